#!/usr/bin/env python3
"""VERDICT r05 item 3, the part that can be settled without a device: cv2.inpaint's heap order (orc_telea_fmm) reproduced BIT FOR BIT
by a conservative parallel simulation (orc_telea_windows: windows of 0.70 in T -- FastMarching_solve adds at least 1/sqrt(2) -- with
sorted pops, order-free activation, dependency-ordered estimates), and what its structure would cost on a device: windows (each a
sort + an activation pass + a dependency walk), pops per window (the sort's size), and the length of the dependency chains inside
the windows -- the number of steps that cannot overlap.  Product-default seed images (mesh, --infill_mask, 65 mm, xfov 45, with and
without a 2.5 m convergence) rendered by the oracle.  CPU only, oracle only.
usage: python tests/report_infill_heap_windows.py [--size 1920x1080] [--frames 1] [--out profiles/r06_infill_heap_windows.md]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as orc
from oracle import oracle_np as onp
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
from metric_depth_video_toolbox_amd.depth_map_tools import compute_camera_matrix

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    K = compute_camera_matrix(45.0, None, W, H)
    rows = []
    t0 = time.time()
    for k in range(a.frames):
        for conv in (2.5, 0.0):
            d, c = SyntheticScene(W, H, config_id=2).frame(3 * k)
            ang = onp.convergence_angle(conv, 0.065) if conv else 0.0
            p = orc.make_params(W, H, K, ipd_m=0.065, mode=orc.MODE_MESH, remove_edges=True, edge_points=1, key_rgb=(0, 255, 0), conv_angle=ang)
            r = orc.render_stereo(p, d, c, want_seed=True)
            for eye in ("left", "right"):
                seed = r[f"{eye}_seed"]
                green = np.all(seed == (0, 255, 0), -1)
                mask = (green | np.all(seed == 0, -1)).astype(np.uint8)
                t = time.time(); ref = orc.telea_fmm(seed, mask); t_f = time.time() - t
                t = time.time(); got, T, st = orc.telea_windows(seed, mask); t_w = time.time() - t
                lev, _ = orc.telea_levels(seed, mask, must_fill=green.astype(np.uint8))
                rows.append(dict(frame=3 * k, conv=conv, eye=eye, equal=bool(np.array_equal(ref, got)), diff=int((ref != got).any(-1).sum()),
                                 inside=int(mask.sum()), green=int(green.sum()), t_fmm=t_f, t_win=t_w,
                                 levels_differ=int(((lev != ref).any(-1) & green).sum()), **st))
                print(rows[-1], flush=True)
    hdr = ("| frame | view | eye | = heap order | pixels to fill (key-coloured) | windows | pops: all / most in a window | most activations in a window | "
           "dependent steps, T (sum over windows of the longest 4-neighbour chain) | dependent steps, colour (distance <= 4) | longest colour chain in one window | "
           "look-ahead violations | key-coloured px the level order gets differently |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    body = "".join(f"| {r['frame']} | {'toe-in 2.5 m' if r['conv'] else 'pure shift'} | {r['eye']} | {'yes' if r['equal'] else 'NO: %d px' % r['diff']} | "
                   f"{r['inside']} ({r['green']}) | {r['windows']} | {r['pops']} / {r['max_pops_per_window']} | {r['max_activations_per_window']} | {r['sum_T_chain']} | "
                   f"{r['sum_colour_chain']} | {r['max_colour_chain_in_a_window']} | {r['lookahead_violations']} | {r['levels_differ']} |\n" for r in rows)
    text = (f"`python tests/report_infill_heap_windows.py --size {W}x{H} --frames {a.frames}` (CPU, oracle only, {time.time() - t0:.0f} s).\n\n" + hdr + body)
    print(text)
    if a.out:
        open(a.out, "w").write(text)
