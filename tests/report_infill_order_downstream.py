#!/usr/bin/env python3
"""What the level-synchronous order of the infill-mask completion costs DOWNSTREAM: product-default frames
(mesh, --infill_mask, convergence) are rendered with the oracle, their infill mask is finished twice -- with the device's order
(orc_telea_levels) and with cv2.inpaint's sequential fast-marching order (orc_telea_fmm, the same estimator and decrees) --, both
go through sr:807-808 + masked_blur, and basic_nomal_infill.normal_infill (orc_normal_infill) runs with each.  Reported: the angle
between the two masks' directions (the r, g channels infill_common.py:10 reads) over the hole pixels, and how many pixels of the
final image differ and by how much.  CPU only.
usage: python tests/report_infill_order_downstream.py [--size 1920x1080] [--frames 2] [--out profiles/r04_infill_order_downstream.md]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as orc
from oracle import oracle_np as onp
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
from metric_depth_video_toolbox_amd.depth_map_tools import compute_camera_matrix


def finish_with(seed, order):
    """sr:803-808 + 816 with the chosen inpaint order -> the u8 infill mask written to the video."""
    green = np.all(seed == (0, 255, 0), -1)
    mask = (green | np.all(seed == 0, -1)).astype(np.uint8)
    if order == "levels":
        filled, _ = orc.telea_levels(seed, mask, must_fill=green.astype(np.uint8))
    else:
        filled = orc.telea_fmm(seed, mask)
    merged = seed.copy()
    merged[green] = filled[green]                 # (the u8 -> float -> u8 round trips of sr:807-808, 816 are the identity: tested)
    return orc.masked_blur(merged), green


def direction_angle_deg(m0, m1, sel):
    d0 = (m0[sel][:, :2].astype(np.float64) / 255.0) * 2 - 1
    d1 = (m1[sel][:, :2].astype(np.float64) / 255.0) * 2 - 1
    n0, n1 = np.linalg.norm(d0, axis=1), np.linalg.norm(d1, axis=1)
    ok = (n0 > 1e-6) & (n1 > 1e-6)
    c = np.clip((d0[ok] * d1[ok]).sum(1) / (n0[ok] * n1[ok]), -1, 1)
    return np.degrees(np.arccos(c))


def one_frame(W, H, cfg, t, conv):
    d, c = SyntheticScene(W, H, config_id=cfg).frame(t)
    K = compute_camera_matrix(45.0, None, W, H)
    ang = onp.convergence_angle(conv, 0.065) if conv else 0.0
    p = orc.make_params(W, H, K, ipd_m=0.065, mode=orc.MODE_MESH, remove_edges=True, edge_points=1, key_rgb=(0, 255, 0), conv_angle=ang)
    r = orc.render_stereo(p, d, c, want_seed=True)
    rows = []
    for eye in ("left", "right"):
        seed, img, hole = r[f"{eye}_seed"], r[f"{eye}_rgb"], r[f"{eye}_mask"] > 0
        m_lev, green = finish_with(seed, "levels")
        m_fmm, _ = finish_with(seed, "fmm")
        a = direction_angle_deg(m_lev, m_fmm, green)
        o_lev, o_fmm = orc.normal_infill(img, m_lev), orc.normal_infill(img, m_fmm)
        dif = np.abs(o_lev.astype(int) - o_fmm.astype(int)).max(-1)
        mdiff = np.abs(m_lev.astype(int) - m_fmm.astype(int)).max(-1)[green]
        rows.append(dict(eye=eye, t=t, holes=int(green.sum()), a_mean=a.mean(), a50=np.percentile(a, 50), a90=np.percentile(a, 90),
                         a99=np.percentile(a, 99), a_max=a.max(), mask_p99=np.percentile(mdiff, 99),
                         px_diff=float((dif > 0).mean()), px_diff_holes=float((dif[hole] > 0).mean()) if hole.any() else 0.0,
                         px_diff_gt8=float((dif > 8).mean()), d_max=int(dif.max()), d_mean_where=float(dif[dif > 0].mean()) if (dif > 0).any() else 0.0))
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    rows = []
    for k in range(a.frames):
        rows += [dict(r, conv=cv) for cv in (2.5, 0.0) for r in one_frame(W, H, 2, 3 * k, cv)]
    hdr = ("| view | frame | eye | hole px | direction angle (deg): mean | p50 | p90 | p99 | max | mask p99 (LSB) | final px that differ | of the hole px | differ by > 8 LSB | max | mean where they differ |\n"
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    body = "".join(f"| {'converged 2.5 m' if r['conv'] else 'pure shift'} | {r['t']} | {r['eye']} | {r['holes']} | {r['a_mean']:.2f} | {r['a50']:.2f} | {r['a90']:.2f} | "
                   f"{r['a99']:.1f} | {r['a_max']:.0f} | {r['mask_p99']:.0f} | {100 * r['px_diff']:.3f} % | {100 * r['px_diff_holes']:.1f} % | {100 * r['px_diff_gt8']:.3f} % | "
                   f"{r['d_max']} | {r['d_mean_where']:.1f} |\n" for r in rows)
    text = ("# What the level-synchronous inpaint order costs downstream (verdict r03 item 5b)\n\n"
            f"`python tests/report_infill_order_downstream.py --size {a.size} --frames {a.frames}` (CPU, oracle only).  Product-default frames of the synthetic clip\n"
            "(mesh, `--infill_mask`, 65 mm, xfov 45; with and without a 2.5 m convergence) rendered by the oracle; the infill mask finished with the\n"
            "device's order (`orc_telea_levels`) and with `cv2.inpaint`'s sequential order (`orc_telea_fmm`: the same estimator, restated -- OpenCV\n"
            "is absent), sr:807-808 and `masked_blur` after either; then `basic_nomal_infill.normal_infill` (`orc_normal_infill`) with each mask.\n"
            "Direction = the (r, g) channels as `infill_common.py:10` reads them; angle over the key-coloured (hole) pixels.  Final image: the\n"
            "infilled eye, max over channels of the difference.\n\n" + hdr + body +
            """
Reading: the order moves a hole pixel's direction by less than a degree in the median and by 3-5 degrees at the 90th percentile;
the last per cent (where two fronts meet, and at the rims the heap order reaches diagonally first) turns by 50-80 degrees.  Because
`normal_infill` marches along that direction to pick a source pixel, 40-65 % of the hole pixels end up with another sample of the same
neighbourhood -- and these synthetic colour frames carry +-64 LSB of per-pixel noise, so ANY other sample shows as a large difference
(on smooth content it is the local gradient times the displacement).  2-3 % of a frame's pixels change, all of them hole pixels or their
6-px blur band.  A maintainer who needs cv2.inpaint's exact directions cannot have them from a level-synchronous front; one who needs
'a direction pointing out of the hole' (what infill_common.py and the infill models consume) gets it to within a few degrees for 90 % of the pixels.
""")
    print(text)
    if a.out:
        open(a.out, "w").write(text)
