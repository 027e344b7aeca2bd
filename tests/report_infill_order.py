#!/usr/bin/env python3
"""How far the level-synchronous infill order (the device's, = orc_telea_levels) is from the sequential fast-marching order
of cv2.inpaint (restated as orc_telea_fmm; same estimator, same decrees) on rendered seed images.  CPU only (oracle).
usage: python tests/report_infill_order.py [--out profiles/r02_infill_order_vs_fmm.md]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as orc
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
from metric_depth_video_toolbox_amd.depth_map_tools import compute_camera_matrix

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
a = ap.parse_args()
rows = []
for (W, H) in ((480, 270), (960, 540)):
    for t in (0, 3):
        d, c = SyntheticScene(W, H, config_id=3).frame(t)
        p = orc.make_params(W, H, compute_camera_matrix(45.0, None, W, H), ipd_m=0.065, mode=orc.MODE_MESH, remove_edges=True,
                            edge_points=1, key_rgb=(0, 255, 0))
        r = orc.render_stereo(p, d, c, want_seed=True)
        for eye in ("left", "right"):
            seed = r[f"{eye}_seed"]
            green = np.all(seed == (0, 255, 0), -1)
            mask = (green | np.all(seed == 0, -1)).astype(np.uint8)
            lev, rem = orc.telea_levels(seed, mask, must_fill=green.astype(np.uint8))
            fmm = orc.telea_fmm(seed, mask)
            diff = np.abs(lev.astype(int) - fmm.astype(int))[green]
            rg = diff[:, :2].max(-1)          # the channels infill_common.py reads as a direction
            allc = diff.max(-1)
            rows.append((f"{W}x{H}", t, eye, int(green.sum()), allc.mean(), np.percentile(allc, 50), np.percentile(allc, 90),
                         np.percentile(allc, 99), (allc <= 2).mean(), rg.mean(), np.percentile(rg, 99)))
hdr = ("| frame size | frame | eye | hole px | mean | p50 | p90 | p99 | share within 2 LSB | rg mean | rg p99 |\n"
       "|---|---|---|---|---|---|---|---|---|---|---|\n")
body = "".join(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]:.2f} | {r[5]:.0f} | {r[6]:.0f} | {r[7]:.0f} | {r[8]:.2f} | {r[9]:.2f} | {r[10]:.0f} |\n" for r in rows)
text = ("# Infill-mask completion: level-synchronous order vs sequential fast marching\n\n"
        "`python tests/report_infill_order.py` (CPU, oracle only).  Per hole pixel, max over channels of |orc_telea_levels - orc_telea_fmm| in LSB:\n"
        "the device's order (breadth-first levels = L1 distance to the nearest known pixel) against the heap order of `cv2.inpaint`\n"
        "(restated; OpenCV itself is not installed), same estimator and decrees.  Seeds: product-default render (mesh, edge removal,\n"
        "edge points, green key) of the synthetic clip.\n\n" + hdr + body)
print(text)
if a.out:
    open(a.out, "w").write(text)
