#!/usr/bin/env python3
"""How often does the decree's f32 evaluation of an edge point's pixel (ex = ((gx - cx) sW) + cx,
u = ex +- dl / Z, row i) differ from the reference's f64 chain for the pure stereo shift?  The reference's chain is NumPy /
Open3D / OpenCV in f64 end to end (dmt:1117-1128 unprojection, sr:599-600 undo of the off-by-one scale, Open3D translate,
cv2.projectPoints with the camera matrix cast to f32, np.round: sr:592-600, 733-746); for the pure shift it is closed form and is
restated here in NumPy f64 (cv2.projectPoints by its published arithmetic: x' = X * (1 / Z), u = x' * fx + cx).
CPU only.  usage: python tests/report_edge_points_f64.py [W H frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
from metric_depth_video_toolbox_amd.depth_map_tools import compute_camera_matrix
from oracle import c_oracle as orc

W, H, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 8)
K = compute_camera_matrix(45.0, None, W, H)
fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
ipd = 0.065
tot = bad_col = bad_row = 0
for cfg in (2, 3):
    sc = SyntheticScene(W, H, config_id=cfg)
    for t in range(N):
        d_rgb, _ = sc.frame(t)
        z = orc.decode_depth(d_rgb, 100.0, 1.0)                       # f32, the reference's decode (pinned by goldens)
        _, unused, _ = orc.edge_filter(z, K, True)                    # vertices of removed triangles (pinned by goldens)
        idx = np.flatnonzero(unused)
        i, j = idx // W, idx % W
        zz = z.reshape(-1)[idx]
        ok = zz > 1e-4
        i, j, zz = i[ok], j[ok], zz[ok]
        gx = (j.astype(np.float32) * np.float32((W + 1) / W)); gy = (i.astype(np.float32) * np.float32((H + 1) / H))
        for sign in (+1.0, -1.0):
            # the decree, f32
            ex = ((gx - np.float32(cx)) * np.float32((W - 1) / W)) + np.float32(cx)
            dl = np.float32(fx * ipd / 2)
            u32 = ex + np.float32(sign) * (dl / zz)
            col32 = np.rint(u32).astype(np.int64)
            # the reference, f64
            Z = zz.astype(np.float64)
            X = (gx.astype(np.float64) - cx) * Z / fx
            Y = (gy.astype(np.float64) - cy) * Z / fy
            X = X * ((W - 1) / W) + sign * (ipd / 2)
            Y = Y * ((H - 1) / H)
            fxr, fyr, cxr, cyr = (float(np.float32(v)) for v in (fx, fy, cx, cy))
            iz = 1.0 / Z
            u64 = (X * iz) * fxr + cxr
            v64 = (Y * iz) * fyr + cyr
            col64, row64 = np.round(u64).astype(np.int64), np.round(v64).astype(np.int64)
            inb = (col64 >= 0) & (col64 < W)
            tot += int(inb.sum())
            bad_col += int(((col32 != col64) & inb).sum())
            bad_row += int(((row64 != i) & inb).sum())
print(f"{W}x{H}, {2 * N} frames, both eyes: {tot} edge points inside the frame; column differs for {bad_col} ({bad_col / max(tot, 1):.2e}), "
      f"row differs from the source row for {bad_row} ({bad_row / max(tot, 1):.2e})")
# where do the rows differ?
W, H = 1920, 1080
K = compute_camera_matrix(45.0, None, W, H)
fy, cy = K[1, 1], K[1, 2]
rng = np.random.default_rng(1)
for i in (0, 1, 2, 539, 540, 541, 1079):
    Z = rng.uniform(0.5, 20.0, 200000).astype(np.float32).astype(np.float64)
    gy = np.float64(np.float32(i) * np.float32((H + 1) / H))
    Y = ((gy - cy) * Z / fy) * ((H - 1) / H)
    v = (Y * (1.0 / Z)) * float(np.float32(fy)) + float(np.float32(cy))
    r = np.round(v).astype(int)
    print("source row", i, "-> rows", dict(zip(*np.unique(r, return_counts=True))), "v range", v.min(), v.max())
