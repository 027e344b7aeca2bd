#!/usr/bin/env python3
"""How much does the open question "does Open3D's legacy renderer cull back faces?"  matter?  The oracle
renders a 1080p synthetic frame with cull = 0 (the default) and cull = 1, for the pure shift, a toe-in and a posed view, with
and without edge removal, and counts the pixels that differ.  CPU only.  usage: python tests/report_cull_risk.py [W H]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as orc
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene, synthetic_pose_track
from metric_depth_video_toolbox_amd.depth_map_tools import compute_camera_matrix

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
K = compute_camera_matrix(45.0, None, W, H)
d, c = SyntheticScene(W, H, config_id=2).frame(0)
views = (("pure shift", {}), ("convergence at 2.5 m", {"conv_angle": math.atan((0.065 / 2) / 2.5)}),
         ("pose (frame 29 of the C4 track)", {"T": synthetic_pose_track(30)[29]}))
for name, kw in views:
    for re in (False, True):
        outs = []
        for cull in (0, 1):
            op = orc.make_params(W, H, K, ipd_m=0.065, max_depth=100.0, depth_scale=1.0, mode=orc.MODE_MESH, remove_edges=re,
                                 edge_points=1 if re else 0, cull=cull, key_rgb=(0, 255, 0) if re else (0, 0, 0), **kw)
            outs.append(orc.render_stereo(op, d, c))
        dm = sum(int((outs[0][e + "_mask"] != outs[1][e + "_mask"]).sum()) for e in ("left", "right"))
        dc = sum(int(np.any(outs[0][e + "_rgb"] != outs[1][e + "_rgb"], -1).sum()) for e in ("left", "right"))
        print(f"{name:34s} remove_edges={re!s:5s}: hole mask differs at {dm} px, colour at {dc} px of {2 * W * H}")
