#!/usr/bin/env python3
"""Soak check (not collected by pytest): the infill-mask completion at full size on seeds from real renders -- pure shift and
converged views, both eyes, several frames per pass -- against the oracle, bit for bit.  usage: python tests/dbg_finish_fullsize.py [W H]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
bad = 0
for cfg, conv in ((3, None), (2, 2.5), (4, 1.5)):
    d, c = synthetic.SyntheticScene(W, H, config_id=cfg).clip(3)
    r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)
    p = r.frame_params(xfov=45.0, convergence_distance=conv)
    res = r.render(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), p, want_seed=True)
    seed = res["seed"]
    fin, rem = r.finish_infill_mask_sbs(seed, want_remaining=True)
    fin = fin.cpu().numpy(); seed_np = seed.cpu().numpy()
    for k in range(3):
        for eye, sl in (("L", slice(0, W)), ("R", slice(W, 2 * W))):
            t0 = time.time()
            want, wrem = orc.finish_infill_mask(np.ascontiguousarray(seed_np[k][:, sl]), max_rounds=256)
            ok = np.array_equal(fin[k][:, sl], want)
            bad += not ok
            print(f"config {cfg} conv {conv} frame {k} {eye}: {'OK' if ok else 'MISMATCH'} remaining {wrem} (oracle {time.time() - t0:.1f} s)", flush=True)
    r.close()
print("ALL OK" if not bad else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
