#!/usr/bin/env python3
"""Time mdvt_normal_infill (basic_nomal_infill.normal_infill on the device) on the masks the product default produces:
1080p synthetic frames -> mesh + --infill_mask (+ optional convergence) render -> finished infill mask -> normal infill of
both eyes.  usage: python tools/normal_infill_bench.py [--frames 8] [--conv 2.5] [--config 2] [--reps 5]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic, basic_nomal_infill as bni

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--conv", type=float, default=0.0)
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--size", default="1920x1080")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
sc = synthetic.SyntheticScene(W, H, config_id=a.config)
d, c = zip(*(sc.frame(t) for t in range(a.frames)))
d, c = torch.from_numpy(np.stack(d)).cuda(), torch.from_numpy(np.stack(c)).cuda()
r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)
recs = [r.frame_params(xfov=45.0, convergence_distance=a.conv if a.conv > 0 else None) for _ in range(a.frames)]
res = r.render(d, c, recs, want_seed=True)
fin = r.finish_infill_mask_sbs(res["seed"])
sbs = res["sbs"]
out = torch.empty_like(sbs)
bg = (fin != 0).all(-1).float().mean().item()
for _ in range(2):
    bni.normal_infill_sbs(sbs, fin, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); bni.normal_infill_sbs(sbs, fin, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts))
print(f"normal_infill {W}x{H} x {a.frames} stereo frames (conv {a.conv}): bg {bg * 100:.2f} % of pixels; "
      f"{ms:.3f} ms per call = {ms / a.frames:.4f} ms per stereo frame ({a.frames / ms * 1e3:.0f} frames/s), "
      f"algorithmic 18 B/px/eye -> {2 * 18 * W * H * a.frames / ms / 1e6:.1f} GB/s")
