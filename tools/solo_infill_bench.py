#!/usr/bin/env python3
"""Per-image time of the two stand-alone marching entry points (mdvt_infill_using_normals with float normals,
mdvt_mark_lower_side) on a 1080p product-default eye.  usage: python tools/solo_infill_bench.py"""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metric_depth_video_toolbox_amd import stereo_rerender as sr, infill_common, synthetic
W, H = 1920, 1080
sc = synthetic.SyntheticScene(W, H, config_id=2)
d, c = sc.frame(0)
r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)
res = r.render(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), r.frame_params(xfov=45.0, convergence_distance=2.5), want_seed=True)
fin = r.finish_infill_mask_sbs(res["seed"])
img = res["sbs"][:, :W].contiguous(); hole = (res["mask"][:, :W] > 0).to(torch.uint8).contiguous(); m = fin[:, :W].contiguous()
nrm = ((m.to(torch.float32) / 255.0) * 2 - 1).contiguous()
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("mdvt_infill_using_normals 1080p: %.1f us per image" % t(lambda: sr.infill_using_normals(img, hole, nrm)))
print("mdvt_mark_lower_side 1080p: %.1f us per image" % t(lambda: infill_common.mark_lower_side(m)))
