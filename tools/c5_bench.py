#!/usr/bin/env python3
"""BASELINE config C5: Depth-Anything-V2 (PyTorch-ROCm) -> on-device 16-bit quantisation -> HIP reproject kernel,
one stream, 1080p.  Random-init weights (no checkpoint available); timing only.
usage: python tools/c5_bench.py [--encoder vits|vitb|vitl] [--frames 4] [--iters 10] [--bf16]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metric_depth_video_toolbox_amd import model_hop, stereo_rerender as sr, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="vits")
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--bf16", action="store_true")
a = ap.parse_args()
W, H, N = 1920, 1080, a.frames
_, color = synthetic.SyntheticScene(W, H, config_id=5).clip(N)
color_t = torch.from_numpy(color).cuda()
model = model_hop.build_depth_anything_v2(a.encoder, max_depth=20, seed=0)
r = sr.StereoRerenderer(W, H, pupillary_distance=65, max_depth=20, render_as_pointcloud=True)
p = r.frame_params(xfov=45.0)
ac = torch.bfloat16 if a.bf16 else None
def step():
    return model_hop.color_to_stereo(model, color_t, r, p, input_height=518, autocast_dtype=ac)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
t0 = time.perf_counter(); e0.record()
for _ in range(a.iters): step()
e1.record(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
# the model alone
e1.record()
for _ in range(a.iters): model_hop.infer_depth(model, color_t, 518, ac)
e2.record(); torch.cuda.synchronize()
tot, mod = e0.elapsed_time(e1) / a.iters / N, e1.elapsed_time(e2) / a.iters / N
print(f"C5 {a.encoder}{' bf16' if a.bf16 else ' fp32'}: {N / (dt / a.iters):.1f} stereo frames/s end to end on one stream "
      f"({tot:.2f} ms/frame, of which depth model {mod:.2f} ms, quantise + reproject {tot - mod:.3f} ms)")
