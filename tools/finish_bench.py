#!/usr/bin/env python3
"""Timing of the infill-mask completion (mdvt_finish_infill_mask) on real seeds: N 1080p frames rendered in the
product-default mode (mesh, --infill_mask), both eyes finished.  Prints per-stage times."""
import argparse
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--rounds", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--config", type=int, default=3, help="synthetic scene id (bench.py uses 2)")
ap.add_argument("--conv", type=float, default=0.0, help="convergence distance in metres (0: none)")
ap.add_argument("--per-eye", action="store_true", help="one call per eye instead of the stereo entry point")
ap.add_argument("--streams", type=int, default=1, help="split the frames over this many contexts, each on its own stream")
a = ap.parse_args()
W, H, N = a.width, a.height, a.frames
d, c = synthetic.SyntheticScene(W, H, config_id=a.config).clip(N)
r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)
p = r.frame_params(xfov=45.0, **({'convergence_distance': a.conv} if a.conv > 0 else {}))
res = r.render(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), p, want_seed=True)
seed = res["seed"]
out = torch.empty_like(seed)
streams = [torch.cuda.Stream() for _ in range(a.streams)]
rs = [sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True) for _ in range(a.streams)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rep in range(a.reps):
    torch.cuda.synchronize()
    ev[0].record()
    if a.streams > 1:
        cur = torch.cuda.current_stream()
        per = (N + a.streams - 1) // a.streams
        for k in range(a.streams):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                rs[k].finish_infill_mask_sbs(seed[k * per:(k + 1) * per], out=out[k * per:(k + 1) * per], max_rounds=a.rounds)
        for k in range(a.streams):
            cur.wait_stream(streams[k])
        rem = torch.zeros(1)
    elif a.per_eye:
        for eye in range(2):
            _, rem = r.finish_infill_mask(seed[:, :, eye * W:(eye + 1) * W], out=out[:, :, eye * W:(eye + 1) * W], max_rounds=a.rounds,
                                          want_remaining=True)
    else:
        _, rem = r.finish_infill_mask_sbs(seed, out=out, max_rounds=a.rounds, want_remaining=True)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    print(f"finish {2 * N} images {W}x{H}: {ms:.2f} ms  -> {ms / N:.3f} ms/frame, remaining {int(rem.sum())}, "
          f"holes {int((res['mask'] > 0).sum()) / (2 * N * W * H) * 100:.1f} %")
