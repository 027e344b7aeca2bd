#!/usr/bin/env python3
"""BASELINE config C3's shape end to end through the clip driver: N synthetic 1080p frames from .npy frame
dumps on disk -> pinned double-buffered H2D -> render -> D2H -> .npy dumps (host I/O included).
usage: python tools/clip_c3.py [--frames 300] [--dir /tmp/c3] [--mesh] [--infill] [--normal_infill | --basic]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metric_depth_video_toolbox_amd import clip
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--dir", default="/tmp/c3")
ap.add_argument("--mesh", action="store_true")
ap.add_argument("--infill", action="store_true")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--normal_infill", action="store_true", help="with --mesh --infill: also write <output>_infilled.npy (basic_nomal_infill.py's step)")
ap.add_argument("--basic", action="store_true", help="with --mesh --infill: --do_basic_infill")
a = ap.parse_args()
os.makedirs(a.dir, exist_ok=True)
W, H, N = 1920, 1080, a.frames
dp, cp = os.path.join(a.dir, "clip_depth.npy"), os.path.join(a.dir, "clip_color.npy")
if int(os.environ.get("RANK", "0")) == 0 and not os.path.exists(dp):
    sc = SyntheticScene(W, H, config_id=3)
    d = np.lib.format.open_memmap(dp, mode="w+", dtype=np.uint8, shape=(N, H, W, 3))
    c = np.lib.format.open_memmap(cp, mode="w+", dtype=np.uint8, shape=(N, H, W, 3))
    base_d, base_c = sc.clip(8)
    for t in range(N):          # 8 distinct frames cycled: generation time is not what is measured
        d[t], c[t] = base_d[t % 8], base_c[t % 8]
    d.flush(); c.flush()
t0 = time.perf_counter()
stats, final = clip.run(dp, cp, batch=a.batch, xfov=45.0, pupillary_distance=65, render_as_pointcloud=not a.mesh,
                        infill_mask=a.infill, normal_infill=a.normal_infill, do_basic_infill=a.basic)
dt = time.perf_counter() - t0
if int(os.environ.get("RANK", "0")) == 0:
    print(f"C3 clip: {N} frames 1080p, {'mesh' if a.mesh else 'points'}{'+infill' if a.infill else ''}{'+normal_infill' if a.normal_infill else ''}{'+basic' if a.basic else ''}: "
          f"{stats[:, 0].sum() / stats[:, 1].max():.1f} frames/s in the render loop incl. host copies and PCIe, "
          f"{N / dt:.1f} frames/s wall incl. output file creation; steady state after the first batch "
          f"{getattr(clip.render_clip, 'last_steady_fps', float('nan')):.1f} frames/s; holes {int(stats[:, 2].sum())}")
