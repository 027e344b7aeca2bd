#!/usr/bin/env python3
"""In-process A/B timing of kernel configurations (env MDVT_POINTS_CFG is re-read per launch).
usage: python tools/kbench.py cfgA cfgB ... [--rounds R] [--calls C] [--mode points|mesh]"""
import os, sys, argparse, statistics
os.environ.setdefault("MDVT_LIB_VARIANT", "tuning")      # the hooks this tool drives live in the tuning build (csrc/mdvt_internal.h);
                                                         # MDVT_LIB_VARIANT= (empty) times the product library
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get('KB_DIST'):
    import torch.distributed as dist
    print('dist imported', dist.is_available())
if os.environ.get('KB_SETDEV'):
    torch.cuda.set_device(0)
from metric_depth_video_toolbox_amd.stereo_rerender import StereoRerenderer
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene

ap = argparse.ArgumentParser()
ap.add_argument("cfgs", nargs="+")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--calls", type=int, default=10)
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--env", default="MDVT_POINTS_CFG")
ap.add_argument("--mesh", action="store_true")
ap.add_argument("--infill", action="store_true")
ap.add_argument("--zout", action="store_true")
ap.add_argument("--conv", type=float, default=None)
ap.add_argument("--pose", action="store_true")
ap.add_argument("--c4", action="store_true", help="BASELINE configs[3] as bench.py's extra.c4_* runs it: scene 4 + contention band + frames 40, 70, ... of the pose track (use with --width 3840 --height 2160 --frames 8)")
ap.add_argument("--bits", action="store_true")
ap.add_argument("--counts", action="store_true")
ap.add_argument("--nomask", action="store_true", help="with --bits: no byte mask (want_mask=False)")
ap.add_argument("--edges", action="store_true", help="remove_edges without --infill_mask (edge points painted, no seed image)")
ap.add_argument("--split", type=int, default=1, help="the batch as this many sub-batches, each with its own context on its own stream")
ap.add_argument("--stagger-us", type=float, default=0.0, help="with --split: sub-batch k starts k x this late and the streams free-run (joined once per timing)")
ap.add_argument("--noedgepts", action="store_true", help="with --edges / --infill: dont_place_points_in_edges")
a = ap.parse_args()
W, H, N = a.width, a.height, a.frames
if os.environ.get('KB_ORDER'):
    r = StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=not a.mesh, infill_mask=a.infill, remove_edges=a.edges or a.infill, dont_place_points_in_edges=a.noedgepts)
    d, c = SyntheticScene(W, H, config_id=2).clip(N)
    d, c = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
else:
    d, c = SyntheticScene(W, H, config_id=2).clip(N)
    d, c = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
    r = StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=not a.mesh, infill_mask=a.infill, remove_edges=a.edges or a.infill, dont_place_points_in_edges=a.noedgepts)
from metric_depth_video_toolbox_amd.synthetic import synthetic_pose_track
Ts = synthetic_pose_track(N) if a.pose else [None] * N
if a.c4:
    from metric_depth_video_toolbox_amd.synthetic import c4_clip
    d4, c4, Ts = c4_clip(N, W, H)          # (the clip bench.py times and tests/test_gpu_bench_sizes.py holds to the oracle)
    d, c, Ts = torch.from_numpy(d4).cuda(), torch.from_numpy(c4).cuda(), list(Ts)
p = [r.frame_params(xfov=45.0, convergence_distance=a.conv, transformation=Ts[k]) for k in range(N)]
sbs = torch.empty((N, H, 2 * W, 3), dtype=torch.uint8, device="cuda")
mask = torch.empty((N, H, 2 * W), dtype=torch.uint8, device="cuda")
zo = torch.empty((N, H, 2 * W), dtype=torch.float32, device="cuda") if a.zout else None
job = r.prepare(d, c, p, out_sbs=sbs, out_mask=mask, want_depth=a.zout, out_depth=zo, want_maskbits=a.bits, want_hole_counts=a.counts, want_mask=not a.nomask)
stream = torch.cuda.current_stream()
if a.split > 1:
    per = (N + a.split - 1) // a.split
    rs = [StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=not a.mesh, infill_mask=a.infill, remove_edges=a.edges or a.infill,
                           dont_place_points_in_edges=a.noedgepts) for _ in range(a.split)]
    streams = [torch.cuda.Stream() for _ in range(a.split)]
    jobs = [rs[k].prepare(d[k * per:(k + 1) * per], c[k * per:(k + 1) * per], p[k * per:(k + 1) * per], out_sbs=sbs[k * per:(k + 1) * per],
                          out_mask=mask[k * per:(k + 1) * per], want_depth=a.zout, out_depth=None if zo is None else zo[k * per:(k + 1) * per],
                          want_maskbits=a.bits, want_hole_counts=a.counts, want_mask=not a.nomask) for k in range(a.split)]
def run():
    if a.split > 1:
        for k in range(a.split):
            streams[k].wait_stream(stream)
            jobs[k].launch(streams[k])
        for k in range(a.split):
            stream.wait_stream(streams[k])
        return
    job.launch(stream)
res = {k: [] for k in a.cfgs}
for k in a.cfgs:
    os.environ[a.env] = '' if k == 'default' else k; run()
torch.cuda.synchronize()
for _ in range(a.rounds):
    for k in a.cfgs:
        os.environ[a.env] = '' if k == 'default' else k
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if a.split > 1 and a.stagger_us > 0:
            for j in range(a.split):
                streams[j].wait_stream(stream)
                with torch.cuda.stream(streams[j]):
                    if j: torch.cuda._sleep(int(j * a.stagger_us * 2400))
                for _ in range(a.calls): jobs[j].launch(streams[j])
            for j in range(a.split): stream.wait_stream(streams[j])
        else:
            for _ in range(a.calls): run()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / a.calls * 1e3)
print("library:", "libmdvt_hip_tuning.so" if os.environ.get("MDVT_LIB_VARIANT") else "libmdvt_hip.so (product)")
bpp = 14 + (8 if a.zout else 0)
for k in a.cfgs:
    med = statistics.median(res[k])
    print(f"{k:>10s}: median {med:7.1f} us  min {min(res[k]):7.1f}  ->  {bpp*W*H*N/med/1e6:.2f} TB/s  ({N*1e6/med:.0f} fps)")
