#!/bin/bash
# r06 GPU job 4: clip tests incl. the Matroska / FFV1 end-to-end CLI test; codec throughput on the GPU box's host cores; CLI from .mkv timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_clip.py tests/test_video_cpu.py -x -q > $OUT/pytest_clip.log 2>&1; tail -8 $OUT/pytest_clip.log
python - > $OUT/codec.log 2>&1 <<'PY'
import numpy as np, time, os, sys
sys.path.insert(0, os.getcwd())
from metric_depth_video_toolbox_amd import video_io as v, clip
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
W, H, N = 1920, 1080, 48
print("usable cores:", clip._usable_cores(), "os.cpu_count:", os.cpu_count())
d, c = SyntheticScene(W, H, config_id=2).clip(8)
for name, fr in (("depth-coded", d), ("noise colour", c)):
    p = f"/tmp/t_{name[:5]}.mkv"
    for th in (1, 4, 16):
        t = time.time()
        with v.VideoWriter(p, W, H, 24, threads=th) as w:
            for k in range(N): w.write(fr[k % 8])
        te = time.time() - t
        r = v.VideoReader(p, threads=th); out = np.empty((H, W, 3), np.uint8); t = time.time(); n = 0
        while r.read_into(out): n += 1
        td = time.time() - t
        print(f"{name:13s} threads {th:2d}: encode {N/te:6.1f} fps, decode {n/td:6.1f} fps, {os.path.getsize(p)/N/1e6:.2f} MB/frame (raw 6.22)")
# clip-level: sink with 12 store threads
sink = clip.VideoSink("/tmp/s.mkv", 2*W, H, 24.0)
from concurrent.futures import ThreadPoolExecutor
sbs = np.concatenate([c, c], axis=2)
t = time.time()
with ThreadPoolExecutor(12) as ex:
    jobs = [ex.submit(sink.write_from, sbs[k % 8:k % 8 + 1], k, 1) for k in range(N)]
    [j.result() for j in jobs]
sink.close(); print(f"VideoSink 3840x1080 noise, 12 threads: {N/(time.time()-t):.1f} fps")
PY
cat $OUT/codec.log
# the CLI from Matroska files, 1080p product default, 48 frames
python - > $OUT/cli_mkv.log 2>&1 <<'PY'
import numpy as np, time, os, sys
sys.path.insert(0, os.getcwd())
from metric_depth_video_toolbox_amd import video_io as v, stereo_rerender as sr
from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
W, H, N = 1920, 1080, 48
sc = SyntheticScene(W, H, config_id=2)
with v.VideoWriter("/tmp/in_depth.mkv", W, H, 24) as wd, v.VideoWriter("/tmp/in_color.mkv", W, H, 24) as wc:
    for k in range(N):
        d, c = sc.frame(k); wd.write(d); wc.write(c)
for rep in range(2):
    t = time.time()
    sr.main(["--depth_video", "/tmp/in_depth.mkv", "--color_video", "/tmp/in_color.mkv", "--xfov", "45", "--infill_mask", "--batch", "16"])
    print(f"CLI from / to Matroska, product default, {N} frames of 1080p: {N/(time.time()-t):.2f} fps end to end (rep {rep})")
PY
tail -4 $OUT/cli_mkv.log
