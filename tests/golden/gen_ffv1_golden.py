#!/usr/bin/env python3
"""Cross-check vectors for libmdvt_video.so against FFmpeg ITSELF -- for a machine that has one (the build container and the GPU box
do not: interoperability is unpinned here).

    python tests/golden/gen_ffv1_golden.py --ffmpeg /usr/bin/ffmpeg          # writes tests/golden/ffv1_ffmpeg_*.mkv + ffv1_ffmpeg.npz

What it makes, from a small deterministic clip (no reference code involved):
  * ffv1_ffmpeg_<mode>.mkv: the clip encoded by FFmpeg as bgr0 FFV1 in Matroska in the modes OpenCV / the toolbox's tools produce
    (`-c:v ffv1` defaults = version 3, Golomb-Rice, 4 slices, g 12; `-level 1`; `-coder 1`; `-coder 2`; `-g 1`; bgra) -- what
    tests/test_video_cpu.py::test_reader_against_ffmpeg_files then decodes with the build's reader and compares with the frames
    in ffv1_ffmpeg.npz;
  * the other direction: the build's writer's file decoded by FFmpeg (`-f rawvideo -pix_fmt rgb24`) must equal the frames; the
    script checks that on the spot and records the verdict in the .npz.
"""
import argparse
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

MODES = {"default": [], "level1": ["-level", "1"], "coder1": ["-coder", "1"], "coder2": ["-coder", "2"], "intra": ["-g", "1"],
         "slices16": ["-slices", "16", "-slicecrc", "1"], "context1": ["-context", "1"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ffmpeg", required=True)
    args = ap.parse_args()
    from metric_depth_video_toolbox_amd import video_io
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
    W, H, N = 96, 54, 14
    d, c = SyntheticScene(W, H, seed=5, n_fg=4).clip(N)
    frames = np.stack([d[k] if k % 2 else c[k] for k in range(N)])
    raw = os.path.join(HERE, "_ffv1_raw.rgb")
    frames.tofile(raw)
    base = [args.ffmpeg, "-y", "-v", "error", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{W}x{H}", "-r", "24", "-i", raw]
    for name, extra in MODES.items():
        subprocess.check_call(base + ["-pix_fmt", "bgr0", "-c:v", "ffv1"] + extra + [os.path.join(HERE, f"ffv1_ffmpeg_{name}.mkv")])
    subprocess.check_call(base + ["-pix_fmt", "bgra", "-c:v", "ffv1", os.path.join(HERE, "ffv1_ffmpeg_alpha.mkv")])
    ours = os.path.join(HERE, "_ffv1_ours.mkv")
    with video_io.VideoWriter(ours, W, H, 24) as w:
        for f in frames:
            w.write(f)
    back = subprocess.check_output([args.ffmpeg, "-v", "error", "-i", ours, "-f", "rawvideo", "-pix_fmt", "rgb24", "-"])
    ok = np.array_equal(np.frombuffer(back, np.uint8).reshape(frames.shape), frames)
    print("FFmpeg decodes the build's file:", "bit-exact" if ok else "MISMATCH")
    np.savez_compressed(os.path.join(HERE, "ffv1_ffmpeg.npz"), frames=frames, ffmpeg_reads_ours=np.bool_(ok),
                        ffmpeg=np.array(subprocess.check_output([args.ffmpeg, "-version"]).decode().splitlines()[0]))
    os.remove(raw); os.remove(ours)


if __name__ == "__main__":
    main()
