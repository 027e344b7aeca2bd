#!/bin/bash
# r06 GPU job 1: the render tests against the oracle after the fill-rule / point-rule change (GL convention), + bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06a; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_bench_sizes.py tests/test_gpu_edge_points.py -x -q -m gpu > $OUT/pytest_render.log 2>&1
tail -15 $OUT/pytest_render.log
timeout 600 python bench.py > $OUT/bench.log 2>&1; tail -3 $OUT/bench.log | cut -c1-1500
