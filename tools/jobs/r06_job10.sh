#!/bin/bash
# r06 GPU job 10: the new tests (bank crowd, 7-frame 4K split, asynchronous completion), profiles of the fused points variant
# (kernel trace + FETCH / WRITE passes, product library), then a soak of the round's library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06j; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_fresh_context.py tests/test_gpu_bench_sizes.py tests/test_gpu_render.py -x -q -m gpu -k "crowd or c4 or host_wait or fused" > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
MDVT_LIB_VARIANT= bash tools/profile_kbench.sh r06_points_fused --bits --counts --frames 128 > $OUT/prof_fused.log 2>&1; tail -2 $OUT/prof_fused.log
MDVT_LIB_VARIANT= bash tools/profile_kbench.sh r06_points_fused_nomask --bits --counts --nomask --frames 128 > $OUT/prof_fused_nomask.log 2>&1; tail -2 $OUT/prof_fused_nomask.log
