#!/bin/bash
# r06 GPU job 3: the whole GPU suite on the default grid (incl. the GL fixtures through the 4-bit kernels), then the render tests once more
# with every renderer on the 4-bit grid (MDVT_TEST_SUBPIXEL_BITS=4: the second copy of the rasterising kernels against the oracle)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -12 $OUT/pytest_all.log
MDVT_TEST_SUBPIXEL_BITS=4 timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_bench_sizes.py -q -m gpu > $OUT/pytest_grid4.log 2>&1; tail -12 $OUT/pytest_grid4.log
