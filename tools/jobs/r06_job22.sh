#!/bin/bash
# r06 GPU job 22: soak of the SECOND copy of the rasterising kernels -- every renderer on the 4-bit sub-pixel grid (MDVT_TEST_SUBPIXEL_BITS=4:
# mdvt::grid4, the grid of the pinned GL), oracle on the same grid
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06s; mkdir -p $OUT/traces
MDVT_TEST_SUBPIXEL_BITS=4 MDVT_SEGV_TRACE=1 MDVT_SEGV_TRACE_DIR=$ROOT/$OUT/traces python tools/soak.py --tag r06_grid4 --commit ${SOAK_COMMIT:-unknown} --seed0 646000 --seeds 600 --cases 400 --full 60 \
     --aux-seeds 0 --batch-seeds 500 --batch-cases 100 --finish 0 --procs 14 --budget-min 12 > $OUT/soak.log 2>&1
tail -12 gpurun_out/soak_r06_grid4/summary.md | cut -c1-250
find gpurun_out/soak_r06_grid4 -name "*.log" -size -3k -delete
