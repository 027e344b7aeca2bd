#!/bin/bash
# r06 GPU job 12: whole GPU suite, then the round's soak on the final library (single-frame sweeps incl. packed mask / counts,
# full-size cases, multi-frame calls, stand-alone entry points, the completion at full size), native backtraces armed
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06l; mkdir -p $OUT/traces
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -4 $OUT/pytest_all.log
MDVT_SEGV_TRACE=1 MDVT_SEGV_TRACE_DIR=$ROOT/$OUT/traces python tools/soak.py --tag r06 --commit ${SOAK_COMMIT:-unknown} --seed0 606000 --seeds 700 --cases 400 --full 160 \
     --aux-seeds 30 --aux-cases 250 --batch-seeds 500 --batch-cases 100 --finish 1 --procs 14 --budget-min 20 > $OUT/soak.log 2>&1
tail -14 gpurun_out/soak_r06/summary.md | cut -c1-300
find gpurun_out/soak_r06 -name "*.log" -size -3k -delete
find $OUT/traces -size 0 -delete; ls $OUT/traces | head
