#!/bin/bash
# r05 GPU job 27: last soak of the round on the final library (single-frame sweeps, full-size cases, multi-frame calls, entry points), native backtraces armed
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05z; mkdir -p $OUT/traces
MDVT_SEGV_TRACE=1 MDVT_SEGV_TRACE_DIR=$ROOT/$OUT/traces python tools/soak.py --tag r05e --commit ${SOAK_COMMIT:-unknown} --seed0 574000 --seeds 1200 --cases 400 --full 240 \
     --aux-seeds 40 --aux-cases 250 --batch-seeds 1000 --batch-cases 100 --finish 1 --procs 14 --budget-min 24 > $OUT/soak.log 2>&1
tail -14 gpurun_out/soak_r05e/summary.md | cut -c1-300
find gpurun_out/soak_r05e -name "*.log" -size -3k -delete
find $OUT/traces -size 0 -delete; ls $OUT/traces | head
