#!/bin/bash
# r06 GPU job 19: third soak of the round's last library (fresh seeds, more multi-frame calls and full-size cases)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06r; mkdir -p $OUT/traces
MDVT_SEGV_TRACE=1 MDVT_SEGV_TRACE_DIR=$ROOT/$OUT/traces python tools/soak.py --tag r06c --commit ${SOAK_COMMIT:-unknown} --seed0 626000 --seeds 1800 --cases 400 --full 120 \
     --aux-seeds 80 --aux-cases 250 --batch-seeds 1600 --batch-cases 100 --finish 2 --procs 14 --budget-min 38 > $OUT/soak.log 2>&1
tail -12 gpurun_out/soak_r06c/summary.md | cut -c1-250
find gpurun_out/soak_r06c -name "*.log" -size -3k -delete
find $OUT/traces -size 0 -delete; ls $OUT/traces | head -3
