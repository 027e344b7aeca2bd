#!/bin/bash
# r06 GPU job 18: the round's last library -- whole GPU suite, smoke, a second soak (fresh seeds), the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06q; mkdir -p $OUT/traces
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -4 $OUT/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
MDVT_SEGV_TRACE=1 MDVT_SEGV_TRACE_DIR=$ROOT/$OUT/traces python tools/soak.py --tag r06b --commit ${SOAK_COMMIT:-unknown} --seed0 616000 --seeds 900 --cases 400 --full 200 \
     --aux-seeds 40 --aux-cases 250 --batch-seeds 700 --batch-cases 100 --finish 1 --procs 14 --budget-min 15 > $OUT/soak.log 2>&1
tail -12 gpurun_out/soak_r06b/summary.md | cut -c1-250
find gpurun_out/soak_r06b -name "*.log" -size -3k -delete
find $OUT/traces -size 0 -delete; ls $OUT/traces | head -3
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -c 600 $OUT/bench.log
