#!/bin/bash
# r05 GPU job 21: long soak of the round's final library + the fresh-context stress on the product library (12 x 5000) and on the r04 layout with the pool
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05u; mkdir -p $OUT
MDVT_LIB_VARIANT= SWEEP_CASE=231 bash tools/fresh_context_ab.sh $OUT/ab 5000 "MDVT_WS_LAYOUT=joint" > $OUT/ab_tuning.log 2>&1; grep "== config" $OUT/ab_tuning.log
for k in $(seq 1 12); do ( MDVT_SWEEP_SEED=504249 MDVT_SWEEP_CASES=400 CASE=231 FRESH=1 ITERS=5000 timeout 900 python tests/dbg_stress_case.py > $OUT/prod_p$k.log 2>&1 ) & done; wait
echo "== product library, 12 x 5000 fresh contexts: $(grep -h '^case ' $OUT/prod_p*.log | awk '{s += $NF} END {print s + 0}') bad first renders, $(grep -L '^case ' $OUT/prod_p*.log | wc -l) processes without a result"
python tools/soak.py --tag r05c --commit ${SOAK_COMMIT:-unknown} --seed0 544000 --seeds 2000 --cases 400 --full 400 \
     --aux-seeds 60 --aux-cases 250 --batch-seeds 400 --batch-cases 100 --finish 2 --procs 14 --budget-min 34 > $OUT/soak.log 2>&1
tail -14 gpurun_out/soak_r05c/summary.md
