#!/bin/bash
# r05 GPU job 1: the suite, the bench line, and the fresh-context A/B of the r04 failing conditions (tuning hooks).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 3000 $OUT/bench.json
J="MDVT_WS_POOL=off,MDVT_WS_LAYOUT=joint"
bash tools/fresh_context_ab.sh $OUT/ab 2500 \
   "$J,MDVT_WS_FRESH=none" \
   "$J,MDVT_WS_FRESH=canary" \
   "$J,MDVT_WS_FRESH=memset" \
   "$J,MDVT_WS_FRESH=devsync" \
   "MDVT_WS_LAYOUT=joint" > $OUT/ab.log 2>&1
grep "== config" $OUT/ab.log
