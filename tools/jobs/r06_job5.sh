#!/bin/bash
# r06 GPU job 5: whole GPU suite on the restored tree, the bench line, the headline profile (kernel trace + FETCH/WRITE passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06e; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -6 $OUT/pytest_all.log
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -1 $OUT/bench.log | cut -c1-3000
bash tools/profile.sh r06_points > $OUT/profile.log 2>&1; tail -4 $OUT/profile.log | cut -c1-600
