#!/bin/bash
# Kernel-trace + FETCH_SIZE / WRITE_SIZE passes (separate runs) of a tools/kbench.py configuration -- for the paths
# bench.py's main line does not time (mesh + --infill_mask + convergence, pose, ...).
# Usage: bash tools/profile_kbench.sh <tag> <kbench args...>     (run on the GPU box through gpurun)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="default --rounds 3 --calls 5 $*"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/tools/kbench.py $ARGS > "$OUT/kbench_trace.txt" 2> "$OUT/trace.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python $ROOT/tools/kbench.py $ARGS > /dev/null 2> "$OUT/fetch.log"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- python $ROOT/tools/kbench.py $ARGS > /dev/null 2> "$OUT/write.log"
python $ROOT/tools/kbench.py $ARGS > "$OUT/kbench_plain.txt" 2>&1
tail -1 "$OUT/kbench_plain.txt"
