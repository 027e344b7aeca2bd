#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + separate PMC passes.
# Usage: bash tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra $*"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python $ROOT/bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- python $ROOT/bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.log"
python $ROOT/bench.py $* > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"
find "$OUT" -name "*.csv" | head -20
tail -2 "$OUT/bench_plain.json"
