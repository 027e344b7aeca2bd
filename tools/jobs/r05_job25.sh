#!/bin/bash
# r05 GPU job 25: the round's last library -- smoke, bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
mkdir -p gpurun_out/r05y
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/r05y/bench.json 2> gpurun_out/r05y/bench.err; echo "bench rc $?"; tail -c 600 gpurun_out/r05y/bench.json
