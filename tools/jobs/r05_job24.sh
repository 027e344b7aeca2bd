#!/bin/bash
# r05 GPU job 24: the banks' side stream and events pooled process-wide (never destroyed) -- suite, then the multi-frame sweeps with native
# backtraces again (tools/jobs/r05_job23.sh: 3 process deaths in ~3 000 such jobs before the change, the traced one in the HSA runtime's callback thread)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
mkdir -p gpurun_out/r05x
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05x/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05x/pytest.log
rm -rf gpurun_out/soak_r05d gpurun_out/r05w
BUDGET_MIN=${BUDGET_MIN:-21} bash tools/jobs/r05_job23.sh
