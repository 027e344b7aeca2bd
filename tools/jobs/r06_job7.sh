#!/bin/bash
# r06 GPU job 7: where the fused variant's 13 % go -- ablations of the BITS tail (tuning build), 128 frames
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06g; mkdir -p $OUT
{
echo "== plain"; timeout 90 python tools/kbench.py default --rounds 7 --calls 10 --frames 128 2>&1 | tail -2
echo "== bits+counts, skip: default(0) 8(no bit stores) 16(no global atomics) 32(no counting) 64(no BITS work) 72"
timeout 90 python tools/kbench.py default 8 16 32 64 72 --env MDVT_DEBUG_SKIP --bits --counts --rounds 7 --calls 10 --frames 128 2>&1 | tail -8
echo "== bits only"; timeout 90 python tools/kbench.py default 8 64 --env MDVT_DEBUG_SKIP --bits --rounds 7 --calls 10 --frames 128 2>&1 | tail -5
echo "== counts only"; timeout 90 python tools/kbench.py default 16 32 64 --env MDVT_DEBUG_SKIP --counts --rounds 7 --calls 10 --frames 128 2>&1 | tail -6
echo "== bits+counts, no byte mask"; timeout 90 python tools/kbench.py default 8 16 32 64 72 --env MDVT_DEBUG_SKIP --bits --counts --nomask --rounds 7 --calls 10 --frames 128 2>&1 | tail -8
} > $OUT/ablate.log 2>&1
cat $OUT/ablate.log
