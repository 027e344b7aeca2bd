#!/bin/bash
# r06 GPU job 14: the fused variant with its set of outputs fixed at compile time and the counts taken from the packed mask: tests, same-box timings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "compact or fused or randomised_parity or raw_abi" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
{
for rep in 1; do
echo "== plain 128"; MDVT_LIB_VARIANT= timeout 90 python tools/kbench.py default --rounds 7 --calls 10 --frames 128 2>&1 | tail -1
echo "== bits+counts 128 (spec 15)"; MDVT_LIB_VARIANT= timeout 90 python tools/kbench.py default --bits --counts --rounds 7 --calls 10 --frames 128 2>&1 | tail -1
echo "== bits only 128 (spec 11)"; MDVT_LIB_VARIANT= timeout 90 python tools/kbench.py default --bits --rounds 7 --calls 10 --frames 128 2>&1 | tail -1
echo "== counts only 128 (spec 13)"; MDVT_LIB_VARIANT= timeout 90 python tools/kbench.py default --counts --rounds 7 --calls 10 --frames 128 2>&1 | tail -1
echo "== bits+counts no byte mask 128 (spec 14)"; MDVT_LIB_VARIANT= timeout 90 python tools/kbench.py default --bits --counts --nomask --rounds 7 --calls 10 --frames 128 2>&1 | tail -1
echo "== generic form (tuning library, MDVT_DEBUG_SKIP=0) bits+counts 128"; timeout 90 python tools/kbench.py 0 --env MDVT_DEBUG_SKIP --bits --counts --rounds 7 --calls 10 --frames 128 2>&1 | tail -1
echo "== plain 32"; MDVT_LIB_VARIANT= timeout 90 python tools/kbench.py default --rounds 9 --calls 20 --frames 32 2>&1 | tail -1
echo "== bits+counts 32"; MDVT_LIB_VARIANT= timeout 90 python tools/kbench.py default --bits --counts --rounds 9 --calls 20 --frames 32 2>&1 | tail -1
done
} 2>&1 | grep -v "amdgpu.ids\|library:" > $OUT/ab.log
cat $OUT/ab.log
