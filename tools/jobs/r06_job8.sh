#!/bin/bash
# r06 GPU job 8: the proven short division, same-box A/B (MDVT_DEBUG_SKIP=128: IEEE expansion), plain and fused, 32 and 128 frames
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06h; mkdir -p $OUT
{
for rep in 1 2; do
echo "== plain 128 (default = short division where proven, 128 = IEEE)"; timeout 90 python tools/kbench.py default 128 default 128 --env MDVT_DEBUG_SKIP --rounds 7 --calls 10 --frames 128 2>&1 | tail -4
echo "== plain 32"; timeout 90 python tools/kbench.py default 128 default 128 --env MDVT_DEBUG_SKIP --rounds 9 --calls 20 --frames 32 2>&1 | tail -4
echo "== bits+counts 128"; timeout 90 python tools/kbench.py default 128 --env MDVT_DEBUG_SKIP --bits --counts --rounds 7 --calls 10 --frames 128 2>&1 | tail -2
echo "== bits+counts 32"; timeout 90 python tools/kbench.py default 128 --env MDVT_DEBUG_SKIP --bits --counts --rounds 9 --calls 20 --frames 32 2>&1 | tail -2
echo "== bits+counts no byte mask 128"; timeout 90 python tools/kbench.py default 128 --env MDVT_DEBUG_SKIP --bits --counts --nomask --rounds 7 --calls 10 --frames 128 2>&1 | tail -2
done
} 2>&1 | grep -v "amdgpu.ids\|library:" > $OUT/ab.log
cat $OUT/ab.log
