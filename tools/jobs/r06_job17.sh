#!/bin/bash
# r06 GPU job 17: early-z reject in the queue / huge walks of the general mesh path: parity (render suite, 4K launch sets, sweeps), then C4 mesh and the 1080p general paths timed
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06p; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_bench_sizes.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
{
echo "== C4 mesh"; MDVT_LIB_VARIANT= timeout 200 python tools/kbench.py default --mesh --c4 --width 3840 --height 2160 --frames 8 --rounds 7 --calls 5 2>&1 | tail -1
echo "== C4 mesh + infill"; MDVT_LIB_VARIANT= timeout 200 python tools/kbench.py default --mesh --infill --c4 --width 3840 --height 2160 --frames 8 --rounds 7 --calls 5 2>&1 | tail -1
echo "== product default"; MDVT_LIB_VARIANT= timeout 120 python tools/kbench.py default --mesh --infill --conv 2.5 --frames 32 --rounds 7 --calls 5 2>&1 | tail -1
echo "== mesh + convergence"; MDVT_LIB_VARIANT= timeout 120 python tools/kbench.py default --mesh --conv 2.5 --frames 32 --rounds 7 --calls 5 2>&1 | tail -1
echo "== mesh + pose 1080p"; MDVT_LIB_VARIANT= timeout 120 python tools/kbench.py default --mesh --pose --frames 32 --rounds 7 --calls 5 2>&1 | tail -1
} 2>&1 | grep -v "amdgpu.ids\|library:" > $OUT/ab.log
cat $OUT/ab.log
