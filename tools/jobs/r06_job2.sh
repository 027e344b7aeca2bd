#!/bin/bash
# r06 GPU job 2: A/B of the points kernels after the point-rule change (product library) against the r05 library on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06b; mkdir -p $OUT
for rep in 1 2 3; do
  for v in "" r05; do
    echo "== variant '${v}' rep $rep" >> $OUT/ab.log
    MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 9 --calls 10 --frames 128 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT/ab.log
  done
done
for v in "" r05; do
  echo "== infill points variant '${v}'" >> $OUT/ab.log
  MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 7 --calls 10 --frames 32 --infill 2>&1 | tail -1 >> $OUT/ab.log
  echo "== c4 points variant '${v}'" >> $OUT/ab.log
  MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 7 --calls 5 --frames 8 --width 3840 --height 2160 --c4 2>&1 | tail -1 >> $OUT/ab.log
done
cat $OUT/ab.log
timeout 900 python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "points" 2>&1 | tail -3
