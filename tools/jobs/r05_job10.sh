#!/bin/bash
# r05 GPU job 10: rows of cells per workgroup chosen per kernel (scanline walk with edge removal: 1, else 4) = "new"; the same with
# an occupancy hint of 6 / 5 waves per SIMD on the cell walks (libmdvt_hip_h6.so / _h5.so); against q (one row everywhere) and r4 (four everywhere)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05j; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in q r4 "" h6 h5 q r4 "" h6 h5; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab mesh_pose --mesh --pose --frames 32 | tee -a $OUT/ab.log
ab mesh_pose_edges --mesh --pose --infill --frames 32 | tee -a $OUT/ab.log
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
