#!/bin/bash
# r05 GPU job 17: all four cell walks over four rows of cells with opaque per-row indices, scanline walks and small<0> at 80 VGPRs ("new");
# small<2> at 80 VGPRs too (s6); against the commit before (d1)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05q; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in d1 "" s6 d1 "" s6; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab mesh_pose --mesh --pose --frames 32 | tee -a $OUT/ab.log
ab mesh_pose_edges --mesh --pose --infill --frames 32 | tee -a $OUT/ab.log
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
ab c4_mesh_edges --mesh --infill --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
