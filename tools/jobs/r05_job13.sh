#!/bin/bash
# r05 GPU job 13: edge-point splat as list + chain ("new") against the one-kernel splat (libmdvt_hip_d0.so); the scanline walk with edge
# removal over 4 / 2 rows of cells at 80 VGPRs (libmdvt_hip_c4.so / _c2.so, list splat too)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05m; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in d0 "" c4 c2 d0 "" c4 c2; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab product_default_8 --mesh --infill --conv 2.5 --frames 8 | tee -a $OUT/ab.log
ab product_default_1 --mesh --infill --conv 2.5 --frames 1 | tee -a $OUT/ab.log
ab mesh_pose_edges --mesh --pose --infill --frames 32 | tee -a $OUT/ab.log
ab c4_mesh_edges --mesh --infill --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
MDVT_LIB_VARIANT= bash tools/profile_kbench.sh r05_pd8_list --mesh --infill --conv 2.5 --frames 8
