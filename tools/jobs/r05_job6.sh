#!/bin/bash
# r05 GPU job 6: cell walks over kRowsWG rows of cells per workgroup (vertex rows in a ring): suite + A/B of 2 / 4 / 8 rows against
# the library of the commit before (libmdvt_hip_q.so) and of d53eb0d (libmdvt_hip_prev.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in prev q r2 "" r8 q r2 "" r8; do
    echo "== $tag on '${v:-r4}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab mesh_pose --mesh --pose --frames 32 | tee -a $OUT/ab.log
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
