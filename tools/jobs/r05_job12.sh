#!/bin/bash
# r05 GPU job 12: the general vertex programme's divisions by a shared reciprocal -- self-test, suite, A/B against the commit before (libmdvt_hip_d0.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_arith.py -x -q > $OUT/pytest_arith.log 2>&1; echo "arith rc $?"; tail -3 $OUT/pytest_arith.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in d0 "" d0 ""; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab mesh_pose --mesh --pose --frames 32 | tee -a $OUT/ab.log
ab points_conv --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
ab c4_points --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
