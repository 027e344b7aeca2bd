#!/bin/bash
# r05 GPU job 20: the filter's flag plane zeroed by one kernel of the library's instead of the runtime's fill launches -- suite, smoke, A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05t; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in z0 "" z0 ""; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab product_default_1 --mesh --infill --conv 2.5 --frames 1 | tee -a $OUT/ab.log
ab mesh_infill --mesh --infill --frames 32 | tee -a $OUT/ab.log
ab points_edges --infill --frames 32 | tee -a $OUT/ab.log
