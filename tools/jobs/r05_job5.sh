#!/bin/bash
# r05 GPU job 5: k_mesh_band with its arguments read through the kernarg segment and no carried row geometry -- suite, A/B against
# libmdvt_hip_prev.so (d53eb0d: the band kernel of r04), SQ counters; the grid-barrier probe (verdict item 5b)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05e; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in prev "" prev ""; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 7 --calls 10 "$@" 2>&1 | tail -1)"
  done
}
ab mesh --mesh --frames 32 | tee $OUT/ab.log
ab mesh_infill --mesh --infill --frames 32 | tee -a $OUT/ab.log
ab mesh_zout --mesh --zout --frames 32 | tee -a $OUT/ab.log
ab mesh_1frame --mesh --frames 1 | tee -a $OUT/ab.log
hipcc --offload-arch=gfx950 -O3 tools/probe/grid_barrier_probe.hip -o /tmp/grid_probe 2> /dev/null
timeout 120 /tmp/grid_probe 260 | tee $OUT/grid_probe.log
export MDVT_LIB_VARIANT=
bash tools/pmc_kbench.sh r05_mesh --mesh --frames 32 > $OUT/pmc_mesh.log 2>&1; grep "k_mesh_band" gpurun_out/pmck_r05_mesh/summary.txt | head -30
