#!/bin/bash
# r05 GPU job 18: the round's candidate library -- suite, bench line, A/B of the queue walk's occupancy (q5 / q6: 96 / 80 VGPRs), soak, profiles
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05r; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
ab() {
  tag=$1; shift
  for v in "" q5 q6 "" q5 q6; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_pose --mesh --pose --frames 32 | tee -a $OUT/ab.log
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
python tools/soak.py --tag r05b --commit ${SOAK_COMMIT:-unknown} --seed0 534000 --seeds 500 --cases 400 --full 120 \
     --aux-seeds 20 --aux-cases 250 --batch-seeds 200 --batch-cases 100 --procs 14 --budget-min 11 > $OUT/soak.log 2>&1
tail -12 gpurun_out/soak_r05b/summary.md
export MDVT_LIB_VARIANT=
bash tools/profile_kbench.sh r05f_product_default --mesh --infill --conv 2.5 --frames 32
bash tools/profile_kbench.sh r05f_c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8
bash tools/profile_kbench.sh r05f_mesh_conv --mesh --conv 2.5 --frames 32
bash tools/profile_kbench.sh r05f_mesh_pose --mesh --pose --frames 32
