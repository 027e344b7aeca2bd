#!/bin/bash
# r05 GPU job 3: replay of the soak's one failed batch job, the suite on the two-level queue counters, A/B against the previous
# commit's library (libmdvt_hip_prev.so, built from a worktree of HEAD~), queue census of C4 mesh, the headline's r05 profile.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05c; mkdir -p $OUT
for v in prev prev ""; do
  MDVT_LIB_VARIANT=$v AMD_LOG_LEVEL=1 MDVT_SWEEP_SEED=594109 MDVT_BATCH_CASES=100 timeout 600 python -m pytest tests/test_gpu_render.py -x -q -k test_randomised_batch_sweep > $OUT/replay_109_${v:-new}_$RANDOM.log 2>&1
  echo "replay batch_109 on '${v:-new}': rc $?"
done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {   # tag, kbench args
  tag=$1; shift
  for v in prev "" prev ""; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab mesh_pose --mesh --pose --frames 32 | tee -a $OUT/ab.log
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
MDVT_LIB_VARIANT=tuning MDVT_QUEUE_DUMP=1 python tools/kbench.py default --rounds 1 --calls 1 --mesh --c4 --width 3840 --height 2160 --frames 8 2>&1 | grep -i "queued" | sort | uniq -c | tee $OUT/queue_c4.log
MDVT_LIB_VARIANT=tuning MDVT_QUEUE_DUMP=1 python tools/kbench.py default --rounds 1 --calls 1 --mesh --conv 2.5 --frames 32 2>&1 | grep -i "queued" | sort | uniq -c | tee $OUT/queue_conv.log
bash tools/profile.sh r05 > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
export MDVT_LIB_VARIANT=
bash tools/profile_kbench.sh r05b_mesh_conv --mesh --conv 2.5 --frames 32
bash tools/profile_kbench.sh r05b_c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8
