#!/bin/bash
# r05 GPU job 23: the two process crashes of the round's soaks (both in multi-frame sweeps: an abort, a segfault inside
# mdvt_render_stereo_batch; none in 3 500 single-frame sweep jobs) -- multi-frame sweeps only, with a native backtrace of whatever
# thread dies (tools/probe/segv_trace.c via tests/conftest.py, MDVT_SEGV_TRACE=1), 14 processes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05w; mkdir -p $OUT/traces
MDVT_SEGV_TRACE=1 MDVT_SEGV_TRACE_DIR=$ROOT/$OUT/traces python tools/soak.py --tag r05d --commit ${SOAK_COMMIT:-unknown} --seed0 564000 --seeds 0 --cases 0 --full 0 --finish 0 \
     --aux-seeds 0 --batch-seeds 2400 --batch-cases 100 --procs 14 --budget-min ${BUDGET_MIN:-19} > $OUT/soak.log 2>&1
tail -8 gpurun_out/soak_r05d/summary.md | cut -c1-300
find gpurun_out/soak_r05d -name "batch_*.log" -size -3k -delete
find $OUT/traces -size 0 -delete
ls $OUT/traces | head; for f in $(ls $OUT/traces/* 2>/dev/null | head -4); do echo "== $f"; head -40 $f; done
python - <<'PY'
import re, subprocess, glob
for f in glob.glob("gpurun_out/r05w/traces/*")[:4]:
    print("== resolved", f)
    for line in open(f):
        m = re.match(r"(\S+)\((\S*)\+0x([0-9a-f]+)\)", line)
        if m and ("libmdvt" in m.group(1) or "libamdhip" in m.group(1) or "libhsa" in m.group(1)):
            r = subprocess.run(["addr2line", "-f", "-C", "-e", m.group(1), "0x" + m.group(3)], capture_output=True, text=True).stdout.split("\n")[0]
            print("  ", m.group(1).split("/")[-1], "+0x" + m.group(3), r[:120])
PY
