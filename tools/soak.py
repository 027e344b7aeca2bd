#!/usr/bin/env python3
"""Widened parity soak on the GPU box: the seeded random sweeps of tests/test_gpu_render.py over many fresh seeds and at
full size, plus the full-size infill-mask completion check, run as parallel pytest processes (the oracle is the slow
side: one CPU core per process).  Writes one log per job under gpurun_out/soak_<tag>/ and a summary
gpurun_out/soak_<tag>/summary.md -- copy it to profiles/<round>_soak_summary.md (tracked) after the run.

    python tools/soak.py --tag r03 --commit $(git rev-parse --short HEAD) --seeds 100 --cases 400 --full 104 \
                         --aux-seeds 24 --aux-cases 200 --procs 12 --budget-min 60

The product is only ever the thing checked: every job is `pytest tests/test_gpu_render.py -k <sweep>` with the
MDVT_SWEEP_* environment the tests document, or tests/dbg_finish_fullsize.py.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--commit", default="unknown")
    ap.add_argument("--seed0", type=int, default=303000, help="first seed (fresh range per round)")
    ap.add_argument("--seeds", type=int, default=100, help="render-sweep seeds at the widened small sizes")
    ap.add_argument("--cases", type=int, default=400, help="cases per seed")
    ap.add_argument("--full", type=int, default=104, help="full-size cases in total (split into jobs of --full-per-job)")
    ap.add_argument("--full-per-job", type=int, default=4)
    ap.add_argument("--full-sizes", default="1920x1080,1280x720,3840x48,1000x1000")
    ap.add_argument("--aux-seeds", type=int, default=24)
    ap.add_argument("--aux-cases", type=int, default=200)
    ap.add_argument("--batch-seeds", type=int, default=0, help="multi-frame call sweeps (test_randomised_batch_sweep: 2..40 frames per call, banks)")
    ap.add_argument("--batch-cases", type=int, default=100, help="calls per batch seed")
    ap.add_argument("--finish", type=int, default=1, help="run tests/dbg_finish_fullsize.py (both seed kinds) this many times")
    ap.add_argument("--procs", type=int, default=12)
    ap.add_argument("--budget-min", type=float, default=60.0, help="stop starting new jobs after this many minutes")
    a = ap.parse_args()

    out = os.path.join(REPO, "gpurun_out", f"soak_{a.tag}")
    os.makedirs(out, exist_ok=True)
    py = sys.executable
    jobs = []          # (name, kind, env, argv, cases)
    for k in range(a.full // a.full_per_job):
        jobs.append((f"full_{k:03d}", "render sweep, full size", {"MDVT_SWEEP_SEED": str(a.seed0 + 50000 + k), "MDVT_SWEEP_CASES": str(a.full_per_job),
                                                                   "MDVT_SWEEP_SIZES": a.full_sizes},
                     [py, "-m", "pytest", "tests/test_gpu_render.py", "-x", "-q", "-k", "test_randomised_parity_sweep"], a.full_per_job))
    for k in range(a.finish):
        jobs.append((f"finish_{k}", "infill-mask completion, 1080p, pure-shift and converged seeds", {}, [py, "tests/dbg_finish_fullsize.py"], 18))
    for k in range(a.seeds):
        jobs.append((f"sweep_{k:03d}", "render sweep, widened sizes", {"MDVT_SWEEP_SEED": str(a.seed0 + k), "MDVT_SWEEP_CASES": str(a.cases)},
                     [py, "-m", "pytest", "tests/test_gpu_render.py", "-x", "-q", "-k", "test_randomised_parity_sweep"], a.cases))
    for k in range(a.batch_seeds):
        jobs.append((f"batch_{k:03d}", "multi-frame calls (2..40 frames, mixed pure / converged / posed, launch sets on two banks)",
                     {"MDVT_SWEEP_SEED": str(a.seed0 + 70000 + k), "MDVT_BATCH_CASES": str(a.batch_cases)},
                     [py, "-m", "pytest", "tests/test_gpu_render.py", "-x", "-q", "-k", "test_randomised_batch_sweep"], a.batch_cases))
    for k in range(a.aux_seeds):
        jobs.append((f"aux_{k:03d}", "stand-alone entry points sweep", {"MDVT_SWEEP_SEED": str(a.seed0 + 90000 + k), "MDVT_SWEEP_CASES": str(a.aux_cases)},
                     [py, "-m", "pytest", "tests/test_gpu_render.py", "-x", "-q", "-k", "test_randomised_aux_sweep"], a.aux_cases))

    t0 = time.time()
    running, done, pending = [], [], list(jobs)
    skipped = []
    while pending or running:
        while pending and len(running) < a.procs:
            if (time.time() - t0) / 60.0 > a.budget_min:
                skipped, pending = pending, []
                break
            name, kind, env, argv, cases = pending.pop(0)
            log = open(os.path.join(out, name + ".log"), "w")
            p = subprocess.Popen(argv, cwd=REPO, env=dict(os.environ, **env, OMP_NUM_THREADS="1"), stdout=log, stderr=subprocess.STDOUT)
            running.append((name, kind, env, cases, p, log, time.time()))
        still = []
        for r in running:
            rc = r[4].poll()
            if rc is None:
                still.append(r)
            else:
                r[5].close()
                done.append({"name": r[0], "kind": r[1], "env": r[2], "cases": r[3], "rc": rc, "seconds": round(time.time() - r[6], 1)})
        running = still
        time.sleep(0.5)

    fails = [d for d in done if d["rc"] != 0]
    kinds = {}
    for d in done:
        k = kinds.setdefault(d["kind"], {"jobs": 0, "cases": 0, "failed_jobs": 0, "seconds": 0.0})
        k["jobs"] += 1; k["cases"] += d["cases"]; k["failed_jobs"] += d["rc"] != 0; k["seconds"] += d["seconds"]
    mism = []
    for d in fails:
        txt = open(os.path.join(out, d["name"] + ".log")).read()
        m = re.findall(r"(AssertionError[^\n]*|MISMATCH[^\n]*|Error[^\n]*)", txt)
        mism.append((d["name"], d["env"], (m[:3] if m else [txt[-300:]])))
    dev = "?"
    try:
        import torch
        dev = torch.cuda.get_device_name(0)
    except Exception:
        pass
    with open(os.path.join(out, "summary.md"), "w") as fh:
        fh.write(f"# Parity soak {a.tag}\n\n")
        fh.write(f"* commit under test: `{a.commit}` (the tree gpurun shipped; `libmdvt_hip.so` built from it)\n")
        fh.write(f"* device: {dev}; {a.procs} parallel pytest processes; wall {(time.time() - t0) / 60:.1f} min\n")
        fh.write(f"* seeds: render sweep {a.seed0}..{a.seed0 + a.seeds - 1} x {a.cases} cases (sizes of `test_randomised_parity_sweep` + the soak sizes), "
                 f"full size {a.seed0 + 50000}.. x {a.full_per_job} cases over {a.full_sizes}, multi-frame calls {a.seed0 + 70000}.. x {a.batch_cases} calls, "
                 f"aux {a.seed0 + 90000}.. x {a.aux_cases} cases\n")
        fh.write(f"* jobs finished {len(done)} of {len(jobs)} ({len(skipped)} not started: time budget), **failed jobs: {len(fails)}**\n\n")
        fh.write("| kind | jobs | cases | failed jobs | CPU-seconds |\n|---|---|---|---|---|\n")
        for kind, k in kinds.items():
            fh.write(f"| {kind} | {k['jobs']} | {k['cases']} | {k['failed_jobs']} | {k['seconds']:.0f} |\n")
        fh.write("\nEvery case compares every output plane (RGB, hole mask, depth planes, seed image, finished infill mask) with the\n"
                 "oracle bit for bit; a job stops at its first mismatch (`-x`).\n")
        if mism:
            fh.write("\n## Mismatches\n\n")
            for name, env, lines in mism:
                fh.write(f"* `{name}` {json.dumps(env)}: " + " / ".join(l.strip()[:300] for l in lines) + "\n")
        else:
            fh.write("\n**Mismatches: 0.**\n")
    print(open(os.path.join(out, "summary.md")).read())
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
