#!/usr/bin/env python3
"""profiles/<tag>_summary.md + _kernel_stats.csv from a tools/profile_kbench.sh run.
usage: python tools/summarize_kbench_profile.py gpurun_out/prof_<tag> <tag> <frames per launch> "<what was run>" """
import csv, glob, os, statistics, sys

src, tag, frames, what = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(pattern):
    for f in glob.glob(os.path.join(src, pattern), recursive=True):
        yield from csv.DictReader(open(f))


dur, pmc = {}, {}
for r in rows("trace/**/*kernel_trace.csv"):
    if "mdvt::" in r["Kernel_Name"]:
        dur.setdefault(r["Kernel_Name"].split("(")[0], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for name in ("fetch", "write"):
    for r in rows(f"pmc_{name}/**/*counter_collection.csv"):
        if "mdvt::" in r["Kernel_Name"]:
            pmc.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 16      # submissions: profile_kbench.sh runs 1 warm-up + 3 rounds x 5 calls;
                                                                # a submission may be split into several launches (frame chunks)
with open(os.path.join(repo, "profiles", f"{tag}_summary.md"), "w") as fo:
    fo.write(f"# rocprofv3 summary `{tag}`\n\n{what}\n\nCommand: `bash tools/profile_kbench.sh {tag} ...` (kernel trace + separate "
             f"FETCH_SIZE / WRITE_SIZE passes); {frames} frames per submission.  HBM bytes as MI355X_MICROARCH.md prescribes "
             "(counter x 1024 B, FETCH_SIZE x 2 on gfx950).\n\n")
    fo.write("| kernel | launches | avg us per launch | us per frame | HBM read MB per frame | HBM write MB per frame |\n|---|---|---|---|---|---|\n")
    tot = 0.0
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        us_frame = sum(v) / 1e3 / (launches * frames)
        f, w = pmc.get((k, "FETCH_SIZE")), pmc.get((k, "WRITE_SIZE"))
        rd = 2.0 * sum(f) * 1024 / 1e6 / (launches * frames) if f else float("nan")
        wr = sum(w) * 1024 / 1e6 / (launches * frames) if w else float("nan")
        tot += us_frame
        fo.write(f"| `{k}` | {len(v)} | {statistics.mean(v)/1e3:.1f} | {us_frame:.2f} | {rd:.1f} | {wr:.1f} |\n")
    fo.write(f"\nSum of kernel time: {tot:.1f} us per frame.\n")
    p = os.path.join(src, "kbench_plain.txt")
    if os.path.exists(p):
        fo.write("\nUn-profiled timing of the same configuration (HIP events around 5 back-to-back submissions, median of 3):\n\n```\n"
                 + open(p).read().strip().splitlines()[-1] + "\n```\n")
for f in glob.glob(os.path.join(src, "trace/**/*kernel_stats.csv"), recursive=True):
    open(os.path.join(repo, "profiles", f"{tag}_kernel_stats.csv"), "w").write(open(f).read())
print(open(os.path.join(repo, "profiles", f"{tag}_summary.md")).read())
