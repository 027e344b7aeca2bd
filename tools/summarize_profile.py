#!/usr/bin/env python3
"""Summarise a tools/profile.sh run (rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes) into
profiles/<tag>_summary.json + .md.   usage: python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> [kernel-substring]

HBM traffic is taken from the PMC counters exactly as MI355X_MICROARCH.md (section HBM) prescribes:
separate --pmc passes; FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B (hbm_bytes = counter * 1024);
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read, so the read
side is doubled (the kernel's loads are 12-byte-per-lane dwordx3 streams; the corrected figure lands on
the algorithmic read bytes to 4 digits, which is itself the calibration the guide asks for)."""
import csv, glob, json, os, sys, statistics

src, tag = sys.argv[1], sys.argv[2]
needle = sys.argv[3] if len(sys.argv) > 3 else "::k_"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"tag": tag, "kernels": {}}
try:        # the tree the summary is written in (the GPU box has no .git: the profile ran on a snapshot of this tree or an older one)
    import subprocess
    out["commit"] = subprocess.check_output(["git", "-C", repo, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    out["commit"] = None

def rows(pattern):
    for f in glob.glob(os.path.join(src, pattern), recursive=True):
        yield from csv.DictReader(open(f))

dur = {}
for r in rows("trace/**/*kernel_trace.csv"):
    if needle in r["Kernel_Name"]:
        dur.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = {}
for name in ("fetch", "write"):
    for r in rows(f"pmc_{name}/**/*counter_collection.csv"):
        if needle in r["Kernel_Name"]:
            pmc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
bench = {}
for name in ("plain", "trace"):
    p = os.path.join(src, f"bench_{name}.json")
    if os.path.exists(p):
        try:
            bench[name] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            pass
steps = bench.get("trace", {}).get("steps", 0)
for k, v in dur.items():
    timed = v[-steps:] if steps and len(v) >= steps else v
    e = {"calls": len(v), "avg_ns_all": statistics.mean(v), "avg_ns_timed_region": statistics.mean(timed),
         "median_ns_timed_region": statistics.median(timed), "min_ns": min(v), "max_ns": max(v)}
    f = pmc.get((k, "FETCH_SIZE")); w = pmc.get((k, "WRITE_SIZE"))
    if f and w:
        e["FETCH_SIZE_raw"] = statistics.mean(f); e["WRITE_SIZE_raw"] = statistics.mean(w)
        e["hbm_read_bytes_per_launch"] = 2.0 * statistics.mean(f) * 1024      # gfx950 x2 correction
        e["hbm_write_bytes_per_launch"] = statistics.mean(w) * 1024
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    out["kernels"][k] = e
out["bench"] = bench
os.makedirs(os.path.join(repo, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(repo, "profiles", f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join(repo, "profiles", f"{tag}_summary.md"), "w") as fo:
    fo.write(f"# rocprofv3 summary `{tag}`\n\nCommand: `bash tools/profile.sh {tag}` (kernel trace + separate FETCH_SIZE / WRITE_SIZE passes of the same bench command).\n\n")
    fo.write("| kernel | calls | avg us (all) | avg us (timed region) | HBM read MB/launch (2x-corrected) | HBM write MB/launch |\n|---|---|---|---|---|---|\n")
    for k, e in out["kernels"].items():
        fo.write(f"| `{k.split('(')[0]}` | {e['calls']} | {e['avg_ns_all']/1e3:.1f} | {e['avg_ns_timed_region']/1e3:.1f} | "
                 f"{e.get('hbm_read_bytes_per_launch', float('nan'))/1e6:.1f} | {e.get('hbm_write_bytes_per_launch', float('nan'))/1e6:.1f} |\n")
    if "plain" in bench:
        fo.write("\nUn-profiled bench line of the same run:\n\n```json\n" + json.dumps(bench["plain"]) + "\n```\n")
for f in glob.glob(os.path.join(src, "trace/**/*kernel_stats.csv"), recursive=True):
    with open(f) as fi, open(os.path.join(repo, "profiles", f"{tag}_kernel_stats.csv"), "w") as fo:
        fo.write(fi.read())
print(open(os.path.join(repo, "profiles", f"{tag}_summary.md")).read())
