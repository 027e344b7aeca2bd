#!/usr/bin/env python3
"""Where a kernel's lane spills (v_readlane / v_writelane from spilled SGPRs) sit: per basic block of the ISA listing, with the
block's loop depth worked out from its back edges -- a static count says nothing about a kernel whose hot loop holds none.
    make -C metric_depth_video_toolbox_amd/csrc asm SRC=mdvt_mesh_band
    python tools/isa_blocks.py /tmp/mdvt_mesh_band.s 'k_mesh_bandILi0ELi512ELb0'
Loop depth of a block = what LLVM prints on its label in the listing."""
import re, sys

src, pat = sys.argv[1], sys.argv[2]
s = open(src).read()
m = re.search(r'^(_ZN4mdvt\w*%s\w*):' % re.escape(pat), s, re.M)
if not m: sys.exit("kernel not found")
body = s[m.start():s.index('.Lfunc_end', m.start())]
blocks, cur = [], [m.group(1), []]
for line in body.split('\n')[1:]:
    lm = re.match(r'^(\.LBB\d+_\d+):', line)
    if lm:
        blocks.append(cur); cur = [lm.group(1), []]
    elif line.startswith('\t') and not line.strip().startswith(('.', ';')):
        cur[1].append(line.strip())
blocks.append(cur)
# loop depth: LLVM's own annotation on the block label ("; in Loop: Header=BB1_20 Depth=2", "; =>This Inner Loop Header: Depth=4")
depth = [0] * len(blocks)
lab_depth = {}
for lm in re.finditer(r'^(\.LBB\d+_\d+):[^\n]*?Depth=(\d+)', body, re.M):
    lab_depth[lm.group(1)] = int(lm.group(2))
for k, (lab, ins) in enumerate(blocks):
    depth[k] = lab_depth.get(lab, 0)
tot = {}
for k, (lab, ins) in enumerate(blocks):
    d = depth[k]
    t = tot.setdefault(d, [0, 0, 0, 0])
    t[0] += len(ins); t[1] += sum(i.startswith('v_readlane') for i in ins); t[2] += sum(i.startswith('v_writelane') for i in ins)
    t[3] += sum(i.startswith('v_') for i in ins)
print(f"{m.group(1)}: {sum(len(b[1]) for b in blocks)} instructions in {len(blocks)} blocks")
print("| loop depth | instructions | VALU | v_readlane | v_writelane | lane spills / VALU |")
print("|---|---|---|---|---|---|")
for d in sorted(tot):
    n, r, w, v = tot[d]
    print(f"| {d} | {n} | {v} | {r} | {w} | {100.0 * (r + w) / max(v, 1):.1f} % |")
