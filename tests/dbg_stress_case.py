"""Renders one case of the randomised sweep many times (FRESH=1: a new context per render) and counts the renders that differ from
the oracle -- the replay that found what profiles/r04_soak_summary.md describes.  Debug helper, not collected by pytest
(tests/test_gpu_fresh_context.py runs it as its trip-wire).
  MDVT_SWEEP_SEED=504249 MDVT_SWEEP_CASES=400 CASE=231 FRESH=1 ITERS=2500 python tests/dbg_stress_case.py   (run 12 at once)

With MDVT_LIB_VARIANT=tuning a render that differs is DIAGNOSED before the context goes: the general mesh path's triangle-queue
block is read back (mdvt_debug_read), the same context renders the frame again (a context's later renders were never wrong), the
block is read back again and the two are compared region by region -- which words of the bad render's block differ from the good
one's, and what they hold: zero, the canary of MDVT_WS_FRESH=canary (0xC5C5C5C5), or something else.  The tuning hooks that
put the r04 conditions back: MDVT_WS_POOL=off (hipMalloc / hipFree per context), MDVT_WS_FRESH=none|canary|devsync|memset,
MDVT_WS_LAYOUT=joint (both huge lists in the queue's block: 2.2 MB at 100 x 31), MDVT_WS_PAD=<bytes>."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc
from test_gpu_render import sweep_cases

CANARY = 0xC5C5C5C5


def kind_of(v):
    return "zero" if v == 0 else ("canary" if v == CANARY else "0x%08x" % v)


def diagnose(r, bad_blk, info, render_again, tag):
    """bad_blk: the queue block after the differing render; renders again on the same context and compares."""
    ok = render_again()
    good_blk, _ = r.ctx.debug_read_queue_block()
    nbytes, off_cnt, nseg_cap, off_huge, off_tie, W, H, slots = info
    nseg = H                                              # a single-frame render
    lines = [f"{tag}: second render on the same context {'matches the oracle' if ok else 'DIFFERS TOO'}; block {nbytes} B, "
             f"counters at dword {off_cnt}, huge list at {off_huge}, tie flags at {off_tie} (byte {4 * off_tie})"]
    cb, cg = bad_blk[off_cnt:off_cnt + nseg], good_blk[off_cnt:off_cnt + nseg]
    sh = 4                                               # (queue_shift_for, mdvt_mesh_general.hip: blocks of 2^sh segments)
    while ((nseg + (1 << sh) - 1) >> sh) > 4096: sh += 1
    nc = (nseg + (1 << sh) - 1) >> sh
    pb, pg = bad_blk[off_cnt + nseg:off_cnt + nseg + nc], good_blk[off_cnt + nseg:off_cnt + nseg + nc]
    lines.append(f"  queued triangles: bad {int(cb.sum())} (block counters' total {int(pb.sum())}), good {int(cg.sum())} (block counters' total {int(pg.sum())})")
    for sgm in np.nonzero(cb != cg)[0]:
        lines.append(f"  segment {sgm}: counter bad {int(cb[sgm])} ({kind_of(int(cb[sgm]))}) good {int(cg[sgm])}")
    if not np.array_equal(pb, np.add.reduceat(cb, np.arange(0, nseg, 1 << sh))):
        lines.append("  bad render: the block counters are NOT the sums of their segments' counters")
    ent_b = bad_blk[:off_cnt].reshape(-1, 2); ent_g = good_blk[:off_cnt].reshape(-1, 2)
    for sgm in range(nseg):
        nb, ng = int(cb[sgm]), int(cg[sgm])
        eb = ent_b[sgm * 4 * W: sgm * 4 * W + min(nb, 4 * W)].copy(); eg = ent_g[sgm * 4 * W: sgm * 4 * W + ng].copy()
        eb[:, 1] &= 0x7FFFFFFF; eg[:, 1] &= 0x7FFFFFFF           # (bit 31: "its row blocks are in the huge list")
        sb = set(map(tuple, eb.tolist())); sg = set(map(tuple, eg.tolist()))
        if sb != sg or nb != ng:
            miss = sorted(sg - sb); extra = sorted(sb - sg)
            lines.append(f"  segment {sgm}: {len(miss)} entries of the good render missing, {len(extra)} entries only in the bad one: "
                         + ", ".join(f"({kind_of(a)}, {kind_of(b)})" for a, b in extra[:6]))
            # what sits in the bad block where the good block's missing entries are (positions differ between runs; show raw slots)
            raw = ent_b[sgm * 4 * W: sgm * 4 * W + max(nb, ng)]
            holes = [k for k in range(len(raw)) if tuple((int(raw[k, 0]), int(raw[k, 1]) & 0x7FFFFFFF)) not in sg]
            lines.append("    raw slots of the bad block not holding an entry of the good set: "
                         + ", ".join(f"[{k}]=({kind_of(int(raw[k, 0]))}, {kind_of(int(raw[k, 1]))})" for k in holes[:8]))
    hb, hg = bad_blk[off_huge:off_tie], good_blk[off_huge:off_tie]
    nlist = (off_tie - off_huge) // (2 * orc_huge_cap() + 2)
    for li in range(nlist):
        o = li * (2 * orc_huge_cap() + 2)
        lines.append(f"  huge list {li}: counters bad {hb[o + 2 * orc_huge_cap():o + 2 * orc_huge_cap() + 2].tolist()} good {hg[o + 2 * orc_huge_cap():o + 2 * orc_huge_cap() + 2].tolist()}")
    tb, tg = bad_blk[off_tie:], good_blk[off_tie:]
    d = np.nonzero(tb != tg)[0]
    lines.append(f"  tie flag + tile bits ({tb.size} dwords): {d.size} differ" + "".join(f"; [{k}] bad {kind_of(int(tb[k]))} good {kind_of(int(tg[k]))}" for k in d[:6]))
    print("\n".join(lines), flush=True)


def orc_huge_cap():
    return 1 << 17            # mdvt_internal.h kHugeCap


target = int(os.environ["CASE"]); iters = int(os.environ.get("ITERS", "1500")); fresh = os.environ.get("FRESH", "1") == "1"
tuning = os.environ.get("MDVT_LIB_VARIANT", "") == "tuning"
for cs in sweep_cases(synthetic):
    if target >= 0 and cs["case"] != target: continue
    if target < 0 and not (cs["mesh"] and cs["conv_d"] is not None and cs["T"] is None and not cs["infill"]): continue
    W, H, mesh, infill, T = cs["W"], cs["H"], cs["mesh"], cs["infill"], cs["T"]
    d, c = torch.from_numpy(cs["depth_rgb"]).cuda()[None], torch.from_numpy(cs["color"]).cuda()[None]
    mk = lambda: sr.StereoRerenderer(W, H, pupillary_distance=cs["ipd"], max_depth=cs["max_depth"], master_xfov=cs["master"],
                                     render_as_pointcloud=not mesh, infill_mask=infill, dont_place_points_in_edges=cs["no_pts"])
    r = mk(); p = r.frame_params(xfov=cs["xfov"], convergence_distance=cs["conv_d"], transformation=T)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(W, H, K, ipd_m=cs["ipd"] / 1000, max_depth=cs["max_depth"], depth_scale=p.depth_scale, mode=orc.MODE_MESH if mesh else orc.MODE_POINTS,
                         remove_edges=r.remove_edges, edge_points=int(r.edge_points), conv_angle=p.convergence_angle, T=T, key_rgb=r.key_rgb)
    want = orc.render_stereo(op, cs["depth_rgb"], cs["color"], want_depth=True)
    wm = np.concatenate([want["left_mask"], want["right_mask"]], 1); wc = np.concatenate([want["left_rgb"], want["right_rgb"]], 1)
    wm_t, wc_t = torch.from_numpy(wm).cuda(), torch.from_numpy(wc).cuda()
    bad = 0

    def same(got):
        return torch.equal(got["mask"][0], wm_t) and torch.equal(got["sbs"][0], wc_t)

    for it in range(iters):
        if fresh and it: r.close(); r = mk()
        got = r.render(d, c, [p], want_depth=True)
        if not same(got):
            bad += 1
            if bad <= 3:
                print("iter", it, "mask diffs", int((got["mask"][0] != wm_t).sum()), "rgb diffs", int((got["sbs"][0] != wc_t).any(-1).sum()), flush=True)
                if tuning and mesh:
                    try:
                        blk, info = r.ctx.debug_read_queue_block()
                        for rep in range(3):                 # is the BLOCK itself incoherent between XCDs (mdvt_selftest.hip)?
                            co = r.ctx.debug_coherence()
                            print(f"pid {os.getpid()} iter {it}: coherence test {rep} on the bad render's block: {int(co[72])} wrong words"
                                  + (f"; pattern words by writer x reader XCD {co[:64].reshape(8, 8).tolist()}, atomic sums by reader {co[64:72].tolist()}, "
                                     f"first saw 0x{int(co[73]):08x} want 0x{int(co[74]):08x} at dword {int(co[75])}" if co[72] else ""), flush=True)
                        if blk.size: diagnose(r, blk, info, lambda: same(r.render(d, c, [p], want_depth=True)), f"pid {os.getpid()} iter {it}")
                    except Exception as e:      # the diagnosis must not hide the count
                        print("diagnosis failed:", repr(e), flush=True)
    print("case", cs["case"], W, H, "fresh" if fresh else "same ctx", "iters", iters, "bad", bad, flush=True)
    break
