"""The first render of a fresh context and the process-wide pools behind it.

r04's last soak lost queued triangles in one FIRST render of a fresh context in ~3 000 (and one process died of a GPU memory
fault) once a workspace block of the posed / converged mesh path had grown past 2 MB.  r05's diagnosis (DESIGN.md section 9,
profiles/r05_first_render_diagnosis.md): a block of >= 2 MB that a process frees and allocates again at the same address -- a
context created and destroyed per render -- is now and then seen through a STALE TRANSLATION by one of the eight XCDs: that XCD's
stores and atomics go to the pages the address had before, the other seven never see them, and a filled + synchronised block
does not help (the fill runs wherever it runs).  Below the HIP API; the library's part is not to create the condition: workspace
blocks live in a process-wide pool and are not unmapped while the process lives (csrc/mdvt_api.hip, ws_malloc / ws_free).

This file is the trip-wire: the stress that found it (tests/dbg_stress_case.py: the 100 x 31 mesh + convergence case of seed
504249, a new context per render, several processes sharing the GPU), on the product library and -- through the tuning library's
hooks -- on the r04 layout that failed (both huge lists inside the queue's block: 2.2 MB) with the pool in place.  Both must
render every first frame right.  (The failing condition itself, MDVT_WS_POOL=off, is a matter of chance -- 2 to 5 bad renders
and now and then a dead process in 30 000 contexts -- and is run by tools/fresh_context_ab.sh, not here.)
"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROCS, ITERS = 6, 2000


def _stress(extra_env, procs=PROCS, iters=ITERS):
    env = {k: v for k, v in os.environ.items() if not k.startswith("MDVT_")}
    env.update(MDVT_SWEEP_SEED="504249", MDVT_SWEEP_CASES="400", CASE="231", FRESH="1", ITERS=str(iters), OMP_NUM_THREADS="1")
    env.update(extra_env)
    ps = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dbg_stress_case.py")], cwd=REPO, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(procs)]
    outs = [p.communicate(timeout=900)[0] for p in ps]
    bad, finished = 0, 0
    for p, o in zip(ps, outs):
        m = re.search(r"^case 231 100 31 fresh iters (\d+) bad (\d+)$", o, re.M)
        if p.returncode == 0 and m and int(m.group(1)) == iters:
            finished += 1
            bad += int(m.group(2))
    return bad, finished, "\n".join(o[-1500:] for o in outs)


@pytest.mark.parametrize("variant", ["product", "r04_layout_with_the_pool"])
def test_first_render_of_fresh_contexts(variant):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    extra = {} if variant == "product" else {"MDVT_LIB_VARIANT": "tuning", "MDVT_WS_LAYOUT": "joint"}
    bad, finished, tail = _stress(extra)
    assert finished == PROCS, f"{PROCS - finished} of {PROCS} processes did not finish (a GPU fault kills the process):\n{tail}"
    assert bad == 0, f"{bad} first renders of {PROCS * ITERS} fresh contexts differ from the oracle:\n{tail}"


def test_pool_blocks_go_back_to_contexts_of_their_own_gpu_only(monkeypatch):
    """ADVICE r03 / VERDICT r04 item 7: both process-wide pools -- pinned + device parameter blocks, device workspace blocks --
    hand a block back only to a context of the GPU it was allocated on.  A box with one GPU cannot run two devices, so the tuning
    library lets a context claim another GPU's tag (MDVT_POOL_TAG: the pools' matching key, the device index in the product): a
    context tagged 1 must leave the idle blocks tagged 0 alone and give its own back under its tag, and a context tagged 0 then
    takes the tag-0 blocks, not the others."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("MDVT_LIB_VARIANT", "tuning")
    from metric_depth_video_toolbox_amd import _lib, stereo_rerender as sr, synthetic
    W, H = 200, 120
    d, c = synthetic.SyntheticScene(W, H, config_id=1, n_fg=5).frame(0)
    d, c = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()

    def one_context(tag):
        monkeypatch.setenv("MDVT_POOL_TAG", str(tag))
        r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)       # the general mesh path: workspace blocks of several sizes
        monkeypatch.delenv("MDVT_POOL_TAG")
        before = r.ctx.debug_pools()
        out = r.render(d, c, r.frame_params(xfov=45.0, convergence_distance=2.5), want_seed=True)
        torch.cuda.synchronize()
        during = r.ctx.debug_pools()
        sbs = out["sbs"].clone()
        return r, before, during, sbs

    _lib.release_cached_memory(-1)
    r0, b0, d0, ref = one_context(0)
    assert b0["tag"] == 0
    held = r0.ctx.workspace_bytes()
    assert held > 0
    r0.close()                                               # its blocks are idle now, tagged 0
    probe = _lib.Context(0, 16, 16)                          # (a context to ask through; tag 0)
    idle0 = probe.debug_pools()
    assert idle0["ws_mine"] > 0 and idle0["param_mine"] > 0, idle0

    r1, b1, d1, sbs1 = one_context(1)
    assert b1["tag"] == 1 and torch.equal(sbs1, ref)
    # seen from tag 1 the tag-0 blocks are "other": none of them was taken by the render
    assert d1["ws_other"] == idle0["ws_mine"] and d1["param_other"] == idle0["param_mine"], (idle0, b1, d1)
    assert d1["ws_mine"] == 0 and d1["param_mine"] == 0
    r1.close()
    after1 = probe.debug_pools()
    assert after1["ws_mine"] == idle0["ws_mine"] and after1["ws_other"] > 0 and after1["param_other"] > 0, after1

    r2, b2, d2, sbs2 = one_context(0)                        # tag 0 again: served from the tag-0 blocks, tag 1's untouched
    assert torch.equal(sbs2, ref)
    assert d2["ws_mine"] < after1["ws_mine"] and d2["param_mine"] < after1["param_mine"], (after1, d2)
    assert d2["ws_other"] == after1["ws_other"] and d2["param_other"] == after1["param_other"]
    r2.close()
    probe.close()
    _lib.release_cached_memory(-1)


def test_bank_resources_crowd(tmp_path):
    """The crowd that found both runtime bugs, on the path the r05 fix protects: 8 processes share the GPU, each
    makes 300 multi-frame calls that take the two-bank path (side stream + four events from the process-wide pool of
    `bank_res_take`), a context per call, native backtraces armed.  Every process must exit 0 with every call's bytes equal to its
    first call's, and the first calls must agree across processes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    procs, calls = 8, 300
    env = {k: v for k, v in os.environ.items() if not k.startswith("MDVT_")}
    env.update(CALLS=str(calls), MDVT_SEGV_TRACE_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    ps = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dbg_bank_crowd.py")], cwd=REPO, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(procs)]
    outs = [p.communicate(timeout=1200)[0] for p in ps]
    traces = "\n".join(open(os.path.join(tmp_path, f)).read()[-3000:] for f in sorted(os.listdir(tmp_path)) if f.endswith(".log"))
    tail = "\n".join(o[-800:] for o in outs) + "\n" + traces
    for p, o in zip(ps, outs):
        assert p.returncode == 0, f"a process died (rc {p.returncode}):\n{tail}"
        m = re.search(r"^bank crowd calls (\d+) bad (\d+)$", o, re.M)
        assert m and int(m.group(1)) == calls and int(m.group(2)) == 0, tail
    for mesh in (0, 1):
        firsts = [np.load(os.path.join(tmp_path, f)) for f in sorted(os.listdir(tmp_path)) if f.startswith(f"bank_crowd_first_{mesh}_")]
        assert len(firsts) == procs
        for f in firsts[1:]:
            assert np.array_equal(f["sbs"], firsts[0]["sbs"]) and np.array_equal(f["mask"], firsts[0]["mask"])


def test_cached_memory_limit_is_per_gpu_and_configurable():
    """the idle workspace the process keeps is capped per GPU, the cap is configurable, and a caller that is done
    can hand everything back.  Contexts of two frame sizes leave blocks of different size classes behind; lowering the limit returns
    the oldest blocks at once, 0 keeps nothing at all, and renders stay correct throughout (fresh blocks are filled before use)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from metric_depth_video_toolbox_amd import _lib, stereo_rerender as sr, synthetic
    dev = torch.cuda.current_device()
    _lib.release_cached_memory(-1)
    assert _lib.cached_memory(dev) == (0, 0)

    def render_once(W, H):
        d, c = synthetic.SyntheticScene(W, H, config_id=1, n_fg=5).frame(0)
        r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)
        out = r.render(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), r.frame_params(xfov=45.0, convergence_distance=2.5))
        torch.cuda.synchronize()
        held = r.ctx.workspace_bytes()
        sbs = out["sbs"].clone()
        r.close()
        return held, sbs

    try:
        h1, ref1 = render_once(200, 120)
        h2, ref2 = render_once(320, 180)
        idle, blocks = _lib.cached_memory(dev)
        assert idle >= max(h1, h2) and blocks > 0                      # both contexts' blocks wait in the pool (default cap: 4 GiB)
        _lib.set_cached_memory_limit(h2)                                # room for the second context's blocks only
        idle_b, _ = _lib.cached_memory(dev)
        assert idle_b <= h2 < idle
        _lib.set_cached_memory_limit(0)                                 # keep nothing
        assert _lib.cached_memory(dev) == (0, 0)
        h1b, again1 = render_once(200, 120)                             # every block fresh from the driver; nothing retained afterwards
        assert torch.equal(again1, ref1) and _lib.cached_memory(dev) == (0, 0)
        _lib.set_cached_memory_limit(4 << 30)
        _, again2 = render_once(320, 180)
        assert torch.equal(again2, ref2) and _lib.cached_memory(dev)[0] > 0
        r = sr.StereoRerenderer(64, 48, pupillary_distance=65)
        r.close(release_cached_memory=True)
        assert _lib.cached_memory(dev) == (0, 0)
    finally:
        _lib.set_cached_memory_limit(4 << 30)
