"""Renders one case of the randomised sweep many times (FRESH=1: a new context per render) and counts the renders that differ from
the oracle -- the replay that found what profiles/r04_soak_summary.md describes.  Debug helper, not collected by pytest.
  MDVT_SWEEP_SEED=504249 MDVT_SWEEP_CASES=400 CASE=231 FRESH=1 ITERS=2500 python tests/dbg_stress_case.py   (run 12 at once)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc
from test_gpu_render import sweep_cases
target = int(os.environ["CASE"]); iters = int(os.environ.get("ITERS", "1500")); fresh = os.environ.get("FRESH", "1") == "1"
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
    for it in range(iters):
        if fresh and it: r.close(); r = mk()
        got = r.render(d, c, [p], want_depth=True)
        if not (torch.equal(got["mask"][0], wm_t) and torch.equal(got["sbs"][0], wc_t)):
            bad += 1
            if bad <= 3: print("iter", it, "mask diffs", int((got["mask"][0] != wm_t).sum()), "rgb diffs", int((got["sbs"][0] != wc_t).any(-1).sum()), flush=True)
    print("case", cs["case"], W, H, "fresh" if fresh else "same ctx", "iters", iters, "bad", bad, flush=True)
    break
