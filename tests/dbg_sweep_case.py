"""Replays one case of test_randomised_parity_sweep (MDVT_SWEEP_SEED / MDVT_SWEEP_CASES / MDVT_SWEEP_SIZES as in the failing
job, CASE = its number) and prints where the device and the oracle differ.  Debug helper, not collected by pytest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc
from test_gpu_render import sweep_cases

target = int(os.environ.get("CASE", "0"))
for cs in sweep_cases(synthetic):
    if cs["case"] != target:
        continue
    W, H, mesh, infill, T = cs["W"], cs["H"], cs["mesh"], cs["infill"], cs["T"]
    depth_rgb, color = cs["depth_rgb"], cs["color"]
    r = sr.StereoRerenderer(W, H, pupillary_distance=cs["ipd"], max_depth=cs["max_depth"], master_xfov=cs["master"],
                            render_as_pointcloud=not mesh, infill_mask=infill, dont_place_points_in_edges=cs["no_pts"])
    p = r.frame_params(xfov=cs["xfov"], convergence_distance=cs["conv_d"], transformation=T)
    pre = dict(out_sbs=torch.full((1, H, 2 * W, 3), 7, dtype=torch.uint8, device="cuda"), out_mask=torch.full((1, H, 2 * W), 7, dtype=torch.uint8, device="cuda"),
               out_depth=torch.full((1, H, 2 * W), -7.0, dtype=torch.float32, device="cuda"))
    got = r.render(torch.from_numpy(depth_rgb).cuda()[None], torch.from_numpy(color).cuda()[None], [p], want_depth=True, **pre)
    got = {k: v[0] for k, v in got.items()}
    torch.cuda.synchronize()
    print("pixels the device never wrote (sentinel 7 left in the mask):", int((got["mask"] == 7).sum()), "of", 2 * W * H)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(W, H, K, ipd_m=cs["ipd"] / 1000, max_depth=cs["max_depth"], depth_scale=p.depth_scale,
                         mode=orc.MODE_MESH if mesh else orc.MODE_POINTS, remove_edges=r.remove_edges, edge_points=int(r.edge_points),
                         conv_angle=p.convergence_angle, T=T, key_rgb=r.key_rgb)
    orc.stats_reset()
    want = orc.render_stereo(op, depth_rgb, color, want_depth=True)
    print({k: v for k, v in cs.items() if k not in ("depth_rgb", "color")}, "scale", p.depth_scale, "conv angle", p.convergence_angle)
    print("oracle stats", orc.stats())
    m = got["mask"].cpu().numpy(); c = got["sbs"].cpu().numpy(); z = got["depth"].cpu().numpy()
    zs = orc.decode_depth(depth_rgb, cs["max_depth"], p.depth_scale)
    print("decoded z range", float(zs.min()), float(zs.max()))
    for eye, sl in (("left", slice(0, W)), ("right", slice(W, 2 * W))):
        dm = np.argwhere(m[:, sl] != want[eye + "_mask"])
        dc = np.argwhere(np.any(c[:, sl] != want[eye + "_rgb"], -1))
        print(eye, "mask diffs", len(dm), dm.tolist()[:6], "rgb diffs", len(dc), dc.tolist()[:6], "holes got/want", int((m[:, sl] > 0).sum()), int((want[eye + "_mask"] > 0).sum()))
        for (y, x) in dm[:3]:
            print("  at", y, x, "got mask", m[y, sl][x], "want", want[eye + "_mask"][y, x], "got rgb", c[y, sl][x], "want rgb", want[eye + "_rgb"][y, x],
                  "got z", z[y, sl][x], "want z", want[eye + "_depth"][y, x])
    for env in ("MDVT_FORCE_GLOBAL", "MDVT_MESH_CONV"):
        os.environ[env] = "1"
        alt = r.render(torch.from_numpy(depth_rgb).cuda(), torch.from_numpy(color).cuda(), p, want_depth=True)
        print(env, "changes the device result:", not (torch.equal(alt["mask"], got["mask"]) and torch.equal(alt["sbs"], got["sbs"])))
        del os.environ[env]
    break
