"""Replays a SEQUENCE of cases of test_randomised_parity_sweep in one process (CASES="148,150"; MDVT_SWEEP_* as in the failing job),
each with sentinel-filled outputs, and reports which differ from the oracle.  Debug helper for state carried between cases."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc
from test_gpu_render import sweep_cases

targets = [int(v) for v in os.environ.get("CASES", "0").split(",")]
sentinel = os.environ.get("SENTINEL", "1") == "1"
for cs in sweep_cases(synthetic):
    if cs["case"] not in targets:
        continue
    W, H, mesh, infill, T = cs["W"], cs["H"], cs["mesh"], cs["infill"], cs["T"]
    depth_rgb, color = cs["depth_rgb"], cs["color"]
    r = sr.StereoRerenderer(W, H, pupillary_distance=cs["ipd"], max_depth=cs["max_depth"], master_xfov=cs["master"],
                            render_as_pointcloud=not mesh, infill_mask=infill, dont_place_points_in_edges=cs["no_pts"])
    p = r.frame_params(xfov=cs["xfov"], convergence_distance=cs["conv_d"], transformation=T)
    pre = {}
    if sentinel:
        pre = dict(out_sbs=torch.full((1, H, 2 * W, 3), 7, dtype=torch.uint8, device="cuda"), out_mask=torch.full((1, H, 2 * W), 7, dtype=torch.uint8, device="cuda"),
                   out_depth=torch.full((1, H, 2 * W), -7.0, dtype=torch.float32, device="cuda"))
    got = r.render(torch.from_numpy(depth_rgb).cuda()[None], torch.from_numpy(color).cuda()[None], [p], want_depth=True, **pre)
    got = {k: v[0] for k, v in got.items()}
    torch.cuda.synchronize()
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(W, H, K, ipd_m=cs["ipd"] / 1000, max_depth=cs["max_depth"], depth_scale=p.depth_scale,
                         mode=orc.MODE_MESH if mesh else orc.MODE_POINTS, remove_edges=r.remove_edges, edge_points=int(r.edge_points),
                         conv_angle=p.convergence_angle, T=T, key_rgb=r.key_rgb)
    want = orc.render_stereo(op, depth_rgb, color, want_depth=True)
    m = got["mask"].cpu().numpy(); c = got["sbs"].cpu().numpy()
    bad = 0
    for eye, sl in (("left", slice(0, W)), ("right", slice(W, 2 * W))):
        bad += int((m[:, sl] != want[eye + "_mask"]).sum()) + int(np.any(c[:, sl] != want[eye + "_rgb"], -1).sum())
    print(f"case {cs['case']} {W}x{H} mesh={mesh} infill={infill} kind={cs['kind']}: differing px {bad}, never written {int((got['mask'] == 7).sum())}", flush=True)
    r.close()
