"""Replays one case of test_randomised_parity_sweep (MDVT_SWEEP_SEED / MDVT_SWEEP_CASES / CASE) and prints where the
device and the oracle differ.  Debug helper, not collected by pytest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc

seed = int(os.environ.get("MDVT_SWEEP_SEED", "1")); n_cases = int(os.environ.get("MDVT_SWEEP_CASES", "400")); target = int(os.environ.get("CASE", "109"))
rng = np.random.default_rng(seed)
sizes = [(2, 2), (3, 2), (4, 4), (5, 3), (8, 8), (17, 9), (36, 20), (61, 33), (64, 32), (100, 31), (128, 16), (200, 12)]
if n_cases > 60:
    sizes += [(320, 200), (257, 129), (96, 96)]
for case in range(n_cases):
    W, H = sizes[int(rng.integers(len(sizes)))]
    mesh = bool(rng.integers(2)); infill = bool(rng.integers(2)); no_pts = bool(rng.integers(4) == 0)
    ipd = int(rng.choice([0, 1, 30, 63, 65, 120, 400])); xfov = float(rng.choice([20.0, 45.0, 60.0, 90.0, 120.0]))
    master = float(rng.choice([25.0, 45.0, 70.0])); max_depth = int(rng.choice([5, 20, 100, 655])); kind = int(rng.integers(4))
    depth_rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    style = int(rng.integers(4))
    if style == 0:
        code = (2000 + 40 * np.arange(W)[None, :] + 7 * np.arange(H)[:, None]).astype(np.uint32)
        code[:, W // 2:] //= 3
        depth_rgb[..., 0] = (code >> 8) & 0xFF; depth_rgb[..., 2] = code & 0xFF
    elif style == 1:
        depth_rgb[..., 0] = 0; depth_rgb[..., 2] = rng.integers(0, 4, (H, W))
    elif style == 2:
        depth_rgb[..., 0] = 3; depth_rgb[..., 2] = 77
    color = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if rng.integers(2):
        color[rng.integers(H), rng.integers(W)] = (0, 255, 0) if infill else (0, 0, 0)
    T = None
    if kind >= 2:
        T = synthetic.synthetic_pose_track(64)[int(rng.integers(1, 64))]
        T[:3, 3] *= float(rng.choice([1.0, 20.0]))
    conv = float(rng.uniform(0.3, 8.0)) if kind in (1, 3) else None
    if case != target:
        continue
    r = sr.StereoRerenderer(W, H, pupillary_distance=ipd, max_depth=max_depth, master_xfov=master,
                            render_as_pointcloud=not mesh, infill_mask=infill, dont_place_points_in_edges=no_pts)
    p = r.frame_params(xfov=xfov, convergence_distance=conv, transformation=T)
    got = r.render(torch.from_numpy(depth_rgb).cuda(), torch.from_numpy(color).cuda(), p, want_depth=True)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(W, H, K, ipd_m=ipd / 1000, max_depth=max_depth, depth_scale=p.depth_scale,
                         mode=orc.MODE_MESH if mesh else orc.MODE_POINTS, remove_edges=r.remove_edges, edge_points=int(r.edge_points),
                         conv_angle=p.convergence_angle, T=T, key_rgb=r.key_rgb)
    want = orc.render_stereo(op, depth_rgb, color, want_depth=True)
    print(f"case {case}: {W}x{H} mesh={mesh} infill={infill} no_pts={no_pts} ipd={ipd} xfov={xfov} master={master} md={max_depth} kind={kind} style={style} scale={p.depth_scale}")
    m = got["mask"].cpu().numpy(); c = got["sbs"].cpu().numpy(); z = got["depth"].cpu().numpy()
    zs = orc.decode_depth(depth_rgb, max_depth, p.depth_scale)
    tri, unused, _ = orc.edge_filter(zs, K, mesh)
    gt, gu = r.edge_filter(torch.from_numpy(depth_rgb).cuda(), p)
    print("edge filter equal:", np.array_equal(gt.cpu().numpy(), tri), np.array_equal(gu.cpu().numpy(), unused))
    for eye, sl in (("left", slice(0, W)), ("right", slice(W, 2 * W))):
        dm = np.argwhere(m[:, sl] != want[eye + "_mask"])
        dc = np.argwhere(np.any(c[:, sl] != want[eye + "_rgb"], -1))
        print(eye, "mask diffs", dm.tolist()[:10], "rgb diffs", dc.tolist()[:10])
        for (y, x) in dm[:4]:
            print("  at", y, x, "got mask", m[y, sl][x], "want", want[eye + "_mask"][y, x], "got rgb", c[y, sl][x], "want rgb", want[eye + "_rgb"][y, x],
                  "got z", z[y, sl][x], "want z", want[eye + "_depth"][y, x])
            y0, y1, x0, x1 = max(0, y - 2), min(H, y + 2), max(0, x - 2), min(W, x + 3)
            print("  src codes around:\n", depth_rgb[y0:y1, x0:x1, 2], "\n  z:\n", zs[y0:y1, x0:x1])
            ncell = (W - 1) * (H - 1)
            for yy in range(max(0, y - 1), min(H - 1, y + 1)):
                print("   tri_invalid row", yy, "t1", tri[yy * (W - 1) + x0: yy * (W - 1) + x1].tolist(), "t2", tri[ncell + yy * (W - 1) + x0: ncell + yy * (W - 1) + x1].tolist())
    break
