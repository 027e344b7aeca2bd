"""Edge points (sr:589-606, 615-619, 727-735, 745-750, 838-858; dmt:1057-1060) on the GPU: the pixel each
vertex of a removed triangle is splatted on must be the one the reference's f64 chain of operations rounds to -- row AND
column, by construction: the general kernels evaluate the chain itself, the LDS row kernels of pure-shift frames an f32
estimate of the column wherever that provably rounds like the chain, the chain otherwise, and the source row except on the
scanlines the frame's camera matrix sends to the next row (k_edge_rows_exact).  Checked through the C ABI
(mdvt_edge_point_pixels) against tests/golden/edge_points.npz -- the loop body's statements run on the reference's own
functions -- and, at full HD for every vertex of a frame, against the NumPy restatement that the same golden pins."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IMIN = np.iinfo(np.int32).min


def _params(lib, K, scale, ang, T):
    p = lib.MdvtFrameParams()
    flat = np.asarray(K, np.float64).reshape(9)
    for k in range(9):
        p.K[k] = flat[k]
        p.Krender[k] = flat[k]
    p.depth_scale = scale
    p.convergence_angle = ang
    p.has_T = 0
    if T is not None:
        for k in range(16):
            p.T[k] = float(np.asarray(T).reshape(16)[k])
        p.has_T = 1
    return p


def _renderer(W, H, ipd_mm, pointcloud):
    from metric_depth_video_toolbox_amd.stereo_rerender import StereoRerenderer
    return StereoRerenderer(W, H, device=0, pupillary_distance=ipd_mm, render_as_pointcloud=pointcloud, infill_mask=True)


CASES = ["mesh_shift", "points_shift", "mesh_master", "mesh_conv", "points_conv_master", "mesh_pose", "mesh_pose_conv",
         "points_pose", "mesh_k_pow2", "points_k_pow2", "mesh_fy_below", "mesh_fy_above", "mesh_tall"]


@pytest.mark.parametrize("name", CASES)
def test_edge_point_pixels_are_the_reference_chains(golden, name):
    import torch
    from metric_depth_video_toolbox_amd import _lib
    from oracle import oracle_np as onp
    g = golden("edge_points")
    W, H, xfov, master, ipd_mm, pc, conv = g[name + "_par"]
    W, H, pc = int(W), int(H), bool(pc)
    T = g[name + "_T"]
    T = None if T.shape[0] == 0 else T
    K, scale = g[name + "_K"], float(g[name + "_scale"][0])
    ang = 0.0 if math.isnan(conv) else onp.convergence_angle(float(conv) * scale, ipd_mm / 1000)
    r = _renderer(W, H, int(ipd_mm), pc)
    p = _params(_lib, K, scale, ang, T)
    d_rgb = g[name + "_depth_rgb"]
    d = torch.from_numpy(d_rgb).cuda()
    idx = g[name + "_unused"]
    # the 89-degree filter's vertex set first (pinned on its own elsewhere; here: same set as the reference's run)
    _, unused = r.edge_filter(d, p)
    assert np.array_equal(np.flatnonzero(unused.cpu().numpy()), idx)
    live = (d_rgb.reshape(-1, 3)[idx][:, 0] != 0) | (d_rgb.reshape(-1, 3)[idx][:, 2] != 0)     # depth code 0 is not splatted (decree)
    hows = (0,) if (T is not None or ang != 0.0) else (0, 1)
    for how in hows:
        px = r.edge_point_pixels(d, p, how=how).cpu().numpy().astype(np.int64)
        for e, eye in enumerate("LR"):
            want = g[f"{name}_{eye}_px"]
            inside = (want[:, 0] >= 0) & (want[:, 0] < W) & (want[:, 1] >= 0) & (want[:, 1] < H) & live
            got = px[idx, e]
            assert np.array_equal(got[inside], want[inside]), (name, how, eye)
            assert np.all(got[~inside] == IMIN), (name, how, eye)
    r.close()


@pytest.mark.parametrize("pointcloud", [False, True])
@pytest.mark.parametrize("xfov", [45.0, 60.0, 97.3])
def test_every_vertex_of_a_full_hd_frame_rows_and_columns(pointcloud, xfov):
    """tests/report_edge_points_f64.py as an assertion: 1920x1080, EVERY vertex (2 073 600 x 2 eyes, not only the removed
    ones), pure shift: 0 row and 0 column differences between the reference's f64 chain and what the kernels take -- both
    ways of taking it (the chain per vertex; the row kernels' estimate + guard + deferred scanlines)."""
    import torch
    from oracle import c_oracle as orc
    from oracle import oracle_np as onp
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
    W, H = 1920, 1080
    r = _renderer(W, H, 65, pointcloud)
    p = r.frame_params(xfov=xfov)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    flip_rows = set()
    for cfg, t in ((2, 0), (3, 5)):
        d_rgb, _ = SyntheticScene(W, H, config_id=cfg).frame(t)
        depth = orc.decode_depth(d_rgb, 100.0, p.depth_scale)
        want, _, _ = onp.edge_point_chain(depth, K, K, W, H, not pointcloud, 0.065)
        live = depth.reshape(-1) > 1e-4
        inside = (want[..., 0] >= 0) & (want[..., 0] < W) & (want[..., 1] >= 0) & (want[..., 1] < H) & live[:, None]
        d = torch.from_numpy(d_rgb).cuda()
        for how in (0, 1):
            got = r.edge_point_pixels(d, p, how=how).cpu().numpy().astype(np.int64)
            col_bad = int(((got[..., 0] != want[..., 0]) & inside).sum())
            row_bad = int(((got[..., 1] != want[..., 1]) & inside).sum())
            assert (col_bad, row_bad) == (0, 0), (cfg, how, col_bad, row_bad)
            assert np.all(got[~inside] == IMIN)
        rows = np.arange(H * W) // W
        flip_rows |= set(rows[(want[:, 0, 1] != rows) & inside[:, 0]].tolist())
    if xfov == 45.0 and not pointcloud:
        assert flip_rows == {0}, "with this camera matrix the vertices of source row 0 (and only they) land on row 1"
    r.close()


@pytest.mark.parametrize("case", ["conv", "pose", "pose_conv_master"])
def test_every_vertex_of_a_general_full_hd_frame(case):
    """The same for converged and posed frames (the general kernels' path): every vertex, both eyes."""
    import torch
    from oracle import c_oracle as orc
    from oracle import oracle_np as onp
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene, synthetic_pose_track
    from metric_depth_video_toolbox_amd.stereo_rerender import StereoRerenderer
    W, H = 1920, 1080
    master = 45.0 if case != "pose_conv_master" else 60.0
    r = StereoRerenderer(W, H, device=0, pupillary_distance=65, infill_mask=True, master_xfov=master)
    T = None if case == "conv" else synthetic_pose_track(80)[79]
    p = r.frame_params(xfov=45.0, convergence_distance=None if case == "pose" else 2.5, transformation=T)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    d_rgb, _ = SyntheticScene(W, H, config_id=3).frame(2)
    depth = orc.decode_depth(d_rgb, 100.0, p.depth_scale)
    want, _, _ = onp.edge_point_chain(depth, K, K, W, H, True, 0.065, p.convergence_angle, T)
    inside = (want[..., 0] >= 0) & (want[..., 0] < W) & (want[..., 1] >= 0) & (want[..., 1] < H) & (depth.reshape(-1) > 1e-4)[:, None]
    got = r.edge_point_pixels(torch.from_numpy(d_rgb).cuda(), p, how=0).cpu().numpy().astype(np.int64)
    assert inside.sum() > 3_000_000
    assert int(((got != want).any(axis=-1) & inside).sum()) == 0
    assert np.all(got[~inside] == IMIN)
    r.close()
