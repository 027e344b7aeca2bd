#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference's own pure-NumPy functions.

Run ONLY in the build container (needs /root/reference).  Nothing of the reference travels: the
fixtures hold inputs and the reference functions' outputs, nothing else.  cv2 / open3d / OpenGL /
glfw are absent here, so empty stub modules are registered before the import; only
functions that never touch those packages are called.

    python tests/golden/gen_golden.py            # rewrites the fixtures in tests/golden/

Recorded with every fixture: numpy / scipy versions (the unprojection dtype is NumPy-major
dependent).
"""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("cv2"); _stub("glfw"); _stub("OpenGL"); _stub("OpenGL.GL")
    _stub("OpenGL.GL.shaders", compileProgram=None, compileShader=None)

    class _TriangleMesh:  # ndarray-backed stand-in for o3d.geometry.TriangleMesh
        def __init__(s):
            s.vertices = np.zeros((0, 3)); s.triangles = np.zeros((0, 3), np.int32); s.vertex_colors = np.zeros((0, 3))

        def transform(s, T):
            pass

    o3d = _stub("open3d")
    o3d.geometry = types.SimpleNamespace(TriangleMesh=_TriangleMesh)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.array(a, np.float64),
                                        Vector3iVector=lambda a: np.array(a, np.int32))
    sys.path.insert(0, REF)
    import depth_frames_helper as dfh
    import depth_map_tools as dmt
    import stereo_rerender as sr
    import infill_common as ic
    return dfh, dmt, sr, ic



# ---------------------------------------------------------------------------------------------------------------------
# edge points (sr:589-606, 615-619, 707-735, 745-756, 831-858): where the vertices of removed triangles are splatted
# ---------------------------------------------------------------------------------------------------------------------
class _O3dPoints:
    """Stand-in for o3d.geometry.PointCloud with the three in-place operations the loop body uses, restated from
    Open3D's published Geometry3D::TransformPoints / RotatePoints / TranslatePoints in f64: a 4x4 times (x, y, z, 1)
    divided by its w; R (p - centre) + centre; p += t.  Sums run left to right (Eigen's own order inside a 3- or
    4-term dot product is not observable here: it moves a coordinate by an ulp of f64, i.e. the pixel of a point
    that sits within ~1e-13 px of a rounding tie)."""

    def __init__(s):
        s.points = np.zeros((0, 3))

    def transform(s, T):
        T = np.asarray(T, np.float64)
        p = np.asarray(s.points)
        h = [((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3] * 1.0 for r in range(4)]
        s.points = np.stack([h[0] / h[3], h[1] / h[3], h[2] / h[3]], axis=1)
        return s

    def rotate(s, R, center=(0, 0, 0)):
        R, c = np.asarray(R, np.float64), np.asarray(center, np.float64)
        d = np.asarray(s.points) - c
        s.points = np.stack([((R[r, 0] * d[:, 0] + R[r, 1] * d[:, 1]) + R[r, 2] * d[:, 2]) + c[r] for r in range(3)], axis=1)
        return s

    def translate(s, t, relative=True):
        assert relative
        s.points = np.asarray(s.points) + np.asarray(t, np.float64)
        return s


def _rotation_matrix_from_xyz(rot):
    """open3d get_rotation_matrix_from_xyz = Rx(a) Ry(b) Rz(c) with the textbook matrices (for (0, b, 0): Ry(b) itself,
    the products with the identity being exact)."""
    a, b, c = (float(v) for v in rot)
    Rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    Ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    Rz = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def _project_points(obj, rvec, tvec, K, dist):
    """cv2.projectPoints by its published arithmetic (cvProjectPoints2): everything is converted to double first (so a
    float32 camera matrix contributes f64(f32(fx)) ...), R = Rodrigues(0) = I and t = 0 leave the point as it is,
    z = z ? 1/z : 1, x *= z, y *= z, zero distortion coefficients leave x and y as they are, u = x fx + cx."""
    assert not np.any(np.asarray(rvec)) and not np.any(np.asarray(tvec)) and not np.any(np.asarray(dist))
    K = np.asarray(K).astype(np.float64)
    P = np.asarray(obj, np.float64).reshape(-1, 3)
    z = P[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        iz = np.where(z != 0.0, 1.0 / z, 1.0)
    x, y = P[:, 0] * iz, P[:, 1] * iz
    return np.stack([x * K[0, 0] + K[0, 2], y * K[1, 1] + K[1, 2]], axis=1).reshape(-1, 1, 2), None


def edge_point_goldens(dfh, dmt, sr, meta):
    """The loop body's own statements for the edge points, run on the reference's functions (get_mesh_from_depth_map,
    pts_2_pcd, project_3d_points_to_2d, convergence_angle) with Open3D's point cloud and cv2.projectPoints standing in
    as above.  Recorded per case and eye, for EVERY vertex of a removed triangle in the order of `unused_indices`:
    np.round(points_2d) (sr:746, 858), the depth the painter's order sorts by (sr:752), and the un-normalised
    `unprojected_normals` (sr:733, 849)."""
    import cv2 as cv2_stub
    import open3d as o3d_stub
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene, synthetic_pose_track
    cv2_stub.projectPoints = _project_points
    o3d_stub.geometry.PointCloud = _O3dPoints

    def run_case(W, H, depth_rgb, color, xfov, K, master_xfov, ipd_mm, pointcloud, conv_dist, T, max_depth=100):
        frame_width, frame_height = W, H
        cam_matrix = dmt.compute_camera_matrix(xfov, None, W, H) if K is None else np.array(K, np.float64)
        render_cam_matrix = cam_matrix
        depth = dfh.decode_rgb_depth_frame(depth_rgb, max_depth, True)
        scale = 1.0
        if master_xfov is not None:                                           # sr:537-541
            scale = 1.0 / (math.tan(math.radians(master_xfov / 2)) / math.tan(math.radians(xfov / 2)))
            depth *= scale
        left_shift = -(ipd_mm / 1000) / 2                                      # sr:458-459
        right_shift = +(ipd_mm / 1000) / 2
        mesh, unused_indices, removed_normals = dmt.get_mesh_from_depth_map(
            depth, cam_matrix, color, None, remove_edges=True, of_by_one=not pointcloud, return_normals_of_removed=True)
        in_edge = np.zeros(len(mesh.vertices), dtype=bool)
        in_edge[unused_indices] = True
        edge_points = np.asarray(mesh.vertices)[in_edge]                       # (a copy: boolean indexing)
        world_normals = removed_normals + edge_points                          # sr:596, before the undo
        edge_points[:, 0] *= (frame_width - 1) / frame_width                   # sr:599-600
        edge_points[:, 1] *= (frame_height - 1) / frame_height
        edge_pcd, normal_pcd = dmt.pts_2_pcd(edge_points), dmt.pts_2_pcd(world_normals)
        if T is not None:                                                      # sr:615-619
            edge_pcd.transform(np.array(T)); normal_pcd.transform(np.array(T))
        rot_plus = rot_minus = None
        if conv_dist is not None:                                              # sr:707-722
            cd = float(conv_dist) * scale
            ang = sr.convergence_angle(cd, ipd_mm / 1000)
            rot_plus, rot_minus = _rotation_matrix_from_xyz((0, ang, 0)), _rotation_matrix_from_xyz((0, -ang, 0))
        out = {}
        # left eye, sr:727-735, 745-752
        if rot_minus is not None:
            edge_pcd.rotate(rot_minus, center=(0, 0, 0)); normal_pcd.rotate(rot_minus, center=(0, 0, 0))
        edge_pcd.translate([-left_shift, 0.0, 0.0]); normal_pcd.translate([-left_shift, 0.0, 0.0])
        for eye in ("L", "R"):
            if eye == "R":                                                     # sr:838-850: on top of the left eye's state
                edge_pcd.translate([left_shift, 0.0, 0.0]); normal_pcd.translate([left_shift, 0.0, 0.0])
                if rot_plus is not None:
                    for _ in range(2):
                        edge_pcd.rotate(rot_plus, center=(0, 0, 0)); normal_pcd.rotate(rot_plus, center=(0, 0, 0))
                edge_pcd.translate([-right_shift, 0.0, 0.0]); normal_pcd.translate([-right_shift, 0.0, 0.0])
            unprojected_normals = np.asarray(normal_pcd.points) - np.asarray(edge_pcd.points)
            points_3d = np.asarray(edge_pcd.points)
            points_2d = dmt.project_3d_points_to_2d(points_3d, render_cam_matrix)
            with np.errstate(invalid="ignore"):
                pr = np.round(points_2d)
            # (astype(int) of a non-finite or huge value is undefined: such points fail the reference's bounds test anyway)
            ok = np.all(np.isfinite(pr) & (np.abs(pr) < 2.0 ** 30), axis=1)
            out[eye + "_px"] = np.where(ok[:, None], pr, -(2.0 ** 30)).astype(np.int64)
            out[eye + "_z"] = points_3d[:, 2].copy()
            out[eye + "_n"] = unprojected_normals.copy()
        out["unused"] = np.flatnonzero(in_edge).astype(np.int64)
        out["K"] = cam_matrix
        out["scale"] = np.array([scale], np.float64)
        return out

    def scene(W, H, seed, zero_patch=False):
        d, c = SyntheticScene(W, H, seed=seed, n_fg=6).frame(0)
        d = d.copy()
        d[:, W // 3] = d[:, W // 3 + 1] = (3, 3, 0)          # a near pole over the full height: edges in rows 0 and H-1
        d[H // 2, :] = (5, 5, 128)                            # and a wire over the full width: edges in columns 0 and W-1
        if zero_patch:
            d[4:7, 9:12] = 0
        return d, c

    # a focal length whose f32 rounding is BELOW / ABOVE it by most of half an ulp (the rows that flip depend on the sign), found by search
    def xfov_with_delta(W, H, want_negative):
        best = None
        for k in range(4000):
            xf = 40.0 + k * 0.01
            Kc = dmt.compute_camera_matrix(xf, None, W, H)
            rel = (float(np.float32(Kc[1, 1])) - Kc[1, 1]) / Kc[1, 1]
            if (rel < 0) == want_negative and (best is None or abs(rel) > abs(best[1])):
                best = (xf, rel)
        return best[0]

    cases = {}
    pose = synthetic_pose_track(60)[59]
    big_pose = np.eye(4); a = math.radians(4.0)
    big_pose[:3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]) @ \
        np.array([[1, 0, 0], [0, math.cos(a / 2), -math.sin(a / 2)], [0, math.sin(a / 2), math.cos(a / 2)]])
    big_pose[:3, 3] = (0.03, -0.02, 0.05)
    spec = {
        # name: W, H, seed, xfov, K, master_xfov, ipd_mm, pointcloud, convergence distance, pose, zero patch
        "mesh_shift": (64, 48, 21, 45.0, None, None, 65, False, None, None, False),
        "points_shift": (64, 48, 22, 45.0, None, None, 65, True, None, None, True),
        "mesh_master": (48, 32, 23, 60.0, None, 45.0, 63, False, None, None, False),
        "mesh_conv": (64, 48, 24, 45.0, None, None, 65, False, 2.0, None, False),
        "points_conv_master": (64, 48, 25, 50.0, None, 45.0, 65, True, 1.5, None, False),
        "mesh_pose": (64, 48, 26, 45.0, None, None, 65, False, None, pose, False),
        "mesh_pose_conv": (48, 32, 27, 45.0, None, None, 65, False, 3.0, big_pose, True),
        "points_pose": (48, 32, 28, 70.0, None, None, 65, True, None, big_pose, False),
        # fy exactly representable in f32: in exact arithmetic source row 0 sits ON the tie 0.5, the f64 roundings of each point decide
        "mesh_k_pow2": (64, 48, 29, None, [[512.0, 0, 32.0], [0, 512.0, 24.0], [0, 0, 1]], None, 65, False, None, None, False),
        "points_k_pow2": (64, 48, 30, None, [[512.0, 0, 32.0], [0, 512.0, 24.0], [0, 0, 1]], None, 65, True, None, None, False),
        "mesh_fy_below": (96, 64, 31, xfov_with_delta(96, 64, True), None, None, 65, False, None, None, False),
        "mesh_fy_above": (96, 64, 32, xfov_with_delta(96, 64, False), None, None, 65, False, None, None, False),
        "mesh_tall": (16, 400, 33, xfov_with_delta(16, 400, True), None, None, 65, False, None, None, False),
    }
    for name, (W, H, seed, xfov, K, master, ipd, pc, conv, T, zp) in spec.items():
        d, c = scene(W, H, seed, zp)
        r = run_case(W, H, d, c, xfov, K, master, ipd, pc, conv, T)
        cases[name + "_depth_rgb"] = d
        cases[name + "_par"] = np.array([W, H, np.nan if xfov is None else xfov, np.nan if master is None else master, ipd,
                                         1.0 if pc else 0.0, np.nan if conv is None else conv], np.float64)
        cases[name + "_T"] = np.zeros((0, 4)) if T is None else np.array(T, np.float64)
        for k, v in r.items():
            cases[f"{name}_{k}"] = v
    # full HD, the camera of the benchmark: pixels only, for every removed vertex (int16) -- rows 0, 1, H-1, columns 0, W-1 included
    W, H = 1920, 1080
    d, c = SyntheticScene(W, H, config_id=2).frame(3)
    d = d.copy(); d[:, 700] = d[:, 701] = (3, 3, 0); d[600, :] = (5, 5, 128)
    r = run_case(W, H, d, c, 45.0, None, None, 65, False, None, None)
    cases["hd_frame"] = np.array([2, 3], np.int64)          # SyntheticScene(1920, 1080, config_id=2).frame(3) + the pole and the wire above
    cases["hd_unused"] = r["unused"].astype(np.int32)
    for eye in "LR":
        cases[f"hd_{eye}_px"] = np.clip(r[eye + "_px"], -32768, 32767).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "edge_points.npz"), meta=json.dumps(meta), names=np.array(sorted(spec)), **cases)


def main():
    import scipy
    dfh, dmt, sr, ic = import_reference()
    sys.path.insert(0, REPO)
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene

    meta = {"numpy": np.__version__, "scipy": scipy.__version__, "reference_snapshot": "2026-05-15"}
    rng = np.random.default_rng(20260927)
    if "--only-edge-points" in sys.argv:
        edge_point_goldens(dfh, dmt, sr, meta)
        print("edge_points.npz", os.path.getsize(os.path.join(HERE, "edge_points.npz")), "bytes")
        return

    # ------------------------------------------------------------------ codec
    kat_rgb = np.array([[0, 0, 0], [0, 0, 1], [0, 9, 1], [1, 1, 0], [2, 2, 133], [252, 252, 5],
                        [255, 255, 255], [255, 0, 0], [0, 255, 255], [128, 7, 64]], np.uint8).reshape(1, -1, 3)
    kat_dec = {str(md): dfh.decode_rgb_depth_frame(kat_rgb, md, True) for md in (100, 20, 655)}
    rnd_rgb = rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)
    rnd_dec = dfh.decode_rgb_depth_frame(rnd_rgb, 100, True)
    enc_in = np.concatenate([np.array([1.0, 2.5, 0.0015, 100.0, 150.0, -3.0, 0.0, 99.9999, 1.5499e-3, 1.55e-3], np.float32),
                             rng.uniform(0, 110, 86).astype(np.float32)]).reshape(8, 12)
    enc_u32 = dfh.encode_depth_as_uint32(enc_in, 100)
    enc_bgr = dfh.encode_data_as_BGR(enc_u32, 12, 8, bit16=True)
    enc_u32_20 = dfh.encode_depth_as_uint32(enc_in, 20)
    # round trip through the decoder (what a depth video delivers)
    rt = dfh.decode_rgb_depth_frame(np.ascontiguousarray(enc_bgr[..., ::-1]), 100, True)
    np.savez_compressed(os.path.join(HERE, "codec.npz"),
                        kat_rgb=kat_rgb, kat_dec_100=kat_dec["100"], kat_dec_20=kat_dec["20"], kat_dec_655=kat_dec["655"],
                        rnd_rgb=rnd_rgb, rnd_dec=rnd_dec, enc_in=enc_in, enc_u32=enc_u32, enc_bgr=enc_bgr,
                        enc_u32_20=enc_u32_20, roundtrip=rt, meta=json.dumps(meta))

    # ------------------------------------------------------------------ camera / scalars
    cam_cases = [(45, None, 640, 480), (45, None, 1920, 1080), (45, None, 3840, 2160), (60, None, 1920, 1080),
                 (None, 30, 1920, 1080), (70, 40, 1280, 720), (90.5, None, 16, 12), (45, None, 3, 2)]
    cam_in = np.array([[np.nan if a is None else a, np.nan if b is None else b, w, h] for a, b, w, h in cam_cases], np.float64)
    cam_K = np.stack([dmt.compute_camera_matrix(a, b, w, h) for a, b, w, h in cam_cases])
    cam_fov = np.array([dmt.fov_from_camera_matrix(K) for K in cam_K], np.float64)
    conv_in = np.array([[2.0, 0.065], [0.5, 0.063], [10.0, 0.065], [1e-3, 0.065], [100.0, 0.07]], np.float64)
    conv_out = np.array([sr.convergence_angle(d, p) for d, p in conv_in], np.float64)
    cos89 = np.array([np.cos(np.radians(89.0))], np.float64)
    nan_series = [float("nan"), 2.0, float("nan"), float("nan"), 3.5, 4.0, float("nan")]
    nan_filled = sr.fill_nan_with_closest(list(nan_series))
    series = {}
    for name, n in (("s7", 7), ("s120", 120), ("s300", 300)):
        y = (3.0 + np.sin(np.arange(n) / 9.0) + 0.1 * rng.standard_normal(n)).tolist()
        series[name + "_in"] = np.array(y, np.float64)
        series[name + "_out"] = np.asarray(sr.curve_fit(list(y)), np.float64)
    np.savez_compressed(os.path.join(HERE, "camera.npz"), cam_in=cam_in, cam_K=cam_K, cam_fov=cam_fov,
                        conv_in=conv_in, conv_out=conv_out, cos89=cos89,
                        nan_in=np.array(nan_series, np.float64), nan_out=np.array(nan_filled, np.float64),
                        meta=json.dumps(meta), **series)

    # ------------------------------------------------------------------ unprojection + mesh / edge filter
    geo = {}
    tiny = np.array([[2, 2, 4], [2, 1, 4]], np.float32)
    Kt = dmt.compute_camera_matrix(45, None, 3, 2)
    geo["tiny_depth"] = tiny
    geo["tiny_K"] = Kt
    for obo in (False, True):
        pts, h, w = dmt.create_point_cloud_from_depth(tiny, Kt, obo)
        geo[f"tiny_pts_obo{int(obo)}"] = pts

    scenes = {
        # name: (W, H, xfov, scene seed, max_depth, zero-depth patch)
        "a": (48, 32, 45.0, 11, 100, False),
        "b": (64, 48, 60.0, 12, 100, True),
        "c": (40, 24, 45.0, 13, 20, False),
    }
    for name, (W, H, xfov, seed, md, zero_patch) in scenes.items():
        sc = SyntheticScene(W, H, seed=seed, n_fg=5)
        depth_rgb, color = sc.frame(0, md)
        if zero_patch:
            depth_rgb[5:8, 9:13] = 0          # Z = 0 vertices (SURVEY.md 9a "zero depth")
            depth_rgb[20, 30] = (0, 0, 1)     # one LSB
        K = dmt.compute_camera_matrix(xfov, None, W, H)
        depth = dfh.decode_rgb_depth_frame(depth_rgb, md, True)
        geo[f"{name}_depth_rgb"] = depth_rgb
        geo[f"{name}_color"] = color
        geo[f"{name}_K"] = K
        geo[f"{name}_max_depth"] = np.array([md], np.float64)
        geo[f"{name}_depth"] = depth
        for obo in (False, True):
            pts, _, _ = dmt.create_point_cloud_from_depth(depth, K, obo)
            geo[f"{name}_pts_obo{int(obo)}"] = pts
            assert pts.dtype == np.float64, "fixture assumes NumPy >= 2 promotion (SURVEY.md 9 quirk 5)"
            mesh, unused, normals = dmt.get_mesh_from_depth_map(depth, K, color, None, remove_edges=True,
                                                                of_by_one=obo, return_normals_of_removed=True)
            tris = np.asarray(mesh.triangles)
            # zeroed (0,0,0) == removed (dmt:1372); a real triangle never has three equal indices
            geo[f"{name}_tri_invalid_obo{int(obo)}"] = np.all(tris == 0, axis=1).copy()
            geo[f"{name}_unused_obo{int(obo)}"] = np.asarray(unused, np.int64)
            geo[f"{name}_removed_normals_obo{int(obo)}"] = np.asarray(normals, np.float64)
            geo[f"{name}_colors_obo{int(obo)}"] = np.array(mesh.vertex_colors, np.float64, copy=True)
            # mesh re-use path (dmt:1264-1269): second frame through the same mesh object
            depth_rgb2, color2 = sc.frame(1, md)
            depth2 = dfh.decode_rgb_depth_frame(depth_rgb2, md, True)
            mesh2, unused2, _ = dmt.get_mesh_from_depth_map(depth2, K, color2, mesh, remove_edges=True,
                                                            of_by_one=obo, return_normals_of_removed=True)
            geo[f"{name}_f1_depth_rgb"] = depth_rgb2
            geo[f"{name}_f1_tri_invalid_obo{int(obo)}"] = np.all(np.asarray(mesh2.triangles) == 0, axis=1)
            geo[f"{name}_f1_unused_obo{int(obo)}"] = np.asarray(unused2, np.int64)
    # master-FOV scale (sr:537-541) applied the way the loop does it: in place on the f32 depth
    d = dfh.decode_rgb_depth_frame(geo["a_depth_rgb"], 100, True)
    import math
    scale = 1.0 / (math.tan(math.radians(45.0 / 2)) / math.tan(math.radians(60.0 / 2)))
    d *= scale
    geo["a_depth_scaled_60_to_45"] = d
    geo["scale_60_to_45"] = np.array([scale], np.float64)
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), meta=json.dumps(meta), **geo)

    # ------------------------------------------------------------------ infill_using_normals (sr:155-240)
    inf = {}
    for name, (W, H, seed) in {"a": (64, 40, 1), "b": (96, 64, 2)}.items():
        r2 = np.random.default_rng(900 + seed)
        color = r2.integers(0, 256, (H, W, 3), dtype=np.uint8)
        hole = np.zeros((H, W), bool)
        for _ in range(6):                                   # rectangular holes, some touching the border
            x0, y0 = int(r2.integers(0, W - 4)), int(r2.integers(0, H - 4))
            hole[y0:y0 + int(r2.integers(3, H // 3)), x0:x0 + int(r2.integers(3, W // 4))] = True
        hole[:, 0] = True; hole[0, W // 2:] = True
        ang = r2.uniform(0, 2 * np.pi, (H, W))
        mag = r2.uniform(0.2, 1.0, (H, W))
        normal = np.stack([np.cos(ang) * mag, np.sin(ang) * mag, r2.uniform(-1, 1, (H, W))], -1).astype(np.float32)
        normal[hole & (r2.uniform(size=(H, W)) < 0.1)] = (0.0, 1.0, 0.0)      # green-coded: skipped (sr:182)
        normal[hole & (r2.uniform(size=(H, W)) < 0.05)] = (0.0, 0.0, 1.0)     # zero XY direction: invalid (sr:178)
        normal[2, :] = (1.0, 0.0, 0.0)                                         # axis-aligned marches
        normal[:, 3] = (0.0, -1.0, 0.0)
        inf[f"{name}_color"] = color
        inf[f"{name}_hole"] = hole
        inf[f"{name}_normal"] = normal
        inf[f"{name}_out"] = sr.infill_using_normals(color.copy(), hole.copy(), normal.copy())
        inf[f"{name}_out_12"] = sr.infill_using_normals(color.copy(), hole.copy(), normal.copy(), max_steps=12)
    # mark_lower_side (infill_common.py:4-49) on normal-coloured mask images
    for name, (W, H, seed) in {"m1": (72, 48, 5), "m2": (120, 64, 6)}.items():
        r2 = np.random.default_rng(700 + seed)
        img = np.zeros((H, W, 3), np.uint8)
        for _ in range(7):
            x0, y0 = int(r2.integers(0, W - 6)), int(r2.integers(0, H - 6))
            w, h = int(r2.integers(4, W // 3)), int(r2.integers(4, H // 3))
            ang = r2.uniform(0, 2 * np.pi)
            base = np.array([(np.cos(ang) + 1) / 2 * 255, (np.sin(ang) + 1) / 2 * 255, 128.0])
            patch = np.clip(base[None, None, :] + r2.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
            img[y0:y0 + h, x0:x0 + w] = patch[:img[y0:y0 + h, x0:x0 + w].shape[0], :img[y0:y0 + h, x0:x0 + w].shape[1]]
        img[3, 5] = (127, 127, 9)                    # |dir| ~ 0.004: still marches; and an exact-zero direction:
        img[4, 5] = (128, 127, 200)
        inf[f"{name}_img"] = img
        inf[f"{name}_out"] = ic.mark_lower_side(img.copy())
        inf[f"{name}_out_8"] = ic.mark_lower_side(img.copy(), max_steps=8)
    np.savez_compressed(os.path.join(HERE, "infill.npz"), meta=json.dumps(meta), **inf)

    # ------------------------------------------------------------------ convert_to_equirectangular maps (sr:25-86)
    # cv2.remap is absent: a stub captures the float32 maps the reference hands to it (everything up to sr:82 is
    # NumPy).  Small cases keep the full 2-D maps; the VR180 size keeps the centre row / column (the maps are
    # separable: map_x depends on x only, map_y on y only, a pixel is invalid (-1,-1) if either angle is out of range).
    import cv2 as cv2_stub
    captured = {}

    def _remap(image, map_x, map_y, **kw):
        captured["x"], captured["y"] = map_x.copy(), map_y.copy()
        return image
    cv2_stub.remap = _remap
    cv2_stub.INTER_LINEAR, cv2_stub.BORDER_CONSTANT = 1, 0
    eq = {}
    for name, (W, H, fov) in {"s0": (64, 48, 100), "s1": (33, 17, 120.5), "s2": (50, 50, 75), "s3": (40, 24, 179.0),
                              "vr": (1920, 1920, 75), "vr100": (1920, 1920, 100), "hd": (1920, 1080, 90.0)}.items():
        sr.convert_to_equirectangular(np.zeros((H, W, 3), np.uint8), input_fov=fov)
        eq[f"{name}_whf"] = np.array([W, H, fov], np.float64)
        if W * H <= 4096:
            eq[f"{name}_map_x"], eq[f"{name}_map_y"] = captured["x"], captured["y"]
        else:
            eq[f"{name}_row_x"], eq[f"{name}_row_y"] = captured["x"][H // 2], captured["y"][H // 2]
            eq[f"{name}_col_x"], eq[f"{name}_col_y"] = captured["x"][:, W // 2], captured["y"][:, W // 2]
            eq[f"{name}_corner"] = np.array([captured["x"][0, 0], captured["y"][0, 0], captured["x"][-1, -1]], np.float32)
    np.savez_compressed(os.path.join(HERE, "equirect.npz"), meta=json.dumps(meta), **eq)

    # ------------------------------------------------------------------ masked_blur (sr:114-153)
    # cv2 is absent: getGaussianKernel is stubbed with OpenCV's published formula and filter2D with
    # scipy.ndimage.correlate (f32, zero border, anchor 3 for the 6-tap kernel).  What this pins is the reference's
    # own NumPy around those two calls (black handling, normalisation, clip, truncation); the f32 summation order
    # of the stub differs from any real filter2D, so consumers compare within 1 LSB.
    from scipy import ndimage

    def _gauss(n, sigma):
        sigma = 0.3 * ((n - 1) * 0.5 - 1) + 0.8 if sigma <= 0 else sigma
        x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
        g = np.exp(-0.5 / (sigma * sigma) * x * x)
        return (g * (1.0 / g.sum())).reshape(n, 1)

    def _filter2d(src, ddepth, kernel, borderType=None):
        k = kernel.astype(np.float32)
        if src.ndim == 3:
            return np.stack([ndimage.correlate(src[..., c], k, mode="constant", cval=0.0) for c in range(src.shape[2])], -1)
        return ndimage.correlate(src, k, mode="constant", cval=0.0)
    cv2_stub.getGaussianKernel, cv2_stub.filter2D, cv2_stub.BORDER_ISOLATED = _gauss, _filter2d, 16
    mb = {}
    r3 = np.random.default_rng(4242)
    img = r3.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    img[5:20, 8:30] = 0; img[30:, :6] = 0; img[0, :] = 0; img[22:26, 40:44] = (0, 0, 7)
    mb["a_img"], mb["a_out"] = img, sr.masked_blur(img.copy())
    img2 = np.zeros((24, 24, 3), np.uint8); img2[10:14, 10:14] = (255, 128, 1); img2[3, 3] = (9, 9, 9)
    mb["b_img"], mb["b_out"] = img2, sr.masked_blur(img2.copy())
    np.savez_compressed(os.path.join(HERE, "masked_blur.npz"), meta=json.dumps(meta), **mb)

    # ------------------------------------------------------------------ normal_infill (basic_nomal_infill.py:46-119)
    # The reference's own normal_infill / blur_under_mask (with its masked_blur, infill_using_normals, mark_lower_side and
    # SciPy's binary_dilation) around the same two cv2 stand-ins as above plus cv2.blur, restated exactly (integer box
    # sums over a BORDER_REFLECT_101 frame, cvRound(sum / 16)).  Pinned exactly: which pixels each step touches
    # (bni:88-91, 107, 111-118); within a few LSB: the values that went through the stand-in filter2D's summation order.
    def _blur(src, ksize, **kw):
        assert tuple(ksize) == (4, 4) and src.dtype == np.uint8
        p_ = np.pad(src.astype(np.int64), ((2, 1), (2, 1), (0, 0)), mode="reflect")
        H_, W_ = src.shape[:2]
        s_ = sum(p_[dy:dy + H_, dx:dx + W_] for dy in range(4) for dx in range(4))
        return np.rint(s_ / 16.0).astype(np.uint8)
    cv2_stub.blur = _blur
    import basic_nomal_infill as bni
    ni = {}
    for name, (W, H, seed) in {"n1": (96, 64, 11), "n2": (160, 90, 12)}.items():
        r4 = np.random.default_rng(5000 + seed)
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.clip(np.stack([xx * 255 // W, yy * 255 // H, (xx + yy) * 255 // (W + H)], -1) + r4.integers(-40, 41, (H, W, 3)), 1, 255).astype(np.uint8)
        mask = np.zeros((H, W, 3), np.uint8)
        for k in range(7):                       # holes: rectangles and discs coloured by a direction, some touching the border
            ang = r4.uniform(0, 2 * np.pi)
            base = np.array([(np.cos(ang) + 1) / 2 * 255, (np.sin(ang) + 1) / 2 * 255, r4.uniform(40, 255)])
            if k % 2:
                x0, y0 = int(r4.integers(-4, W - 6)), int(r4.integers(-4, H - 6))
                sel = (xx >= x0) & (xx < x0 + int(r4.integers(4, W // 4))) & (yy >= y0) & (yy < y0 + int(r4.integers(4, H // 3)))
            else:
                cx_, cy_, rad = r4.uniform(0, W), r4.uniform(0, H), r4.uniform(3, H / 4)
                sel = (xx - cx_) ** 2 + (yy - cy_) ** 2 < rad * rad
            col = np.clip(base[None, :] + r4.normal(0, 10, (int(sel.sum()), 3)), 0, 255).astype(np.uint8)
            mask[sel] = col
        mask[5:9, 10:14] = (200, 0, 90)          # non-black but one channel zero: not "bg" at bni:88, still marches in mark_lower_side
        mask[20, 30] = (128, 127, 200)           # a direction of ~zero length
        img[np.all(mask != 0, axis=-1)] = 0      # (the renderer leaves holes black)
        img[40:44, 50:60] = 0                    # black pixels outside the holes (masked_blur ignores them)
        ni[f"{name}_img"], ni[f"{name}_mask"] = img.copy(), mask.copy()
        ni[f"{name}_out"] = bni.normal_infill(img.copy(), mask.copy())
        m = r4.uniform(size=(H, W)) < 0.3
        m[10:30, 20:50] = True
        ni[f"{name}_bum_mask"] = m
        ni[f"{name}_bum_out"] = bni.blur_under_mask(img.copy(), m.copy())
    np.savez_compressed(os.path.join(HERE, "normal_infill.npz"), meta=json.dumps(meta), **ni)

    edge_point_goldens(dfh, dmt, sr, meta)

    for f in ("codec.npz", "camera.npz", "geometry.npz", "infill.npz", "equirect.npz", "masked_blur.npz", "normal_infill.npz", "edge_points.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
