"""NumPy restatement of the stages of the hot path that the reference's own NumPy functions pin.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference lines it
restates (paths relative to the read-only reference tree; abbreviations as in SURVEY.md:
dfh = depth_frames_helper.py, dmt = depth_map_tools.py, sr = stereo_rerender.py).

These are checked against golden vectors produced by importing the reference itself
(tests/golden/gen_golden.py) and in turn cross-check the plain-C oracle (mdvt_oracle.c).  The
rasteriser is not here: it only exists in the C oracle because nothing in the reference pins it.
"""
from __future__ import annotations

import math

import numpy as np

CODE_SCALE = 255 ** 4  # dfh:8, dfh:22 -- NOT 2**32


# ----------------------------------------------------------------------------- codec (dfh)
def encode_depth_as_uint32(depth, max_depth):
    """dfh:5-11: clip to [0, max], f64 multiply by 255**4/max, truncate to u32."""
    d = np.minimum(np.maximum(np.asarray(depth), 0.0), max_depth)
    return (d.astype(np.float64) * (CODE_SCALE / float(max_depth))).astype(np.uint32)


def encode_data_as_rgb16(code_u32):
    """dfh:48-61 with bit16=True, expressed in RGB order: R = G = byte 3, B = byte 2."""
    code = np.asarray(code_u32, np.uint32)
    hi = (code >> np.uint32(24)).astype(np.uint8)
    lo = ((code >> np.uint32(16)) & np.uint32(0xFF)).astype(np.uint8)
    return np.stack([hi, hi, lo], axis=-1)


def decode_rgb16_as_uint32(rgb):
    """dfh:63-69 (bit16): byte 3 <- R, byte 2 <- B; G is ignored, bytes 0/1 stay zero."""
    rgb = np.asarray(rgb, np.uint8)
    return (rgb[..., 0].astype(np.uint32) << np.uint32(24)) | (rgb[..., 2].astype(np.uint32) << np.uint32(16))


def decode_rgb_depth_frame(rgb, max_depth, depth_scale=None):
    """dfh:99-103 -> dfh:13-24: f32(code) * f32(max/255**4); optionally sr:541's f32 in-place scale."""
    code = decode_rgb16_as_uint32(rgb)
    depth = code.astype(np.float32) * np.float32(float(max_depth) / CODE_SCALE)
    if depth_scale is not None:
        depth = depth * np.float32(depth_scale)
    return depth


# ----------------------------------------------------------------------------- camera (dmt)
def compute_camera_matrix(xfov_deg, yfov_deg, W, H):
    """dmt:902-934."""
    if xfov_deg is None and yfov_deg is None:
        raise ValueError("either xfov or yfov is required")
    fx = fy = None
    if xfov_deg is not None:
        fx = W / (2.0 * np.tan(np.deg2rad(xfov_deg) / 2.0))
    if yfov_deg is not None:
        fy = H / (2.0 * np.tan(np.deg2rad(yfov_deg) / 2.0))
    fy = fx if fy is None else fy
    fx = fy if fx is None else fx
    return np.array([[fx, 0.0, W / 2.0], [0.0, fy, H / 2.0], [0.0, 0.0, 1.0]], np.float64)


def fov_from_camera_matrix(K):
    """dmt:1640-1649."""
    w, h = K[0][2] * 2, K[1][2] * 2
    return (np.rad2deg(2 * np.arctan2(w, 2 * K[0][0])), np.rad2deg(2 * np.arctan2(h, 2 * K[1][1])))


def master_fov_scale_depth(xfov_deg, master_xfov_deg):
    """sr:537-538."""
    scale_disp = math.tan(math.radians(master_xfov_deg / 2)) / math.tan(math.radians(xfov_deg / 2))
    return 1.0 / scale_disp


def convergence_angle(distance, ipd_m):
    """sr:94-112."""
    if distance == 0:
        raise ValueError("Distance must be non-zero to compute a valid angle.")
    return math.atan((ipd_m / 2) / distance)


# ----------------------------------------------------------------------------- geometry (dmt)
def grid_coords(n, of_by_one):
    """dmt:1114-1122: integer grid, or the f32 grid pre-scaled by (n+1)/n ("off by one" fix)."""
    g = np.arange(n)
    if of_by_one:
        g = g.astype(np.float32) * np.float32((n + 1) / n)
    return g


def unproject(depth, K, of_by_one):
    """dmt:1112-1133 as NumPy >= 2 evaluates it: f64 result from an f32 depth map (NEP 50)."""
    depth = np.asarray(depth)
    H, W = depth.shape
    gx = grid_coords(W, of_by_one).astype(np.float64)[None, :]
    gy = grid_coords(H, of_by_one).astype(np.float64)[:, None]
    z = depth.astype(np.float64)
    X = (gx - K[0][2]) * z / K[0][0]
    Y = (gy - K[1][2]) * z / K[1][1]
    return np.stack([X, Y, z], axis=-1).reshape(-1, 3)


def grid_triangles(H, W):
    """dmt:1243-1254: per cell tri1=(v[i,j],v[i+1,j],v[i+1,j+1]), tri2=(v[i,j],v[i+1,j+1],v[i,j+1]);
    all tri1 (row-major over cells) first, then all tri2."""
    top = (np.arange(H - 1)[:, None] * W + np.arange(W - 1)[None, :]).ravel()
    a, b, c, d = top, top + W, top + W + 1, top + 1
    return np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)


def edge_filter(points, H, W, angle_threshold_deg=89.0):
    """dmt:1283-1294 + 1339-1366.

    -> (tri_invalid bool[2(H-1)(W-1)], unused_idx int[], vertex_normals f64[H*W,3]) where
    vertex_normals is the last-writer-wins scatter of unit triangle normals (dmt:1358-1364)."""
    tris = grid_triangles(H, W)
    p0, p1, p2 = points[tris[:, 0]], points[tris[:, 1]], points[tris[:, 2]]
    n = np.cross(p1 - p0, p2 - p0)
    view = -(p0 + p1 + p2) / 3.0
    dot = np.einsum("ij,ij->i", n, view)
    ln = np.sqrt(np.einsum("ij,ij->i", n, n))
    lv = np.sqrt(np.einsum("ij,ij->i", view, view))
    cosines = dot / (ln * lv + 1e-15)
    invalid = cosines < np.cos(np.radians(angle_threshold_deg))
    touched = np.zeros(H * W, bool)
    touched[tris[invalid].ravel()] = True
    nlen = np.linalg.norm(n, axis=1)
    unit = np.ones_like(n)
    ok = nlen > 0
    unit[ok] = n[ok] / nlen[ok, None]
    vn = np.zeros((H * W, 3), np.float64)
    flat = tris.reshape(-1)
    rep = np.repeat(unit, 3, axis=0)
    # explicit last-writer-wins (NumPy's fancy assignment is sequential in practice; make it so by
    # keeping, for every vertex, the last position at which it occurs in the flattened index list)
    last = np.full(H * W, -1, np.int64)
    last[flat] = np.arange(flat.size)
    has = last >= 0
    vn[has] = rep[last[has]]
    return invalid, np.nonzero(touched)[0], vn


# ----------------------------------------------------------------------------- convergence pre-pass (sr)
def fill_nan_with_closest(values):
    """sr:244-251: every NaN takes the value of the nearest non-NaN sample (first one on a tie)."""
    vals = list(values)
    good = [i for i, v in enumerate(vals) if not math.isnan(v)]
    if good:
        for i, v in enumerate(vals):
            if math.isnan(v):
                vals[i] = vals[min(good, key=lambda g: abs(g - i))]
    return vals


def curve_fit(values):
    """sr:253-268: Savitzky-Golay (poly 2) over the clip, tail-extended by <= 50 samples."""
    from scipy.signal import savgol_filter
    y = np.array(values)
    n_tail = min(50, len(y))
    ext = np.concatenate([y, y[-n_tail:]])
    win = min(100, len(ext))
    if win % 2 == 0:
        win -= 1
    sm = savgol_filter(ext, window_length=win, polyorder=2)
    return sm[:-n_tail] if n_tail > 0 else sm


# ----------------------------------------------------------------------------- edge points (sr)
def edge_point_chain(depth, K, Kr, W, H, of_by_one, ipd_m, conv_angle=0.0, T=None, normals=None):
    """Where the reference splats the vertices of removed triangles, for EVERY vertex of the (decoded, scaled, f32)
    depth map: the f64 operations of dmt:1117-1128, sr:596-600, Open3D's transform / rotate / translate on the cloud
    (sr:615-619, 727-732 and, on top of the left eye's state, 838-847), cv2.projectPoints with the f32-cast camera matrix
    (dmt:1058) and np.round (sr:746, 858), one NumPy operation per node.
    -> (px int64[H*W, 2 eyes, 2 (x, y)] = np.round(points_2d), z f64[H*W, 2], unprojected normals f64[H*W, 2, 3] | None)."""
    depth = np.asarray(depth, np.float32)
    K, Kr = np.asarray(K, np.float64), np.asarray(Kr, np.float64).astype(np.float32).astype(np.float64)
    x, y = np.meshgrid(np.arange(W), np.arange(H))
    if of_by_one:
        x = x.astype(np.float32); y = y.astype(np.float32)
        x *= (W + 1) / W
        y *= (H + 1) / H
    z = depth
    X = (x - K[0, 2]) * z / K[0, 0]
    Y = (y - K[1, 2]) * z / K[1, 1]
    P = np.stack((X, Y, z), axis=-1).reshape(-1, 3).astype(np.float64)
    clouds = [P * np.array([(W - 1) / W, (H - 1) / H, 1.0])]
    if normals is not None:
        clouds.append(np.asarray(normals, np.float64).reshape(-1, 3) + P)
    h = ipd_m / 2.0
    has_conv = conv_angle == conv_angle and conv_angle != 0.0
    c, s = math.cos(conv_angle), math.sin(conv_angle)

    def roty(q, sn):
        return np.stack([c * q[:, 0] + sn * q[:, 2], q[:, 1], (-sn) * q[:, 0] + c * q[:, 2]], axis=1)

    eyes = []
    for q in clouds:
        if T is not None:
            Tm = np.asarray(T, np.float64).reshape(4, 4)
            hh = [((Tm[r, 0] * q[:, 0] + Tm[r, 1] * q[:, 1]) + Tm[r, 2] * q[:, 2]) + Tm[r, 3] * 1.0 for r in range(4)]
            q = np.stack([hh[0] / hh[3], hh[1] / hh[3], hh[2] / hh[3]], axis=1)
        if has_conv:
            q = roty(q, -s)
        left = q + np.array([h, 0.0, 0.0])
        q = left + np.array([-h, 0.0, 0.0])
        if has_conv:
            q = roty(roty(q, s), s)
        right = q + np.array([-h, 0.0, 0.0])
        eyes.append((left, right))
    px = np.empty((H * W, 2, 2), np.int64)
    zz = np.empty((H * W, 2), np.float64)
    for e in range(2):
        q = eyes[0][e]
        with np.errstate(divide="ignore", invalid="ignore"):
            iz = np.where(q[:, 2] != 0.0, 1.0 / q[:, 2], 1.0)
            u = (q[:, 0] * iz) * Kr[0, 0] + Kr[0, 2]
            v = (q[:, 1] * iz) * Kr[1, 1] + Kr[1, 2]
            ru, rv = np.round(u), np.round(v)
        ok = np.isfinite(ru) & np.isfinite(rv) & (np.abs(ru) < 2.0 ** 30) & (np.abs(rv) < 2.0 ** 30)
        px[:, e, 0] = np.where(ok, ru, -(2.0 ** 30)).astype(np.int64)
        px[:, e, 1] = np.where(ok, rv, -(2.0 ** 30)).astype(np.int64)
        zz[:, e] = q[:, 2]
    nrm = None
    if normals is not None:
        nrm = np.stack([eyes[1][0] - eyes[0][0], eyes[1][1] - eyes[0][1]], axis=1)
    return px, zz, nrm
