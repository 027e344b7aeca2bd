"""ctypes binding of the plain-C oracle (oracle/mdvt_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmdvt_oracle.so")

MODE_POINTS = 0
MODE_MESH = 1


class OrcParams(C.Structure):
    _fields_ = [
        ("W", C.c_int32), ("H", C.c_int32),
        ("mode", C.c_int32), ("remove_edges", C.c_int32), ("edge_points", C.c_int32),
        ("general", C.c_int32), ("has_T", C.c_int32), ("cull", C.c_int32),
        ("K", C.c_double * 4), ("Kr", C.c_double * 4),
        ("ipd_m", C.c_double), ("max_depth", C.c_double), ("depth_scale", C.c_double),
        ("conv_angle", C.c_double),
        ("T", C.c_double * 16),
        ("key_rgb", C.c_uint8 * 4),
        ("subpixel_bits", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "mdvt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "mdvt_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, f32p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.orc_decode_depth.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_double, f32p]
        L.orc_decode_depth.restype = None
        L.orc_encode_depth.argtypes = [f32p, C.c_int, C.c_int, C.c_double, u8p]
        L.orc_encode_depth.restype = None
        L.orc_camera_matrix.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, f64p]
        L.orc_camera_matrix.restype = C.c_int
        L.orc_unproject_f64.argtypes = [f32p, C.c_int, C.c_int, f64p, C.c_int, f64p]
        L.orc_unproject_f64.restype = None
        L.orc_edge_filter.argtypes = [f32p, C.c_int, C.c_int, f64p, C.c_int, u8p, u8p, f64p]
        L.orc_edge_filter.restype = None
        L.orc_convergence_angle.argtypes = [C.c_double, C.c_double]
        L.orc_convergence_angle.restype = C.c_double
        L.orc_stats_reset.argtypes = []
        L.orc_stats_reset.restype = None
        L.orc_stats_get.argtypes = [C.POINTER(C.c_int64)]
        L.orc_stats_get.restype = None
        L.orc_render_stereo.argtypes = [C.POINTER(OrcParams), u8p, u8p, u8p, u8p, u8p, u8p, f32p, f32p]
        L.orc_render_stereo.restype = C.c_int
        L.orc_render_stereo_seed.argtypes = [C.POINTER(OrcParams), u8p, u8p, u8p, u8p, u8p, u8p, f32p, f32p, u8p, u8p]
        L.orc_render_stereo_seed.restype = C.c_int
        L.orc_edge_point_chain.argtypes = [C.POINTER(OrcParams), f32p, f64p, C.POINTER(C.c_int32), f64p, f64p]
        L.orc_edge_point_chain.restype = None
        L.orc_infill_using_normals.argtypes = [u8p, u8p, f32p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_infill_using_normals.restype = None
        L.orc_mark_lower_side.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_mark_lower_side.restype = None
        L.orc_equirect_tables.argtypes = [C.c_int, C.c_int, C.c_double, f32p, f32p]
        L.orc_equirect_tables.restype = None
        L.orc_remap_linear.argtypes = [u8p, C.c_int, C.c_int, f32p, f32p, u8p]
        L.orc_remap_linear.restype = None
        L.orc_masked_blur_kernel.argtypes = [f32p]
        L.orc_masked_blur_kernel.restype = None
        L.orc_masked_blur.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.orc_masked_blur.restype = None
        L.orc_telea_levels.argtypes = [u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_telea_levels.restype = C.c_int
        L.orc_telea_fmm.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_telea_fmm.restype = None
        L.orc_telea_bands.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_float, u8p]
        L.orc_telea_bands.restype = C.c_int
        L.orc_telea_windows.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, u8p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        L.orc_telea_windows.restype = None
        L.orc_finish_infill_mask.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, u8p, u8p]
        L.orc_finish_infill_mask.restype = C.c_int
        L.orc_box_blur4.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.orc_box_blur4.restype = None
        L.orc_dilate_cross.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_dilate_cross.restype = None
        L.orc_blur_under_mask.argtypes = [u8p, u8p, C.c_int, C.c_int, u8p]
        L.orc_blur_under_mask.restype = None
        L.orc_normal_infill.argtypes = [u8p, u8p, C.c_int, C.c_int, u8p, u8p]
        L.orc_normal_infill.restype = None
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def decode_depth(rgb: np.ndarray, max_depth: float, depth_scale: float = 1.0) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, np.uint8)
    H, W = rgb.shape[:2]
    out = np.empty((H, W), np.float32)
    lib().orc_decode_depth(_p(rgb, C.c_uint8), W, H, float(max_depth), float(depth_scale), _p(out, C.c_float))
    return out


def encode_depth(depth: np.ndarray, max_depth: float) -> np.ndarray:
    depth = np.ascontiguousarray(depth, np.float32)
    H, W = depth.shape
    out = np.empty((H, W, 3), np.uint8)
    lib().orc_encode_depth(_p(depth, C.c_float), W, H, float(max_depth), _p(out, C.c_uint8))
    return out


def camera_matrix(xfov, yfov, W, H) -> np.ndarray:
    K = np.empty(9, np.float64)
    rc = lib().orc_camera_matrix(np.nan if xfov is None else float(xfov),
                                 np.nan if yfov is None else float(yfov), int(W), int(H), _p(K, C.c_double))
    if rc != 0:
        raise ValueError("either xfov or yfov is required")
    return K.reshape(3, 3)


def k4(K: np.ndarray) -> np.ndarray:
    K = np.asarray(K, np.float64)
    return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.float64)


def unproject_f64(depth: np.ndarray, K: np.ndarray, of_by_one: bool) -> np.ndarray:
    depth = np.ascontiguousarray(depth, np.float32)
    H, W = depth.shape
    out = np.empty((H * W, 3), np.float64)
    kk = k4(K)
    lib().orc_unproject_f64(_p(depth, C.c_float), W, H, _p(kk, C.c_double), int(of_by_one), _p(out, C.c_double))
    return out


def edge_filter(depth: np.ndarray, K: np.ndarray, of_by_one: bool, want_normals: bool = False):
    """-> (tri_invalid u8[2*(H-1)*(W-1)] in draw order, unused u8[H*W], normals f64[H*W,3] | None)"""
    depth = np.ascontiguousarray(depth, np.float32)
    H, W = depth.shape
    tri = np.empty(2 * (H - 1) * (W - 1), np.uint8)
    unused = np.empty(H * W, np.uint8)
    normals = np.empty((H * W, 3), np.float64) if want_normals else None
    kk = k4(K)
    lib().orc_edge_filter(_p(depth, C.c_float), W, H, _p(kk, C.c_double), int(of_by_one),
                          _p(tri, C.c_uint8), _p(unused, C.c_uint8),
                          _p(normals, C.c_double) if want_normals else None)
    return tri, unused, normals


def convergence_angle(distance: float, ipd_m: float) -> float:
    return float(lib().orc_convergence_angle(float(distance), float(ipd_m)))


def make_params(W, H, K, *, Kr=None, ipd_m=0.065, max_depth=100.0, depth_scale=1.0, mode=MODE_POINTS,
                remove_edges=False, edge_points=False, conv_angle=0.0, T=None, key_rgb=(0, 0, 0),
                force_general=False, cull=0, subpixel_bits=None) -> OrcParams:
    p = OrcParams()
    p.W, p.H = int(W), int(H)
    p.mode = int(mode)
    p.remove_edges = int(bool(remove_edges))
    p.cull = int(cull)
    # (MDVT_TEST_SUBPIXEL_BITS: the whole GPU suite on another sub-pixel grid, tests/conftest.py -- the renderers AND this checker)
    p.subpixel_bits = int(os.environ.get("MDVT_TEST_SUBPIXEL_BITS", 0)) if subpixel_bits is None else int(subpixel_bits)
    p.edge_points = int(edge_points)
    kk = k4(K)
    kr = kk if Kr is None else k4(Kr)
    for i in range(4):
        p.K[i] = kk[i]
        p.Kr[i] = kr[i]
    p.ipd_m = float(ipd_m)
    p.max_depth = float(max_depth)
    p.depth_scale = float(depth_scale)
    p.conv_angle = float(conv_angle or 0.0)
    p.has_T = int(T is not None)
    if T is not None:
        Tm = np.asarray(T, np.float64).reshape(16)
        for i in range(16):
            p.T[i] = Tm[i]
    p.general = int(bool(force_general) or T is not None or p.conv_angle != 0.0 or not np.array_equal(kk, kr))
    for i in range(3):
        p.key_rgb[i] = int(key_rgb[i])
    return p


STAT_NAMES = ("near_partial", "near_all", "depth_ties", "fragments", "culled")


def stats_reset():
    lib().orc_stats_reset()


def stats():
    """Rasteriser counters since stats_reset(): dict of STAT_NAMES (see mdvt_oracle.h)."""
    v = (C.c_int64 * 5)()
    lib().orc_stats_get(v)
    return dict(zip(STAT_NAMES, [int(x) for x in v]))


def render_stereo(p: OrcParams, depth_rgb: np.ndarray, color_rgb: np.ndarray, want_depth: bool = False, want_seed: bool = False):
    """-> dict(left_rgb, right_rgb, left_mask, right_mask[, left_depth, right_depth][, left_seed, right_seed])"""
    depth_rgb = np.ascontiguousarray(depth_rgb, np.uint8)
    color_rgb = np.ascontiguousarray(color_rgb, np.uint8)
    H, W = p.H, p.W
    assert depth_rgb.shape == (H, W, 3) and color_rgb.shape == (H, W, 3)
    out = {
        "left_rgb": np.empty((H, W, 3), np.uint8), "right_rgb": np.empty((H, W, 3), np.uint8),
        "left_mask": np.empty((H, W), np.uint8), "right_mask": np.empty((H, W), np.uint8),
    }
    ld = rd = None
    if want_depth:
        out["left_depth"] = np.empty((H, W), np.float32)
        out["right_depth"] = np.empty((H, W), np.float32)
        ld, rd = _p(out["left_depth"], C.c_float), _p(out["right_depth"], C.c_float)
    ls = rs = None
    if want_seed:
        out["left_seed"] = np.empty((H, W, 3), np.uint8)
        out["right_seed"] = np.empty((H, W, 3), np.uint8)
        ls, rs = _p(out["left_seed"], C.c_uint8), _p(out["right_seed"], C.c_uint8)
    rc = lib().orc_render_stereo_seed(C.byref(p), _p(depth_rgb, C.c_uint8), _p(color_rgb, C.c_uint8),
                                      _p(out["left_rgb"], C.c_uint8), _p(out["right_rgb"], C.c_uint8),
                                      _p(out["left_mask"], C.c_uint8), _p(out["right_mask"], C.c_uint8), ld, rd, ls, rs)
    if rc != 0:
        raise ValueError(f"orc_render_stereo failed: {rc}")
    return out


class OrcGlOpts(C.Structure):
    _fields_ = [("near_clip", C.c_int32), ("samples", C.c_int32), ("pattern", C.c_int32), ("resolve", C.c_int32),
                ("depth_tie_tol", C.c_double)]


# |1/Z - 1/Z'| below which a GL's depth buffer cannot tell two fragments apart: window depth is 1 - near/Z to 24 bits (or an
# f32 just below 1.0: the same 2^-24 spacing) with near = 1e-4 (dmt:1520), and its interpolation is good to a few of those
# steps -- 4 * 2^-24 / 1e-4.
GL_DEPTH_TIE_TOL = 4.0 * 2.0 ** -24 / 1e-4


def render_stereo_gl(p: OrcParams, depth_rgb: np.ndarray, color_rgb: np.ndarray, *, near_clip=False, samples=0, pattern=0, resolve=0,
                     depth_tie_tol=0.0):
    """The GL candidates of mdvt_oracle.c (diagnostics) -> dict(left_rgb, right_rgb, left_mask, right_mask[, left_ambiguous,
    right_ambiguous] when depth_tie_tol > 0)."""
    depth_rgb = np.ascontiguousarray(depth_rgb, np.uint8)
    color_rgb = np.ascontiguousarray(color_rgb, np.uint8)
    H, W = p.H, p.W
    assert depth_rgb.shape == (H, W, 3) and color_rgb.shape == (H, W, 3)
    g = OrcGlOpts(int(bool(near_clip)), int(samples), int(pattern), int(resolve), float(depth_tie_tol))
    out = {"left_rgb": np.empty((H, W, 3), np.uint8), "right_rgb": np.empty((H, W, 3), np.uint8),
           "left_mask": np.empty((H, W), np.uint8), "right_mask": np.empty((H, W), np.uint8)}
    la = ra = None
    if depth_tie_tol > 0:
        out["left_ambiguous"], out["right_ambiguous"] = np.empty((H, W), np.uint8), np.empty((H, W), np.uint8)
        la, ra = _p(out["left_ambiguous"], C.c_uint8), _p(out["right_ambiguous"], C.c_uint8)
    L = lib()
    L.orc_render_stereo_gl.restype = C.c_int
    rc = L.orc_render_stereo_gl(C.byref(p), C.byref(g), _p(depth_rgb, C.c_uint8), _p(color_rgb, C.c_uint8),
                                _p(out["left_rgb"], C.c_uint8), _p(out["right_rgb"], C.c_uint8),
                                _p(out["left_mask"], C.c_uint8), _p(out["right_mask"], C.c_uint8), la, ra)
    if rc != 0:
        raise ValueError(f"orc_render_stereo_gl failed: {rc}")
    return out


def edge_point_chain(p: OrcParams, depth: np.ndarray, normals: np.ndarray | None = None):
    """The edge points' f64 chain for EVERY vertex of a decoded, scaled depth map -> (px i32[H*W, 2 eyes, 2 (x, y)] with
    INT32_MIN outside the frame / for depth code 0, z f64[H*W, 2], unprojected normals f64[H*W, 2, 3] | None)."""
    depth = np.ascontiguousarray(depth, np.float32)
    n = p.W * p.H
    assert depth.shape == (p.H, p.W)
    px = np.empty((n, 2, 2), np.int32)
    z = np.empty((n, 2), np.float64)
    nrm = None
    if normals is not None:
        normals = np.ascontiguousarray(normals, np.float64)
        nrm = np.empty((n, 2, 3), np.float64)
    lib().orc_edge_point_chain(C.byref(p), _p(depth, C.c_float), None if normals is None else _p(normals, C.c_double),
                               _p(px, C.c_int32), _p(z, C.c_double), None if nrm is None else _p(nrm, C.c_double))
    return px, z, nrm


def infill_using_normals(color: np.ndarray, hole_mask: np.ndarray, normal_map: np.ndarray, max_steps: int = 400) -> np.ndarray:
    color = np.ascontiguousarray(color, np.uint8)
    hole = np.ascontiguousarray(hole_mask, np.uint8)
    normal = np.ascontiguousarray(normal_map, np.float32)
    H, W = hole.shape
    out = np.empty_like(color)
    lib().orc_infill_using_normals(_p(color, C.c_uint8), _p(hole, C.c_uint8), _p(normal, C.c_float), W, H, int(max_steps),
                                   _p(out, C.c_uint8))
    return out


def mark_lower_side(normals_img: np.ndarray, max_steps: int = 30) -> np.ndarray:
    img = np.ascontiguousarray(normals_img, np.uint8)
    H, W = img.shape[:2]
    out = np.empty_like(img)
    lib().orc_mark_lower_side(_p(img, C.c_uint8), W, H, int(max_steps), _p(out, C.c_uint8))
    return out


def equirect_tables(W: int, H: int, input_fov: float = 100.0):
    """-> (mx f32[W], my f32[H]); -1 marks an angle outside the input fov (sr:25-78)."""
    mx, my = np.empty(W, np.float32), np.empty(H, np.float32)
    lib().orc_equirect_tables(int(W), int(H), float(input_fov), _p(mx, C.c_float), _p(my, C.c_float))
    return mx, my


def equirect_maps(W: int, H: int, input_fov: float = 100.0):
    """The 2-D float32 maps the reference hands to cv2.remap (sr:77-78)."""
    mx, my = equirect_tables(W, H, input_fov)
    bad = (my == -1)[:, None] | (mx == -1)[None, :]
    X = np.where(bad, np.float32(-1), np.broadcast_to(mx[None, :], (H, W))).astype(np.float32)
    Y = np.where(bad, np.float32(-1), np.broadcast_to(my[:, None], (H, W))).astype(np.float32)
    return np.ascontiguousarray(X), np.ascontiguousarray(Y)


def remap_linear(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    H, W = src.shape[:2]
    map_x, map_y = np.ascontiguousarray(map_x, np.float32), np.ascontiguousarray(map_y, np.float32)
    assert map_x.shape == (H, W) and map_y.shape == (H, W)
    out = np.empty_like(src)
    lib().orc_remap_linear(_p(src, C.c_uint8), W, H, _p(map_x, C.c_float), _p(map_y, C.c_float), _p(out, C.c_uint8))
    return out


def convert_to_equirectangular(image: np.ndarray, input_fov: float = 100.0) -> np.ndarray:
    H, W = image.shape[:2]
    X, Y = equirect_maps(W, H, input_fov)
    return remap_linear(image, X, Y)


def masked_blur_kernel() -> np.ndarray:
    K = np.empty(36, np.float32)
    lib().orc_masked_blur_kernel(_p(K, C.c_float))
    return K.reshape(6, 6)


def masked_blur(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    out = np.empty_like(img)
    lib().orc_masked_blur(_p(img, C.c_uint8), W, H, _p(out, C.c_uint8))
    return out


def telea_levels(img: np.ndarray, mask: np.ndarray, must_fill=None, radius: int = 3, max_rounds: int = 65000):
    """-> (filled image, number of must_fill / mask pixels left unfilled)"""
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    H, W = mask.shape
    mf = None if must_fill is None else np.ascontiguousarray(must_fill, np.uint8)
    out = np.empty_like(img)
    rem = lib().orc_telea_levels(_p(img, C.c_uint8), _p(mask, C.c_uint8), None if mf is None else _p(mf, C.c_uint8),
                                 W, H, int(radius), int(max_rounds), _p(out, C.c_uint8))
    return out, rem


def finish_infill_mask(seed: np.ndarray, key_rgb=(0, 255, 0), max_rounds: int = 65000, want_blur: bool = False):
    seed = np.ascontiguousarray(seed, np.uint8)
    H, W = seed.shape[:2]
    key = np.array(list(key_rgb) + [0], np.uint8)
    out = np.empty_like(seed)
    blur = np.empty_like(seed) if want_blur else None
    rem = lib().orc_finish_infill_mask(_p(seed, C.c_uint8), W, H, _p(key, C.c_uint8), int(max_rounds), _p(out, C.c_uint8),
                                       _p(blur, C.c_uint8) if want_blur else None)
    return (out, blur, rem) if want_blur else (out, rem)


def telea_fmm(img: np.ndarray, mask: np.ndarray, radius: int = 3) -> np.ndarray:
    """Sequential fast-marching order (cv2.inpaint's), same estimator: for measuring the effect of the order only."""
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    H, W = mask.shape
    out = np.empty_like(img)
    lib().orc_telea_fmm(_p(img, C.c_uint8), _p(mask, C.c_uint8), W, H, int(radius), _p(out, C.c_uint8))
    return out


def telea_bands(img: np.ndarray, mask: np.ndarray, delta: float, radius: int = 3):
    """The march of telea_fmm with its pops grouped into steps of `delta` in T (r05 experiment) -> (image, steps)."""
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    H, W = mask.shape
    out = np.empty_like(img)
    steps = lib().orc_telea_bands(_p(img, C.c_uint8), _p(mask, C.c_uint8), W, H, int(radius), float(delta), _p(out, C.c_uint8))
    return out, int(steps)


def box_blur4(img: np.ndarray) -> np.ndarray:
    """cv2.blur(img, (4, 4)) on uint8 [H,W,3] (basic_nomal_infill.py:103)."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    out = np.empty_like(img)
    lib().orc_box_blur4(_p(img, C.c_uint8), W, H, _p(out, C.c_uint8))
    return out


def dilate_cross(mask: np.ndarray, iterations: int = 6) -> np.ndarray:
    """scipy.ndimage.binary_dilation(mask, iterations=iterations) (basic_nomal_infill.py:112) -> bool [H,W]."""
    m = np.ascontiguousarray(mask, np.uint8)
    H, W = m.shape
    out = np.empty_like(m)
    lib().orc_dilate_cross(_p(m, C.c_uint8), W, H, int(iterations), _p(out, C.c_uint8))
    return out.astype(bool)


def blur_under_mask(img: np.ndarray, bool_mask: np.ndarray) -> np.ndarray:
    """basic_nomal_infill.blur_under_mask (basic_nomal_infill.py:46-85) with its default ksize / sigma."""
    img = np.ascontiguousarray(img, np.uint8)
    m = np.ascontiguousarray(bool_mask, np.uint8)
    H, W = m.shape
    out = np.empty_like(img)
    lib().orc_blur_under_mask(_p(img, C.c_uint8), _p(m, C.c_uint8), W, H, _p(out, C.c_uint8))
    return out


def normal_infill(img: np.ndarray, infill_mask: np.ndarray, want_stages: bool = False):
    """basic_nomal_infill.normal_infill (basic_nomal_infill.py:87-119).  -> out, or (out, dict of the stages)."""
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(infill_mask, np.uint8)
    H, W = img.shape[:2]
    n = H * W
    out = np.empty_like(img)
    st = np.empty(11 * n, np.uint8) if want_stages else None
    lib().orc_normal_infill(_p(img, C.c_uint8), _p(mask, C.c_uint8), W, H, _p(out, C.c_uint8), None if st is None else _p(st, C.c_uint8))
    if not want_stages:
        return out
    return out, {"blur": st[:3 * n].reshape(H, W, 3), "filled": st[3 * n:6 * n].reshape(H, W, 3),
                 "merged": st[6 * n:9 * n].reshape(H, W, 3), "bg": st[9 * n:10 * n].reshape(H, W).astype(bool),
                 "grown": st[10 * n:11 * n].reshape(H, W).astype(bool)}


def telea_windows(img: np.ndarray, mask: np.ndarray, radius: int = 3):
    """telea_fmm's heap order from order-free steps (orc_telea_windows: windows of 0.70 in T, sorted pops, order-free activation,
    dependency-ordered estimates) -> (image, T field, stats dict).  Must equal telea_fmm bit for bit."""
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    H, W = mask.shape
    out = np.empty_like(img)
    T = np.empty((H, W), np.float32)
    st = (C.c_uint64 * 8)()
    lib().orc_telea_windows(_p(img, C.c_uint8), _p(mask, C.c_uint8), W, H, int(radius), _p(out, C.c_uint8), _p(T, C.c_float), st)
    names = ("windows", "max_pops_per_window", "pops", "sum_T_chain", "sum_colour_chain", "max_colour_chain_in_a_window",
             "lookahead_violations", "max_activations_per_window")
    return out, T, dict(zip(names, (int(v) for v in st)))
