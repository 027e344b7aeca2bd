"""Host-side driver of the MI355X stereo-rerender path: the counterpart of the frame loop of the
reference's stereo_rerender.py (sr:471-944) with the per-pixel stages replaced by the HIP kernels
behind include/mdvt.h.

What stays on the host (as in the reference): argument handling, per-frame scalars (camera matrix,
master-FOV depth scale, convergence angle), the convergence pre-pass.  What moved to the GPU:
decode, unprojection, eye/pose transform, z-buffered render, hole mask, edge filter, edge points.

PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .depth_map_tools import compute_camera_matrix, fov_from_camera_matrix


# ------------------------------------------------------------------------------------------------
# scalar helpers with the reference's names
# ------------------------------------------------------------------------------------------------
def convergence_angle(distance, pupillary_distance):
    """Angle (rad) each eye rotates inward to converge at `distance`.  Reference: sr:94-112."""
    if distance == 0:
        raise ValueError("convergence_angle: a convergence distance of 0 has no toe-in angle")
    half_baseline = pupillary_distance / 2
    return math.atan(half_baseline / distance)


def fill_nan_with_closest(values):
    """NaNs take the nearest non-NaN sample (earlier one on a tie), in place.  Reference: sr:244-251."""
    known = [i for i, v in enumerate(values) if not math.isnan(v)]
    if known:
        for i, v in enumerate(values):
            if math.isnan(v):
                values[i] = values[min(known, key=lambda k: abs(k - i))]
    return values


def curve_fit(values):
    """Savitzky-Golay smoothing (window <= 99, order 2) with a repeated tail.  Reference: sr:253-268."""
    from scipy.signal import savgol_filter
    y = np.array(values)
    n_tail = min(50, len(y))
    y_ext = np.concatenate([y, y[-n_tail:]])
    window = min(100, len(y_ext))
    if window % 2 == 0:
        window -= 1
    out = savgol_filter(y_ext, window_length=window, polyorder=2)
    out = out[:-n_tail] if n_tail > 0 else out
    if len(out) != len(y):
        raise AssertionError(f"curve_fit: smoothed series has {len(out)} samples for {len(y)} inputs")
    return out


def master_fov_scale_depth(xfov, master_xfov):
    """sr:537-538: depth scale that accounts for viewing at `master_xfov` what was filmed at `xfov`."""
    scale_disp = math.tan(math.radians(master_xfov / 2)) / math.tan(math.radians(xfov / 2))
    return 1.0 / scale_disp


def infill_using_normals(color_img, hole_mask, normal_map, max_steps=400, out=None):
    """Device version of the reference's infill_using_normals (sr:155-240), same argument meaning:
    color_img uint8 [H,W,3], hole_mask bool/uint8 [H,W] (True = fill), normal_map float32 [H,W,3] whose XY
    components give the march direction.  All CUDA tensors; returns a new uint8 [H,W,3] tensor."""
    import torch
    from .depth_frames_helper import _ctx
    assert color_img.is_cuda and color_img.dtype == torch.uint8 and color_img.dim() == 3 and color_img.shape[2] == 3
    H, W = int(color_img.shape[0]), int(color_img.shape[1])
    color_img = color_img.contiguous()
    hole = hole_mask.to(torch.uint8).contiguous()
    normal = normal_map.to(torch.float32).contiguous()
    assert tuple(hole.shape) == (H, W) and tuple(normal.shape) == (H, W, 3)
    if out is None:
        out = torch.empty_like(color_img)
    ctx = _ctx(color_img.device.index or 0, W, H)
    s = torch.cuda.current_stream(color_img.device)
    ctx.check(_lib.load().mdvt_infill_using_normals(ctx.handle, color_img.data_ptr(), 3 * W, hole.data_ptr(), W,
                                                    normal.data_ptr(), 12 * W, out.data_ptr(), 3 * W, int(max_steps),
                                                    C.c_void_p(s.cuda_stream)))
    return out


def infill_using_mask_normals(img, hole_mask, infill_mask, max_steps=400, out=None):
    """sr:809-812 (--do_basic_infill) for whole batches: `infill_using_normals(image, bg_mask, infill_mask * 2 - 1)` with the
    normals taken straight from the finished infill-mask image ((u8 / 255) * 2 - 1 in f32, sr:808 + 810).  img, infill_mask: uint8
    CUDA [H,W,3] or [N,H,W,3]; hole_mask: uint8 / bool CUDA [H,W] or [N,H,W] (non-zero = hole); rows and images may be strided
    (one half of side-by-side frames).  out None: a filled copy of img is returned; out given (may be img itself): filled in
    place after img has been copied into it."""
    import torch
    from .depth_frames_helper import _ctx
    assert img.is_cuda and img.dtype == torch.uint8 and img.dim() in (3, 4) and img.shape[-1] == 3
    assert infill_mask.shape == img.shape and infill_mask.dtype == torch.uint8 and tuple(hole_mask.shape) == tuple(img.shape[:-1])
    if hole_mask.dtype == torch.bool:
        hole_mask = hole_mask.to(torch.uint8)
    if out is None:
        out = img.clone()
    elif out.data_ptr() != img.data_ptr():
        out.copy_(img)
    for t in (out, infill_mask):
        assert t.stride(-1) == 1 and t.stride(-2) == 3, "pixels must be packed RGB"
    assert hole_mask.stride(-1) == 1
    batched = img.dim() == 4
    N = int(img.shape[0]) if batched else 1
    H, W = int(img.shape[-3]), int(img.shape[-2])
    ctx = _ctx(img.device.index or 0, W, H)
    s = torch.cuda.current_stream(img.device)
    stride = (lambda t: t.stride(0) if batched else 0)
    ctx.check(_lib.load().mdvt_infill_using_mask_normals(ctx.handle, out.data_ptr(), out.stride(-3), stride(out),
                                                         hole_mask.data_ptr(), hole_mask.stride(-2), stride(hole_mask),
                                                         infill_mask.data_ptr(), infill_mask.stride(-3), stride(infill_mask),
                                                         N, int(max_steps), C.c_void_p(s.cuda_stream)))
    return out


def touchly_depth(depth, touchly_max_depth=5, touchly_min_depth=0, zero_is_far=False, out=None):
    """float32 CUDA depth [H,W] -> uint8 [H,W,3] Touchly reverse-depth plane (sr:549-551; with zero_is_far the
    variant used after a render, sr:689-691 / 825-829).  --touchly1 without a pose file is
    vconcat([color_frame, touchly_depth(decode(depth_rgb) * scale)]) (sr:548-552)."""
    import torch
    from .depth_frames_helper import _ctx
    assert depth.is_cuda and depth.dtype == torch.float32 and depth.dim() == 2 and depth.stride(1) == 1   # rows may be strided
    H, W = int(depth.shape[0]), int(depth.shape[1])
    if out is None:
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=depth.device)
    assert out.is_contiguous()
    ctx = _ctx(depth.device.index or 0, W, H)
    s = torch.cuda.current_stream(depth.device)
    ctx.check(_lib.load().mdvt_touchly_depth(ctx.handle, depth.data_ptr(), 4 * depth.stride(0), out.data_ptr(), 3 * W,
                                             float(touchly_max_depth), float(touchly_min_depth), int(bool(zero_is_far)),
                                             C.c_void_p(s.cuda_stream)))
    return out


def equirect_tables(W, H, input_fov=100):
    """The lookup tables of convert_to_equirectangular (sr:41-78) as float32 NumPy arrays (map_x[W], map_y[H];
    -1 = outside the input fov).  Host arithmetic of the C library (mdvt_equirect_tables), no GPU needed."""
    mx, my = np.empty(int(W), np.float32), np.empty(int(H), np.float32)
    rc = _lib.load().mdvt_equirect_tables(int(W), int(H), float(input_fov), mx.ctypes.data_as(C.POINTER(C.c_float)),
                                          my.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != _lib.MDVT_OK:
        raise ValueError(f"bad equirect table request: W={W} H={H} input_fov={input_fov}")
    return mx, my


_equirect_cache = {}


def convert_to_equirectangular(image, input_fov=100, out=None):
    """Device version of the reference's convert_to_equirectangular (sr:25-86): uint8 CUDA image(s) [H,W,3] or
    [N,H,W,3] rendered at `input_fov` -> 180-degree equirectangular image(s) of the same size, the valid region
    centred and everything outside the input fov black.  Rows / images may be strided views (e.g. one eye of a
    side-by-side buffer); pixels must be packed RGB."""
    import torch
    from .depth_frames_helper import _ctx
    assert image.is_cuda and image.dtype == torch.uint8 and image.dim() in (3, 4) and image.shape[-1] == 3
    assert image.stride(-1) == 1 and image.stride(-2) == 3, "pixels must be packed RGB"
    batched = image.dim() == 4
    N = int(image.shape[0]) if batched else 1
    H, W = int(image.shape[-3]), int(image.shape[-2])
    if out is None:
        out = torch.empty(tuple(image.shape), dtype=torch.uint8, device=image.device)
    assert out.shape == image.shape and out.stride(-1) == 1 and out.stride(-2) == 3 and out.data_ptr() != image.data_ptr()
    dev = image.device.index or 0
    key = (dev, W, H, float(input_fov))
    if key not in _equirect_cache:
        mx, my = equirect_tables(W, H, input_fov)
        _equirect_cache[key] = (torch.from_numpy(mx).to(image.device), torch.from_numpy(my).to(image.device))
    tx, ty = _equirect_cache[key]
    ctx = _ctx(dev, W, H)
    s = torch.cuda.current_stream(image.device)
    ctx.check(_lib.load().mdvt_equirect_remap(ctx.handle, image.data_ptr(), image.stride(-3), image.stride(0) if batched else 0,
                                              out.data_ptr(), out.stride(-3), out.stride(0) if batched else 0, N,
                                              tx.data_ptr(), ty.data_ptr(), C.c_void_p(s.cuda_stream)))
    return out


def masked_blur(img, out=None):
    """Device version of the reference's masked_blur(img, ksize=(6,6), sigma=0) (sr:114-153): a 6x6 Gaussian that
    ignores pure black pixels.  uint8 CUDA [H,W,3] (rows may be strided) -> uint8 [H,W,3]."""
    import torch
    from .depth_frames_helper import _ctx
    assert img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3
    assert img.stride(2) == 1 and img.stride(1) == 3, "pixels must be packed RGB"
    H, W = int(img.shape[0]), int(img.shape[1])
    if out is None:
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=img.device)
    assert out.shape == img.shape and out.stride(2) == 1 and out.stride(1) == 3
    ctx = _ctx(img.device.index or 0, W, H)
    s = torch.cuda.current_stream(img.device)
    ctx.check(_lib.load().mdvt_masked_blur(ctx.handle, img.data_ptr(), img.stride(0), out.data_ptr(), out.stride(0),
                                           C.c_void_p(s.cuda_stream)))
    return out


VR180_SIZE = 1920      # sr:528: the VR180 render is always 1920 x 1920


def vr180_render_fov(K):
    """sr:527-535: the fov of the (square) VR180 render camera for input camera matrix K."""
    fovx, fovy = fov_from_camera_matrix(K)
    max_fov = max(fovx, fovy)
    if max_fov >= 180:
        raise ValueError("fov cant be 180 or over, the tool is not built to handle fisheye distorted input video")   # sr:531-532
    return max(75, max_fov)


def make_frame_params(W, H, xfov=None, yfov=None, *, master_xfov=45.0, pupillary_distance=63,
                      convergence_distance=None, transformation=None, vr180=False):
    """The per-frame scalars the reference computes before its render calls (sr:515-541, 563-566,
    707-721) as one mdvt_frame_params record.  Pure host code (no GPU needed).
    vr180: the render camera becomes the square `vr180_render_fov` camera and replaces the master fov
    (sr:527-535); the render target is 1920 x 1920, which this build supports for 1920 x 1920 inputs only
    (for any other input size the reference's own writer, sized 2W x H at sr:409-437, rejects the frames)."""
    if xfov is None and yfov is None:
        raise ValueError("Error: Either --xfov_file, --xfov or --yfov must be provided.")    # sr:319-320
    K = compute_camera_matrix(xfov, yfov, W, H)
    p = _lib.MdvtFrameParams()
    flat = K.reshape(9)
    for k in range(9):
        p.K[k] = flat[k]
        p.Krender[k] = flat[k]
    xf = xfov
    if xf is None:      # sr:537 needs xf; with only --yfov the reference raises TypeError. Use K's xfov.
        xf = float(np.rad2deg(2 * np.arctan2(W, 2 * K[0, 0])))
    if vr180:
        if (W, H) != (VR180_SIZE, VR180_SIZE):
            raise ValueError(f"--vr180 renders {VR180_SIZE}x{VR180_SIZE} (sr:528); only inputs of that size are supported, got {W}x{H}")
        render_fov = vr180_render_fov(K)
        master_xfov = render_fov                                                             # sr:533
        Kr = compute_camera_matrix(render_fov, render_fov, VR180_SIZE, VR180_SIZE).reshape(9)  # sr:535
        for k in range(9):
            p.Krender[k] = Kr[k]
    scale = master_fov_scale_depth(xf, master_xfov)
    p.depth_scale = scale
    p.convergence_angle = 0.0
    if convergence_distance is not None and not math.isnan(float(convergence_distance)):
        cd = float(convergence_distance)
        if cd != 0:                                                     # sr:711-713: zero = skip
            cd *= scale                                                 # sr:716
            p.convergence_angle = convergence_angle(cd, pupillary_distance / 1000)
    p.has_T = 0
    if transformation is not None:
        T = np.asarray(transformation, np.float64).reshape(16)
        for k in range(16):
            p.T[k] = T[k]
        p.has_T = 1
    return p


# ------------------------------------------------------------------------------------------------
# renderer
# ------------------------------------------------------------------------------------------------
class PreparedRender:
    """One pre-validated submission: launch() = one mdvt_render_stereo_batch call on a stream."""

    def __init__(self, renderer, n, params_arr, io, results, keepalive, device):
        self._fn = renderer._L.mdvt_render_stereo_batch
        self._check = renderer.ctx.check
        self._h = renderer.ctx.handle
        self._n, self._arr, self._io, self._ioref = n, params_arr, io, C.byref(io)
        self.results = results
        self._keep = keepalive
        self._device = device
        self._torch = renderer.torch

    def launch(self, stream=None):
        s = stream if stream is not None else self._torch.cuda.current_stream(self._device)
        self._check(self._fn(self._h, self._n, self._arr, self._ioref, C.c_void_p(s.cuda_stream)))
        return self.results


class StereoRerenderer:
    """One render context for W x H frames on one GPU.

    Keyword names follow the CLI flags of the reference (sr:273-316):
      pupillary_distance  int mm (default 63)        max_depth            default 100
      master_xfov         default 45.0               render_as_pointcloud point splat instead of mesh
      remove_edges / infill_mask / do_basic_infill   turn the 89-degree edge filter on (sr:568-570)
      dont_remove_edges                              overrides the above (sr:572-573)
      dont_place_points_in_edges                     no edge points (sr:589)
      workspace_mib                                  budget for the posed / converged mesh path's workspace (mdvt.h; 0 = 4096 MiB)
      cull                                           0 draw both faces of the mesh (default), 1 cull back faces, 2 front faces
                                                     (dmt:1507-1556 leaves Open3D's mesh_show_back_face at its default)
      subpixel_bits                                  the rasteriser's sub-pixel grid (GL_SUBPIXEL_BITS of the GL that ran the
                                                     reference): 0 = 8, the default; 4 = the grid of the GL the fixtures
                                                     tests/golden/render_gl_*.npz were rendered with (mdvt.h)
    """

    FINISH_SPLIT_FRAMES = 32         # finish_infill_mask_sbs: from this many frames per call, two concurrent half passes

    def __init__(self, width: int, height: int, *, device: Optional[int] = None, pupillary_distance=63,
                 max_depth=100, master_xfov: float = 45.0, render_as_pointcloud: bool = False,
                 remove_edges: bool = False, infill_mask: bool = False, do_basic_infill: bool = False,
                 dont_remove_edges: bool = False, dont_place_points_in_edges: bool = False, cull: int = 0,
                 workspace_mib: int = 0, subpixel_bits: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm GPU visible: the stereo-rerender path has no CPU fallback")
        self.torch = torch
        self.W, self.H = int(width), int(height)
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.pupillary_distance = pupillary_distance
        self.max_depth = max_depth
        self.master_xfov = float(master_xfov)
        self.mode = _lib.MODE_POINTS if render_as_pointcloud else _lib.MODE_MESH
        rm = bool(infill_mask or remove_edges or do_basic_infill)          # sr:568-570
        if dont_remove_edges:
            rm = False                                                     # sr:572-573
        self.remove_edges = rm
        self.edge_points = rm and not dont_place_points_in_edges           # sr:589
        self.do_basic_infill = bool(do_basic_infill)                       # edge points then only feed the mask (sr:809-812)
        self.key_rgb = (0, 255, 0) if infill_mask else (0, 0, 0)           # sr:555-558
        self.ctx = _lib.Context(self.device, self.W, self.H)
        cfg = _lib.MdvtConfig()
        cfg.mode = self.mode
        cfg.remove_edges = int(self.remove_edges)
        cfg.edge_points = (2 if self.do_basic_infill else 1) if self.edge_points else 0
        cfg.cull = int(cull)
        cfg.workspace_mib = int(workspace_mib)                             # 0: the library's default (4096)
        self.cull = int(cull)
        cfg.subpixel_bits = int(subpixel_bits)
        self.subpixel_bits = int(subpixel_bits)
        cfg.ipd_m = self.pupillary_distance / 1000                         # sr:458-459
        cfg.max_depth = float(self.max_depth)
        for k in range(3):
            cfg.key_rgb[k] = self.key_rgb[k]
        self.ctx.check(self._L.mdvt_set_config(self.ctx.handle, C.byref(cfg)))
        self._cfg, self._ctx2, self._side = cfg, None, None      # (finish_infill_mask_sbs's second context, made on first use)

    @property
    def _L(self):
        return _lib.load()

    # -- per-frame scalars (sr:515-541, 563-566, 707-721) -------------------------------------
    def frame_params(self, xfov=None, yfov=None, convergence_distance=None, transformation=None, vr180=False):
        return make_frame_params(self.W, self.H, xfov, yfov, master_xfov=self.master_xfov,
                                 pupillary_distance=self.pupillary_distance,
                                 convergence_distance=convergence_distance, transformation=transformation, vr180=vr180)

    # -- the per-frame loop body, batched -----------------------------------------------------
    def prepare(self, depth_rgb, color_rgb, params, *, out_sbs=None, out_mask=None, want_depth: bool = False,
                out_depth=None, want_maskbits: bool = False, want_hole_counts: bool = False, want_seed: bool = False,
                want_mask: bool = True):
        """Validate once and pack everything one submission needs (buffer table, parameter records).
        Returns a PreparedRender whose launch() is a single C-ABI call -- use it when the same buffers
        are rendered into repeatedly (streaming loops, benchmarks).
        want_mask=False (with want_maskbits): no byte mask is written, the packed 1 bit/px mask is the hole mask -- where the
        compaction is fused into the render kernel (pure-shift point frames; the library refuses elsewhere)."""
        torch = self.torch
        single = depth_rgb.dim() == 3
        if single:
            depth_rgb, color_rgb = depth_rgb[None], color_rgb[None]
        N, H, W = depth_rgb.shape[0], self.H, self.W
        assert tuple(depth_rgb.shape) == (N, H, W, 3) and tuple(color_rgb.shape) == (N, H, W, 3), \
            "color image and depth image need to have same width and height"          # sr:507
        assert depth_rgb.dtype == torch.uint8 and color_rgb.dtype == torch.uint8
        assert depth_rgb.is_cuda and color_rgb.is_cuda and depth_rgb.is_contiguous() and color_rgb.is_contiguous()
        arr = self.pack_params(params, N)
        dev = depth_rgb.device
        sbs = out_sbs if out_sbs is not None else _lib.retry_after_release(lambda: torch.empty((N, H, 2 * W, 3), dtype=torch.uint8, device=dev))
        assert want_mask or want_maskbits, "want_mask=False needs want_maskbits"
        mask = None
        if want_mask:
            mask = out_mask if out_mask is not None else _lib.retry_after_release(lambda: torch.empty((N, H, 2 * W), dtype=torch.uint8, device=dev))
        assert tuple(sbs.shape[-3:]) == (H, 2 * W, 3) and (mask is None or tuple(mask.shape[-2:]) == (H, 2 * W))
        zout = None
        if want_depth:
            zout = out_depth if out_depth is not None else _lib.retry_after_release(lambda: torch.empty((N, H, 2 * W), dtype=torch.float32, device=dev))

        io = _lib.MdvtIO()
        io.depth_rgb, io.depth_pitch, io.depth_stride = depth_rgb.data_ptr(), 3 * W, 3 * W * H
        io.color_rgb, io.color_pitch, io.color_stride = color_rgb.data_ptr(), 3 * W, 3 * W * H
        io.left_rgb, io.right_rgb = sbs.data_ptr(), sbs.data_ptr() + 3 * W
        io.rgb_pitch, io.rgb_stride = 6 * W, 6 * W * H
        if mask is not None:
            io.left_mask, io.right_mask = mask.data_ptr(), mask.data_ptr() + W
            io.mask_pitch, io.mask_stride = 2 * W, 2 * W * H
        if zout is not None:
            io.left_depth, io.right_depth = zout.data_ptr(), zout.data_ptr() + 4 * W
            io.zout_pitch, io.zout_stride = 8 * W, 8 * W * H
        res = {"sbs": sbs[0] if single else sbs}
        if mask is not None:
            res["mask"] = mask[0] if single else mask
        if zout is not None:
            res["depth"] = zout[0] if single else zout
        bits = counts = None
        if want_maskbits:       # [N, H, 2, 4*ceil(W/32)] u8: packed 1 bit/px rows, left then right eye
            rowb = 4 * ((W + 31) // 32)
            bits = torch.zeros((N, H, 2, rowb), dtype=torch.uint8, device=dev)
            io.left_maskbits, io.right_maskbits = bits.data_ptr(), bits.data_ptr() + rowb
            io.maskbits_pitch, io.maskbits_stride = 2 * rowb, 2 * rowb * H
            res["maskbits"] = bits[0] if single else bits
        if want_hole_counts:    # [N, 2] int32 (left, right)
            counts = torch.zeros((N, 2), dtype=torch.int32, device=dev)
            io.hole_counts = counts.data_ptr()
            res["hole_counts"] = counts[0] if single else counts
        seed = None
        if want_seed:           # [N, H, 2W, 3] u8: the infill-mask seed images, left | right (sr:787-803, 921-928)
            seed = _lib.retry_after_release(lambda: torch.empty((N, H, 2 * W, 3), dtype=torch.uint8, device=dev))
            io.left_seed, io.right_seed = seed.data_ptr(), seed.data_ptr() + 3 * W
            io.seed_pitch, io.seed_stride = 6 * W, 6 * W * H
            res["seed"] = seed[0] if single else seed
        return PreparedRender(self, N, arr, io, res, (depth_rgb, color_rgb, sbs, mask, zout, bits, counts, seed), dev)

    def render(self, depth_rgb, color_rgb, params, *, out_sbs=None, out_mask=None, want_depth: bool = False,
               out_depth=None, stream=None, want_maskbits: bool = False, want_hole_counts: bool = False, want_mask: bool = True,
               want_seed: bool = False):
        """depth_rgb, color_rgb: uint8 device tensors [N,H,W,3] (or [H,W,3]); params: one
        MdvtFrameParams or a sequence of N.  Returns dict(sbs=[N,H,2W,3] u8 (left | right, sr:918),
        mask=[N,H,2W] u8 (255 = hole), depth=[N,H,2W] f32 (optional; 0 = background))."""
        return self.prepare(depth_rgb, color_rgb, params, out_sbs=out_sbs, out_mask=out_mask,
                            want_depth=want_depth, out_depth=out_depth, want_maskbits=want_maskbits,
                            want_hole_counts=want_hole_counts, want_seed=want_seed, want_mask=want_mask).launch(stream)

    def finish_infill_mask_sbs(self, seed_sbs, out=None, max_rounds: int = 0, want_remaining: bool = False, no_host_wait: bool = False):
        """finish_infill_mask for side-by-side seed buffers [N,H,2W,3] (render(want_seed=True)["seed"]): both eyes of
        all frames in one pass (mdvt_finish_infill_mask_stereo).  Returns [N,H,2W,3] (and, with want_remaining, an
        int32 tensor [2,N]: left eyes, right eyes).  no_host_wait: the asynchronous form (max_rounds, default 256, levels are
        launched without reading the deepest level back: nothing waits on the stream)."""
        if no_host_wait:
            max_rounds = -(int(max_rounds) if max_rounds > 0 else 256)
        torch = self.torch
        W, H = self.W, self.H
        single = seed_sbs.dim() == 3
        if single:
            seed_sbs = seed_sbs[None]
        assert seed_sbs.is_cuda and seed_sbs.dtype == torch.uint8 and tuple(seed_sbs.shape[1:]) == (H, 2 * W, 3)
        assert seed_sbs.stride(-1) == 1 and seed_sbs.stride(-2) == 3, "pixels must be packed RGB"
        N = int(seed_sbs.shape[0])
        if out is None:
            out = torch.empty((N, H, 2 * W, 3), dtype=torch.uint8, device=seed_sbs.device)
        elif out.dim() == 3:
            out = out[None]
        assert tuple(out.shape) == (N, H, 2 * W, 3) and out.stride(-1) == 1 and out.stride(-2) == 3
        rem = torch.zeros((2, N), dtype=torch.int32, device=seed_sbs.device) if want_remaining else None
        s = torch.cuda.current_stream(seed_sbs.device)

        def one_pass(ctx, a, b, stream, r):
            sd, o = seed_sbs[a:b], out[a:b]
            ctx.check(self._L.mdvt_finish_infill_mask_stereo(
                ctx.handle, sd.data_ptr(), sd.data_ptr() + 3 * W, sd.stride(1), sd.stride(0),
                o.data_ptr(), o.data_ptr() + 3 * W, o.stride(1), o.stride(0), b - a, int(max_rounds),
                r.data_ptr() if r is not None else None, C.c_void_p(stream.cuda_stream)))

        if N >= self.FINISH_SPLIT_FRAMES:
            # Two halves, two contexts, two streams (r04): the completion is ~260 dependent launches per pass whose marking half
            # waits 84 % of its wave cycles; from 32 frames per call a second pass running beside it fills them (0.434 -> 0.376 ms
            # per 1080p frame; at 16 frames the halves are launch-floor bound and lose).  The second work area is allocated on
            # first use (14 B per pixel and image, at most 32 images).  Same bytes: a frame's mask does not depend on its batch.
            if self._ctx2 is None:
                self._ctx2 = _lib.Context(self.device, self.W, self.H)
                self._ctx2.check(self._L.mdvt_set_config(self._ctx2.handle, C.byref(self._cfg)))
                self._side = torch.cuda.Stream(device=seed_sbs.device)
            h = N // 2
            # (the counters of both halves are allocated and zeroed on the caller's stream BEFORE the side stream waits for it, and
            #  the second half's tensor is recorded on the side stream: the fill and the library's writes are ordered, and the
            #  caching allocator will not hand the block on while the side stream may still write it -- advisor, r04)
            r1 = torch.zeros((2, h), dtype=torch.int32, device=seed_sbs.device) if want_remaining else None
            r2 = torch.zeros((2, N - h), dtype=torch.int32, device=seed_sbs.device) if want_remaining else None
            self._side.wait_stream(s)
            if r2 is not None:
                r2.record_stream(self._side)
            one_pass(self._ctx2, h, N, self._side, r2)         # (each call returns once its pass A has been read back)
            one_pass(self.ctx, 0, h, s, r1)
            s.wait_stream(self._side)
            if rem is not None:
                rem[:, :h], rem[:, h:] = r1, r2
        else:
            one_pass(self.ctx, 0, N, s, rem)
        res = out[0] if single else out
        return (res, rem) if want_remaining else res

    def finish_infill_mask(self, seed, out=None, max_rounds: int = 0, want_remaining: bool = False):
        """sr:803-808 + 816 on the device: seed image(s) from render(want_seed=True) -> the finished infill-mask
        image(s): Telea-weighted inpaint of the key-coloured / black pixels (level by level, see include/mdvt.h),
        key-coloured pixels keep the inpainted normal, then masked_blur.  seed: uint8 CUDA [H,W,3] or [N,H,W,3],
        rows / images may be strided (e.g. one eye of the side-by-side seed buffer).  Returns the image(s)
        (and, with want_remaining, an int32 tensor [N] of key-coloured pixels the front did not reach)."""
        torch = self.torch
        assert seed.is_cuda and seed.dtype == torch.uint8 and seed.dim() in (3, 4) and seed.shape[-1] == 3
        assert seed.stride(-1) == 1 and seed.stride(-2) == 3, "pixels must be packed RGB"
        assert tuple(seed.shape[-3:-1]) == (self.H, self.W)
        batched = seed.dim() == 4
        N = int(seed.shape[0]) if batched else 1
        if out is None:
            out = torch.empty(tuple(seed.shape), dtype=torch.uint8, device=seed.device)
        assert out.shape == seed.shape and out.stride(-1) == 1 and out.stride(-2) == 3
        rem = torch.zeros(N, dtype=torch.int32, device=seed.device) if want_remaining else None
        s = torch.cuda.current_stream(seed.device)
        self.ctx.check(self._L.mdvt_finish_infill_mask(
            self.ctx.handle, seed.data_ptr(), seed.stride(-3), seed.stride(0) if batched else 0, out.data_ptr(),
            out.stride(-3), out.stride(0) if batched else 0, N, int(max_rounds), rem.data_ptr() if rem is not None else None,
            C.c_void_p(s.cuda_stream)))
        return (out, rem) if want_remaining else out

    @staticmethod
    def pack_params(params, n_frames: int):
        """One record, a sequence of N records, or an already packed ctypes array -> ctypes array of N.
        Pack once and pass the array to render() to keep the per-call host cost off the launch path."""
        if isinstance(params, C.Array):
            if len(params) != n_frames:
                raise ValueError(f"need {n_frames} frame parameter records, got {len(params)}")
            return params
        if isinstance(params, _lib.MdvtFrameParams):
            params = [params] * n_frames
        if len(params) != n_frames:
            raise ValueError(f"need {n_frames} frame parameter records, got {len(params)}")
        return (_lib.MdvtFrameParams * n_frames)(*params)

    def edge_filter(self, depth_rgb, params, of_by_one: Optional[bool] = None, stream=None):
        """The filter part of dmt.get_mesh_from_depth_map(remove_edges=True) for one frame:
        -> (tri_invalid u8[2(H-1)(W-1)] in draw order, unused u8[H*W])."""
        torch = self.torch
        H, W = self.H, self.W
        assert tuple(depth_rgb.shape) == (H, W, 3) and depth_rgb.is_cuda and depth_rgb.is_contiguous()
        dev = depth_rgb.device
        tri = torch.empty(2 * (H - 1) * (W - 1), dtype=torch.uint8, device=dev)
        unused = torch.empty(H * W, dtype=torch.uint8, device=dev)
        if of_by_one is None:
            of_by_one = self.mode == _lib.MODE_MESH
        K = (C.c_double * 9)(*[params.K[k] for k in range(9)])
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        self.ctx.check(self._L.mdvt_edge_filter(self.ctx.handle, depth_rgb.data_ptr(), 3 * W, K, params.depth_scale,
                                                int(of_by_one), tri.data_ptr(), unused.data_ptr(),
                                                C.c_void_p(s.cuda_stream)))
        return tri, unused

    def edge_point_pixels(self, depth_rgb, params, how: int = 0, stream=None):
        """Where the edge point of every vertex of a frame lands (sr:589-606, 727-735, 745-750, 838-858) ->
        int32[H*W, 2 eyes, 2 (x, y)], INT32_MIN outside the frame.  how = 0: the reference's f64 chain per vertex; 1: as the
        row kernels of a pure-shift frame take it (mdvt.h)."""
        torch = self.torch
        H, W = self.H, self.W
        assert tuple(depth_rgb.shape) == (H, W, 3) and depth_rgb.is_cuda and depth_rgb.is_contiguous()
        px = torch.empty((H * W, 2, 2), dtype=torch.int32, device=depth_rgb.device)
        s = stream if stream is not None else torch.cuda.current_stream(depth_rgb.device)
        self.ctx.check(self._L.mdvt_edge_point_pixels(self.ctx.handle, C.byref(params), depth_rgb.data_ptr(), 3 * W, int(how),
                                                      px.data_ptr(), C.c_void_p(s.cuda_stream)))
        return px

    def close(self, release_cached_memory: bool = False):
        """Destroy the context(s).  Their workspace blocks stay in the library's process-wide pool for the next context of this GPU
        (up to mdvt_set_cached_memory_limit, 4 GiB per GPU by default); release_cached_memory=True hands them back to the driver
        now -- what a long-lived process that is done rendering should do (torch's allocator cannot reclaim them)."""
        if getattr(self, "_ctx2", None) is not None:
            self._ctx2.close()
            self._ctx2 = None
        self.ctx.close()
        if release_cached_memory:
            _lib.release_cached_memory(self.device)


# ------------------------------------------------------------------------------------------------
# command line (the reference's flags, sr:273-316, for raw frame dumps -- see clip.py)
# ------------------------------------------------------------------------------------------------
def build_arg_parser():
    import argparse
    ap = argparse.ArgumentParser(description="Convert an RGB-encoded depth frame dump and optional colour frame dump "
                                             "into a stereoscopic side-by-side output on MI355X GPUs.")
    ap.add_argument("--master_xfov", type=float, default=45.0)
    ap.add_argument("--depth_video", type=str, required=True, help="uint8 [N,H,W,3] .npy dump of the RGB-coded depth frames")
    ap.add_argument("--color_video", type=str, required=False, help="uint8 [N,H,W,3] .npy dump of the colour frames")
    ap.add_argument("--xfov", type=float, required=False)
    ap.add_argument("--yfov", type=float, required=False)
    ap.add_argument("--xfov_file", type=str, required=False)
    ap.add_argument("--max_depth", default=100, type=int)
    ap.add_argument("--transformation_file", type=str, required=False)
    ap.add_argument("--transformation_lock_frame", default=0, type=int)
    ap.add_argument("--pupillary_distance", default=63, type=int)
    ap.add_argument("--max_frames", default=-1, type=int)
    ap.add_argument("--render_as_pointcloud", action="store_true")
    ap.add_argument("--convergence_file", type=str, required=False)
    ap.add_argument("--dont_place_points_in_edges", action="store_true")
    ap.add_argument("--dont_remove_edges", action="store_true")
    ap.add_argument("--infill_mask", action="store_true")
    ap.add_argument("--remove_edges", action="store_true")
    ap.add_argument("--create_sbs_depth_video", action="store_true")
    ap.add_argument("--batch", default=16, type=int, help="frames per GPU submission")
    ap.add_argument("--green_and_black_infill_mask", action="store_true")
    ap.add_argument("--vr180", action="store_true", help="VR180 180-degree side-by-side (1920x1920 inputs)")
    ap.add_argument("--touchly0", action="store_true", help="Touchly0 format: left | right | left depth, needs VR180")
    ap.add_argument("--touchly1", action="store_true", help="Touchly1 format: colour over depth, no stereo rendering")
    ap.add_argument("--touchly_max_depth", default=5, type=float)
    ap.add_argument("--touchly_min_depth", default=0, type=float)
    ap.add_argument("--do_basic_infill", action="store_true", help="fill the holes by marching along the infill-mask normals")
    ap.add_argument("--normal_infill", action="store_true",
                    help="not a reference flag: also run basic_nomal_infill.py's normal_infill on every frame while it is on the device "
                         "and write <output>_infilled.npy (needs --infill_mask)")
    for flag in ("--compressed", "--mask_video", "--save_background", "--load_background"):
        ap.add_argument(flag, nargs="?", const=True, default=None, help="reference flag outside the built hot path")
    return ap


def main(argv=None):
    from . import clip
    args = build_arg_parser().parse_args(argv)
    for flag in ("compressed", "mask_video", "save_background", "load_background"):
        if getattr(args, flag) is not None:
            raise NotImplementedError(f"--{flag} is outside the hot path this build covers (DESIGN.md section 1)")
    if args.xfov is None and args.yfov is None and args.xfov_file is None:
        raise ValueError("Error: Either --xfov_file, --xfov or --yfov must be provided.")       # sr:319-320
    if args.xfov is None and args.xfov_file is None:
        raise NotImplementedError("--yfov without --xfov: the reference itself fails at sr:537 in this case")
    if not os.path.isfile(args.depth_video):
        raise FileNotFoundError(f"Depth video not found: {args.depth_video}")                  # sr:326
    if args.color_video and not os.path.isfile(args.color_video):
        raise FileNotFoundError(f"Color video not found: {args.color_video}")                  # sr:331
    stats, final = clip.run(args.depth_video, args.color_video, batch=args.batch,
                            create_sbs_depth_video=args.create_sbs_depth_video, max_frames=args.max_frames,
                            green_and_black_infill_mask=args.green_and_black_infill_mask,
                            xfov=args.xfov, xfov_file=args.xfov_file, convergence_file=args.convergence_file,
                            transformation_file=args.transformation_file,
                            transformation_lock_frame=args.transformation_lock_frame,
                            pupillary_distance=args.pupillary_distance, max_depth=args.max_depth,
                            master_xfov=args.master_xfov, render_as_pointcloud=args.render_as_pointcloud,
                            remove_edges=args.remove_edges, dont_remove_edges=args.dont_remove_edges,
                            infill_mask=args.infill_mask,
                            dont_place_points_in_edges=args.dont_place_points_in_edges,
                            vr180=args.vr180, touchly0=args.touchly0, touchly1=args.touchly1,
                            do_basic_infill=args.do_basic_infill, normal_infill=args.normal_infill,
                            touchly_max_depth=args.touchly_max_depth, touchly_min_depth=args.touchly_min_depth)
    if int(os.environ.get("RANK", "0")) == 0:
        frames, secs = float(stats[:, 0].sum()), float(stats[:, 1].max())
        print(f"Processing complete. Output saved to: {final}  ({frames:.0f} frames, {frames / secs:.1f} frames/s incl. host I/O)")
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
