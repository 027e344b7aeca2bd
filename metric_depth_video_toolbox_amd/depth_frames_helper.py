"""Device-side mirror of the reference's depth_frames_helper.py codec entry points (same names and
argument meaning), operating on PyTorch-ROCm tensors through the HIP kernels.  No CPU fallback.

  decode_rgb_depth_frame(rgb, max_depth, bit16)     dfh:99-103 (-> dfh:63-75, 13-24)
  encode_depth_frame(depth, max_depth, bgr=True)    dfh:5-11 + dfh:48-61 with bit16=True (sr:930-936)
"""
from __future__ import annotations

import ctypes as C

from . import _lib

_ctx_cache = {}


def _ctx(device_index: int, W: int, H: int):
    import os
    key = (device_index, W, H, os.environ.get("MDVT_LIB_VARIANT", ""))      # a context belongs to the library that made it
    if key not in _ctx_cache:
        _ctx_cache[key] = _lib.Context(device_index, W, H)
    return _ctx_cache[key]


def decode_rgb_depth_frame(rgb, max_depth, bit16=True, depth_scale: float = 1.0, out=None):
    """uint8 device tensor [H,W,3] (RGB order) -> float32 [H,W] metres.  Only the 16-bit format the
    hot path uses is built (bit16 must be True; the reference's 24-bit branch is not on the path)."""
    import torch
    if not bit16:
        raise NotImplementedError("only the bit16 depth format is on the stereo-rerender path")
    assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.dim() == 3 and rgb.shape[2] == 3 and rgb.is_contiguous()
    H, W = int(rgb.shape[0]), int(rgb.shape[1])
    if out is None:
        out = torch.empty((H, W), dtype=torch.float32, device=rgb.device)
    ctx = _ctx(rgb.device.index or 0, W, H)
    s = torch.cuda.current_stream(rgb.device)
    ctx.check(_lib.load().mdvt_decode_depth(ctx.handle, rgb.data_ptr(), 3 * W, out.data_ptr(), 4 * W,
                                            float(max_depth), float(depth_scale), C.c_void_p(s.cuda_stream)))
    return out


def encode_depth_frame(depth, max_depth, bgr: bool = True, out=None):
    """float32 device tensor [H,W] metres -> uint8 [H,W,3] 16-bit depth code, B,G,R order by default
    (what encode_data_as_BGR hands to cv2.VideoWriter) or R,G,B with bgr=False."""
    import torch
    assert depth.is_cuda and depth.dtype == torch.float32 and depth.dim() == 2 and depth.is_contiguous()
    H, W = int(depth.shape[0]), int(depth.shape[1])
    if out is None:
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=depth.device)
    ctx = _ctx(depth.device.index or 0, W, H)
    s = torch.cuda.current_stream(depth.device)
    ctx.check(_lib.load().mdvt_encode_depth(ctx.handle, depth.data_ptr(), 4 * W, out.data_ptr(), 3 * W,
                                            float(max_depth), int(bool(bgr)), C.c_void_p(s.cuda_stream)))
    return out


def swap_rb(frames, out=None):
    """cv2.cvtColor(frame, COLOR_BGR2RGB) / COLOR_RGB2BGR (sr:493, 505, 928, 941) on the device: uint8 CUDA [H,W,3] or
    [N,H,W,3] (rows / images may be strided) -> the same with bytes 0 and 2 of every pixel swapped.  out may be
    `frames` itself (in place)."""
    import torch
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() in (3, 4) and frames.shape[-1] == 3
    assert frames.stride(-1) == 1 and frames.stride(-2) == 3, "pixels must be packed"
    batched = frames.dim() == 4
    N = int(frames.shape[0]) if batched else 1
    H, W = int(frames.shape[-3]), int(frames.shape[-2])
    if out is None:
        out = torch.empty(tuple(frames.shape), dtype=torch.uint8, device=frames.device)
    assert out.shape == frames.shape and out.stride(-1) == 1 and out.stride(-2) == 3
    ctx = _ctx(frames.device.index or 0, W, H)
    s = torch.cuda.current_stream(frames.device)
    ctx.check(_lib.load().mdvt_swap_rb(ctx.handle, frames.data_ptr(), frames.stride(-3), frames.stride(0) if batched else 0,
                                       out.data_ptr(), out.stride(-3), out.stride(0) if batched else 0, N, C.c_void_p(s.cuda_stream)))
    return out
