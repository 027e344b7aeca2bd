"""Device-side mirror of the reference's infill_common.mark_lower_side (infill_common.py:4-49), same name
and argument meaning, on PyTorch-ROCm tensors through the HIP kernel.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

from . import _lib


def mark_lower_side(normals_img, max_steps=30, out=None):
    """normals_img: uint8 CUDA tensor [H,W,3] (the normal-coloured infill mask; black = not a hole).
    Returns a uint8 [H,W,3] tensor that is (0,0,255) where a hole pixel's march along its encoded XY
    direction is about to leave the hole ("the lower side"), 0 elsewhere."""
    import torch
    from .depth_frames_helper import _ctx
    assert normals_img.is_cuda and normals_img.dtype == torch.uint8 and normals_img.dim() == 3 and normals_img.shape[2] == 3
    img = normals_img.contiguous()
    H, W = int(img.shape[0]), int(img.shape[1])
    if out is None:
        out = torch.empty_like(img)
    ctx = _ctx(img.device.index or 0, W, H)
    s = torch.cuda.current_stream(img.device)
    ctx.check(_lib.load().mdvt_mark_lower_side(ctx.handle, img.data_ptr(), 3 * W, out.data_ptr(), 3 * W, int(max_steps),
                                              C.c_void_p(s.cuda_stream)))
    return out
