"""Upstream model hop (BASELINE config C5): Depth-Anything-V2 inference in
PyTorch-ROCm feeding the HIP reproject kernels on ONE HIP stream, without the depth-video round trip.

In the reference this hop goes through files: a depth generator (video_metric_convert.py:97-148,
other/metric_dpt_func.py:7-16) writes `*_depth.mkv` with dfh.save_depth_video (dfh:125-161: optional
resize, encode_depth_as_uint32, encode_data_as_BGR(bit16)), and stereo_rerender.py reads it back.  Here
the model's depth tensor is quantised to the same 16-bit code on the device (mdvt_encode_depth), so the
render kernels see exactly the bytes a depth video would have delivered, and everything is enqueued on
torch's current stream: model kernels -> resize -> encode -> render, no host synchronisation in between.

The Depth-Anything-V2 sources and weights are not part of the reference tree (external clone,
install_mdvtoolbox.sh:265-273) and cannot be downloaded here; the architecture comes from
`transformers` (DepthAnythingForDepthEstimation with a DINOv2 backbone) with seeded random weights
unless a state dict is supplied.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _lib

_ENCODERS = {   # Depth-Anything-V2 model cards: (hidden, layers, heads, out stages, neck sizes, fusion)
    "vits": (384, 12, 6, (3, 6, 9, 12), (48, 96, 192, 384), 64),
    "vitb": (768, 12, 12, (3, 6, 9, 12), (96, 192, 384, 768), 128),
    "vitl": (1024, 24, 16, (5, 12, 18, 24), (256, 512, 1024, 1024), 256),
}


def build_depth_anything_v2(encoder: str = "vits", max_depth: int = 20, seed: int = 0, state_dict=None,
                            device="cuda", dtype=None, num_layers: Optional[int] = None):
    """Metric Depth-Anything-V2 (DPT head on DINOv2).  `num_layers` shrinks the backbone for tests."""
    import torch
    from transformers import DepthAnythingConfig, DepthAnythingForDepthEstimation, Dinov2Config
    hidden, layers, heads, stages, neck, fusion = _ENCODERS[encoder]
    if num_layers is not None:
        layers = num_layers
        stages = tuple(max(1, round(layers * k / 4)) for k in (1, 2, 3, 4))
    bc = Dinov2Config(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, image_size=518,
                      patch_size=14, out_features=[f"stage{s}" for s in stages], reshape_hidden_states=False,
                      apply_layernorm=True)
    cfg = DepthAnythingConfig(backbone_config=bc, reassemble_hidden_size=hidden, neck_hidden_sizes=list(neck),
                              fusion_hidden_size=fusion, head_hidden_size=32, depth_estimation_type="metric",
                              max_depth=int(max_depth))
    torch.manual_seed(seed)
    model = DepthAnythingForDepthEstimation(cfg)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    model = model.eval().to(device)
    if dtype is not None:
        model = model.to(dtype)
    return model


def depth_to_rgb_code(depth, max_depth: float, out=None):
    """f32 device tensor [N,H,W] metres -> uint8 [N,H,W,3] 16-bit depth code in R,G,B order
    (dfh:5-11 + dfh:48-61 with the channel order the render kernels read).  One launch for the batch:
    the code is per pixel, so the batch is encoded as one (N*H) x W image."""
    import torch
    assert depth.is_cuda and depth.dtype == torch.float32 and depth.dim() == 3 and depth.is_contiguous()
    N, H, W = (int(v) for v in depth.shape)
    if out is None:
        out = torch.empty((N, H, W, 3), dtype=torch.uint8, device=depth.device)
    from .depth_frames_helper import _ctx
    s = torch.cuda.current_stream(depth.device)
    group = max(1, 32767 // H)                      # a context takes at most 32767 rows
    for a in range(0, N, group):
        n = min(group, N - a)
        ctx = _ctx(depth.device.index or 0, W, n * H)
        ctx.check(_lib.load().mdvt_encode_depth(ctx.handle, depth[a].data_ptr(), 4 * W, out[a].data_ptr(), 3 * W,
                                                float(max_depth), 0, C.c_void_p(s.cuda_stream)))
    return out


_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def infer_depth(model, color_u8, input_height: int = 518, autocast_dtype=None):
    """color_u8: uint8 device tensor [N,H,W,3] RGB.  -> f32 [N,H,W] metric depth at the frame size.
    Preprocessing as the Depth-Anything-V2 image processor: resize so the short side is `input_height`
    rounded to a multiple of 14 (bicubic), ImageNet normalisation; output resized back bilinearly."""
    import torch
    import torch.nn.functional as F
    N, H, W, _ = color_u8.shape
    x = color_u8.permute(0, 3, 1, 2).to(torch.float32).div_(255.0)
    scale = input_height / min(H, W)
    h = max(14, int(round(H * scale / 14)) * 14)
    w = max(14, int(round(W * scale / 14)) * 14)
    x = F.interpolate(x, size=(h, w), mode="bicubic", align_corners=False)
    mean = torch.tensor(_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(_STD, device=x.device).view(1, 3, 1, 1)
    x = (x - mean) / std
    with torch.no_grad():
        if autocast_dtype is not None:
            with torch.autocast("cuda", dtype=autocast_dtype):
                d = model(pixel_values=x).predicted_depth
        else:
            d = model(pixel_values=x.to(next(model.parameters()).dtype)).predicted_depth
    d = F.interpolate(d.to(torch.float32).unsqueeze(1), size=(H, W), mode="bilinear", align_corners=False).squeeze(1)
    return d.contiguous()


def color_to_stereo(model, color_u8, renderer, params, *, input_height: int = 518, autocast_dtype=None,
                    want_depth: bool = False):
    """colour frames -> depth model -> 16-bit quantisation -> stereo render, all on the current stream.
    Returns (render result dict, depth_rgb code tensor)."""
    depth = infer_depth(model, color_u8, input_height, autocast_dtype)
    depth_rgb = depth_to_rgb_code(depth, renderer.max_depth)
    res = renderer.render(depth_rgb, color_u8.contiguous(), params, want_depth=want_depth)
    return res, depth_rgb
