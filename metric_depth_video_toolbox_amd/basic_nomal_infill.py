"""Device-side mirror of the reference's basic_nomal_infill.py (movie_2_3D.py's infill step after stereo_rerender): the
same names and argument meaning, on PyTorch-ROCm tensors through libmdvt_hip.so.  No CPU fallback.

    normal_infill(img, infill_mask)            basic_nomal_infill.py:87-119, one eye
    process_pair(sbs_color, sbs_mask)          basic_nomal_infill.py:124-236, on frame dumps (clip.py's formats)
    python -m metric_depth_video_toolbox_amd.basic_nomal_infill --sbs_color_video X.npy --sbs_mask_video Y.npy

The reference reads and writes FFV1 videos through OpenCV; here the inputs are the `.npy` dumps the clip driver writes
(`<depth>_stereo.npy`, `<depth>_stereo.npy_infillmask.npy`: uint8 [N, H, 2W, 3], RGB order) and the output is
`<sbs_color>_infilled.npy` (the reference appends `_infilled.mkv` to the colour video's full name, bni:141), written as
`<sbs_color>_tmp_infilled.npy` and renamed once every frame is in
(depth_frames_helper.verify_and_move, dfh:163-179).
"""
from __future__ import annotations

import argparse
import ctypes as C
import os

import numpy as np

from . import _lib


def _packed(t):
    return t.stride(-1) == 1 and t.stride(-2) == 3


def normal_infill(img, infill_mask, out=None):
    """basic_nomal_infill.normal_infill (basic_nomal_infill.py:87-119).  img, infill_mask: uint8 CUDA tensors [H,W,3] or
    [N,H,W,3] (RGB; rows and images may be strided, e.g. one half of a side-by-side frame).  Returns the image with its
    holes filled.  Unlike the reference, `img` itself is left untouched (the reference blackens and refills it in place
    and returns a new array; its caller only uses the returned one, basic_nomal_infill.py:186)."""
    import torch
    from .depth_frames_helper import _ctx
    for t in (img, infill_mask):
        assert t.is_cuda and t.dtype == torch.uint8 and t.dim() in (3, 4) and t.shape[-1] == 3 and _packed(t), \
            "uint8 CUDA [H,W,3] or [N,H,W,3] with packed RGB pixels"
    assert img.shape == infill_mask.shape
    batched = img.dim() == 4
    N = int(img.shape[0]) if batched else 1
    H, W = int(img.shape[-3]), int(img.shape[-2])
    if out is None:
        out = torch.empty(tuple(img.shape), dtype=torch.uint8, device=img.device)
    assert out.shape == img.shape and out.is_cuda and out.dtype == torch.uint8 and _packed(out)
    ctx = _ctx(img.device.index or 0, W, H)
    s = torch.cuda.current_stream(img.device)
    stride = (lambda t: t.stride(0) if batched else 0)
    ctx.check(_lib.load().mdvt_normal_infill(ctx.handle, img.data_ptr(), img.stride(-3), stride(img),
                                             infill_mask.data_ptr(), infill_mask.stride(-3), stride(infill_mask),
                                             out.data_ptr(), out.stride(-3), stride(out), N, C.c_void_p(s.cuda_stream)))
    return out


def normal_infill_sbs(sbs, sbs_mask, out=None):
    """Both eyes of side-by-side frames (basic_nomal_infill.py:172-228): uint8 CUDA [N,H,2W,3] (or [H,2W,3]) -> the same
    layout with both halves infilled."""
    import torch
    if out is None:
        out = torch.empty(tuple(sbs.shape), dtype=torch.uint8, device=sbs.device)
    W = int(sbs.shape[-2]) // 2
    for half in (slice(0, W), slice(W, 2 * W)):
        normal_infill(sbs[..., half, :], sbs_mask[..., half, :], out=out[..., half, :])
    return out


def process_pair(sbs_color_video_path: str, sbs_mask_video_path: str, max_frames: int = -1, batch: int = 8, device=None):
    """basic_nomal_infill.process_pair (basic_nomal_infill.py:124-236) on frame dumps.  A mask clip shorter than the colour
    clip means "no holes" for the remaining frames (basic_nomal_infill.py:165-167).  Returns the output path."""
    import torch
    from .clip import verify_and_move
    from .clip import open_output
    # (a clip rendered by several ranks exists as per-rank segments + index: open_output reads either form)
    if not (os.path.isfile(sbs_color_video_path) or os.path.isfile(sbs_color_video_path + ".index.json")):
        raise Exception(f"input sbs_color_video does not exist: {sbs_color_video_path}")
    if not (os.path.isfile(sbs_mask_video_path) or os.path.isfile(sbs_mask_video_path + ".index.json")):
        raise Exception(f"input sbs_mask_video does not exist: {sbs_mask_video_path}")
    color = open_output(sbs_color_video_path)
    mask = open_output(sbs_mask_video_path)
    assert color.ndim == 4 and color.shape[-1] == 3 and color.dtype == np.uint8, "uint8 [N, H, 2W, 3] expected"
    assert color.shape[1:] == mask.shape[1:], "mask and color video not same resolution"
    if color.shape[2] % 2:
        raise ValueError(f"side-by-side frames need an even width, got {color.shape[2]}")     # (the reference gives the right eye the odd column: bni:180-181)
    if max_frames == 0:
        raise ValueError("max_frames = 0: the reference still processes one frame (bni:226-228); ask for -1 (all) or a positive count")
    n = color.shape[0] if max_frames == -1 else min(color.shape[0], max_frames)
    tmp, final = sbs_color_video_path + "_tmp_infilled.npy", sbs_color_video_path + "_infilled.npy"      # bni:140-141
    out = np.lib.format.open_memmap(tmp, mode="w+", dtype=np.uint8, shape=(n,) + tuple(color.shape[1:]))
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    for a in range(0, n, batch):
        b = min(a + batch, n)
        d_color = torch.from_numpy(np.array(color[a:b])).to(dev)
        m = np.zeros((b - a,) + tuple(color.shape[1:]), np.uint8)
        have = max(0, min(b, mask.shape[0]) - a)
        if have:
            m[:have] = mask[a:a + have]
        d_out = normal_infill_sbs(d_color, torch.from_numpy(m).to(dev))
        out[a:b] = d_out.cpu().numpy()
    out.flush()
    del out
    verify_and_move(tmp, n, final)
    return final


def _is_txt(path) -> bool:
    return isinstance(path, str) and path.lower().endswith(".txt")                      # bni:29-30


def _read_list_file(path: str):
    """Stripped lines of a list file, blank lines and lines starting with '#' ignored (bni:32-43)."""
    with open(path, "r", encoding="utf-8") as f:
        return [ln.strip() for ln in f if ln.strip() and not ln.strip().startswith("#")]


def pairs_from_arguments(sbs_color_video: str, sbs_mask_video: str):
    """bni:246-260: one pair, or -- if the colour argument is a .txt list -- the pairs of two lists of equal length."""
    if not _is_txt(sbs_color_video):
        return [(sbs_color_video, sbs_mask_video)]
    if not _is_txt(sbs_mask_video):
        raise ValueError("If --sbs_color_video is a .txt file, then --sbs_mask_video must also be a .txt file.")
    colors, masks = _read_list_file(sbs_color_video), _read_list_file(sbs_mask_video)
    if len(colors) != len(masks):
        raise ValueError(f"List length mismatch: {sbs_color_video} has {len(colors)} entries, {sbs_mask_video} has {len(masks)} entries.")
    return list(zip(colors, masks))


def main(argv=None):
    p = argparse.ArgumentParser(description="Normal infill script (frame dumps)")
    p.add_argument("--sbs_color_video", type=str, required=True, help="side by side stereo frames (.npy) rendered with point clouds in the masked area, or a .txt list of them")
    p.add_argument("--sbs_mask_video", type=str, required=True, help="side by side infill mask frames (.npy), or the matching .txt list")
    p.add_argument("--max_frames", default=-1, type=int, help="quit after max_frames nr of frames", required=False)
    args = p.parse_args(argv)
    pairs = pairs_from_arguments(args.sbs_color_video, args.sbs_mask_video)
    if _is_txt(args.sbs_color_video):
        # (the reference runs two clips at a time with the GPU sections serialised, bni:262-274; here a clip is I/O and one
        #  launch set per batch, so the clips simply follow each other)
        print(f"Batch mode: {len(pairs)} pairs")
        for c_path, m_path in pairs:
            try:
                print("Done. Wrote:", process_pair(c_path, m_path, args.max_frames))
            except Exception as e:                                    # bni:270-274: surface the error, keep the other clips going
                print(f"[ERROR] A clip failed: {e}")
        return 0
    print("Done. Wrote:", process_pair(*pairs[0], args.max_frames))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
