"""Clip-level driver: the counterpart of the reference's whole `stereo_rerender.py` run (sr:318-968)
for frame dumps, with frames streamed through the GPU in batches and sharded over ranks.

On-disk formats.  The reference reads/writes FFV1-in-MKV through OpenCV (sr:327-341, 435-444, 941).  This driver does the
same when --depth_video is a Matroska file (video_io.py / libmdvt_video.so: a host-side FFV1 + Matroska reader and writer written
from the RFCs; a device-side FFV1 codec is out of scope): `x.mkv` [+ `y.mkv`] in, `x.mkv_stereo.mkv`, `x.mkv_stereo.mkv_infillmask.mkv`,
`x.mkv_stereo.mkv_depth.mkv` (the reference's names, sr:411-444) and `x.mkv_stereo.mkv_holemask.mkv` (this build's extra, grey) out,
each written as `x.mkv_tmp_...` and renamed once its frame count is right (dfh:163-179).  A colour video in another codec
(H.264 ...) cannot be decoded here.  Otherwise it works on raw frame dumps, the format for benchmarking:

  <name>.npy            uint8 [N, H, W, 3]  RGB frames (np.load(..., mmap_mode="r") compatible)
  depth dump            the same layout holding the 16-bit RGB depth code of dfh:48-61 (RGB order)
  <depth>_stereo.npy    uint8 [N, H, 2W, 3] left | right (sr:918), written as <depth>_tmp_stereo.npy and
                        renamed only when every frame was written (the reference's verify_and_move,
                        dfh:163-179); <depth>_Touchly0.npy / <depth>_Touchly1.npy with --touchly0 / --touchly1
                        (sr:411-418; the side outputs follow the main name, as the reference's do)
  <depth>_stereo.npy_holemask.npy   uint8 [N, H, 2W]   255 = hole (sr:740 / 854)
  <depth>_stereo.npy_depth.npy      uint8 [N, H, 2W, 3] B,G,R 16-bit depth code of both eyes (sr:930-939)
  <depth>_stereo.npy_infillmask.npy uint8 [N, H, 2W, 3] with --infill_mask (sr:787-808, 921-928; RGB order): the
                        normal-coloured mask -- black outside holes, inside them the screen-space normal pointing
                        out of the hole as (n+1)/2*255, inpainted from the removed-vertex normals at the splatted
                        edge points and blurred (device: StereoRerenderer.finish_infill_mask) -- or, with
                        --green_and_black_infill_mask, just the key colour (0,255,0) at holes.
  --do_basic_infill     the holes of the stereo output are filled by marching along those normals (sr:809-812).

Output-format variants (sr:406-422, 548-552, 677-702, 825-829, 914-918); the VR180 ones need 1920x1920 inputs:
  --vr180      both eyes rendered with the square VR180 camera, each put through convert_to_equirectangular
  --touchly0   = --vr180 plus a third image, the left eye's Touchly reverse-depth plane: [N, H, 3W, 3]
  --touchly1   colour over Touchly depth, [N, 2H, W, 3]; straight from the input without a pose file (sr:548-552),
               else a mono render of the posed mesh (sr:673-691)

Side-cars are the reference's own JSON formats: xfov list (sr:351-359), convergence list with NaNs
(sr:343-349), transformations list of 4x4 (sr:362-373).

Pipelining: per batch, H2D on a copy stream from pinned staging -> render on the compute stream ->
D2H on a second copy stream into pinned staging, HIP events order them; the file <-> pinned-staging copies are
pread / pwrite calls (no mapping, no intermediate array) fanned out over helper threads in frame sub-ranges, one batch
ahead of / behind the submitting thread; three staging sets rotate.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import numpy as np

from . import distributed as D
from .stereo_rerender import StereoRerenderer, curve_fit, fill_nan_with_closest


def load_clip_parameters(n_frames: int, W: int, H: int, *, xfov=None, xfov_file=None, convergence_file=None,
                         transformation_file=None, transformation_lock_frame: int = 0, pupillary_distance=63,
                         max_depth=100, master_xfov: float = 45.0, render_as_pointcloud=False, remove_edges=False,
                         infill_mask=False, dont_place_points_in_edges=False, vr180=False, touchly0=False,
                         touchly1=False, touchly_max_depth=5.0, touchly_min_depth=0.0, do_basic_infill=False,
                         dont_remove_edges=False, n_use: Optional[int] = None) -> D.ClipParameters:
    """What sr:318-373 does before the loop, on rank 0.  n_frames is the length of the input clip: the side-cars are
    checked against it and the convergence curve is smoothed over all of it, as the reference does (sr:343-349, 403);
    n_use (--max_frames) then keeps the first frames only."""
    if xfov is None and xfov_file is None:
        raise ValueError("Error: Either --xfov_file, --xfov or --yfov must be provided.")            # sr:319-320
    if xfov_file is not None:
        if not os.path.isfile(xfov_file):
            raise FileNotFoundError(f"XFOV file not found: {xfov_file}")
        with open(xfov_file) as fh:
            xf = json.load(fh)
        if not isinstance(xf, list) or not all(isinstance(x, (int, float)) for x in xf):
            raise ValueError("XFOV file must contain a list of numbers.")                             # sr:358-359
        if len(xf) != n_frames:
            raise ValueError(f"XFOV file must have the same number of frames as the input video ({n_frames} vs xfov={len(xf)}).")
        xfovs = np.asarray(xf, np.float64)
    else:
        xfovs = np.full(n_frames, float(xfov))
    conv = np.zeros(n_frames)
    if convergence_file is not None:
        if not os.path.isfile(convergence_file):
            raise FileNotFoundError(f"Convergence file not found: {convergence_file}")
        with open(convergence_file) as fh:
            vals = fill_nan_with_closest([float(v) for v in json.load(fh)])                            # sr:348
        conv = np.asarray(curve_fit(vals), np.float64)                                                 # sr:349
        if len(conv) < n_frames:            # (the reference indexes the list by frame, sr:709: a short list fails there)
            raise ValueError("convergence file has fewer entries than frames")
        conv = conv[:n_frames]
    T = None
    if transformation_file is not None:
        if not os.path.isfile(transformation_file):
            raise Exception("input transformation_file does not exist")                                # sr:364-365
        with open(transformation_file) as fh:
            T = D.rebase_on_lock_frame(json.load(fh), transformation_lock_frame)                        # sr:369-373
        if len(T) < n_frames:
            raise ValueError("transformation file has fewer entries than frames")
        T = T[:n_frames]
    if n_use is not None and n_use < n_frames:
        xfovs, conv = xfovs[:n_use], conv[:n_use]
        if T is not None:
            T = T[:n_use]
        n_frames = n_use
    rm = bool(infill_mask or remove_edges or do_basic_infill) and not dont_remove_edges                # sr:568-573
    flags = (1 if render_as_pointcloud else 0) | (2 if rm else 0) | (4 if (rm and not dont_place_points_in_edges) else 0) \
        | (8 if infill_mask else 0)
    if touchly0:
        vr180 = True                                                                                   # sr:406-407
    if touchly0 and touchly1:
        raise ValueError("--touchly0 and --touchly1 are different output formats; pick one")
    flags |= (16 if vr180 else 0) | (32 if touchly0 else 0) | (64 if touchly1 else 0) | (128 if do_basic_infill else 0)
    if not float(touchly_max_depth) > float(touchly_min_depth):
        raise ValueError("touchly_max_depth must exceed touchly_min_depth")
    return D.ClipParameters(W, H, n_frames, pupillary_distance / 1000, float(max_depth), float(master_xfov), flags,
                            xfovs, conv, T, float(touchly_max_depth), float(touchly_min_depth))


def output_shape(clip: D.ClipParameters):
    """Frame shape (rows, cols) of the main output for the clip's format variant (sr:409-422)."""
    f = clip.mode_flags
    if f & 64:
        return 2 * clip.H, clip.W           # touchly1: vconcat([colour, depth])
    if f & 32:
        return clip.H, 3 * clip.W           # touchly0: hconcat([left, right, left depth])
    return clip.H, 2 * clip.W


def renderer_for(clip: D.ClipParameters, device: Optional[int] = None) -> StereoRerenderer:
    f = clip.mode_flags
    pd = clip.ipd_m * 1000
    if f & 64:
        pd = 0                                  # touchly1 draws the posed mesh once, unshifted (sr:673-683)
    if abs(pd - round(pd)) < 1e-9:
        pd = int(round(pd))                     # --pupillary_distance is an int in mm (sr:288)
    return StereoRerenderer(clip.W, clip.H, device=device, pupillary_distance=pd,
                            max_depth=clip.max_depth, master_xfov=clip.master_xfov,
                            render_as_pointcloud=bool(f & 1), remove_edges=bool(f & 2) and not bool(f & 8),
                            infill_mask=bool(f & 8), dont_place_points_in_edges=not bool(f & 4),
                            do_basic_infill=bool(f & 128) and bool(f & 2), dont_remove_edges=not bool(f & 2))


def frame_param_records(r: StereoRerenderer, clip: D.ClipParameters, lo: int, hi: int):
    out = []
    for t in range(lo, hi):
        cd = float(clip.convergence[t])
        if clip.mode_flags & 64:
            cd = 0.0                                # no toe-in in the touchly1 branch (sr:673-702)
        out.append(r.frame_params(xfov=float(clip.xfov[t]),
                                  convergence_distance=None if (cd == 0.0 or math.isnan(cd)) else cd,
                                  transformation=None if clip.transformations is None else clip.transformations[t],
                                  vr180=bool(clip.mode_flags & 16)))
    return out


def _post_vr180(r, clip, recs, d_sbs, d_z, d_post, n):
    """sr:914-918 on the device: every image of the frame through convert_to_equirectangular with that
    frame's render fov, then hconcat.  d_post: [n, H, 2W or 3W, 3]."""
    from .stereo_rerender import convert_to_equirectangular, touchly_depth, vr180_render_fov
    W = clip.W
    fovs = [vr180_render_fov(np.array([rec.K[k] for k in range(9)]).reshape(3, 3)) for rec in recs[:n]]
    a = 0
    while a < n:                                        # runs of equal fov share one remap call per eye
        b = a + 1
        while b < n and fovs[b] == fovs[a]:
            b += 1
        for eye in range(2):
            convert_to_equirectangular(d_sbs[a:b, :, eye * W:(eye + 1) * W], fovs[a], out=d_post[a:b, :, eye * W:(eye + 1) * W])
        a = b
    if clip.mode_flags & 32:                            # touchly0: the left eye's depth plane (sr:825-829)
        import torch
        plane = torch.empty((clip.H, W, 3), dtype=torch.uint8, device=d_sbs.device)
        for f in range(n):
            touchly_depth(d_z[f, :, :W], clip.touchly_max_depth, clip.touchly_min_depth, zero_is_far=True, out=plane)
            convert_to_equirectangular(plane, fovs[f], out=d_post[f, :, 2 * W:])
    return d_post[:n]


def _post_touchly1(r, clip, scales, d_depth_in, d_color_in, d_sbs, d_mask, d_z, d_post, n, posed):
    """sr:548-552 (no pose file: colour over the decoded depth) / sr:673-691 (mono render of the posed mesh)."""
    import torch
    from . import depth_frames_helper as dfh
    from .stereo_rerender import touchly_depth
    W, H = clip.W, clip.H
    for f in range(n):
        if not posed:
            d_post[f, :H] = d_color_in[f]
            z = dfh.decode_rgb_depth_frame(d_depth_in[f], clip.max_depth, True, depth_scale=scales[f])
            touchly_depth(z, clip.touchly_max_depth, clip.touchly_min_depth, out=d_post[f, H:])
        else:
            img = d_sbs[f, :, :W]
            if r.key_rgb != (0, 0, 0):      # the reference keeps the raw render here: holes show the key colour (sr:683-684)
                key = torch.tensor(r.key_rgb, dtype=torch.uint8, device=img.device)
                img = torch.where((d_mask[f, :, :W] > 0)[..., None], key, img)
            d_post[f, :H] = img
            touchly_depth(d_z[f, :, :W], clip.touchly_max_depth, clip.touchly_min_depth, zero_is_far=True, out=d_post[f, H:])
    return d_post[:n]


class _RawFrames:
    """Frame dumps that live in a file (np.load(..., mmap_mode=...) / open_memmap arrays) are read and written with
    pread / pwrite straight between the file and the pinned staging buffers: one kernel copy per batch and direction,
    no page-fault storm through a mapping and no intermediate NumPy array.  Anything else (plain arrays, lists) is
    indexed the ordinary way."""

    def __init__(self, arr, writable: bool):
        self.arr = arr
        self.fd = -1
        if isinstance(arr, (VideoFrames, VideoSink)):           # a video file: its own decode / encode-and-append
            self.read_into = arr.read_into if isinstance(arr, VideoFrames) else None
            self.write_from = arr.write_from if isinstance(arr, VideoSink) else None
            return
        if isinstance(arr, np.memmap) and arr.flags["C_CONTIGUOUS"] and getattr(arr, "filename", None) and arr.ndim >= 2:
            # A slice of a memmap (depth[k:]) is still an np.memmap with the parent's filename AND the parent's `offset`:
            # the file position of its first byte is the root mapping's offset plus the distance of the data pointers.
            root = arr
            while isinstance(getattr(root, "base", None), np.memmap):
                root = root.base
            try:
                delta = int(arr.__array_interface__["data"][0]) - int(root.__array_interface__["data"][0])
                if delta < 0 or len(arr) == 0:
                    raise OSError("not a forward slice of its mapping")
                self.fd = os.open(str(arr.filename), os.O_RDWR if writable else os.O_RDONLY)
                self.base = int(root.offset) + delta
                self.frame_bytes = int(arr[0].nbytes)
            except (OSError, KeyError, TypeError):
                self.fd = -1

    def read_into(self, dst: np.ndarray, a: int, n: int):
        if self.fd < 0:
            dst[...] = self.arr[a:a + n]
            return
        mv = memoryview(dst).cast("B")
        off, done, total = self.base + a * self.frame_bytes, 0, n * self.frame_bytes
        while done < total:
            got = os.preadv(self.fd, [mv[done:total]], off + done)
            if got <= 0:
                raise IOError("short read from a frame dump")
            done += got

    def write_from(self, src: np.ndarray, a: int, n: int):
        if self.fd < 0:
            self.arr[a:a + n] = src
            return
        mv = memoryview(src).cast("B")
        off, done, total = self.base + a * self.frame_bytes, 0, n * self.frame_bytes
        while done < total:
            done += os.pwritev(self.fd, [mv[done:total]], off + done)

    def close(self):
        if self.fd >= 0:
            os.close(self.fd)
            self.fd = -1


class VideoFrames:
    """An FFV1-in-Matroska file as the read-only [N, H, W, 3] uint8 frame array render_clip / open_output expect (RGB order).
    Reads go through a small pool of decoders: a sequential reader continues where its decoder stands, anything else is a
    seek (free in an intra-only stream; an inter-coded one -- FFmpeg's default, a key frame every 12 -- decodes forward from the
    last key frame, so it gets ONE decoder and all the slice threads instead of several decoders leap-frogging)."""

    def __init__(self, path: str, readers: int = 2):
        import threading
        from . import video_io
        first = video_io.VideoReader(path)
        self.path, self.fps, self.info = path, first.fps, first.info
        self.shape = (first.frames, first.height, first.width, 3)
        self.dtype, self.ndim = np.dtype(np.uint8), 4
        n = max(1, int(readers)) if first.info.intra else 1
        cores = _usable_cores()
        first.threads = max(1, cores // (2 * n)) if n > 1 else 0
        self._readers = [first] + [video_io.VideoReader(path, threads=first.threads) for _ in range(n - 1)]
        self._pos = [0] * n
        self._busy = [False] * n
        self._cv = threading.Condition()

    def __len__(self):
        return self.shape[0]

    def read_into(self, dst: np.ndarray, a: int, n: int):
        with self._cv:
            while True:
                free = [k for k in range(len(self._readers)) if not self._busy[k]]
                if free:
                    k = next((k for k in free if self._pos[k] == a), None)
                    if k is None:
                        behind = [k for k in free if self._pos[k] <= a]
                        k = max(behind, key=lambda q: self._pos[q]) if behind else free[0]
                    self._busy[k] = True
                    break
                self._cv.wait()
        try:
            r = self._readers[k]
            if self._pos[k] != a:
                r.seek(a)
            for i in range(n):
                if not r.read_into(dst[i]):
                    raise IOError(f"{self.path}: ends at frame {a + i}")
            self._pos[k] = a + n
        except BaseException:
            self._pos[k] = -1 << 60         # unknown position: the next use seeks
            raise
        finally:
            with self._cv:
                self._busy[k] = False
                self._cv.notify()

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            lo, hi, step = idx.indices(len(self))
            if step != 1:
                return np.stack([self[t] for t in range(lo, hi, step)]) if hi > lo else np.empty((0,) + self.shape[1:], np.uint8)
            out = np.empty((max(0, hi - lo),) + self.shape[1:], np.uint8)
            if hi > lo:
                self.read_into(out, lo, hi - lo)
            return out
        t = int(idx)
        if t < 0:
            t += len(self)
        out = np.empty((1,) + self.shape[1:], np.uint8)
        self.read_into(out, t, 1)
        return out[0]

    def __array__(self, dtype=None, copy=None):
        a = self[0:len(self)]
        return a if dtype is None else a.astype(dtype)

    def close(self):
        for r in self._readers:
            r.close()


class VideoSink:
    """An output video being written: render_clip's store threads hand it frame sub-ranges in any order (write_from), each
    thread encodes its frames itself (one FFV1 packet per frame, slices on one thread: the parallelism is across frames) and
    the packets are appended in frame order.  Grey frames (the hole mask) are written as R = G = B."""

    def __init__(self, path: str, width: int, height: int, fps: float, grey: bool = False, slices=(4, 4), bgr: bool = False):
        import threading
        from . import video_io
        self._vio, self.bgr = video_io, bgr
        self.slices = (min(slices[0], width), min(slices[1], height))
        self._w = video_io.VideoWriter(path, width, height, fps, slices=self.slices)
        self.path, self.grey, self.shape_hw = path, grey, (height, width)
        self._pending, self._next, self._lock = {}, 0, threading.Lock()

    def write_from(self, src: np.ndarray, a: int, n: int):
        for i in range(n):
            f = src[i]
            if self.grey:
                f = np.repeat(f[..., None], 3, axis=-1)
            pkt, _ = self._vio.encode_frame(f, slices=self.slices, threads=1, bgr=self.bgr)
            with self._lock:
                self._pending[a + i] = pkt
                while self._next in self._pending:
                    self._w.write_packet(self._pending.pop(self._next))
                    self._next += 1

    def close(self) -> int:
        with self._lock:
            if self._pending:
                missing = self._next
                self._pending.clear()
                self._w.close()
                raise RuntimeError(f"{self.path}: frame {missing} was never written")
            return self._w.close()


def render_clip(depth_frames, color_frames, out_sbs, out_mask, clip: D.ClipParameters, *, lo: int = 0,
                hi: Optional[int] = None, batch: int = 16, out_depth_rgb=None, out_infill=None, green_and_black: bool = False,
                device: Optional[int] = None, out_base: int = 0, io_threads: int = 12, out_infilled=None):
    """Render frames [lo, hi) of a clip.  depth_frames / color_frames / out_*: array-likes indexed
    [frame] (NumPy arrays or memmaps, uint8); frame t is written to out_*[t - out_base] (a rank that owns the output
    segment [lo, hi) passes out_base = lo).  out_infilled (optional, [frames, H, 2W, 3]): the stereo frames after
    basic_nomal_infill.normal_infill of both eyes with the finished infill mask -- movie_2_3D.py's next step, run here
    while frame and mask are still on the device instead of through two more video files.  Returns (frames, seconds, hole_pixels)."""
    import time
    import torch
    from . import depth_frames_helper as dfh

    hi = clip.n_frames if hi is None else hi
    W, H = clip.W, clip.H
    r = renderer_for(clip, device)
    dev = torch.device("cuda", r.device)
    recs = frame_param_records(r, clip, lo, hi)
    B = max(1, min(batch, hi - lo))
    want_zrgb = out_depth_rgb is not None
    vr180, touchly0, touchly1 = bool(clip.mode_flags & 16), bool(clip.mode_flags & 32), bool(clip.mode_flags & 64)
    posed = clip.transformations is not None
    skip_render = touchly1 and not posed                      # sr:548-552: "fast path we can skip the full render pass"
    want_z = want_zrgb or touchly0 or (touchly1 and posed)
    basic_infill = bool(clip.mode_flags & 128) and not touchly1
    want_infill = out_infill is not None and not skip_render   # (the reference writes no infill-mask frame on its fast path)
    want_infilled = out_infilled is not None
    if want_infilled and (not (clip.mode_flags & 2) or not (clip.mode_flags & 8) or vr180 or touchly1 or basic_infill):
        raise ValueError("normal_infill needs --infill_mask with edge removal and the plain stereo output "
                         "(basic_nomal_infill.py consumes <depth>_stereo and its infill mask)")
    if not (clip.mode_flags & 2):
        green_and_black = True       # --infill_mask --dont_remove_edges: no normals to show, the mask is the key colour at holes
        basic_infill = False
    if want_infilled and green_and_black:
        raise ValueError("normal_infill marches along the normal-coloured mask: not with --green_and_black_infill_mask")
    want_seed = ((want_infill and not green_and_black) or basic_infill or want_infilled) and bool(clip.mode_flags & 2)
    # without --infill_mask the key colour is black and every black pixel counts as "to fill" (sr:803-805): the front
    # then has to cross the whole frame, as the reference's own cv2.inpaint call does
    telea_rounds = 0 if (clip.mode_flags & 8) else W + H
    oH, oW = output_shape(clip)
    post = vr180 or touchly1                                   # the main output is a post-processed image

    def pinned(shape, dtype):
        return torch.empty(shape, dtype=dtype, pin_memory=True)

    NSETS = 3
    sets = []
    for _ in range(NSETS):
        sets.append({
            "h_d": pinned((B, H, W, 3), torch.uint8), "h_c": pinned((B, H, W, 3), torch.uint8),
            "h_sbs": pinned((B, oH, oW, 3), torch.uint8), "h_mask": pinned((B, H, 2 * W), torch.uint8),
            "h_counts": pinned((B, 2), torch.int32), "h_rem": pinned((2, B), torch.int32),
            "h_zrgb": pinned((B, H, 2 * W, 3), torch.uint8) if want_zrgb else None,
            "h_seed": pinned((B, H, 2 * W, 3), torch.uint8) if want_infill else None,
            "h_infilled": pinned((B, H, 2 * W, 3), torch.uint8) if want_infilled else None,
            "d_infilled": torch.empty((B, H, 2 * W, 3), dtype=torch.uint8, device=dev) if want_infilled else None,
            "d_infill": torch.empty((B, H, 2 * W, 3), dtype=torch.uint8, device=dev) if (want_seed or want_infill) else None,
            "d_d": torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev),
            "d_c": torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev),
            "d_sbs": torch.empty((B, H, 2 * W, 3), dtype=torch.uint8, device=dev),
            "d_mask": torch.empty((B, H, 2 * W), dtype=torch.uint8, device=dev),
            "d_z": torch.empty((B, H, 2 * W), dtype=torch.float32, device=dev) if want_z else None,
            "d_zrgb": torch.empty((B, H, 2 * W, 3), dtype=torch.uint8, device=dev) if want_zrgb else None,
            "d_post": torch.zeros((B, oH, oW, 3), dtype=torch.uint8, device=dev) if post else None,
            "in_done": torch.cuda.Event(), "render_done": torch.cuda.Event(), "out_done": torch.cuda.Event(),
            "loaded": None, "stored": None,
        })
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    s_cmp = torch.cuda.current_stream(dev)

    # Host side of the pipeline: the two big memcpys per batch (frame dump -> pinned inputs, pinned outputs -> output
    # dumps; 29 MB per 1080p frame, NumPy drops the GIL inside them) run on helper threads, one batch ahead / behind
    # the thread that feeds the GPU.  Three staging sets keep the three stages out of each other's buffers.
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=4)        # one task per batch and direction ...
    io_pool = ThreadPoolExecutor(max_workers=max(2, int(io_threads)))    # ... each of which fans its file copies out in frame sub-ranges

    def fan(fn, host, a, n, parts):
        """fn(host[k0:k1], a + k0, k1 - k0) over `parts` frame sub-ranges in parallel: one thread moves ~2-3 GB/s into
        fresh page-cache pages, a 1080p batch needs several."""
        parts = max(1, min(parts, n))
        edges = [(n * k) // parts for k in range(parts + 1)]
        return [io_pool.submit(fn, host[edges[k]:edges[k + 1]], a + edges[k], edges[k + 1] - edges[k]) for k in range(parts)]

    f_depth, f_color = _RawFrames(depth_frames, False), _RawFrames(color_frames, False)
    f_sbs, f_mask = _RawFrames(out_sbs, True), _RawFrames(out_mask, True)
    f_zrgb = _RawFrames(out_depth_rgb, True) if out_depth_rgb is not None else None
    f_infill = _RawFrames(out_infill, True) if out_infill is not None else None
    f_infilled = _RawFrames(out_infilled, True) if out_infilled is not None else None

    trace = [] if os.environ.get("MDVT_CLIP_TRACE") else None

    def load(st, a, n):
        t_l = time.perf_counter()
        st["in_done"].synchronize()                 # the H2D copies that last read these pinned buffers are done
        t_l1 = time.perf_counter()
        if color_frames is depth_frames and isinstance(depth_frames, VideoFrames):      # sr:508-509 on one video: decode it once
            for j in fan(f_depth.read_into, st["h_d"][:n].numpy(), a, n, 2):
                j.result()
            np.copyto(st["h_c"][:n].numpy(), st["h_d"][:n].numpy())
            jobs = []
        else:
            jobs = fan(f_color.read_into, st["h_c"][:n].numpy(), a, n, 2) + fan(f_depth.read_into, st["h_d"][:n].numpy(), a, n, 2)
        for j in jobs:
            j.result()
        if trace is not None:
            trace.append(("load", a, t_l - t0, t_l1 - t_l, time.perf_counter() - t_l1))

    def store(st, a, n):
        t_s = time.perf_counter()
        st["out_done"].synchronize()
        t_s1 = time.perf_counter()
        o = a - out_base
        jobs = fan(f_sbs.write_from, st["h_sbs"][:n].numpy(), o, n, 6) + fan(f_mask.write_from, st["h_mask"][:n].numpy(), o, n, 1)
        h = int(st["h_counts"][:n].sum())           # hole pixels, counted on the device (mdvt_io.hole_counts)
        if st.get("check_rem") and int(st["h_rem"][:, :n].sum()) != 0:
            redo.append((a, n))                     # a hole deeper than the default 256 levels: finished again below
        if want_zrgb:
            jobs += fan(f_zrgb.write_from, st["h_zrgb"][:n].numpy(), o, n, 3)
        if want_infill:
            jobs += fan(f_infill.write_from, st["h_seed"][:n].numpy(), o, n, 3)
        if want_infilled:
            jobs += fan(f_infilled.write_from, st["h_infilled"][:n].numpy(), o, n, 3)
        for j in jobs:
            j.result()
        store_done.append(time.perf_counter())
        if trace is not None:
            trace.append(("store", a, t_s - t0, t_s1 - t_s, time.perf_counter() - t_s1))
        return h

    def device_stage(st, a, n, rounds, count_holes):
        """Everything the compute stream does for one batch once its inputs are on the device (sr:512-941 minus file I/O):
        render, SBS depth code, infill-mask completion, basic infill, green / black mask, VR180 / Touchly post.  Shared by
        the streaming loop and the re-do of batches whose holes were deeper than the default level budget."""
        brecs = recs[a - lo:a - lo + n]
        res = None
        if skip_render:
            st["d_mask"][:n].zero_()
        else:
            res = r.render(st["d_d"][:n], st["d_c"][:n], brecs, out_sbs=st["d_sbs"][:n],
                           out_mask=st["d_mask"][:n], want_depth=want_z, out_depth=st["d_z"][:n] if want_z else None,
                           want_seed=want_seed, want_hole_counts=count_holes)
        if want_zrgb:                               # sr:930-939: both eyes through the 16-bit code, B,G,R
            for f in range(n):
                dfh.encode_depth_frame(st["d_z"][f], clip.max_depth, bgr=True, out=st["d_zrgb"][f])
        st["d_rem"] = None
        if want_seed and res is not None:              # sr:803-808: finish the normal-coloured mask on the device, per eye
            _, st["d_rem"] = r.finish_infill_mask_sbs(res["seed"], out=st["d_infill"][:n], max_rounds=rounds, want_remaining=True)
            if basic_infill:                           # sr:809-812: march along the normals into the holes, all frames of an eye at once
                from .stereo_rerender import infill_using_mask_normals
                for eye in range(2):
                    sl = slice(eye * W, (eye + 1) * W)
                    infill_using_mask_normals(st["d_sbs"][:n, :, sl], st["d_mask"][:n, :, sl], st["d_infill"][:n, :, sl], out=st["d_sbs"][:n, :, sl])
        if want_infilled and res is not None:          # basic_nomal_infill.py:172-228: both eyes of every frame
            from .basic_nomal_infill import normal_infill_sbs
            normal_infill_sbs(st["d_sbs"][:n], st["d_infill"][:n], out=st["d_infilled"][:n])
        if want_infill and green_and_black and res is not None:      # sr:787-793: the key colour at holes, black elsewhere
            key = torch.tensor(r.key_rgb, dtype=torch.uint8, device=dev)
            torch.mul((st["d_mask"][:n] > 0)[..., None], key, out=st["d_infill"][:n])
        main = st["d_sbs"][:n]
        if touchly1:
            main = _post_touchly1(r, clip, [rec.depth_scale for rec in brecs], st["d_d"], st["d_c"], st["d_sbs"], st["d_mask"], st["d_z"], st["d_post"], n, posed)
        elif vr180:
            main = _post_vr180(r, clip, brecs, st["d_sbs"], st["d_z"], st["d_post"], n)
        return res, main

    redo = []
    store_done = []
    starts = list(range(lo, hi, B))
    t0 = time.perf_counter()
    if starts:
        sets[0]["loaded"] = pool.submit(load, sets[0], starts[0], min(B, hi - starts[0]))
    stores = []
    for k, a in enumerate(starts):
        n = min(B, hi - a)
        st = sets[k % NSETS]
        st["loaded"].result()                       # this batch's inputs are in pinned memory
        if k + 1 < len(starts):                     # next batch's inputs: start copying now
            nx = sets[(k + 1) % NSETS]
            nx["loaded"] = pool.submit(load, nx, starts[k + 1], min(B, hi - starts[k + 1]))
        if st["stored"] is not None:
            st["stored"].result()                   # this set's pinned outputs have been written out
        with torch.cuda.stream(s_in):
            s_in.wait_event(st["render_done"])      # device inputs free again
            st["d_d"][:n].copy_(st["h_d"][:n], non_blocking=True)
            st["d_c"][:n].copy_(st["h_c"][:n], non_blocking=True)
            st["in_done"].record(s_in)
        s_cmp.wait_event(st["in_done"])
        s_cmp.wait_event(st["out_done"])            # device outputs free again
        res, main = device_stage(st, a, n, telea_rounds, True)
        st["render_done"].record(s_cmp)
        with torch.cuda.stream(s_out):
            s_out.wait_event(st["render_done"])
            st["h_sbs"][:n].copy_(main, non_blocking=True)
            st["h_mask"][:n].copy_(st["d_mask"][:n], non_blocking=True)
            if res is not None:
                st["h_counts"][:n].copy_(res["hole_counts"], non_blocking=True)
                res["hole_counts"].record_stream(s_out)
            else:
                st["h_counts"][:n].zero_()
            if want_zrgb:
                st["h_zrgb"][:n].copy_(st["d_zrgb"][:n], non_blocking=True)
            if want_infill and res is not None:
                st["h_seed"][:n].copy_(st["d_infill"][:n], non_blocking=True)
            if want_infilled:
                st["h_infilled"][:n].copy_(st["d_infilled"][:n], non_blocking=True)
            st["check_rem"] = st.get("d_rem") is not None
            if st["check_rem"]:
                st["h_rem"][:, :n].copy_(st["d_rem"], non_blocking=True)
                st["d_rem"].record_stream(s_out)
            st["out_done"].record(s_out)
        st["stored"] = pool.submit(store, st, a, n)
        stores.append(st["stored"])
    holes = sum(f.result() for f in stores)
    torch.cuda.synchronize(dev)
    # steady-state rate: from the moment the first batch has left the pipeline (the first render call pays the one-time
    # code-object load, ~0.2 s) to the moment the last one has
    if len(store_done) >= 3:
        render_clip.last_steady_fps = (hi - lo - min(B, hi - lo)) / max(store_done[-1] - store_done[0], 1e-9)
    else:
        render_clip.last_steady_fps = float("nan")
    for a, n in redo:
        # cv2.inpaint fills every masked pixel (sr:806).  The default 256 levels did not reach the bottom of a hole of this
        # batch: run the batch's whole device stage again and let the front travel as far as the frame is large (W + H
        # levels always suffice); every output the completion feeds is written again (the infill mask, and with
        # --do_basic_infill the stereo frames, through the VR180 post-processing if that is on).
        st = sets[0]
        f_depth.read_into(st["h_d"][:n].numpy(), a, n)
        f_color.read_into(st["h_c"][:n].numpy(), a, n)
        st["d_d"][:n].copy_(st["h_d"][:n])
        st["d_c"][:n].copy_(st["h_c"][:n])
        _, main = device_stage(st, a, n, W + H, False)
        if basic_infill:
            f_sbs.write_from(main.cpu().numpy(), a - out_base, n)
        if want_infill:
            f_infill.write_from(st["d_infill"][:n].cpu().numpy(), a - out_base, n)
        if want_infilled:
            f_infilled.write_from(st["d_infilled"][:n].cpu().numpy(), a - out_base, n)
    dt = time.perf_counter() - t0
    if trace is not None:
        for ev in sorted(trace, key=lambda e: e[2]):
            print("clip trace: %-5s frame %4d  t=%7.1f ms  wait %6.1f ms  copy %6.1f ms" % (ev[0], ev[1], ev[2] * 1e3, ev[3] * 1e3, ev[4] * 1e3))
    pool.shutdown()
    io_pool.shutdown()
    for f in (f_depth, f_color, f_sbs, f_mask, f_zrgb, f_infill, f_infilled):
        if f is not None:
            f.close()
    r.close()
    return hi - lo, dt, holes


def npy_shape(path: str):
    """Shape recorded in a .npy header (no mapping: an empty segment cannot be mapped)."""
    with open(path, "rb") as fh:
        major, _ = np.lib.format.read_magic(fh)
        shape, _, _ = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(fh)
    return shape


def frames_in(path: str) -> int:
    """Frames in an output file of either kind (a .npy header, or the frames indexed in a Matroska file: what the reference asks
    cv2.CAP_PROP_FRAME_COUNT for, dfh:169)."""
    from . import video_io
    if video_io.is_matroska(path):
        with video_io.VideoReader(path) as r:
            return r.frames
    return npy_shape(path)[0]


def verify_and_move(tmp_path: str, expected_frames: int, final_path: str):
    """The reference's tmp -> final protocol (dfh:163-179): rename only if the frame count matches."""
    got = frames_in(tmp_path)
    if got != expected_frames:
        raise RuntimeError(f"{tmp_path}: {got} frames written, expected {expected_frames}; left in place")
    os.replace(tmp_path, final_path)


OUTPUT_KINDS = {"sbs": "", "mask": "_holemask", "depth": "_depth", "infill": "_infillmask", "infilled": "_infilled"}


def segment_path(path: str, rank: int, world: int) -> str:
    """File of rank `rank`'s output segment: `<path>` itself for a single rank, else `<path>.rank<r>of<R><ext of path>`."""
    return path if world == 1 else f"{path}.rank{rank}of{world}{os.path.splitext(path)[1] or '.npy'}"


def plan_outputs(depth_path: str, clip: D.ClipParameters, world: int, *, create_sbs_depth_video: bool = False,
                 infill_mask: bool = False, normal_infill: bool = False, ext: str = ".npy"):
    """Names and shapes of everything a run writes (pure host logic).  With one rank every output is one `.npy` dump, as the
    reference writes one file per output (sr:411-444).  With R ranks every rank owns the SEGMENT of each output that holds
    its contiguous frame range -- its own file `<output>.rank<r>of<R>.npy`, written as `<tmp>.rank<r>of<R>.npy` and renamed
    by the rank itself once all its frames are in (dfh:163-179 per segment) -- plus one small `<output>.index.json` from
    rank 0: like the reference's one-file-per-scene outputs (m23d:433-452), no two writers share a file (writers of one
    file serialise on its inode lock at ~13 GB/s on the GPU box: tools/probe/io_probe.py).  open_output() / merge_output()
    read either form.  Returns {kind: dict(final, tmp, frame_shape, segments=[(rank, lo, hi)])} ."""
    N, W, H = clip.n_frames, clip.W, clip.H
    kind = "Touchly1" if clip.mode_flags & 64 else ("Touchly0" if clip.mode_flags & 32 else "stereo")     # sr:411-422
    final = depth_path + f"_{kind}{ext}"              # ext ".mkv": the reference's own names (sr:411-444)
    tmp = depth_path + f"_tmp_{kind}{ext}"
    oH, oW = output_shape(clip)
    shapes = {"sbs": (oH, oW, 3), "mask": (H, 2 * W)}
    if create_sbs_depth_video:
        shapes["depth"] = (H, 2 * W, 3)
    if infill_mask and not ((clip.mode_flags & 64) and clip.transformations is None):
        shapes["infill"] = (H, 2 * W, 3)
    if normal_infill:
        shapes["infilled"] = (H, 2 * W, 3)
    segs = [(r,) + tuple(D.frame_range(r, world, N)) for r in range(world)]
    return {k: dict(final=final + (OUTPUT_KINDS[k] + ext if OUTPUT_KINDS[k] else ""), tmp=tmp + (OUTPUT_KINDS[k] + ext if OUTPUT_KINDS[k] else ""),
                    frame_shape=shp, segments=segs, frames=N)
            for k, shp in shapes.items()}


class SegmentedFrames:
    """Read-only view of an output written as per-rank segments: indexable like the single [N, ...] array."""

    def __init__(self, parts, bounds):
        self.parts, self.bounds = parts, bounds          # bounds[k] = first frame of part k; bounds[-1] = N
        self.shape = (bounds[-1],) + tuple(parts[0].shape[1:])
        self.dtype = parts[0].dtype
        self.ndim = len(self.shape)

    def __len__(self):
        return self.bounds[-1]

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            lo, hi, step = idx.indices(len(self))
            if step != 1:
                return np.stack([self[t] for t in range(lo, hi, step)])
            out = [p[max(lo, b0) - b0:min(hi, b1) - b0] for p, b0, b1 in zip(self.parts, self.bounds[:-1], self.bounds[1:])
                   if max(lo, b0) < min(hi, b1)]
            return np.concatenate(out) if out else np.empty((0,) + self.shape[1:], self.dtype)
        t = int(idx)
        if t < 0:
            t += len(self)
        k = int(np.searchsorted(self.bounds, t, side="right")) - 1
        return self.parts[k][t - self.bounds[k]]

    def __array__(self, dtype=None, copy=None):
        a = self[0:len(self)]
        return a if dtype is None else a.astype(dtype)


def open_output(path: str, mmap_mode: Optional[str] = "r"):
    """An output of run(): the single dump `<path>` or the per-rank segments named by `<path>.index.json` (run() leaves only the
    form it wrote; should both exist -- files copied together by hand -- the newer one is taken)."""
    from . import video_io

    def one(f, mapped=True):
        return VideoFrames(f, readers=1) if video_io.is_matroska(f) else np.load(f, mmap_mode=mmap_mode if mapped else None)
    ip = path + ".index.json"
    if os.path.exists(path) and not (os.path.exists(ip) and os.path.getmtime(ip) > os.path.getmtime(path)):
        return one(path)
    with open(path + ".index.json") as fh:
        idx = json.load(fh)
    here = os.path.dirname(path)
    parts = [one(os.path.join(here, s["file"]), s["hi"] > s["lo"]) for s in idx["segments"]]
    bounds = [s["lo"] for s in idx["segments"]] + [idx["frames"]]
    for p, s in zip(parts, idx["segments"]):
        if p.shape[0] != s["hi"] - s["lo"]:
            raise RuntimeError(f"{s['file']}: {p.shape[0]} frames, the index says {s['hi'] - s['lo']}")
    return SegmentedFrames(parts, bounds)


def _remove_segments(path: str, keep=()):
    """Remove `<path>.index.json` and the segment files it names (an earlier multi-rank run's form of the output), except the
    files named in `keep` (base names: the segments the run that calls this has just written)."""
    ip = path + ".index.json"
    if not os.path.exists(ip):
        return
    try:
        with open(ip) as fh:
            idx = json.load(fh)
        for s in idx.get("segments", []):
            if s["file"] in keep:
                continue
            f = os.path.join(os.path.dirname(path), s["file"])
            if os.path.exists(f):
                os.remove(f)
    finally:
        os.remove(ip)


def merge_output(path: str, remove_segments: bool = True) -> str:
    """Concatenate the segments of `<path>.index.json` into the single dump `<path>` (tmp -> final rename)."""
    seg = open_output(path)
    if not isinstance(seg, SegmentedFrames):
        return path
    if isinstance(seg.parts[0], VideoFrames):
        # video segments: the packets are copied as they are (every segment was written with the same size and slice counts)
        from . import video_io
        tmp = path + ".merge_tmp.mkv"
        first = video_io.VideoReader(seg.parts[0].path)
        H, W = first.height, first.width
        sl = (min(4, W), min(4, H))                    # VideoSink's slice grid
        if first.info.slices != sl[0] * sl[1]:
            raise RuntimeError(f"{seg.parts[0].path}: {first.info.slices} slices per frame, not one of this driver's segments")
        first.close()
        with video_io.VideoWriter(tmp, W, H, seg.parts[0].fps, slices=sl) as w:
            for part in seg.parts:
                with video_io.VideoReader(part.path) as r:
                    while True:
                        pkt = r.next_packet()
                        if pkt is None:
                            break
                        w.write_packet(pkt)
        for part in seg.parts:
            part.close()
    else:
        tmp = path + ".merge_tmp.npy"
        out = np.lib.format.open_memmap(tmp, mode="w+", dtype=seg.dtype, shape=seg.shape)
        for p, b0 in zip(seg.parts, seg.bounds[:-1]):
            out[b0:b0 + p.shape[0]] = p
        out.flush()
        del out
    os.replace(tmp, path)
    if remove_segments:
        with open(path + ".index.json") as fh:
            idx = json.load(fh)
        for s in idx["segments"]:
            os.remove(os.path.join(os.path.dirname(path), s["file"]))
        os.remove(path + ".index.json")
    return path


def _usable_cores() -> int:
    """Cores this process may use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def pin_to_gpu_numa_node(device_index: int) -> Optional[int]:
    """Restrict this process (and the I/O threads it starts afterwards) to the CPUs of the NUMA node its GPU hangs off:
    the pinned staging buffers and the page-cache copies of a rank then stay on the memory controller next to its PCIe
    root.  Best effort: returns the node, or None when the topology cannot be read (containers often hide it)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def run(depth_path: str, color_path: Optional[str], *, batch: int = 16, create_sbs_depth_video: bool = False,
        max_frames: int = -1, green_and_black_infill_mask: bool = False, backend: Optional[str] = None,
        normal_infill: bool = False, **clip_kwargs):
    """File-level entry (what `python stereo_rerender.py --depth_video ...` is to the reference).
    Multi-process aware: under torchrun every rank renders its own contiguous frame range into its own output segment
    files (plan_outputs); rank 0 adds the index.  `backend`: torch.distributed backend (default: RCCL when a GPU is
    visible; MDVT_DIST_BACKEND overrides -- the two-ranks-on-one-GPU tests use gloo)."""
    from . import video_io
    rank, world = D.init_process_group(backend or os.environ.get("MDVT_DIST_BACKEND"))
    video = video_io.is_matroska(depth_path)                 # the reference's own format (sr:326-341): outputs follow it
    depth = VideoFrames(depth_path) if video else np.load(depth_path, mmap_mode="r")
    if color_path is None:
        color = depth                                                                                  # sr:508-509
    else:
        color = VideoFrames(color_path) if video_io.is_matroska(color_path) else np.load(color_path, mmap_mode="r")
    if depth.ndim != 4 or depth.shape[3] != 3 or depth.dtype != np.uint8:
        raise ValueError("depth dump must be uint8 [N, H, W, 3]")
    if color.shape != depth.shape:
        raise ValueError(f"Depth video and Color video must have the same dimensions "
                         f"(Depth: {depth.shape[2]}x{depth.shape[1]} vs Color {color.shape[2]}x{color.shape[1]}).")   # sr:387-388
    N, H, W = depth.shape[:3]
    n_total = N
    if max_frames >= 0:
        N = min(N, max_frames)      # the reference processes max_frames+1 and then fails its own check: dropped
    clip = load_clip_parameters(n_total, W, H, n_use=N, **clip_kwargs) if rank == 0 else None
    clip = D.broadcast_clip_parameters(clip, src=0)
    plan = plan_outputs(depth_path, clip, world, create_sbs_depth_video=create_sbs_depth_video,
                        infill_mask=bool(clip_kwargs.get("infill_mask")), normal_infill=normal_infill, ext=".mkv" if video else ".npy")
    final = plan["sbs"]["final"]
    lo, hi = D.frame_range(rank, world, N)
    io_threads = 12
    if world > 1:
        import torch
        if torch.cuda.is_available():
            pin_to_gpu_numa_node(torch.cuda.current_device())
        io_threads = max(2, min(12, _usable_cores() // world))      # the ranks share the box's CPU quota
    # every rank creates, fills, checks and renames its own segment files: no shared inode, no barrier before the loop
    outs = {}
    for k, pl in plan.items():
        t = segment_path(pl["tmp"], rank, world)
        if video:                                   # cv2.VideoWriter(tmp, 'FFV1', frame_rate, out_size) (sr:435-444)
            shp = pl["frame_shape"]
            # (the depth-code frames are B, G, R arrays -- what sr:930-939 hands cv2 --, everything else is RGB)
            outs[k] = VideoSink(t, shp[1], shp[0], depth.fps or 30.0, grey=len(shp) == 2, bgr=k == "depth")
        elif hi > lo:
            outs[k] = np.lib.format.open_memmap(t, mode="w+", dtype=np.uint8, shape=(hi - lo,) + pl["frame_shape"])
        else:                                   # more ranks than frames: an empty segment (cannot be mapped)
            outs[k] = np.empty((0,) + pl["frame_shape"], np.uint8)
            np.save(t, outs[k])
    frames, secs, holes = render_clip(depth, color, outs["sbs"], outs["mask"], clip, lo=lo, hi=hi, batch=batch,
                                      out_depth_rgb=outs.get("depth"), out_infill=outs.get("infill"),
                                      green_and_black=green_and_black_infill_mask, out_base=lo, io_threads=io_threads,
                                      out_infilled=outs.get("infilled"))
    # (no msync: the dumps were written through the page cache, which every later reader shares; forcing 6 GB of dirty pages
    #  to the disk before the rename is what the reference's writers do not do either, and costs seconds on a container fs)
    if video:
        for o in outs.values():
            o.close()                               # out.release() (sr:950)
    del outs
    for k, pl in plan.items():
        verify_and_move(segment_path(pl["tmp"], rank, world), hi - lo, segment_path(pl["final"], rank, world))
    stats = D.gather_rank_stats(frames, secs, holes)          # (a collective: every rank's segments are in place after it)
    if world > 1 and rank == 0:
        for k, pl in plan.items():
            idx = {"frames": N, "world": world, "frame_shape": list(pl["frame_shape"]), "dtype": "uint8",
                   "segments": [{"rank": r, "lo": a, "hi": b, "file": os.path.basename(segment_path(pl["final"], r, world))}
                                for r, a, b in pl["segments"]]}
            # (advisor r04) the segments of an earlier run with ANOTHER world size (`<final>.rankXofY.npy`) are named by the old index
            # only: they go with it, before the new index takes its place -- multi-GB files nobody would find again
            _remove_segments(pl["final"], keep={sg["file"] for sg in idx["segments"]})
            with open(pl["final"] + ".index.json.tmp", "w") as fh:
                json.dump(idx, fh)
            os.replace(pl["final"] + ".index.json.tmp", pl["final"] + ".index.json")
            # a single-file dump of an earlier one-rank run (or merge) of the same clip would shadow these segments in open_output
            if os.path.exists(pl["final"]):
                os.remove(pl["final"])
    elif world == 1:
        # ... and the segments + index of an earlier multi-rank run would outlive this run's single file
        for k, pl in plan.items():
            _remove_segments(pl["final"])
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return stats, final
