"""FFV1-in-Matroska files, the reference's on-disk format (cv2.VideoCapture / cv2.VideoWriter with fourcc 'FFV1',
stereo_rerender.py:326-341, 426-444, 941; depth_frames_helper.py:125-161): ctypes binding of libmdvt_video.so
(include/mdvt_video.h, csrc_host/mdvt_video.cpp -- host C++, written from RFC 9043 / RFC 9559; no FFmpeg in the image).

    with VideoReader("x_depth.mkv") as r:            # frames come back as H x W x 3 uint8, RGB unless bgr=True
        for frame in r: ...
    with VideoWriter("out.mkv", W, H, fps) as w:
        w.write(rgb)

No fallback: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from fractions import Fraction

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmdvt_video.so")

SYMBOLS = ("mdvt_video_last_error", "mdvt_video_abi", "mdvt_video_open", "mdvt_video_read", "mdvt_video_rewind", "mdvt_video_seek",
           "mdvt_video_next_packet", "mdvt_video_config_record", "mdvt_video_close", "mdvt_video_create", "mdvt_video_write",
           "mdvt_video_write_packet", "mdvt_video_finish", "mdvt_ffv1_encode_frame")

RGB, BGR = 0, 1


class VideoInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("frames", C.c_int64), ("fps", C.c_double),
                ("ffv1_version", C.c_int32), ("ffv1_micro_version", C.c_int32), ("coder_type", C.c_int32), ("slices", C.c_int32),
                ("alpha", C.c_int32), ("intra", C.c_int32), ("ec", C.c_int32), ("reserved", C.c_int32)]


class VideoError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or `make -C metric_depth_video_toolbox_amd/csrc_host`); there is no fallback codec")
        L = C.CDLL(LIB_PATH)
        L.mdvt_video_last_error.restype = C.c_char_p
        L.mdvt_video_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(VideoInfo)]
        L.mdvt_video_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.mdvt_video_rewind.argtypes = [C.c_void_p]
        L.mdvt_video_seek.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.mdvt_video_next_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mdvt_video_config_record.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mdvt_video_write_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.mdvt_video_close.argtypes = [C.c_void_p]
        L.mdvt_video_close.restype = None
        L.mdvt_video_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mdvt_video_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.mdvt_video_finish.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.mdvt_ffv1_encode_frame.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                             C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise VideoError(load().mdvt_video_last_error().decode(errors="replace"))
    return rc


def is_matroska(path: str) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(4) == b"\x1a\x45\xdf\xa3"
    except OSError:
        return False


class VideoReader:
    """cv2.VideoCapture for FFV1-in-Matroska: sequential reads (inter-coded streams keep their context state from frame to frame)."""

    def __init__(self, path: str, bgr: bool = False, threads: int = 0):
        self._h = C.c_void_p()
        self.info = VideoInfo()
        self.path, self.order, self.threads = path, BGR if bgr else RGB, int(threads)
        _check(load().mdvt_video_open(os.fsencode(path), C.byref(self._h), C.byref(self.info)))
        self.width, self.height, self.frames, self.fps = self.info.width, self.info.height, int(self.info.frames), float(self.info.fps)

    def read_into(self, out: np.ndarray) -> bool:
        """Decodes the next frame into `out` (H x W x 3 uint8, rows contiguous; e.g. a pinned staging buffer).  False at the end."""
        assert out.dtype == np.uint8 and out.shape == (self.height, self.width, 3) and out.strides[2] == 1 and out.strides[1] == 3
        return _check(load().mdvt_video_read(self._h, out.ctypes.data, out.strides[0], self.order, self.threads)) == 0

    def read(self):
        out = np.empty((self.height, self.width, 3), np.uint8)
        return out if self.read_into(out) else None

    def rewind(self):
        _check(load().mdvt_video_rewind(self._h))

    def seek(self, frame: int):
        """The next read returns `frame` (an inter-coded stream is decoded forward from its last key frame)."""
        _check(load().mdvt_video_seek(self._h, int(frame), self.threads))

    def next_packet(self):
        """The next frame's FFV1 packet as stored (bytes), or None at the end."""
        cap = self.width * self.height * 8 + (1 << 16)
        buf, n = np.empty(cap, np.uint8), C.c_size_t()
        return buf[:n.value].tobytes() if _check(load().mdvt_video_next_packet(self._h, buf.ctypes.data, cap, C.byref(n))) == 0 else None

    def config_record(self) -> bytes:
        buf, n = np.empty(1 << 16, np.uint8), C.c_size_t()
        _check(load().mdvt_video_config_record(self._h, buf.ctypes.data, buf.size, C.byref(n)))
        return buf[:n.value].tobytes()

    def __iter__(self):
        while True:
            f = self.read()
            if f is None:
                return
            yield f

    def close(self):
        if self._h:
            load().mdvt_video_close(self._h)
            self._h = C.c_void_p()

    __enter__ = lambda self: self
    __exit__ = lambda self, *a: self.close()
    __del__ = close


class VideoWriter:
    """cv2.VideoWriter(path, fourcc('F','F','V','1'), fps, (W, H)): FFV1 version 3, intra-only, in Matroska."""

    def __init__(self, path: str, width: int, height: int, fps: float, slices=(0, 0), bgr: bool = False, threads: int = 0):
        fr = Fraction(float(fps)).limit_denominator(1001)
        self._h = C.c_void_p()
        self.path, self.width, self.height, self.order, self.threads = path, int(width), int(height), BGR if bgr else RGB, int(threads)
        self.slices = (int(slices[0]) or min(4, self.width), int(slices[1]) or min(4, self.height))
        _check(load().mdvt_video_create(os.fsencode(path), self.width, self.height, fr.numerator, fr.denominator, int(slices[0]), int(slices[1]),
                                        C.byref(self._h)))
        self.frames = 0

    def write(self, frame: np.ndarray):
        assert frame.dtype == np.uint8 and frame.shape == (self.height, self.width, 3) and frame.strides[2] == 1 and frame.strides[1] == 3
        _check(load().mdvt_video_write(self._h, frame.ctypes.data, frame.strides[0], self.order, self.threads))
        self.frames += 1

    def write_packet(self, packet: bytes):
        """Appends a frame already encoded by encode_frame() with this writer's size and slice counts."""
        _check(load().mdvt_video_write_packet(self._h, packet, len(packet)))
        self.frames += 1

    def close(self) -> int:
        if not self._h:
            return self.frames
        n = C.c_int64()
        h, self._h = self._h, C.c_void_p()
        _check(load().mdvt_video_finish(h, C.byref(n)))
        return int(n.value)

    release = close
    __enter__ = lambda self: self

    def __exit__(self, *a):
        self.close()


def encode_frame(frame: np.ndarray, slices=(2, 2), bgr: bool = False, threads: int = 1):
    """One frame -> (FFV1 packet bytes, configuration record bytes): the codec without the container (tests)."""
    H, W = frame.shape[:2]
    frame = np.ascontiguousarray(frame, np.uint8)
    cap = W * H * 6 + 4096 * slices[0] * slices[1]
    pkt, cfg = np.empty(cap, np.uint8), np.empty(4096, np.uint8)
    ps, cs = C.c_size_t(), C.c_size_t()
    _check(load().mdvt_ffv1_encode_frame(W, H, slices[0], slices[1], frame.ctypes.data, frame.strides[0], BGR if bgr else RGB, threads,
                                         pkt.ctypes.data, cap, C.byref(ps), cfg.ctypes.data, 4096, C.byref(cs)))
    return pkt[:ps.value].tobytes(), cfg[:cs.value].tobytes()
