"""ctypes binding of libmdvt_hip.so (the C ABI declared in include/mdvt.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is present, the
render entry points raise.  ``load()`` only dlopen()s the library (possible without a GPU);
``Context`` needs a device.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmdvt_hip.so")

MDVT_OK = 0
MODE_POINTS = 0
MODE_MESH = 1

STATUS_NAMES = {0: "MDVT_OK", -1: "MDVT_ERR_INVALID_ARG", -2: "MDVT_ERR_HIP", -3: "MDVT_ERR_UNSUPPORTED",
                -4: "MDVT_ERR_NO_DEVICE", -5: "MDVT_ERR_OOM"}

# every symbol include/mdvt.h declares
SYMBOLS = ("mdvt_version", "mdvt_create", "mdvt_destroy", "mdvt_last_error", "mdvt_set_config",
           "mdvt_render_stereo", "mdvt_render_stereo_batch", "mdvt_decode_depth", "mdvt_encode_depth",
           "mdvt_edge_filter", "mdvt_infill_using_normals", "mdvt_mark_lower_side", "mdvt_touchly_depth",
           "mdvt_equirect_tables", "mdvt_equirect_remap", "mdvt_masked_blur", "mdvt_finish_infill_mask",
           "mdvt_finish_infill_mask_stereo", "mdvt_swap_rb", "mdvt_selftest", "mdvt_normal_infill", "mdvt_infill_using_mask_normals",
           "mdvt_edge_point_pixels", "mdvt_workspace_bytes", "mdvt_release_cached_memory", "mdvt_cached_memory", "mdvt_set_cached_memory_limit",
           "mdvt_debug_read")


class MdvtError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {text}")
        self.code = code


class MdvtConfig(C.Structure):
    _fields_ = [("mode", C.c_int32), ("remove_edges", C.c_int32), ("edge_points", C.c_int32), ("cull", C.c_int32),
                ("ipd_m", C.c_double), ("max_depth", C.c_double), ("key_rgb", C.c_uint8 * 4), ("workspace_mib", C.c_uint32),
                ("subpixel_bits", C.c_int32), ("reserved2", C.c_int32)]


class MdvtFrameParams(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("Krender", C.c_double * 9), ("depth_scale", C.c_double),
                ("convergence_angle", C.c_double), ("T", C.c_double * 16), ("has_T", C.c_int32), ("reserved", C.c_int32)]


class MdvtIO(C.Structure):
    _fields_ = [("depth_rgb", C.c_void_p), ("depth_pitch", C.c_size_t), ("depth_stride", C.c_size_t),
                ("color_rgb", C.c_void_p), ("color_pitch", C.c_size_t), ("color_stride", C.c_size_t),
                ("left_rgb", C.c_void_p), ("right_rgb", C.c_void_p), ("rgb_pitch", C.c_size_t), ("rgb_stride", C.c_size_t),
                ("left_mask", C.c_void_p), ("right_mask", C.c_void_p), ("mask_pitch", C.c_size_t), ("mask_stride", C.c_size_t),
                ("left_depth", C.c_void_p), ("right_depth", C.c_void_p), ("zout_pitch", C.c_size_t), ("zout_stride", C.c_size_t),
                ("left_maskbits", C.c_void_p), ("right_maskbits", C.c_void_p), ("maskbits_pitch", C.c_size_t),
                ("maskbits_stride", C.c_size_t), ("hole_counts", C.c_void_p),
                ("left_seed", C.c_void_p), ("right_seed", C.c_void_p), ("seed_pitch", C.c_size_t), ("seed_stride", C.c_size_t)]


_libs = {}


def lib_path(variant: str = "") -> str:
    return LIB_PATH if not variant else os.path.join(_PKG, f"libmdvt_hip_{variant}.so")


def load():
    """dlopen libmdvt_hip.so and declare prototypes.  Raises if the library has not been built.

    The product library has no tuning / ablation hooks.  With MDVT_LIB_VARIANT=tuning in the environment (read at every
    call, so a test can set it for its own duration) the MDVT_TUNING build of the same objects is loaded instead --
    libmdvt_hip_tuning.so, whose launchers re-read the MDVT_* hooks of csrc/mdvt_internal.h: tools/ and hook-driven tests only."""
    variant = os.environ.get("MDVT_LIB_VARIANT", "")
    if variant in _libs:
        return _libs[variant]
    path = lib_path(variant)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built and there is no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C metric_depth_video_toolbox_amd/csrc`).")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.mdvt_version.restype = C.c_int
    L.mdvt_version.argtypes = []
    L.mdvt_create.restype = C.c_int
    L.mdvt_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_uint32]
    L.mdvt_destroy.restype = C.c_int
    L.mdvt_destroy.argtypes = [vp]
    L.mdvt_last_error.restype = C.c_char_p
    L.mdvt_last_error.argtypes = [vp]
    L.mdvt_set_config.restype = C.c_int
    L.mdvt_set_config.argtypes = [vp, C.POINTER(MdvtConfig)]
    L.mdvt_selftest.restype = C.c_int
    L.mdvt_selftest.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(C.c_uint64)]
    L.mdvt_render_stereo.restype = C.c_int
    L.mdvt_render_stereo.argtypes = [vp, C.POINTER(MdvtFrameParams), C.POINTER(MdvtIO), vp]
    L.mdvt_render_stereo_batch.restype = C.c_int
    L.mdvt_render_stereo_batch.argtypes = [vp, C.c_int, C.POINTER(MdvtFrameParams), C.POINTER(MdvtIO), vp]
    L.mdvt_decode_depth.restype = C.c_int
    L.mdvt_decode_depth.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_double, C.c_double, vp]
    L.mdvt_encode_depth.restype = C.c_int
    L.mdvt_encode_depth.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_double, C.c_int, vp]
    L.mdvt_edge_filter.restype = C.c_int
    L.mdvt_edge_filter.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_double), C.c_double, C.c_int, vp, vp, vp]
    L.mdvt_infill_using_normals.restype = C.c_int
    L.mdvt_infill_using_normals.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp]
    L.mdvt_mark_lower_side.restype = C.c_int
    L.mdvt_mark_lower_side.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp]
    L.mdvt_touchly_depth.restype = C.c_int
    L.mdvt_touchly_depth.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_double, C.c_double, C.c_int, vp]
    L.mdvt_equirect_tables.restype = C.c_int
    L.mdvt_equirect_tables.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mdvt_equirect_remap.restype = C.c_int
    L.mdvt_equirect_remap.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_int, vp, vp, vp]
    L.mdvt_masked_blur.restype = C.c_int
    L.mdvt_masked_blur.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.mdvt_finish_infill_mask.restype = C.c_int
    L.mdvt_finish_infill_mask.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.mdvt_finish_infill_mask_stereo.restype = C.c_int
    L.mdvt_finish_infill_mask_stereo.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.mdvt_swap_rb.restype = C.c_int
    L.mdvt_swap_rb.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_int, vp]
    L.mdvt_normal_infill.restype = C.c_int
    L.mdvt_normal_infill.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_int, vp]
    L.mdvt_infill_using_mask_normals.restype = C.c_int
    L.mdvt_infill_using_mask_normals.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t,
                                                 C.c_int, C.c_int, vp]
    L.mdvt_edge_point_pixels.restype = C.c_int
    L.mdvt_edge_point_pixels.argtypes = [vp, C.POINTER(MdvtFrameParams), vp, C.c_size_t, C.c_int, vp, vp]
    L.mdvt_workspace_bytes.restype = C.c_int
    L.mdvt_workspace_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mdvt_release_cached_memory.restype = C.c_int
    L.mdvt_release_cached_memory.argtypes = [C.c_int]
    L.mdvt_cached_memory.restype = C.c_int
    L.mdvt_cached_memory.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mdvt_set_cached_memory_limit.restype = C.c_int
    L.mdvt_set_cached_memory_limit.argtypes = [C.c_uint64]
    L.mdvt_debug_read.restype = C.c_int
    L.mdvt_debug_read.argtypes = [vp, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    _libs[variant] = L
    return L


def cached_memory(device: int = -1):
    """(idle bytes, idle blocks) of the process-wide workspace pool for GPU `device` (-1: all): mdvt_cached_memory."""
    L = load()
    b, n = C.c_uint64(), C.c_uint64()
    L.mdvt_cached_memory(int(device), C.byref(b), C.byref(n))
    return int(b.value), int(n.value)


def release_cached_memory(device: int = -1) -> None:
    """Return the pool's idle workspace blocks to the driver (mdvt_release_cached_memory)."""
    load().mdvt_release_cached_memory(int(device))


def set_cached_memory_limit(bytes_per_gpu: int) -> None:
    """Idle workspace bytes the process keeps per GPU for the next context (mdvt_set_cached_memory_limit; default 4 GiB)."""
    load().mdvt_set_cached_memory_limit(int(bytes_per_gpu))


def retry_after_release(alloc):
    """Run `alloc()` (a torch allocation); on torch's out-of-memory error hand the library's idle workspace blocks back to the
    driver (torch's allocator cannot see them) and try once more."""
    import torch
    try:
        return alloc()
    except torch.cuda.OutOfMemoryError:
        release_cached_memory(-1)
        torch.cuda.empty_cache()
        return alloc()


def exported_symbols():
    """Names from SYMBOLS that the loaded library actually exports (used by the CPU-only tests)."""
    L = load()
    return [s for s in SYMBOLS if hasattr(L, s)]


class Context:
    """Owns one mdvt_ctx.  ``check`` turns a negative status into MdvtError with the library's text."""

    def __init__(self, device: int, width: int, height: int):
        self._L = load()
        self._h = C.c_void_p()
        rc = self._L.mdvt_create(C.byref(self._h), int(device), int(width), int(height), 0)
        if rc != MDVT_OK:
            raise MdvtError(rc, (self._L.mdvt_last_error(None) or b"").decode())
        self.device, self.W, self.H = int(device), int(width), int(height)

    @property
    def handle(self):
        return self._h

    def check(self, rc: int):
        if rc != MDVT_OK:
            raise MdvtError(rc, (self._L.mdvt_last_error(self._h) or b"").decode())

    def workspace_bytes(self) -> int:
        """Device memory the context owns right now (mdvt_workspace_bytes)."""
        n = C.c_uint64()
        self.check(self._L.mdvt_workspace_bytes(self._h, C.byref(n)))
        return int(n.value)

    def debug_read_queue_block(self):
        """(bytes of the general mesh path's queue block as a uint32 array, info[8]) -- tuning library only (mdvt_debug_read)."""
        import numpy as np
        info = (C.c_uint64 * 8)()
        self.check(self._L.mdvt_debug_read(self._h, 0, None, 0, info))
        buf = np.zeros(int(info[0]) // 4, dtype=np.uint32)
        if buf.size:
            self.check(self._L.mdvt_debug_read(self._h, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes, info))
        return buf, [int(v) for v in info]

    def debug_coherence(self):
        """mdvt_debug_read(what = 1): the cross-XCD coherence test on the queue block (tuning library; overwrites the block).
        -> uint32[80]: [8 w + r] wrong pattern words by writer / reader XCD, [64 + r] wrong atomic sums, [72] total, [73..75] first."""
        import numpy as np
        info = (C.c_uint64 * 8)()
        out = np.zeros(80, dtype=np.uint32)
        self.check(self._L.mdvt_debug_read(self._h, 1, out.ctypes.data_as(C.c_void_p), out.nbytes, info))
        return out

    def debug_pools(self):
        """mdvt_debug_read(what = 2), tuning library: idle blocks of the two process-wide pools as this context's GPU sees them ->
        dict(param_mine, param_other, ws_mine, ws_other, tag)."""
        info = (C.c_uint64 * 8)()
        self.check(self._L.mdvt_debug_read(self._h, 2, None, 0, info))
        return dict(param_mine=int(info[0]), param_other=int(info[1]), ws_mine=int(info[2]), ws_other=int(info[3]), tag=int(info[4]))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.mdvt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
