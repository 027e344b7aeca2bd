"""MI355X-native stereo-rerender path of calledit/metric_depth_video_toolbox.

Only the hot path of the reference's stereo_rerender.py lives here:

  csrc/                  hand-written gfx950 HIP kernels + the C ABI of include/mdvt.h
  _lib.py                ctypes binding of libmdvt_hip.so (no CPU fallback)
  stereo_rerender.py     host driver: the counterpart of the reference's frame loop
  depth_frames_helper.py device-side 16-bit RGB depth codec (reference names)
  depth_map_tools.py     camera-matrix helpers (reference names)
  distributed.py         frame sharding over ranks + RCCL broadcast of the parameter block
  synthetic.py           deterministic synthetic frames for tests / bench
"""

__all__ = ["_lib", "stereo_rerender", "depth_frames_helper", "depth_map_tools", "synthetic"]
__version__ = "0.1.0"
