"""CPU oracle for the stereo-rerender hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / the timed CPU baseline.  The product package
(``metric_depth_video_toolbox_amd``) never imports it and has no CPU fallback.

  oracle.c_oracle   ctypes binding of oracle/libmdvt_oracle.so (mdvt_oracle.c, the arbiter)
  oracle.oracle_np  NumPy restatement of the stages the reference's own NumPy functions pin
"""
