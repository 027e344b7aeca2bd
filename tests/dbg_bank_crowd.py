"""One process of the bank-resource crowd (tests/test_gpu_fresh_context.py::test_bank_resources_crowd): CALLS multi-frame calls that
take the two-bank path (a run of general frames longer than one launch set: the side stream and the four events of
csrc/mdvt_api.hip `bank_res_take`), each on a context created for the call and destroyed after it -- the pattern of the r05 soak's
multi-frame sweeps, in which three processes in ~3 000 jobs died inside the HSA runtime's callback thread before the streams and
events were pooled.  A native backtrace is armed (tools/probe/segv_trace.c).  Every call's output must equal the first call's (which
the parent checks against the oracle).  Debug helper, not collected by pytest.
  CALLS=300 python tests/dbg_bank_crowd.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic

so = f"/tmp/libsegv_trace_{os.getpid()}.so"
try:
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(os.getcwd(), "tools", "probe", "segv_trace.c")])
    ctypes.CDLL(so).segv_trace_install(os.path.join(os.environ.get("MDVT_SEGV_TRACE_DIR", "/tmp"), f"bank_crowd_{os.getpid()}.log").encode())
except Exception as e:      # no compiler: run unarmed
    print("no backtrace hook:", e)

CALLS = int(os.environ.get("CALLS", "300"))
W, H, N = 96, 54, 11
d, c = synthetic.SyntheticScene(W, H, seed=31, n_fg=5).clip(N)
dd, cc = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
first = {}
bad = 0
for k in range(CALLS):
    mesh = k % 2 == 1                                     # points: launch sets of 4 -> banks of 2; mesh (small budget): sets of 2 -> banks of 1
    r = sr.StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=not mesh, workspace_mib=1 if mesh else 0)
    p = [r.frame_params(xfov=45.0, transformation=T) for T in synthetic.synthetic_pose_track(N)] if mesh else \
        [r.frame_params(xfov=45.0, convergence_distance=2.0 + 0.1 * t) for t in range(N)]
    got = r.render(dd, cc, p)
    torch.cuda.synchronize()
    out = (got["sbs"].cpu().numpy(), got["mask"].cpu().numpy())
    if mesh not in first:
        first[mesh] = out
        np.savez(os.path.join(os.environ.get("MDVT_SEGV_TRACE_DIR", "/tmp"), f"bank_crowd_first_{int(mesh)}_{os.getpid()}.npz"), sbs=out[0], mask=out[1])
    elif not (np.array_equal(out[0], first[mesh][0]) and np.array_equal(out[1], first[mesh][1])):
        bad += 1
    r.close()
print(f"bank crowd calls {CALLS} bad {bad}")
