"""The ctypes stub printed in INTEGRATION.md is executed as written (only the library path is made absolute) and must
give the same frames as the package's own driver -- so the document cannot drift from the ABI."""
import argparse
import os
import re
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_the_package():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
    from metric_depth_video_toolbox_amd.depth_map_tools import compute_camera_matrix
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C.*?)```", md, re.S).group(1)
    code = code.replace('C.CDLL("libmdvt_hip.so")', f'C.CDLL("{REPO}/metric_depth_video_toolbox_amd/libmdvt_hip.so")')
    mod = types.ModuleType("mdvt_hip_stub")
    exec(compile(code, "INTEGRATION.md", "exec"), mod.__dict__)
    W, H = 256, 144
    d, c = synthetic.SyntheticScene(W, H, config_id=2).frame(0)
    T = synthetic.synthetic_pose_track(4)[3]
    for kw, conv, pose in ((dict(render_as_pointcloud=True, infill_mask=False), None, None),
                           (dict(render_as_pointcloud=False, infill_mask=True), 0.013, None),
                           (dict(render_as_pointcloud=False, infill_mask=False), None, T)):
        args = argparse.Namespace(remove_edges=False, do_basic_infill=False, dont_remove_edges=False, dont_place_points_in_edges=False,
                                  pupillary_distance=65, max_depth=100, **kw)
        R = mod.Renderer(W, H, args)
        K = compute_camera_matrix(45.0, None, W, H)
        out, hole = R.render(d, c, K, K, 1.0, conv, pose)
        r = sr.StereoRerenderer(W, H, pupillary_distance=65, **kw)
        p = r.frame_params(xfov=45.0, transformation=pose)
        p.convergence_angle = conv or 0.0
        ref = r.render(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), p)
        assert np.array_equal(out, ref["sbs"].cpu().numpy()) and np.array_equal(hole, ref["mask"].cpu().numpy()), kw
        r.close()


def test_integration_md_normal_infill_snippet_runs_as_written():
    """Section 2c's call sequence, executed verbatim on top of the stub of section 1, gives the package's own result."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from metric_depth_video_toolbox_amd import basic_nomal_infill as bni
    from test_gpu_normal_infill import ni_scene                        # (tests/ is on sys.path: rootdir conftest, prepend mode)
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    stub = re.search(r"```python\n(import ctypes as C.*?)```", md, re.S).group(1)
    stub = stub.replace('C.CDLL("libmdvt_hip.so")', f'C.CDLL("{REPO}/metric_depth_video_toolbox_amd/libmdvt_hip.so")')
    snippet = re.search(r"## 2c\..*?```python\n(.*?)```", md, re.S).group(1)
    W, H, n = 96, 64, 3
    rng = np.random.default_rng(4)
    rgb = np.zeros((n, H, 2 * W, 3), np.uint8); mask = np.zeros_like(rgb)
    for f in range(n):
        for e in range(2):
            rgb[f, :, e * W:(e + 1) * W], mask[f, :, e * W:(e + 1) * W] = ni_scene(rng, W, H)
    env = types.ModuleType("mdvt_hip_stub").__dict__
    exec(compile(stub, "INTEGRATION.md", "exec"), env)
    h = env["C"].c_void_p()
    env["_check"](None, env["_L"].mdvt_create(env["C"].byref(h), torch.cuda.current_device(), W, H, 0))
    env.update(h=h, W=W, H=H, n=n, d_rgb=torch.from_numpy(rgb).cuda(), d_mask=torch.from_numpy(mask).cuda(),
               d_out=torch.zeros((n, H, 2 * W, 3), dtype=torch.uint8, device="cuda"),
               stream=env["C"].c_void_p(torch.cuda.current_stream().cuda_stream))
    exec(compile(snippet, "INTEGRATION.md#2c", "exec"), env)
    torch.cuda.synchronize()
    want = bni.normal_infill_sbs(torch.from_numpy(rgb).cuda(), torch.from_numpy(mask).cuda())
    assert np.array_equal(env["d_out"].cpu().numpy(), want.cpu().numpy())
