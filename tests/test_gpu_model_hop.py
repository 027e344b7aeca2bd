"""BASELINE config C5's plumbing: Depth-Anything-V2 (transformers class, seeded random weights -- no
checkpoint is downloadable here) -> on-device 16-bit quantisation -> HIP reproject kernels, one stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("W,H,N", [(448, 252, 2), (1920, 1080, 1)])
def test_depth_model_feeds_the_render_kernels_on_one_stream(orc, W, H, N):
    """(1920 x 1080 is BASELINE config C5's own size.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    pytest.importorskip("transformers")
    from metric_depth_video_toolbox_amd import model_hop, stereo_rerender as sr, synthetic
    _, color = synthetic.SyntheticScene(W, H, config_id=5, n_fg=6).clip(N)
    color_t = torch.from_numpy(color).cuda()
    model = model_hop.build_depth_anything_v2("vits", max_depth=20, seed=7, num_layers=4)
    r = sr.StereoRerenderer(W, H, pupillary_distance=65, max_depth=20, render_as_pointcloud=True)
    p = r.frame_params(xfov=45.0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                       # everything is enqueued on ONE (non-default) stream
        res, depth_rgb = model_hop.color_to_stereo(model, color_t, r, p, input_height=252, want_depth=True)
    side.synchronize()
    code = depth_rgb.cpu().numpy()
    assert code.shape == (N, H, W, 3) and np.array_equal(code[..., 0], code[..., 1])      # R == G (dfh:53-54)
    # the quantiser is the reference codec: decode(encode(model depth)) is within one 16-bit LSB, one-sided
    # (a second inference of the same frames: the model's convolutions are not bit-reproducible from run to run at every
    #  size, so a 16-bit code may differ by one step in a few pixels -- the codec check below is on the depth itself)
    depth = model_hop.infer_depth(model, color_t, 252).cpu().numpy()
    want = np.stack([orc.encode_depth(depth[k], 20) for k in range(N)])
    c16 = code[..., 0].astype(np.int32) * 256 + code[..., 2]
    w16 = want[..., 0].astype(np.int32) * 256 + want[..., 2]
    assert np.abs(c16 - w16).max() <= 1 and (c16 != w16).mean() < 1e-3
    back = np.stack([orc.decode_depth(code[k], 20) for k in range(N)])
    err = depth.astype(np.float64) - back
    lsb = 20 * 65536 / 255 ** 4
    assert err.min() > -lsb - 1e-6 and err.max() < 2 * lsb + 1e-6
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(W, H, K, ipd_m=0.065, max_depth=20, depth_scale=p.depth_scale, mode=orc.MODE_POINTS)
    for k in range(N):
        want = orc.render_stereo(op, code[k], color[k], want_depth=True)
        sbs, mask = res["sbs"][k].cpu().numpy(), res["mask"][k].cpu().numpy()
        assert np.array_equal(mask[:, :W], want["left_mask"]) and np.array_equal(mask[:, W:], want["right_mask"])
        assert np.array_equal(sbs[:, :W], want["left_rgb"]) and np.array_equal(sbs[:, W:], want["right_rgb"])
    r.close()


def test_depth_to_rgb_code_on_batches_taller_than_one_context(orc):
    """40 frames of 1080 rows exceed the 32767-row limit of one context: the batch is encoded in groups, every frame
    still equals the reference encoder (dfh:5-11, 48-61)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from metric_depth_video_toolbox_amd import model_hop
    rng = np.random.default_rng(9)
    d = rng.uniform(0, 25, (40, 1080, 64)).astype(np.float32)
    got = model_hop.depth_to_rgb_code(torch.from_numpy(d).cuda(), 20).cpu().numpy()
    for k in (0, 29, 30, 39):
        assert np.array_equal(got[k], orc.encode_depth(d[k], 20))
