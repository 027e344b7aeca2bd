"""GPU tests of the clip-level driver (BASELINE configs C3 / C4 shapes at test sizes + one 4K frame):
streamed batches, per-frame side-car parameters, SBS depth dump, z-buffer contention stressor."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from metric_depth_video_toolbox_amd import clip, stereo_rerender, synthetic
    return clip, stereo_rerender, synthetic


def _oracle_frame(orc, clipp, r, rec, d, c, T):
    K = np.array([rec.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(clipp.W, clipp.H, K, ipd_m=r.pupillary_distance / 1000, max_depth=clipp.max_depth, depth_scale=rec.depth_scale,
                         mode=orc.MODE_POINTS if r.mode == 0 else orc.MODE_MESH, remove_edges=r.remove_edges,
                         edge_points=r.edge_points, conv_angle=rec.convergence_angle, T=T, key_rgb=r.key_rgb)
    return orc.render_stereo(op, d, c, want_depth=True)


@pytest.mark.parametrize("variant", ["plain_points", "mesh_infill_sidecars"])
def test_clip_through_files(mods, orc, tmp_path, variant):
    clip, sr, synthetic = mods
    W, H, N = 192, 108, 23
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=6).clip(N)
    dp, cp = str(tmp_path / "clip_depth.npy"), str(tmp_path / "clip_color.npy")
    np.save(dp, d); np.save(cp, c)
    kw = dict(pupillary_distance=65, render_as_pointcloud=True, xfov=45.0)
    if variant == "mesh_infill_sidecars":
        xf = [44.0 + 0.25 * k for k in range(N)]
        conv = [2.0 + 0.05 * k for k in range(N)]
        conv[0] = conv[5] = conv[6] = float("nan")
        T = synthetic.synthetic_pose_track(N)
        (tmp_path / "xfov.json").write_text(json.dumps(xf))
        (tmp_path / "conv.json").write_text(json.dumps(conv))
        (tmp_path / "T.json").write_text(json.dumps(T.tolist()))
        kw = dict(pupillary_distance=63, xfov_file=str(tmp_path / "xfov.json"), convergence_file=str(tmp_path / "conv.json"),
                  transformation_file=str(tmp_path / "T.json"), transformation_lock_frame=4, infill_mask=True)
    stats, final = clip.run(dp, cp, batch=5, create_sbs_depth_video=True, green_and_black_infill_mask=True, **kw)
    assert final == dp + "_stereo.npy" and os.path.exists(final) and not os.path.exists(dp + "_tmp_stereo.npy")
    assert stats.shape == (1, 3) and stats[0, 0] == N
    sbs, mask, zrgb = np.load(final), np.load(final + "_holemask.npy"), np.load(final + "_depth.npy")
    assert sbs.shape == (N, H, 2 * W, 3) and mask.shape == (N, H, 2 * W) and zrgb.shape == (N, H, 2 * W, 3)
    if kw.get("infill_mask"):
        im = np.load(final + "_infillmask.npy")
        assert np.array_equal(im, (mask > 0)[..., None] * np.array([0, 255, 0], np.uint8))
    else:
        assert not os.path.exists(final + "_infillmask.npy")
    cl = clip.load_clip_parameters(N, W, H, **kw)
    r = clip.renderer_for(cl)
    recs = clip.frame_param_records(r, cl, 0, N)
    for t in range(N):
        T = None if cl.transformations is None else cl.transformations[t]
        want = _oracle_frame(orc, cl, r, recs[t], d[t], c[t], T)
        assert np.array_equal(mask[t][:, :W], want["left_mask"]) and np.array_equal(mask[t][:, W:], want["right_mask"]), t
        assert np.array_equal(sbs[t][:, :W], want["left_rgb"]) and np.array_equal(sbs[t][:, W:], want["right_rgb"]), t
        for sl, key in ((slice(0, W), "left_depth"), (slice(W, 2 * W), "right_depth")):
            assert np.array_equal(zrgb[t][:, sl][..., ::-1], orc.encode_depth(want[key], cl.max_depth)), t   # sr:930-936, B,G,R
    assert int(stats[0, 2]) == int(np.count_nonzero(mask))
    r.close()


def test_4k_pose_driven_with_contention_band(mods, orc):
    """BASELINE config C4's shape: 3840x2160, pose-driven novel view, a band where ~512 sources fold onto
    one or two target pixels (z-buffer atomic contention)."""
    clip, sr, synthetic = mods
    W, H = 3840, 2160
    sc = synthetic.SyntheticScene(W, H, config_id=4)
    from metric_depth_video_toolbox_amd.depth_map_tools import compute_camera_matrix
    K = compute_camera_matrix(45.0, None, W, H)
    z = synthetic.contention_band(sc.depth_m(0), K[0, 0], 0.065, row0=1000, rows=64)
    depth_rgb = synthetic.quantise_depth_to_rgb(z)
    _, color = sc.frame(0)
    T = synthetic.synthetic_pose_track(60)[50]
    for pose in (None, T):
        r = sr.StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=True)
        p = r.frame_params(xfov=45.0, transformation=pose)
        got = r.render(torch.from_numpy(depth_rgb).cuda(), torch.from_numpy(color).cuda(), p, want_depth=True)
        op = orc.make_params(W, H, K, ipd_m=0.065, max_depth=100, depth_scale=p.depth_scale, mode=orc.MODE_POINTS, T=pose)
        want = orc.render_stereo(op, depth_rgb, color, want_depth=True)
        sbs, mask, zz = got["sbs"].cpu().numpy(), got["mask"].cpu().numpy(), got["depth"].cpu().numpy()
        assert np.array_equal(mask[:, :W], want["left_mask"]) and np.array_equal(mask[:, W:], want["right_mask"])
        assert np.array_equal(sbs[:, :W], want["left_rgb"]) and np.array_equal(sbs[:, W:], want["right_rgb"])
        assert np.array_equal(zz[:, :W], want["left_depth"]) and np.array_equal(zz[:, W:], want["right_depth"])
        if pose is None:    # the stressor really folds: in the band, one left-eye column wins over hundreds of sources
            band_holes = (mask[1000:1064, :W] > 0).sum(axis=1)
            assert band_holes.min() > 300
        r.close()


def test_4k_mesh_pose_single_frame(mods, orc):
    clip, sr, synthetic = mods
    W, H = 3840, 2160
    depth_rgb, color = synthetic.SyntheticScene(W, H, config_id=4).frame(0)
    T = synthetic.synthetic_pose_track(60)[25]
    r = sr.StereoRerenderer(W, H, pupillary_distance=65)
    p = r.frame_params(xfov=45.0, transformation=T)
    got = r.render(torch.from_numpy(depth_rgb).cuda(), torch.from_numpy(color).cuda(), p)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(W, H, K, ipd_m=0.065, max_depth=100, depth_scale=p.depth_scale, mode=orc.MODE_MESH, T=T)
    want = orc.render_stereo(op, depth_rgb, color)
    sbs, mask = got["sbs"].cpu().numpy(), got["mask"].cpu().numpy()
    assert np.array_equal(mask[:, :W], want["left_mask"]) and np.array_equal(mask[:, W:], want["right_mask"])
    assert np.array_equal(sbs[:, :W], want["left_rgb"]) and np.array_equal(sbs[:, W:], want["right_rgb"])
    r.close()


def test_cli_end_to_end(mods, orc, tmp_path, capsys):
    """`python -m metric_depth_video_toolbox_amd.stereo_rerender` with the reference's flag names."""
    clip, sr, synthetic = mods
    W, H, N = 160, 90, 7
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=5).clip(N)
    dp, cp = str(tmp_path / "v_depth.npy"), str(tmp_path / "v.npy")
    np.save(dp, d); np.save(cp, c)
    rc = sr.main(["--depth_video", dp, "--color_video", cp, "--xfov", "50", "--pupillary_distance", "65",
                  "--infill_mask", "--max_frames", "5", "--batch", "3"])
    finished = np.load(dp + "_stereo.npy_infillmask.npy")                 # the normal-coloured mask (sr:787-808)
    os.remove(dp + "_stereo.npy")
    rc = sr.main(["--depth_video", dp, "--color_video", cp, "--xfov", "50", "--pupillary_distance", "65",
                  "--infill_mask", "--green_and_black_infill_mask", "--max_frames", "5", "--batch", "3"])
    assert rc == 0 and "Processing complete" in capsys.readouterr().out
    sbs, mask = np.load(dp + "_stereo.npy"), np.load(dp + "_stereo.npy_holemask.npy")
    assert sbs.shape == (5, H, 2 * W, 3) and os.path.exists(dp + "_stereo.npy_infillmask.npy")
    r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)
    p = r.frame_params(xfov=50.0)
    for t in range(5):
        K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
        op = orc.make_params(W, H, K, ipd_m=0.065, max_depth=100, depth_scale=p.depth_scale, mode=orc.MODE_MESH,
                             remove_edges=True, edge_points=True, key_rgb=(0, 255, 0))
        want = orc.render_stereo(op, d[t], c[t], want_seed=True)
        assert np.array_equal(finished[t][:, :W], orc.finish_infill_mask(want["left_seed"])[0])
        assert np.array_equal(finished[t][:, W:], orc.finish_infill_mask(want["right_seed"])[0])
        assert np.array_equal(sbs[t][:, :W], want["left_rgb"]) and np.array_equal(sbs[t][:, W:], want["right_rgb"])
        assert np.array_equal(mask[t][:, :W], want["left_mask"]) and np.array_equal(mask[t][:, W:], want["right_mask"])
    r.close()


def test_cli_end_to_end_on_matroska_files(mods, orc, tmp_path, capsys):
    """The reference's real interface: `--depth_video x.mkv --color_video y.mkv` in, FFV1-in-Matroska out
    under the reference's names -- x.mkv_stereo.mkv, ..._infillmask.mkv, ..._depth.mkv (sr:411-444) -- through the tmp -> final
    protocol (dfh:163-179), every frame equal to the oracle's render of the decoded inputs; the 16-bit depth code survives both
    files (lossless)."""
    from metric_depth_video_toolbox_amd import video_io
    clip, sr, synthetic = mods
    W, H, N = 160, 90, 7
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=5).clip(N)
    dp, cp = str(tmp_path / "v_depth.mkv"), str(tmp_path / "v.mkv")
    for path, frames in ((dp, d), (cp, c)):
        with video_io.VideoWriter(path, W, H, 24000 / 1001, bgr=True) as w:          # as cv2.VideoWriter receives them: BGR
            for f in frames:
                w.write(np.ascontiguousarray(f[..., ::-1]))
    rc = sr.main(["--depth_video", dp, "--color_video", cp, "--xfov", "50", "--pupillary_distance", "65", "--infill_mask",
                  "--create_sbs_depth_video", "--max_frames", "5", "--batch", "3"])
    assert rc == 0 and "Processing complete" in capsys.readouterr().out
    final = dp + "_stereo.mkv"
    for f in (final, final + "_infillmask.mkv", final + "_depth.mkv", final + "_holemask.mkv"):
        assert os.path.exists(f) and video_io.is_matroska(f), f
    assert not [f for f in os.listdir(tmp_path) if "_tmp_" in f]
    with video_io.VideoReader(final) as r:
        assert (r.width, r.height, r.frames) == (2 * W, H, 5) and abs(r.fps - 24000 / 1001) < 1e-3
        sbs = np.stack(list(r))
    mask = np.stack(list(video_io.VideoReader(final + "_holemask.mkv")))
    im = np.stack(list(video_io.VideoReader(final + "_infillmask.mkv")))
    zrgb = np.stack(list(video_io.VideoReader(final + "_depth.mkv")))
    assert np.array_equal(mask[..., 0], mask[..., 1]) and np.array_equal(mask[..., 0], mask[..., 2])
    r = sr.StereoRerenderer(W, H, pupillary_distance=65, infill_mask=True)
    p = r.frame_params(xfov=50.0)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    op = orc.make_params(W, H, K, ipd_m=0.065, max_depth=100, depth_scale=p.depth_scale, mode=orc.MODE_MESH,
                         remove_edges=True, edge_points=True, key_rgb=(0, 255, 0))
    for t in range(5):
        want = orc.render_stereo(op, d[t], c[t], want_depth=True, want_seed=True)
        assert np.array_equal(sbs[t][:, :W], want["left_rgb"]) and np.array_equal(sbs[t][:, W:], want["right_rgb"])
        assert np.array_equal(mask[t][:, :W, 0], want["left_mask"]) and np.array_equal(mask[t][:, W:, 0], want["right_mask"])
        assert np.array_equal(im[t][:, :W], orc.finish_infill_mask(want["left_seed"])[0])
        for eye, sl in (("left", slice(0, W)), ("right", slice(W, 2 * W))):          # sr:930-939: the depth code of both eyes, B, G, R
            assert np.array_equal(zrgb[t][:, sl], orc.encode_depth(want[eye + "_depth"], 100.0))          # (read back as RGB: R = high byte)
    r.close()


def _touchly_np(depth, tmax, tmin, zero_is_far):
    """sr:549-551 / 687-691 literally."""
    d8 = np.rint(np.maximum(0, np.minimum(depth, tmax) - tmin) * (255 / (tmax - tmin))).astype(np.uint8)
    if zero_is_far:
        d8[d8 == 0] = 255
    return np.repeat((255 - d8)[..., np.newaxis], 3, axis=-1)


@pytest.mark.parametrize("posed", [False, True])
def test_touchly1_clip(mods, orc, tmp_path, posed):
    """--touchly1: colour over Touchly depth.  Without a pose file straight from the input (sr:548-552); with one,
    a single unshifted render of the posed mesh (sr:673-691), holes showing the background colour."""
    clip, sr, synthetic = mods
    W, H, N = 160, 96, 5
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=5).clip(N)
    dp, cp = str(tmp_path / "d.npy"), str(tmp_path / "c.npy")
    np.save(dp, d); np.save(cp, c)
    kw = dict(xfov=60.0, touchly1=True, touchly_max_depth=12.0, touchly_min_depth=1.0, infill_mask=posed)
    if posed:
        (tmp_path / "T.json").write_text(json.dumps(synthetic.synthetic_pose_track(N).tolist()))
        kw["transformation_file"] = str(tmp_path / "T.json")
    stats, final = clip.run(dp, cp, batch=2, **kw)
    assert final == dp + "_Touchly1.npy" and not os.path.exists(dp + "_tmp_Touchly1.npy")       # sr:411-413
    assert os.path.exists(final + "_infillmask.npy") == posed
    out = np.load(final)
    assert out.shape == (N, 2 * H, W, 3)
    if not posed:       # the fast path (sr:548-552) writes no infill-mask frames even when one is asked for
        stats, final2 = clip.run(dp, cp, batch=2, **dict(kw, infill_mask=True))
        assert not os.path.exists(final2 + "_infillmask.npy") and np.array_equal(np.load(final2), out)
    cl = clip.load_clip_parameters(N, W, H, **kw)
    r = clip.renderer_for(cl)
    recs = clip.frame_param_records(r, cl, 0, N)
    for t in range(N):
        if not posed:
            z = orc.decode_depth(d[t], cl.max_depth, recs[t].depth_scale)       # dfh:99-103 + sr:541 (scale 60 -> 45 deg)
            assert np.array_equal(out[t, :H], c[t])
            assert np.array_equal(out[t, H:], _touchly_np(z, 12.0, 1.0, False)), t
        else:
            assert r.pupillary_distance == 0 and recs[t].convergence_angle == 0.0
            want = _oracle_frame(orc, cl, r, recs[t], d[t], c[t], cl.transformations[t])
            img = want["left_rgb"].copy()
            img[want["left_mask"] > 0] = (0, 255, 0)                                # raw render: background colour in holes
            assert np.array_equal(out[t, :H], img), t
            assert np.array_equal(out[t, H:], _touchly_np(want["left_depth"], 12.0, 1.0, True)), t
    r.close()


@pytest.mark.parametrize("touchly0", [False, True])
def test_vr180_clip(mods, orc, tmp_path, touchly0):
    """--vr180 / --touchly0 (sr:406-407, 527-535, 825-829, 914-918): 1920x1920 render with the square VR180 camera,
    every image through convert_to_equirectangular; other input sizes are refused."""
    clip, sr, synthetic = mods
    W = H = 1920
    N = 2
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=6).clip(N)
    dp, cp = str(tmp_path / "d.npy"), str(tmp_path / "c.npy")
    np.save(dp, d); np.save(cp, c)
    kw = dict(xfov=90.0, pupillary_distance=65, vr180=not touchly0, touchly0=touchly0, render_as_pointcloud=touchly0)
    (tmp_path / "xfov.json").write_text(json.dumps([90.0, 70.0]))                 # per-frame fov: 90 -> render fov 90, 70 -> 75
    kw.pop("xfov"); kw["xfov_file"] = str(tmp_path / "xfov.json")
    stats, final = clip.run(dp, cp, batch=2, **kw)
    assert final == dp + ("_Touchly0.npy" if touchly0 else "_stereo.npy")                          # sr:414-422
    out = np.load(final)
    assert out.shape == (N, H, (3 if touchly0 else 2) * W, 3)
    cl = clip.load_clip_parameters(N, W, H, **kw)
    assert cl.mode_flags & 16
    r = clip.renderer_for(cl)
    recs = clip.frame_param_records(r, cl, 0, N)
    for t, fov in enumerate((90.0, 75.0)):
        rec = recs[t]
        K = np.array([rec.K[k] for k in range(9)]).reshape(3, 3)
        Kr = np.array([rec.Krender[k] for k in range(9)]).reshape(3, 3)
        assert abs(sr.vr180_render_fov(K) - fov) < 1e-9 and abs(Kr[0, 0] - 960 / np.tan(np.radians(fov / 2))) < 1e-9
        fov = sr.vr180_render_fov(K)        # 89.99999999999999 for the 90-degree frame: fov_from_camera_matrix round trip (sr:529-533)
        op = orc.make_params(W, H, K, Kr=Kr, ipd_m=cl.ipd_m, max_depth=cl.max_depth, depth_scale=rec.depth_scale,
                             mode=orc.MODE_POINTS if r.mode == 0 else orc.MODE_MESH)
        want = orc.render_stereo(op, d[t], c[t], want_depth=True)
        assert np.array_equal(out[t, :, :W], orc.convert_to_equirectangular(want["left_rgb"], fov)), t
        assert np.array_equal(out[t, :, W:2 * W], orc.convert_to_equirectangular(want["right_rgb"], fov)), t
        if touchly0:
            plane = _touchly_np(want["left_depth"], 5.0, 0.0, True)
            assert np.array_equal(out[t, :, 2 * W:], orc.convert_to_equirectangular(plane, fov)), t
    r.close()
    with pytest.raises(ValueError):
        sr.make_frame_params(1920, 1080, 60.0, vr180=True)


@pytest.mark.parametrize("basic", [False, True])
def test_infill_mask_clip(mods, orc, tmp_path, basic):
    """--infill_mask through files: <out>_infillmask.npy holds the finished normal-coloured mask of both eyes
    (sr:787-808, 921-928); with --do_basic_infill the stereo frames have their holes filled along it (sr:809-812)."""
    clip, sr, synthetic = mods
    W, H, N = 192, 108, 5
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=6).clip(N)
    dp, cp = str(tmp_path / "d.npy"), str(tmp_path / "c.npy")
    np.save(dp, d); np.save(cp, c)
    conv = [2.0, float("nan"), 2.2, 2.3, 0.0]
    (tmp_path / "conv.json").write_text(json.dumps(conv))
    kw = dict(xfov=45.0, pupillary_distance=65, infill_mask=True, do_basic_infill=basic, convergence_file=str(tmp_path / "conv.json"))
    stats, final = clip.run(dp, cp, batch=2, **kw)
    sbs, mask, im = np.load(final), np.load(final + "_holemask.npy"), np.load(final + "_infillmask.npy")
    assert im.shape == (N, H, 2 * W, 3) and not os.path.exists(final + "_infillmask_seed.npy")
    cl = clip.load_clip_parameters(N, W, H, **kw)
    r = clip.renderer_for(cl)
    assert r.do_basic_infill == basic and r.remove_edges and r.edge_points
    recs = clip.frame_param_records(r, cl, 0, N)
    for t in range(N):
        K = np.array([recs[t].K[k] for k in range(9)]).reshape(3, 3)
        op = orc.make_params(W, H, K, ipd_m=0.065, depth_scale=recs[t].depth_scale, mode=orc.MODE_MESH, remove_edges=True,
                             edge_points=2 if basic else 1, conv_angle=recs[t].convergence_angle, key_rgb=(0, 255, 0))
        want = orc.render_stereo(op, d[t], c[t], want_seed=True)
        for eye, sl in (("left", slice(0, W)), ("right", slice(W, 2 * W))):
            wfin, wrem = orc.finish_infill_mask(want[eye + "_seed"])
            assert wrem == 0 and np.array_equal(im[t][:, sl], wfin), (t, eye)
            assert np.array_equal(mask[t][:, sl], want[eye + "_mask"])
            wimg = want[eye + "_rgb"]
            if basic:
                wn = ((wfin.astype(np.float32) / np.float32(255.0)) * 2 - 1).astype(np.float32)
                wimg = orc.infill_using_normals(wimg, want[eye + "_mask"] > 0, wn)
            assert np.array_equal(sbs[t][:, sl], wimg), (t, eye)
    r.close()


def test_normal_infill_in_the_same_run(mods, orc, tmp_path, capsys):
    """--normal_infill (not a reference flag): movie_2_3D.py's next step, basic_nomal_infill.normal_infill of both eyes, run while
    frame and finished mask are on the device.  Its file = what process_pair makes of the run's two other outputs, and both =
    the oracle's normal_infill of the oracle's frames and masks."""
    clip, sr, synthetic = mods
    from metric_depth_video_toolbox_amd import basic_nomal_infill as bni
    W, H, N = 192, 108, 5
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=6).clip(N)
    dp, cp = str(tmp_path / "d.npy"), str(tmp_path / "c.npy")
    np.save(dp, d); np.save(cp, c)
    (tmp_path / "conv.json").write_text(json.dumps([2.0, float("nan"), 2.2, 2.3, 0.0]))
    rc = sr.main(["--depth_video", dp, "--color_video", cp, "--xfov", "45", "--pupillary_distance", "65", "--infill_mask", "--normal_infill",
                  "--convergence_file", str(tmp_path / "conv.json"), "--batch", "2"])
    assert rc == 0
    final = dp + "_stereo.npy"
    filled = np.load(final + "_infilled.npy")
    sbs, im = np.load(final), np.load(final + "_infillmask.npy")
    assert filled.shape == sbs.shape and not os.path.exists(dp + "_tmp_stereo.npy_infilled.npy")
    os.rename(final + "_infilled.npy", str(tmp_path / "fused.npy"))
    again = np.load(bni.process_pair(final, final + "_infillmask.npy", batch=3))
    assert np.array_equal(again, filled)
    for t in range(N):
        for sl in (slice(0, W), slice(W, 2 * W)):
            assert np.array_equal(filled[t][:, sl], orc.normal_infill(np.ascontiguousarray(sbs[t][:, sl]), np.ascontiguousarray(im[t][:, sl]))), t
    holes = np.all(sbs == 0, -1)
    assert (filled[holes].max(-1) > 0).mean() > 0.8
    with pytest.raises(ValueError):
        clip.run(dp, cp, xfov=45.0, normal_infill=True)                       # no --infill_mask: no normals to march along


def test_max_frames_with_full_length_side_cars_and_unremoved_edges(mods, orc, tmp_path):
    """(a) --max_frames with side-cars that cover the whole clip: the reference checks the xfov list against the video's
    frame count (sr:403) and indexes the convergence list by frame, so this combination works upstream and must here.
    (b) --infill_mask --dont_remove_edges: only remove_edges is cleared (sr:572-573) -- the background stays green and
    the infill-mask output is still written (sr:555-558, 921-928), with nothing but the key colour in it."""
    clip, sr, synthetic = mods
    W, H, N = 128, 72, 6
    d, c = synthetic.SyntheticScene(W, H, config_id=3, n_fg=5).clip(N)
    dp, cp = str(tmp_path / "d.npy"), str(tmp_path / "c.npy")
    np.save(dp, d); np.save(cp, c)
    (tmp_path / "xfov.json").write_text(json.dumps([44.0 + k for k in range(N)]))
    (tmp_path / "conv.json").write_text(json.dumps([2.0, float("nan"), 2.2, 2.4, 2.1, 2.0]))
    stats, final = clip.run(dp, cp, batch=4, max_frames=3, xfov_file=str(tmp_path / "xfov.json"),
                            convergence_file=str(tmp_path / "conv.json"), pupillary_distance=65, infill_mask=True,
                            dont_remove_edges=True)
    sbs, mask, im = np.load(final), np.load(final + "_holemask.npy"), np.load(final + "_infillmask.npy")
    assert sbs.shape[0] == 3 and im.shape == sbs.shape
    cl = clip.load_clip_parameters(N, W, H, n_use=3, xfov_file=str(tmp_path / "xfov.json"), convergence_file=str(tmp_path / "conv.json"),
                                   pupillary_distance=65, infill_mask=True, dont_remove_edges=True)
    full = clip.load_clip_parameters(N, W, H, xfov_file=str(tmp_path / "xfov.json"), convergence_file=str(tmp_path / "conv.json"),
                                     pupillary_distance=65, infill_mask=True, dont_remove_edges=True)
    assert cl.n_frames == 3 and np.array_equal(cl.convergence, full.convergence[:3]) and not (cl.mode_flags & 2) and (cl.mode_flags & 8)
    r = clip.renderer_for(cl)
    assert r.key_rgb == (0, 255, 0) and not r.remove_edges
    recs = clip.frame_param_records(r, cl, 0, 3)
    for t in range(3):
        want = _oracle_frame(orc, cl, r, recs[t], d[t], c[t], None)
        assert np.array_equal(mask[t, :, :W], want["left_mask"]) and np.array_equal(sbs[t, :, W:], want["right_rgb"]), t
        green = np.zeros((H, 2 * W, 3), np.uint8)
        green[mask[t] > 0] = (0, 255, 0)
        assert np.array_equal(im[t], green), t
    r.close()
