"""What bench.py times, parity-tested at the size it times it ("what's weak" 2): the launch sets of the posed / converged
paths -- two banks of workspace slots on two streams, event-ordered reuse -- only exist for calls of more than one launch set, and
their segment counts, queue sizes and slot strides all depend on W, H and the number of frames.  The small-frame bank tests
(test_gpu_render.py::test_two_banks_on_other_shapes, ::test_converged_mesh_launch_sets_on_two_banks) do not stand in for these.

  * 1920 x 1080 x 32 frames in ONE call: product default (mesh + --infill_mask + convergence, movie_2_3D.py:433-445, with seed
    images) and mesh + convergence -- `extra.product_default` / `extra.mesh_convergence` of the bench line;
  * 3840 x 2160 x 8 frames in one call, the pose track and the contention band of BASELINE configs[3] (synthetic.c4_clip, the
    very clip bench.py's `extra.c4_4k_pose_points` / `c4_4k_pose_mesh` render), points and mesh.
Every call is made twice (the z-key slots alternate between two parities), every frame of both must equal the same frame rendered
alone (one launch set, no banks, another context), and a handful of frames are held to the oracle bit for bit.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from test_gpu_render import _K, _compare, mods  # noqa: E402,F401  (the module fixture and the comparison of the render tests)


def _oracle_frame(orc, r, p, d, c, T=None, seed=False):
    op = orc.make_params(r.W, r.H, _K(p), ipd_m=r.pupillary_distance / 1000, max_depth=r.max_depth, depth_scale=p.depth_scale,
                         mode=orc.MODE_POINTS if r.mode == 0 else orc.MODE_MESH, remove_edges=r.remove_edges,
                         edge_points=int(r.edge_points), conv_angle=p.convergence_angle, T=T, key_rgb=r.key_rgb)
    return orc.render_stereo(op, d, c, want_depth=True, want_seed=seed)


def _check_batch(sr, orc, W, H, d, c, make_renderer, make_params, Ts, oracle_frames, seed, tag):
    N = d.shape[0]
    dt, ct = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
    r = make_renderer()
    ps = make_params(r)
    keys = ("sbs", "mask", "depth") + (("seed",) if seed else ())
    runs = []
    for rep in range(2):                                    # both key parities; the second call also re-uses every slot and list
        got = r.render(dt, ct, ps, want_depth=True, want_seed=seed)
        torch.cuda.synchronize()
        runs.append({k: got[k].clone() for k in keys})
    assert r.ctx.workspace_bytes() > 0
    for k in keys:
        assert torch.equal(runs[0][k], runs[1][k]), f"{tag}: {k} differs between the first and the second call"
    # every frame alone: one launch set, no banks, its own context
    r1 = make_renderer()
    for f in range(N):
        one = r1.render(dt[f], ct[f], ps[f], want_depth=True, want_seed=seed)
        for k in keys:
            if not torch.equal(one[k], runs[0][k][f]):
                ndiff = int((one[k] != runs[0][k][f]).sum())
                raise AssertionError(f"{tag}: frame {f} of the {N}-frame call differs from the frame rendered alone: {k} at {ndiff} elements")
    r1.close()
    t0 = time.time()
    for f in oracle_frames:
        want = _oracle_frame(orc, r, ps[f], d[f], c[f], T=None if Ts is None else Ts[f], seed=seed)
        _compare({k: runs[0][k][f] for k in ("sbs", "mask", "depth")}, want, W, f"{tag} frame {f}")
        if seed:
            sd = runs[0]["seed"][f].cpu().numpy()
            for eye, sl in (("left", slice(0, W)), ("right", slice(W, 2 * W))):
                assert np.array_equal(sd[:, sl], want[eye + "_seed"]), f"{tag} seed {eye} frame {f}"
    print(f"{tag}: oracle {len(oracle_frames)} frames in {time.time() - t0:.1f} s")
    r.close()


@pytest.mark.parametrize("variant", ["product_default", "mesh_convergence"])
def test_bench_launch_sets_at_full_hd(mods, orc, variant):
    """bench.py `extra.product_default` / `extra.mesh_convergence`: 32 frames of 1920 x 1080 in one call = four launch sets of 8 on
    two banks and two streams (mdvt_render_stereo_batch)."""
    _lib, sr, synthetic = mods
    W, H, N = 1920, 1080, 32
    d, c = synthetic.SyntheticScene(W, H, config_id=2).clip(N, t0=3)
    kw = dict(infill_mask=True) if variant == "product_default" else {}
    _check_batch(sr, orc, W, H, d, c,
                 lambda: sr.StereoRerenderer(W, H, pupillary_distance=65, **kw),
                 lambda r: [r.frame_params(xfov=45.0, convergence_distance=2.5) for _ in range(N)],
                 None, (0, 7, 8, 15, 16, 31), variant == "product_default", f"{variant} 1080p x {N}")


@pytest.mark.parametrize("mode,N", [("points", 8), ("mesh", 8), ("mesh", 7)])
def test_bench_c4_launch_sets_at_4k(mods, orc, mode, N):
    """bench.py `extra.c4_4k_pose_points` / `c4_4k_pose_mesh`: 8 frames of 3840 x 2160 under the pose track with the contention band
    (in mesh mode: the queued and the huge triangles) in one call -- banks of 2 (mesh: 4 slots by the 4 GiB budget; points: 4 slots).
    7 frames: the odd split of a run that fits one launch set into two banks (halves of 3 frames of 4K and more are split: sets of
    3, 3 and 1)."""
    _lib, sr, synthetic = mods
    W, H = 3840, 2160
    d, c, Ts = synthetic.c4_clip(N, W, H)
    _check_batch(sr, orc, W, H, d, c,
                 lambda: sr.StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=(mode == "points")),
                 lambda r: [r.frame_params(xfov=45.0, transformation=Ts[k]) for k in range(N)],
                 Ts, (0, 3, N - 1), False, f"C4 {mode} 4K x {N}")
