"""world_size-2 gloo tests of the N>1 path: one broadcast of the clip parameter block from rank 0,
contiguous frame ranges, all-gather of the per-rank report, max-over-ranks timing."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from metric_depth_video_toolbox_amd import distributed as D
    from metric_depth_video_toolbox_amd.synthetic import synthetic_pose_track
    r, w = D.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    clip = None
    if rank == 0:       # only rank 0 knows the clip (xfov file, smoothed convergence, poses)
        clip = D.ClipParameters(1920, 1080, 301, 0.065, 100.0, 45.0, 5, np.linspace(40, 50, 301),
                                np.linspace(2, 4, 301), synthetic_pose_track(301))
    got = D.broadcast_clip_parameters(clip, src=0)
    lo, hi = D.frame_range(rank, world, got.n_frames)
    stats = D.gather_rank_stats(hi - lo, 0.5 + rank, 1000.0 * rank)
    slow = D.max_over_ranks(0.5 + rank)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), blk=got.pack(), lo=lo, hi=hi, stats=stats, slow=slow)
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding(tmp_path):
    import torch.multiprocessing as mp
    from metric_depth_video_toolbox_amd import distributed as D
    from metric_depth_video_toolbox_amd.synthetic import synthetic_pose_track
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = D.ClipParameters(1920, 1080, 301, 0.065, 100.0, 45.0, 5, np.linspace(40, 50, 301),
                            np.linspace(2, 4, 301), synthetic_pose_track(301)).pack()
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in range(world):
        assert np.array_equal(res[r]["blk"], want), "every rank holds rank 0's parameter block bit for bit"
        assert float(res[r]["slow"]) == 1.5
        assert np.array_equal(res[r]["stats"], np.array([[150, 0.5, 0.0], [151, 1.5, 1000.0]]))
    assert (int(res[0]["lo"]), int(res[0]["hi"]), int(res[1]["lo"]), int(res[1]["hi"])) == (0, 150, 150, 301)


def test_single_process_is_identity():
    from metric_depth_video_toolbox_amd import distributed as D
    c = D.ClipParameters(64, 48, 3, 0.063, 100.0, 45.0, 1, np.full(3, 45.0), np.zeros(3))
    assert D.broadcast_clip_parameters(c) is c
    assert D.max_over_ranks(2.5) == 2.5
    assert D.gather_rank_stats(3, 1.0, 7).shape == (1, 3)


# ---------------------------------------------------------------------------------------------------------------------
# clip.run with more than one rank: per-rank output segments ("each rank owns its output segment")
# ---------------------------------------------------------------------------------------------------------------------
def _fake_render_clip(depth_frames, color_frames, out_sbs, out_mask, clip, *, lo=0, hi=None, batch=16, out_depth_rgb=None,
                      out_infill=None, green_and_black=False, device=None, out_base=0, io_threads=12, out_infilled=None):
    """Stand-in for the GPU stage (CPU tests cannot render): every output frame is a pure function of its input frame, and
    goes through the driver's own file writer (_RawFrames: pwrite at the segment's offset)."""
    from metric_depth_video_toolbox_amd.clip import _RawFrames
    W = clip.W
    writers = {"sbs": _RawFrames(out_sbs, True), "mask": _RawFrames(out_mask, True)}
    if out_depth_rgb is not None:
        writers["depth"] = _RawFrames(out_depth_rgb, True)
    if out_infill is not None:
        writers["infill"] = _RawFrames(out_infill, True)
    for a in range(lo, hi, batch):
        n = min(batch, hi - a)
        d, c = np.asarray(depth_frames[a:a + n]), np.asarray(color_frames[a:a + n])
        sbs = np.concatenate([c, d], axis=2)
        writers["sbs"].write_from(np.ascontiguousarray(sbs), a - out_base, n)
        writers["mask"].write_from(np.ascontiguousarray(sbs[..., 0] ^ 0x5A), a - out_base, n)
        if "depth" in writers:
            writers["depth"].write_from(np.ascontiguousarray(sbs[..., ::-1]), a - out_base, n)
        if "infill" in writers:
            writers["infill"].write_from(np.ascontiguousarray(255 - sbs), a - out_base, n)
    for w in writers.values():
        w.close()
    return hi - lo, 0.25 + 0.01 * lo, float(hi - lo) * 7.0


def _clip_worker(rank, world, port, work_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MDVT_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from metric_depth_video_toolbox_amd import clip
    clip.render_clip = _fake_render_clip
    stats, final = clip.run(os.path.join(work_dir, "d.npy"), os.path.join(work_dir, "c.npy"), batch=3,
                            create_sbs_depth_video=True, xfov=45.0, infill_mask=True,
                            convergence_file=os.path.join(work_dir, "conv.json") if rank == 0 else "only-rank-0-reads-the-side-cars")
    np.save(os.path.join(work_dir, f"stats{rank}.npy"), stats)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 11), (3, 2)])
def test_multi_rank_run_writes_per_rank_segments(tmp_path, world, N):
    """clip.run under `world` ranks (gloo): every rank creates, fills, verifies and renames the segment files of its own
    frame range, rank 0 writes the index; read back through open_output / merge_output the outputs equal the 1-rank run
    byte for byte.  (N < world: a rank with no frames writes an empty segment.)"""
    import json
    import torch.multiprocessing as mp
    from metric_depth_video_toolbox_amd import clip
    rng = np.random.default_rng(5)
    H, W = 6, 8
    d, c = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8), rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    np.save(tmp_path / "d.npy", d); np.save(tmp_path / "c.npy", c)
    (tmp_path / "conv.json").write_text(json.dumps([2.0 + 0.1 * k for k in range(N)]))
    mp.spawn(_clip_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    final = str(tmp_path / "d.npy") + "_stereo.npy"
    want = {"": np.concatenate([c, d], axis=2)}
    want["_holemask.npy"] = want[""][..., 0] ^ 0x5A
    want["_depth.npy"] = want[""][..., ::-1]
    want["_infillmask.npy"] = 255 - want[""]
    for suffix, arr in want.items():
        path = final + suffix
        assert not os.path.exists(path), "no rank may write the shared single file"
        idx = json.load(open(path + ".index.json"))
        assert idx["frames"] == N and idx["world"] == world and [s["rank"] for s in idx["segments"]] == list(range(world))
        for s in idx["segments"]:
            assert s["file"] == os.path.basename(path) + f".rank{s['rank']}of{world}.npy"
            seg = np.load(tmp_path / s["file"])
            assert seg.shape[0] == s["hi"] - s["lo"] and np.array_equal(seg, arr[s["lo"]:s["hi"]])
        view = clip.open_output(path)
        assert len(view) == N and view.shape == arr.shape and np.array_equal(np.asarray(view), arr)
        assert np.array_equal(view[N - 1], arr[N - 1]) and np.array_equal(view[1:N], arr[1:N]) and np.array_equal(view[-1], arr[-1])
    assert not [f for f in os.listdir(tmp_path) if "_tmp_" in f], "every segment was renamed by its rank"
    stats = np.load(tmp_path / "stats0.npy")
    assert stats.shape == (world, 3) and stats[:, 0].sum() == N and stats[:, 2].sum() == 7.0 * N
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / f"stats{r}.npy"), stats)
    merged = clip.merge_output(final)
    assert merged == final and np.array_equal(np.load(final), want[""]) and not os.path.exists(final + ".index.json")
    assert not [f for f in os.listdir(tmp_path) if f.startswith("d.npy_stereo.npy.rank")]


def test_output_plan_names_and_shapes():
    from metric_depth_video_toolbox_amd import clip, distributed as D
    cp = D.ClipParameters(64, 48, 10, 0.063, 100.0, 45.0, 2 | 4 | 8, np.full(10, 45.0), np.zeros(10))
    one = clip.plan_outputs("/x/depth.npy", cp, 1, create_sbs_depth_video=True, infill_mask=True)
    assert sorted(one) == ["depth", "infill", "mask", "sbs"]
    assert one["sbs"]["final"] == "/x/depth.npy_stereo.npy" and one["sbs"]["tmp"] == "/x/depth.npy_tmp_stereo.npy"
    assert one["mask"]["final"] == "/x/depth.npy_stereo.npy_holemask.npy" and one["mask"]["frame_shape"] == (48, 128)
    assert one["infill"]["tmp"] == "/x/depth.npy_tmp_stereo.npy_infillmask.npy" and one["sbs"]["segments"] == [(0, 0, 10)]
    assert clip.segment_path(one["sbs"]["final"], 0, 1) == one["sbs"]["final"]          # a single rank writes the reference's one file
    four = clip.plan_outputs("/x/depth.npy", cp, 4)
    assert sorted(four) == ["mask", "sbs"] and four["sbs"]["segments"] == [(0, 0, 2), (1, 2, 5), (2, 5, 7), (3, 7, 10)]
    assert clip.segment_path(four["sbs"]["tmp"], 2, 4) == "/x/depth.npy_tmp_stereo.npy.rank2of4.npy"
    t1 = D.ClipParameters(64, 48, 4, 0.063, 100.0, 45.0, 64, np.full(4, 45.0), np.zeros(4))
    assert clip.plan_outputs("/x/d.npy", t1, 1)["sbs"]["final"] == "/x/d.npy_Touchly1.npy"
    assert clip.plan_outputs("/x/d.npy", t1, 1)["sbs"]["frame_shape"] == (96, 64, 3)
