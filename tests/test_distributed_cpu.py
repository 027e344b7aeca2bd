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
