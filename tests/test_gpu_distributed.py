"""The multi-rank path on real hardware: backend "nccl" (= RCCL on ROCm) with the helpers of distributed.py, and
bench.py's refusal to measure fewer GPUs than it was asked for (the reference's analogue is one OS
process per scene with argv parameters, movie_2_3D.py:433-452)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RCCL_SCRIPT = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MDVT_REPO"])
from metric_depth_video_toolbox_amd import distributed as D
rank, world = D.init_process_group()            # RANK / WORLD_SIZE / MASTER_* from the env -> backend nccl
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == world
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
N = 7
T = np.tile(np.eye(4), (N, 1, 1)); T[:, 0, 3] = np.arange(N) * 1e-3
clip = D.ClipParameters(1920, 1080, N, 0.065, 100.0, 45.0, 0b1110, np.linspace(40, 50, N), np.linspace(2, 3, N), T) if rank == 0 else None
got = D.broadcast_clip_parameters(clip, src=0, device=dev)      # device tensors through RCCL
assert (got.W, got.H, got.n_frames, got.mode_flags) == (1920, 1080, N, 0b1110)
assert np.array_equal(got.xfov, np.linspace(40, 50, N)) and np.array_equal(got.convergence, np.linspace(2, 3, N))
assert np.array_equal(got.transformations, T)
st = D.gather_rank_stats(5.0 + rank, 0.25, 1234.0, device=dev)
assert st.shape == (world, 3) and st[rank, 0] == 5.0 + rank
assert D.max_over_ranks(1.5 + rank, device=dev) == 1.5 + (world - 1)
t = torch.ones(1 << 20, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
assert float(t[0]) == world
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", world)
"""


def test_rccl_backend_carries_the_parameter_block_and_the_rank_statistics():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               MDVT_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "RCCL_OK 1" in p.stdout, p.stdout + p.stderr


def test_bench_never_measures_fewer_gpus_than_requested():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    ask = have + 1
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(ask), "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and f"--gpus {ask}" in p.stderr and "{" not in p.stdout, p.stdout + p.stderr
    if have >= 2:           # on a multi-GPU node plain `python bench.py --gpus 2` becomes two RCCL ranks by itself
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--frames", "4", "--no-extra", "--no-cpu-baseline", "--prewarm-ms", "0"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and len(line["config"]["devices"]) == 2


def test_bench_line_under_torchrun_with_one_rank():
    """The driver's launch line with N = 1 ranks: the RCCL process group exists and n_gpus comes from it."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--frames", "4", "--no-cpu-baseline", "--prewarm-ms", "0", "--clip-frames", "12"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["scaling"] == "weak" and line["roofline"]["bound"] == "hbm"
    assert line["extra"]["clip_c3"]["frames"] == 12 and line["extra"]["clip_c3"]["scaling"] == "strong"
    assert "mesh" in line["extra"] and "product_default" in line["extra"]
    for k in ("c4_4k_pose_points", "c4_4k_pose_mesh", "c5_model_hop"):       # BASELINE configs[3] and [4] ride on the N = 1 line
        assert k in line["extra"] and line["extra"][k]["fps"] > 0, line["extra"].get("c4_error") or line["extra"].get("c5_error")
    assert line["extra"]["c4_4k_pose_mesh"]["frames_per_launch"] == 8 and 0 < line["extra"]["c5_model_hop"]["render_share"] < 0.5 and line["extra"]["c5_model_hop"]["quantise_and_render_ms_per_frame"] > 0


def test_bench_line_with_two_ranks_sharing_the_gpu():
    """bench.py's N > 1 code path end to end on the hardware at hand (the driver's SCALE run is otherwise the first time it
    executes): the driver's launch line with two ranks, backend gloo, both on cuda:0 (RCCL refuses two ranks on one
    device; MDVT_BENCH_SHARE_GPU is read by bench.py for this test only).  The parameter block is broadcast, every rank
    renders its own frames, rank 0 prints ONE line that counts both ranks' frames, and the strong-scaling clips split
    their frames over the ranks."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MDVT_DIST_BACKEND="gloo", MDVT_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--frames", "8", "--prewarm-ms", "0", "--clip-frames", "30", "--clip-repeats", "3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and len(line["config"]["devices"]) == 2 and line["scaling"] == "weak"
    assert line["config"]["frames_per_step_per_gpu"] == 8
    # the line says what the ranks really ran on: two ranks, ONE physical GPU, and that this was the tests-only sharing mode
    cfg = line["config"]
    assert len(cfg["device_uuids"]) == 2 and cfg["distinct_gpus"] == 1 and cfg["ranks_share_a_gpu"] is True and cfg["ranks_seen_by_the_process_group"] == 2
    # whole-job value: both ranks' frames over the slower rank's wall clock
    assert abs(line["value"] - 2 * 8 * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-6 * line["value"]
    assert line["cpu_baseline"] is None and line["roofline"]["bound"] == "hbm"
    ex = line["extra"]
    assert ex["clip_c3"]["frames"] == 30 and ex["clip_c3"]["n_gpus"] == 2 and ex["clip_c3"]["scaling"] == "strong"
    assert ex["clip_c3_long"]["frames"] == 90 and ex["clip_c3_long"]["repeats"] == 3 and ex["clip_c3_long"]["n_gpus"] == 2
    assert "mesh" not in ex                               # the N = 1 extras stay at N = 1


def test_ranks_sharing_a_gpu_without_the_opt_in_are_refused():
    """Eight ranks on one device must not pass as eight GPUs: without MDVT_BENCH_SHARE_GPU every rank takes cuda:LOCAL_RANK,
    and two ranks whose devices carry the same UUID (here: LOCAL_RANK 1 does not exist on a one-GPU box, or both map to the
    same GPU) end the job with a message instead of a line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a one-GPU box (with two GPUs the launch line is legitimate)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MDVT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--frames", "4", "--prewarm-ms", "0", "--clip-frames", "8", "--clip-repeats", "1", "--no-cpu-baseline", "--no-extra"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")], p.stdout + p.stderr


def test_bench_line_with_eight_ranks_sharing_the_gpu():
    """The width the driver's SCALE run uses, before it does: the driver's launch line with EIGHT ranks (gloo, all on cuda:0).
    One line from rank 0, n_gpus == 8 with eight device names, the weak-scaling value counts all eight ranks' frames, both
    strong-scaling clips split their frames eight ways (clip_c3_long is the entry to read for the >= 6x criterion), and the
    host-side setup time of every rank is on the line and on stderr."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MDVT_DIST_BACKEND="gloo", MDVT_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--frames", "4", "--prewarm-ms", "0", "--clip-frames", "40", "--clip-repeats", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and len(line["config"]["devices"]) == 8 and line["scaling"] == "weak"
    assert len(line["config"]["device_uuids"]) == 8 and line["config"]["distinct_gpus"] == 1 and line["config"]["ranks_share_a_gpu"] is True
    assert abs(line["value"] - 8 * 4 * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-6 * line["value"]
    assert len(line["config"]["setup_seconds_per_rank"]) == 8 and "host-side setup per rank" in p.stderr
    ex = line["extra"]
    assert ex["clip_c3"]["frames"] == 40 and ex["clip_c3"]["n_gpus"] == 8 and "clip_c3_long" in ex["clip_c3"]["note"]
    assert ex["clip_c3_long"]["frames"] == 80 and ex["clip_c3_long"]["n_gpus"] == 8 and ex["clip_c3_long"]["scaling"] == "strong"
    assert "mesh" not in ex and "c4_4k_pose_mesh" not in ex
    # a 40-frame clip over 8 ranks is 5 frames each: contiguous, complete, disjoint
    from metric_depth_video_toolbox_amd import distributed as D
    ranges = [D.frame_range(r, 8, 40) for r in range(8)]
    assert ranges[0][0] == 0 and ranges[-1][1] == 40 and all(ranges[k][1] == ranges[k + 1][0] for k in range(7))


@pytest.mark.parametrize("variant", ["points", "product_default"])
def test_two_ranks_render_one_clip_into_per_rank_segments(tmp_path, variant):
    """BASELINE config C3's multi-rank leg end to end on the hardware at hand: `clip.run` under torchrun with TWO ranks
    (backend gloo, both on cuda:0 -- RCCL refuses two ranks on one device), through the CLI with the reference's flag
    names.  Every rank renders its contiguous frame range into its own segment files; read back through open_output they
    equal the single-rank run's files byte for byte (frames are independent: sharding must not change a bit)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    from metric_depth_video_toolbox_amd import clip
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene
    W, H, N = 256, 144, 21
    d, c = SyntheticScene(W, H, config_id=3, n_fg=6).clip(N)
    flags = ["--xfov", "45", "--pupillary_distance", "65", "--batch", "4", "--create_sbs_depth_video"]
    if variant == "points":
        flags += ["--render_as_pointcloud"]
    else:
        (tmp_path / "conv.json").write_text(json.dumps([2.5 + 0.02 * k if k % 7 else float("nan") for k in range(N)]))
        flags += ["--infill_mask", "--convergence_file", str(tmp_path / "conv.json")]
    outs = {}
    for world in (1, 2):
        wd = tmp_path / f"w{world}"
        wd.mkdir()
        dp, cp = str(wd / "v_depth.npy"), str(wd / "v.npy")
        np.save(dp, d); np.save(cp, c)
        env = dict(os.environ, MDVT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=REPO)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "-m", "metric_depth_video_toolbox_amd.stereo_rerender",
               "--depth_video", dp, "--color_video", cp] + flags
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
        assert p.returncode == 0 and "Processing complete" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
        final = dp + "_stereo.npy"
        kinds = ["", "_holemask.npy", "_depth.npy"] + (["_infillmask.npy"] if variant == "product_default" else [])
        if world == 2:
            assert not os.path.exists(final) and os.path.exists(final + ".rank0of2.npy") and os.path.exists(final + ".rank1of2.npy")
            idx = json.load(open(final + ".index.json"))
            assert [(s["lo"], s["hi"]) for s in idx["segments"]] == [(0, 10), (10, 21)]
        outs[world] = {k: np.asarray(clip.open_output(final + k)) for k in kinds}
        assert not [f for f in os.listdir(wd) if "_tmp_" in f]
    for k in outs[1]:
        assert outs[1][k].shape[0] == N and np.array_equal(outs[1][k], outs[2][k]), f"output {k!r} differs between 1 and 2 ranks"
    assert (outs[1]["_holemask.npy"] > 0).any()
