"""Frame sharding over ranks.

The reference's only parallelism is one OS process per scene with the parameters passed as argv
(movie_2_3D.py:433-452).  Here: one process per GPU; rank 0 owns the clip-level parameters (per-frame
xfov, the NaN-filled and smoothed convergence curve, the lock-frame re-based poses) and broadcasts
them as ONE small block (80 + 144*N bytes) over RCCL/xGMI; every rank then renders its own contiguous
frame range with no further exchange.  A final all-gather of three doubles per rank feeds the report.

Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

HEADER_DOUBLES = 10    # W, H, N, ipd_m, max_depth, master_xfov, mode_flags, has_T, touchly_max_depth, touchly_min_depth
PER_FRAME_DOUBLES = 18  # xfov, convergence distance, T[16]


@dataclass
class ClipParameters:
    """What stereo_rerender.py sets up before its loop (sr:343-373, 458-459)."""
    W: int
    H: int
    n_frames: int
    ipd_m: float
    max_depth: float
    master_xfov: float
    mode_flags: int                      # bit0 pointcloud, bit1 remove_edges, bit2 edge_points, bit3 infill key colour,
                                         # bit4 vr180, bit5 touchly0, bit6 touchly1 (sr:291-303, 406-407)
    xfov: np.ndarray                     # [N] degrees (sr:351-359, or the constant --xfov)
    convergence: np.ndarray              # [N] metres, already NaN-filled + smoothed (sr:343-349); 0 = none
    transformations: Optional[np.ndarray] = None   # [N,4,4], already re-based on the lock frame (sr:369-373)
    touchly_max_depth: float = 5.0       # sr:302-303
    touchly_min_depth: float = 0.0

    def pack(self) -> np.ndarray:
        N = self.n_frames
        blk = np.zeros(HEADER_DOUBLES + PER_FRAME_DOUBLES * N, np.float64)
        blk[:HEADER_DOUBLES] = [self.W, self.H, N, self.ipd_m, self.max_depth, self.master_xfov, self.mode_flags,
                                0.0 if self.transformations is None else 1.0, self.touchly_max_depth, self.touchly_min_depth]
        body = blk[HEADER_DOUBLES:].reshape(N, PER_FRAME_DOUBLES)
        body[:, 0] = self.xfov
        body[:, 1] = self.convergence
        if self.transformations is not None:
            body[:, 2:] = np.asarray(self.transformations, np.float64).reshape(N, 16)
        return blk

    @staticmethod
    def unpack(blk: np.ndarray) -> "ClipParameters":
        blk = np.asarray(blk, np.float64)
        W, H, N = int(blk[0]), int(blk[1]), int(blk[2])
        body = blk[HEADER_DOUBLES:HEADER_DOUBLES + PER_FRAME_DOUBLES * N].reshape(N, PER_FRAME_DOUBLES)
        T = body[:, 2:].reshape(N, 4, 4).copy() if blk[7] != 0.0 else None
        return ClipParameters(W, H, N, float(blk[3]), float(blk[4]), float(blk[5]), int(blk[6]),
                              body[:, 0].copy(), body[:, 1].copy(), T, float(blk[8]), float(blk[9]))


def frame_range(rank: int, world: int, n_frames: int):
    """Contiguous range [lo, hi) of rank `rank` (keeps per-rank video decode sequential)."""
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def rebase_on_lock_frame(transformations: Sequence, lock_frame: int) -> np.ndarray:
    """sr:369-373: T_i <- T_i @ inv(T_lock)."""
    T = np.asarray(transformations, np.float64)
    if lock_frame != 0:
        inv = np.linalg.inv(T[lock_frame])
        T = np.stack([t @ inv for t in T])
    return T


def init_process_group(backend: Optional[str] = None):
    """Join the job torchrun started (RANK / WORLD_SIZE / MASTER_* in the env).  Returns (rank, world)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # torchrun: join even with one rank
    if (world > 1 or launched) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def _collective_device(device):
    """Where a collective's tensors live: the caller's device under RCCL (default: the current one), host memory under
    gloo (whose device-tensor support covers only some collectives: the tests that put two ranks on one GPU use it)."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() != "nccl":
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else device


def broadcast_clip_parameters(params: Optional[ClipParameters], src: int = 0, device=None) -> ClipParameters:
    """One broadcast of the packed parameter block from `src` (two tiny collectives: length, payload).
    With a single process this is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        assert params is not None
        return params
    device = _collective_device(device)
    rank = dist.get_rank()
    blk = params.pack() if rank == src else None
    n = torch.tensor([0 if blk is None else blk.size], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    t = torch.from_numpy(blk).to(device) if rank == src else torch.empty(int(n.item()), dtype=torch.float64, device=device)
    dist.broadcast(t, src=src)
    return ClipParameters.unpack(t.cpu().numpy())


def gather_rank_stats(frames: float, seconds: float, hole_px: float, device=None) -> np.ndarray:
    """All-gather of (frames, seconds, hole pixels) per rank -> [world, 3]."""
    import torch
    import torch.distributed as dist
    mine = np.array([frames, seconds, hole_px], np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return mine[None]
    device = _collective_device(device)
    t = torch.from_numpy(mine).to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    device = _collective_device(device)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
