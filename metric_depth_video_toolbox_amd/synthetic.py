"""Deterministic synthetic RGB-coded-depth + colour frames for tests and bench.py.

Data generation only -- this is not part of the render path.  Depth is produced in metres and then
quantised through the 16-bit RGB depth code of the toolbox's depth videos (255**4 scale,
truncation, R = G = high byte, B = low byte; reference depth_frames_helper.py:5-11, 48-61), so the
u8 frames are the ground-truth input exactly as a ``*_depth.mkv`` would deliver them.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 20260927
_CODE_SCALE = 255 ** 4


def quantise_depth_to_rgb(depth_m: np.ndarray, max_depth: float = 100.0) -> np.ndarray:
    """f32/f64 metres -> H x W x 3 u8 (RGB order) 16-bit depth code."""
    d = np.clip(np.asarray(depth_m), 0.0, max_depth).astype(np.float64)
    code = (d * (_CODE_SCALE / float(max_depth))).astype(np.uint32)
    hi = (code >> np.uint32(24)).astype(np.uint8)
    lo = ((code >> np.uint32(16)) & np.uint32(0xFF)).astype(np.uint8)
    return np.ascontiguousarray(np.stack([hi, hi, lo], axis=-1))


class SyntheticScene:
    """Background ramp + ripple with K_fg axis-aligned foreground rectangles."""

    def __init__(self, W: int, H: int, config_id: int = 2, n_fg: int = 12, seed: int | None = None):
        self.W, self.H = int(W), int(H)
        self.seed = BASE_SEED + int(config_id) if seed is None else int(seed)
        rng = np.random.default_rng(self.seed)
        self.rect_w = np.maximum(2, (rng.uniform(0.05, 0.25, n_fg) * W).astype(np.int64))
        self.rect_h = np.maximum(2, (rng.uniform(0.05, 0.25, n_fg) * H).astype(np.int64))
        self.rect_x = rng.integers(0, W, n_fg)
        self.rect_y = rng.integers(0, H, n_fg)
        self.rect_z = rng.uniform(1.0, 3.0, n_fg)
        x = np.arange(W, dtype=np.float64)[None, :]
        y = np.arange(H, dtype=np.float64)[:, None]
        self._bg = 10.0 + 2.0 * (x / W) + 0.25 * np.sin(2 * np.pi * 3 * x / W) * np.cos(2 * np.pi * 2 * y / H)
        gx = (x / max(W - 1, 1)) * 255.0
        gy = (y / max(H - 1, 1)) * 255.0
        self._grad = np.stack([np.broadcast_to(gx, (H, W)), np.broadcast_to(gy, (H, W)),
                               np.broadcast_to(255.0 - 0.5 * (gx + gy), (H, W))], axis=-1)

    def depth_m(self, t: int = 0) -> np.ndarray:
        """Metric depth of frame ``t`` (rectangles translated by (3t, t) px with wrap-around)."""
        W, H = self.W, self.H
        z = self._bg.copy()
        # far-to-near so that nearer rectangles overwrite
        for k in np.argsort(-self.rect_z):
            x0 = int((self.rect_x[k] + 3 * t) % W)
            y0 = int((self.rect_y[k] + t) % H)
            xs = (x0 + np.arange(self.rect_w[k])) % W
            ys = (y0 + np.arange(self.rect_h[k])) % H
            z[np.ix_(ys, xs)] = self.rect_z[k]
        return z

    def frame(self, t: int = 0, max_depth: float = 100.0, key_rgb=((0, 0, 0), (0, 255, 0))):
        """-> (depth_rgb u8[H,W,3], color_rgb u8[H,W,3]) for frame ``t``.

        Colour = smooth gradient + uniform noise; any pixel that lands exactly on a key colour is
        bumped by +1 in blue so that coverage == colour-key mask on the benchmark set."""
        rng = np.random.default_rng(self.seed * 1000003 + int(t))
        noise = rng.integers(0, 256, (self.H, self.W, 3)).astype(np.float64)
        col = np.clip(0.5 * self._grad + 0.5 * noise, 0, 255).astype(np.uint8)
        for key in key_rgb:
            hit = np.all(col == np.array(key, np.uint8), axis=-1)
            col[hit, 2] += 1
        return quantise_depth_to_rgb(self.depth_m(t), max_depth), np.ascontiguousarray(col)

    def clip(self, n_frames: int, t0: int = 0, max_depth: float = 100.0):
        """-> (depth_rgb u8[N,H,W,3], color_rgb u8[N,H,W,3])"""
        d = np.empty((n_frames, self.H, self.W, 3), np.uint8)
        c = np.empty((n_frames, self.H, self.W, 3), np.uint8)
        for k in range(n_frames):
            d[k], c[k] = self.frame(t0 + k, max_depth)
        return d, c


def contention_band(depth_m: np.ndarray, fx: float, ipd_m: float, row0: int, rows: int = 64,
                    c: float | None = None) -> np.ndarray:
    """C4's z-buffer stressor: in ``rows`` rows set Z(u) = fx*b/(c-u) for c-u in [2, 513] so that
    ~512 sources fold onto one or two target pixels of the left eye."""
    H, W = depth_m.shape
    out = depth_m.copy()
    b = ipd_m / 2.0
    c = float(W // 2 + 300) if c is None else float(c)
    u = np.arange(W, dtype=np.float64)
    gap = c - u
    sel = (gap >= 2) & (gap <= 513)
    z = np.where(sel, fx * b / np.where(sel, gap, 1.0), 0.0)
    r1 = min(H, row0 + rows)
    out[row0:r1, sel] = z[sel][None, :]
    return out


def synthetic_pose_track(n_frames: int) -> np.ndarray:
    """C4's camera track in the align_3d_points JSON shape: list of 4x4, frame 0 identity, frame t =
    yaw 0.02deg*t, pitch 0.01deg*t, translation (1, 0.5, 2) mm * t."""
    Ts = np.zeros((n_frames, 4, 4), np.float64)
    for t in range(n_frames):
        yaw, pitch = np.deg2rad(0.02 * t), np.deg2rad(0.01 * t)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        T = np.eye(4)
        T[:3, :3] = Ry @ Rx
        T[:3, 3] = np.array([1.0, 0.5, 2.0]) * 1e-3 * t
        Ts[t] = T
    return Ts


def c4_clip(n_frames: int = 8, W: int = 3840, H: int = 2160, host_frames: int = 4, xfov: float = 45.0, ipd_m: float = 0.065):
    """BASELINE configs[3]'s clip as bench.py times it and tests/test_gpu_bench_sizes.py checks it: ``host_frames`` scenes of
    SyntheticScene(config_id=4) with the contention band moving down 8 rows per frame; frame k >= host_frames is frame
    k % host_frames rolled by (16, 24) * (k // host_frames) whole pixels.  -> (depth_rgb, color_rgb) u8[N, H, W, 3] (host) and
    the poses: frames 40, 70, ... of synthetic_pose_track (up to 5 degrees of yaw, half a metre of travel)."""
    sc = SyntheticScene(W, H, config_id=4)
    fx = W / (2.0 * np.tan(np.deg2rad(xfov) / 2.0))          # dmt.compute_camera_matrix's fx (dmt:902-934)
    nh = min(host_frames, n_frames)
    d = np.empty((n_frames, H, W, 3), np.uint8)
    c = np.empty((n_frames, H, W, 3), np.uint8)
    for t in range(nh):
        z = contention_band(sc.depth_m(t), fx, ipd_m, row0=H // 2 - 32 + 8 * t, rows=64)
        d[t] = quantise_depth_to_rgb(z)
        _, c[t] = sc.frame(t)
    for k in range(nh, n_frames):
        sh = (16 * (k // nh), 24 * (k // nh))
        d[k] = np.roll(d[k % nh], shift=sh, axis=(0, 1))
        c[k] = np.roll(c[k % nh], shift=sh, axis=(0, 1))
    Ts = synthetic_pose_track(max(300, 40 + 30 * n_frames))[40:40 + n_frames * 30:30]      # one pose per frame (frames 40, 70, ... of the track)
    return d, c, Ts
