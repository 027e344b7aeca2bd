"""Stress for state carried from one context to the next (found by the r03 soak: once in 120 000 cases a frame was rendered with
the PREVIOUS context's per-frame parameter block -- the device address of the block is recycled by hipMalloc).  Alternates
contexts whose parameters give obviously different pictures (a: pure shift with ipd 0 = everything covered; b: pose looking
away = everything a hole) and counts frames that show the other context's picture.  usage: python tests/dbg_param_stress.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
W, H = 96, 96
rng = np.random.default_rng(os.getpid())
code = (2000 + 40 * np.arange(W)[None, :] + 7 * np.arange(H)[:, None]).astype(np.uint32)
d = np.zeros((H, W, 3), np.uint8); d[..., 0] = (code >> 8) & 0xFF; d[..., 2] = code & 0xFF
c = rng.integers(1, 256, (H, W, 3), dtype=np.uint8)
dt, ct = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
T = np.array([[-0.7449975, -0.40574716, 0.52947896, 0.014], [-0.33871512, -0.45370943, -0.82426926, 0.007],
              [0.57467451, -0.79342107, 0.20057969, 0.028], [0, 0, 0, 1.0]])
bad = n = 0
t0 = time.time()
while time.time() - t0 < secs:
    for kind in (0, 1):
        if kind == 0:
            r = sr.StereoRerenderer(W, H, pupillary_distance=0, max_depth=655, master_xfov=25.0, render_as_pointcloud=bool(n & 2))
            p = r.frame_params(xfov=120.0)
        else:
            r = sr.StereoRerenderer(W, H, pupillary_distance=65, max_depth=5, master_xfov=25.0, render_as_pointcloud=bool(n & 2))
            p = r.frame_params(xfov=20.0, convergence_distance=4.7, transformation=T)
        got = r.render(dt, ct, p)
        holes = int((got["mask"] > 0).sum())
        ok = holes == 2 * W * H if kind == 1 else holes < W * H // 4
        if not ok:
            bad += 1
            print(f"pid {os.getpid()} iteration {n} kind {kind}: {holes} hole px -- the other context's picture", flush=True)
        r.close()
        n += 1
print(f"pid {os.getpid()} mode {os.environ.get('MDVT_PARAM_UPLOAD', 'pooled')}: {n} contexts, {bad} wrong")
