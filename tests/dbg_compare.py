import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc
W, H = 64, 48
depth_rgb, color = synthetic.SyntheticScene(W, H, seed=W*1000+H, n_fg=6).frame(0)
r = sr.StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=True)
p = r.frame_params(xfov=45.0)
got = r.render(torch.from_numpy(depth_rgb).cuda(), torch.from_numpy(color).cuda(), p, want_depth=True)
K = np.array([p.K[k] for k in range(9)]).reshape(3,3)
op = orc.make_params(W, H, K, ipd_m=0.065, max_depth=100, depth_scale=p.depth_scale, mode=0)
want = orc.render_stereo(op, depth_rgb, color, want_depth=True)
m = got["mask"].cpu().numpy(); z = got["depth"].cpu().numpy(); c = got["sbs"].cpu().numpy()
for eye, sl in (("left", slice(0, W)), ("right", slice(W, 2*W))):
    dm = m[:, sl] != want[eye+"_mask"]
    dz = z[:, sl] != want[eye+"_depth"]
    dc = np.any(c[:, sl] != want[eye+"_rgb"], axis=-1)
    print(eye, "mask diff", dm.sum(), "depth diff", dz.sum(), "rgb diff", dc.sum())
    ys, xs = np.nonzero(dz)
    print(" rows", sorted(set(ys.tolist()))[:20], "cols", sorted(set(xs.tolist()))[:40])
    for y, x in list(zip(ys, xs))[:6]:
        print("  ", y, x, "got z", z[y, sl][x], "want", want[eye+"_depth"][y, x], "got rgb", c[y, sl][x], "want", want[eye+"_rgb"][y, x])
