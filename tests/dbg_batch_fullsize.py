"""One-off soak: a 1080p batch mixing pure-shift / convergence / pose frames, every mode, against the oracle frame by
frame (maskbits and hole counts included).  Debug helper, not collected by pytest."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metric_depth_video_toolbox_amd import stereo_rerender as sr, synthetic
from oracle import c_oracle as orc

W, H, N = 1920, 1080, 10
d, c = synthetic.SyntheticScene(W, H, config_id=3).clip(N)
T = synthetic.synthetic_pose_track(N)
kinds = ["pure", "pure", "conv", "pose", "pure", "conv", "conv", "pure", "pose", "pure"]
for mode in ("points", "points_infill", "mesh", "mesh_infill"):
    r = sr.StereoRerenderer(W, H, pupillary_distance=65, render_as_pointcloud=mode.startswith("points"), infill_mask=mode.endswith("infill"))
    recs = [r.frame_params(xfov=45.0 + (t % 3), convergence_distance=2.5 if k == "conv" else None, transformation=T[t] if k == "pose" else None)
            for t, k in enumerate(kinds)]
    got = r.render(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), recs, want_depth=True, want_maskbits=True, want_hole_counts=True,
                   want_seed=mode.endswith("infill"))
    fin = r.finish_infill_mask_sbs(got["seed"]).cpu().numpy() if mode.endswith("infill") else None
    sbs, mask, z = got["sbs"].cpu().numpy(), got["mask"].cpu().numpy(), got["depth"].cpu().numpy()
    bits, hc = got["maskbits"].cpu().numpy(), got["hole_counts"].cpu().numpy()
    t0 = time.time()
    for t, k in enumerate(kinds):
        K = np.array([recs[t].K[q] for q in range(9)]).reshape(3, 3)
        op = orc.make_params(W, H, K, ipd_m=0.065, depth_scale=recs[t].depth_scale, mode=orc.MODE_POINTS if mode.startswith("points") else orc.MODE_MESH,
                             remove_edges=r.remove_edges, edge_points=int(r.edge_points), conv_angle=recs[t].convergence_angle,
                             T=T[t] if k == "pose" else None, key_rgb=r.key_rgb)
        want = orc.render_stereo(op, d[t], c[t], want_depth=True, want_seed=fin is not None)
        for e, (eye, sl) in enumerate((("left", slice(0, W)), ("right", slice(W, 2 * W)))):
            ok = (np.array_equal(mask[t][:, sl], want[eye + "_mask"]) and np.array_equal(sbs[t][:, sl], want[eye + "_rgb"]) and
                  np.array_equal(z[t][:, sl].view(np.uint32), want[eye + "_depth"].view(np.uint32)))
            pb = np.unpackbits(bits[t][:, e], axis=1, bitorder="little")[:, :W] * 255
            ok = ok and np.array_equal(pb, want[eye + "_mask"]) and hc[t][e] == np.count_nonzero(want[eye + "_mask"])
            if fin is not None:
                ok = ok and np.array_equal(fin[t][:, sl], orc.finish_infill_mask(want[eye + "_seed"], max_rounds=256)[0])
            print(mode, "frame", t, k, eye, "OK" if ok else "MISMATCH", flush=True)
    print(mode, "oracle time", round(time.time() - t0, 1), "s", flush=True)
    r.close()
