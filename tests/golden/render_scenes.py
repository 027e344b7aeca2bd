"""The fixture scenes for pinning the rasteriser (stage 7, dmt:1422-1572) against the literal reference.  Shared by
gen_render_golden.py (which needs a machine where Open3D can open a GL window) and the tests that consume its output."""
import numpy as np

# name, W, H, scene seed, kwargs of the reference CLI that matter for the frame
RENDER_SCENES = [
    dict(name="mesh_64x48", W=64, H=48, seed=11, pointcloud=False, remove_edges=False, ipd_mm=65, xfov=45.0, convergence=None, pose=None),
    dict(name="mesh_320x240", W=320, H=240, seed=12, pointcloud=False, remove_edges=False, ipd_mm=65, xfov=45.0, convergence=None, pose=None),
    dict(name="mesh_edges_96x64", W=96, H=64, seed=13, pointcloud=False, remove_edges=True, ipd_mm=65, xfov=45.0, convergence=None, pose=None),
    dict(name="points_96x64", W=96, H=64, seed=14, pointcloud=True, remove_edges=False, ipd_mm=65, xfov=45.0, convergence=None, pose=None),
    dict(name="mesh_conv_96x64", W=96, H=64, seed=15, pointcloud=False, remove_edges=False, ipd_mm=65, xfov=45.0, convergence=2.5, pose=None),
    dict(name="mesh_pose_96x64", W=96, H=64, seed=16, pointcloud=False, remove_edges=False, ipd_mm=63, xfov=60.0, convergence=None, pose=37),
]


def scene_inputs(sc):
    """-> (depth_rgb u8[H,W,3], color_rgb u8[H,W,3], T 4x4 or None): deterministic, no reference code involved."""
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene, synthetic_pose_track
    depth_rgb, color = SyntheticScene(sc["W"], sc["H"], seed=sc["seed"], n_fg=6).frame(0)
    T = None if sc["pose"] is None else synthetic_pose_track(sc["pose"] + 1)[sc["pose"]]
    return depth_rgb, color, T
