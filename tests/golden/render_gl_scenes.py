"""The fixture scenes of tests/golden/render_gl_*.npz (the rasteriser stage against a conformant OpenGL, gen_gl_golden.py)
and the (cull, samples) variants rendered for each.  Shared by the generator and by the tests that consume its output; no
reference code is involved in making the inputs."""
import numpy as np

# (cull back faces?, multisample count): the two GL states of dmt.render that the reference leaves to Open3D's defaults
VARIANTS = ((False, 0), (True, 0), (False, 4), (True, 4))

# texture: "smooth" = the gradient only (neighbouring vertex colours differ by a few LSB), "noise" = SyntheticScene's default
# (gradient + uniform noise: neighbouring vertices differ by up to 128 LSB -- every interpolation difference shows).
_D = dict(pointcloud=False, remove_edges=False, ipd_mm=65, xfov=45.0, convergence=None, pose=None, texture="noise",
          band=False, zero_patch=False, key_px=False, n_fg=6)
GL_SCENES = [
    dict(_D, name="mesh_shift_smooth_96x64", W=96, H=64, seed=11, texture="smooth"),
    dict(_D, name="mesh_shift_noise_96x60", W=96, H=60, seed=11),
    dict(_D, name="mesh_shift_noise_320x240", W=320, H=240, seed=12, key_px=True),
    dict(_D, name="mesh_edges_160x90", W=160, H=90, seed=13, remove_edges=True, key_px=True),
    dict(_D, name="mesh_conv_160x90", W=160, H=90, seed=15, convergence=2.5),
    dict(_D, name="mesh_pose_160x90", W=160, H=90, seed=16, ipd_mm=63, xfov=60.0, pose=37),
    dict(_D, name="mesh_pose_conv_edges_160x90", W=160, H=90, seed=17, pose=25, convergence=3.0, remove_edges=True),
    dict(_D, name="mesh_band_256x100", W=256, H=100, seed=18, band=True),                       # C4's contention band: exact depth ties
    dict(_D, name="mesh_band_pose_256x100", W=256, H=100, seed=18, band=True, pose=40),
    dict(_D, name="mesh_zero_patch_96x60", W=96, H=60, seed=19, zero_patch=True),             # Z = 0 vertices: near-plane clipping
    dict(_D, name="points_shift_96x64", W=96, H=64, seed=14, pointcloud=True, key_px=True),   # v sits exactly on pixel corners
    dict(_D, name="points_shift_320x240", W=320, H=240, seed=20, pointcloud=True),
    dict(_D, name="points_edges_160x90", W=160, H=90, seed=21, pointcloud=True, remove_edges=True),
    dict(_D, name="points_conv_160x90", W=160, H=90, seed=22, pointcloud=True, convergence=2.5),
    dict(_D, name="points_pose_160x90", W=160, H=90, seed=23, pointcloud=True, pose=37, xfov=60.0),
    dict(_D, name="points_band_256x100", W=256, H=100, seed=24, pointcloud=True, band=True),
    dict(_D, name="points_zero_patch_96x64", W=96, H=64, seed=25, pointcloud=True, zero_patch=True),
]


def gl_scene_inputs(sc):
    """-> (depth_rgb u8[H,W,3], color_rgb u8[H,W,3], T 4x4 or None): deterministic."""
    from metric_depth_video_toolbox_amd.synthetic import (SyntheticScene, contention_band, quantise_depth_to_rgb,
                                                          synthetic_pose_track)
    W, H = sc["W"], sc["H"]
    scene = SyntheticScene(W, H, seed=sc["seed"], n_fg=sc["n_fg"])
    depth_rgb, color = scene.frame(0)
    if sc["band"]:
        fx = W / (2.0 * np.tan(np.deg2rad(sc["xfov"]) / 2.0))
        z = contention_band(scene.depth_m(0), fx, sc["ipd_mm"] / 1000.0, row0=H // 3, rows=H // 3, c=W // 2 + 100)
        depth_rgb = quantise_depth_to_rgb(z)
    if sc["texture"] == "smooth":
        color = np.clip(scene._grad, 0, 255).astype(np.uint8)
        color[np.all(color == 0, axis=-1), 2] = 1
    if sc["zero_patch"]:
        depth_rgb[3:6, 5:11] = 0                      # Z = 0: on the wrong side of the near plane
        depth_rgb[H // 2, W // 2] = 0                 # a single Z = 0 vertex in the middle of a surface
        depth_rgb[H - 2, W - 3] = (0, 0, 1)           # one depth LSB: 1.55 mm
    if sc["key_px"]:
        color[1, 2] = (0, 0, 0)                       # exact key colours inside the image: the colour-key rule
        color[2, 7] = (0, 255, 0)
        color[H // 2, W // 2] = (0, 0, 0)
        color[H // 2 + 3, W // 2 - 9:W // 2 - 5] = (0, 255, 0) if sc["remove_edges"] else (0, 0, 0)
    T = None if sc["pose"] is None else synthetic_pose_track(sc["pose"] + 1)[sc["pose"]]
    return np.ascontiguousarray(depth_rgb), np.ascontiguousarray(color), T
