#!/usr/bin/env python3
"""Pin the rasteriser against the LITERAL reference: run the reference's own frame-loop calls -- including
depth_map_tools.render (Open3D legacy Visualizer -> OpenGL, dmt:1422-1572) -- on the fixture scenes of
render_scenes.py and write tests/golden/render_<name>.npz.

This cannot run in the build container or on the GPU box (no open3d, no cv2, no GL; dmt.render is Windows-only at
the reference's commit, dmt:1461-1475).  Run it once on any machine where the reference itself runs:

    pip install open3d opencv-python numpy scipy            # what install_mdvtoolbox.sh:18 installs
    python tests/golden/gen_render_golden.py --reference /path/to/metric_depth_video_toolbox

and commit the .npz files it writes.  tests/test_oracle_golden.py::test_rasteriser_against_reference_renders (oracle)
and tests/test_gpu_render.py::test_hip_against_reference_renders (HIP path) then compare against them -- bit-exact
hole mask, RGB within 1 LSB, the bar BASELINE.json states -- and until then skip with a message that says the
rasteriser is unpinned.  Nothing of the reference is copied: the script imports it and records inputs and outputs.

What is recorded per scene and eye: the colour read-back as the loop converts it ((img * 255).astype(uint8), sr:819),
the colour-key hole mask (np.all(img == bg_color), sr:740), the linearised depth read-back (0 = background, dmt:1563),
the edge-point composite where the scene uses it, and the versions of open3d / numpy / the GL renderer string.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def render_scene(dfh, dmt, sr, sc, depth_rgb, color, T):
    """One iteration of the reference's loop body (sr:512-907) for a single frame, through the reference's functions."""
    import copy
    W, H = sc["W"], sc["H"]
    cam = dmt.compute_camera_matrix(sc["xfov"], None, W, H)                                   # sr:515-525
    depth = dfh.decode_rgb_depth_frame(depth_rgb, 100, True)                                  # sr:512
    scale = 1.0 / (np.tan(np.radians(45.0 / 2)) / np.tan(np.radians(sc["xfov"] / 2)))        # sr:537-538, master_xfov 45
    depth *= scale                                                                            # sr:541
    bg = np.array([0.0, 1.0, 0.0]) if sc["remove_edges"] else np.array([0.0, 0.0, 0.0])       # sr:555-558 (--infill_mask)
    mesh, unused, normals = dmt.get_mesh_from_depth_map(depth, cam, color, None, remove_edges=sc["remove_edges"],
                                                        of_by_one=not sc["pointcloud"], return_normals_of_removed=True)   # sr:583
    draw = dmt.convert_mesh_to_pcd(mesh, unused, None) if sc["pointcloud"] else mesh          # sr:609-611
    if T is not None:
        draw.transform(T)                                                                     # sr:615-616
    half = sc["ipd_mm"] / 1000 / 2                                                            # sr:458-459
    rot_minus = rot_plus = None
    if sc["convergence"]:
        a = sr.convergence_angle(sc["convergence"] * scale, sc["ipd_mm"] / 1000)              # sr:716-719
        rot_plus = mesh.get_rotation_matrix_from_xyz((0, a, 0))
        rot_minus = mesh.get_rotation_matrix_from_xyz((0, -a, 0))
    out = {}
    if rot_minus is not None:
        draw.rotate(rot_minus, center=(0, 0, 0))                                              # sr:723-724
    draw.translate([half, 0.0, 0.0])                                                          # sr:725 (left_shift = -ipd/2)
    for eye in ("left", "right"):
        img, dep = dmt.render([draw], cam, depth=-2, bg_color=bg)                             # sr:738 / 852
        out[eye + "_mask"] = (np.all(img == bg, axis=-1)).astype(np.uint8) * 255              # sr:740 / 854
        out[eye + "_rgb"] = (img * 255).astype(np.uint8)                                      # sr:819 / 907
        out[eye + "_depth"] = np.asarray(dep, np.float32)
        if eye == "left":                                                                     # sr:831-836
            draw.translate([-half, 0.0, 0.0])
            if rot_plus is not None:
                draw.rotate(rot_plus, center=(0, 0, 0)); draw.rotate(rot_plus, center=(0, 0, 0))
            draw.translate([-half, 0.0, 0.0])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of calledit/metric_depth_video_toolbox")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    try:
        import open3d as o3d
        import cv2  # noqa: F401  (the reference imports it at module level)
    except ImportError as e:
        sys.exit(f"gen_render_golden.py needs the reference's own dependencies (open3d, opencv-python): {e}")
    sys.path.insert(0, os.path.abspath(args.reference))
    import depth_frames_helper as dfh
    import depth_map_tools as dmt
    import stereo_rerender as sr
    from render_scenes import RENDER_SCENES, scene_inputs
    for sc in RENDER_SCENES:
        depth_rgb, color, T = scene_inputs(sc)
        res = render_scene(dfh, dmt, sr, sc, depth_rgb, color, T)
        path = os.path.join(args.out, f"render_{sc['name']}.npz")
        np.savez_compressed(path, depth_rgb=depth_rgb, color_rgb=color, T=np.zeros((0,)) if T is None else T,
                            versions=np.array(f"open3d {o3d.__version__}; numpy {np.__version__}"), **res)
        print("wrote", path, {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
