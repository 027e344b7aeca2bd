#!/usr/bin/env python3
"""Pin the rasteriser stage (depth_map_tools.render, dmt:1422-1572: Open3D legacy Visualizer -> OpenGL) against a CONFORMANT
OpenGL that runs in the build container: SwiftShader's OpenGL ES 3.0 (gl_swiftshader.py), headless.

    python tests/golden/gen_gl_golden.py            # rewrites tests/golden/render_gl_*.npz    (build container only)

What is the reference's and what is restated, per stage of the loop body (sr:489-941) for one frame:

  * decode, master-fov scale, camera matrix, grid mesh / point cloud, the 89-degree filter, parked vertices: THE REFERENCE'S OWN
    functions, imported from /root/reference over the usual stub modules (gen_golden.import_reference):
    dfh.decode_rgb_depth_frame, dmt.compute_camera_matrix, dmt.get_mesh_from_depth_map, dmt.convert_mesh_to_pcd,
    sr.convergence_angle.
  * Open3D's in-place geometry operations (transform / rotate / translate, sr:615-616, 723-725, 831-836): restated in f64
    from their published definitions (gen_golden._O3dPoints, the same stand-in edge_points.npz was made with).
  * dmt.render itself: restated as the GL calls Open3D's legacy Visualizer issues for it, as far as its published source is
    remembered -- NOT observable here, so every state that is not certain is a recorded parameter of the fixture instead of a
    silent choice:
      - geometry copied, Y scaled by cam[1][1] / cam[0][0] in f64 (dmt:1532-1541), uploaded as float32;
      - vertex colours c / 255 in f64 (dmt:1228), uploaded as float32;
      - MVP = Perspective(fovy, W / H, 1e-4, far) * LookAt(eye 0, looking along +z, up = -y) with
        fovy = 2 atan(H / (2 cam[0][0])) (the "focal length in the fy slot" hack of dmt:1543-1548; Open3D clamps fovy to
        [5, 90] degrees), near = 1e-4 (dmt:1520), far = |z of the bounding-box centre| + 3 x its largest extent
        (ViewControl::SetProjectionParameters), composed in f64 and cast to float32;
      - `gl_Position = MVP * vec4(position, 1)`, colour passed through, light off (dmt:1553), GL_LESS depth test, clear to
        bg_color, GL_TRIANGLES in the order of mesh.triangles / GL_POINTS of size 1 (dmt:1510) in vertex order;
      - `cull`: Open3D enables GL_CULL_FACE unless RenderOption.mesh_show_back_face is set (the reference never sets it) --
        both settings are rendered;
      - `samples`: Open3D's window asks GLFW for a 4x multisampled framebuffer -- single-sample and 4x are both rendered;
      - read back RGBA8 (what glReadPixels(GL_FLOAT) * 255 -> uint8 returns for a UNORM8 buffer), flipped so
        that row 0 is the top of the window; hole mask = (rgb == bg) (sr:740, 854), holes black (sr:793, 819).

Each fixture holds the INPUTS (depth_rgb, color_rgb, pose), the scene parameters, and per (cull, samples) variant and eye the
GL's colour and hole mask, plus the GL strings (renderer, version, GL_SUBPIXEL_BITS).  Nothing of the reference is stored.
SwiftShader snaps vertices to a 1/16-pixel grid (GL_SUBPIXEL_BITS 4); the decree's default is 8 bits (what desktop GPUs
report), so the tests run the oracle and the HIP path with `subpixel_bits = 4` against these files.
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from render_gl_scenes import GL_SCENES, VARIANTS, gl_scene_inputs        # noqa: E402


# ------------------------------------------------------------------------------------------------ Open3D's view set-up
def open3d_mvp(W, H, cam, positions):
    """ViewControl::ConvertFromPinholeCameraParameters + SetProjectionParameters + gl_util::Perspective / LookAt for the
    parameters dmt:1512-1549 hands over (extrinsic = identity; intrinsic = [[999999, 0, cx], [0, cam[0][0], cy], [0, 0, 1]]).
    -> (MVP float32 4x4, dict of what went into it)."""
    fy_slot = float(cam[0][0])
    tan_half = H / (fy_slot * 2.0)
    fov = min(max(math.atan(tan_half) * 2.0 * 180.0 / math.pi, 5.0), 90.0)
    lo, hi = positions.min(axis=0), positions.max(axis=0)
    centre, extent = (lo + hi) / 2.0, float(np.max(hi - lo))
    front = np.array([0.0, 0.0, -1.0])
    ideal_distance = abs(float(np.dot(-centre, front)))                  # eye = 0
    tan_fov = math.tan(fov * 0.5 / 180.0 * math.pi)
    zoom = min(max(ideal_distance * tan_fov / extent, 0.02), 2.0)
    distance = zoom * extent / tan_fov
    near, far = 1e-4, distance + 3.0 * extent
    P = np.zeros((4, 4))
    t = math.tan(fov / 180.0 * math.pi / 2.0)
    P[0, 0] = 1.0 / (W / H) / t
    P[1, 1] = 1.0 / t
    P[2, 2] = -(far + near) / (far - near)
    P[3, 2] = -1.0
    P[2, 3] = -2.0 * far * near / (far - near)
    V = np.diag([1.0, -1.0, -1.0, 1.0])                                  # LookAt(0, (0,0,d), up = (0,-1,0))
    mvp = P.astype(np.float32) @ V.astype(np.float32)
    return mvp, dict(fovy=fov, near=near, far=far)


def dmt_render_gl(gl, positions, colors, triangles, cam, W, H, bg, cull, samples, mirror_y=False):
    """dmt.render(objects, cam, depth=False, bg_color=bg) through the GL. -> (rgb u8[H,W,3] as sr:819 makes it, mask u8[H,W])"""
    pos = np.array(positions, np.float64)
    pos[:, 1] *= cam[1][1] / cam[0][0]                                    # dmt:1532-1541
    mvp, view = open3d_mvp(W, H, cam, pos)
    rgba, _ = gl.render(W, H, pos.astype(np.float32), np.asarray(colors, np.float64).astype(np.float32), mvp, triangles,
                        bg=bg, cull=cull, samples=samples, read_depth=False, mirror_y=mirror_y)
    img = rgba[..., :3].astype(np.float32) / np.float32(255.0)           # capture_screen_float_buffer
    mask = np.all(img == np.asarray(bg, np.float64), axis=-1)            # sr:740
    out = (img * 255).astype(np.uint8)                                   # sr:819
    out[mask] = 0                                                        # sr:793
    return out, mask.astype(np.uint8) * 255, view


def render_scene(gl, dfh, dmt, sr, sc, depth_rgb, color, T, variants=VARIANTS, mirror_y=False):
    from gen_golden import _O3dPoints, _rotation_matrix_from_xyz
    W, H = sc["W"], sc["H"]
    cam = dmt.compute_camera_matrix(sc["xfov"], None, W, H)                                   # sr:515-525
    depth = dfh.decode_rgb_depth_frame(depth_rgb, 100, True)                                  # sr:512
    scale = 1.0 / (np.tan(np.radians(45.0 / 2)) / np.tan(np.radians(sc["xfov"] / 2)))        # sr:537-538, master_xfov 45
    depth = depth * np.float32(scale) if scale != 1.0 else depth                              # sr:541 (f32 in place)
    bg = np.array([0.0, 1.0, 0.0]) if sc["remove_edges"] else np.array([0.0, 0.0, 0.0])       # sr:555-558
    mesh, unused, _normals = dmt.get_mesh_from_depth_map(depth, cam, color, None, remove_edges=sc["remove_edges"],
                                                         of_by_one=not sc["pointcloud"], return_normals_of_removed=True)
    triangles = None if sc["pointcloud"] else np.asarray(mesh.triangles).copy()
    geo = _O3dPoints()
    if sc["pointcloud"]:
        geo.points = np.zeros_like(np.asarray(mesh.vertices, np.float64))
        geo.colors = np.zeros_like(np.asarray(mesh.vertex_colors, np.float64))
        dmt.convert_mesh_to_pcd(mesh, unused, geo)                                            # sr:609: parks the removed vertices (dmt:1091)
        colors = geo.colors
    else:
        geo.points = np.asarray(mesh.vertices, np.float64).copy()
        colors = np.asarray(mesh.vertex_colors, np.float64)
    if T is not None:
        geo.transform(T)                                                                      # sr:615-616
    half = sc["ipd_mm"] / 1000 / 2                                                            # sr:458-459
    rot_minus = rot_plus = None
    if sc["convergence"]:
        a = sr.convergence_angle(sc["convergence"] * scale, sc["ipd_mm"] / 1000)              # sr:716-719
        rot_plus, rot_minus = _rotation_matrix_from_xyz((0, a, 0)), _rotation_matrix_from_xyz((0, -a, 0))
    out = {}
    if rot_minus is not None:
        geo.rotate(rot_minus, center=(0, 0, 0))                                               # sr:723-724
    geo.translate([half, 0.0, 0.0])                                                           # sr:725
    for eye in ("left", "right"):
        for cull, samples in variants:
            rgb, mask, view = dmt_render_gl(gl, geo.points, colors, triangles, cam, W, H, bg, cull, samples, mirror_y)
            tag = f"{eye}_c{int(cull)}s{samples}"
            out[tag + "_rgb"], out[tag + "_mask"] = rgb, mask
            out[eye + "_far"] = np.float64(view["far"])
        if eye == "left":                                                                     # sr:831-836
            geo.translate([-half, 0.0, 0.0])
            if rot_plus is not None:
                geo.rotate(rot_plus, center=(0, 0, 0)); geo.rotate(rot_plus, center=(0, 0, 0))
            geo.translate([-half, 0.0, 0.0])
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default=None, help="comma separated scene names")
    args = ap.parse_args()
    from gen_golden import import_reference
    from gl_swiftshader import GL
    dfh, dmt, sr, _ic = import_reference()
    gl = GL()
    print(gl.meta())
    for sc in GL_SCENES:
        if args.only and sc["name"] not in args.only.split(","):
            continue
        depth_rgb, color, T = gl_scene_inputs(sc)
        res = render_scene(gl, dfh, dmt, sr, sc, depth_rgb, color, T)
        path = os.path.join(args.out, f"render_gl_{sc['name']}.npz")
        np.savez_compressed(path, depth_rgb=depth_rgb, color_rgb=color, T=np.zeros((0,)) if T is None else T,
                            gl=np.array(gl.meta()), subpixel_bits=np.int32(gl.subpixel_bits),
                            numpy=np.array(np.__version__), **res)
        print("wrote", path, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
