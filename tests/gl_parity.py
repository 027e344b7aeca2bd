"""How a render is held to the GL fixtures tests/golden/render_gl_*.npz (made by tests/golden/gen_gl_golden.py with a
conformant OpenGL, SwiftShader) -- shared by tests/test_oracle_golden.py (the oracle), tests/test_gpu_render.py (the HIP path)
and tests/report_gl_parity.py.

What can be asked of two correct rasterisers that do not share their float arithmetic:

  * POINTS: hole mask and colour bit for bit, except where the GL's own depth buffer cannot order the candidates -- the oracle's
    GL-candidate renderer marks those pixels (`ambiguous` bit 0: a losing fragment of another colour within the window-depth
    resolution of the winner, mdvt_oracle.c) -- and except a vertex that sits within float noise of a sub-pixel snapping tie
    and so lands one pixel further (bounded: SNAP_FLIP_FRAC of the pixels).
  * MESH: hole mask bit for bit up to such snap flips at a hole's rim; colour within 1 LSB, except at ambiguous pixels, on
    rubber-sheet triangles that span more than a factor 2 in 1/Z (`ambiguous` bit 1: perspective-correct interpolation there
    amplifies every rounding, and the GL interpolates with plane equations, the decree with barycentrics), next to either, and
    on the few triangles a snap flip reshaped (bounded in number and, outside steep triangles, by the vertex-colour step over the
    sub-pixel grid).

The fixtures were rendered on a 1/16-pixel grid (GL_SUBPIXEL_BITS 4): a flip moves a vertex by 1/16 px there, by 1/256 px on the
default grid of the product, where none of the colour effects would reach 1 LSB.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

SNAP_FLIP_FRAC = 5e-4          # pixels a snap flip may move (hole mask) / recolour beyond the explained set
FLIP_MAX_LSB = 16              # ... and by how much, away from steep triangles (vertex colours step by up to 255 per px, / 16)


def fixture_names():
    import glob
    return sorted(os.path.basename(f)[len("render_gl_"):-len(".npz")] for f in glob.glob(os.path.join(GOLDEN, "render_gl_*.npz")))


def load_fixture(name):
    import sys
    sys.path.insert(0, GOLDEN)
    from render_gl_scenes import GL_SCENES
    sc = {s["name"]: s for s in GL_SCENES}[name]
    g = np.load(os.path.join(GOLDEN, f"render_gl_{name}.npz"), allow_pickle=False)
    T = None if g["T"].size == 0 else g["T"]
    return sc, g, T


def oracle_params(orc, sc, T, cull, subpixel_bits):
    from metric_depth_video_toolbox_amd import stereo_rerender as sr
    W, H = sc["W"], sc["H"]
    p = sr.make_frame_params(W, H, xfov=sc["xfov"], pupillary_distance=sc["ipd_mm"], convergence_distance=sc["convergence"],
                             transformation=T)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    return orc.make_params(W, H, K, ipd_m=sc["ipd_mm"] / 1000, depth_scale=p.depth_scale,
                           mode=orc.MODE_POINTS if sc["pointcloud"] else orc.MODE_MESH, remove_edges=sc["remove_edges"],
                           edge_points=False, conv_angle=p.convergence_angle, T=T,
                           key_rgb=(0, 255, 0) if sc["remove_edges"] else (0, 0, 0), cull=1 if cull else 0,
                           subpixel_bits=subpixel_bits)


def _dilate(m, r=1):
    out = m.copy()
    for _ in range(r):
        p = np.pad(out, 1)
        out = p[1:-1, 1:-1] | p[:-2, 1:-1] | p[2:, 1:-1] | p[1:-1, :-2] | p[1:-1, 2:] | p[:-2, :-2] | p[:-2, 2:] | p[2:, :-2] | p[2:, 2:]
    return out


def compare(got_rgb, got_mask, gl_rgb, gl_mask, ambiguous, points):
    """-> dict of counts; `ok` says whether the rules of the module docstring hold."""
    n = got_mask.size
    mask_diff = (got_mask > 0) != (gl_mask > 0)
    both = (got_mask == 0) & (gl_mask == 0)
    d = np.abs(got_rgb.astype(np.int32) - gl_rgb.astype(np.int32)).max(axis=-1)
    over = (d > (0 if points else 1)) & both
    tie, steep = (ambiguous & 1) > 0, (ambiguous & 2) > 0
    explained = _dilate(tie | steep | mask_diff) if not points else (tie | _dilate(mask_diff))
    unexplained = over & ~explained
    allowed = max(2, int(np.ceil(SNAP_FLIP_FRAC * n)))
    res = dict(pixels=n, covered=int(both.sum()), mask_diff=int(mask_diff.sum()), rgb_over=int(over.sum()),
               rgb_over_frac=float(over.sum()) / max(1, int(both.sum())), unexplained=int(unexplained.sum()),
               unexplained_max=int(d[unexplained].max(initial=0)), ambiguous=int(tie.sum()), steep=int(steep.sum()), allowed=allowed)
    res["ok"] = (res["mask_diff"] <= allowed and res["unexplained"] <= allowed and
                 (points or res["unexplained_max"] <= FLIP_MAX_LSB or res["unexplained"] <= 4))
    return res
