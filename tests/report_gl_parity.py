#!/usr/bin/env python3
"""profiles/r06_gl_parity.md: what each GL state that dmt.render (dmt:1422-1572) leaves to the implementation changes in the
output, measured with a conformant OpenGL (SwiftShader, tests/golden/gl_swiftshader.py) on the reference's own geometry at
640x480 and 1920x1080.  BUILD CONTAINER ONLY (needs /root/reference and the kaleido wheel); a report of the test side: it uses
the oracle and is never imported by the product.

    python tests/report_gl_parity.py > profiles/r06_gl_parity.md
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))

import gl_parity                                                  # noqa: E402
import gen_gl_golden as G                                         # noqa: E402
from gen_golden import import_reference                            # noqa: E402
from gl_swiftshader import GL                                      # noqa: E402
from render_gl_scenes import _D, gl_scene_inputs                   # noqa: E402
from oracle import c_oracle as orc                                # noqa: E402


def diff(a_rgb, a_mask, b_rgb, b_mask):
    """-> (hole-mask px that differ, px covered in both whose colour differs by > 1 LSB, covered px)"""
    both = (a_mask == 0) & (b_mask == 0)
    d = np.abs(a_rgb.astype(np.int32) - b_rgb.astype(np.int32)).max(-1)
    return int(((a_mask > 0) != (b_mask > 0)).sum()), int(((d > 1) & both).sum()), int(both.sum())


def main():
    dfh, dmt, sr, _ = import_reference()
    gl = GL()
    print("# r06: the rasteriser stage against a conformant OpenGL -- what each free GL state changes\n")
    print(f"GL: `{gl.meta()}` (the `kaleido` wheel's SwiftShader, headless EGL; `tests/golden/gl_swiftshader.py`).  Geometry: the "
          "reference's own `get_mesh_from_depth_map` / `convert_mesh_to_pcd` (imported from `/root/reference`), Open3D's view set-up "
          "restated (`tests/golden/gen_gl_golden.py`).  Scenes: `SyntheticScene` (noise-textured colours: neighbouring vertices differ "
          "by up to 128 LSB, so every interpolation difference shows), 65 mm, xfov 45.  Numbers are left eye / right eye; `mask` = "
          "hole-mask pixels that differ, `rgb` = pixels covered in both renders whose colour differs by more than 1 LSB.\n")
    rows = []
    for (W, H) in ((640, 480), (1920, 1080)):
        for kind in ("mesh", "mesh+edges", "mesh+conv", "points"):
            sc = dict(_D, name=f"{kind}_{W}x{H}", W=W, H=H, seed=31, pointcloud=kind == "points", remove_edges=kind == "mesh+edges",
                      convergence=2.5 if kind == "mesh+conv" else None, n_fg=12)
            d, c, T = gl_scene_inputs(sc)
            t0 = time.time()
            res = G.render_scene(gl, dfh, dmt, sr, sc, d, c, T, variants=((False, 0), (True, 0), (False, 4)))
            tgl = time.time() - t0
            line = {}
            for bits in (4, 8):
                op = gl_parity.oracle_params(orc, sc, T, False, subpixel_bits=bits)
                o = orc.render_stereo(op, d, c)
                line[f"decree@{bits}"] = [diff(o[e + "_rgb"], o[e + "_mask"], res[e + "_c0s0_rgb"], res[e + "_c0s0_mask"]) for e in ("left", "right")]
            op = gl_parity.oracle_params(orc, sc, T, False, subpixel_bits=4)
            amb = orc.render_stereo_gl(op, d, c, depth_tie_tol=orc.GL_DEPTH_TIE_TOL)
            o = orc.render_stereo(op, d, c)
            expl = []
            for e in ("left", "right"):
                r = gl_parity.compare(o[e + "_rgb"], o[e + "_mask"], res[e + "_c0s0_rgb"], res[e + "_c0s0_mask"], amb[e + "_ambiguous"], sc["pointcloud"])
                expl.append((r["unexplained"], r["unexplained_max"], r["ok"]))
            line["unexplained"] = expl
            line["cull"] = [diff(res[e + "_c1s0_rgb"], res[e + "_c1s0_mask"], res[e + "_c0s0_rgb"], res[e + "_c0s0_mask"]) for e in ("left", "right")]
            line["msaa"] = [diff(res[e + "_c0s4_rgb"], res[e + "_c0s4_mask"], res[e + "_c0s0_rgb"], res[e + "_c0s0_mask"]) for e in ("left", "right")]
            ms = orc.render_stereo_gl(op, d, c, samples=4, pattern=1, resolve=1)
            line["msaa_cand"] = [diff(ms[e + "_rgb"], ms[e + "_mask"], res[e + "_c0s4_rgb"], res[e + "_c0s4_mask"]) for e in ("left", "right")]
            std = orc.render_stereo_gl(op, d, c, samples=4, pattern=0, resolve=0)
            line["msaa_pattern"] = [diff(std[e + "_rgb"], std[e + "_mask"], ms[e + "_rgb"], ms[e + "_mask"]) for e in ("left", "right")]
            rows.append((sc["name"], W * H, tgl, line))
            print(f"<!-- {sc['name']}: GL {tgl:.1f} s -->", file=sys.stderr)

    def cell(v):
        return " / ".join(f"{m} mask, {r} rgb" for m, r, _ in v)

    print("## 1. The decree against the GL (single sample, no culling)\n")
    print("| scene | decree on the GL's grid (4 bits) | of those, not explained (count, max LSB) | decree on its default grid (8 bits) |")
    print("|---|---|---|---|")
    for name, n, tgl, L in rows:
        ue = " / ".join(f"{u} (max {m})" for u, m, _ in L["unexplained"])
        print(f"| {name} | {cell(L['decree@4'])} | {ue} | {cell(L['decree@8'])} |")
    print("\n\"Explained\" = at a pixel the oracle's candidate renderer marks: another fragment of another colour within the window-depth "
          "resolution of the winner (near = 1e-4: |1/Z - 1/Z'| <= 4 x 2^-24 / 1e-4), or a winning triangle that spans more than a "
          "factor 2 in 1/Z (a rubber sheet across a depth edge), or a neighbour of either / of a moved hole pixel (`tests/gl_parity.py`).  "
          "What is left are vertices within float noise of a snapping tie: the GL computes `MVP * v`, the divide and the viewport in its "
          "own f32 order, the decree `u = (fx X)/Z + cx`; a flip moves a vertex by 1/16 px on this grid (1/256 px on the default grid).\n")
    print("## 2. What the states dmt.render never sets change, GL against GL (4-bit grid)\n")
    print("| scene | back-face culling on vs off | 4x multisampling vs one sample | oracle's 4x candidate (SwiftShader's positions) vs the GL's 4x | standard D3D positions + exact mean vs SwiftShader's (candidate vs candidate) |")
    print("|---|---|---|---|---|")
    for name, n, tgl, L in rows:
        print(f"| {name} | {cell(L['cull'])} | {cell(L['msaa'])} | {cell(L['msaa_cand'])} | {cell(L['msaa_pattern'])} |")
    print()


if __name__ == "__main__":
    main()
