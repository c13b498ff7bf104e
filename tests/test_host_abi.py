"""Host-side tests that need no GPU: the C-ABI library loads and exports every symbol include/lmx.h declares, the
bank container and its YAML wire format (readLinemod / writeLinemod: /root/reference/src/rgbdDetector.cpp:1668-1680,
src/renderer.cpp:56-70), status codes for upstream's CV_Assert cases, and the host merge (std::sort + std::unique)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from linemod_pose_estimation_amd import _lib, synth, NativeBank, Detector, merge_raw
from linemod_pose_estimation_amd.bank import TemplateBank, DEFAULT_COLOR_GRADIENT, DEFAULT_DEPTH_NORMAL
from oracle import oracle as o

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _same_bank(a, b):
    assert a.T == b.T and len(a.modalities) == len(b.modalities)
    for ma, mb in zip(a.modalities, b.modalities):
        assert ma["type"] == mb["type"]
        for k in ma:
            assert float(ma[k]) == float(mb[k]) if k != "type" else True
    ca = sorted(a.classes, key=lambda c: c[0])
    cb = sorted(b.classes, key=lambda c: c[0])
    assert [c[0] for c in ca] == [c[0] for c in cb]
    for (_, ta, fa), (_, tb, fb) in zip(ca, cb):
        assert np.array_equal(ta, tb) and np.array_equal(fa, fb)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lmx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lmx_[a-z_0-9A-Z]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS)
    L = C.CDLL(_lib.SO_PATH)
    for s in sorted(declared):
        assert hasattr(L, s), s
    assert _lib.lib().lmx_version().decode().startswith("lmx")
    names = [_lib.lib().lmx_kernel_name(i).decode() for i in range(_lib.lib().lmx_num_kernels())]
    assert "k_score_coarse" in names and len(set(names)) == len(names)


def test_bank_roundtrip_through_c_abi():
    bank = synth.make_bank(7, seed=3, classes=["b_cls", "a_cls"], size_range=(20.0, 50.0))
    nb = NativeBank.from_bank(bank)
    assert nb.class_ids() == ["a_cls", "b_cls"]            # std::map order, like Detector::classIds()
    back = nb.to_bank()
    _same_bank(bank, back)
    L = _lib.lib()
    assert L.lmx_bank_num_templates(nb.h, None) == 14 and L.lmx_bank_num_templates(nb.h, b"a_cls") == 7
    assert L.lmx_bank_pyramid_levels(nb.h) == 2 and L.lmx_bank_T(nb.h, 0) == 5 and L.lmx_bank_T(nb.h, 1) == 8
    t = bank.get_templates("a_cls", 3)
    assert len(t) == 4 and t[0][2] == 0 and t[2][2] == 1 and t[0][3].shape == (63, 3) and t[2][3].shape == (31, 3)


def test_yaml_roundtrip(tmp_path):
    bank = synth.make_bank(5, seed=4, classes=["obj", "other"], size_range=(20.0, 60.0))
    p = tmp_path / "bank_templates.yml"
    NativeBank.from_bank(bank).save_yaml(p)
    text = p.read_text()
    assert text.startswith("%YAML:1.0") and "template_pyramids:" in text and "- [ " in text
    _same_bank(bank, NativeBank.load_yaml(p).to_bank())
    # ColorGradient-only bank (the ensenso trainers write those: src/renderer_only_image.cpp:130-136)
    b1 = synth.make_bank(3, modalities=("ColorGradient",), seed=5, size_range=(20.0, 40.0))
    NativeBank.from_bank(b1).save_yaml(p)
    _same_bank(b1, NativeBank.load_yaml(p).to_bank())


def test_yaml_opencv_style_fixture():
    back = NativeBank.load_yaml(os.path.join(GOLDEN, "opencv_style_templates.yml")).to_bank()
    assert back.T == [5, 8]
    assert [m["type"] for m in back.modalities] == ["ColorGradient", "DepthNormal"]
    assert back.modalities[0]["weak_threshold"] == 10.0 and back.modalities[0]["strong_threshold"] == 55.0
    assert back.modalities[1]["distance_threshold"] == 2000 and back.modalities[1]["extract_threshold"] == 2
    cls = dict((c, (t, f)) for c, t, f in back.classes)
    assert sorted(cls) == ["obj", "zeta"] and cls["zeta"][0].shape[0] == 0
    t, f = cls["obj"]
    assert t.shape == (8, 5)
    assert t[:, :3].tolist() == [[12, 10, 0], [12, 10, 0], [6, 5, 1], [6, 5, 1], [20, 8, 0], [20, 8, 0], [10, 4, 1], [10, 4, 1]]
    assert t[:, 4].tolist() == [3, 1, 2, 1, 2, 1, 1, 1]
    assert f.tolist() == [[0, 0, 1], [12, 3, 7], [5, 10, 4], [6, 5, 2], [0, 0, 1], [6, 1, 7], [3, 2, 2],
                          [20, 8, 0], [1, 2, 3], [10, 4, 5], [10, 4, 0], [5, 2, 5]]


def test_status_codes(tmp_path):
    L = _lib.lib()
    h = C.c_void_p()
    assert L.lmx_bank_load_yaml(b"/nonexistent/file.yml", C.byref(h)) == _lib.LMX_ERR_IO
    bad = tmp_path / "bad.yml"
    bad.write_text("%YAML:1.0\npyramid_levels: 2\nT: [ 5 ]\n")
    assert L.lmx_bank_load_yaml(str(bad).encode(), C.byref(h)) == _lib.LMX_ERR_PARSE
    assert b"T must list" in L.lmx_last_error()
    bad.write_text("%YAML:1.0\npyramid_levels: 1\nT: [ 5 ]\nmodalities:\n   -\n      type: Foo\n")
    assert L.lmx_bank_load_yaml(str(bad).encode(), C.byref(h)) == _lib.LMX_ERR_PARSE
    # > 63 features: upstream CV_Assert(templ.features.size() <= 63) in similarity()
    templ = np.array([[10, 10, 0, 0, 64], [5, 5, 1, 64, 1]], np.int32)
    feats = np.zeros((65, 3), np.int32)
    with pytest.raises(_lib.LmxError) as e:
        NativeBank.from_bank(TemplateBank(T=[5, 8], modalities=[dict(DEFAULT_COLOR_GRADIENT)], classes=[("obj", templ, feats)]))
    assert e.value.status == _lib.LMX_ERR_SHAPE
    feats = np.zeros((2, 3), np.int32)
    feats[0] = (-1, 0, 0)
    templ = np.array([[10, 10, 0, 0, 1], [5, 5, 1, 1, 1]], np.int32)
    with pytest.raises(_lib.LmxError) as e:
        NativeBank.from_bank(TemplateBank(T=[5, 8], modalities=[dict(DEFAULT_COLOR_GRADIENT)], classes=[("obj", templ, feats)]))
    assert e.value.status == _lib.LMX_ERR_INVALID_ARG
    with pytest.raises(_lib.LmxError) as e:
        NativeBank.from_bank(TemplateBank(T=[5, 8, 8, 8, 8], modalities=[dict(DEFAULT_COLOR_GRADIENT)]))
    assert e.value.status == _lib.LMX_ERR_INVALID_ARG


@pytest.mark.skipif(has_gpu(), reason="checks the no-device error path")
def test_no_cpu_fallback_without_device():
    bank = synth.make_bank(2, seed=1, size_range=(20.0, 30.0))
    with pytest.raises(_lib.LmxError) as e:
        Detector(bank, 160, 160)
    assert e.value.status == _lib.LMX_ERR_NO_DEVICE and "no CPU path" in str(e.value)


def test_merge_raw_equals_upstream_sort_unique():
    """lmx_merge_raw (host, no GPU needed) on the oracle's pre-sort records == the oracle's final output, for any
    arrival order and any split of the records into template shards (what the all-gather delivers)."""
    bank = synth.make_bank(60, seed=8, size_range=(30.0, 70.0), classes=["a", "b"])
    sources, _ = synth.make_scene(bank, 320, 240, seed=9)
    det = o.OracleDetector(bank)
    final = det.match(sources, 78.0)
    raw = det.last_raw()
    assert len(raw) > len(final) > 10
    for seed in range(3):
        perm = np.random.default_rng(seed).permutation(len(raw))
        got = merge_raw(raw[perm])
        assert len(got) == len(final)
        for k in final.dtype.names:
            assert np.array_equal(got[k], final[k]), k
    assert len(merge_raw(raw[:0])) == 0


def test_cluster_matches_equals_reference_restatement():
    """lmx_cluster_matches against the oracle's function-by-function restatement of the reference's own
    rcd_voting / cluster_filter / cluster_scoring / nonMaximaSuppressionUsingIOU on realistic match lists."""
    from linemod_pose_estimation_amd.detector import cluster_matches
    bank = synth.make_bank(400, seed=81)
    sources, _ = synth.make_scene(bank, 640, 480, seed=82, n_instances=6)
    det = o.OracleDetector(bank)
    rng = np.random.default_rng(5)
    n_t = 400
    dists = 0.5 + 0.1 * (np.arange(n_t) % 6) + rng.uniform(-0.005, 0.005, n_t)       # six distance rings like the renderer
    rects = np.stack([np.zeros(n_t), np.zeros(n_t), [m["width"] for m in bank.meta["obj"]], [m["height"] for m in bank.meta["obj"]]], 1).astype(np.int32)
    seen = 0
    for thr, step, cthr in [(75.0, 10, 2), (80.0, 16, 0), (70.0, 8, 3), (85.0, 10, 2), (99.9, 10, 2)]:
        m = det.match(sources, thr)
        ref_c, ref_m = o.cluster_matches(m, dists, rects, step, 0.5, 0.1, cthr)
        got_c, got_m = cluster_matches(m, dists, rects, step, 0.5, 0.1, cthr)
        assert len(got_c) == len(ref_c)
        for k in ("index", "rect", "score", "member_begin", "member_count"):
            assert np.array_equal(got_c[k], ref_c[k]), k
        nm = int(ref_c["member_count"].sum()) if len(ref_c) else 0
        assert np.array_equal(got_m[:nm], ref_m[:nm])
        if len(ref_c) > 1:
            assert (np.diff(ref_c["score"]) <= 0).all()          # sorted by score
            seen += 1
        for c in ref_c:                                            # clusters respect the size filter
            assert c["member_count"] > cthr
    assert seen >= 2
    # matches left of / above the image origin (templates larger than the clamp range give negative x, y): the reference's
    # `int X; X /= matches.size();` divides the int as a size_t, so a negative sum comes out as (2^64 + X) / n truncated to int
    m = det.match(sources, 75.0).copy()
    m["x"] -= 700          # whole clusters move left of the origin: sums of two and more negative x
    m["y"][: len(m) // 2] -= 500
    ref_c, ref_m = o.cluster_matches(m, dists, rects, 10, 0.5, 0.1, 1)
    got_c, got_m = cluster_matches(m, dists, rects, 10, 0.5, 0.1, 1)
    assert len(got_c) == len(ref_c) > 3 and (ref_c["rect"][:, :2] > 1 << 20).any()
    for k in ("index", "rect", "score", "member_begin", "member_count"):
        assert np.array_equal(got_c[k], ref_c[k]), k


def _random_lut(seed):
    return np.random.default_rng(seed).choice(np.array([0, 1, 2, 4, 8, 16, 32, 64, 128], np.uint8), (20, 20, 20), p=[0.04] + [0.12] * 8)


def test_normal_lut_is_data_on_the_bank(tmp_path):
    """NORMAL_LUT[20][20][20] (upstream normal_lut.i) is pluggable: set/get, validation, the two file forms, the default generator."""
    L = _lib.lib()
    bank = synth.make_bank(3, seed=8, size_range=(20.0, 40.0))
    nb = NativeBank.from_bank(bank)
    assert nb.normal_lut_origin() == _lib.LMX_LUT_DEFAULT
    assert np.array_equal(nb.normal_lut(), o.normal_lut())          # library default == the oracle's default table
    lut = _random_lut(1)
    nb.set_normal_lut(lut)
    assert nb.normal_lut_origin() == _lib.LMX_LUT_USER and np.array_equal(nb.normal_lut(), lut)
    bad = lut.copy()
    bad[3, 4, 5] = 3                                                # two bits: not a one-hot label
    with pytest.raises(_lib.LmxError) as e:
        nb.set_normal_lut(bad)
    assert e.value.status == _lib.LMX_ERR_INVALID_ARG and np.array_equal(nb.normal_lut(), lut)
    nb.set_normal_lut(None)
    assert nb.normal_lut_origin() == _lib.LMX_LUT_DEFAULT and np.array_equal(nb.normal_lut(), o.normal_lut())
    # raw 8000-byte file
    raw = tmp_path / "normal_lut.bin"
    raw.write_bytes(lut.tobytes())
    nb.load_normal_lut(raw)
    assert np.array_equal(nb.normal_lut(), lut)
    # C initialiser text in the style of OpenCV's normal_lut.i: declaration with dimensions, nested braces, comments
    txt = tmp_path / "normal_lut.i"
    rows = []
    for v3 in range(20):
        plane = ",\n".join("  {" + ", ".join(str(int(v)) for v in lut[v3, v2]) + "}" for v2 in range(20))
        rows.append(" {\n" + plane + "\n }")
    txt.write_text("// generated 20x20x20 table\nstatic const unsigned char NORMAL_LUT[20][20][20] = {\n" + ",\n".join(rows) + "\n}; /* 8000 values */\n")
    nb.set_normal_lut(None)
    nb.load_normal_lut(txt)
    assert np.array_equal(nb.normal_lut(), lut)
    short = tmp_path / "short.i"
    short.write_text("{1, 2, 4}")
    with pytest.raises(_lib.LmxError) as e:
        nb.load_normal_lut(short)
    assert e.value.status == _lib.LMX_ERR_PARSE
    assert L.lmx_bank_set_normal_lut(None, None) == _lib.LMX_ERR_INVALID_ARG


def test_yaml_carries_the_normal_lut_and_flags_foreign_depth_banks(tmp_path, monkeypatch):
    monkeypatch.delenv("LMX_NORMAL_LUT", raising=False)
    bank = synth.make_bank(3, seed=9, size_range=(20.0, 40.0))
    p = tmp_path / "b_templates.yml"
    nb = NativeBank.from_bank(bank)
    nb.save_yaml(p)
    assert "lmx_normal_lut: default" in p.read_text() and not os.path.exists(str(p) + ".normal_lut")
    assert NativeBank.load_yaml(p).normal_lut_origin() == _lib.LMX_LUT_DEFAULT
    lut = _random_lut(2)
    nb.set_normal_lut(lut)
    nb.save_yaml(p)
    assert "lmx_normal_lut: sidecar" in p.read_text() and os.path.getsize(str(p) + ".normal_lut") == 8000
    back = NativeBank.load_yaml(p)
    assert back.normal_lut_origin() == _lib.LMX_LUT_SIDECAR and np.array_equal(back.normal_lut(), lut)
    assert np.array_equal(back.to_bank().normal_lut, lut)
    nb.set_normal_lut(None)
    nb.save_yaml(p)                                                  # back to the default: the stale side-car goes away
    assert not os.path.exists(str(p) + ".normal_lut") and NativeBank.load_yaml(p).normal_lut_origin() == _lib.LMX_LUT_DEFAULT
    # a yml written by OpenCV (no marker, no side-car) with a DepthNormal modality: trained against a table we cannot see
    foreign = NativeBank.load_yaml(os.path.join(GOLDEN, "opencv_style_templates.yml"))
    assert foreign.normal_lut_origin() == _lib.LMX_LUT_UNKNOWN
    desc = _lib.CtxDesc(0, 160, 160, 1, 0, 0, 1, None, 0)
    h = C.c_void_p()
    assert _lib.lib().lmx_ctx_create(foreign.h, C.byref(desc), C.byref(h)) == _lib.LMX_ERR_INVALID_ARG   # before any device is touched
    assert b"normal_lut.i" in _lib.lib().lmx_last_error()
    # ... unless the environment names the table, or the caller decides
    raw = tmp_path / "lut.bin"
    raw.write_bytes(lut.tobytes())
    monkeypatch.setenv("LMX_NORMAL_LUT", str(raw))
    envb = NativeBank.load_yaml(os.path.join(GOLDEN, "opencv_style_templates.yml"))
    assert envb.normal_lut_origin() == _lib.LMX_LUT_SIDECAR and np.array_equal(envb.normal_lut(), lut)
    foreign.set_normal_lut(None)
    assert foreign.normal_lut_origin() == _lib.LMX_LUT_DEFAULT
    # ColorGradient-only banks (the ensenso banks the north star names) are never affected
    cg = tmp_path / "cg.yml"
    cg.write_text(open(os.path.join(GOLDEN, "opencv_style_templates.yml")).read().split("   -\n      type: DepthNormal")[0] + "classes:\n")
    monkeypatch.delenv("LMX_NORMAL_LUT")
    assert NativeBank.load_yaml(cg).normal_lut_origin() == _lib.LMX_LUT_DEFAULT


def test_merge_gathered_reports_dropped_candidates():
    """A gather block whose header says that the rank's candidate list overflowed (word 0 > word 2, the list's capacity) carries
    incomplete matches: lmx_merge_gathered must fail instead of merging them (advisor finding, round 1)."""
    from linemod_pose_estimation_amd.detector import merge_gathered, RAW_MATCH_DTYPE
    from linemod_pose_estimation_amd.dist import make_block, block_bytes
    recs = np.zeros(3, RAW_MATCH_DTYPE)
    recs["similarity"] = [95.0, 93.0, 97.0]
    recs["order_key"] = [1, 2, 3]
    blk = make_block(recs, 16)
    hdr = blk[:64].view(np.uint32)
    hdr[0], hdr[2] = 50, 100                      # 50 candidates, list capacity 100: complete
    out = merge_gathered(np.concatenate([blk, make_block(recs[:0], 16)]), 2, block_bytes(16), 16, 1)
    assert len(out[0]) == 3 and out[0]["similarity"][0] == 97.0
    hdr[0] = 101                                  # the scoring kernel dropped a candidate
    with pytest.raises(_lib.LmxError) as e:
        merge_gathered(np.concatenate([blk, make_block(recs[:0], 16)]), 2, block_bytes(16), 16, 1)
    assert e.value.status == _lib.LMX_ERR_OVERFLOW and "candidate list overflow" in str(e.value)


def test_renderer_params_sidecar_reader_and_writer(tmp_path):
    """`<object>_renderer_params.yml` (readLinemodTemplateParams, /root/reference/src/rgbdDetector.cpp:1681-1749): the side-car the consumer
    chain needs (Ori_dist, Rect, renderer_radius_min / _step).  The fixture is a verbatim sample of the reference's own data file
    (config/data/boxNew_longDistance_linemod_xtion_renderer_params.yml: 6 templates + footer, OpenCV FileStorage with !!opencv-matrix
    values); values checked against the file's text, then a write / read round trip."""
    import ctypes as C
    import re
    from conftest import ROOT
    L = _lib.lib()
    path = os.path.join(ROOT, "tests", "golden", "renderer_params_sample.yml")
    p = C.POINTER(_lib.RendererParams)()
    _lib.check(L.lmx_renderer_params_load(path.encode(), C.byref(p)))
    r = p.contents
    assert r.n_templates == 6 and r.renderer_n_points == 150 and r.renderer_angle_step == 10 and (r.renderer_width, r.renderer_height) == (640, 480)
    assert (r.renderer_radius_min, r.renderer_radius_max, r.renderer_radius_step) == (0.5, 1.0, 0.1) and abs(r.renderer_focal_length_x - 535.566011) < 1e-9
    txt = open(path).read()
    rects = np.asarray([[int(v) for v in m.split(",")] for m in re.findall(r"Rect: \[(.*?)\]", txt)], np.int32)
    assert np.array_equal(np.ctypeslib.as_array(r.rects, (6, 4)), rects)
    ori = np.asarray([float(v) for v in re.findall(r"Ori_dist: (\S+)", txt)])
    assert np.array_equal(np.ctypeslib.as_array(r.obj_origin_dists, (6,)), ori.astype(np.float32).astype(np.float64))   # through a float, like the reference
    R0 = np.ctypeslib.as_array(r.R, (6, 9))[0]
    assert abs(R0[0] - 9.7591209808210677e-01) < 1e-15 and abs(np.linalg.det(R0.reshape(3, 3)) - 1.0) < 1e-9
    out = tmp_path / "again.yml"
    _lib.check(L.lmx_renderer_params_save(p, str(out).encode()))
    q = C.POINTER(_lib.RendererParams)()
    _lib.check(L.lmx_renderer_params_load(str(out).encode(), C.byref(q)))
    for name, n in (("obj_origin_dists", 6), ("distances", 6), ("R", 54), ("T", 18), ("K", 54)):
        assert np.array_equal(np.ctypeslib.as_array(getattr(r, name), (n,)), np.ctypeslib.as_array(getattr(q.contents, name), (n,))), name
    assert np.array_equal(np.ctypeslib.as_array(q.contents.rects, (24,)), rects.reshape(-1)) and q.contents.renderer_radius_step == 0.1
    # the arrays plug straight into the consumer chain
    m = np.zeros(3, np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"), ("class_index", "<i4")]))
    m["x"], m["y"], m["similarity"], m["template_id"] = [100, 101, 102], [50, 50, 51], [95.0, 94.0, 93.0], [0, 1, 2]
    from linemod_pose_estimation_amd.detector import cluster_matches
    cl, mem = cluster_matches(m, np.ctypeslib.as_array(r.obj_origin_dists, (6,)), np.ctypeslib.as_array(r.rects, (6, 4)), 8, r.renderer_radius_min, r.renderer_radius_step, 2)
    assert len(cl) == 1 and cl[0]["member_count"] == 3
    L.lmx_renderer_params_free(p)
    L.lmx_renderer_params_free(q)
    bad = tmp_path / "bad.yml"
    bad.write_text(txt.replace("Ori_dist", "Oops"))
    assert L.lmx_renderer_params_load(str(bad).encode(), C.byref(q)) == _lib.LMX_ERR_PARSE and b"incomplete" in L.lmx_last_error()


def test_renderer_params_reader_on_the_reference_file():
    """The complete side-car the reference ships (config/data/boxNew_longDistance_linemod_xtion_renderer_params.yml, 2 MB, 2652 templates =
    26 directions x 6 distances x 17 in-plane rotations) through lmx_renderer_params_load; only where the reference checkout exists
    (the build container), the committed 6-template sample above is what travels."""
    import ctypes as C
    path = "/root/reference/config/data/boxNew_longDistance_linemod_xtion_renderer_params.yml"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    L = _lib.lib()
    p = C.POINTER(_lib.RendererParams)()
    _lib.check(L.lmx_renderer_params_load(path.encode(), C.byref(p)))
    r = p.contents
    assert r.n_templates == 2652 and (r.renderer_radius_min, r.renderer_radius_max, r.renderer_radius_step) == (0.5, 1.0, 0.1)
    rects = np.ctypeslib.as_array(r.rects, (2652, 4))
    assert rects[:, 2].min() == 55 and rects[:, 3].max() == 194 and np.median(rects[:, 2]) == 114     # the distribution synth.make_bank draws from
    d = np.ctypeslib.as_array(r.obj_origin_dists, (2652,))
    assert np.allclose(np.unique(np.round(d, 3)), [0.5, 0.6, 0.7, 0.8, 0.9, 1.0])
    R = np.ctypeslib.as_array(r.R, (2652, 3, 3))
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-9)
    L.lmx_renderer_params_free(p)


def test_damaged_files_end_in_a_status_never_in_the_process(tmp_path):
    """The readers take files the deployment did not write (banks, the renderer-params side-car, the binary cache).  A mutant of the
    side-car with a stray '}' inside a flow sequence used to make the YAML reader append empty items until std::bad_alloc unwound
    through the C ABI and ended the process (found by scripts/fuzz_files.py): now every reader makes progress or fails, values that
    do not fit an int32 are parse errors, and the entry points catch what still throws.  Plus 300 mutants of each kind of file."""
    import ctypes as C
    import subprocess
    import sys
    from conftest import ROOT
    L = _lib.lib()
    txt = open(os.path.join(ROOT, "tests", "golden", "renderer_params_sample.yml")).read()
    bad = tmp_path / "stray_brace.yml"
    bad.write_text(txt.replace("240., 0., 0.,", "240.} 0., 0.,", 1))
    q = C.POINTER(_lib.RendererParams)()
    assert L.lmx_renderer_params_load(str(bad).encode(), C.byref(q)) == _lib.LMX_ERR_PARSE and b"expected" in L.lmx_last_error()
    big = tmp_path / "huge_rect.yml"
    big.write_text(txt.replace("Rect: [ ", "Rect: [ 99999999999, ", 1))
    assert L.lmx_renderer_params_load(str(big).encode(), C.byref(q)) == _lib.LMX_ERR_PARSE
    yml = open(os.path.join(ROOT, "tests", "golden", "opencv_style_templates.yml")).read()
    h = C.c_void_p()
    for mutant in (yml.replace("[", "[ }", 1), yml.replace("63", "4294967297", 1), yml[: len(yml) // 2], yml.replace(":", "", 3)):
        f = tmp_path / "m.yml"
        f.write_text(mutant)
        st = L.lmx_bank_load_yaml(str(f).encode(), C.byref(h))
        assert st in (_lib.LMX_OK, _lib.LMX_ERR_PARSE, _lib.LMX_ERR_SHAPE, _lib.LMX_ERR_INVALID_ARG)
        if st == _lib.LMX_OK:
            L.lmx_bank_destroy(h)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_files.py"), "1200", "3"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert res.returncode == 0 and "file fuzz ok: 1200 mutants" in res.stdout, res.stdout[-1500:] + res.stderr[-1500:]


def test_cluster_chain_refuses_inputs_whose_depth_ring_is_undefined():
    """`(int)((dist - radius_min) / radius_step)` (the vote's third index, src/rgbdDetector.cpp:48-56) is undefined in the reference for a
    step of zero, a NaN distance or a quotient an int cannot hold, and would differ between the host and the device chain here:
    refused with a message instead (found by UBSan under random inputs); rect sums wrap like the reference's ints."""
    from linemod_pose_estimation_amd.detector import cluster_matches
    m = np.zeros(3, np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"), ("class_index", "<i4")]))
    m["x"], m["y"], m["similarity"], m["template_id"] = [100, 101, 102], [50, 50, 51], [95.0, 94.0, 93.0], [0, 1, 2]
    dists = np.asarray([0.5, 0.5, 0.5])
    rects = np.asarray([[0, 0, 40, 30]] * 3, np.int32)
    cl, _ = cluster_matches(m, dists, rects, 8, 0.5, 0.1, 2)
    assert len(cl) == 1 and tuple(cl[0]["rect"]) == (101, 50, 40, 30)
    for kw, msg in ((dict(step=0.0), "renderer_radius_step"), (dict(step=-0.1), "renderer_radius_step"), (dict(step=1e-30, dists=np.asarray([0.6, 0.6, 0.6])), "depth ring"),
                    (dict(dists=np.asarray([0.5, np.nan, 0.5])), "depth ring"), (dict(dists=np.asarray([0.5, np.inf, 0.5])), "depth ring")):
        with pytest.raises(_lib.LmxError) as e:
            cluster_matches(m, kw.get("dists", dists), rects, 8, 0.5, kw.get("step", 0.1), 2)
        assert e.value.status == _lib.LMX_ERR_INVALID_ARG and msg in str(e.value), (kw, str(e.value))
    huge = np.asarray([[0, 0, 2 ** 31 - 1, 2 ** 31 - 1]] * 3, np.int32)
    cl, _ = cluster_matches(m, dists, huge, 8, 0.5, 0.1, 2)       # 3 * (2^31 - 1) wraps to 2^31 - 3 like an int sum; / 3 as size_t
    assert len(cl) == 1 and cl[0]["rect"][2] == (3 * (2 ** 31 - 1) - 2 ** 32) // 3
