"""Parity tests proper (-m gpu): the HIP path, called through the C ABI (liblmx.so), against the CPU oracle on the same
seeded inputs -- bit-exact on every intermediate map (quantised labels, linear memories, colour pyramid) and on the
final (x, y, similarity, template_id, class) list, including its order.  Float tolerance: none; the two float
stages (fastAtan2, normal normalisation) feed integer quantisers and must agree bit for bit, so the comparison is
array_equal everywhere.  Full-size cases (BASELINE.json configs) add size-independent properties: idempotence,
batch == single frame, shard-merge == whole bank, sortedness.
Reference call being replaced: /root/reference/src/rgbdDetector.cpp:31-34."""
import ctypes as C

import numpy as np
import pytest

from linemod_pose_estimation_amd import Detector, NativeBank, _lib, merge_raw, synth, RAW_MATCH_DTYPE
from linemod_pose_estimation_amd.bank import TemplateBank, DEFAULT_COLOR_GRADIENT
from oracle import oracle as o

pytestmark = pytest.mark.gpu


def same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for k in ("x", "y", "similarity", "template_id", "class_index"):
        assert np.array_equal(a[k], b[k]), k


def check_stages(det, od, W, H, L, M, frame=0):
    for l in range(L):
        for m in range(M):
            assert np.array_equal(det.debug_quantized(frame, l, m), od.quantized(l, m, (H >> l, W >> l))), ("quant", l, m)
            assert np.array_equal(det.debug_linear_memory(frame, l, m), od.linear_memory(l, m, (H >> l, W >> l))), ("lm", l, m)


CASES = [
    # W, H, n, modalities, T, thr, size_range, row_pad
    (160, 160, 10, ("ColorGradient",), (5, 8), 70.0, (20.0, 36.0), 0),
    (160, 160, 10, ("ColorGradient", "DepthNormal"), (5, 8), 72.0, (20.0, 36.0), 24),
    (320, 240, 80, ("ColorGradient", "DepthNormal"), (5, 8), 80.0, (30.0, 80.0), 112),
    (640, 480, 200, ("ColorGradient",), (5, 8), 88.0, (55.0, 194.0), 112),   # config 1 shape: ensenso ROI view, stride 752*3
    (640, 480, 200, ("ColorGradient", "DepthNormal"), (5, 8), 85.0, (55.0, 194.0), 0),
    (256, 192, 30, ("ColorGradient", "DepthNormal"), (4, 8), 75.0, (24.0, 60.0), 0),   # upstream default T = {4, 8}
    (320, 320, 30, ("DepthNormal",), (5, 8), 70.0, (30.0, 80.0), 6),
    (240, 240, 20, ("ColorGradient",), (5,), 75.0, (24.0, 60.0), 0),                    # single pyramid level
    (480, 480, 20, ("ColorGradient", "DepthNormal"), (5, 8, 10), 70.0, (40.0, 100.0), 0),  # three levels
]


@pytest.mark.parametrize("W,H,n,mods,T,thr,size_range,row_pad", CASES)
def test_stagewise_and_final_parity(W, H, n, mods, T, thr, size_range, row_pad):
    bank = synth.make_bank(n, modalities=mods, T=T, seed=41, size_range=size_range)
    sources, _ = synth.make_scene(bank, W, H, seed=42, row_pad=row_pad)
    od = o.OracleDetector(bank)
    ref = od.match(sources, thr)
    det = Detector(bank, W, H)
    got = det.match(sources, thr)
    check_stages(det, od, W, H, len(T), len(mods))
    if "ColorGradient" in mods and len(T) > 1:
        assert np.array_equal(det.debug_pyramid_bgr(0, 1, mods.index("ColorGradient")), o.pyrdown(np.ascontiguousarray(sources[mods.index("ColorGradient")])))
    assert det.stats()["candidates"] == od.last_candidates()
    assert det.stats()["raw_matches"] == len(od.last_raw())
    same(got, ref)
    det.close()


@pytest.mark.parametrize("W,H,T", [(320, 240, (5, 8)), (160, 160, (5, 8)), (256, 192, (4, 8)), (480, 480, (5, 8, 10)), (384, 288, (6, 8))])   # T=6: the generic spread kernel
def test_banded_and_flat_spread_images_give_the_same_matches(W, H, T, monkeypatch):
    """k_refine gathers its 16 x 16 patches from the finer levels' spread image, kept banded (LevelGeom::ls_bands, two copies of
    every cell so that a patch is 16 consecutive 32-byte rows) when the level has a multiple of 16 cell columns, flat otherwise
    (LMX_LS_FLAT forces flat).  Low threshold: thousands of refined candidates, patches clamped at every border."""
    bank = synth.make_bank(40, T=T, seed=47, size_range=(20.0, min(W, H) * 0.45))
    sources, _ = synth.make_scene(bank, W, H, seed=470, n_instances=6)
    od = o.OracleDetector(bank)
    ref = od.match(sources, 55.0)
    assert len(ref) > 50 and od.last_candidates() > 300
    for flat in (False, True):
        if flat:
            monkeypatch.setenv("LMX_LS_FLAT", "1")
        det = Detector(bank, W, H, max_candidates=1 << 18)
        got = det.match(sources, 55.0)
        check_stages(det, od, W, H, len(T), 2)
        assert det.stats()["raw_matches"] == len(od.last_raw())
        same(got, ref)
        det.close()


def test_thresholds_and_rerun_idempotent():
    bank = synth.make_bank(120, seed=43, size_range=(40.0, 120.0))
    sources, _ = synth.make_scene(bank, 640, 480, seed=44)
    od = o.OracleDetector(bank)
    det = Detector(bank, 640, 480, max_candidates=1 << 18)   # thr 50 yields tens of thousands of coarse candidates
    seen = set()
    for thr in (50.0, 65.0, 80.0, 92.0, 94.0, 99.0, 100.0):
        ref = od.match(sources, thr)
        a = det.match(sources, thr)
        b = det.match(sources, thr)
        same(a, ref)
        same(a, b)
        seen.add(len(ref))
    assert len(seen) > 3 and max(seen) > 1000
    det.close()


def test_class_filter_and_two_classes():
    bank = synth.make_bank(40, seed=45, size_range=(30.0, 80.0), classes=["memoryChip2", "cpu_binary"])
    sources, _ = synth.make_scene(bank, 320, 240, seed=46, n_instances=5)
    od = o.OracleDetector(bank)
    det = Detector(bank, 320, 240)
    assert det.classIds() == ["cpu_binary", "memoryChip2"] == od.class_ids()
    for cids in [(), ("memoryChip2",), ("cpu_binary",), ("memoryChip2", "cpu_binary"), ("nope",), ("cpu_binary", "nope")]:
        same(det.match(sources, 75.0, cids), od.match(sources, 75.0, cids))
    assert len(det.match(sources, 75.0, ("nope",))) == 0
    det.close()


def test_batch_equals_single_frames():
    bank = synth.make_bank(60, seed=47, size_range=(30.0, 80.0))
    frames = [synth.make_scene(bank, 320, 240, seed=48 + f, row_pad=8 * (f % 2))[0] for f in range(5)]
    od = o.OracleDetector(bank)
    det = Detector(bank, 320, 240, max_batch=5)
    outs = det.match_batch(frames, 78.0)
    for f, src in enumerate(frames):
        ref = od.match(src, 78.0)
        same(outs[f], ref)
        check_stages(det, od, 320, 240, 2, 2, frame=f)
    # split-phase API: resident frames, repeated enqueue/collect, partial batch
    det.upload(frames)
    det.enqueue(5, 78.0)
    outs2 = det.collect(5)
    det.enqueue(3, 78.0)
    outs3 = det.collect(3)
    for f in range(5):
        same(outs2[f], outs[f])
    for f in range(3):
        same(outs3[f], outs[f])
    det.close()


def test_template_shards_merge_to_whole_bank():
    """What the multi-GPU job does, on one GPU: R sharded contexts, raw records read back, host merge."""
    bank = synth.make_bank(90, seed=49, size_range=(30.0, 90.0), classes=["a", "b"])
    sources, _ = synth.make_scene(bank, 320, 240, seed=50, n_instances=5)
    ref = o.OracleDetector(bank).match(sources, 76.0)
    assert len(ref) > 20
    hip = C.CDLL("libamdhip64.so")
    for world in (2, 3, 8):
        recs = []
        for r in range(world):
            det = Detector(bank, 320, 240, shard_rank=r, shard_world=world)
            det.upload([sources])
            det.enqueue(1, 76.0)
            det.sync()
            rec_ptr, cnt_ptr, cap = det.raw_matches_ptrs()
            hdr = np.zeros(16, np.uint32)
            assert hip.hipMemcpy(C.c_void_p(hdr.ctypes.data), C.c_void_p(cnt_ptr), C.c_size_t(64), C.c_int(2)) == 0
            n = int(hdr[1])
            assert n <= cap
            buf = np.zeros(n, RAW_MATCH_DTYPE)
            if n:
                assert hip.hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(rec_ptr), C.c_size_t(n * 32), C.c_int(2)) == 0
            recs.append(buf)
            det.close()
        allrec = np.concatenate(recs[::-1])       # arrival order must not matter
        same(merge_raw(allrec), ref)


def test_edge_cases_empty_bank_big_template_and_border_features():
    # empty bank: no templates at all
    empty = TemplateBank(T=[5, 8], modalities=[dict(DEFAULT_COLOR_GRADIENT)],
                         classes=[("obj", np.zeros((0, 5), np.int32), np.zeros((0, 3), np.int32))])
    img = np.random.default_rng(0).integers(0, 255, (160, 160, 3), dtype=np.uint8)
    det = Detector(empty, 160, 160)
    assert len(det.match([img], 50.0)) == 0
    det.close()
    # a template larger than image - 16T (refinement clamp goes below the border: upstream's max_x < border case),
    # a template larger than the image (template_positions <= 0), features sitting exactly at x == width / y == height
    rng = np.random.default_rng(1)
    def tmpl(w, h, nf0, nf1):
        f0 = np.stack([rng.integers(0, w + 1, nf0), rng.integers(0, h + 1, nf0), rng.integers(0, 8, nf0)], 1)
        f0[0] = (w, h, 3)
        if nf0 > 1:
            f0[1] = (w, 0, 5)
        f1 = np.stack([rng.integers(0, w // 2 + 1, nf1), rng.integers(0, h // 2 + 1, nf1), rng.integers(0, 8, nf1)], 1)
        f1[0] = (w // 2, h // 2, 3)
        return [(w, h, 0, f0), (w // 2, h // 2, 1, f1)]
    entries = tmpl(40, 30, 20, 10) + tmpl(150, 140, 30, 12) + tmpl(200, 60, 12, 6) + tmpl(10, 10, 1, 1)
    templates, feats, fb = [], [], 0
    for w, h, lv, f in entries:
        templates.append((w, h, lv, fb, len(f)))
        feats.append(f)
        fb += len(f)
    bank = TemplateBank(T=[5, 8], modalities=[dict(DEFAULT_COLOR_GRADIENT)],
                        classes=[("obj", np.asarray(templates, np.int32), np.concatenate(feats).astype(np.int32))])
    sources, _ = synth.make_scene(synth.make_bank(4, modalities=("ColorGradient",), seed=3, size_range=(20.0, 40.0)), 160, 160, seed=51)
    od = o.OracleDetector(bank)
    det = Detector(bank, 160, 160)
    for thr in (30.0, 55.0, 70.0):
        same(det.match(sources, thr), od.match(sources, thr))
    det.close()
    # a level-0 feature far outside any image (x / T >= 4096 does not fit the banded feature table's column field): the context falls
    # back to the flat spread image for that level, the feature itself is skipped like upstream skips it
    far = [(w, h, lv, f.copy()) for w, h, lv, f in tmpl(40, 30, 20, 10) + tmpl(60, 50, 24, 12)]
    far[0][3][2] = (25000, 3, 1)
    templates, feats, fb = [], [], 0
    for w, h, lv, f in far:
        templates.append((w, h, lv, fb, len(f)))
        feats.append(f)
        fb += len(f)
    bank = TemplateBank(T=[5, 8], modalities=[dict(DEFAULT_COLOR_GRADIENT)],
                        classes=[("obj", np.asarray(templates, np.int32), np.concatenate(feats).astype(np.int32))])
    od = o.OracleDetector(bank)
    det = Detector(bank, 160, 160)
    for thr in (30.0, 55.0):
        ref = od.match(sources, thr)
        same(det.match(sources, thr), ref)
    assert len(ref) > 0
    det.close()


def test_shape_asserts_and_overflow():
    bank = synth.make_bank(4, seed=52, size_range=(20.0, 30.0))
    with pytest.raises(_lib.LmxError) as e:
        Detector(bank, 164, 160)                     # 164 % 5 != 0: upstream linearize CV_Assert
    assert e.value.status == _lib.LMX_ERR_SHAPE
    det = Detector(bank, 160, 160)
    with pytest.raises(_lib.LmxError) as e:
        det.match([np.zeros((160, 160, 3), np.uint8)], 80.0)      # sources.size() != modalities.size()
    assert e.value.status == _lib.LMX_ERR_SHAPE
    with pytest.raises(_lib.LmxError) as e:
        det.match([np.zeros((80, 160, 3), np.uint8), np.zeros((80, 160), np.uint16)], 80.0)
    assert e.value.status == _lib.LMX_ERR_SHAPE
    det.close()
    # an enqueue for more frames than the latest upload holds is refused (it would match stale frames of an earlier batch)
    det = Detector(synth.make_bank(4, seed=3, size_range=(20.0, 40.0)), 160, 160, max_batch=4)
    fr = [synth.make_scene(synth.make_bank(4, seed=3, size_range=(20.0, 40.0)), 160, 160, seed=52 + k)[0] for k in range(4)]
    with pytest.raises(_lib.LmxError) as e:
        det.enqueue(1, 80.0)                      # nothing uploaded yet
    assert e.value.status == _lib.LMX_ERR_INVALID_ARG
    det.upload(fr)
    det.enqueue(4, 80.0); det.collect(4)
    det.upload(fr[:2])
    det.enqueue(2, 80.0); det.collect(2)
    with pytest.raises(_lib.LmxError) as e:
        det.enqueue(3, 80.0)
    assert e.value.status == _lib.LMX_ERR_INVALID_ARG and "most recent upload holds 2" in str(e.value)
    det.close()
    # candidate capacity exceeded -> explicit overflow status, never a silent truncation
    bank = synth.make_bank(40, seed=53, size_range=(20.0, 40.0))
    sources, _ = synth.make_scene(bank, 320, 240, seed=54)
    det = Detector(bank, 320, 240, max_candidates=64)
    with pytest.raises(_lib.LmxError) as e:
        det.match(sources, 40.0)
    assert e.value.status == _lib.LMX_ERR_OVERFLOW
    det.close()


@pytest.mark.parametrize("frames", [1, 5])
def test_candidate_capacity_is_exact_with_the_striped_list(frames):
    """The candidate list is striped over up to 64 counters with a spill region (lmx_internal.hpp).  Its contract is the single
    list's: a batch succeeds -- with exactly the oracle's matches -- whenever the number of coarse candidates fits max_candidates
    * max_batch, however unevenly they fall on the stripes (one template passing at thousands of placements, capacities below the
    stripe count where everything spills), and reports LMX_ERR_OVERFLOW as soon as it does not.  One frame = the 8-stripe
    small-batch form, five frames = 64 stripes."""
    bank = synth.make_bank(6, seed=77, size_range=(20.0, 36.0))
    W, H, thr = 320, 240, 50.0
    fr = [synth.make_scene(bank, W, H, seed=78 + f, texture=1.0)[0] for f in range(frames)]
    od = o.OracleDetector(bank)
    refs = [od.match(f, thr) for f in fr]
    big = Detector(bank, W, H, max_batch=frames, max_candidates=1 << 18)
    outs = big.match_batch(fr, thr, cap=1 << 18)
    total = big.stats()["candidates"]
    big.close()
    assert total > 300 * frames               # six templates: dozens of candidates per (template, frame) wave, far more than 64 stripes x 1
    for a, b in zip(outs, refs):
        same(a, b)
    per_frame = -(-total // frames)           # capacity = max_candidates * max_batch
    for cap_pf, ok in ((per_frame, True), (per_frame - 1 if per_frame * frames - frames >= total else per_frame, True), ((total - 1) // frames, False), (7, False)):
        det = Detector(bank, W, H, max_batch=frames, max_candidates=cap_pf)
        if cap_pf * frames >= total:
            outs = det.match_batch(fr, thr, cap=1 << 18)
            for a, b in zip(outs, refs):
                same(a, b)
            assert det.stats()["candidates"] == total
        else:
            assert not ok
            with pytest.raises(_lib.LmxError) as e:
                det.match_batch(fr, thr, cap=1 << 18)
            assert e.value.status == _lib.LMX_ERR_OVERFLOW
        det.close()


def test_yaml_bank_through_readlinemod(tmp_path):
    bank = synth.make_bank(12, seed=55, size_range=(24.0, 50.0))
    p = tmp_path / "obj_templates.yml"
    NativeBank.from_bank(bank).save_yaml(p)
    sources, _ = synth.make_scene(bank, 240, 160, seed=56)
    det = Detector.readLinemod(p, 240, 160)
    assert det.numTemplates() == 12 and det.classIds() == ["obj"] and det.pyramidLevels() == 2 and det.getT(1) == 8
    same(det.match(sources, 72.0), o.OracleDetector(bank).match(sources, 72.0))
    t = det.getTemplates("obj", 5)
    assert np.array_equal(t[0][3], bank.get_templates("obj", 5)[0][3])
    det.close()


@pytest.mark.parametrize("config", ["config1_cg_3000", "config2_rgbd_3000", "config3_1280x960_two_objects"])
def test_full_size_baseline_configs(config):
    """BASELINE.json configs at full size: parity against the oracle (it finishes in well under a second per frame)
    plus size-independent properties."""
    if config == "config1_cg_3000":
        W, H, thr = 640, 480, 92.0
        bank = synth.make_bank(3000, modalities=("ColorGradient",), seed=20250214)
    elif config == "config2_rgbd_3000":
        W, H, thr = 640, 480, 92.0
        bank = synth.make_bank(3000, seed=20250215)
    else:
        # 1280x1024 violates T=5 divisibility: the 1280x960 crop with T={5,8} is used
        W, H, thr = 1280, 960, 90.0
        bank = synth.make_bank(3000, seed=20250216, classes=["memoryChip2", "cpu_binary"])   # BASELINE: 2 objects, ~6000 templates
    frames = [synth.make_scene(bank, W, H, seed=3000 + f, n_instances=6)[0] for f in range(2)]
    od = o.OracleDetector(bank)
    det = Detector(bank, W, H, max_batch=2)
    outs = det.match_batch(frames, thr)
    total = 0
    for f in range(2):
        ref = od.match(frames[f], thr)
        same(outs[f], ref)
        s, t = outs[f]["similarity"], outs[f]["template_id"]
        assert all(s[i] > s[i + 1] or (s[i] == s[i + 1] and t[i] <= t[i + 1]) for i in range(len(s) - 1))
        total += len(ref)
    assert total > 0
    again = det.match_batch(frames, thr)
    for f in range(2):
        same(again[f], outs[f])                                   # idempotent
    same(det.match(frames[1], thr), outs[1])                      # batch slot 1 == single-frame call
    det.close()


@pytest.mark.parametrize("diff_thr,dist_thr", [(50, 2000), (200, 900), (201, 2000), (5000, 70000)])
def test_depth_normal_parameters(diff_thr, dist_thr):
    """Non-default DepthNormal parameters; difference_threshold > 200 takes the 64-bit accumulation path on the GPU."""
    bank = synth.make_bank(12, modalities=("DepthNormal",), seed=57, size_range=(24.0, 60.0))
    bank.modalities[0]["difference_threshold"] = diff_thr
    bank.modalities[0]["distance_threshold"] = dist_thr
    sources, _ = synth.make_scene(bank, 240, 240, seed=58)
    d = sources[0].copy()
    d[40:80, 60:120] += 400          # a depth step larger than the default difference threshold
    d[150:170, 10:60] = 60000        # far pixels
    od = o.OracleDetector(bank)
    ref = od.match([d], 60.0)
    det = Detector(bank, 240, 240, max_candidates=1 << 17)
    got = det.match([d], 60.0)
    check_stages(det, od, 240, 240, 2, 1)
    same(got, ref)
    det.close()


def _random_normal_lut(seed):
    return np.random.default_rng(seed).choice(np.array([0, 1, 2, 4, 8, 16, 32, 64, 128], np.uint8), (20, 20, 20), p=[0.04] + [0.12] * 8)


@pytest.mark.parametrize("seed,diff_thr", [(0, 50), (1, 50), (2, 300)])
def test_pluggable_normal_lut_matches_the_oracle(seed, diff_thr):
    """NORMAL_LUT is data (lmx_bank_set_normal_lut; upstream: normal_lut.i indexed [v3][v2][v1]).  Random one-hot tables that
    depend on v3 as well: labels before/after the median, linear memories and matches equal the oracle's with the same table,
    in both accumulation widths of the device kernel; a curved depth surface exercises many (v3, v2, v1) cells, in-plane
    normals (v3 = 20) land past the table = no label."""
    bank = synth.make_bank(14, modalities=("ColorGradient", "DepthNormal"), seed=571 + seed, size_range=(24.0, 60.0))
    bank.modalities[1]["difference_threshold"] = diff_thr
    bank.normal_lut = _random_normal_lut(seed)
    sources, _ = synth.make_scene(bank, 240, 240, seed=580 + seed)
    ys, xs = np.mgrid[0:240, 0:240]
    d = sources[1].astype(np.float64) + 60 * np.sin(xs / 9.0) * np.cos(ys / 11.0) + 0.8 * ys
    d[30:60, 150:200] += 45
    d = np.clip(d, 1, 65535).astype(np.uint16)
    src = [sources[0], d]
    od = o.OracleDetector(bank)
    ref = od.match(src, 55.0)
    det = Detector(bank, 240, 240, max_candidates=1 << 17)
    got = det.match(src, 55.0)
    check_stages(det, od, 240, 240, 2, 2)
    same(got, ref)
    q = det.debug_quantized(0, 0, 1)
    assert len(np.unique(q)) >= 8                                         # the scene really spreads over the labels
    plain = o.OracleDetector(synth.make_bank(14, modalities=("ColorGradient", "DepthNormal"), seed=571 + seed, size_range=(24.0, 60.0)))
    plain.match(src, 55.0)
    assert not np.array_equal(q, plain.quantized(0, 1, (240, 240)))       # and the table, not the default rule, produced them
    det.close()


def test_pipelined_enqueue_collect():
    """Two outstanding enqueues (double-buffered outputs): results arrive oldest first; a third enqueue is refused."""
    bank = synth.make_bank(40, seed=59, size_range=(30.0, 80.0))
    frames = [synth.make_scene(bank, 320, 240, seed=60 + f)[0] for f in range(4)]
    od = o.OracleDetector(bank)
    refs = [od.match(f, 78.0) for f in frames]
    det = Detector(bank, 320, 240, max_batch=4)
    det.upload(frames)
    det.enqueue(4, 78.0)
    det.enqueue(2, 99.5)
    with pytest.raises(_lib.LmxError) as e:
        det.enqueue(4, 78.0)
    assert e.value.status == _lib.LMX_ERR_INVALID_ARG
    a = det.collect(4)
    det.enqueue(3, 78.0)
    b = det.collect(2)
    c = det.collect(3)
    for f in range(4):
        same(a[f], refs[f])
    for f in range(2):
        same(b[f], od.match(frames[f], 99.5))
    for f in range(3):
        same(c[f], refs[f])
    with pytest.raises(_lib.LmxError):
        det.collect(1)
    det.close()


def test_orientation_quantiser_exhaustive():
    """The only float stage of the colour path, over its WHOLE input domain: Sobel outputs of 8-bit images are integers
    in [-1020, 1020], so all 2041^2 gradients are compared (device fastAtan2 + round-half-even vs the oracle)."""
    v = np.arange(-1020, 1021, dtype=np.int16)
    dx, dy = np.meshgrid(v, v)
    dx, dy = np.ascontiguousarray(dx.reshape(-1)), np.ascontiguousarray(dy.reshape(-1))
    got = np.empty(dx.size, np.uint8)
    _lib.check(_lib.lib().lmx_debug_orientation_labels(0, dx.ctypes.data, dy.ctypes.data, dx.size, got.ctypes.data))
    ref = o.orientation_labels(dx, dy)
    assert got.max() == 16 and np.array_equal(got, ref)


def test_config4_one_rank_of_eight_on_a_50k_bank():
    """BASELINE config 4 (50 000 templates sharded over 8 GPUs, 6250 per rank), the part one GPU can run: rank 3's context
    must emit exactly the oracle's pre-sort records whose template ids fall into its shard (size-independent property:
    shard outputs partition the whole-bank output)."""
    from golden_util import bank_50k
    bank = bank_50k()
    frames = [synth.make_scene(bank, 640, 480, seed=3100 + f)[0] for f in range(2)]
    rank, world = 3, 8
    b, e = bank.shard(rank, world)["obj"]
    assert e - b == 6250
    od = o.OracleDetector(bank)
    det = Detector(bank, 640, 480, max_batch=2, shard_rank=rank, shard_world=world)
    det.upload(frames)
    det.enqueue(2, 92.0)
    got = det.collect(2)             # this rank's matches, sorted/uniqued among themselves
    hip = C.CDLL("libamdhip64.so")
    for f in range(2):
        od.match(frames[f], 92.0)
        raw = od.last_raw()
        mine = raw[(raw["template_id"] >= b) & (raw["template_id"] < e)]
        same(got[f], merge_raw(mine))
    det.close()


@pytest.mark.parametrize("mono", [False, True])
def test_upload_raw_node_side_preprocessing(mono):
    """SURVEY 8f row 4 on the device: raw 752x480 frame (+ float-metre depth) -> (MONO8->BGR) + GaussianBlur 3x3 on the full
    frame + crop Rect(56, 0, 640, 480) + depth * 1000 -> u16, then the usual path.  Checked against the oracle's restatement
    of the reference's detect_cb steps (src/linemod_ensenso_detect_3_mult_detect_service.cpp:293-326, 837-858)."""
    W, H, SW, SH, bias_x = 640, 480, 752, 480, 56
    bank = synth.make_bank(150, seed=63)
    det = Detector(bank, W, H, max_batch=2)
    od = o.OracleDetector(bank)
    raw_frames, ref_sources = [], []
    for f in range(2):
        (bgr, depth), _ = synth.make_scene(bank, SW, SH, seed=64 + f, texture=0.8)
        color = np.ascontiguousarray(bgr[:, :, 1]) if mono else np.ascontiguousarray(bgr)
        z = depth.astype(np.float32) / np.float32(1000.0)
        z[depth == 0] = np.nan                       # what an organised cloud holds at invalid points
        z[5, 100:110] = np.inf
        raw_frames.append([color, z])
        ref_sources.append([o.pre_color(color, (bias_x, 0), (W, H), True), o.pre_depth(z, (bias_x, 0), (W, H))])
    det.upload_raw(raw_frames, (SW, SH), (bias_x, 0), blur3=True, mono=mono, depth_float_m=True)
    det.enqueue(2, 85.0)
    outs = det.collect(2)
    for f in range(2):
        assert np.array_equal(det.debug_pyramid_bgr(f, 0, 0), ref_sources[f][0])
        assert np.array_equal(det.debug_depth(f, 1), ref_sources[f][1])
        ref = od.match(ref_sources[f], 85.0)
        assert len(ref) > 0
        same(outs[f], ref)
    # frame-sized u16 depth next to a full-size colour frame, no blur
    (bgr, depth), _ = synth.make_scene(bank, SW, SH, seed=70)
    color = np.ascontiguousarray(bgr[:, :, 1]) if mono else np.ascontiguousarray(bgr)
    d_crop = np.ascontiguousarray(depth[:, bias_x:bias_x + W])
    det.upload_raw([[color, d_crop]], (SW, SH), (bias_x, 0), blur3=False, mono=mono, depth_float_m=False)
    det.enqueue(1, 85.0)
    got = det.collect(1)[0]
    same(got, od.match([o.pre_color(color, (bias_x, 0), (W, H), False), d_crop], 85.0))
    with pytest.raises(_lib.LmxError) as e:
        det.upload_raw([[color, d_crop]], (SW, SH), (200, 0), blur3=False, mono=mono)   # crop leaves the frame
    assert e.value.status == _lib.LMX_ERR_SHAPE
    det.close()


def test_add_template_trainer_parity(tmp_path):
    """lmx_bank_add_template (device quantisation + host feature selection) builds the same template
    pyramids, ids and bounding boxes as the oracle's restatement of Detector::addTemplate; failures return -1 and add
    nothing; a bank trained this way, written to YAML and read back, finds its own training views."""
    import train_util
    from linemod_pose_estimation_amd.bank import DEFAULT_DEPTH_NORMAL
    for mods in (["ColorGradient", "DepthNormal"], ["ColorGradient"]):
        mdesc = [dict(DEFAULT_COLOR_GRADIENT) if m == "ColorGradient" else dict(DEFAULT_DEPTH_NORMAL) for m in mods]
        empty = TemplateBank(T=[5, 8], modalities=mdesc)
        od = o.OracleDetector(empty)
        nb = NativeBank.create([5, 8], mdesc)
        if len(mods) == 2:
            # the trainer quantises normals with the bank's NORMAL_LUT too (a v3-dependent stand-in for upstream's normal_lut.i)
            dflt = o.normal_lut().astype(np.uint16)
            lut = np.zeros((20, 20, 20), np.uint8)
            for v3 in range(20):       # the default octants, rotated by one bin every four v3 planes
                r = (v3 // 4) % 8
                lut[v3] = ((dflt[v3] << r) | (dflt[v3] >> (8 - r))) & 0xff
            od.set_normal_lut(lut)
            nb.set_normal_lut(lut)
        views = []
        for seed in (91, 93, 95, 97, 99):
            v = train_util.rendered_view(seed)
            if v is None:
                continue
            bgr, depth, mask = v
            src = [bgr, depth][:len(mods)]
            use_mask = mask if seed != 95 else None          # one view without a mask (upstream allows an empty mask)
            ref_tid, ref_bb = od.add_template(src, "obj", use_mask)
            got_tid, got_bb = nb.add_template(src, "obj", use_mask)
            assert got_tid == ref_tid and (ref_tid < 0 or got_bb == ref_bb), (seed, got_tid, ref_tid, got_bb, ref_bb)
            if ref_tid >= 0:
                views.append((src, ref_bb))
        tiny = np.zeros((240, 320), np.uint8)
        tiny[50:54, 60:64] = 255
        assert nb.add_template(views[0][0], "obj", tiny)[0] == -1 == od.add_template(views[0][0], "obj", tiny)[0]
        trained = nb.to_bank()
        n = trained.num_templates("obj")
        assert n == len(views) >= 3
        for tid in range(n):
            for (w, h, lvl, f), (rw, rh, rl, rf) in zip(trained.get_templates("obj", tid), od.get_templates("obj", tid)):
                assert (w, h, lvl) == (rw, rh, rl) and np.array_equal(f, rf)
        # persist like writeLinemod, reload like readLinemod, detect the training views
        yml = tmp_path / ("trained_%d.yml" % len(mods))
        nb.save_yaml(yml)
        det = Detector.readLinemod(yml, 320, 240)
        for src, bb in views[:2]:
            m = det.match(src, 90.0)
            same(m, od.match(src, 90.0))
            assert len(m) > 0 and m["similarity"][0] > 95.0
        det.close()


def test_hipgraph_replay_matches_eager():
    """LMX_CTX_HIPGRAPH: the captured chain replays with identical results across thresholds, batch sizes, both output slots."""
    bank = synth.make_bank(60, seed=65, size_range=(30.0, 80.0))
    frames = [synth.make_scene(bank, 320, 240, seed=66 + f)[0] for f in range(3)]
    od = o.OracleDetector(bank)
    det = Detector(bank, 320, 240, max_batch=3, hipgraph=True)
    det.upload(frames)
    for rep in range(3):
        for thr, n in ((78.0, 3), (85.0, 2), (78.0, 3)):
            det.enqueue(n, thr)
            det.enqueue(n, thr)           # second slot -> its own graph
            a = det.collect(n)
            b = det.collect(n)
            for f in range(n):
                ref = od.match(frames[f], thr)
                same(a[f], ref)
                same(b[f], ref)
    same(det.match(frames[1], 78.0), od.match(frames[1], 78.0))   # lmx_match goes through the same graph path
    det.close()


def test_config5_hipgraph_lanes_64_frames_on_one_rank_of_the_50k_bank():
    """BASELINE configs[4], the part one GPU runs: 64 concurrent 640x480 frames x the 6250-template shard (rank 3 of 8) of the
    50 000-template bank, the per-batch chain captured as hipGraphs and replayed on all device lanes at once
    (LMX_CTX_HIPGRAPH | LMX_CTX_OVERLAP), pipelined to the context's depth.  Every frame of every step must equal the oracle's
    pre-sort records filtered to the shard (shard outputs partition the whole-bank output), merged like the multi-GPU job does.
    Round 1 had forbidden this flag combination after "incomplete read-backs"; those came from the hipMemsetAsync /
    hipMemcpyAsync(DeviceToHost) nodes the captured chain contained at the time: the chain is kernels only
    now (header clear inside the first kernel, read-back by k_publish_records)."""
    from golden_util import bank_50k
    bank = bank_50k()
    B = 64
    frames = [synth.make_scene(bank, 640, 480, seed=3100 + f)[0] for f in range(B)]
    rank, world = 3, 8
    b, e = bank.shard(rank, world)["obj"]
    assert e - b == 6250
    od = o.OracleDetector(bank)
    refs = []
    for f in range(B):
        od.match(frames[f], 92.0)
        raw = od.last_raw()
        refs.append(merge_raw(raw[(raw["template_id"] >= b) & (raw["template_id"] < e)]))
    assert sum(len(r) for r in refs) > 8
    det = Detector(bank, 640, 480, max_batch=B, shard_rank=rank, shard_world=world, hipgraph=True, overlap=True)
    assert det.max_outstanding >= 4
    det.upload(frames)
    inflight = 0
    for step in range(3 * det.max_outstanding):      # every slot's graph is captured once and replayed twice
        if inflight == det.max_outstanding:
            got = det.collect(B)
            inflight -= 1
            for f in range(B):
                same(got[f], refs[f])
        det.enqueue(B, 92.0)
        inflight += 1
    while inflight:
        got = det.collect(B)
        inflight -= 1
        for f in range(B):
            same(got[f], refs[f])
    det.close()


@pytest.mark.parametrize("mode", ["pageable", "pageable_roi_graph", "pinned_async"])
def test_host_frames_pipelined_fresh_frames_every_step(mode):
    """The reference's boundary hands match() host images on every call (..._service.cpp:324-344).  Pipelined form of that:
    every step uploads DIFFERENT frames (upload of step i+1 is queued while the kernels of earlier steps run, frame sets rotate)
    and every frame of every step must equal the oracle.  pageable: plain numpy memory through the threaded staging copy;
    pageable_roi_graph: strided ROI views (row stride 752*3 like the ensenso crop) and the chain replayed as hipGraphs (one per
    slot and frame set); pinned_async: frames in pinned memory, DMA straight from the caller's buffer, LMX_CTX_ASYNC_INPUT."""
    from linemod_pose_estimation_amd import PinnedArena
    bank = synth.make_bank(80, seed=131, size_range=(30.0, 80.0))
    B, n_batches = 6, 5
    row_pad = 112 if mode == "pageable_roi_graph" else 0
    batches = [[synth.make_scene(bank, 320, 240, seed=1400 + 10 * k + f, row_pad=row_pad)[0] for f in range(B)] for k in range(n_batches)]
    od = o.OracleDetector(bank)
    refs = [[od.match(fr, 80.0) for fr in batch] for batch in batches]
    assert len({len(r) for batch in refs for r in batch}) > 3          # the batches really differ
    arena = None
    if mode == "pinned_async":
        arena = PinnedArena(n_batches * B * (320 * 240 * 5 + 1024))
        batches = [[[arena.put(np.ascontiguousarray(src)) for src in fr] for fr in batch] for batch in batches]
    det = Detector(bank, 320, 240, max_batch=B, overlap=True, hipgraph=(mode == "pageable_roi_graph"), async_input=(mode == "pinned_async"))
    pending = []
    for step in range(4 * n_batches):
        k = (step * 3) % n_batches                  # not the rotation period of the frame sets
        if len(pending) == det.max_outstanding:
            kk = pending.pop(0)
            got = det.collect(B)
            for f in range(B):
                same(got[f], refs[kk][f])
        det.upload(batches[k])
        det.enqueue(B, 80.0)
        pending.append(k)
    while pending:
        kk = pending.pop(0)
        got = det.collect(B)
        for f in range(B):
            same(got[f], refs[kk][f])
    # the synchronous wrappers ride the same path
    for k in (1, 3):
        outs = det.match_batch(batches[k], 80.0)
        for f in range(B):
            same(outs[f], refs[k][f])
        same(det.match(batches[k][2], 80.0), refs[k][2])
    od.match(batches[3][2], 80.0)
    check_stages(det, od, 320, 240, 2, 2, frame=0)   # the last device call was match(batches[3][2])
    det.close()
    if arena is not None:
        arena.close()


def test_overlapped_lanes_give_identical_results():
    """LMX_CTX_OVERLAP: the two output slots run on two streams with their own intermediate buffers.  Different thresholds,
    batch sizes and a class-filter change in flight, re-uploads between rounds, stage read-back after an enqueue on either
    lane, the gather block of an enqueue that ran on lane 1: everything equals the oracle / the single-lane context."""
    bank = synth.make_bank(50, seed=71, size_range=(30.0, 80.0))
    od = o.OracleDetector(bank)
    det = Detector(bank, 320, 240, max_batch=4, overlap=True)
    plain = Detector(bank, 320, 240, max_batch=4)
    for rnd in range(3):
        frames = [synth.make_scene(bank, 320, 240, seed=700 + 10 * rnd + f)[0] for f in range(4)]
        det.upload(frames)
        plain.upload(frames)
        plan = [(4, 76.0), (3, 88.0), (2, 76.0), (1, 99.0), (4, 88.0), (2, 99.0), (3, 76.0), (1, 88.0)][:det.max_outstanding]
        assert len(plan) == det.max_outstanding >= 4
        for n, thr in plan:       # slots alternate between the lanes: all of them run concurrently, two deep
            det.enqueue(n, thr)
        with pytest.raises(_lib.LmxError):
            det.enqueue(1, 76.0)  # every slot is outstanding
        results = [det.collect(n) for n, _ in plan]   # oldest first
        for (n, thr), res in zip(plan, results):
            for f in range(n):
                same(res[f], od.match(frames[f], thr))
        a = results[0]
        # the most recent enqueue ran on lane 0; one more puts the view on lane 1: stage buffers of both lanes are complete
        det.enqueue(4, 76.0)
        det.enqueue(4, 76.0)      # lane 1 is the most recent
        det.collect(4)
        det.collect(4)
        plain.enqueue(4, 76.0)
        plain.collect(4)
        for lvl in range(2):
            for m in range(2):
                assert np.array_equal(det.debug_quantized(2, lvl, m), plain.debug_quantized(2, lvl, m))
            assert np.array_equal(det.debug_pyramid_bgr(3, lvl, 0), plain.debug_pyramid_bgr(3, lvl, 0))
        for m in range(2):
            assert np.array_equal(det.debug_linear_memory(1, 1, m), plain.debug_linear_memory(1, 1, m))
    # gather block of an enqueue that ran on lane 1 (export is ordered behind it on lane 0's stream)
    import torch
    det.enqueue(4, 76.0)
    det.collect(4)
    det.enqueue(4, 76.0)          # lane 1
    blk = torch.zeros(64 + 4096 * 32, dtype=torch.uint8, device="cuda")
    det.export_raw(blk.data_ptr(), 4096)
    det.sync()
    plain.enqueue(4, 76.0)
    blk2 = torch.zeros(64 + 4096 * 32, dtype=torch.uint8, device="cuda")
    plain.export_raw(blk2.data_ptr(), 4096)
    plain.sync()
    h1, h2 = blk.cpu().numpy(), blk2.cpu().numpy()
    n1, n2 = int(h1[4:8].view(np.uint32)[0]), int(h2[4:8].view(np.uint32)[0])
    assert n1 == n2 and n1 > 0
    r1 = np.sort(h1[64:64 + 32 * n1].view(RAW_MATCH_DTYPE), order=["frame", "order_key"])
    r2 = np.sort(h2[64:64 + 32 * n2].view(RAW_MATCH_DTYPE), order=["frame", "order_key"])
    assert np.array_equal(r1, r2)
    det.close()
    plain.close()


@pytest.mark.parametrize("mods,nfeat,levels", [
    (("ColorGradient",), 63, (8,)),                      # 63 features at the only level: byte sums up to 252
    (("ColorGradient", "DepthNormal"), 31, (8,)),        # 62 in total -> u8 kernel
    (("ColorGradient", "DepthNormal"), 32, (8,)),        # 64 in total -> generic kernel
    (("DepthNormal", "ColorGradient"), 63, (4, 8)),      # 31 + 31 at the coarsest level, modalities swapped
    (("ColorGradient", "DepthNormal"), 20, (5, 8)),      # 10 + 10: not a multiple of the group size
])
def test_both_scoring_kernels_agree_with_the_oracle(mods, nfeat, levels, monkeypatch):
    """The three scoring kernels -- k_score_coarse_sb (default for templates with <= 63 coarsest-level features: feature table in
    16-dword scalar blocks, every group sharing its funnel shift, leftovers padded with zero-run entries), k_score_coarse_u8 (its
    predecessor: table broadcast with v_readlane) and the generic k_score_coarse -- must each reproduce the oracle, candidate
    counts included; LMX_SCORE_KERNEL selects one."""
    bank = synth.make_bank(60, modalities=mods, T=levels, seed=81, num_features=nfeat, size_range=(30.0, 80.0))
    sources, _ = synth.make_scene(bank, 320, 240, seed=82)
    od = o.OracleDetector(bank)
    names = set()
    for variant, no_prune in (("sb", False), ("u8", False), ("generic", False), ("sb", True), ("u8", True), ("generic", True)):
        monkeypatch.setenv("LMX_SCORE_KERNEL", variant)      # read when the context is created
        if no_prune:
            # LMX_SCORE_NO_PRUNE: the exact early exits compiled out, similarity()'s full work -- identical candidates
            monkeypatch.setenv("LMX_SCORE_NO_PRUNE", "1")
        det = Detector(bank, 320, 240, max_candidates=1 << 18)
        monkeypatch.delenv("LMX_SCORE_KERNEL", raising=False)
        monkeypatch.delenv("LMX_SCORE_NO_PRUNE", raising=False)
        for thr in (55.0, 80.0, 92.0):
            ref = od.match(sources, thr)
            same(det.match(sources, thr), ref)
            assert det.stats()["candidates"] == od.last_candidates()
        names.add(det.device_kernel_name("k_score_coarse"))
        det.close()
    total = nfeat * len(mods) // (2 ** (len(levels) - 1))
    assert names == ({"k_score_coarse_sb", "k_score_coarse_u8", "k_score_coarse"} if total <= 63 else {"k_score_coarse"})


@pytest.mark.parametrize("n_frames", [8, 11, 16, 19])
def test_batches_of_eight_frames_and_more(n_frames):
    """From 8 frames per batch on the scoring and spread kernels place each frame's workgroups on one XCD (frame % 8), with a
    ragged last group when the batch is not a multiple of 8: every frame's stages and matches against the oracle, for sub-batches too."""
    bank = synth.make_bank(60, seed=97, size_range=(30.0, 80.0))
    frames = [synth.make_scene(bank, 320, 240, seed=980 + f)[0] for f in range(n_frames)]
    od = o.OracleDetector(bank)
    det = Detector(bank, 320, 240, max_batch=n_frames)
    det.upload(frames)
    for n in sorted({n_frames, max(8, n_frames - 3)}):
        det.enqueue(n, 78.0)
        got = det.collect(n)
        for f in range(n):
            ref = od.match(frames[f], 78.0)
            same(got[f], ref)
            if f in (0, 7, n - 1):
                check_stages(det, od, 320, 240, 2, 2, frame=f)
    det.close()


def test_bench_workload_every_frame_against_the_oracle():
    """Exactly what bench.py times (BASELINE configs[1]: 640x480 RGB-D, 3000 templates, T = {5, 8}, threshold 92, 64 resident
    frames, device lanes, pipelined to the context's depth): the matches of every frame of every step equal the oracle's."""
    bank = synth.make_bank(3000, modalities=("ColorGradient", "DepthNormal"), T=(5, 8), seed=20250215)
    frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(64)]
    od = o.OracleDetector(bank)
    refs = [od.match(f, 92.0) for f in frames]
    assert sum(len(r) for r in refs) > 64
    det = Detector(bank, 640, 480, max_batch=64, overlap=True)
    det.upload(frames)
    inflight = 0
    for step in range(2 * det.max_outstanding + 1):
        if inflight == det.max_outstanding:
            got = det.collect(64)
            inflight -= 1
            for f in range(64):
                same(got[f], refs[f])
        det.enqueue(64, 92.0)
        inflight += 1
    while inflight:
        got = det.collect(64)
        inflight -= 1
        for f in range(64):
            same(got[f], refs[f])
    det.close()


def test_lanes_soak_every_step_identical():
    """300 pipelined steps over the device lanes with changing batch sizes, thresholds and re-uploads: every single result is
    compared with the oracle's (computed once per distinct request), so a rare race between lanes would show."""
    bank = synth.make_bank(60, seed=95, size_range=(30.0, 80.0))
    sets = [[synth.make_scene(bank, 320, 240, seed=960 + 10 * k + f)[0] for f in range(4)] for k in range(2)]
    od = o.OracleDetector(bank)
    expect = {}
    def ref(k, f, thr):
        if (k, f, thr) not in expect:
            expect[(k, f, thr)] = od.match(sets[k][f], thr)
        return expect[(k, f, thr)]
    det = Detector(bank, 320, 240, max_batch=4, overlap=True)
    rng = np.random.default_rng(7)
    pending = []
    k = 0
    det.upload(sets[k])
    for step in range(300):
        if step in (100, 200):            # re-upload: waits for every lane, then the other frame set is resident
            while pending:
                n, thr, kk = pending.pop(0)
                res = det.collect(n)
                for f in range(n):
                    same(res[f], ref(kk, f, thr))
            k ^= 1
            det.upload(sets[k])
        if len(pending) == det.max_outstanding:
            n, thr, kk = pending.pop(0)
            res = det.collect(n)
            for f in range(n):
                same(res[f], ref(kk, f, thr))
        n, thr = int(rng.integers(1, 5)), float(rng.choice([74.0, 82.0, 90.0]))
        det.enqueue(n, thr)
        pending.append((n, thr, k))
    while pending:
        n, thr, kk = pending.pop(0)
        res = det.collect(n)
        for f in range(n):
            same(res[f], ref(kk, f, thr))
    det.close()


def test_release_and_block_copy_entry_points():
    """lmx_ctx_release frees the oldest slot without a read-back (error when nothing is outstanding); lmx_stream_copy_blocks
    copies, per gather block, the header and exactly the records it counts (and nothing else) into another buffer."""
    import torch
    bank = synth.make_bank(40, seed=91, size_range=(30.0, 80.0))
    frames = [synth.make_scene(bank, 320, 240, seed=92 + f)[0] for f in range(2)]
    det = Detector(bank, 320, 240, max_batch=2)
    assert det.max_outstanding == 2
    with pytest.raises(_lib.LmxError):
        det.release()
    det.upload(frames)
    cap = 8192
    blk_bytes = 64 + cap * 32
    src = torch.zeros(2 * blk_bytes, dtype=torch.uint8, device="cuda")
    for r, thr in enumerate((75.0, 88.0)):      # two "ranks": the same context at two thresholds
        det.enqueue(2, thr)
        det.export_raw(src[r * blk_bytes:].data_ptr(), cap)
        det.release()
    det.sync()
    host = torch.full((2 * blk_bytes,), 0xAB, dtype=torch.uint8).pin_memory()
    _lib.check(_lib.lib().lmx_stream_copy_blocks(host.data_ptr(), src.data_ptr(), 2, blk_bytes, cap, None))
    torch.cuda.synchronize()
    h, d = host.numpy(), src.cpu().numpy()
    counts = []
    for r in range(2):
        n = int(d[r * blk_bytes + 4:r * blk_bytes + 8].view(np.uint32)[0])
        counts.append(n)
        assert n <= cap
        used = 64 + 32 * n
        assert np.array_equal(h[r * blk_bytes:r * blk_bytes + used], d[r * blk_bytes:r * blk_bytes + used])
        assert (h[r * blk_bytes + used:(r + 1) * blk_bytes] == 0xAB).all()      # untouched
    assert max(counts) > 0 and counts[0] >= counts[1]
    with pytest.raises(_lib.LmxError):
        _lib.check(_lib.lib().lmx_stream_copy_blocks(host.data_ptr() + 4, src.data_ptr(), 2, blk_bytes, cap, None))   # misaligned
    det.close()


def test_three_modalities():
    """More than two modalities (upstream's addSimilarities keeps adding u8 maps into the u16 total)."""
    bank = synth.make_bank(30, modalities=("ColorGradient", "DepthNormal", "ColorGradient"), seed=67, size_range=(30.0, 80.0))
    sources, _ = synth.make_scene(bank, 320, 240, seed=68)
    assert len(sources) == 3
    od = o.OracleDetector(bank)
    det = Detector(bank, 320, 240)
    for thr in (70.0, 82.0):
        same(det.match(sources, thr), od.match(sources, thr))
    check_stages(det, od, 320, 240, 2, 3)
    det.close()


def _random_bank(rng, T, mods, n_templates, W, H):
    """Banks with arbitrary (not contour-like) features: random positions incl. x == width / y == height, random labels,
    random feature counts 1..63, template boxes up to the image size."""
    from linemod_pose_estimation_amd.bank import DEFAULT_DEPTH_NORMAL
    L, M = len(T), len(mods)
    templates, feats, fb = [], [], 0
    for _ in range(n_templates):
        w0 = int(rng.integers(8, max(9, W // 2)))
        h0 = int(rng.integers(8, max(9, H // 2)))
        for l in range(L):
            w, h = w0 >> l, h0 >> l
            for m in range(M):
                nf = int(rng.integers(1, 64 >> l if (64 >> l) > 1 else 2))
                f = np.stack([rng.integers(0, w + 1, nf), rng.integers(0, h + 1, nf), rng.integers(0, 8, nf)], 1).astype(np.int32)
                templates.append((w, h, l, fb, nf))
                feats.append(f)
                fb += nf
    mdesc = [dict(DEFAULT_COLOR_GRADIENT) if m == "ColorGradient" else dict(DEFAULT_DEPTH_NORMAL) for m in mods]
    return TemplateBank(T=list(T), modalities=mdesc, classes=[("obj", np.asarray(templates, np.int32), np.concatenate(feats))])


@pytest.mark.parametrize("seed", range(12))
def test_randomized_configurations(seed):
    """Seeded random geometry (sizes, T per level, modality mix, row padding), random banks and thresholds: every path of the
    generic code (odd cell counts, T without a fast kernel, out-of-image features, multi-pass scoring) against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    Ts = [(5, 8), (4, 8), (8,), (5,), (2, 4), (3, 6), (6, 4), (4, 8, 8), (7, 5), (10, 8)][seed % 10]
    L = len(Ts)
    unit = int(np.lcm.reduce([t << l for l, t in enumerate(Ts)]))
    while (unit * unit) % (16 << (2 * (L - 1))) or unit < 8:
        unit *= 2
    W = unit * int(rng.integers(max(1, 160 // unit), max(2, 420 // unit) + 1))
    H = unit * int(rng.integers(max(1, 160 // unit), max(2, 420 // unit) + 1))
    mods = [("ColorGradient",), ("ColorGradient", "DepthNormal"), ("DepthNormal",), ("DepthNormal", "ColorGradient")][int(rng.integers(0, 4))]
    bank = _random_bank(rng, Ts, mods, int(rng.integers(5, 40)), W, H)
    scene_bank = synth.make_bank(6, modalities=mods, T=(5, 8), seed=2000 + seed, size_range=(20.0, min(W, H) / 2.5))
    sources, _ = synth.make_scene(scene_bank, W, H, seed=3000 + seed, row_pad=int(rng.integers(0, 3)) * 4, texture=float(rng.uniform(0.3, 1.2)))
    od = o.OracleDetector(bank)
    det = Detector(bank, W, H, max_candidates=1 << 19)
    for thr in (float(rng.uniform(40, 60)), float(rng.uniform(60, 90))):
        ref = od.match(sources, thr)
        got = det.match(sources, thr, cap=1 << 17)
        assert det.stats()["candidates"] == od.last_candidates(), (Ts, W, H, mods, thr)
        same(got, ref)
    check_stages(det, od, W, H, L, len(mods))
    det.close()


@pytest.mark.parametrize("thr,step,cthr", [(75.0, 10, 2), (80.0, 16, 0), (70.0, 8, 3), (88.0, 10, 2), (99.9, 10, 2)])
def test_device_side_finalise_and_cluster_chain(thr, step, cthr):
    """SURVEY 8f row 2 ON THE DEVICE (lmx_ctx_collect_clusters, csrc/lmx_f2.hip): std::sort + std::unique of Detector::match and the
    reference's rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU (src/rgbdDetector.cpp:36-144,
    462-574) run in one kernel on the raw-match slot.  Compared per frame with the oracle: its final match list (which fixes
    libstdc++'s order of ties that std::unique and the greedy NMS observe) and its restatement of the reference's own functions
    on that list.  Match lists of several hundred to two thousand records with many ties; one frame beyond the LDS path's 2048
    records takes the host fallback inside the same call."""
    bank = synth.make_bank(400, seed=81)
    frames = [synth.make_scene(bank, 640, 480, seed=82 + f, n_instances=6)[0] for f in range(3)]
    rng = np.random.default_rng(5)
    n_t = 400
    dists = 0.5 + 0.1 * (np.arange(n_t) % 6) + rng.uniform(-0.005, 0.005, n_t)       # six distance rings like the renderer
    rects = np.stack([np.zeros(n_t), np.zeros(n_t), [m["width"] for m in bank.meta["obj"]], [m["height"] for m in bank.meta["obj"]]], 1).astype(np.int32)
    od = o.OracleDetector(bank)
    det = Detector(bank, 640, 480, max_batch=3, max_candidates=1 << 18)
    det.set_cluster_sidecar(dists, rects, step, 0.5, 0.1, cthr)
    det.upload(frames)
    det.enqueue(3, thr)
    got = det.collect_clusters(3, cap_total=1 << 18)
    sizes = []
    for f in range(3):
        ref_m = od.match(frames[f], thr)
        sizes.append(len(od.last_raw()))
        ref_c, ref_mem = o.cluster_matches(ref_m, dists, rects, step, 0.5, 0.1, cthr)
        m, c, mem = got[f]
        same(m, ref_m)
        assert len(c) == len(ref_c), (f, len(c), len(ref_c))
        for k in ("index", "rect", "score", "member_count"):
            assert np.array_equal(c[k], ref_c[k]), (f, k)
        for a, b in zip(c, ref_c):
            assert np.array_equal(mem[a["member_begin"]:a["member_begin"] + a["member_count"]], ref_mem[b["member_begin"]:b["member_begin"] + b["member_count"]])
    if thr <= 75.0:
        assert max(sizes) > 2048 > min(sizes) or max(sizes) > 500       # both the LDS path and (at the lowest thresholds) the host fallback ran
    # the plain collect of the same enqueue agrees too
    det.enqueue(3, thr)
    plain = det.collect(3, cap_total=1 << 18)
    for f in range(3):
        same(plain[f], got[f][0])
    det.close()


def test_randomised_configurations_sweep():
    """scripts/fuzz_parity.py, 40 draws: sizes, pyramid depths, T, modality sets, feature counts, thresholds, batch sizes, row strides,
    lanes and hipGraph at random (seeded); stages and matches against the oracle.  (600 draws were run at the end of round 2.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "40", "31"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "fuzz ok: 40 configurations" in res.stdout


def test_mesh_rendered_bank_full_size_against_the_oracle():
    """The realistic bank ("what's missing" 3): 2652 templates = the reference's view grid (26 directions x 6 distances x 17
    in-plane rotations, config/data/..._renderer_params.yml) rendered from its own memoryChip2.stl and trained by addTemplate
    (tests/golden/mesh_bank_memoryChip2.npz).  Neighbouring templates are neighbouring views of ONE object, and the scenes contain
    rendered chips: many templates respond to every instance.  HIP == oracle on matches, order, candidate counts; every planted
    instance is found by (at least) its own training view at the planted position."""
    from linemod_pose_estimation_amd import meshsynth as ms
    bank, rects, dists, views_idx = ms.load_bank("memoryChip2")
    assert bank.num_templates() == 2652
    chip, cpu, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
    scenes = [ms.make_scene(chip, views, seed=40 + f, n_instances=3, other_tri=cpu, n_other=2) for f in range(3)]
    od = o.OracleDetector(bank)
    det = Detector(bank, 640, 480, max_batch=3, max_candidates=1 << 17)
    for thr in (90.0, 80.0):
        outs = det.match_batch([s for s, _ in scenes], thr, cap=1 << 16)
        for f, (src, truth) in enumerate(scenes):
            ref = od.match(src, thr)
            same(outs[f], ref)
            assert len(ref) > 5
            for t in truth:       # the planted pose: template id = view index (every view was accepted by the trainer)
                tid = int(np.nonzero(views_idx == t["view"])[0][0])
                # a match reports the corner of the FEATURE bounding box (cropTemplates), a few pixels inside the silhouette's
                hit = ref[(ref["template_id"] == tid) & (np.abs(ref["x"] - t["x"]) <= 12) & (np.abs(ref["y"] - t["y"]) <= 12)]
                assert len(hit) and hit["similarity"].max() >= 90.0, (f, t, thr)
    assert det.stats()["candidates"] > 3 * 1000      # threshold 80: thousands of neighbouring views pass the coarse level per frame
    det.close()


def test_two_object_mesh_banks_config3_against_the_oracle():
    """BASELINE configs[2] with rendered banks: the two objects it names (memoryChip2 + cpu_binary, 2 x 2652 templates trained from the
    reference's meshes over its view grid) as two classes of one detector, 1280x960 scenes (the T = 5 crop of 1280x1024) holding rendered
    instances of BOTH; HIP == oracle on matches and order, and every planted instance of either class is reported by its own view."""
    from linemod_pose_estimation_amd import meshsynth as ms
    bank, side = ms.load_banks(("memoryChip2", "cpu_binary"))
    assert bank.num_templates() == 2 * 2652 and [c[0] for c in bank.classes] == ["memoryChip2", "cpu_binary"]
    chip, cpu, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
    W, H, thr = 1280, 960, 90.0
    scenes = [ms.make_scene(chip, views, W, H, seed=60 + f, n_instances=4, other_tri=cpu, n_other=3, other_class="cpu_binary") for f in range(2)]
    od = o.OracleDetector(bank)
    det = Detector(bank, W, H, max_batch=2, max_candidates=1 << 17)
    outs = det.match_batch([s for s, _ in scenes], thr, cap=1 << 16)
    for f, (src, truth) in enumerate(scenes):
        ref = od.match(src, thr)
        same(outs[f], ref)
        assert {t["class"] for t in truth} == {"obj", "cpu_binary"}
        for t in truth:
            name = "memoryChip2" if t["class"] == "obj" else "cpu_binary"
            ci = sorted(c[0] for c in bank.classes).index(name)      # class_index follows upstream's std::map order of the class ids
            tid = int(np.nonzero(side[name][2] == t["view"])[0][0])
            hit = ref[(ref["class_index"] == ci) & (ref["template_id"] == tid) & (np.abs(ref["x"] - t["x"]) <= 12) & (np.abs(ref["y"] - t["y"]) <= 12)]
            assert len(hit) and hit["similarity"].max() >= 90.0, (f, t)
    same(det.match(scenes[1][0], thr, cap=1 << 16), outs[1])
    det.close()


def test_hip_trainer_on_mesh_renders_equals_the_committed_bank():
    """addTemplate on the device (lmx_bank_add_template) over rendered views of the reference's mesh == the bank the oracle's trainer
    produced for the same views (the committed fixture): widths, heights, every feature, for views spread over the whole grid."""
    from linemod_pose_estimation_amd import meshsynth as ms
    bank, rects, dists, views_idx = ms.load_bank("memoryChip2")
    chip, views = ms.load_mesh("memoryChip2"), ms.view_grid()
    nb = NativeBank.create(bank.T, bank.modalities)
    picks = list(range(0, 2652, 45))
    for i in picks:
        bgr, depth, mask, rect = ms.training_view(chip, *views[i])
        tid, bb = nb.add_template([bgr, depth], "obj", mask)
        assert tid == picks.index(i)
        assert tuple(rects[i]) == rect
    got = nb.to_bank()
    for k, i in enumerate(picks):
        for a, b in zip(got.get_templates("obj", k), bank.get_templates("obj", i)):
            assert a[:3] == b[:3] and np.array_equal(a[3], b[3]), (i, a[:3], b[:3])


def test_match_with_masks_equals_the_oracle():
    """Detector::match's sixth argument ("what's missing" 4): per-modality masks, halved per pyramid level with INTER_NEAREST,
    labels copied through them.  Random blocky masks, a modality without a mask next to a masked one, full masks (== no masks), batches
    where only some frames carry masks, and masks dropped again by the next upload; label images and matches against the oracle."""
    W, H = 320, 240
    bank = synth.make_bank(40, seed=301, size_range=(30.0, 70.0))
    od = o.OracleDetector(bank)
    det = Detector(bank, W, H, max_batch=3)
    rng = np.random.default_rng(9)

    def blocky(p):
        m = (rng.uniform(0, 1, (H // 16, W // 16)) < p).astype(np.uint8) * rng.integers(1, 255, (H // 16, W // 16), dtype=np.uint8)
        return np.ascontiguousarray(np.kron(m, np.ones((16, 16), np.uint8)))
    frames = [synth.make_scene(bank, W, H, seed=302 + f)[0] for f in range(3)]
    for case in range(6):
        masks = [blocky(0.7), blocky(0.8)]
        if case == 1:
            masks[1] = None
        if case == 2:
            masks[0] = None
        if case == 3:
            masks = [np.full((H, W), 255, np.uint8), np.full((H, W), 1, np.uint8)]
        src = frames[case % 3]
        ref = od.match(src, 72.0, masks=masks)
        same(det.match_masked(src, masks, 72.0), ref)
        for l in range(2):
            for m in range(2):
                assert np.array_equal(det.debug_quantized(0, l, m), od.quantized(l, m, (H >> l, W >> l))), (case, l, m)
        if case == 3:
            same(ref, od.match(src, 72.0))
    assert len(od.match(frames[0], 72.0)) > len(od.match(frames[0], 72.0, masks=[blocky(0.3), None]))
    # a batch: frame 1 without masks; then a plain upload drops the masks again
    bm = [[blocky(0.7), blocky(0.7)], [None, None], [blocky(0.6), None]]
    det.upload(frames)
    det.upload_masks(bm)
    det.enqueue(3, 72.0)
    outs = det.collect(3)
    for f in range(3):
        same(outs[f], od.match(frames[f], 72.0, masks=None if f == 1 else bm[f]))
    det.upload(frames)
    det.enqueue(3, 72.0)
    outs = det.collect(3)
    for f in range(3):
        same(outs[f], od.match(frames[f], 72.0))
    # upstream's CV_Asserts: masks.size() == modalities.size(), mask.size() == source.size(), 8UC1
    with pytest.raises(_lib.LmxError, match="masks.size"):
        det.upload_masks([[blocky(0.5)]] * 3)
    with pytest.raises(_lib.LmxError, match="size"):
        det.upload_masks([[np.ones((H, W // 2), np.uint8), None]] * 3)
    with pytest.raises(_lib.LmxError, match="most recent upload"):
        det.upload([frames[0]])
        det.upload_masks(bm)
    det.close()


def test_masks_for_fewer_frames_than_uploaded_and_a_second_mask_upload():
    """lmx_ctx_upload_masks for n_frames < the uploaded batch marked the whole set as masked, so frames
    [n_frames, n_uploaded) were filtered through mask memory nobody had written; and a second upload_masks for the same frames was not
    ordered behind the kernels of the previous enqueue.  Upload 4 frames, mask 2, enqueue 4: frames 2 and 3 equal the UNMASKED oracle;
    then new masks for the same upload while the first enqueue may still be running (lanes), twice in a row."""
    W, H = 320, 240
    bank = synth.make_bank(40, seed=311, size_range=(30.0, 70.0))
    od = o.OracleDetector(bank)
    rng = np.random.default_rng(19)

    def blocky(p):
        m = (rng.uniform(0, 1, (H // 16, W // 16)) < p).astype(np.uint8)
        return np.ascontiguousarray(np.kron(m, np.ones((16, 16), np.uint8)))
    frames = [synth.make_scene(bank, W, H, seed=312 + f)[0] for f in range(4)]
    for overlap in (False, True):
        det = Detector(bank, W, H, max_batch=4, overlap=overlap)
        # poison the mask buffers first: a full-batch upload of all-zero masks, so that stale memory would drop every label
        det.upload(frames)
        det.upload_masks([[np.zeros((H, W), np.uint8)] * 2] * 4)
        det.enqueue(4, 72.0)
        assert all(len(m) == 0 for m in det.collect(4))
        det.upload(frames)
        bm = [[blocky(0.7), blocky(0.7)], [blocky(0.6), None]]
        det.upload_masks(bm)          # 2 of the 4 uploaded frames
        det.enqueue(4, 72.0)
        bm2 = [[blocky(0.5), None], [None, blocky(0.8)], [blocky(0.7), blocky(0.7)]]
        det.upload_masks(bm2)         # the same frames, other masks, while the first enqueue is in flight
        det.enqueue(4, 72.0)
        first, second = det.collect(4), det.collect(4)
        for f in range(4):
            same(first[f], od.match(frames[f], 72.0, masks=bm[f] if f < 2 else None))
            same(second[f], od.match(frames[f], 72.0, masks=bm2[f] if f < 3 else None))
        assert sum(len(m) for m in first[2:]) > 0
        det.close()


def test_score_no_prune_on_the_bench_workload_shape():
    """The full-work leg of the roofline (bench.py extra.score_full_work): 640x480 RGB-D, a 64-frame batch on the XCD-aware path, 300 templates,
    threshold 92 -- with the pruning compiled out the candidate lists and the matches are the ones the shipped kernel gives, frame by frame."""
    import os
    bank = synth.make_bank(300, seed=20250215)
    frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(16)]
    outs = {}
    for flag in ("0", "1"):
        os.environ["LMX_SCORE_NO_PRUNE"] = flag
        try:
            det = Detector(bank, 640, 480, max_batch=16, overlap=True)
        finally:
            del os.environ["LMX_SCORE_NO_PRUNE"]
        det.upload(frames)
        det.enqueue(16, 92.0)
        outs[flag] = (det.collect(16), det.stats()["candidates"])
        det.close()
    assert outs["0"][1] == outs["1"][1] > 0
    od = o.OracleDetector(bank)
    for f in range(16):
        same(outs["0"][0][f], outs["1"][0][f])
        if f < 4:
            same(outs["1"][0][f], od.match(frames[f], 92.0))


def test_streamed_frame_stores_of_the_one_frame_call(monkeypatch):
    """lmx_match with a fresh host frame launches the level-0 quantisers BEFORE the frame is on the device; their workgroups wait,
    tile by tile, for the rows the calling thread is still storing (a progress word behind every band of rows).  Same results as store-then-launch
    (LMX_NO_STREAM_STORE=1) and as the oracle: RGB-D and ColorGradient only, strided ROI views, two frames per call, many calls in a row on one
    context (the sequence number in the progress word tells a call's stores from the previous call's)."""
    W, H = 640, 480
    for mods in (("ColorGradient", "DepthNormal"), ("ColorGradient",)):
        bank = synth.make_bank(120, modalities=mods, seed=611, size_range=(55.0, 150.0))
        od = o.OracleDetector(bank)
        frames = [synth.make_scene(bank, W, H, seed=612 + f, row_pad=112 if f % 2 else 0)[0] for f in range(5)]
        refs = [od.match(fr, 86.0) for fr in frames]
        assert sum(len(r) for r in refs) > 5
        dets = {}
        for name, env in (("streamed", None), ("stored", "1")):
            if env:
                monkeypatch.setenv("LMX_NO_STREAM_STORE", env)
            dets[name] = Detector(bank, W, H, max_batch=2)
            monkeypatch.delenv("LMX_NO_STREAM_STORE", raising=False)
        for rep in range(3):
            for f, fr in enumerate(frames):
                for det in dets.values():
                    same(det.match(fr, 86.0), refs[f])
        for det in dets.values():
            got = det.match_batch([frames[1], frames[4]], 86.0)
            same(got[0], refs[1])
            same(got[1], refs[4])
            det.close()


def test_streamed_stores_from_both_ends_band_heights_and_switches(monkeypatch):
    """Round 4: two threads store each modality of the one-frame call from both ends (colour first, then depth), claiming bands of rows from a shared
    counter; a tile starts when its rows lie below the top front, above the bottom front, or the fronts have met.  Band heights that do not divide
    the image (and one band for the whole frame), two frames per call, a second image size, strided views -- and the switches that take threads away
    (LMX_ONE_STORE_THREAD: top-down only; LMX_NO_LAUNCH_THREAD: no helper at all) -- all give the oracle's matches, call after call."""
    for (W, H) in ((640, 480), (320, 400)):
        bank = synth.make_bank(60, seed=631, size_range=(40.0, 110.0) if W == 640 else (30.0, 70.0))
        od = o.OracleDetector(bank)
        frames = [synth.make_scene(bank, W, H, seed=632 + f, row_pad=48 if f == 1 else 0)[0] for f in range(3)]
        refs = [od.match(fr, 84.0) for fr in frames]
        assert sum(len(r) for r in refs) > 3
        for env in ({"LMX_STREAM_BAND_ROWS": "8"}, {"LMX_STREAM_BAND_ROWS": "33"}, {"LMX_STREAM_BAND_ROWS": "96"}, {"LMX_STREAM_BAND_ROWS": "4096"}, {},
                    {"LMX_ONE_STORE_THREAD": "1", "LMX_STREAM_BAND_ROWS": "40"}, {"LMX_NO_LAUNCH_THREAD": "1"}, {"LMX_NO_DELEGATE_FIRST_LAUNCH": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            det = Detector(bank, W, H, max_batch=2)
            for k in env:
                monkeypatch.delenv(k)
            for rep in range(4):
                for f, fr in enumerate(frames):
                    same(det.match(fr, 84.0), refs[f])
            got = det.match_batch([frames[2], frames[1]], 84.0)
            same(got[0], refs[2])
            same(got[1], refs[1])
            det.close()


def test_a_streamed_store_that_never_arrives_is_an_error_not_a_hang(monkeypatch):
    """The bounded wait of the streamed stores: with the test hook that leaves the depth rows out (LMX_TEST_DROP_STREAM_STORE) the depth quantiser's
    workgroups give up after LMX_STREAM_TIMEOUT_US, the call reports the failure, and the device is fine afterwards: another context matches."""
    import time
    W, H = 320, 240
    bank = synth.make_bank(40, seed=621, size_range=(30.0, 70.0))
    src = synth.make_scene(bank, W, H, seed=622)[0]
    monkeypatch.setenv("LMX_TEST_DROP_STREAM_STORE", "1")
    monkeypatch.setenv("LMX_STREAM_TIMEOUT_US", "3000")
    bad = Detector(bank, W, H)
    monkeypatch.delenv("LMX_TEST_DROP_STREAM_STORE")
    monkeypatch.delenv("LMX_STREAM_TIMEOUT_US")
    t0 = time.time()
    with pytest.raises(_lib.LmxError, match="never reached the device"):
        bad.match(src, 80.0)
    assert time.time() - t0 < 5.0
    bad.close()
    det = Detector(bank, W, H)
    same(det.match(src, 80.0), o.OracleDetector(bank).match(src, 80.0))
    det.close()
