"""include/lmx_cv_linemod.hpp: the cv::linemod-shaped facade (cv::linemod::Detector / Template / Feature / Match / Modality with
read(FileNode), readClass(FileNode), write(FileStorage), writeClass, addTemplate, getTemplates, match(..., noArray())), compiled
with a caller written in the reference's style (tests/cpp/cv_facade_main.cpp) against a stand-in for the OpenCV core types
(tests/cpp/cv_standin, this image has no OpenCV).  CPU: it builds with plain g++ and FileNode -> Detector -> FileStorage is
lossless.  GPU: matching a cropped ROI view, the per-request detector rebuild hitting the device-context cache, and training
through addTemplate all equal the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from linemod_pose_estimation_amd import NativeBank, _lib, synth
from linemod_pose_estimation_amd.bank import TemplateBank, DEFAULT_COLOR_GRADIENT, DEFAULT_DEPTH_NORMAL
from oracle import oracle as o

SRC = os.path.join(ROOT, "tests", "cpp", "cv_facade_main.cpp")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cvfacade") / "cv_facade_main")
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "cv_standin"), "-I", os.path.join(ROOT, "include"),
           SRC, "-o", out, "-pthread", "-L", _lib.CSRC, "-llmx", "-Wl,-rpath," + _lib.CSRC, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return out


def test_filenode_to_detector_to_filestorage_is_lossless(exe, tmp_path):
    bank = synth.make_bank(5, seed=4, classes=["obj", "other"], size_range=(20.0, 60.0))
    a, b = tmp_path / "a.yml", tmp_path / "b.yml"
    NativeBank.from_bank(bank).save_yaml(a)
    res = subprocess.run([exe, "rewrite", str(a), str(b)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert a.read_text() == b.read_text()          # Detector::write / writeClass emit what lmx_bank_save_yaml (== upstream's layout) emits
    # the hand-authored OpenCV-style fixture goes through read(FileNode) / readClass(FileNode) as well
    res = subprocess.run([exe, "rewrite", os.path.join(ROOT, "tests", "golden", "opencv_style_templates.yml"), str(b)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    back = NativeBank.load_yaml(b).to_bank()
    ref = NativeBank.load_yaml(os.path.join(ROOT, "tests", "golden", "opencv_style_templates.yml")).to_bank()
    assert back.T == ref.T and [c[0] for c in sorted(back.classes)] == [c[0] for c in sorted(ref.classes)]
    for (_, ta, fa), (_, tb, fb) in zip(sorted(back.classes, key=lambda c: c[0]), sorted(ref.classes, key=lambda c: c[0])):
        assert np.array_equal(ta, tb) and np.array_equal(fa, fb)
    res = subprocess.run([exe, "rewrite", "/nonexistent.yml", str(b)], capture_output=True, text=True)
    assert res.returncode == 1 and "cannot open" in res.stderr


def test_cache_trim_drops_unreferenced_cached_banks(tmp_path):
    """lmx_cache_trim: cached banks nobody references are dropped (a later cached load parses again: a new object), referenced ones stay."""
    import ctypes as C
    L = _lib.lib()
    p = tmp_path / "t.yml"
    NativeBank.from_bank(synth.make_bank(3, seed=9, size_range=(20.0, 40.0))).save_yaml(p)
    h1, h2, h3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h1)))
    L.lmx_cache_trim()                                   # h1 is referenced: stays
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h2)))
    assert h1.value == h2.value
    fp = L.lmx_bank_fingerprint(h1)
    L.lmx_bank_release(h1); L.lmx_bank_release(h2)
    L.lmx_cache_trim()                                   # nobody references it any more: dropped
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h3)))
    assert L.lmx_bank_fingerprint(h3) == fp and L.lmx_bank_num_templates(h3, None) == 3
    L.lmx_bank_release(h3)
    L.lmx_cache_trim()


def test_loaded_detector_is_copy_on_write(exe):
    """cv::linemod::Detector::load hands out the templates file's cached, shared bank; a detector that is modified afterwards switches to a
    private copy and other detectors of the same file do not see the change (CPU only)."""
    res = subprocess.run([exe, "cow", os.path.join(ROOT, "tests", "golden", "opencv_style_templates.yml")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.strip() == "a 2 classes 2 | b 4 classes 3 | before 2 levels 2 T0 5 modalities 2"


def test_generated_message_header_after_the_facade(exe):
    """the service node includes rgbdDetector.h (the facade, with `#define linemod lmx_linemod` active from there on) BEFORE
    the generated message header linemod_pose_estimation/linemod.h (src/linemod_ensenso_detect_3_mult_detect_service.cpp:1, :16), whose
    `typedef ... linemod;` is therefore renamed.  cv_facade_main.cpp includes stand-ins of the generated headers in exactly that order and uses
    the message, its Ptr typedefs, the service's Request / Response and cv::linemod::Match side by side: it compiles (with -Werror) and the
    message's type name -- a string literal, which the macro cannot touch -- is still the one on the wire (CPU only)."""
    res = subprocess.run([exe, "publish", os.path.join(ROOT, "tests", "golden", "opencv_style_templates.yml")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.strip() == "linemod_pose_estimation/linemod 2 messages, last id 1 templates 0, match template_id 1, pose.x 0.001"


def test_bank_and_yaml_tree_caches(tmp_path):
    """lmx_bank_load_yaml_cached: one parse per (path, mtime, size); the document tree API walks a yml like cv::FileNode."""
    import ctypes as C
    import time
    L = _lib.lib()
    bank = synth.make_bank(4, seed=5, size_range=(20.0, 40.0))
    p = tmp_path / "c.yml"
    NativeBank.from_bank(bank).save_yaml(p)
    h1, h2, h3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h1)))
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h2)))
    assert h1.value == h2.value and L.lmx_bank_num_templates(h1, None) == 4
    fp = L.lmx_bank_fingerprint(h1)
    assert fp == L.lmx_bank_fingerprint(NativeBank.load_yaml(p).h) != 0
    time.sleep(0.02)
    NativeBank.from_bank(synth.make_bank(6, seed=6, size_range=(20.0, 40.0))).save_yaml(p)      # the file changes: new parse
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h3)))
    assert h3.value != h1.value and L.lmx_bank_num_templates(h3, None) == 6 and L.lmx_bank_fingerprint(h3) != fp
    for h in (h1, h2, h3):
        L.lmx_bank_release(h)
    clone = C.c_void_p()
    _lib.check(L.lmx_bank_clone(h3, C.byref(clone)))
    assert L.lmx_bank_fingerprint(clone) == L.lmx_bank_fingerprint(h3)
    L.lmx_bank_destroy(clone)
    # compact binary form: lossless, checksummed; the cached loader leaves one next to the yml and a "new process" (here: a cache
    # miss in memory is not possible to force, so the side file is loaded directly) reads it instead of the YAML
    side = str(p) + ".lmxcache"
    assert 8000 < os.path.getsize(side) < os.path.getsize(str(p))
    binp = tmp_path / "bank.lmx"
    _lib.check(L.lmx_bank_save_binary(h3, str(binp).encode()))
    hb = C.c_void_p()
    _lib.check(L.lmx_bank_load_binary(str(binp).encode(), C.byref(hb)))
    assert L.lmx_bank_fingerprint(hb) == L.lmx_bank_fingerprint(h3) and L.lmx_bank_num_templates(hb, None) == 6
    assert open(side, "rb").read()[:8] == b"LMXCACH2"
    assert open(side, "rb").read()[32:] == open(binp, "rb").read()          # the side file = {magic, mtime, size, table-inputs key} + the same bytes
    L.lmx_bank_destroy(hb)
    raw = bytearray(open(binp, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    open(binp, "wb").write(bytes(raw))
    assert L.lmx_bank_load_binary(str(binp).encode(), C.byref(hb)) == _lib.LMX_ERR_PARSE and b"checksum" in L.lmx_last_error()
    open(binp, "wb").write(b"not a bank")
    assert L.lmx_bank_load_binary(str(binp).encode(), C.byref(hb)) == _lib.LMX_ERR_PARSE
    doc = C.c_void_p()
    _lib.check(L.lmx_yaml_open(str(p).encode(), C.byref(doc)))
    root = L.lmx_yaml_root(doc)
    assert L.lmx_yaml_kind(root) == 3 and L.lmx_yaml_scalar(L.lmx_yaml_get(root, b"pyramid_levels")) == b"2"
    T = L.lmx_yaml_get(root, b"T")
    assert L.lmx_yaml_kind(T) == 2 and [L.lmx_yaml_scalar(L.lmx_yaml_item(T, i)) for i in range(L.lmx_yaml_size(T))] == [b"5", b"8"]
    assert L.lmx_yaml_key(root, 0) == b"pyramid_levels" and L.lmx_yaml_get(root, b"nope") is None
    L.lmx_yaml_close(doc)


def _fnv1a(data):
    h = 0xcbf29ce484222325
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xffffffffffffffff
    return h


def test_cached_bank_follows_the_normal_lut_inputs(tmp_path, monkeypatch):
    """Advisor finding (round 2): the bank caches were keyed on the yml alone although yaml_load also folds `<yml>.normal_lut` and
    LMX_NORMAL_LUT into the bank.  A side-car that appears later, or the environment's table, must show up in the next cached load --
    in memory and through the .lmxcache file -- and a foreign DepthNormal bank first seen without a table must not stay UNKNOWN."""
    import ctypes as C
    L = _lib.lib()
    monkeypatch.delenv("LMX_NORMAL_LUT", raising=False)
    bank = synth.make_bank(3, seed=15, size_range=(20.0, 40.0))
    p = tmp_path / "rgbd.yml"
    NativeBank.from_bank(bank).save_yaml(p)                         # DepthNormal bank, marker `lmx_normal_lut: default`
    h1, h2, h3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h1)))
    assert L.lmx_bank_normal_lut_origin(h1) == _lib.LMX_LUT_DEFAULT
    table = np.empty(8000, np.uint8)
    _lib.check(L.lmx_bank_get_normal_lut(h1, table.ctypes.data))
    table = np.roll(table, 7).copy()                                            # some other valid table
    (tmp_path / "rgbd.yml.normal_lut").write_bytes(table.tobytes())
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h2)))
    got = np.empty(8000, np.uint8)
    _lib.check(L.lmx_bank_get_normal_lut(h2, got.ctypes.data))
    assert h2.value != h1.value and L.lmx_bank_normal_lut_origin(h2) == _lib.LMX_LUT_SIDECAR and np.array_equal(got, table)
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h3)))      # unchanged inputs: the cached bank
    assert h3.value == h2.value
    for h in (h1, h2, h3):
        L.lmx_bank_release(h)
    # a yml without the marker (what OpenCV writes): unknown table at first, the environment's table once it is named
    q = tmp_path / "foreign.yml"
    q.write_text("".join(l for l in open(p) if not l.startswith("lmx_normal_lut")))
    f1, f2 = C.c_void_p(), C.c_void_p()
    _lib.check(L.lmx_bank_load_yaml_cached(str(q).encode(), C.byref(f1)))
    assert L.lmx_bank_normal_lut_origin(f1) == _lib.LMX_LUT_UNKNOWN
    lut_file = tmp_path / "normal_lut.bin"
    lut_file.write_bytes(table.tobytes())
    monkeypatch.setenv("LMX_NORMAL_LUT", str(lut_file))
    _lib.check(L.lmx_bank_load_yaml_cached(str(q).encode(), C.byref(f2)))
    _lib.check(L.lmx_bank_get_normal_lut(f2, got.ctypes.data))
    assert f2.value != f1.value and L.lmx_bank_normal_lut_origin(f2) == _lib.LMX_LUT_SIDECAR and np.array_equal(got, table)
    L.lmx_bank_release(f1)
    L.lmx_bank_release(f2)


def test_binary_bank_is_validated_not_only_checksummed(tmp_path):
    """Advisor finding (round 2): a binary bank (lmx_bank_load_binary, and the .lmxcache the cached loader picks up by itself) was only
    checksummed.  A damaged-but-rehashed or crafted file -- label 9, 64+ features, a feature range past the array, T = 0 -- must be
    refused like lmx_bank_add_class refuses it, and a bad .lmxcache must fall back to parsing the yml."""
    import ctypes as C
    import struct
    L = _lib.lib()
    bank = synth.make_bank(3, seed=16, size_range=(20.0, 40.0), modalities=("ColorGradient",))
    nb = NativeBank.from_bank(bank)
    binp = tmp_path / "b.lmx"
    _lib.check(L.lmx_bank_save_binary(nb.h, str(binp).encode()))
    good = open(binp, "rb").read()
    feats = np.ascontiguousarray(bank.classes[0][2], np.int32).tobytes()
    templ = np.ascontiguousarray(bank.classes[0][1], np.int32).tobytes()
    f_off, t_off = good.index(feats), good.index(templ)
    hb = C.c_void_p()

    def load(mutated):
        body = bytes(mutated[:-8])
        open(binp, "wb").write(body + struct.pack("<Q", _fnv1a(body)))      # a VALID checksum over the damaged body
        return L.lmx_bank_load_binary(str(binp).encode(), C.byref(hb)), L.lmx_last_error()

    st, _ = load(bytearray(good))
    assert st == _lib.LMX_OK
    L.lmx_bank_destroy(hb)
    m = bytearray(good); m[f_off + 8:f_off + 12] = struct.pack("<i", 9)          # label of feature 0
    st, msg = load(m)
    assert st == _lib.LMX_ERR_PARSE and b"out of range" in msg
    m = bytearray(good); m[t_off + 16:t_off + 20] = struct.pack("<i", 64)        # feature count of template 0
    st, msg = load(m)
    assert st == _lib.LMX_ERR_PARSE and b"63" in msg
    m = bytearray(good); m[t_off + 12:t_off + 16] = struct.pack("<i", 1 << 28)   # feature range past the array
    st, msg = load(m)
    assert st == _lib.LMX_ERR_PARSE and b"out of bounds" in msg
    m = bytearray(good); m[16:20] = struct.pack("<i", 0)                         # T[0] = 0
    st, msg = load(m)
    assert st == _lib.LMX_ERR_PARSE and b"outside 1..16" in msg
    # the cached loader: a .lmxcache with a matching key but an invalid bank inside is ignored, the yml is parsed
    p = tmp_path / "c.yml"
    nb.save_yaml(p)
    h1 = C.c_void_p()
    _lib.check(L.lmx_bank_load_yaml_cached(str(p).encode(), C.byref(h1)))
    fp = L.lmx_bank_fingerprint(h1)
    L.lmx_bank_release(h1)
    side = bytearray(open(str(p) + ".lmxcache", "rb").read())
    inner = side[32:]
    k = bytes(inner).index(feats)
    inner[k + 8:k + 12] = struct.pack("<i", 9)
    body = bytes(inner[:-8])
    open(str(p) + ".lmxcache", "wb").write(bytes(side[:32]) + body + struct.pack("<Q", _fnv1a(body)))
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\nfrom linemod_pose_estimation_amd import _lib\nL = _lib.lib(); h = C.c_void_p()\n"
            "_lib.check(L.lmx_bank_load_yaml_cached(%r.encode(), C.byref(h)))\nprint(L.lmx_bank_fingerprint(h))" % (ROOT, str(p)))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)    # a fresh process: no in-memory entry
    assert res.returncode == 0 and int(res.stdout.strip()) == fp, res.stderr


@pytest.mark.gpu
def test_cv_style_caller_matches_oracle_and_reuses_the_resident_bank(exe, tmp_path):
    bank = synth.make_bank(24, seed=61, size_range=(24.0, 60.0))
    yml = tmp_path / "obj_templates.yml"
    NativeBank.from_bank(bank).save_yaml(yml)
    W, H, frame_cols, crop_x = 320, 240, 376, 28                   # the crop of a wider camera frame, like Rect(56, 0, 640, 480) of 752
    sources, _ = synth.make_scene(bank, W, H, seed=62)
    raw_bgr = np.random.default_rng(1).integers(0, 255, (H, frame_cols, 3), dtype=np.uint8)
    raw_bgr[:, crop_x:crop_x + W] = sources[0]
    raw_d = np.zeros((H, frame_cols), np.uint16)
    raw_d[:, crop_x:crop_x + W] = sources[1]
    (tmp_path / "bgr.raw").write_bytes(raw_bgr.tobytes())
    (tmp_path / "depth.raw").write_bytes(raw_d.tobytes())
    res = subprocess.run([exe, "match", str(yml), str(W), str(H), str(frame_cols), str(crop_x), "74", str(tmp_path / "bgr.raw"), str(tmp_path / "depth.raw")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0] == "request 0 context_cached 0" and lines[1] == "request 1 context_cached 1"   # the rebuilt detector found its bank on the device
    assert lines[2] == "classes 1 templates 24 levels 2 T0 5 modalities 2" and lines[-1] == "exception"
    od = o.OracleDetector(bank)
    ref = od.match(sources, 74.0)
    got = [l.split() for l in lines[3:-2]]
    assert len(got) == len(ref) > 0
    for g, r in zip(got, ref):
        assert (int(g[0]), int(g[1]), int(g[4])) == (r["x"], r["y"], r["template_id"]) and g[3] == "obj" and int(g[5]) == 126
        assert np.float32(float(g[2])) == r["similarity"]
    qsum = sum(int(od.quantized(l, m, (H >> l, W >> l)).astype(np.uint64).sum()) for l in range(2) for m in range(2))
    assert lines[-2] == "quantized 4 %dx%d %d same 1" % (W, H, qsum)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["requests", "requests_cached"])
def test_per_request_detector_rebuild_both_ways(exe, tmp_path, mode):
    """The service node's callback builds its detector on every request (src/linemod_ensenso_detect_3_mult_detect_service.cpp:1784-1786).
    `requests`: readLinemod as the reference writes it (FileStorage + read + readClass); `requests_cached`: the one-line variant,
    cv::linemod::Detector::load (shared cached bank, cached device context).  Either way the C++ caller compares the detector and its
    matches with the other way's and must find them identical; the cached way must find the context resident."""
    bank = synth.make_bank(30, seed=71, size_range=(24.0, 60.0))
    yml = tmp_path / "obj_templates.yml"
    NativeBank.from_bank(bank).save_yaml(yml)
    W, H = 320, 240
    sources, _ = synth.make_scene(bank, W, H, seed=72)
    (tmp_path / "bgr.raw").write_bytes(np.ascontiguousarray(sources[0]).tobytes())
    (tmp_path / "depth.raw").write_bytes(np.ascontiguousarray(sources[1]).tobytes())
    res = subprocess.run([exe, mode, str(yml), str(W), str(H), "74", "5", str(tmp_path / "bgr.raw"), str(tmp_path / "depth.raw")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    n_ref = len(o.OracleDetector(bank).match(sources, 74.0))
    assert lines[0].startswith("requests 5 matches %d " % n_ref) and n_ref > 0
    assert lines[1] == "load_equals_readLinemod 1 context_cached 1"


@pytest.mark.gpu
def test_more_candidates_than_the_default_lists_hold(exe, tmp_path):
    """Upstream's match() has no capacity limit.  At threshold 30 this scene yields more than the 16384 coarse candidates the
    device lists hold by default: the facade re-acquires a context with larger lists and repeats the call (twice the same result)."""
    bank = synth.make_bank(300, seed=75, size_range=(24.0, 90.0))
    yml = tmp_path / "obj_templates.yml"
    NativeBank.from_bank(bank).save_yaml(yml)
    W, H = 320, 240
    bgr, depth = synth.make_scene(bank, W, H, seed=76, n_instances=5, texture=1.0)[0]
    (tmp_path / "bgr.raw").write_bytes(np.ascontiguousarray(bgr).tobytes())
    (tmp_path / "depth.raw").write_bytes(np.ascontiguousarray(depth).tobytes())
    od = o.OracleDetector(bank)
    ref = od.match([bgr, depth], 30.0)
    assert od.last_candidates() > 16384 + 4096
    s = 0
    for r in ref:
        s = (s * 1000003 + (int(r["x"]) * 7 + int(r["y"]) * 13 + int(r["template_id"]) * 31 + int(np.float32(r["similarity"]) * np.float32(1000.0)))) % (1 << 64)
    res = subprocess.run([exe, "count", str(yml), str(W), str(H), "30", str(tmp_path / "bgr.raw"), str(tmp_path / "depth.raw")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.strip().splitlines() == ["matches %d checksum %d" % (len(ref), s)] * 2


@pytest.mark.gpu
def test_two_threads_with_their_own_detectors_share_one_context_safely(exe, tmp_path):
    """Two cv::linemod::Detector objects read from the same yml in two threads get the same cached device context
    (lmx_ctx_acquire); lmx_match serialises them (lmx_ctx_lock).  80 concurrent matches of two different frames against the
    one-thread results, which in turn equal the oracle."""
    bank = synth.make_bank(60, seed=71, size_range=(24.0, 70.0))
    yml = tmp_path / "obj_templates.yml"
    NativeBank.from_bank(bank).save_yaml(yml)
    W, H = 320, 240
    scenes = [synth.make_scene(bank, W, H, seed=72 + k, n_instances=4)[0] for k in range(2)]
    args = []
    for k, (bgr, depth) in enumerate(scenes):
        (tmp_path / ("bgr%d.raw" % k)).write_bytes(np.ascontiguousarray(bgr).tobytes())
        (tmp_path / ("depth%d.raw" % k)).write_bytes(np.ascontiguousarray(depth).tobytes())
        args += [str(tmp_path / ("bgr%d.raw" % k)), str(tmp_path / ("depth%d.raw" % k))]
    res = subprocess.run([exe, "threads", str(yml), str(W), str(H), "70"] + args, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    od = o.OracleDetector(bank)
    n = [len(od.match(s, 70.0)) for s in scenes]
    assert min(n) > 0 and n[0] != n[1]
    assert res.stdout.strip().splitlines()[-1] == "threads expect %d %d mismatches 0 0 shared_context 2" % (n[0], n[1])


@pytest.mark.gpu
@pytest.mark.parametrize("n_mod", [1, 2])
def test_cv_style_trainer_equals_the_oracle(exe, tmp_path, n_mod):
    import train_util
    views = [v for v in (train_util.rendered_view(s) for s in (91, 93, 97)) if v is not None]
    assert len(views) >= 2
    raw = b"".join(bgr.tobytes() + depth.tobytes() + mask.tobytes() for bgr, depth, mask in views)
    (tmp_path / "views.raw").write_bytes(raw)
    out = tmp_path / "trained.yml"
    res = subprocess.run([exe, "train", str(out), "320", "240", str(len(views)), str(tmp_path / "views.raw"), str(n_mod)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    mdesc = [dict(DEFAULT_COLOR_GRADIENT), dict(DEFAULT_DEPTH_NORMAL)][:n_mod]
    od = o.OracleDetector(TemplateBank(T=[5, 8], modalities=mdesc))
    lines = res.stdout.strip().splitlines()
    for v, (bgr, depth, mask) in enumerate(views):
        tid, bb = od.add_template([bgr, depth][:n_mod], "obj", mask)
        assert lines[v] == "view %d template %d bb %d %d %d %d" % ((v, tid) + tuple(bb))
    trained = NativeBank.load_yaml(out).to_bank()
    assert trained.num_templates("obj") == len(views)
    for tid in range(len(views)):
        for (w, h, lvl, f), (rw, rh, rl, rf) in zip(trained.get_templates("obj", tid), od.get_templates("obj", tid)):
            assert (w, h, lvl) == (rw, rh, rl) and np.array_equal(f, rf)
