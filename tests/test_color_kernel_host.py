"""The colour quantiser's kernel body on the CPU (tests/cpp/cq_host.cpp: linemod_pose_estimation_amd/csrc/lmx_color_quantize.hpp compiled with
LMX_CQ_HOST, the workgroup's threads emulated stage by stage) against the oracle: label images (quantizedOrientations + hysteresisGradient,
SURVEY A.2), cv::pyrDown of the source (A.3) and the trainer's squared magnitudes, bit for bit -- image borders (replicate for the blur and
Sobel, reflect-101 for pyrDown), sizes that are no multiple of the 64 x 16 / 64 x 32 tile, odd sizes, both tile heights.  This checks the
device algorithm's index arithmetic and weight vectors in this container (no GPU); the machine instructions themselves are checked on the GPU
by tests/test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as o

CSRC = os.path.join(ROOT, "linemod_pose_estimation_amd", "csrc")


@pytest.fixture(scope="module")
def cq(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("cqhost") / "libcqhost.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-I", CSRC, "-o", so,
                           os.path.join(ROOT, "tests", "cpp", "cq_host.cpp")])
    lib = C.CDLL(so)
    lib.cq_host_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]
    lib.cq_host_label_check.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    lib.cq_host_label_check.restype = C.c_long
    return lib


def run(lib, bgr, th, weak=10.0, pyr=True, mag=False):
    H, W = bgr.shape[:2]
    dst = np.full((H, W), 0x5a, np.uint8)
    pd = np.full((H // 2, W // 2, 3), 0x5a, np.uint8) if pyr else None
    mg = np.full((H, W), -1.0, np.float32) if mag else None
    rc = lib.cq_host_run(bgr.ctypes.data, dst.ctypes.data, pd.ctypes.data if pyr else None, mg.ctypes.data if mag else None, H, W, C.c_float(weak), th)
    assert rc == 0
    return dst, pd, mg


def image(rng, H, W, kind):
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == "smooth":   # gradients strong enough to pass the weak threshold, low-pass so that votes reach 5 of 9
        a = rng.uniform(0, 255, (H // 8 + 2, W // 8 + 2, 3))
        a = np.kron(a, np.ones((8, 8, 1)))[:H, :W]
        return np.ascontiguousarray(np.clip(a + rng.normal(0, 3, a.shape), 0, 255).astype(np.uint8))
    if kind == "extreme":  # saturating blocks: Sobel outputs reach +-1020, equal channel magnitudes (first-channel precedence)
        a = (rng.integers(0, 2, (H // 4 + 1, W // 4 + 1, 1)) * 255).astype(np.uint8)
        return np.ascontiguousarray(np.repeat(np.kron(a, np.ones((4, 4, 1), np.uint8))[:H, :W], 3, axis=2))
    raise ValueError(kind)


@pytest.mark.parametrize("H,W", [(96, 128), (64, 64), (37, 70), (33, 131), (16, 8), (5, 4), (4, 5), (130, 66), (67, 257), (240, 320)])
@pytest.mark.parametrize("th", [16, 32])
def test_labels_and_pyrdown_equal_the_oracle(cq, H, W, th):
    rng = np.random.default_rng(H * 1000 + W + th)
    for kind in ("smooth", "noise", "extreme"):
        bgr = image(rng, H, W, kind)
        ref_q, ref_mag, _ = o.quantized_orientations(bgr, 10.0)
        dst, pd, mg = run(cq, bgr, th, mag=True)
        assert np.array_equal(dst, ref_q), (kind, np.argwhere(dst != ref_q)[:5])
        assert np.array_equal(pd, o.pyrdown(bgr)), (kind, np.argwhere(pd != o.pyrdown(bgr))[:5])
        assert np.array_equal(mg, ref_mag), (kind, np.argwhere(mg != ref_mag)[:5])
        if kind == "smooth" and H * W > 4000:
            assert (ref_q != 0).mean() > 0.05           # the comparison is not about empty images
    # no pyramid output (the coarsest level), another weak threshold
    bgr = image(rng, H, W, "smooth")
    dst, _, _ = run(cq, bgr, th, weak=25.0, pyr=False)
    assert np.array_equal(dst, o.quantized_orientations(bgr, 25.0)[0])


def test_label_rule_without_compares_is_the_16_bin_rule_mod_8(cq):
    """orientation_label8 (sign masks, cross-product signs; what the kernel runs) == orientation_label16 & 7 == the oracle's float pipeline & 7
    for every Sobel gradient an 8-bit image can produce: all 2041 x 2041 (dx, dy)."""
    v = np.arange(-1020, 1021, dtype=np.int16)
    dx, dy = np.meshgrid(v, v)
    dx, dy = np.ascontiguousarray(dx.reshape(-1)), np.ascontiguousarray(dy.reshape(-1))
    out = np.empty(dx.size, np.uint8)
    assert cq.cq_host_label_check(dx.ctypes.data, dy.ctypes.data, dx.size, out.ctypes.data) == 0
    assert np.array_equal(out, o.orientation_labels(dx, dy) & 7)


def test_lds_budget(cq):
    """Six workgroups of the tall tile per CU (160 KB of LDS): the layout must stay below 160 KB / 6."""
    assert cq.cq_host_lds_bytes(32) <= 160 * 1024 // 6 and cq.cq_host_lds_bytes(16) <= 160 * 1024 // 9
