"""csrc/lmx_sort_emul.hpp restates the algorithm of libstdc++'s std::sort so that the DEVICE can reproduce upstream's order of ties
(Detector::match: std::sort + std::unique on the matches; nonMaximaSuppressionUsingIOU: std::sort on the clusters,
/root/reference/src/rgbdDetector.cpp:462-530).  Here its host build is compared with the real std::sort (called by the oracle
library) on the permutation it produces: random inputs with heavy ties, all-equal, sorted, reversed, organ-pipe and
median-of-three-killer sequences (the last ones drive introsort into its heap-sort fallback), sizes around the insertion-sort
threshold of 16."""
import ctypes as C

import numpy as np
import pytest

from linemod_pose_estimation_amd import _lib
from oracle import oracle as o


def _std_perm(sim, tid):
    L = o.lib()
    L.lmo_std_sort_perm.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    L.lmo_std_sort_perm.restype = None
    perm = np.empty(len(sim), np.int32)
    L.lmo_std_sort_perm(sim.ctypes.data, tid.ctypes.data, len(sim), perm.ctypes.data)
    return perm


def _emu_perm(sim, tid):
    perm = np.empty(len(sim), np.int32)
    _lib.check(_lib.lib().lmx_debug_introsort_perm(sim.ctypes.data, tid.ctypes.data, len(sim), perm.ctypes.data))
    return perm


def _check(sim, tid):
    sim = np.ascontiguousarray(sim, np.float32)
    tid = np.ascontiguousarray(tid, np.int32)
    a, b = _std_perm(sim, tid), _emu_perm(sim, tid)
    assert np.array_equal(a, b), (len(sim), np.flatnonzero(a != b)[:5])
    return a


def median_of_three_killer(n):
    """Musser's sequence: quicksort with median-of-three pivots degrades on it, so introsort reaches its depth limit."""
    k = n // 2
    a = np.zeros(n, np.int64)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def test_random_tie_heavy_inputs():
    rng = np.random.default_rng(0)
    total = 0
    for n in list(range(0, 40)) + [63, 64, 65, 100, 255, 256, 257, 1000, 2047, 2048, 5000]:
        for levels, tids in ((1, 1), (2, 2), (3, 50), (10, 5), (40, 3000), (10 ** 6, 3000)):
            for rep in range(6 if n < 300 else 2):
                sim = rng.integers(0, levels, n).astype(np.float32) * 0.25 + 90
                tid = rng.integers(0, tids, n)
                _check(sim, tid)
                total += 1
    assert total > 1000


@pytest.mark.parametrize("n", [17, 33, 100, 500, 2048, 6000, 20000])
def test_structured_inputs_including_the_heap_sort_fallback(n):
    idx = np.arange(n)
    zeros = np.zeros(n, np.int32)
    _check(np.full(n, 95.0), zeros)                                   # every element equal
    _check(idx.astype(np.float32), zeros)                             # ascending similarity = reversed for this comparator
    _check(-idx.astype(np.float32), zeros)                            # already sorted
    _check(np.full(n, 95.0), idx)                                     # ties broken by template_id, sorted
    _check(np.full(n, 95.0), idx[::-1])                               # ... reversed
    _check(np.minimum(idx, n - 1 - idx).astype(np.float32), zeros)    # organ pipe
    _check(np.full(n, 95.0), median_of_three_killer(n))               # drives the depth limit -> heap sort
    _check(-median_of_three_killer(n).astype(np.float32), idx % 3)
    _check((idx % 7).astype(np.float32), (idx * 31) % 11)             # periodic with ties


def test_score_comparator():
    L = o.lib()
    L.lmo_std_sort_perm_score.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    L.lmo_std_sort_perm_score.restype = None
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 15, 16, 17, 18, 40, 300, 3000]:
        for levels in (1, 3, 1000):
            score = (rng.integers(0, levels, n) / 7.0).astype(np.float64)
            a = np.empty(n, np.int32)
            b = np.empty(n, np.int32)
            L.lmo_std_sort_perm_score(score.ctypes.data, n, a.ctypes.data)
            _lib.check(_lib.lib().lmx_debug_introsort_perm_score(score.ctypes.data, n, b.ctypes.data))
            assert np.array_equal(a, b), (n, levels)


def _device_perm(sim, tid):
    perm = np.empty(len(sim), np.int32)
    _lib.check(_lib.lib().lmx_debug_device_sort_perm(0, sim.ctypes.data, tid.ctypes.data, len(sim), perm.ctypes.data))
    return perm


@pytest.mark.gpu
def test_device_block_sort_equals_std_sort():
    """csrc/lmx_sort_block.hpp -- the workgroup-parallel form k_f2_finalize_cluster runs (level-parallel partitions, stable rank inside the
    leaves, heap-sort fallback) -- against the REAL std::sort: the same permutation, ties included, for the same inputs as the host
    restatement above: random with heavy ties, sizes around 16 and up to 2048, all-equal, sorted, reversed, organ pipe, periodic, and
    median-of-three killers (depth limit -> heap sort)."""
    rng = np.random.default_rng(5)
    total = 0

    def check(sim, tid):
        sim = np.ascontiguousarray(sim, np.float32)
        tid = np.ascontiguousarray(tid, np.int32)
        a, b = _std_perm(sim, tid), _device_perm(sim, tid)
        assert np.array_equal(a, b), (len(sim), np.flatnonzero(a != b)[:5], a[:8], b[:8])

    for n in list(range(0, 40)) + [63, 64, 65, 100, 255, 256, 257, 511, 1000, 1024, 2047, 2048]:
        for levels, tids in ((1, 1), (2, 2), (3, 50), (10, 5), (40, 3000), (10 ** 6, 3000)):
            for rep in range(4 if n < 300 else 2):
                check(rng.integers(0, levels, n).astype(np.float32) * 0.25 + 90, rng.integers(0, tids, n))
                total += 1
    for n in (17, 33, 100, 500, 2048):
        idx = np.arange(n)
        zeros = np.zeros(n, np.int32)
        check(np.full(n, 95.0), zeros)
        check(idx.astype(np.float32), zeros)
        check(-idx.astype(np.float32), zeros)
        check(np.full(n, 95.0), idx)
        check(np.full(n, 95.0), idx[::-1])
        check(np.minimum(idx, n - 1 - idx).astype(np.float32), zeros)
        check(np.full(n, 95.0), median_of_three_killer(n))
        check(-median_of_three_killer(n).astype(np.float32), idx % 3)
        check((idx % 7).astype(np.float32), (idx * 31) % 11)
        total += 9
    assert total > 1000
