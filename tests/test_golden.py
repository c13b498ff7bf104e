"""Committed golden fixtures (tests/golden/case_*.npz, made by tests/golden/make_golden.py).
CPU: the oracle reproduces them (regression pin of the checker).  GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

import golden_util as G
from oracle import oracle as o


@pytest.mark.parametrize("path", G.CASES, ids=[os.path.basename(p) for p in G.CASES])
def test_oracle_reproduces_golden(path):
    z, bank, sources = G.load(path)
    det = o.OracleDetector(bank)
    got = det.match(sources, float(z["threshold"]))
    G.same_matches(got, z["matches"])
    raw = det.last_raw()
    assert np.array_equal(raw["order_key"], z["raw"]["order_key"]) and np.array_equal(raw["similarity"], z["raw"]["similarity"])
    assert det.last_candidates() == int(z["candidates"])
    H, W = sources[0].shape[:2]
    for l in range(len(bank.T)):
        for m in range(len(bank.modalities)):
            assert np.array_equal(det.quantized(l, m, (H >> l, W >> l)), z["quant_l%d_m%d" % (l, m)])
            assert G.crc(det.linear_memory(l, m, (H >> l, W >> l))) == z["lm_crc_l%d_m%d" % (l, m)]


@pytest.mark.gpu
@pytest.mark.parametrize("path", G.CASES, ids=[os.path.basename(p) for p in G.CASES])
def test_gpu_reproduces_golden(path):
    from linemod_pose_estimation_amd import Detector
    z, bank, sources = G.load(path)
    H, W = sources[0].shape[:2]
    det = Detector(bank, W, H)
    got = det.match(sources, float(z["threshold"]))
    G.same_matches(got, z["matches"])
    assert det.stats()["candidates"] == int(z["candidates"]) and det.stats()["raw_matches"] == len(z["raw"])
    for l in range(len(bank.T)):
        for m in range(len(bank.modalities)):
            assert np.array_equal(det.debug_quantized(0, l, m), z["quant_l%d_m%d" % (l, m)])
            assert G.crc(det.debug_linear_memory(0, l, m)) == z["lm_crc_l%d_m%d" % (l, m)]
    det.close()
