"""Loader for tests/golden/case_*.npz (see tests/golden/make_golden.py)."""
import glob
import os
import zlib

import numpy as np

from linemod_pose_estimation_amd.bank import TemplateBank, DEFAULT_COLOR_GRADIENT, DEFAULT_DEPTH_NORMAL

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "case_*.npz")))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def load(path):
    z = np.load(path)
    mods = [dict(DEFAULT_COLOR_GRADIENT if str(m) == "ColorGradient" else DEFAULT_DEPTH_NORMAL) for m in z["modalities"]]
    bank = TemplateBank(T=[int(t) for t in z["T"]], modalities=mods)
    for ci, cid in enumerate(z["class_ids"]):
        bank.classes.append((str(cid), z["templates_%d" % ci].astype(np.int32), z["features_%d" % ci].astype(np.int32)))
    if "normal_lut" in z.files:
        bank.normal_lut = z["normal_lut"]
    sources = [z["source_%d" % m] for m in range(len(mods))]
    return z, bank, sources


def same_matches(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for k in ("x", "y", "similarity", "template_id", "class_index"):
        assert np.array_equal(a[k], b[k]), k


_BIG = {}


def bank_50k():
    """BASELINE configs[3] / [4]'s bank: 50 000 synthetic templates (SURVEY 8d generator, seed 20250213 + 4).  25 s to generate, so it is
    made once per test process."""
    if "bank" not in _BIG:
        from linemod_pose_estimation_amd import synth
        _BIG["bank"] = synth.make_bank(50000, seed=20250217)
    return _BIG["bank"]
