"""The header-only C++ facade (include/lmx_linemod.hpp) over the C ABI: builds with plain g++ against liblmx.so
(CPU check) and, on the GPU box, produces the oracle's matches from a YAML bank and raw strided images."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from linemod_pose_estimation_amd import NativeBank, _lib, synth
from oracle import oracle as o

SRC = os.path.join(ROOT, "tests", "cpp", "facade_main.cpp")


def _build(tmp_path):
    exe = str(tmp_path / "facade_main")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", _lib.CSRC, "-llmx",
           "-Wl,-rpath," + _lib.CSRC, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_facade_compiles_and_reports_missing_bank(tmp_path):
    exe = _build(tmp_path)
    res = subprocess.run([exe, "/nonexistent.yml", "160", "160", "160", "80", "/dev/null"], capture_output=True, text=True)
    assert res.returncode == 1 and "cannot open" in res.stderr


@pytest.mark.gpu
def test_facade_matches_oracle(tmp_path):
    exe = _build(tmp_path)
    bank = synth.make_bank(24, seed=61, size_range=(24.0, 60.0))
    yml = tmp_path / "obj_templates.yml"
    NativeBank.from_bank(bank).save_yaml(yml)
    sources, _ = synth.make_scene(bank, 320, 240, seed=62, row_pad=56)   # strided ROI view like the ensenso crop
    stride_px = sources[0].strides[0] // 3
    raw_bgr = np.zeros((240, stride_px, 3), np.uint8)
    raw_bgr[:, :320] = sources[0]
    raw_d = np.zeros((240, stride_px), np.uint16)
    raw_d[:, :320] = sources[1]
    (tmp_path / "bgr.raw").write_bytes(raw_bgr.tobytes())
    (tmp_path / "depth.raw").write_bytes(raw_d.tobytes())
    res = subprocess.run([exe, str(yml), "320", "240", str(stride_px), "74", str(tmp_path / "bgr.raw"), str(tmp_path / "depth.raw")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0] == "classes 1 templates 24 levels 2" and lines[-1] == "exception status 2"
    ref = o.OracleDetector(bank).match(sources, 74.0)
    got = [l.split() for l in lines[1:-1]]
    assert len(got) == len(ref) > 0
    for g, r in zip(got, ref):
        assert (int(g[0]), int(g[1]), int(g[4])) == (r["x"], r["y"], r["template_id"]) and g[3] == "obj"
        assert np.float32(float(g[2])) == r["similarity"]
