#!/usr/bin/env python3
"""Mutation fuzz of the three file readers a deployment feeds with files it did not write itself (SURVEY 8f row 1 and the side-car):
  lmx_bank_load_yaml          <object>_templates.yml   (OpenCV FileStorage YAML, /root/reference/src/rgbdDetector.cpp:1668-1680)
  lmx_renderer_params_load    <object>_renderer_params.yml (rgbdDetector.cpp:1681-1749)
  lmx_bank_load_binary        <yml>.lmxcache            (this library's own cache format)
Every mutant (byte flips, digit changes, truncations, spliced and duplicated lines, huge numbers) must come back with LMX_OK or a clean
error status; run it under the sanitized host build (scripts/sanitize_host.sh builds it; then
  LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0 LMX_SO_PATH=build/asan/liblmx.so python scripts/fuzz_files.py 3000)
so that an out-of-bounds read or an overflow inside a reader is a report, not luck.  No GPU involved.
usage: python scripts/fuzz_files.py [n_mutants] [seed]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linemod_pose_estimation_amd import _lib, synth, NativeBank  # noqa: E402


def mutate(rng, data):
    b = bytearray(data)
    kind = int(rng.integers(0, 8))
    n = len(b)
    if kind == 0:      # flip a few bytes
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, n))] = int(rng.integers(0, 256))
    elif kind == 1:    # truncate
        b = b[: int(rng.integers(0, n))]
    elif kind == 2:    # change digits
        idx = [i for i in rng.integers(0, n, 40) if 48 <= b[i] <= 57]
        for i in idx[:8]:
            b[i] = 48 + int(rng.integers(0, 10))
    elif kind == 3:    # duplicate a line
        lines = bytes(b).split(b"\n")
        i = int(rng.integers(0, len(lines)))
        lines.insert(i, lines[int(rng.integers(0, len(lines)))])
        b = bytearray(b"\n".join(lines))
    elif kind == 4:    # delete a line
        lines = bytes(b).split(b"\n")
        del lines[int(rng.integers(0, len(lines)))]
        b = bytearray(b"\n".join(lines))
    elif kind == 5:    # a huge / negative number in place of a digit run
        i = int(rng.integers(0, n))
        while i < n and not (48 <= b[i] <= 57):
            i += 1
        j = i
        while j < n and 48 <= b[j] <= 57:
            j += 1
        b[i:j] = rng.choice([b"4294967297", b"-1", b"99999999999999999999", b"2147483647", b"1e309", b"0x10"])
    elif kind == 6:    # splice two halves at random points
        i, j = sorted(int(v) for v in rng.integers(0, n, 2))
        b = b[:i] + b[j:]
    else:              # brackets and braces
        for _ in range(3):
            b[int(rng.integers(0, n))] = int(rng.choice(list(b"[]{}:,-\n ")))
    return bytes(b)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = _lib.lib()
    tmp = tempfile.mkdtemp(prefix="lmx_fuzz_files_")
    bank = synth.make_bank(5, seed=3, classes=["a", "b"], size_range=(24.0, 40.0))
    nb = NativeBank.from_bank(bank)
    yml, bin_ = os.path.join(tmp, "t.yml"), os.path.join(tmp, "t.lmxcache")
    nb.save_yaml(yml)
    _lib.check(L.lmx_bank_save_binary(nb.h, bin_.encode()))
    seeds = {"bank_yaml": open(yml, "rb").read(), "bank_binary": open(bin_, "rb").read(),
             "bank_yaml_cv": open(os.path.join(ROOT, "tests", "golden", "opencv_style_templates.yml"), "rb").read(),
             "renderer_params": open(os.path.join(ROOT, "tests", "golden", "renderer_params_sample.yml"), "rb").read()}
    stats = {}
    path = os.path.join(tmp, "mutant")
    for i in range(n):
        kind = list(seeds)[i % len(seeds)]
        data = mutate(rng, seeds[kind])
        if rng.uniform() < 0.3:
            data = mutate(rng, data)
        open(path, "wb").write(data)
        if os.environ.get("LMX_FUZZ_VERBOSE"):
            print(i, kind, path, flush=True)
        if kind == "renderer_params":
            p = C.POINTER(_lib.RendererParams)()
            st = L.lmx_renderer_params_load(path.encode(), C.byref(p))
            if st == 0:
                r = p.contents   # touch what a caller would read
                nt = r.n_templates
                if nt > 0:
                    float(np.ctypeslib.as_array(r.obj_origin_dists, (nt,)).sum())
                    int(np.ctypeslib.as_array(r.rects, (nt, 4)).sum())
                L.lmx_renderer_params_free(p)
        else:
            h = C.c_void_p()
            st = (L.lmx_bank_load_binary if kind == "bank_binary" else L.lmx_bank_load_yaml)(path.encode(), C.byref(h))
            if st == 0:
                got = NativeBank(h.value)
                try:
                    got.to_bank()        # walks every class, template and feature through the C ABI
                except UnicodeDecodeError:
                    pass                 # a class id with flipped bytes: this harness' decode, not the library
                out = os.path.join(tmp, "again.yml")
                got.save_yaml(out)
                del got
        stats.setdefault(kind, {}).setdefault(int(st), 0)
        stats[kind][int(st)] += 1
    for k, v in stats.items():
        print("%-16s %s" % (k, "  ".join("status %d: %d" % (s, c) for s, c in sorted(v.items()))))
    print("file fuzz ok: %d mutants" % n)


if __name__ == "__main__":
    main()
