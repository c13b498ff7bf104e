#!/usr/bin/env python3
"""host-frame rate with the frames in pinned memory, for the three transfer forms of lmx_ctx_upload (LMX_PINNED_MODE = pull | dma |
stage) and for pageable frames with 1..16 staging threads (LMX_UPLOAD_THREADS).  Same workload and pipelining as bench.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from linemod_pose_estimation_amd import Detector, PinnedArena, synth  # noqa: E402

B = 64
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, texture=0.6)[0] for f in range(B)]
perms = [np.random.default_rng(s).permutation(B) for s in (1, 2, 3)]
host = [[[np.array(src, copy=True) for src in frames[i]] for i in p] for p in perms]
arena = PinnedArena(3 * B * (bench.FRAME_BYTES + 1024))
pinned = [[[arena.put(src) for src in fr] for fr in batch] for batch in host]
for mode in ("pull", "dma", "stage"):
    os.environ["LMX_PINNED_MODE"] = mode
    r = bench.secondary_line(torch, Detector, bank, None, B, 92.0, 60, overlap=True, uploads=pinned, async_input=True)
    print("pinned frames, LMX_PINNED_MODE=%-5s  %8.0f frames/s  %.3f ms/step  %.1f GB/s  step_ms %s" % (mode, r["value"], r["ms_per_step"], r["value"] * bench.FRAME_BYTES / 1e9, r["step_ms"]), flush=True)
os.environ.pop("LMX_PINNED_MODE")
for nt in (1, 2, 4, 8, 12):
    os.environ["LMX_UPLOAD_THREADS"] = str(nt)
    r = bench.secondary_line(torch, Detector, bank, None, B, 92.0, 60, overlap=True, uploads=host)
    print("pageable frames, %2d staging thread(s)      %8.0f frames/s  %.3f ms/step  %.1f GB/s" % (nt, r["value"], r["ms_per_step"], r["value"] * bench.FRAME_BYTES / 1e9), flush=True)
# ColorGradient-only (the ensenso banks of config 1): 921600 bytes per frame
bank1 = synth.make_bank(3000, modalities=("ColorGradient",), seed=20250214)
host1 = [[[src[0]] for src in batch] for batch in host]
os.environ.pop("LMX_UPLOAD_THREADS")
r = bench.secondary_line(torch, Detector, bank1, None, B, 92.0, 60, overlap=True, uploads=host1)
print("config 1 shape (ColorGradient only), pageable  %8.0f frames/s  %.3f ms/step  %.1f GB/s" % (r["value"], r["ms_per_step"], r["value"] * 640 * 480 * 3 / 1e9), flush=True)
