#!/usr/bin/env python3
"""Experiment: do two contexts on private streams overlap (score kernel is cache-pipeline bound, the quantisers VALU bound)?
Alternates enqueue/collect over n contexts that hold the same frames; prints frames/s for n = 1, 2, 3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linemod_pose_estimation_amd import synth, Detector

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bank = synth.make_bank(3000, modalities=("ColorGradient", "DepthNormal"), T=(5, 8), seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(B)]
for n in (1, 2, 3):
    for b in sorted({B, B // n}):
        dets = [Detector(bank, 640, 480, device=0, max_batch=b) for _ in range(n)]
        for d in dets:
            d.upload(frames[:b])
        def run(k):
            for d in dets:
                d.enqueue(b, 92.0)
            for _ in range(k - 1):
                for d in dets:
                    d.enqueue(b, 92.0)
                    d.collect(b)
            for d in dets:
                d.collect(b)
        run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        run(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("contexts %d  frames/ctx-step %3d  -> %.1f frames/s  (%.3f ms per %d frames)" % (n, b, n * b * K / dt, dt / K * 1e3, n * b), flush=True)
        del dets
