#!/usr/bin/env python3
"""Where does the pipelined ShardedMatcher spend its time (one rank, no process group)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linemod_pose_estimation_amd import synth
from linemod_pose_estimation_amd.dist import ShardedMatcher
B = 64
torch.cuda.set_device(0)
bank = synth.make_bank(3000, modalities=("ColorGradient", "DepthNormal"), T=(5, 8), seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(B)]
for ov in (True, False, True):
    sm = ShardedMatcher(bank, 640, 480, max_batch=B, overlap=ov)
    sm.upload(frames)
    def run(k, log=None):
        inflight = 0
        for _ in range(k):
            if inflight == sm.depth:
                sm.finish(); inflight -= 1
                if log is not None: log.append(("f", time.perf_counter()))
            sm.submit(B, 92.0); inflight += 1
            if log is not None: log.append(("s", time.perf_counter()))
        while inflight:
            sm.finish(); inflight -= 1
            if log is not None: log.append(("f", time.perf_counter()))
    run(4); run(3)
    torch.cuda.synchronize()
    log = []
    t0 = time.perf_counter()
    run(20, log)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("overlap %s: %.1f frames/s (%.3f ms/step)" % (ov, B * 20 / dt, dt / 20 * 1e3))
    print(" ".join("%s%.2f" % (a, (b - t0) * 1e3) for a, b in log))
    del sm
