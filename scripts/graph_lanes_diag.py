#!/usr/bin/env python3
"""Diagnosis of LMX_CTX_HIPGRAPH | LMX_CTX_OVERLAP (VERDICT r1 "what's weak" 10): concurrent hipGraph replays on the device
lanes at BASELINE config 5's per-GPU shape (64 frames x the 6250-template shard, rank 3 of 8, of the 50k bank).

Reference = the same shard context run eagerly on one lane (itself covered against the oracle by the -m gpu tests).  Every
step of a pipelined run (all output slots in flight) is compared record for record; LMX_DEBUG_COLLECT=1 makes collect()
compare each slot's pinned host mirror with the device-side slot, LMX_GRAPH_DOT=<dir> dumps the captured graphs.
usage: graph_lanes_diag.py [steps] [templates_total] [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LMX_DEBUG_COLLECT", "1")

from linemod_pose_estimation_amd import Detector, synth  # noqa: E402


def differs(a, b):
    if len(a) != len(b):
        return "count %d != %d" % (len(a), len(b))
    for k in ("x", "y", "similarity", "template_id", "class_index"):
        if not np.array_equal(a[k], b[k]):
            return "field %s" % k
    return None


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    n_total = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    bank = synth.make_bank(n_total, seed=20250217)
    frames = [synth.make_scene(bank, 640, 480, seed=3100 + f)[0] for f in range(B)]
    rank, world = 3, 8
    eager = Detector(bank, 640, 480, max_batch=B, shard_rank=rank, shard_world=world)
    eager.upload(frames)
    eager.enqueue(B, 92.0)
    ref = eager.collect(B)
    print("reference: %d matches over %d frames, %d candidates" % (sum(len(r) for r in ref), B, eager.stats()["candidates"]), flush=True)
    eager.close()
    for name, kw in (("graph, one lane", dict(hipgraph=True)), ("graph + lanes", dict(hipgraph=True, overlap=True)), ("eager lanes", dict(overlap=True))):
        det = Detector(bank, 640, 480, max_batch=B, shard_rank=rank, shard_world=world, **kw)
        det.upload(frames)
        bad, inflight, t0 = 0, 0, time.perf_counter()
        checked = 0

        def take():
            nonlocal bad, checked
            got = det.collect(B)
            for f in range(B):
                d = differs(got[f], ref[f])
                if d:
                    bad += 1
                    if bad <= 10:
                        print("  %s: step result %d frame %d differs: %s" % (name, checked, f, d), flush=True)
            checked += 1
        for _ in range(steps):
            if inflight == det.max_outstanding:
                take()
                inflight -= 1
            det.enqueue(B, 92.0)
            inflight += 1
        while inflight:
            take()
            inflight -= 1
        dt = time.perf_counter() - t0
        print("%-16s depth %d: %d steps, %d differing frames, %.3f ms/step (includes the comparisons)" % (name, det.max_outstanding, steps, bad, dt / steps * 1e3), flush=True)
        det.close()


if __name__ == "__main__":
    main()
