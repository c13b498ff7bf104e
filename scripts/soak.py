"""Soak: 600 pipelined steps (three lanes, fresh host frames every step, 8 distinct batches of 6 frames at 320x240 and 2 batches of
8 frames at 640x480) with every step's matches compared with the oracle's."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
from oracle import oracle as o

def same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for k in ("x", "y", "similarity", "template_id", "class_index"):
        assert np.array_equal(a[k], b[k]), k

for (W, H, B, nb, ntmpl, steps, tex) in ((320, 240, 6, 8, 120, 400, 0.6), (640, 480, 8, 2, 300, 200, 1.0)):
    bank = synth.make_bank(ntmpl, seed=77, size_range=(30.0, min(W, H) * 0.4))
    od = o.OracleDetector(bank)
    batches = [[synth.make_scene(bank, W, H, seed=9000 + 10 * k + f, texture=tex)[0] for f in range(B)] for k in range(nb)]
    refs = [[od.match(fr, 80.0) for fr in b] for b in batches]
    for graph in (False, True):
        det = Detector(bank, W, H, max_batch=B, overlap=True, hipgraph=graph, max_candidates=1 << 18)
        prepared = [Detector.prepare_batch(b) for b in batches]
        inflight, checked, t0 = [], 0, time.time()
        for i in range(steps):
            if len(inflight) == 3:
                k = inflight.pop(0)
                got = det.collect(B)
                for f in range(B): same(got[f], refs[k][f])
                checked += B
            k = i % nb
            det.upload(prepared[k]); det.enqueue(B, 80.0); inflight.append(k)
        while inflight:
            k = inflight.pop(0)
            got = det.collect(B)
            for f in range(B): same(got[f], refs[k][f])
            checked += B
        det.close()
        print("%dx%d graph=%s: %d frames checked, %d matches per batch, %.1f s" % (W, H, graph, checked, sum(len(r) for r in refs[0]), time.time() - t0), flush=True)
print("soak ok")
