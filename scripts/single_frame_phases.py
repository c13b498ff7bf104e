"""Where the time of ONE host frame per call goes (640x480 RGB-D, 3000 templates): upload (staging + H2D), wait for the transfer,
enqueue (kernel launches), collect (sync + read-back), with pageable and pinned sources and 1..8 staging threads."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
from linemod_pose_estimation_amd.detector import PinnedArena
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, texture=0.6)[0] for f in range(8)]


def run(det, batches, n=300, warm=30, wait=True):
    ph = {k: [] for k in ("upload", "upload_wait", "enqueue", "collect", "total")}
    for i in range(warm + n):
        b = batches[i % len(batches)]
        t0 = time.perf_counter(); det.upload(b)
        t1 = time.perf_counter()
        if wait: det.upload_wait()
        t2 = time.perf_counter(); det.enqueue(1, 92.0)
        t3 = time.perf_counter(); det.collect(1)
        t4 = time.perf_counter()
        if i >= warm:
            for k, v in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)): ph[k].append(v * 1e6)
    return "  ".join("%s %.1f" % (k, np.median(v)) for k, v in ph.items())


for threads in (os.environ.get("LMX_UPLOAD_THREADS", "default"),):
    det = Detector(bank, 640, 480, max_batch=1)
    print("pageable, split phases          :", run(det, [Detector.prepare_batch([f]) for f in frames]), flush=True)
    print("pageable, no explicit wait      :", run(det, [Detector.prepare_batch([f]) for f in frames], wait=False), flush=True)
    det.close()
    det = Detector(bank, 640, 480, max_batch=1, async_input=True)
    arena = PinnedArena(8 * (640 * 480 * 5) + 4096)
    pinned = [Detector.prepare_batch([[arena.put(m) for m in f]]) for f in frames]
    print("pinned + async_input, split     :", run(det, pinned), flush=True)
    det.close()
