"""The one-frame call as a 1 Hz node issues it (/root/reference's detect loop sleeps between requests): lmx_match with a fresh pageable host frame after the
process has been idle for a while -- helper thread asleep, queues idle, clocks down.  usage: python scripts/single_frame_cold.py [idle_ms ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
idles = [float(a) for a in sys.argv[1:]] or [0.0, 1.0, 20.0, 200.0, 1000.0]
bank = synth.make_bank(3000, seed=20250215)
frames = [[np.array(s, copy=True) for s in synth.make_scene(bank, 640, 480, seed=3000 + f, texture=0.6)[0]] for f in range(8)]
det = Detector(bank, 640, 480, max_batch=1)
prep = [Detector.prepare_batch([f]) for f in frames]
for i in range(30):
    det.match_prepared(prep[i % 8], 92.0)
for idle in idles:
    n = 200 if idle <= 1.0 else (40 if idle <= 20.0 else 12)
    t = []
    for i in range(n):
        if idle > 0:
            time.sleep(idle * 1e-3)
        t0 = time.perf_counter()
        det.match_prepared(prep[i % 8], 92.0)
        t.append(time.perf_counter() - t0)
    t = np.asarray(t) * 1e6
    print("idle %7.1f ms between calls: median %.1f us  p10 %.1f  p90 %.1f  (%d calls)" % (idle, np.median(t), np.percentile(t, 10), np.percentile(t, 90), n), flush=True)
det.close()
