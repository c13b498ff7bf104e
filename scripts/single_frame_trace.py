"""One resident 640x480 RGB-D frame, 3000 templates: 60 enqueue+collect calls (run under rocprofv3 --kernel-trace to see where the
77 us of a single-frame call go: scripts/trace_timeline.py prints kernels and gaps of one call)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
bank = synth.make_bank(3000, seed=20250215)
frame = synth.make_scene(bank, 640, 480, seed=3000, texture=0.6)[0]
det = Detector(bank, 640, 480, max_batch=1, hipgraph="--graph" in sys.argv)
det.upload([frame])
t = []
for i in range(60):
    t0 = time.perf_counter()
    det.enqueue(1, 92.0)
    det.collect(1)
    t.append(time.perf_counter() - t0)
print("median %.1f us" % (np.median(np.asarray(t[10:])) * 1e6))
det.close()
