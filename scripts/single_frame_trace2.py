"""Host-side phases (LMX_MATCH_TRACE=1) of lmx_match with a fresh pageable 640x480 RGB-D host frame per call, 3000 templates; run it under
rocprofv3 --kernel-trace for the device timeline of the same calls (scripts/trace_timeline.py).  usage: python scripts/single_frame_trace2.py [n]"""
import os, sys, time
sys.path.insert(0, ".")
os.environ["LMX_MATCH_TRACE"] = "1"
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bank = synth.make_bank(3000, seed=20250215)
frames = [[np.array(s, copy=True) for s in synth.make_scene(bank, 640, 480, seed=3000 + f, texture=0.6)[0]] for f in range(16)]
det = Detector(bank, 640, 480, max_batch=1)
prep = [Detector.prepare_batch([f]) for f in frames]
t = []
for i in range(30 + n):
    t0 = time.perf_counter()
    det.match_prepared(prep[i % 16], 92.0)
    t.append(time.perf_counter() - t0)
t = np.asarray(t[30:]) * 1e6
print("lmx_match, descriptors built once: median %.1f us  p10 %.1f  p90 %.1f" % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)), flush=True)
det.close()
