import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000+f)[0] for f in range(32)]
det = Detector(bank, 640, 480, max_batch=32)
det.upload(frames)
for i in range(8):
    t0=time.perf_counter(); det.enqueue(32, 92.0); t1=time.perf_counter(); out = det.collect(32); t2=time.perf_counter()
    print("enqueue %.1f us collect %.1f us" % ((t1-t0)*1e6, (t2-t1)*1e6), file=sys.stderr)
