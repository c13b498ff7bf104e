"""Latency of ONE frame per call, the reference's own pattern (..._service.cpp:339-344): lmx_match with a fresh pageable 640x480 RGB-D
host frame (3000 templates), and enqueue + collect on a resident frame.  LMX_NO_SMALL_CHAIN=1 gives the eight-launch chain for an A/B.
usage: python scripts/single_frame_latency.py [n]   (run under rocprofv3 --kernel-trace for the device timeline)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bank = synth.make_bank(3000, seed=20250215)
frames = [[np.array(s, copy=True) for s in synth.make_scene(bank, 640, 480, seed=3000 + f, texture=0.6)[0]] for f in range(16)]
det = Detector(bank, 640, 480, max_batch=1)


def lat(fn, n, warm=30):
    t = []
    for i in range(warm + n):
        t0 = time.perf_counter()
        fn(i)
        t.append(time.perf_counter() - t0)
    t = np.asarray(t[warm:]) * 1e6
    return "median %.1f us  p10 %.1f  p90 %.1f" % (np.median(t), np.percentile(t, 10), np.percentile(t, 90))


prep = [Detector.prepare_batch([f]) for f in frames]
print("lmx_match, fresh pageable host frame per call :", lat(lambda i: det.match(frames[i % 16], 92.0), n), flush=True)
print("   the same, lmx_image descriptors built once  :", lat(lambda i: det.match_prepared(prep[i % 16], 92.0), n), flush=True)
det.upload([frames[0]])


def resident(i):
    det.enqueue(1, 92.0)
    det.collect(1)


print("enqueue + collect, resident frame             :", lat(resident, n), flush=True)
cg = synth.make_bank(3000, modalities=("ColorGradient",), seed=20250214)
d2 = Detector(cg, 640, 480, max_batch=1)
print("lmx_match, ColorGradient only (config 0 shape) :", lat(lambda i: d2.match(frames[i % 16][:1], 92.0), n), flush=True)
