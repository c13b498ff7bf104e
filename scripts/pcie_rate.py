"""Single-frame latency at the reference's boundary (one host frame per call, like the service node: ..._service.cpp:324-344) and
the batch call, for the forms the library offers."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, texture=0.6)[0] for f in range(32)]


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t = np.asarray(t) * 1e6
    return "median %.1f us  p10 %.1f  p90 %.1f" % (np.median(t), np.percentile(t, 10), np.percentile(t, 90))


for name, kw in (("eager", {}), ("hipgraph", {"hipgraph": True})):
    det = Detector(bank, 640, 480, max_batch=1, **kw)
    k = [0]

    def host_match():
        k[0] = (k[0] + 1) % len(frames)
        return det.match(frames[k[0]], 92.0)

    print("lmx_match, one fresh host frame per call, %-9s %s" % (name + ":", timeit(host_match)), flush=True)
    det.upload([frames[0]])

    def resident():
        det.enqueue(1, 92.0)
        return det.collect(1)

    print("enqueue + collect, resident frame,        %-9s %s" % (name + ":", timeit(resident)), flush=True)
    det.close()
det = Detector(bank, 640, 480, max_batch=32)
print("lmx_match_batch, 32 fresh host frames per call (synchronous): %s per call" % timeit(lambda: det.match_batch(frames, 92.0), n=30, warm=5), flush=True)
