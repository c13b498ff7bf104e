"""The explosive regime: threshold 50 defeats the pruning of the coarse scoring kernel, every (template, placement)
that reaches half the maximum score becomes a candidate.  2 frames per step, one lane, 3000 templates; prints the step time, the host
split of collect() (LMX_COLLECT_TRACE) and the one-core oracle's time for the same call.  Run under `rocprofv3 --kernel-trace --stats`
for the kernel side.  usage: python scripts/low_threshold_profile.py [threshold]"""
import os, sys, time
os.environ["LMX_COLLECT_TRACE"] = "1"
sys.path.insert(0, ".")
from linemod_pose_estimation_amd import synth, Detector
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, texture=0.6)[0] for f in range(2)]
det = Detector(bank, 640, 480, max_batch=2, max_candidates=1 << 21)
det.upload(frames)
for i in range(5):
    t0 = time.perf_counter()
    det.enqueue(2, thr)
    out = det.collect(2, 1 << 22)
    dt = time.perf_counter() - t0
    print("step %d: %.1f ms, %s matches per frame, %d candidates" % (i, dt * 1e3, [len(m) for m in out], det.stats()["candidates"]), flush=True)
det.set_profiling(True); det.reset_profiling()
for i in range(3):
    det.enqueue(2, thr); det.collect(2, 1 << 22)
print("kernel ms per step (HIP events):", {k: round(v[0] / 3, 3) for k, v in det.kernel_times().items() if v[1]}, flush=True)
if "--cpu" in sys.argv:
    from oracle import oracle as o
    od = o.OracleDetector(bank)
    t0 = time.perf_counter(); ref = od.match(frames[0], thr); t1 = time.perf_counter()
    print("oracle, one core, one frame at threshold %.0f: %.1f ms, %d matches (same as the device: %s)" % (thr, (t1 - t0) * 1e3, len(ref), len(ref) == len(out[0])), flush=True)
