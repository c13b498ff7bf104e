"""The pipelined headline workload with the SYSTEM HIP runtime (a C++ caller's situation: liblmx.so -> /opt/rocm's libamdhip64) or with the
runtime PyTorch-ROCm bundles (the bench's situation: torch is imported first, liblmx.so shares its libamdhip64).
usage: python scripts/runtime_ab.py system|torch"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
which = sys.argv[1] if len(sys.argv) > 1 else "system"
if which == "system":
    sys.modules["torch"] = None          # `import torch` inside the package raises ImportError: liblmx.so loads the system runtime
from linemod_pose_estimation_amd import synth, Detector  # noqa: E402
import bench  # noqa: E402

bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(64)]
det = Detector(bank, 640, 480, device=0, max_batch=64, overlap=True)
det.upload(frames)
bench.run_pipelined(det, 60, 64, 92.0)
det.sync()
for rep in range(3):
    t = time.perf_counter()
    bench.run_pipelined(det, 300, 64, 92.0)
    det.sync()
    dt = time.perf_counter() - t
    print("%s runtime: %7.0f frames/s  %.4f ms/step" % (which, 64 * 300 / dt, dt / 300 * 1e3), flush=True)
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print("   loaded:", libs)
