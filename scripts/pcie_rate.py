"""PCIe-inclusive rate: host frames through lmx_match_batch (pageable -> pinned staging -> H2D -> kernels -> D2H)."""
import sys, time
sys.path.insert(0, ".")
from linemod_pose_estimation_amd import synth, Detector
bank = synth.make_bank(3000, seed=20250215)
B = 32
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f)[0] for f in range(B)]
det = Detector(bank, 640, 480, max_batch=B)
for _ in range(3):
    det.match_batch(frames, 92.0)
t = time.perf_counter()
K = 10
for _ in range(K):
    det.match_batch(frames, 92.0)
dt = (time.perf_counter() - t) / K
print("match_batch with host frames: %.3f ms per %d frames = %.0f frames/s (%.1f us/frame)" % (dt * 1e3, B, B / dt, dt / B * 1e6))
t = time.perf_counter()
for _ in range(K):
    det.match(frames[0], 92.0)
dt1 = (time.perf_counter() - t) / K
print("single-frame lmx_match with host frame: %.1f us" % (dt1 * 1e6))
