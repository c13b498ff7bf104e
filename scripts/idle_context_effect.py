"""Does an IDLE context in the process slow another context's pipelined matching?  (bench.py's later lines dropped by 4-8 % once an earlier
block left a context in lmx_ctx_acquire's cache.)  One box: the 6250-template shard workload alone, then with idle contexts of several kinds
alive in the process.  usage: python scripts/idle_context_effect.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import bench  # noqa: E402
from linemod_pose_estimation_amd import synth, Detector  # noqa: E402

bank = synth.make_bank(6250, seed=20250217)
small = synth.make_bank(50, seed=3)
frames = [synth.make_scene(bank, 640, 480, seed=6000 + f, row_pad=0)[0] for f in range(64)]


def rate(tag):
    line = bench.secondary_line(torch, Detector, bank, frames, 64, 92.0, 60)
    print("%-70s %7.0f frames/s  %.4f ms/step" % (tag, line["value"], line["ms_per_step"]), flush=True)


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
rate("alone")
rate("alone again")
idle = Detector(small, 640, 480, device=0, max_batch=1, overlap=False)
rate("+ one idle context (1 lane, 50 templates, never used)")
idle.match(frames[0], 92.0)
rate("+ the same after it matched one frame")
if quick:
    idle.close()
    sys.exit(0)
idle2 = Detector(small, 640, 480, device=0, max_batch=64, overlap=True)
idle2.upload(frames[:64])
idle2.enqueue(64, 92.0); idle2.collect(64)
rate("+ a second idle context (3 lanes, 64-frame buffers, used once)")
idle.close(); idle2.close()
rate("after closing both")

# the same question for plain device allocations of various sizes made before the context exists (is it WHERE its buffers land?)
for mb in (16, 50, 64, 100, 256):
    pad = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    rate("+ a %d MB allocation alive" % mb)
    del pad
    torch.cuda.empty_cache()
rate("alone, at the end")
