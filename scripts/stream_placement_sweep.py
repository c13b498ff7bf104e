"""How much does the pipelined rate depend on how many OTHER streams exist in the process when the detector creates its own?
k idle streams are created first (and used once, so that their hardware queues exist), then the default workload and the 6250-template shard
are timed.  usage: python scripts/stream_placement_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import bench  # noqa: E402
from linemod_pose_estimation_amd import synth, Detector  # noqa: E402

bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(64)]
big = synth.make_bank(6250, seed=20250217)
bframes = [synth.make_scene(big, 640, 480, seed=6000 + f, row_pad=0)[0] for f in range(64)]
keep = []
x = torch.zeros(1024, device="cuda")
for k in range(0, 9):
    a = bench.secondary_line(torch, Detector, bank, frames, 64, 92.0, 150)
    b = bench.secondary_line(torch, Detector, big, bframes, 64, 92.0, 60)
    print("%d other streams alive: default %7.0f frames/s   6250-template shard %7.0f frames/s" % (k, a["value"], b["value"]), flush=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1)          # the stream has launched a kernel: its hardware queue exists
    torch.cuda.synchronize()
    keep.append(s)
