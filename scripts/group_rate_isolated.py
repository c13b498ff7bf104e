"""The C++ device group with one member and RCCL in the loop, in a process of its own (bench.py measures it late in a long process, after
dozens of contexts): frames/s of the group against the plain detector, in either order.  usage: python scripts/group_rate_isolated.py group_first|plain_first"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
from linemod_pose_estimation_amd import synth, Detector
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(64)]
order = sys.argv[1]
if order == "plain_first":
    l = bench.secondary_line(torch, Detector, bank, frames, 64, 92.0, 200); print("plain", round(l["value"]), flush=True)
for r in range(2):
    g = bench.group_line(torch, bank, frames, 64, 92.0, 200, 1, "rccl"); print("group_1_member_rccl", round(g["value"]), g.get("host_us_per_batch"), flush=True)
l = bench.secondary_line(torch, Detector, bank, frames, 64, 92.0, 200); print("plain", round(l["value"]), flush=True)
