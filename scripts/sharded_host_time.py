"""Host time of ShardedMatcher.submit / .finish per 64-frame step with one rank and RCCL initialised (what every rank of `bench.py --gpus N`
pays), next to the plain Detector pipeline.  usage: python scripts/sharded_host_time.py"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from linemod_pose_estimation_amd import synth
from linemod_pose_estimation_amd.dist import ShardedMatcher
B = 64
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f)[0] for f in range(B)]
cap = int(os.environ.get("CAP", "8192"))
sm = ShardedMatcher(bank, 640, 480, max_batch=B, gather_capacity=cap)
if os.environ.get("NOCOLL"):
    sm.collective = False
    sm.recv = sm.send
sm.upload(frames)
ts, tf = [], []
inflight = 0
torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(300):
    if inflight == sm.depth:
        t0 = time.perf_counter(); sm.finish(); tf.append(time.perf_counter() - t0); inflight -= 1
    t0 = time.perf_counter(); sm.submit(B, 92.0); ts.append(time.perf_counter() - t0); inflight += 1
while inflight:
    sm.finish(); inflight -= 1
torch.cuda.synchronize()
dt = time.perf_counter() - t_all
print("cap %d nocoll %s" % (cap, bool(os.environ.get("NOCOLL"))), end=" "); print("ShardedMatcher: %.3f ms per step (%.0f frames/s); host: submit median %.1f us (p90 %.1f), finish median %.1f us (p90 %.1f)"
      % (dt / 300 * 1e3, B * 300 / dt, np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6, np.median(tf) * 1e6, np.percentile(tf, 90) * 1e6))
dist.destroy_process_group()
