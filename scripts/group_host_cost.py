"""Host-side cost of driving a C++ device group (csrc/lmx_group.cpp) with 1..8 members from ONE process, per phase
(LMX_GROUP_TRACE=1 prints the per-phase averages when a group is destroyed).  On a one-GPU box the members share the device, so
only the host columns carry over to a multi-GPU node.  usage: python scripts/group_host_cost.py [frames_per_batch]"""
import os
import sys
import time

os.environ["LMX_GROUP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from linemod_pose_estimation_amd import synth, Detector
from linemod_pose_estimation_amd.dist import DeviceGroup

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ONLY = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else None   # members,hipgraph,fresh
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f)[0] for f in range(B)]
host = [Detector.prepare_batch([[np.array(s, copy=True) for s in fr] for fr in frames]) for _ in range(3)]
for members, coll in ((1, "rccl"), (2, "peer_copy"), (4, "peer_copy"), (8, "peer_copy")):
    for hipgraph in (False, True):
        for fresh in (False, True):
            if ONLY and ONLY != (members, int(hipgraph), int(fresh)):
                continue
            g = DeviceGroup(bank, 640, 480, members, devices=[0] * members, max_batch=B, collective=coll, hipgraph=hipgraph)
            g.upload(host[0])
            inflight, t_host = 0, []
            for i in range(40):
                if inflight == g.depth:
                    g.finish(B)
                    inflight -= 1
                t0 = time.perf_counter()
                if fresh:
                    g.upload(host[i % 3])
                g.submit(B, 92.0)
                if i >= 16:
                    t_host.append(time.perf_counter() - t0)
                inflight += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while inflight:
                g.finish(B)
                inflight -= 1
            print("members %d %-9s hipgraph %d fresh_host_frames %d: upload+submit %.0f us per batch (median)" % (members, coll, hipgraph, fresh, np.median(t_host) * 1e6), flush=True)
            sys.stderr.flush()
            g.close()
