import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
from linemod_pose_estimation_amd import synth
from linemod_pose_estimation_amd.dist import ShardedMatcher
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
if len(sys.argv) > 1 and sys.argv[1] == "pg":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
B=64
bank = synth.make_bank(3000, modalities=("ColorGradient", "DepthNormal"), T=(5, 8), seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(B)]
sm = ShardedMatcher(bank, 640, 480, max_batch=B)
sm.upload(frames)
def run(k, acc):
    inflight=0
    for _ in range(k):
        if inflight == sm.depth:
            t=time.perf_counter(); sm.finish(); acc[1]+=time.perf_counter()-t; inflight-=1
        t=time.perf_counter(); sm.submit(B, 92.0); acc[0]+=time.perf_counter()-t; inflight+=1
    while inflight:
        t=time.perf_counter(); sm.finish(); acc[1]+=time.perf_counter()-t; inflight-=1
run(48,[0,0])
torch.cuda.synchronize()
acc=[0,0]; t0=time.perf_counter(); run(40,acc); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("collective" , sm.collective, "frames/s %.0f  per step: total %.3f ms, host in submit %.3f ms, host in finish (incl. waiting) %.3f ms" % (B*40/dt, dt/40*1e3, acc[0]/40*1e3, acc[1]/40*1e3))
