"""Where does the sharded path (ShardedMatcher: export -> RCCL all-gather -> read-back -> host merge per batch) lose against the plain one?
One rank, same box: the plain pipelined rate before and after RCCL is initialised, the sharded loop with and without the collective, with
and without HIP events around the scoring kernel (what bench.py adds for its roofline).  usage: python scripts/sharded_overhead_split.py [plain_first|sharded_first]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
import torch, torch.distributed as dist
import bench
from linemod_pose_estimation_amd import synth, Detector
from linemod_pose_estimation_amd.dist import ShardedMatcher
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(64)]
def plain(tag):
    line = bench.secondary_line(torch, Detector, bank, frames, 64, 92.0, 200, breakdown=True)
    print(tag, round(line["value"]), {k: round(v,4) for k,v in line["kernel_ms_per_step"].items()}, flush=True)
def sharded(tag, collective=True, events=False):
    sm = ShardedMatcher(bank, 640, 480, max_batch=64)
    if events:
        sm.det.set_profiling("k_score_coarse")
    if not collective: sm.collective = False; sm.recv = sm.send
    sm.upload(frames)
    def run(k):
        infl=0
        for _ in range(k):
            if infl == sm.depth: sm.finish(); infl -= 1
            sm.submit(64, 92.0); infl += 1
        while infl: sm.finish(); infl -= 1
    run(60)
    for rep in range(4):   # consecutive windows of 200 steps: does the rate settle?
        torch.cuda.synchronize(); t=time.perf_counter(); run(200); torch.cuda.synchronize(); dt=time.perf_counter()-t
        print(tag, "window", rep, round(64*200/dt), flush=True)
order = sys.argv[1] if len(sys.argv) > 1 else "plain_first"
if order == "plain_first":
    plain("plain before init")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
if order == "plain_first":
    plain("plain after nccl init")
if order == "collective_first":
    t = torch.zeros(64 * 1024, dtype=torch.uint8, device="cuda"); out = torch.empty_like(t)
    dist.all_gather_into_tensor(out, t); dist.barrier(); torch.cuda.synchronize()
    print("(one all_gather + barrier before any matcher exists)", flush=True)
if order == "throwaway_first":
    sm0 = ShardedMatcher(bank, 640, 480, max_batch=64); sm0.upload(frames)
    for _ in range(8): sm0.step(64, 92.0)
    sm0.det.close(); del sm0
    print("(a first ShardedMatcher was created, stepped 8 times and closed)", flush=True)
sharded("sharded, all_gather")
sharded("sharded, all_gather, events around the scoring kernel", events=True)
sharded("sharded, no all_gather", collective=False)
plain("plain again")
