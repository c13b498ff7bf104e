"""One frame per call, eager five-launch chain against a context with LMX_CTX_HIPGRAPH (the chain replayed as one graph): does a graph
launch beat five kernel launches?  It does not on ROCm 7.x: resident 66 us eager, 83 us as a graph; with a host frame 117 against 185 us (the
graph path also gives up the direct-store upload).  usage (GPU box): python scripts/single_frame_graph.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
bank = synth.make_bank(3000, seed=20250215)
frames = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=0.6)[0] for f in range(8)]
def lat(fn, n=300, warm=30):
    for i in range(warm): fn(i)
    ts=[]
    for i in range(n):
        t=time.perf_counter(); fn(i); ts.append((time.perf_counter()-t)*1e6)
    ts=np.sort(ts); return "median %.1f us p10 %.1f p90 %.1f" % (np.median(ts), ts[len(ts)//10], ts[len(ts)*9//10])
for hg in (False, True):
    det = Detector(bank, 640, 480, device=0, max_batch=1, hipgraph=hg, overlap=False)
    fresh = [[np.array(s, copy=True) for s in frames[i % 8]] for i in range(64)]
    print("hipgraph", hg, "host frame :", lat(lambda i: det.match(fresh[i % 64], 92.0)))
    det.upload([frames[0]])
    def res(i):
        det.enqueue(1, 92.0); det.collect(1)
    print("hipgraph", hg, "resident   :", lat(res))
    det.close()
