"""How much more a border tile of the quantisers costs than an interior one: the same number of pixels as 64 frames of 640x480 and as 16 frames of
1280x960 (border tiles: colour 30 of 150 vs 60 of 600 per frame, depth 46 of 150 vs 96 of 600), per-kernel time with one step in flight.
usage: python scripts/border_tile_cost.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
from linemod_pose_estimation_amd import synth, Detector
bank = synth.make_bank(60, seed=5)
for (w, h, n) in ((640, 480, 64), (1280, 960, 16), (2560, 1920, 4)):
    frames = [synth.make_scene(bank, w, h, seed=100 + f, texture=0.6)[0] for f in range(min(n, 4))]
    frames = [frames[i % len(frames)] for i in range(n)]
    det = Detector(bank, w, h, max_batch=n, overlap=False)
    det.upload(frames)
    for _ in range(3):
        det.enqueue(n, 92.0); det.collect(n, 1 << 14)
    det.set_profiling(True); det.reset_profiling()
    for _ in range(5):
        det.enqueue(n, 92.0); det.collect(n, 1 << 14)
    t = {k: v[0] / 5.0 for k, v in det.kernel_times().items() if v[1]}
    print("%dx%d x %d frames: " % (w, h, n), {k: round(v * 1e3, 1) for k, v in t.items()}, "us", flush=True)
    det.close()
