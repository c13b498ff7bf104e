"""k_refine's time on the workloads where it matters -- the default bank on busy scenes, the rendered chip bank at threshold 85, the two rendered
objects at 1280x960 -- for whatever liblmx.so LMX_SO_PATH names (scripts/build_variants.py refine8 builds the kernel as it is and a TIMING build
in the shape of an 8-response-word image; results of the latter are wrong).  usage: LMX_SO_PATH=... python scripts/refine_word_experiment.py"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
import bench
from linemod_pose_estimation_amd import synth, Detector
from linemod_pose_estimation_amd import meshsynth as ms
tag = os.path.basename(os.environ.get("LMX_SO_PATH", "liblmx.so"))
def show(name, ln):
    k = ln["kernel_ms_per_step"]
    print("%-22s %-34s %8.0f frames/s  %.3f ms/step  refine %.4f  score %.4f  candidates/frame %.0f" % (tag, name, ln["value"], ln["ms_per_step"], k.get("k_refine", 0), k.get("k_score_coarse", 0),
                                                                                                  ln["coarse_candidates_per_frame"]), flush=True)
bank = synth.make_bank(3000, modalities=("ColorGradient", "DepthNormal"), T=(5, 8), seed=20250215)
busy = [synth.make_scene(bank, 640, 480, seed=3000 + f, row_pad=0, texture=1.0)[0] for f in range(64)]
show("busy scene, thr 92", bench.secondary_line(torch, Detector, bank, busy, 64, 92.0, 40, breakdown=True))
show("busy scene, thr 85", bench.secondary_line(torch, Detector, bank, busy, 64, 85.0, 20, breakdown=True, max_candidates=1 << 17, collect_cap=1 << 21))
mbank, _, _, _ = ms.load_bank("memoryChip2")
chip, cpu_mesh, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
distinct = [ms.make_scene(chip, views, seed=7000 + f, n_instances=3, other_tri=cpu_mesh, n_other=2, texture=0.6)[0] for f in range(16)]
mframes = [distinct[f % 16] for f in range(64)]
show("chip bank, thr 85", bench.secondary_line(torch, Detector, mbank, mframes, 64, 85.0, 20, breakdown=True, max_candidates=1 << 16, collect_cap=1 << 21))
bank2, _ = ms.load_banks(("memoryChip2", "cpu_binary"))
d2 = [ms.make_scene(chip, views, 1280, 960, seed=7100 + f, n_instances=4, other_tri=cpu_mesh, n_other=4, other_class="cpu_binary", texture=0.6)[0] for f in range(4)]
show("two objects 1280x960, thr 92", bench.secondary_line(torch, Detector, bank2, [d2[f % 4] for f in range(16)], 16, 92.0, 20, width=1280, height=960, breakdown=True, collect_cap=1 << 21))
