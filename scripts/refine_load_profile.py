#!/usr/bin/env python3
"""Where the step goes when MANY candidates pass the coarse level (rendered banks: neighbouring views of one object).
Per workload: frames/s pipelined, candidates and matches per frame, per-kernel ms with one step in flight, and k_refine's cost per
candidate.  Usage (GPU box):  python scripts/refine_load_profile.py [out.txt]
"""
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import bench  # noqa: E402
from linemod_pose_estimation_amd import Detector, meshsynth as ms, synth  # noqa: E402


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 and sys.argv[1] != "-" else sys.stdout
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"   # three workloads: A/B runs of library variants (LMX_SO_PATH) on one box
    chip, cpu, views = ms.load_mesh("memoryChip2"), ms.load_mesh("cpu_binary"), ms.view_grid()
    mbank = ms.load_bank("memoryChip2")[0]
    bank2 = ms.load_banks(("memoryChip2", "cpu_binary"))[0]
    d1 = [ms.make_scene(chip, views, seed=7000 + f, n_instances=3, other_tri=cpu, n_other=2)[0] for f in range(16)]
    f1 = [d1[f % 16] for f in range(64)]
    d2 = [ms.make_scene(chip, views, 1280, 960, seed=7100 + f, n_instances=4, other_tri=cpu, n_other=4, other_class="cpu_binary")[0] for f in range(4)]
    f2 = [d2[f % 4] for f in range(16)]
    sbank = synth.make_bank(3000, seed=20250215)
    fs = [synth.make_scene(sbank, 640, 480, seed=3000 + f, row_pad=0, texture=1.0)[0] for f in range(64)]
    cases = [("synthetic bank, busy scenes, thr 92", sbank, fs, 64, 92.0, 640, 480),
             ("synthetic bank, busy scenes, thr 85", sbank, fs, 64, 85.0, 640, 480),
             ("mesh bank (chip), thr 92", mbank, f1, 64, 92.0, 640, 480),
             ("mesh bank (chip), thr 88", mbank, f1, 64, 88.0, 640, 480),
             ("mesh bank (chip), thr 85", mbank, f1, 64, 85.0, 640, 480),
             ("two rendered banks 1280x960, thr 92", bank2, f2, 16, 92.0, 1280, 960),
             ("two rendered banks 1280x960, thr 88", bank2, f2, 16, 88.0, 1280, 960)]
    if quick:
        cases = [c for c in cases if c[0] in ("synthetic bank, busy scenes, thr 92", "mesh bank (chip), thr 85", "two rendered banks 1280x960, thr 92")]
    for name, bank, frames, B, thr, W, H in cases:
        line = bench.secondary_line(torch, Detector, bank, frames, B, thr, 30, width=W, height=H, breakdown=True, max_candidates=1 << 19, collect_cap=1 << 20)
        k = line["kernel_ms_per_step"]
        cands = line["coarse_candidates_per_frame"] * B
        print("%-40s %8.0f frames/s  %6.3f ms/step  cand/frame %7.0f  matches/frame %6.1f" % (name, line["value"], line["ms_per_step"], line["coarse_candidates_per_frame"],
                                                                                              line["matches_per_frame"]), file=out)
        print("    kernels (one step in flight, ms): " + "  ".join("%s %.3f" % (a, b) for a, b in sorted(k.items(), key=lambda t: -t[1])), file=out)
        rk = [v for a, v in k.items() if "refine" in a.lower()]
        if rk and cands:
            print("    k_refine: %.1f ns per candidate" % (rk[0] * 1e6 / cands), file=out)
        out.flush()


if __name__ == "__main__":
    main()
