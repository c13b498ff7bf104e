#!/usr/bin/env python3
"""Summarise one scripts/profile_r02.sh run: per (device kernel, grid size) medians of every PMC counter, the kernel's
duration from the un-instrumented one-lane kernel trace of the same workload, and the derived utilisations that bench.py
reports as `roofline`.

Units / corrections (/opt/skills/guides/MI355X_MICROARCH.md): FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts
half the bytes of wide coalesced reads, so it is doubled; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles; a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (4 for some, measured by
scripts/microbench/valu_issue -> profiles/*valu_issue*.txt and passed in as a weight table below); the scalar unit is shared
by the 4 SIMDs of a CU and issues one instruction per cycle.
usage: pmc_summary2.py <gpurun_out/tag dir> [--json out.json]"""
import collections
import csv
import glob
import json
import os
import re
import statistics
import sys

N_CU, N_SIMD = 256, 1024
CLOCK_GHZ = 2.4   # max shader clock; GRBM_GUI_ACTIVE (when collected) gives the effective one


def kname(s):
    m = re.search(r"(k_[a-z0-9_]+)", s)
    return m.group(1) if m else s.split("(")[0][:48]


def main():
    d = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else os.path.join(d, "summary.json")
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            key = (kname(r["Kernel_Name"]), int(r["Grid_Size"]))
            cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[key] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]), "wg": int(r["Workgroup_Size"])}
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "kt1", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur[(kname(r["Kernel_Name"]), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)  # us
    rows = {}
    for key in sorted(cnt):
        c = {k: statistics.median(v) for k, v in cnt[key].items()}
        n = max(len(v) for v in cnt[key].values())
        us = statistics.median(dur[key]) if dur.get(key) else None
        row = {"kernel": key[0], "grid_size": key[1], "launches_seen": n, "duration_us_one_lane_trace": us, "counters": c, **meta[key]}
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            row["hbm_bytes_per_launch"] = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        if us:
            cyc = us * 1e-6 * CLOCK_GHZ * 1e9   # cycles the kernel had on every SIMD / scalar unit
            if "SQ_INSTS_VALU" in c:
                row["valu_issue_frac_2cyc"] = c["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * cyc)
            if "SQ_INSTS_SALU" in c:
                row["salu_issue_frac_1cyc"] = c["SQ_INSTS_SALU"] / (N_CU * cyc)
            if "SQ_WAVE_CYCLES" in c:
                row["avg_waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4.0 / (N_SIMD * cyc)
            if "SQ_ACTIVE_INST_VALU" in c:
                row["active_inst_valu_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cyc)
            if "hbm_bytes_per_launch" in row:
                row["hbm_gbs"] = row["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9
            if "TCP_TCC_READ_REQ_sum" in c:
                row["l2_to_l1_tbs_128B"] = c["TCP_TCC_READ_REQ_sum"] * 128.0 / (us * 1e-6) / 1e12
        if c.get("SQ_WAVES"):
            w = c["SQ_WAVES"]
            row["per_wave"] = {k.replace("SQ_INSTS_", "").lower(): c[k] / w for k in c if k.startswith("SQ_INSTS_")}
            if "SQ_WAVE_CYCLES" in c:
                row["per_wave"]["lifetime_cycles"] = c["SQ_WAVE_CYCLES"] * 4.0 / w
        rows["%s@%d" % key] = row
    workload = {}
    try:  # the bench line of the same script run names the workload the counters belong to
        cfg = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])["config"]
        workload = {"frames": cfg["frames_per_step"], "templates": cfg["templates_per_gpu"], "threshold": cfg["threshold"], "texture": cfg["scene_texture"], "score_no_prune": bool(cfg.get("score_no_prune", False))}
    except (OSError, ValueError, KeyError, IndexError):
        pass
    json.dump({"source": os.path.basename(os.path.abspath(d)), "workload": workload, "clock_ghz_assumed": CLOCK_GHZ,
               "note": "per launch medians; PMC passes ran one lane (--no-overlap); durations from the un-instrumented one-lane kernel trace",
               "kernels": rows}, open(out_json, "w"), indent=1)
    for k, r in rows.items():
        if not r["kernel"].startswith("k_") or r["counters"].get("SQ_WAVES", 1e9) < 512:
            continue
        print("%s  (n=%d, wg %d, vgpr %d, sgpr %d, lds %d)  duration %s us" % (k, r["launches_seen"], r["wg"], r["vgpr"], r["sgpr"], r["lds"],
              "%.1f" % r["duration_us_one_lane_trace"] if r["duration_us_one_lane_trace"] else "?"))
        for name in ("valu_issue_frac_2cyc", "salu_issue_frac_1cyc", "avg_waves_per_simd", "active_inst_valu_frac", "hbm_bytes_per_launch", "hbm_gbs", "l2_to_l1_tbs_128B"):
            if name in r:
                print("    %-28s %.4g" % (name, r[name]))
        if "per_wave" in r:
            print("    per wave: " + "  ".join("%s %.0f" % kv for kv in sorted(r["per_wave"].items())))
        print("    " + "  ".join("%s=%.4g" % kv for kv in sorted(r["counters"].items())))


if __name__ == "__main__":
    main()
