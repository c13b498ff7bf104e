#!/usr/bin/env python3
"""Readable digest of one bench.py JSON line.  Usage: python scripts/print_bench.py <file with the JSON line>"""
import json
import sys


def g(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def main():
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e = d.get("extra", {})
    print("value %.0f %s  %.3f ms/step  n_gpus %s" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
    print("kernels ms/step:", {k: round(v, 4) for k, v in d.get("kernel_ms_per_step", {}).items()})
    r = d.get("roofline", {})
    print("roofline: %s %s frac %.3f  excl %.4f ms" % (r.get("bound"), r.get("kernel"), r.get("frac", 0), r.get("avg_launch_ms_exclusive", 0)))
    print("host_frames %.0f  pinned %.0f  raw camera %.0f (%s)" % (g(d, "host_frames", "value", default=0), g(d, "host_frames_pinned", "value", default=0),
                                                                  g(d, "host_frames_raw_camera", "value", default=0), g(d, "host_frames_raw_camera", "error", default="ok")))
    if "cpu_baseline" in d:
        print("cpu_baseline", d["cpu_baseline"].get("value"), "all cores", g(d, "cpu_baseline_all_cores", "value"))
    for k in ("busy_scene", "low_threshold", "config0_cg_only", "config3_1280x960_2x3000", "config3_mesh_two_objects_1280x960", "config4_shard_6250",
              "config5_shard_6250_hipgraph_lanes", "mesh_bank", "group_1_member_rccl"):
        v = e.get(k)
        if not isinstance(v, dict):
            continue
        if "error" in v:
            print("%-38s ERROR %s" % (k, v["error"]))
            continue
        print("%-38s %9.0f frames/s  %7.3f ms/step  cand/frame %8.1f  matches/frame %6.1f" % (k, v.get("value", 0), v.get("ms_per_step", 0), v.get("coarse_candidates_per_frame", 0),
                                                                                             v.get("matches_per_frame", 0)))
    print("config0 resident %.0f  raw mono %.0f" % (g(e, "config0_cg_only", "resident", "value", default=0), g(e, "config0_cg_only", "raw_mono_752x480", "value", default=0)))
    y = g(e, "config0_cg_only", "yml_request_flow", default={})
    if y:
        print("yml request flow:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in y.items() if k not in ("note", "warm_request_us")},
              "warm request us", g(y, "warm_request_us", "median"))
    mb = e.get("mesh_bank", {})
    print("mesh thr85 %.0f  clusters %s  trainer %s" % (g(mb, "threshold_85", "value", default=0), {k: round(v["value"]) for k, v in mb.get("collect_clusters", {}).items() if isinstance(v, dict)},
                                                          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in mb.get("trainer", {}).items() if k != "note"}))
    print("group8", {k: (v.get("host_us_per_batch") if isinstance(v, dict) else None) for k, v in e.get("group_8_members_one_gpu", {}).items() if k != "note"})
    print("single frame", {k: v for k, v in e.get("single_frame_latency", {}).items() if k != "note"})


if __name__ == "__main__":
    main()
