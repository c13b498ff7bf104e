#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel and write profiles/traffic.json
for bench.py's `roofline.traffic`.  Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads, so it is doubled.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel> <frames> <templates> <out.json>"""
import collections
import csv
import json
import re
import statistics
import sys


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        d[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    return d


def main():
    fpath, wpath, kernel, frames, templates, out = sys.argv[1:7]
    f, w = agg(fpath, "FETCH_SIZE"), agg(wpath, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(f) | set(w)):
        fk = statistics.median(f[k]) if f.get(k) else 0.0
        wk = statistics.median(w[k]) if w.get(k) else 0.0
        rows[k] = {"launches_seen": len(f.get(k, [])), "FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk,
                   "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
        print("%-28s n=%3d FETCH %10.1f KiB  WRITE %10.1f KiB  -> %8.2f MB/launch (fetch doubled)" %
              (k, rows[k]["launches_seen"], fk, wk, rows[k]["hbm_bytes_per_launch"] / 1e6))
    json.dump({"kernel": kernel, "frames": int(frames), "templates": int(templates),
               "hbm_bytes_per_launch": rows[kernel]["hbm_bytes_per_launch"],
               "note": "median over launches; (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes", "all_kernels": rows},
              open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
