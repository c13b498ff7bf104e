#!/usr/bin/env python3
"""Print the tail of a rocprofv3 kernel trace as a timeline: start/end (us), queue, kernel."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:30]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f  dur %7.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name))
