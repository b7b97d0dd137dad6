#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a short text summary:
per-kernel average duration from the kernel trace, and per-launch averages of
every PMC counter for the dominant kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))


print("# profile summary for", os.path.basename(root))
for f in find("kt/**/*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print("  %-90s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
            r.get("Name", "")[:90], r.get("Calls"), r.get("AverageNs"), r.get("MinNs"), r.get("MaxNs"),
            r.get("Percentage")))
dom = None
for f in find("kt/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    dur = defaultdict(list)
    meta = {}
    for r in rows:
        k = r["Kernel_Name"]
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
    dom = max(dur, key=lambda k: sum(dur[k]))
    d = sorted(dur[dom][len(dur[dom]) // 6:])      # drop warm-up launches
    print("\n## dominant kernel:", dom[:120])
    print("  launches=%d  median_ns=%d  mean_ns=%.0f  p10=%d p90=%d  vgpr/sgpr/lds/wg/grid=%s" % (
        len(dur[dom]), d[len(d) // 2], sum(d) / len(d), d[len(d) // 10], d[9 * len(d) // 10], meta[dom]))
print("\n## PMC counters, average per launch of the dominant kernel")
for f in find("pmc*/**/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    acc = defaultdict(list)
    for r in rows:
        if dom is None or r["Kernel_Name"] == dom:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        v = v[len(v) // 6:]
        print("  %-28s avg=%.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
