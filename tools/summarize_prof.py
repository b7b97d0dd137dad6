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
avg = {}
for f in find("pmc*/**/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    acc = defaultdict(list)
    for r in rows:
        if dom is None or r["Kernel_Name"] == dom:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        v = v[len(v) // 6:]
        avg[c] = sum(v) / len(v)
        print("  %-28s avg=%.1f  (n=%d)" % (c, avg[c], len(v)))

# HBM bytes per launch.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
# counts each 128-byte read request as 64 bytes, so it is doubled; the raw request
# counters (32/64/128-byte reads, 64-byte writes) give the same total independently.
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    rd = 2.0 * avg["FETCH_SIZE"] * 1024.0
    wr = avg["WRITE_SIZE"] * 1024.0
    print("\n## HBM traffic per launch")
    print("  reads  (2 x FETCH_SIZE KiB): %.1f MB" % (rd / 1e6))
    print("  writes (WRITE_SIZE KiB)    : %.1f MB" % (wr / 1e6))
    if "TCC_EA0_RDREQ_128B_sum" in avg:
        rd_raw = (128 * avg["TCC_EA0_RDREQ_128B_sum"] + 64 * avg.get("TCC_EA0_RDREQ_64B_sum", 0)
                  + 32 * avg.get("TCC_EA0_RDREQ_32B_sum", 0))
        print("  reads  from request counters: %.1f MB" % (rd_raw / 1e6))
    if "TCC_EA0_WRREQ_sum" in avg:
        w64 = avg.get("TCC_EA0_WRREQ_64B_sum", 0)
        wr_raw = 64 * w64 + 32 * (avg["TCC_EA0_WRREQ_sum"] - w64)
        print("  writes from request counters: %.1f MB" % (wr_raw / 1e6))
    print("  total: %.1f MB" % ((rd + wr) / 1e6))
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash          # ties the figure to the kernel source it was measured on
    json.dump({"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
               "kernel": dom, "source": os.path.basename(root), "kernel_source_hash": kernel_source_hash()},
              open(os.path.join(root, "traffic.json"), "w"))
