#!/usr/bin/env python3
"""Copy the judged parts of a tools/profile.sh output directory into profiles/:
kernel stats table, the condensed summary, and the HBM-bytes-per-launch entry of
profiles/hbm_traffic.json (keyed the way bench.py looks it up).

usage: python tools/collect_profile.py gpurun_out/prof_<tag> <profiles name> <traffic key> "<command note>"
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, name, key, note = sys.argv[1:5]
stats = sorted(glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True))
shutil.copy(stats[0], os.path.join(ROOT, "profiles", name + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", name + "_summary.txt"))
tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
table = json.load(open(tpath))
ent = json.load(open(os.path.join(src, "traffic.json")))
ent["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950 counts a "
               "128-byte read request as 64 bytes); cross-checked against TCC_EA0_RDREQ_{32B,64B,128B} and "
               "TCC_EA0_WRREQ{,_64B}; command: " + note)
table[key] = ent
json.dump(table, open(tpath, "w"), indent=1)
print("profiles/%s_* written, hbm_traffic.json[%s] = %.1f MB" % (name, key, ent["hbm_bytes_per_launch"] / 1e6))
