#!/usr/bin/env python3
"""Step() of the resident host driver at 1M groups x 5 peers against the kernel
alone and against BatchedDriver::Step (which moves the whole image both ways).
Run on the MI355X box:  python tools/resident_probe.py [out.json]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_lib

l = host_lib.lib()
l.rh_resident_bench.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_double)]
rows = []
G, n = 1 << 20, 5
for dirty, steps, with_batched in ((0, 200, 1), (100, 200, 0), (1000, 200, 0), (10000, 100, 0), (100000, 30, 0)):
    out = (C.c_double * 6)()
    rc = l.rh_resident_bench(G, n, dirty, steps, 0, with_batched, out)
    assert rc == 0, rc
    row = dict(config="ResidentDriver::Step, %d groups x %d peers, %d groups touched by the host per step" % (G, n, dirty),
               first_step_ms=out[0], us_per_step=out[1], groups_uploaded_per_step=out[2],
               groups_applied_per_step=out[3], rebuilds=out[4])
    if with_batched:
        row["batched_driver_step_ms"] = out[5]
    print(json.dumps(row), flush=True)
    rows.append(row)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "resident.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(rows, open(path, "w"), indent=1)
