#!/bin/bash
# GPU box: mixed workload, library of the side tree ab/old against the current one, alternating.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for i in 1 2; do
  for tree in ab/old .; do
    (cd $tree && timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 --workload mixed --sets 6 "$@" 2>> $GRAFT_REPO_ROOT/gpurun_out/ab.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tree', round(r['roofline']['kernel_ms']*1e3,2), 'us', round(r['roofline']['frac'],4))") || exit 1
  done
done
