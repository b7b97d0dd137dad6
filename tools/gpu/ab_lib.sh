#!/bin/bash
# GPU box: bench.py with another build of librq.so (RQ_LIBRARY) against the in-tree one, alternating.
# usage: bash tools/gpu/ab_lib.sh <other library, relative to the repository> [bench args...]
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
OTHER=$1; shift
for i in 1 2 3; do
  for lib in "$OTHER" ""; do
    RQ_LIBRARY=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 "$@" 2>> gpurun_out/ab.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('${lib:-in-tree}', round(r['roofline']['kernel_ms']*1e3,2), 'us', round(r['roofline']['frac'],4), r['verified']['ok'])" || exit 1
  done
done
