#!/bin/bash
# GPU box: the bench line with every secondary figure, another build of librq.so against the in-tree one.
# usage: bash tools/gpu/ab_extras.sh <other library, relative to the repository>
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
OTHER=${1:-ab/librq_old.so}
for lib in "$OTHER" ""; do
  RQ_LIBRARY=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 400 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>> gpurun_out/abx.err > gpurun_out/abx_${lib:+old}.json || exit 1
done
python - <<'PY'
import json
o=json.load(open('gpurun_out/abx_old.json')); n=json.load(open('gpurun_out/abx_.json'))
print('headline', round(o['roofline']['kernel_ms']*1e3,2), round(n['roofline']['kernel_ms']*1e3,2))
for k in n['extra']:
    a=o['extra'].get(k,{}); b=n['extra'][k]
    f=lambda v: round(v.get('kernel_ms', v.get('ms_per_round',0))*1e3,2)
    print(k, f(a), f(b), b.get('verified',{}).get('ok'))
PY
