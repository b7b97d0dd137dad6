#!/bin/bash
# GPU box: bench.py the way the driver runs it, then with long runs; output under gpurun_out/.
# usage: bash tools/gpu/bench.sh <tag> [bench args...]
TAG=$1; shift
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python bench.py --warmup 5 --steps 20 "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?
cat gpurun_out/bench_$TAG.json
[ $rc -ne 0 ] && tail -20 gpurun_out/bench_$TAG.err
exit $rc
