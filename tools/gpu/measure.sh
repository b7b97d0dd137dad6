#!/bin/bash
# GPU box: the measurements behind profiles/r03_*.json (no profiler attached).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
set -x
timeout -k 10 800 python tools/config_report.py gpurun_out/r03_configs.json > gpurun_out/r03_configs.log 2>&1 || exit 1
for args in "--steps 20 --warmup 5" "--steps 400 --warmup 40" "--steps 400 --warmup 40 --force-exchange" \
            "--steps 400 --warmup 40 --force-exchange --consume" \
            "--steps 400 --warmup 40 --workload mixed --sets 8" "--steps 400 --warmup 40 --workload mixed --sets 8 --no-segments"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args >> gpurun_out/r03_bench_sweep.jsonl 2>> gpurun_out/r03_bench_sweep.err || exit 1
done
for tb in 8 32; do timeout -k 10 300 ./examples/rq_bench 1048576 5 12 400 40 $tb >> gpurun_out/r03_rq_bench_c.txt || exit 1; done
cat gpurun_out/r03_rq_bench_c.txt
