#!/bin/bash
# GPU box: the measurements behind profiles/r02_*.json (no profiler attached).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
set -x
timeout -k 10 700 python tools/config_report.py gpurun_out/r02_configs.json > gpurun_out/r02_configs.log 2>&1 || exit 1
for args in "--steps 20 --warmup 5" "--steps 400 --warmup 40" "--steps 400 --warmup 40 --force-exchange" \
            "--steps 400 --warmup 40 --force-exchange --consume" "--steps 400 --warmup 40 --force-exchange --consume --exchange-impl native" \
            "--steps 400 --warmup 40 --log-term-bits 16" "--steps 400 --warmup 40 --log-window 0"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args >> gpurun_out/r02_bench_sweep.jsonl 2>> gpurun_out/r02_bench_sweep.err || exit 1
done
for tb in 8 32; do timeout -k 10 300 ./examples/rq_bench 1048576 5 12 400 40 $tb >> gpurun_out/r02_rq_bench_c.txt || exit 1; done
cat gpurun_out/r02_rq_bench_c.txt
