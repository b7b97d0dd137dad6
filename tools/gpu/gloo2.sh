#!/bin/bash
# GPU box: two gloo ranks on the one GPU, repeated; stacks dumped if stuck.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
i=0
for args in "--fence event" "--fence signal" "--fence event" "--fence signal --consume" "--fence event --consume" "--fence signal" "--fence signal --sets 3 --rounds-per-exchange 1" "--fence event --sets 3 --rounds-per-exchange 1"; do
  i=$((i+1))
  RQ_BENCH_TRACE_AFTER=25 timeout -k 10 100 python3 bench.py --gpus 2 --backend gloo --groups 65536 --steps 16 --warmup 4 --no-cpu-baseline --no-extras $args > gpurun_out/g$i.json 2> gpurun_out/g$i.err
  echo "[$args] rc=$?"
done
