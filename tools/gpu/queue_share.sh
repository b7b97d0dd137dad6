#!/bin/bash
# GPU box: the start-signal fence with 0, 4, 16 and 64 further busy streams in the process,
# the wait enqueued after (launch-first) and before (wait-first) the launch that satisfies it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for order in launch-first wait-first; do
for k in 0 4 16 64; do
  timeout -k 10 90 python tools/queue_share_probe.py $k $order >> gpurun_out/queue_share.jsonl 2>> gpurun_out/queue_share.err || { echo "probe with $k extra streams ended abnormally"; cat gpurun_out/queue_share.jsonl; tail -5 gpurun_out/queue_share.err; exit 1; }
  tail -n 1 gpurun_out/queue_share.jsonl | grep -q '"completed": false' && break
done
done
cat gpurun_out/queue_share.jsonl
