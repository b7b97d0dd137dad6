#!/bin/bash
# GPU box: the AppendEntries batch with and without the follower commit rule, both layouts.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for lay in planes tiles; do for c in "" clamp; do
  timeout -k 10 200 python tools/append_probe.py 240 0 $lay $c >> gpurun_out/append_probe.txt 2>> gpurun_out/append_probe.err || exit 1
done; done
cat gpurun_out/append_probe.txt
