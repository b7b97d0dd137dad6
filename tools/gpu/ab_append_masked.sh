#!/bin/bash
# GPU box: the masked AppendEntries batch (5 slot planes, leader's slot masked out), another build of librq.so against the in-tree one.
# usage: bash tools/gpu/ab_append_masked.sh <other library, relative to the repository>
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
OTHER=${1:-ab/librq_old.so}
for i in 1 2; do
  for lib in "$OTHER" ""; do
    echo -n "${lib:-in-tree}: "
    RQ_LIBRARY=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python tools/append_probe.py 240 0 planes - masked 2>> gpurun_out/abm.err || exit 1
  done
done
