#!/bin/bash
# GPU box: the gpu-marked tests in one process, log under gpurun_out/.
# usage: bash tools/gpu/tests.sh <tag> [pytest args...]
TAG=$1; shift
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > gpurun_out/tests_$TAG.log 2>&1
rc=$?
tail -8 gpurun_out/tests_$TAG.log
exit $rc
