#!/bin/bash
# GPU box: the three fuzz tests with other seeds and many more cases than the suite runs.
# usage: bash tools/gpu/soak.sh <iterations per test> <seed> [<seed> ...]
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
ITERS=$1; shift
for seed in "$@"; do
  RQ_FUZZ_SEED=$seed RQ_FUZZ_ITERS=$ITERS timeout -k 10 1000 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz" --timeout 900 --timeout-method thread > gpurun_out/soak_$seed.log 2>&1
  rc=$?
  echo "seed $seed: rc=$rc $(tail -n 1 gpurun_out/soak_$seed.log)"
  if [ $rc -ne 0 ]; then tail -n 30 gpurun_out/soak_$seed.log; exit $rc; fi
done
