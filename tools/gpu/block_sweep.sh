#!/bin/bash
# GPU box: block-size sweep (variant bits 8-11 = threads / 64) of the fixed and the member-mask round on the tiled layout.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for wl in fused mixed; do
  for v in 0 0x100 0x200 0x300 0x400; do
    extra=""; [ $wl = mixed ] && extra="--workload mixed --sets 8"
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 --variant $v $extra 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$wl variant $v kernel_us %.2f frac %.4f'%(j['roofline']['kernel_ms']*1e3, j['roofline']['frac']))" || exit 1
  done
done
