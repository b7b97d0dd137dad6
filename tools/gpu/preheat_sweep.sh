set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for rot in 1 1 4 16 64; do
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --preheat-rotations $rot | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); print('rot', $rot, 'kernel_us %.2f step_us %.2f frac %.4f'%(j['roofline']['kernel_ms']*1e3, j['ms_per_step']*1e3, j['roofline']['frac']))"
done
timeout -k 10 120 python3 bench.py --steps 400 --warmup 40 --no-extras --no-cpu-baseline | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); print('long', 'kernel_us %.2f step_us %.2f frac %.4f'%(j['roofline']['kernel_ms']*1e3, j['ms_per_step']*1e3, j['roofline']['frac']))"
