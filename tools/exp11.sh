mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "joint or mixed or variants or cluster_size" 2>&1 | tail -4 || exit 1
run() { timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 120 --workload mixed --sets 6 "$@" > gpurun_out/tmp.json 2>> gpurun_out/exp.err || exit 1; python -c "import json,sys;d=json.load(open('gpurun_out/tmp.json'));print(' '.join(sys.argv[1:]), '| kernel_us %.2f pct %.1f' % (d['roofline']['kernel_ms']*1e3, d['pct_hbm_peak']))" "$@"; }
for v in 0x432 0x472 0x232 0x272 0x231 0x271 0x431 0x131 0x132; do run --variant $v; done
