mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 240 "$@" > gpurun_out/tmp.json 2>> gpurun_out/exp.err || exit 1; python -c "import json,sys;d=json.load(open('gpurun_out/tmp.json'));print(' '.join(sys.argv[1:]), '| kernel_us %.2f pct %.1f' % (d['roofline']['kernel_ms']*1e3, d['pct_hbm_peak']))" "$@"; }
run
run --no-bits
run --no-status
run --no-bits --no-status
run --what vote
run --what commit
run --workload mixed
run --workload mixed --variant 0x401
run --workload mixed --variant 0x201
run --workload mixed --variant 0x202
