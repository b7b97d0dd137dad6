mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 240 "$@" > gpurun_out/tmp.json 2>> gpurun_out/exp.err || exit 1; python -c "import json,sys;d=json.load(open('gpurun_out/tmp.json'));print(' '.join(sys.argv[1:]), '| kernel_us %.2f pct %.1f' % (d['roofline']['kernel_ms']*1e3, d['pct_hbm_peak']))" "$@"; }
for i in 1 2 3; do for v in 0x231 0x271; do run --variant $v; done; done
run --variant 0x271 --sets 16
run --variant 0x271 --sets 4
