mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 240 "$@" > gpurun_out/tmp.json 2>> gpurun_out/exp.err || exit 1; python -c "import json,sys;d=json.load(open('gpurun_out/tmp.json'));print(' '.join(sys.argv[1:]), '| kernel_us %.2f pct %.1f' % (d['roofline']['kernel_ms']*1e3, d['pct_hbm_peak']))" "$@"; }
run
run
echo "--- gloo rehearsal"
RQ_BENCH_TRACE_AFTER=40 timeout -k 10 80 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --sets 2 --groups 65536 --backend gloo > gpurun_out/gloo2.log 2>&1; echo "rc $?"; grep -v "^  File \"/usr" gpurun_out/gloo2.log | tail -40
