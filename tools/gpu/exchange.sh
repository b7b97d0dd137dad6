#!/bin/bash
# GPU box: what the summary exchange and its device-side consumer cost the compute stream (one rank).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -f gpurun_out/r02_exchange.jsonl
for args in "" "--force-exchange --fence event" "--force-exchange --fence signal" "--force-exchange --consume --fence event" "--force-exchange --consume --fence signal" "--force-exchange --consume --exchange-impl native" "--force-exchange --exchange-impl native"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 $args >> gpurun_out/r02_exchange.jsonl 2>> gpurun_out/r02_exchange.err || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_exchange.jsonl"):
    r = json.loads(l); c = r["config"]
    print(c["summary_allgather"], c["summary_consumed_on_device"], c["summary_impl"], c.get("exchange_fence"), "us/step %.2f" % (r["ms_per_step"] * 1e3))
PY
