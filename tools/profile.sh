#!/bin/bash
# rocprofv3 passes for the default bench workload; one counter group per run.
# usage: bash tools/profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PROF_CMD overrides the profiled program (a python script path and its arguments, relative to the repository)
if [ -n "${PROF_CMD:-}" ]; then BENCH="python3 $GRAFT_REPO_ROOT/$PROF_CMD"; else
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 10 $*"; fi
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $OUT/kt.log; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_REQ_sum TCC_BUSY_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($grp) failed"; tail -3 $OUT/pmc$i.log; }
done
cd $GRAFT_REPO_ROOT && python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
