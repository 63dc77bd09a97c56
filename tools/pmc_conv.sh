#!/bin/bash
# PMC passes over the conv micro-benchmark (tools/bench_conv.py), one counter set per run.
# usage: tools/pmc_conv.sh   -> gpurun_out/pmc_conv/<set>/...counter_collection.csv
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_conv; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/tools/bench_conv.py c3 > $out/$name.log 2>&1 || { echo "$name failed"; tail -3 $out/$name.log; exit 1; }
  echo "$name done"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_ANY
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
