#!/bin/bash
# PMC pass over bench.py for the fused radial-MLP forward kernel.  -> gpurun_out/pmc_mlp/
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_mlp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || { echo "$name failed"; tail -3 $out/$name.log; exit 1; }
  echo "$name done"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
