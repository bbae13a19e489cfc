#!/bin/bash
# Issue / cache counters of the count pass (k_project<8,false,false>) on the bench workload; run through gpurun from the
# repo root: bash profiles/pmc_count_pass.sh [substring of the kernel name].  One rocprofv3 --pmc pass per counter set.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
K=${1:-k_project<8, false}
O=gpurun_out/pmc_count
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/p$i -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/p$i.log 2>&1 || { echo "set $i failed"; tail -3 $O/p$i.log; }
  f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 profiles/pmc_kernel.py $f "$K" > $O/p$i.txt
  rm -rf $O/p$i
done
cat $O/p*.txt
