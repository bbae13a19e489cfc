#!/bin/bash
# Issue / LDS counters of the single-pass kernels against the two-pass kernels they replace, on the bench workload.
#   bash profiles/pmc_p1.sh   (through gpurun, from the repo root).  One rocprofv3 --pmc pass per counter set.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/pmc_p1
mkdir -p $O
for sp in 1 0; do
  export BRAMBLE_AMD_SINGLE_PASS=$sp
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/s${sp}p$i -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/s${sp}p$i.log 2>&1 || { echo "set $i failed"; tail -3 $O/s${sp}p$i.log; }
    f=$(ls $O/s${sp}p$i/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 profiles/pmc_kernel.py $f "k_pro" > $O/s${sp}p$i.txt && python3 profiles/pmc_kernel.py $f "k_emit" >> $O/s${sp}p$i.txt
    rm -rf $O/s${sp}p$i
  done
done
cat $O/s*p*.txt
