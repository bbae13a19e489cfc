#!/bin/bash
# Issue / LDS / wait counters of one kernel: bash profiles/pmc_any.sh <kernel-substring> <out-dir> -- python3 <script> args...
# (through gpurun, from the repo root; one rocprofv3 --pmc pass per counter set; the program itself follows `--`)
set -o pipefail
K="$1"; O="$2"; shift 3
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/p$i -o run -- "$@" > $O/p$i.log 2>&1 || { echo "set $i failed"; tail -3 $O/p$i.log; }
  f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 profiles/pmc_kernel.py $f "$K" > $O/p$i.txt
  rm -rf $O/p$i
done
cat $O/p*.txt
