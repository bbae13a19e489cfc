#!/bin/bash
# Issue / memory-pipeline counters of the BAM re-encode kernels (k_bam_tasks or k_bam_encode<G>, k_bam_scan, k_bam_size) on
# `bench_extra.py bam`; run through gpurun from the repo root: bash profiles/pmc_bam.sh [pairs] [kernel substring] [BAM_LANES].
# One rocprofv3 --pmc pass per counter set.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
N=${1:-4000000}
K=${2:-k_bam_}
export BAM_LANES=${3:-0}
O=gpurun_out/pmc_bam
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/p$i -o run -- python3 bench_extra.py bam --reads $N --steps 1 --warmup 0 > $O/p$i.log 2>&1 || { echo "set $i failed: $set"; tail -2 $O/p$i.log; }
  f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 profiles/pmc_kernel.py $f "$K" > $O/p$i.txt
  rm -rf $O/p$i
done
echo "# rocprofv3 --pmc, bench_extra.py bam --reads $N --steps 1 --warmup 0, BAM_LANES=$BAM_LANES, per launch"
cat $O/p*.txt
