set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
N=${1:-2000000}
K=${2:-k_bam_rows}
O=gpurun_out/pmc_bam2
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/p$i -o run -- python3 bench_extra.py bam --reads $N --steps 1 --warmup 0 > $O/p$i.log 2>&1 || { echo "set $i failed: $set"; tail -2 $O/p$i.log; }
  f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 profiles/pmc_kernel.py $f "$K" > $O/p$i.txt
  rm -rf $O/p$i
done
cat $O/p*.txt
