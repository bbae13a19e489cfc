#!/bin/bash
# as ab.sh, for the records -> records path: bash profiles/ab_bundle.sh [bam|bundle]
mkdir -p gpurun_out
W=${1:-bundle}
for r in 1 2 3; do
  for v in base new; do
    cp .ab/$v.so bramble_amd/libbramble_amd.so
    timeout -k 10 300 python3 bench_extra.py $W --reads 10000000 > gpurun_out/abb_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/abb_$v.log; exit 1; }
    tail -1 gpurun_out/abb_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), {k:v for k,v in d['kernel_ms_per_step'].items() if 'bam' in k or 'rec_' in k})" | cut -c1-300
  done
done
