#!/bin/bash
# as ab.sh, for the records -> records path: bash profiles/ab_bundle.sh
mkdir -p gpurun_out
for r in 1 2; do
  for v in base new; do
    cp .ab/$v.so bramble_amd/libbramble_amd.so
    timeout -k 10 300 python3 bench_extra.py bam --reads 10000000 > gpurun_out/abb_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/abb_$v.log; exit 1; }
    tail -1 gpurun_out/abb_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k!='workload'})" | cut -c1-400
  done
done
