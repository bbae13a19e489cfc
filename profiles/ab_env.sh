#!/bin/bash
# A/B of one build under two settings of an environment variable, interleaved on one box (boxes differ by a few per cent):
#   bash profiles/ab_env.sh BRAMBLE_AMD_SINGLE_PASS 0 1 [bench.py args]
set -o pipefail
var=$1; a=$2; b=$3; shift 3
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in $a $b; do
    export $var=$v
    timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie "$@" > gpurun_out/abenv_$v.log 2>&1 || { echo "$var=$v failed"; tail -5 gpurun_out/abenv_$v.log; exit 1; }
    tail -1 gpurun_out/abenv_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$var=$v', round(d['ms_per_step'],3), d['projected_records_per_step'], d['matches_per_step'], {a:round(b,3) for a,b in k.items() if b>0})"
  done
done
