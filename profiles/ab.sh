#!/bin/bash
# A/B of two builds of libbramble_amd.so on one box (boxes differ by a few per cent): .ab/base.so against .ab/new.so,
# interleaved.  bash profiles/ab.sh [bench.py args]
set -o pipefail
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in base new; do
    cp .ab/$v.so bramble_amd/libbramble_amd.so
    timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/ab_$v.log; exit 1; }
    tail -1 gpurun_out/ab_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items() if b>0})"
  done
done
