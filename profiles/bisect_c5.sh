#!/bin/bash
# bench_extra.py <config> with each library build under .ab/bis/ (one box): bash profiles/bisect_c5.sh [c5|c3]
cfg=${1:-c5}
mkdir -p gpurun_out
cp bramble_amd/libbramble_amd.so /tmp/keep.so
for f in .ab/bis/*.so; do
  cp $f bramble_amd/libbramble_amd.so
  timeout -k 10 200 python3 bench_extra.py $cfg > gpurun_out/bis.log 2>&1 || { echo "$f failed"; tail -3 gpurun_out/bis.log; continue; }
  tail -1 gpurun_out/bis.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f', round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items() if b>0.1})"
done
cp /tmp/keep.so bramble_amd/libbramble_amd.so
