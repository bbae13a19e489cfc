#!/bin/bash
# count-pass group width: bash profiles/sweep_group_lanes.sh (through gpurun)
mkdir -p gpurun_out
for g in 8 16 8 16; do
timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --group-lanes $g > gpurun_out/gl.log 2>&1
tail -1 gpurun_out/gl.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('G',$g, round(d['ms_per_step'],3), k['k_project<G,false>'])"
done
