#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh r01
# Writes under gpurun_out/$1/; copy the summaries into profiles/$1/ afterwards.
set -o pipefail
R=${1:-r01}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python3 bench.py > $O/bench_10m.json.log 2>$O/bench_10m.err || exit 1
echo "bench done"; tail -1 $O/bench_10m.json.log | cut -c1-300
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_stats.log 2>&1 || exit 1
echo "stats done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || exit 1
echo "fetch done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1 || exit 1
echo "write done"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_bam -o run -- python3 bench_extra.py bam --reads 10000000 > $O/bam_stats.log 2>&1 || exit 1
echo "bam done"
