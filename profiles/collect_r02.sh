#!/bin/bash
# Round-2 evidence (run through gpurun from the repo root): profiles/collect_r02.sh [tag]
# kernel stats + the two PMC passes of the bench command, kernel stats of the long-read configurations
# (k_ksw, k_project_fa, the SIMF kernels), the records-in/records-out path.  Outputs under gpurun_out/$1/.
set -o pipefail
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie > $O/rocprof_stats.log 2>&1 || exit 1
echo "stats done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_fetch.log 2>&1 || exit 1
echo "fetch done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_write.log 2>&1 || exit 1
echo "write done"
python3 bench_extra.py c3 --reads 1000000 > $O/bench_extra_c3_1m.json.log 2> $O/c3.err || exit 1
echo "c3 done"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_c3 -o run -- python3 bench_extra.py c3 --reads 200000 > $O/c3_stats.log 2>&1 || exit 1
echo "c3 stats done"
python3 bench_extra.py c5 > $O/bench_extra_c5.json.log 2> $O/c5.err || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_c5 -o run -- python3 bench_extra.py c5 > $O/c5_stats.log 2>&1 || exit 1
echo "c5 done"
python3 bench_extra.py bundle > $O/bench_extra_bundle.json.log 2> $O/bundle.err || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_bundle -o run -- python3 bench_extra.py bundle > $O/bundle_stats.log 2>&1 || exit 1
echo "bundle done"
