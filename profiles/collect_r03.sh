#!/bin/bash
# Round-3 evidence (run through gpurun from the repo root): bash profiles/collect_r03.sh [tag]
# The bench line, rocprofv3 kernel stats and the two PMC passes of the same command, the single-pass A/B with its issue
# counters, small calls (bench_extra small), configs[2] / configs[4] shapes, the records path, the command line, and
# `bench.py --gpus 2` starting its own ranks.  Outputs under gpurun_out/$1/.
set -o pipefail
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python3 bench.py > $O/bench_10m.json.log 2> $O/bench_10m.err || exit 1
echo "bench done"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie > $O/rocprof_stats.log 2>&1 || exit 1
echo "stats done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_write.log 2>&1 || exit 1
echo "traffic passes done"
bash profiles/ab_env.sh BRAMBLE_AMD_SINGLE_PASS 0 1 > $O/ab_single_pass.log 2>&1 || exit 1
bash profiles/pmc_p1.sh > $O/pmc_single_pass.txt 2>&1 || exit 1
echo "single-pass A/B done"
python3 bench_extra.py small > $O/bench_extra_small.json.log 2> $O/small.err || exit 1
for p in 5000 52000; do python3 bench.py --pairs $p --steps 50 --warmup 5 --no-cpu-baseline --no-pcie > $O/bench_pairs$p.json.log 2>/dev/null || exit 1; done
echo "small calls done"
python3 bench_extra.py c3 --reads 1000000 > $O/bench_extra_c3_1m.json.log 2> $O/c3.err || exit 1
python3 bench_extra.py c5 > $O/bench_extra_c5.json.log 2> $O/c5.err || exit 1
python3 bench_extra.py bundle > $O/bench_extra_bundle.json.log 2> $O/bundle.err || exit 1
echo "c3 / c5 / bundle done"
CLI_LEVELS=device,device python3 bench_extra.py cli --reads 10000000 > $O/bench_extra_cli10m.json.log 2> $O/cli.err || exit 1
echo "cli done"
BENCH_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --pairs 2000000 > $O/bench_2rank_self_launched.json.log 2> $O/bench_2rank.err || exit 1
echo "2-rank self-launch done"
