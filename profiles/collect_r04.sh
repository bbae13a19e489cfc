#!/bin/bash
# Round-4 evidence (run through gpurun from the repo root): bash profiles/collect_r04.sh [tag]
# The bench line, rocprofv3 kernel stats and the two PMC passes of the same command (direct rows), and the A/B against the
# match-table path.  Outputs under gpurun_out/$1/.
set -o pipefail
R=${1:-r04}
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
bash profiles/ab_env.sh BRAMBLE_AMD_DIRECT_ROWS 0 1 > $O/ab_direct_rows.log 2>&1 || exit 1
echo "A/B done"
f=$(ls $O/pmc_fetch/*counter_collection.csv | head -1); w=$(ls $O/pmc_write/*counter_collection.csv | head -1)
mkdir -p $O/summary
python3 profiles/summarize_pmc.py $f $w 10000000 $O/summary/ > $O/pmc_summary.txt 2>&1 || exit 1
cp $f $O/pmc_fetch_counter_collection.csv; cp $w $O/pmc_write_counter_collection.csv
s=$(ls $O/stats/*kernel_stats.csv | head -1); cp $s $O/kernel_stats_bench10m.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
cat $O/pmc_summary.txt
