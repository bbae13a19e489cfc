#!/bin/bash
# Round-2 evidence (run through gpurun from the repo root): profiles/collect_r02.sh [tag]
# kernel stats + the two PMC passes of the bench command, kernel stats of the long-read configurations
# (k_ksw, k_project_fa, the SIMF kernels), the records-in/records-out path.  Outputs under gpurun_out/$1/.
set -o pipefail
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python3 bench.py > $O/bench_10m.json.log 2> $O/bench_10m.err || exit 1
echo "bench done"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie > $O/rocprof_stats.log 2>&1 || exit 1
echo "stats done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_fetch.log 2>&1 || exit 1
echo "fetch done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_write.log 2>&1 || exit 1
echo "write done"
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_rdsplit -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_rdsplit.log 2>&1 || exit 1
echo "read request sizes done"
python3 bench_extra.py c3 --reads 1000000 > $O/bench_extra_c3_1m.json.log 2> $O/c3.err || exit 1
echo "c3 done"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_c3 -o run -- python3 bench_extra.py c3 --reads 1000000 > $O/c3_stats.log 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $O/pmc_c3 -o run -- python3 bench_extra.py c3 --reads 1000000 --steps 1 --warmup 1 > $O/c3_pmc.log 2>&1 || exit 1
echo "c3 stats + issue counters done"
python3 bench_extra.py c5 > $O/bench_extra_c5.json.log 2> $O/c5.err || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_c5 -o run -- python3 bench_extra.py c5 > $O/c5_stats.log 2>&1 || exit 1
echo "c5 done"
python3 bench_extra.py bam > $O/bench_extra_bam.json.log 2> $O/bam.err || exit 1
BAM_LANES=8 python3 bench_extra.py bam > $O/bench_extra_bam_lanes8.json.log 2> $O/bam8.err || exit 1
bash profiles/pmc_bam.sh 4000000 k_bam_ 0 > $O/pmc_bam_tasks_4m.txt 2>&1 || exit 1
bash profiles/pmc_bam.sh 4000000 k_bam_encode 8 > $O/pmc_bam_encode8_4m.txt 2>&1 || exit 1
profiles/bin/calib_store > $O/calib_store.txt 2>&1 || exit 1
echo "bam stage done"
python3 bench_extra.py bundle > $O/bench_extra_bundle.json.log 2> $O/bundle.err || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_bundle -o run -- python3 bench_extra.py bundle > $O/bundle_stats.log 2>&1 || exit 1
echo "bundle done"
BENCH_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --pairs 2000000 --no-cpu-baseline > $O/bench_2rank_gloo_rehearsal.json.log 2> $O/bench_2rank.err || exit 1
echo "2-rank rehearsal done"
