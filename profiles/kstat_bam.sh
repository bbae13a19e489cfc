# rocprofv3 kernel averages of `bench_extra.py bam` (the BAM re-encode kernels): bash profiles/kstat_bam.sh [pairs]
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/ksb && mkdir -p gpurun_out/ksb
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/ksb -o run -- python3 bench_extra.py bam --reads ${1:-2000000} --steps 2 --warmup 0 > gpurun_out/ksb.log 2>&1
f=$(ls gpurun_out/ksb/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "bam" in r["Name"] or "scan" in r["Name"]:
        print("%-70s calls %5s avg_us %10.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
