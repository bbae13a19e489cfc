cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/ks && mkdir -p gpurun_out/ks
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/ks -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ks.log 2>&1
f=$(ls gpurun_out/ks/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %5s avg_us %10.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
