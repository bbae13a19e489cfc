#!/bin/bash
# kernel stats of the records path (bench_extra.py bundle, 20.9 M records): bash profiles/bundle_prof.sh out-prefix [ENV=VALUE]
set -o pipefail
out="${1:-gpurun_out/bundle_prof}"; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d "$out.d" -o run -- python3 bench_extra.py bundle --steps 3 --warmup 1 > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
f=$(find "$out.d" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out.kernel_stats.csv"; rm -rf "$out.d"
python3 - "$out.kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
