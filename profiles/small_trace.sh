#!/bin/bash
# Kernel timeline of a small device-resident step: bash profiles/small_trace.sh <pairs> <out-prefix>
# rocprofv3 --kernel-trace of bench.py --pairs N; per kernel the mean duration, and the step's kernel time against its wall.
set -o pipefail
N="${1:-5000}"; out="${2:-gpurun_out/small_trace}"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d "$out.d" -o run -- python3 bench.py --pairs "$N" --steps 200 --warmup 20 --no-cpu-baseline --no-pcie > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
f=$(find "$out.d" -name "*kernel_stats.csv" | head -1); t=$(find "$out.d" -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    calls = int(r["Calls"])
    if calls < 200:
        continue
    per_step = float(r["TotalDurationNs"]) / 220.0 / 1e3
    tot += per_step
    print("%-86s %5.1f calls/step  %7.2f us/step" % (r["Name"][:86], calls / 220.0, per_step))
print("kernel time per step: %.1f us" % tot)
# gaps: the last 50 steps' span from the trace
tr = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[2]))), key=lambda x: x[0])
seg = [i for i, x in enumerate(tr) if x[2].startswith("br::k_segment")]
if len(seg) > 60:
    a, b = seg[-51], seg[-1]
    span = (tr[b][0] - tr[a][0]) / 50.0 / 1e3
    busy = sum(e - s for s, e, _ in tr[a:b]) / 50.0 / 1e3
    print("last 50 steps: %.1f us per step from k_segment to k_segment, %.1f us of it inside kernels, %d launches per step" % (span, busy, (b - a) // 50))
PY
grep '"metric"' "$out.log" | cut -c1-200
rm -rf "$out.d"
