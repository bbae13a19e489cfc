#!/bin/bash
# kernel trace of one command-line run (the binary itself goes after `--`: it never re-executes anything)
set -e
out="${1:-gpurun_out/cli_prof}"
export CLI_TMP=/tmp/brcli
CLI_LEVELS=device python3 bench_extra.py cli --reads "${READS:-10000000}" 2> "$out.bench.err" | tail -1 > "$out.bench.json"
cd /tmp && export TMPDIR=/tmp
BRAMBLE_AMD_CLI_CLEANUP=1 rocprofv3 --output-format csv --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out.d" -o cli -- "$GRAFT_REPO_ROOT/bramble_amd/bin/bramble" /tmp/brcli/in.bam -G /tmp/brcli/guides.gtf -o /tmp/brcli/prof.bam -p 16 --device-deflate > "$GRAFT_REPO_ROOT/$out.run.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "$out.d" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out.kernel_stats.csv"
head -30 "$out.kernel_stats.csv" | cut -c1-150
tail -4 "$out.run.log"
rm -rf "$out.d"
