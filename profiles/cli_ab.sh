#!/bin/bash
# A/B of command-line variants on one generated input: CLI_LEVELS entries alternate, e.g.
#   profiles/cli_ab.sh "device@BRAMBLE_AMD_HUGE_PAGES=0,device@BRAMBLE_AMD_HUGE_PAGES=1" 3 out.log
set -e
pair="$1"; rounds="${2:-3}"; out="${3:-gpurun_out/cli_ab.log}"
levels="$pair"; for ((i = 1; i < rounds; i++)); do levels="$levels,$pair"; done
CLI_LEVELS="$levels" python3 bench_extra.py cli --reads "${READS:-10000000}" 2> "$out.err" | tail -1 > "$out.json"
python3 - "$out.json" <<'PY' | tee "$out"
import json, sys
d = json.loads(open(sys.argv[1]).read())
for k, v in d["results"].items():
    rep = v["report"]
    inside = rep.split("summed), ")[1].split("s wall")[0]
    print("%-60s wall %.2f s  inside %s s  user %.2f  sys %.2f  out %s |%s" % (k, v["wall_s"], inside, v["user_s"], v["sys_s"], v.get("out_xxh64", "")[:8], rep.split("stage busy time:")[1][:140]))
PY
