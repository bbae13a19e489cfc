#!/bin/bash
# kernel stats + issue counters of the -S DP kernels on a 200k-read C3 batch: profiles/ksw_prof.sh <tag>
set -o pipefail
T=${1:-ksw}
O=gpurun_out/${ROUND:-r04}/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python3 bench_extra.py c3 --reads ${READS:-200000} --steps 2 --warmup 1 > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $O/pmc -o run -- python3 bench_extra.py c3 --reads ${READS:-200000} --steps 1 --warmup 1 > $O/pmc.log 2>&1 || exit 1
python3 profiles/kstats.py $O/stats/run_kernel_stats.csv | head -12
python3 - $O <<'PY'
import csv,sys,json
from collections import defaultdict
O=sys.argv[1]
d=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for r in csv.DictReader(open(O+'/pmc/run_counter_collection.csv')):
    k=r['Kernel_Name'].split('(')[0].replace('void br::','').replace('br::','')
    if 'ksw' not in k: continue
    d[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
rows=None
for l in open(O+'/pmc.log'):
    if l.startswith('{'): rows=json.loads(l)['ksw_routing']
print(rows)
for k,v in d.items():
    print(k, n[k], {a.replace('SQ_',''):round(b/n[k]/1e6,2) for a,b in v.items()})
PY
