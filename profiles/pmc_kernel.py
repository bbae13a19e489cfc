#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters: profiles/pmc_kernel.py counter_collection.csv [substring]"""
import csv
import sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[k][row["Counter_Name"]] += 1
for k in tot:
    print(k[:70])
    for c in sorted(tot[k]):
        print("    %-28s %16.1f per launch (%d launches)" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
