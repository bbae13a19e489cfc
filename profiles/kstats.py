#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: profiles/kstats.py FILE"""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-52s calls %4s avg %10.1f us %5.1f%%" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
