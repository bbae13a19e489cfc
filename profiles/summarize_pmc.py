#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into per-launch HBM traffic per kernel.

usage: summarize_pmc.py FETCH_counter_collection.csv WRITE_counter_collection.csv PAIRS OUT_DIR

FETCH_SIZE / WRITE_SIZE are collected in separate `rocprofv3 --pmc X --kernel-trace` passes of
`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline` (MI355X_MICROARCH.md, HBM section).
Per dispatch the counter is in KiB.  FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but on gfx950 every L2->fabric read request is
128 bytes wide -- for coalesced streams and for random 4/16/32/64-byte gathers alike (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ,
the 32B / 64B request counters stay at zero: profiles/r02/pmc_calibration.json, profiles/calib_fetch.hip) -- so
bytes = 2 * FETCH_SIZE * 1024 for reads of every kernel here, and WRITE_SIZE * 1024 for writes.  Everything here is averaged PER LAUNCH (sum over the
dispatches of a kernel / number of dispatches), the same normalisation bench.py uses for
`roofline.achieved`.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("br::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name


def main():
    fetch_csv, write_csv, pairs, out_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    full = {}
    for k in sorted(set(ft) | set(wt)):
        if not k.startswith("void br::") and not k.startswith("br::"):
            continue
        f = ft.get(k, 0.0) / max(fc.get(k, 0), 1)
        w = wt.get(k, 0.0) / max(wc.get(k, 0), 1)
        full[short(k)] = {
            "launches_in_pass": fc.get(k, 0),
            "FETCH_SIZE_KB_per_launch": f,
            "WRITE_SIZE_KB_per_launch": w,
            "hbm_bytes_corrected": int((2.0 * f + w) * 1024.0),
            "correction": "reads 2 x FETCH_SIZE (all read requests are 128 B: r02/pmc_calibration.json), writes WRITE_SIZE as is",
        }
    json.dump(full, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)
    # the table bench.py reads: keyed like br_ctx kernel timers (bench.py's kernel_ms_per_step)
    def bench_key(k):
        k = k.replace(", ", ",")
        m = re.match(r"k_project<(\d+),(true|false),(true|false),(\d)>", k)
        if m:
            return "k_project<64,true>" if m.group(2) == "true" else "k_project<G,%s,%s,%s>" % (m.group(2), m.group(3), m.group(4))
        if k.startswith("k_scan"):
            return "k_scan_*"
        if re.match(r"k_(rows|primary)<", k):
            return re.sub(r"<.*>$", "", k)
        m = re.match(r"k_emit_dense<(true|false),(\d),(true|false)>", k)
        if m:
            return "k_emit_dense<%s,%s>" % (m.group(1), m.group(2))   # the timer's name (the third parameter is the -S variant)
        if k == "k_pair_emit":
            return "k_pair<true>"                                      # the emit pass keeps its timer's name
        if k in ("k_big<0>", "k_pair_big"):
            return "k_big<0>+k_pair_big"
        if k == "k_pair_mask_wide":
            return "k_pair_mask"
        return k
    # direct rows (the default path): bench.py's counter diagnostic at the end of the run projects the batch once more through
    # the match-table path (br_ctx_collect_counters); those kernels are not part of the timed step
    direct = any(k.startswith("k_emit_rows") for k in full)
    direct_step = re.compile(r"^(k_segment|k_group_ids|k_project<\d+, ?false, ?false, ?[12]>|k_name_seed|k_pair_mask(_wide)?|k_big<[01]>|k_pair_big|"
                             r"k_scan5_\w+|k_group_desc|k_expand_rows|k_emit_rows<[012]>)$")
    kern = {}
    for k, v in full.items():
        if k in ("k_stats", "k_sum_ncig"):
            continue   # diagnostics outside the timed step
        if direct and not direct_step.match(k):
            continue
        key = bench_key(k)
        if key in kern:
            t = dict(kern[key])
            for f in ("FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "hbm_bytes_corrected"):
                t[f] = t[f] + v[f]
            t["kernels_summed"] = t.get("kernels_summed", 1) + 1
            kern[key] = t
        else:
            kern[key] = dict(v)
    traffic = {
        "pairs": pairs,
        "note": "per launch; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
                "`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                "(gfx950 correction, MI355X_MICROARCH.md HBM section). Made by profiles/summarize_pmc.py from the "
                "CSVs beside pmc_summary.json.",
        "kernels": kern,
    }
    json.dump(traffic, open(os.path.join(os.path.dirname(os.path.abspath(out_dir.rstrip('/'))), "pmc_traffic.json"), "w"), indent=1)
    for k, v in sorted(full.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"]):
        print("%-40s %3d launches  %8.3f GB/launch" % (k, v["launches_in_pass"], v["hbm_bytes_corrected"] / 1e9))


if __name__ == "__main__":
    main()
