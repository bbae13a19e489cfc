"""The per-stage `traffic` of the bench line comes from profiles/pmc_traffic.json, keyed by the context's kernel timers.
A kernel that is renamed (k_pair<true> -> k_pair_emit) or gains a template parameter (k_emit_dense) silently drops out of
that table unless profiles/summarize_pmc.py maps it back: every timer the bench line's stages sum over must be there."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_traffic_covers_every_stage_of_the_bench_line():
    traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the kernel lists of the per-kernel table: entry(..., [ "k_...", ... ])
    wanted = set()
    for lst in re.findall(r"\[((?:\s*\"k_[^\"]+\",?)+)\s*\]\)", src):
        wanted.update(re.findall(r"\"(k_[^\"]+)\"", lst))
    assert {"k_segment", "k_rows", "k_pair<true>", "k_emit_dense<false,1>"} <= wanted
    missing = sorted(k for k in wanted if k not in traffic)
    assert not missing, "profiles/pmc_traffic.json lacks %s (profiles/summarize_pmc.py bench_key)" % missing
    for k in wanted:
        assert traffic[k]["hbm_bytes_corrected"] > 0
