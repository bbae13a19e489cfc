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
    # the kernel lists of the per-kernel table of the default (direct rows) path: entry(..., [ "k_...", ... ]) for segment and
    # count, pair_names / emit_names for the two stages behind them
    wanted = set()
    for lst in re.findall(r"(?:pair_names|emit_names) = \[((?:\s*\"k_[^\"]+\",?)+)\s*\]", src):
        wanted.update(re.findall(r"\"(k_[^\"]+)\"", lst))
    for lst in re.findall(r"\[((?:\s*\"k_[^\"]+\",?)+)\s*\]\)", src):
        names = re.findall(r"\"(k_[^\"]+)\"", lst)
        if any(n.startswith(("k_segment", "k_project<G")) for n in names):
            wanted.update(names)
    assert {"k_segment", "k_pair_mask", "k_group_desc", "k_emit_rows<1>", "k_emit_rows<2>", "k_expand_rows", "k_big<0>+k_pair_big"} <= wanted
    missing = sorted(k for k in wanted if k not in traffic)
    assert not missing, "profiles/pmc_traffic.json lacks %s (profiles/summarize_pmc.py bench_key)" % missing
    for k in wanted:
        assert traffic[k]["hbm_bytes_corrected"] > 0
