#!/usr/bin/env python3
"""One-off soak of the -S rescue DP kernels against the full-matrix checker (tests/ksw2_gotoh.c): N random rescue-shaped
pairs per seed (short ones, long ones up to 900 bases, N bases, foreign tails, very short problems in a row).
  python3 profiles/ksw_soak.py [pairs per seed] [seed ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tests import ksw2_check  # noqa: E402
from tests.test_ksw2_pinned import _long_pair  # noqa: E402
from bramble_amd import lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seeds = [int(x) for x in sys.argv[2:]] or [101, 202, 303]
ann = synth.Annotation("S")
idx = lib.Index(ann.as_dict(), device=0)
ctx = lib.Context(idx)
for seed in seeds:
    rng = np.random.RandomState(seed)
    t0 = time.time()
    pairs = [ksw2_check.random_pair(rng) for _ in range(n)]
    pairs += [_long_pair(rng, int(rng.randint(100, 900))) for _ in range(n // 5)]
    pairs += [("ACGTACGTAC"[:int(rng.randint(1, 11))], "ACGTTGCA"[:int(rng.randint(1, 9))]) for _ in range(n // 20)]
    order = rng.permutation(len(pairs))
    pairs = [pairs[k] for k in order]
    want = [ksw2_check.gotoh(t, q) for t, q in pairs]
    ok, mx, cigs = ctx.ksw_pairs(pairs)
    bad = 0
    n_ok = 0
    for p, g in enumerate(want):
        accept = g["max"] >= 10 and g["score"] != ksw2_check.NEG_INF
        if bool(ok[p]) != accept or int(mx[p]) != g["max"] or (accept and ksw2_check.cigar_text(cigs[p]) != ksw2_check.cigar_text(g["cigar"])):
            bad += 1
        n_ok += accept
    print("seed %d: %d pairs, %d accepted, %d differ from the checker, routing %s, %.0f s" % (seed, len(pairs), n_ok, bad, ctx.ksw_diag(), time.time() - t0), flush=True)
    assert bad == 0
