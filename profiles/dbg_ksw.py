import numpy as np, sys
sys.path.insert(0,'/root/repo')
from tests import ksw2_check
from tests.test_ksw2_pinned import _long_pair
from bramble_amd import lib, synth
ann = synth.Annotation("S"); idx = lib.Index(ann.as_dict(), device=0); ctx = lib.Context(idx)
rng = np.random.RandomState(23)
pairs = [ksw2_check.random_pair(rng) for _ in range(3000)]
pairs += [_long_pair(rng, int(rng.randint(150, 520))) for _ in range(1500)]
pairs += [("ACGTACGTAC"[:int(rng.randint(1, 11))], "ACGTTGCA"[:int(rng.randint(1, 9))]) for _ in range(300)]
pairs += [(p[0][:int(rng.randint(1, 30))], p[1]) for p in (_long_pair(rng, int(rng.randint(60, 300))) for _ in range(300))]
order = rng.permutation(len(pairs)); pairs = [pairs[k] for k in order]
want = [ksw2_check.gotoh(t, q) for t, q in pairs]
for pct in (100, 20, 50, 5):
    ctx.set_param("ksw_tape_pct", pct)
    ok, mx, cigs = ctx.ksw_pairs(pairs)
    bad = [p for p,g in enumerate(want) if int(mx[p]) != g["max"]]
    print(pct, ctx.ksw_diag(), "bad", len(bad), [(p, len(pairs[p][1]), len(pairs[p][0]), want[p]["max"]) for p in bad[:12]])
