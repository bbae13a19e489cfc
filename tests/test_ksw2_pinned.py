"""Pins the ksw2 piece of the oracle (rows a9 / a10 of SURVEY.md section 8) independently of the hand that wrote it:
oracle/oracle_ksw2.hpp (anti-diagonal u/v/x/y difference recurrence, int8, direction bytes -- the shape of
ksw_extz2_sse) against tests/ksw2_gotoh.c (full-matrix int32 Gotoh, traceback by comparing matrix values) on 10^5
random rescue-shaped pairs, and both against the committed vectors tests/golden/ksw2_cases.json."""
import json
import os

import numpy as np

from oracle import oracle_binding as ob
from tests import ksw2_check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(t, q):
    cig, score, mx = ob.ksw_align(t, q)
    return int(score), int(mx), np.asarray(cig, dtype=np.uint32)


def _same(t, q, g, o):
    score, mx, cig = o
    assert mx == g["max"], ("max", t, q, mx, g["max"])
    assert score == g["score"], ("score", t, q, score, g["score"])
    assert ksw2_check.cigar_text(cig) == ksw2_check.cigar_text(g["cigar"]), (t, q, ksw2_check.cigar_text(cig), ksw2_check.cigar_text(g["cigar"]))
    # the CIGAR ends in the maximum cell and its own score is the maximum
    tl = sum(int(w) >> 4 for w in cig if (int(w) & 15) in (0, 2))
    ql = sum(int(w) >> 4 for w in cig if (int(w) & 15) in (0, 1))
    assert (tl, ql) == (g["max_t"] + 1, g["max_q"] + 1)


def _path_score(t, q, cig):
    i = j = 0
    s = 0
    for w in cig:
        op, ln = int(w) & 15, int(w) >> 4
        if op == 0:
            for k in range(ln):
                a, b = t[i + k], q[j + k]
                s += -1 if "N" in (a, b) else (1 if a == b else -4)
            i += ln
            j += ln
        else:
            s -= 4 + ln
            if op == 2:
                i += ln
            else:
                j += ln
    return s


def test_golden_vectors_hold_for_checker_and_oracle():
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "ksw2_cases.json")))
    assert len(doc["cases"]) >= 300
    n_drop = n_ok = 0
    for c in doc["cases"]:
        g = ksw2_check.gotoh(c["target"], c["query"])
        assert (g["score"], g["max"], g["max_t"], g["max_q"], g["zdropped"], ksw2_check.cigar_text(g["cigar"])) == \
            (c["score"], c["max"], c["max_t"], c["max_q"], c["zdropped"], c["cigar"])
        _same(c["target"], c["query"], g, _oracle(c["target"], c["query"]))
        if c["max"] > 0:
            assert _path_score(c["target"], c["query"], g["cigar"]) == c["max"]   # size-independent property: the path is worth the maximum
        n_drop += c["zdropped"]
        n_ok += c["max"] >= 10 and not c["zdropped"]
    assert n_drop >= 30 and n_ok >= 100   # both outcomes of the acceptance rule (src/evaluate.cpp:451-498) are represented


def test_oracle_ksw2_equals_full_matrix_gotoh_on_1e5_random_pairs():
    rng = np.random.RandomState(7)
    n_drop = n_n = 0
    for k in range(100000):
        t, q = ksw2_check.random_pair(rng)
        g = ksw2_check.gotoh(t, q)
        _same(t, q, g, _oracle(t, q))
        n_drop += g["zdropped"]
        n_n += "N" in t or "N" in q
    assert n_drop > 5000 and n_n > 5000


import pytest  # noqa: E402


@pytest.mark.gpu
def test_k_ksw_reproduces_the_golden_vectors_and_the_checker_on_random_pairs():
    """The HIP kernel itself (br_ctx_ksw_pairs: k_ksw alone) against the committed vectors and against the independent
    full-matrix checker on 20 000 random pairs: acceptance (max >= 10 and no z-drop before the last cell), maximum,
    and the traceback CIGAR of every accepted rescue."""
    from bramble_amd import lib, synth
    ann = synth.Annotation("S")
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "ksw2_cases.json")))
    pairs = [(c["target"], c["query"]) for c in doc["cases"]]
    want = [{"max": c["max"], "score": c["score"], "cigar": c["cigar"]} for c in doc["cases"]]
    rng = np.random.RandomState(11)
    for _ in range(20000):
        t, q = ksw2_check.random_pair(rng)
        g = ksw2_check.gotoh(t, q)
        pairs.append((t, q))
        want.append({"max": g["max"], "score": g["score"], "cigar": ksw2_check.cigar_text(g["cigar"])})
    ok, mx, cigs = ctx.ksw_pairs(pairs)
    n_ok = 0
    for p, w in enumerate(want):
        accept = w["max"] >= 10 and w["score"] != ksw2_check.NEG_INF
        assert bool(ok[p]) == accept, (pairs[p], int(ok[p]), w)
        assert int(mx[p]) == w["max"], (pairs[p], int(mx[p]), w)
        if accept:
            assert ksw2_check.cigar_text(cigs[p]) == w["cigar"], (pairs[p], ksw2_check.cigar_text(cigs[p]), w)
            n_ok += 1
    assert n_ok > 5000
    ctx.close()
    idx.close()


def _long_pair(rng, ql):
    """A rescue-shaped pair with a query of ql bases (beyond random_pair's 160): a copy with a few edits, or unrelated."""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    q = acgt[rng.randint(0, 4, ql)].copy()
    if rng.randint(0, 5) == 0:
        t = acgt[rng.randint(0, 4, int(rng.randint(1, ql + 41)))].copy()
    else:
        t = list(q)
        for _ in range(int(rng.randint(0, 1 + ql // 12))):
            k = int(rng.randint(0, len(t)))
            u = rng.randint(0, 3)
            if u == 0:
                t[k] = acgt[rng.randint(0, 4)]
            elif u == 1 and len(t) > 1:
                del t[k]
            else:
                t.insert(k, acgt[rng.randint(0, 4)])
        t = np.array(t + list(acgt[rng.randint(0, 4, int(rng.randint(0, 41)))]), dtype=np.uint8)[:ql + 40]
        if rng.randint(0, 4) == 0 and len(t) > 40:
            cut = int(rng.randint(20, len(t)))
            t[cut:] = acgt[rng.randint(0, 4, len(t) - cut)]
    if rng.randint(0, 6) == 0:
        t[int(rng.randint(0, len(t)))] = ord("N")
        q[int(rng.randint(0, len(q)))] = ord("N")
    return bytes(t).decode(), bytes(q).decode()


@pytest.mark.gpu
def test_streamed_dp_every_array_shape_leftovers_and_pieces():
    """k_ksw_dp (the register arrays of 64 / 128 / 256 / 384 target columns), the problems it leaves to k_ksw (targets
    beyond 384 columns) and a tape budget small enough to force several pieces: every problem against the full-matrix
    checker, and the general kernel alone (ksw_fast = 0) gives the same answers."""
    from bramble_amd import lib, synth
    ann = synth.Annotation("S")
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    rng = np.random.RandomState(23)
    pairs = [ksw2_check.random_pair(rng) for _ in range(3000)]
    pairs += [_long_pair(rng, int(rng.randint(150, 520))) for _ in range(1500)]
    # very short problems in a row (first bases closer than a lane's column count), targets shorter than the query
    pairs += [("ACGTACGTAC"[:int(rng.randint(1, 11))], "ACGTTGCA"[:int(rng.randint(1, 9))]) for _ in range(300)]
    pairs += [(p[0][:int(rng.randint(1, 30))], p[1]) for p in (_long_pair(rng, int(rng.randint(60, 300))) for _ in range(300))]
    pairs += [_long_pair(rng, 5000), _long_pair(rng, 2100)]   # targets beyond 2048: the general kernel keeps u / v / x / y in HBM
    order = rng.permutation(len(pairs))
    pairs = [pairs[k] for k in order]
    want = [ksw2_check.gotoh(t, q) for t, q in pairs]

    def check(ok, mx, cigs):
        n_ok = 0
        for p, g in enumerate(want):
            accept = g["max"] >= 10 and g["score"] != ksw2_check.NEG_INF
            assert bool(ok[p]) == accept, (pairs[p], int(ok[p]), g)
            assert int(mx[p]) == g["max"], (pairs[p], int(mx[p]), g)
            if accept:
                assert ksw2_check.cigar_text(cigs[p]) == ksw2_check.cigar_text(g["cigar"]), (pairs[p],)
                n_ok += 1
        return n_ok

    assert check(*ctx.ksw_pairs(pairs)) > 1500
    d = ctx.ksw_diag()
    assert d["pieces"] == 1 and all(n > 100 for n in d["per_shape"]) and d["leftover_before"] > 50, d
    assert d["leftover_after"] == d["leftover_before"], d       # no group ran out of tape
    ctx.set_param("ksw_tape_mb", 1)
    check(*ctx.ksw_pairs(pairs))
    assert ctx.ksw_diag()["pieces"] > 1
    # a tape too small for the batch: waves that find no chunk hand their problems to the general kernel
    ctx.set_param("ksw_tape_mb", 49152)
    ctx.set_param("ksw_tape_pct", 20)
    check(*ctx.ksw_pairs(pairs))
    d = ctx.ksw_diag()
    assert d["leftover_after"] > d["leftover_before"], d
    ctx.set_param("ksw_tape_pct", 100)
    ctx.set_param("ksw_fast", 0)
    check(*ctx.ksw_pairs(pairs))
    assert ctx.ksw_diag()["pieces"] == 0
    ctx.close()
    idx.close()
