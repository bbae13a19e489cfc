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
