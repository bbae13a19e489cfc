"""Test infrastructure: ctypes binding of tests/ksw2_gotoh.c (the independent full-matrix Gotoh checker of the oracle's
ksw2 restatement) and the seeded generator of rescue-shaped sequence pairs."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NEG_INF = -0x40000000


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(tempfile.gettempdir(), "ksw2_gotoh_%d.so" % os.getuid())
        src = os.path.join(_HERE, "ksw2_gotoh.c")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-shared", "-fPIC", src, "-o", out])
        L = C.CDLL(out)
        L.gotoh_extz.restype = C.c_int
        L.gotoh_extz.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def gotoh(tseq, qseq):
    """-> dict(score, max, max_t, max_q, zdropped, cigar uint32[])"""
    out = (C.c_int32 * 5)()
    cap = len(tseq) + len(qseq) + 4
    cig = np.zeros(cap, dtype=np.uint32)
    n = lib().gotoh_extz(tseq.encode(), qseq.encode(), out, cig.ctypes.data, cap)
    assert 0 <= n <= cap
    return {"score": int(out[0]), "max": int(out[1]), "max_t": int(out[2]), "max_q": int(out[3]), "zdropped": int(out[4]),
            "cigar": cig[:n].copy()}


def cigar_text(words):
    return "".join("%d%s" % (int(w) >> 4, "MID"[int(w) & 15]) for w in words)


def random_pair(rng):
    """One (target, query) shaped like bramble's clip rescue (src/evaluate.cpp:320-365,500-546): the query is a soft clip
    plus overhang (5..~160 bases), the target the neighbouring exon sequence trimmed to qlen + 40.  Mix: a true copy with
    substitutions / indels / N bases, a copy whose tail is foreign (z-drop), an unrelated target, very short ones."""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    kind = rng.randint(0, 10)
    ql = int(rng.randint(1, 8)) if kind == 0 else int(rng.randint(5, 160))
    q = acgt[rng.randint(0, 4, ql)].copy()
    if kind in (1, 2):   # unrelated target
        tl = int(rng.randint(1, ql + 41))
        t = acgt[rng.randint(0, 4, tl)].copy()
    else:
        t = list(q)
        rate = [0.0, 0.02, 0.08, 0.2][rng.randint(0, 4)]
        k = 0
        while k < len(t):
            u = rng.rand()
            if u < rate * 0.5:
                t[k] = acgt[rng.randint(0, 4)]
            elif u < rate * 0.75:
                del t[k]
                continue
            elif u < rate:
                t.insert(k, acgt[rng.randint(0, 4)])
                k += 1
            k += 1
        t = np.array(t + list(acgt[rng.randint(0, 4, int(rng.randint(0, 41)))]), dtype=np.uint8)
        if kind in (3, 4) and len(t) > 12:   # foreign tail: the alignment should stop (z-drop) part of the way
            cut = int(rng.randint(6, len(t)))
            t[cut:] = acgt[rng.randint(0, 4, len(t) - cut)]
            if len(q) > cut:
                q[cut:] = acgt[rng.randint(0, 4, len(q) - cut)]
        t = t[:ql + 40]
        if len(t) == 0:
            t = acgt[rng.randint(0, 4, 1)].copy()
    if kind in (5, 6):   # N bases on either side (scored -e against anything)
        for arr in (t, q):
            for p in rng.randint(0, len(arr), int(rng.randint(0, 4))):
                arr[p] = ord("N")
    return bytes(t).decode(), bytes(q).decode()
