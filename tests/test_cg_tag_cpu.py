"""CIGARs of more than 65535 ops (CG:B,I, SAM spec 4.2.2) in the oracle: htslib restores such a CIGAR when it reads a record
and spills it again when it writes one (bam_read1 / bam_write1), so the reference projects ultra-long reads like any other
(include/bramble.h:29-85 over gclib/GSam.cpp:197-201).  The records here are assembled by tests/bamio.py from the
specification; htslib itself is not in the reference tree (parity unpinned for its primitives)."""
import struct

import numpy as np

from oracle import oracle_binding as ob
from tests import bamio

ANN = {"refnames": ["chr1"], "transcripts": [{"id": "fwd", "ref_id": 0, "strand": "+", "exons": [[1001, 301001]]},
                                             {"id": "rev", "ref_id": 0, "strand": "-", "exons": [[1001, 301001]]}]}


def long_cigar(n_units):
    """(3M 1I 2M 1D) x n_units: four ops, six query bases, six reference bases per unit"""
    unit = [(3 << 4) | 0, (1 << 4) | 1, (2 << 4) | 0, (1 << 4) | 2]
    return unit * n_units, 6 * n_units, 6 * n_units


def run_stream(records, flags):
    stream = bamio.frame(records)
    off, p = [], 0
    for r in records:
        off.append(p + 4)
        p += 4 + len(r)
    rows, _, _, parsed = ob.run_bam(ob.OracleIndex(ANN), ob.make_flags(**flags), stream, np.array(off, np.uint64),
                                    np.array([len(r) for r in records], np.uint32), np.array([0], np.int32))
    return rows, parsed


def test_placeholder_with_cg_tag_reads_like_the_cigar_in_place():
    real = [(20 << 4) | 0, (10 << 4) | 2, (30 << 4) | 0]
    aux = b"NMC\x03" + b"XSA+"
    spilled = bamio.bam_record(b"r", 0, 1499, real, 50, aux=aux, spill=True)
    plain = bamio.bam_record(b"r", 0, 1499, real, 50, aux=aux, spill=False)
    assert b"CGBI" in spilled and b"CGBI" not in plain and len(spilled) == len(plain) - 4 + 8 + 12
    a, pa = run_stream([spilled], {})
    b, pb = run_stream([plain], {})
    assert a["n_rows"] == b["n_rows"] == 1 and list(pa["cigar_off"]) == list(pb["cigar_off"]) == [0, 3]
    assert np.array_equal(a["cigar"], b["cigar"]) and np.array_equal(a["bam_stream"], b["bam_stream"])
    assert b"CG" not in bytes(a["bam_stream"])          # the tag left with the restore
    # not restored: an ordinary CIGAR next to a CG tag, the placeholder without the tag, a CG tag of another type
    for rec in (bamio.bam_record(b"r", 0, 1499, [50 << 4], 50, aux=b"CGBI" + struct.pack("<II", 1, 50 << 4)),
                bamio.bam_record(b"r", 0, 1499, [(50 << 4) | 4, (60 << 4) | 3], 50),
                bamio.bam_record(b"r", 0, 1499, [(50 << 4) | 4, (60 << 4) | 3], 50, aux=b"CGZ" + b"x\0")):
        rows, parsed = run_stream([rec], {})
        assert int(parsed["cigar_off"][1]) == rec[12] | rec[13] << 8


def test_long_cigar_is_spilled_again_on_the_way_out():
    cig, qlen, rlen = long_cigar(17501)                      # 70004 ops
    rec = bamio.bam_record(b"ultra", 0, 1999, cig, qlen, aux=b"NMi" + struct.pack("<i", 5))
    assert struct.unpack_from("<H", rec, 12)[0] == 2        # the placeholder went into the CIGAR field
    rows, parsed = run_stream([rec], {"lr": 1})
    assert int(parsed["cigar_off"][1]) == len(cig)
    assert rows["n_rows"] == 2 and sorted(chr(s) for s in rows["strand"]) == ["+", "-"]
    outs = bamio.split_stream(rows["bam_stream"])
    assert len(outs) == 2
    for k, out in enumerate(outs):
        f = bamio.record_fields(out)
        c0, c1 = int(rows["cigar_off"][k]), int(rows["cigar_off"][k + 1])
        want = [int(w) for w in rows["cigar"][c0:c1]]
        assert len(want) > 65535
        if chr(rows["strand"][k]) == "-":
            want = want[::-1]                                # reverse_complement_bam turns the op order around
        reflen = sum(w >> 4 for w in want if (w & 0xF) in (0, 2, 3, 7, 8))
        assert f["n_cigar_field"] == 2 and f["cigar"] == [(qlen << 4) | 4, (reflen << 4) | 3]
        tail = out[-(8 + 4 * len(want)):]
        assert tail[:4] == b"CGBI" and struct.unpack_from("<I", tail, 4)[0] == len(want)
        assert list(np.frombuffer(tail[8:], dtype="<u4")) == want
        assert f["aux"].count(b"CGBI") == 1 and f["aux"].endswith(tail)
        # and reading that record back gives the projected CIGAR again
        back, pb = run_stream([bamio.bam_record(b"x", 0, 10, want, qlen)], {"lr": 1})
        assert int(pb["cigar_off"][1]) == len(want)
