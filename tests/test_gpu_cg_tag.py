"""CIGARs of more than 65535 ops through the device path: a record whose CIGAR field holds the
placeholder <l_seq>S<ref_len>N and whose real ops sit in a CG:B,I tag (SAM spec 4.2.2) is projected from the tag's ops --
what htslib's bam_read1 gives the reference (include/bramble.h:29-85 over gclib/GSam.cpp:197-201) -- the tag leaves the
output records, and a rewritten CIGAR of more than 65535 ops is spilled the same way on the way out (bam_write1).
Compared byte for byte with the oracle's stream; the records come from tests/bamio.py (assembled from the specification)."""
import struct

import numpy as np
import pytest

from bramble_amd import lib
from oracle import oracle_binding as ob
from tests import bamio
from tests.test_cg_tag_cpu import ANN, long_cigar

pytestmark = pytest.mark.gpu


def both(records, flags, bam_lanes=0):
    stream = bamio.frame(records)
    off, ln, _, _ = lib.bam_split(stream)
    assert len(off) == len(records)
    orc, _, _, parsed = ob.run_bam(ob.OracleIndex(ANN), ob.make_flags(**flags), stream, off, ln, np.array([0], np.int32))
    idx = lib.Index(ANN, device=0)
    ctx = lib.Context(idx)
    ctx.set_param("bam_lanes", bam_lanes)
    got, cnt = ctx.project_bam_bundle(lib.make_config(**flags), stream, off, ln, np.array([0], np.int32))
    ctx.close()
    idx.close()
    return got, cnt, orc, parsed


def test_placeholder_and_tag_combinations():
    real = [(20 << 4) | 0, (10 << 4) | 2, (30 << 4) | 0]
    cg3 = b"CGBI" + struct.pack("<IIII", 3, *real)
    recs = [bamio.bam_record(b"a_long", 0, 1499, real, 50, aux=b"NMC\x03", spill=True),            # restored from the tag
            bamio.bam_record(b"b_long_i", 0, 1499, [(50 << 4) | 4, (60 << 4) | 3], 50, aux=b"CGBi" + cg3[4:] + b"XSA+"),   # B,i counts too
            bamio.bam_record(b"c_plain", 0, 1499, [(50 << 4) | 4, (60 << 4) | 3], 50),             # the placeholder alone: a soft-clipped read
            bamio.bam_record(b"d_tagged", 0, 1499, [50 << 4], 50, aux=cg3 + b"NHC\x01"),           # an ordinary CIGAR keeps its CG tag
            bamio.bam_record(b"e_short", 0, 1499, [(50 << 4) | 4, (60 << 4) | 3], 50, aux=b"CGBI" + struct.pack("<II", 1, 50 << 4)),  # fewer entries than n_cigar
            bamio.bam_record(b"f_z", 0, 1499, [(50 << 4) | 4, (60 << 4) | 3], 50, aux=b"CGZab\0")]
    for flags in ({}, {"lr": 1}):
        for lanes in (0, 8):
            got, cnt, orc, parsed = both(recs, flags, lanes)
            assert list(np.diff(parsed["cigar_off"].astype(np.int64))) == [3, 3, 2, 1, 2, 2]
            assert orc["n_rows"] >= 4
            assert np.array_equal(got, orc["bam_stream"]), (flags, lanes)
    outs = bamio.split_stream(got)
    assert sum(1 for o in outs if o[32:38] == b"a_long" and b"CG" in bamio.record_fields(o)["aux"]) == 0
    assert any(o[32:40] == b"d_tagged" and b"CGBI" in bamio.record_fields(o)["aux"] for o in outs)


@pytest.mark.parametrize("lanes", [0, 8])
def test_ultra_long_read_round_trips_byte_identically(lanes):
    cig, qlen, rlen = long_cigar(17501)                      # 70004 ops
    rng = np.random.RandomState(5)
    recs = []
    for i in range(70):                                      # ordinary reads around the long ones: slow and fast rows share waves
        recs.append(bamio.bam_record(b"r%03d" % i, 0, 1200 + int(rng.randint(0, 5000)), [(90 << 4) | 0], 90, aux=b"NMC\x01"))
        if i in (3, 40):
            recs.append(bamio.bam_record(b"ultra%d" % i, 0, 1999 + i, cig, qlen, aux=b"NMi" + struct.pack("<i", 5) + b"ASi" + struct.pack("<i", 777),
                                         flag=16 if i == 40 else 0))
    got, cnt, orc, parsed = both(recs, {"lr": 1}, lanes)
    assert int(np.diff(parsed["cigar_off"].astype(np.int64)).max()) == len(cig)
    assert orc["n_rows"] == 2 * len(recs) and cnt["n_rows"] == orc["n_rows"]
    assert np.array_equal(got, orc["bam_stream"])
    spilled = [o for o in bamio.split_stream(got) if o[32:37] == b"ultra"]
    assert len(spilled) == 4
    for o in spilled:
        f = bamio.record_fields(o)
        assert f["n_cigar_field"] == 2 and (f["cigar"][0] & 0xF) == 4 and (f["cigar"][0] >> 4) == qlen and f["aux"].count(b"CGBI") == 1
