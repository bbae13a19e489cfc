"""An independent, literal reading of write_to_bam's record rules (src/core.cpp:96-212; src/bam.cpp:474-702) on two tiny
records, assembled byte by byte here and compared with the oracle (CPU) and, in the GPU run, with the device encoders (k_bam_tasks).  This is
the cross-check for the part of the oracle that restates htslib primitives (declared "parity unpinned")."""
import struct

import numpy as np
import pytest

from bramble_amd import lib
from oracle import oracle_binding as ob

CODES = "=ACMGRSVTWYHKDBN"
COMP = {1: 8, 2: 4, 4: 2, 8: 1}            # A<->T, C<->G; everything else becomes N (15)  (src/bam.cpp:658-667)


def pack_seq(seq):
    out = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        out[i >> 1] |= CODES.index(ch) << (4 if i % 2 == 0 else 0)
    return bytes(out)


def record(tid, pos0, name, mapq, bin_, flag, cigar_words, seq, qual, aux, mtid=-1, mpos=-1, tlen=0):
    nm = name.encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", tid, pos0, len(nm), mapq, bin_, len(cigar_words), flag, len(seq), mtid, mpos, tlen)
    body += nm + b"".join(struct.pack("<I", w) for w in cigar_words) + pack_seq(seq) + bytes(qual) + aux
    return struct.pack("<I", len(body)) + body


ANN = {"refnames": ["chr1"], "transcripts": [{"id": "plus", "ref_id": 0, "strand": "+", "exons": [[1000, 1100]]},
                                             {"id": "minus", "ref_id": 0, "strand": "-", "exons": [[5000, 5100]]}]}
SEQ1, QUAL1 = "ACGTNACGTTA", [10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]          # odd length
AUX_IN = b"NMC\x02" + b"NHC\x05" + b"XSA+" + b"MDZ11\x00" + b"HIC\x03"   # XS:A:+ : the read is tried on "+" transcripts only
M = lambda n: (n << 4) | 0


def build_input():
    # read a: on the '+' transcript, secondary in the input; read b: on the '-' transcript, reverse strand in the input
    a = record(0, 1009, "ra", 17, 4681, 0x100, [M(11)], SEQ1, QUAL1, AUX_IN)
    b = record(0, 5019, "rb", 60, 4681, 0x10, [(2 << 4) | 4, M(9)], SEQ1, QUAL1, b"NHi\x01\x00\x00\x00")
    # a proper pair on the '+' transcript: read1 forward at 1011 (1-based), read2 reverse at 1041, 50 bases each: transcript positions 11 and 41
    p1 = record(0, 1010, "pp", 60, 4681, 0x1 | 0x40 | 0x20, [M(50)], "ACGTA" * 10, [30] * 50, b"", mtid=0, mpos=1040, tlen=80)
    p2 = record(0, 1040, "pp", 60, 4681, 0x1 | 0x80 | 0x10, [M(50)], "TTGCA" * 10, [31] * 50, b"", mtid=0, mpos=1010, tlen=-80)
    return np.frombuffer(a + b + p1 + p2, dtype=np.uint8)


def expected_stream():
    def tag_i(tag, v):
        return tag + b"i" + struct.pack("<i", v)
    # a: tid 0 ('plus'), transcript pos = 1010 - 1000 = 10, NH = 1 -> MAPQ 255, primary (secondary bit cleared), unpaired:
    # mate fields reset (-1, -1, 0); aux: the first NH and XS are deleted, NH:i and HI:i appended in that order
    aux_a = b"NMC\x02" + b"MDZ11\x00" + b"HIC\x03"
    aux_a = aux_a.replace(b"HIC\x03", b"") + tag_i(b"NH", 1) + tag_i(b"HI", 1)      # HI is replaced: deleted, then appended
    out_a = record(0, 10, "ra", 255, 4681, 0x000, [M(11)], SEQ1, QUAL1, aux_a)
    # b: tid 1 ('minus'): 2S9M at 5020..5028 -> exon [5000, 5100) on '-': pos = (5100 - 5029) + 0 = 71; the record is reverse-
    # complemented: CIGAR op order reversed, SEQ complemented and reversed (N stays N), QUAL reversed, reverse flag toggled
    rc = "".join(CODES[COMP.get(CODES.index(c), 15)] for c in reversed(SEQ1))
    out_b = record(1, 71, "rb", 255, 4681, 0x00, [M(9), (2 << 4) | 4], rc, list(reversed(QUAL1)), tag_i(b"NH", 1) + tag_i(b"HI", 1))
    # the pair: both mates on 'plus' (same transcript): paired + proper-pair bits set, mate fields = the mate's transcript
    # coordinates, tlen = +-(distance between the outer ends) (src/bam.cpp:531-588); hit_index runs over records, the
    # mate included (src/core.cpp:250-258): HI 1 and 2, NH 2 -> MAPQ 3 (src/core.cpp:46-58)
    out_p1 = record(0, 11, "pp", 3, 4681, 0x1 | 0x2 | 0x40 | 0x20, [M(50)], "ACGTA" * 10, [30] * 50, tag_i(b"NH", 2) + tag_i(b"HI", 1),
                    mtid=0, mpos=41, tlen=80)
    out_p2 = record(0, 41, "pp", 3, 4681, 0x1 | 0x2 | 0x80 | 0x10, [M(50)], "TTGCA" * 10, [31] * 50, tag_i(b"NH", 2) + tag_i(b"HI", 2),
                    mtid=0, mpos=11, tlen=-80)
    return np.frombuffer(out_a + out_b + out_p1 + out_p2, dtype=np.uint8)


def test_oracle_record_bytes_follow_the_rules():
    stream = build_input()
    roff, rlen, _, _ = lib.bam_split(stream)
    orc, _, _, _ = ob.run_bam(ob.OracleIndex(ANN), ob.make_flags(), stream, roff, rlen, np.array([0], dtype=np.int32))
    exp = expected_stream()
    assert orc["n_rows"] == 4
    assert orc["bam_stream"].tobytes() == exp.tobytes()


@pytest.mark.gpu
def test_device_record_bytes_follow_the_rules():
    stream = build_input()
    roff, rlen, _, _ = lib.bam_split(stream)
    idx = lib.Index(ANN, device=0)
    ctx = lib.Context(idx)
    got, counters = ctx.project_bam_bundle(lib.make_config(), stream, roff, rlen, np.array([0], dtype=np.int32))
    ctx.close()
    idx.close()
    assert got.tobytes() == expected_stream().tobytes()
