"""Scope-table row f-1 (BAM bundle entry): raw BAM records in, raw projected records out, everything between on
the device -- against the oracle's restatement of the reader side (process_reads / process_read_in /
process_pairs), convert_reads and write_to_bam.  Streams must be byte-identical."""
import numpy as np
import pytest

from bramble_amd import lib, synth
from oracle import oracle_binding as ob

pytestmark = pytest.mark.gpu


def framed_stream(b, unmapped_every=0):
    """synth records (BAM layout from refID on, contiguous) -> uncompressed BAM alignment section
    [block_size][record]..., optionally with unmapped copies sprinkled in (process_reads skips those)."""
    blob = np.asarray(b["rec_blob"], dtype=np.uint8)
    off = np.asarray(b["rec_off"], dtype=np.uint64)
    out = bytearray()
    n = len(off) - 1
    for i in range(n):
        rec = blob[int(off[i]):int(off[i + 1])].tobytes()
        if unmapped_every and i % unmapped_every == 0:
            un = bytearray(rec)
            flag = un[14] | (un[15] << 8) | 0x4
            un[14], un[15] = flag & 0xff, flag >> 8
            out += len(un).to_bytes(4, "little") + bytes(un)
        out += len(rec).to_bytes(4, "little") + rec
    return np.frombuffer(bytes(out), dtype=np.uint8)


def run_both_bam(ann, stream, ref_map, **flags):
    roff, rlen, n_un, used = lib.bam_split(stream)
    assert used == stream.size
    idx = lib.Index(ann, device=0)
    ctx = lib.Context(idx)
    got, counters = ctx.project_bam_bundle(lib.make_config(**flags), stream, roff, rlen, ref_map)
    ctx.close()
    idx.close()
    orc, _, _, parsed = ob.run_bam(ob.OracleIndex(ann), ob.make_flags(**flags), stream, roff, rlen, ref_map)
    return got, counters, orc, n_un


def assert_streams_equal(got, exp):
    assert len(got) == len(exp)
    if not np.array_equal(got, exp):
        bad = int(np.nonzero(got != exp)[0][0])
        raise AssertionError("BAM streams differ first at byte %d of %d" % (bad, len(exp)))


@pytest.mark.parametrize("mode,flags,kw", [("pe", {}, {"xs_tag": True}), ("pe", {}, {}), ("pe", {"fr": 1}, {}),
                                           ("se", {"strict": 1}, {}), ("hifi", {"lr_hq": 1}, {}), ("ont", {"lr": 1}, {})])
def test_bam_bundle_matches_oracle(mode, flags, kw):
    ann = synth.Annotation("G", n_genes=1500, n_refs=3)
    b = ann.reads(8000, mode, with_records=1, **kw)
    stream = framed_stream(b, unmapped_every=97)
    got, counters, orc, n_un = run_both_bam(ann.as_dict(), stream, np.arange(3, dtype=np.int32), **flags)
    assert n_un > 0 and orc["n_rows"] > 5000
    assert counters["n_rows"] == orc["n_rows"]
    for k in ("total_complete", "total_unique", "dropped_reads", "total_processed"):
        assert counters[k] == orc[k], k
    assert_streams_equal(got, orc["bam_stream"])


def test_bam_bundle_reader_side_equals_flat_batch():
    """The oracle fed with raw records and the oracle fed with the generator's flat batch agree: the synthetic
    records carry the same alignments as the SoA tables the other parity tests use."""
    ann = synth.Annotation("G", n_genes=800, n_refs=3)
    b = ann.reads(4000, "pe", with_records=1, xs_tag=True)
    stream = framed_stream(b)
    roff, rlen, _, _ = lib.bam_split(stream)
    oi = ob.OracleIndex(ann.as_dict())
    r1, _, _, _ = ob.run_bam(oi, ob.make_flags(), stream, roff, rlen, np.arange(3, dtype=np.int32))
    r2, _, _ = ob.run(oi, ob.make_flags(), b, want_matches=False, bam_records=(b["rec_blob"], b["rec_off"]))
    assert np.array_equal(r1["bam_stream"], r2["bam_stream"])


def test_bam_bundle_clip_rescue_reads_sequence_from_records():
    """-S: the query bases come from the records' 4-bit SEQ (k_seq_src / k_seq_ascii)."""
    ann = synth.Annotation("G", n_genes=300, n_refs=2, with_genome=True)
    b = ann.reads(3000, "ont", with_seq=1, with_records=1)
    stream = framed_stream(b)
    got, counters, orc, _ = run_both_bam(ann.as_dict(), stream, np.arange(2, dtype=np.int32), lr=1, use_fasta=1)
    assert (orc["clip_score"] != 0).sum() > 500
    assert_streams_equal(got, orc["bam_stream"])


def test_bam_bundle_reference_names_the_annotation_lacks():
    """Input references that are not in the annotation map to ids without intervals: their records vanish
    (g2tTree::getGuideExons returns false for an unknown refid, src/g2t.cpp:334-344)."""
    ann = synth.Annotation("G", n_genes=800, n_refs=3)
    b = ann.reads(3000, "pe", with_records=1)
    stream = framed_stream(b)
    ref_map = np.array([0, -1, 7], dtype=np.int32)   # ref 1 unknown, ref 2 -> id past the annotation's table
    got, counters, orc, _ = run_both_bam(ann.as_dict(), stream, ref_map)
    assert 0 < orc["n_rows"]
    assert_streams_equal(got, orc["bam_stream"])


def _record(name, ref, pos0, flag, cigar, mref, mpos0, seq_len):
    ops = "MIDNSHP=X"
    cg = []
    num = ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            cg.append((int(num) << 4) | ops.index(ch))
            num = ""
    nm = name.encode() + b"\0"
    body = bytearray()
    body += int(ref).to_bytes(4, "little", signed=True) + int(pos0).to_bytes(4, "little", signed=True)
    body += bytes([len(nm), 60]) + (4680).to_bytes(2, "little") + len(cg).to_bytes(2, "little") + int(flag).to_bytes(2, "little")
    body += int(seq_len).to_bytes(4, "little") + int(mref).to_bytes(4, "little", signed=True)
    body += int(mpos0).to_bytes(4, "little", signed=True) + (0).to_bytes(4, "little")
    body += nm
    for w in cg:
        body += int(w).to_bytes(4, "little")
    body += bytes([0x12] * ((seq_len + 1) // 2)) + bytes([30] * seq_len)
    body += b"NMC\x00"
    return len(body).to_bytes(4, "little") + bytes(body)


def test_bam_bundle_large_name_group_pairs_like_the_hash_map():
    """One read name with several hundred multi-mapping pair records (k_mates_big: wave-wide open-set search) next to
    small groups, including duplicate positions (a later record overwrites the map entry of an earlier one)."""
    txs = [{"id": "t%d" % t, "ref_id": 0, "strand": "+", "exons": [[1000 + 7 * t, 1400 + 7 * t], [2000, 2300]]}
           for t in range(40)]
    ann = {"refnames": ["chr1"], "transcripts": txs}
    rng = np.random.RandomState(11)
    out = bytearray()
    # big group: 150 pairs, mates listed in a shuffled order, some positions duplicated
    pairs = []
    for k in range(150):
        p1 = 1000 + int(rng.randint(0, 60))
        p2 = 1200 + int(rng.randint(0, 60))
        pairs.append((p1, p2))
    recs = []
    for p1, p2 in pairs:
        recs.append(("big", p1, 0x1 | 0x40 | 0x20, p2))
        recs.append(("big", p2, 0x1 | 0x80 | 0x10, p1))
    order = rng.permutation(len(recs))
    for i in order:
        name, pos, flag, mpos = recs[int(i)]
        out += _record(name, 0, pos, flag, "50M", 0, mpos, 50)
    # small groups
    for g in range(200):
        p1 = 1000 + int(rng.randint(0, 100))
        p2 = 1150 + int(rng.randint(0, 100))
        out += _record("s%d" % g, 0, p1, 0x1 | 0x40 | 0x20, "50M", 0, p2, 50)
        if g % 9 == 0:   # duplicate of read1 at the same position: overwrites the map entry
            out += _record("s%d" % g, 0, p1, 0x1 | 0x40 | 0x20 | 0x100, "50M", 0, p2, 50)
        out += _record("s%d" % g, 0, p2, 0x1 | 0x80 | 0x10, "50M", 0, p1, 50)
    stream = np.frombuffer(bytes(out), dtype=np.uint8)
    got, counters, orc, _ = run_both_bam(ann, stream, np.array([0], dtype=np.int32))
    assert orc["n_rows"] > 1000
    assert_streams_equal(got, orc["bam_stream"])


def test_bam_split_stops_at_partial_record_and_rejects_garbage():
    ann = synth.Annotation("G", n_genes=200, n_refs=2)
    b = ann.reads(500, "pe", with_records=1)
    stream = framed_stream(b)
    roff, rlen, _, used = lib.bam_split(stream[:-5])
    full_off, full_len, _, full_used = lib.bam_split(stream)
    assert len(roff) == len(full_off) - 1 and used == int(full_off[-1]) - 4
    assert full_used == stream.size
    bad = stream.copy()
    bad[0:4] = np.frombuffer((7).to_bytes(4, "little"), dtype=np.uint8)   # block_size < 32
    with pytest.raises(lib.BrambleError):
        lib.bam_split(bad)


def test_bam_bundle_empty():
    idx = lib.Index({"refnames": ["chr1"], "transcripts": [{"id": "t", "ref_id": 0, "strand": "+", "exons": [[10, 50]]}]},
                    device=0)
    ctx = lib.Context(idx)
    got, counters = ctx.project_bam_bundle(lib.make_config(), np.zeros(0, np.uint8), np.zeros(0, np.uint64),
                                           np.zeros(0, np.uint32), np.array([0], np.int32))
    assert got.size == 0 and counters["n_rows"] == 0
    ctx.close()
    idx.close()


def _rec_full(name, ref, pos0, flag, cigar, mref, mpos0, seq, qual, aux, mapq=37):
    """One BAM record from explicit parts (seq: ASCII or '', qual: bytes / None for 0xff fill, aux: raw bytes)."""
    ops = "MIDNSHP=X"
    cg, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            cg.append((int(num) << 4) | ops.index(ch))
            num = ""
    nm = name.encode() + b"\0"
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    ls = len(seq)
    packed = bytearray((ls + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= code[ch] << (4 if i % 2 == 0 else 0)
    q = bytes([0xff] * ls) if qual is None else bytes(qual)
    body = bytearray()
    body += int(ref).to_bytes(4, "little", signed=True) + int(pos0).to_bytes(4, "little", signed=True)
    body += bytes([len(nm), mapq]) + (4680).to_bytes(2, "little") + len(cg).to_bytes(2, "little") + int(flag).to_bytes(2, "little")
    body += int(ls).to_bytes(4, "little") + int(mref).to_bytes(4, "little", signed=True)
    body += int(mpos0).to_bytes(4, "little", signed=True) + (0).to_bytes(4, "little")
    body += nm
    for w in cg:
        body += int(w).to_bytes(4, "little")
    body += bytes(packed) + q + aux
    return len(body).to_bytes(4, "little") + bytes(body)


def test_bam_bundle_unusual_records():
    """Records real aligners produce that the generator does not: hard clips, =/X ops, odd and empty sequences, absent
    qualities, IUPAC codes, long and one-character names, every aux value type (incl. Z strings holding tag-like text,
    H, B arrays, f), repeated NH tags, XS as Z, no aux at all, no CIGAR."""
    txs = [{"id": "fw", "ref_id": 0, "strand": "+", "exons": [[1000, 1200], [2000, 2200], [3000, 3300]]},
           {"id": "rv", "ref_id": 0, "strand": "-", "exons": [[900, 1200], [2000, 2200]]},
           {"id": "rv2", "ref_id": 0, "strand": "-", "exons": [[5000, 5400]]}]
    ann = {"refnames": ["chr1"], "transcripts": txs}
    rng = np.random.RandomState(3)

    def rs(n):
        return "".join("ACGTNRYK"[int(x)] for x in rng.randint(0, 8, size=n))
    i32 = lambda v: int(v).to_bytes(4, "little", signed=True)
    recs = []
    aux_all = (b"NHC\x03" + b"XSA-" + b"ZZZNH:i:1 HI\x00" + b"HHH1AE3\x00" + b"fff" + b"\x00\x00\x80?" + b"BBBs\x03\x00\x00\x00\x01\x00\x02\x00\x03\x00" +
               b"ASs\xfe\xff" + b"HIi" + i32(7) + b"NHS\x09\x00" + b"tsA+" + b"BIBI\x01\x00\x00\x00\xff\xff\xff\xff" + b"cc" + b"c\x80")
    recs.append(_rec_full("n1", 0, 1049, 0, "3H5S40=2X53M", -1, -1, rs(100), rng.randint(2, 40, 100).astype(np.uint8).tobytes(), aux_all))
    recs.append(_rec_full("n2", 0, 5099, 16, "101M", -1, -1, rs(101), rng.randint(2, 40, 101).astype(np.uint8).tobytes(), b"XSZ-foo\x00NMC\x01"))
    recs.append(_rec_full("n3", 0, 5099, 0, "33M", -1, -1, rs(33), None, b""))                          # odd length, no quals, no aux
    recs.append(_rec_full("n3", 0, 5149, 256, "33M", -1, -1, "", None, b"NHC\x02"))                     # secondary without SEQ
    recs.append(_rec_full("q" * 250, 0, 1149, 0, "51M800N50M", -1, -1, rs(101), rng.randint(2, 40, 101).astype(np.uint8).tobytes(), b"XSA+"))
    recs.append(_rec_full("x", 0, 949, 0x1 | 0x40, "1M", 0, 1999, rs(1), bytes([30]), b"NHi" + i32(1)))
    recs.append(_rec_full("x", 0, 1999, 0x1 | 0x80 | 0x10, "7M", 0, 949, rs(7), bytes([30] * 7), b""))
    recs.append(_rec_full("nocigar", 0, 1000, 0, "", -1, -1, rs(10), bytes([1] * 10), b"NHC\x01"))
    recs.append(_rec_full("clipH", 0, 2049, 0, "10H100M10H", -1, -1, rs(100), rng.randint(2, 40, 100).astype(np.uint8).tobytes(), b"MDZ100\x00"))
    recs.append(_rec_full("del", 0, 1099, 0, "50M3D2I48M", -1, -1, rs(100), rng.randint(2, 40, 100).astype(np.uint8).tobytes(), b"tsA-"))
    stream = np.frombuffer(b"".join(recs), dtype=np.uint8)
    for flags in ({}, {"lr": 1}, {"strict": 1}, {"fr": 1}):
        got, counters, orc, _ = run_both_bam(ann, stream, np.array([0], dtype=np.int32), **flags)
        assert orc["n_rows"] >= 6
        assert_streams_equal(got, orc["bam_stream"])


def test_staged_bundles_equal_one_shot_calls():
    """br_bam_bundle_stage / br_project_bam_staged: three bundles staged ahead into the three device slots give the
    streams of three one-shot br_project_bam_bundle calls; projecting a slot that holds another bundle is refused."""
    import ctypes as C
    ann = synth.Annotation("G", n_genes=600, n_refs=3)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    L = lib.lib()
    L.br_bam_bundle_stage.argtypes = [C.c_void_p, C.POINTER(lib.BrBamBundle), C.c_int]
    L.br_project_bam_staged.argtypes = [C.c_void_p, C.POINTER(lib.BrConfig), C.POINTER(lib.BrBamBundle), C.c_int, C.POINTER(lib.BrHostBam)]
    ref_map = np.arange(3, dtype=np.int32)
    bundles, expect, keep = [], [], []
    for k, n in enumerate((1500, 400, 2600)):
        b = ann.reads(n, "pe", with_records=1, seed=77 + k)
        stream, roff, rlen = synth.Annotation.frame_records(b)
        exp, _ = ctx.project_bam_bundle(cfg, stream, roff, rlen, ref_map)
        expect.append(exp)
        keep.append((stream, roff, rlen))
        bundles.append(lib.BrBamBundle(stream.ctypes.data, stream.size, roff.ctypes.data, rlen.ctypes.data, len(rlen), ref_map.ctypes.data, 3, 0))
    for k, bb in enumerate(bundles):
        assert L.br_bam_bundle_stage(ctx.h, C.byref(bb), k) == 0
    out = lib.BrHostBam()
    assert L.br_project_bam_staged(ctx.h, C.byref(cfg), C.byref(bundles[0]), 1, C.byref(out)) == -1     # slot 1 holds bundle 1
    for k in (2, 0, 1):
        assert L.br_project_bam_staged(ctx.h, C.byref(cfg), C.byref(bundles[k]), k, C.byref(out)) == 0
        got = np.ctypeslib.as_array(C.cast(out.data, C.POINTER(C.c_uint8)), shape=(int(out.n_bytes),)).copy()
        assert_streams_equal(got, expect[k])
    ctx.close()
    idx.close()


def test_results_that_come_home_later_equal_the_waiting_call():
    """br_project_bam_staged_nowait + br_host_bam_wait (the command line's form: the BGZF blocks of bundle k cross PCIe beside
    the kernels of bundle k + 1, from a second device buffer) against br_project_bam_staged, bundle by bundle: the same bytes
    and counters, also when nobody ever waits for a result before its buffer comes round again."""
    import ctypes as C
    import zlib
    ann = synth.Annotation("G", n_genes=600, n_refs=3)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    L = lib.lib()
    L.br_bam_bundle_stage.argtypes = [C.c_void_p, C.POINTER(lib.BrBamBundle), C.c_int]
    sig = [C.c_void_p, C.POINTER(lib.BrConfig), C.POINTER(lib.BrBamBundle), C.c_int, C.POINTER(lib.BrHostBam)]
    L.br_project_bam_staged.argtypes = sig
    L.br_project_bam_staged_nowait.argtypes = sig
    L.br_host_bam_wait.argtypes = [C.c_void_p, C.POINTER(lib.BrHostBam)]
    ref_map = np.arange(3, dtype=np.int32)
    keep, bundles = [], []
    for k, n in enumerate((1800, 300, 2500, 900, 1200)):
        b = ann.reads(n, "pe", with_records=1, seed=501 + k)
        stream, roff, rlen = synth.Annotation.frame_records(b)
        keep.append((stream, roff, rlen))
        bundles.append(lib.BrBamBundle(stream.ctypes.data, stream.size, roff.ctypes.data, rlen.ctypes.data, len(rlen), ref_map.ctypes.data, 3, 1))
    take = lambda o: np.ctypeslib.as_array(C.cast(o.data, C.POINTER(C.c_uint8)), shape=(int(o.n_bytes),)).copy()
    fields = ("n_bytes", "n_rows", "total_complete", "total_unique", "dropped_reads", "total_processed")
    expect = []
    for k, bb in enumerate(bundles):
        out = lib.BrHostBam()
        assert L.br_bam_bundle_stage(ctx.h, C.byref(bb), k % 3) == 0
        assert L.br_project_bam_staged(ctx.h, C.byref(cfg), C.byref(bb), k % 3, C.byref(out)) == 0
        expect.append((take(out), tuple(int(getattr(out, f)) for f in fields)))
        assert out.n_bytes > 0
    # one call ahead, as the command line's writer does: wait for bundle k only after bundle k + 1 has been launched
    outs = [lib.BrHostBam() for _ in bundles]
    for k, bb in enumerate(bundles):
        assert L.br_bam_bundle_stage(ctx.h, C.byref(bb), k % 3) == 0
        assert L.br_project_bam_staged_nowait(ctx.h, C.byref(cfg), C.byref(bb), k % 3, C.byref(outs[k])) == 0
        assert tuple(int(getattr(outs[k], f)) for f in fields) == expect[k][1]
        if k:
            assert L.br_host_bam_wait(ctx.h, C.byref(outs[k - 1])) == 0
            assert np.array_equal(take(outs[k - 1]), expect[k - 1][0])
    assert L.br_host_bam_wait(ctx.h, C.byref(outs[-1])) == 0
    last = take(outs[-1])
    assert np.array_equal(last, expect[-1][0])
    assert L.br_host_bam_wait(ctx.h, C.byref(outs[-1])) == 0             # a second wait is a no-op
    # the blocks inflate to a BAM record stream (any inflater: tests/test_gpu_codec.py checks the codec itself)
    raw = zlib.decompressobj(31).decompress(bytes(last[:int.from_bytes(bytes(last[16:18]), "little") + 1]))
    assert len(raw) > 36 and int.from_bytes(raw[:4], "little") >= 32
    # nobody waits: the context itself waits before a download buffer is used again
    for k, bb in enumerate(bundles):
        o = lib.BrHostBam()
        assert L.br_bam_bundle_stage(ctx.h, C.byref(bb), k % 3) == 0
        assert L.br_project_bam_staged_nowait(ctx.h, C.byref(cfg), C.byref(bb), k % 3, C.byref(o)) == 0
    assert L.br_host_bam_wait(ctx.h, C.byref(o)) == 0
    assert np.array_equal(take(o), expect[-1][0])
    ctx.close()
    idx.close()


@pytest.mark.parametrize("seed,flags", [(11, {}), (12, {"lr": 1}), (13, {"strict": 1})])
def test_bam_tasks_random_records(seed, flags):
    """k_bam_tasks (a wave per 32 rows, their byte regions as 16-byte copy tasks) on records built to hit every segment kind:
    odd and even SEQ lengths on both strands, records whose bases are only A C G T N and records with '=' / IUPAC codes,
    absent qualities, one-byte to 40-byte names, CIGARs that stay above two ops after the rewrite (soft clips on both
    sides, insertions: the arena path, reversed on '-'), aux areas with the removed tags first, last, in the middle,
    repeated or missing, and several isoforms per locus so that a wave holds rows of few records.  Both encoders
    (bam_lanes 0 = k_bam_tasks, 8 = k_bam_encode<8>) against the oracle, byte for byte."""
    rng = np.random.RandomState(seed)
    txs = []
    for g in range(12):
        base = 1000 + 6000 * g
        strand = "+" if g % 2 == 0 else "-"
        for iso in range(1 + g % 4):
            txs.append({"id": "g%d.%d" % (g, iso), "ref_id": 0, "strand": strand,
                        "exons": [[base, base + 900 + 10 * iso], [base + 2000, base + 2700], [base + 4000 - 3 * iso, base + 4800]]})
    ann = {"refnames": ["chr1"], "transcripts": txs}
    i32 = lambda v: int(v).to_bytes(4, "little", signed=True)
    clean, dirty = "ACGTN", "=ACMGRSVTWYHKDBN"
    tags = [b"NHC\x02", b"HIi" + i32(3), b"XSA+", b"XSA-", b"tsA-", b"ASs\x30\x00", b"ASC\x07", b"NMC\x01", b"MDZ10A5\x00",
            b"RGZgrp1\x00", b"XXBs\x02\x00\x00\x00\x01\x00\x02\x00", b"ffff\x00\x00\x80?"]
    recs = []
    for k in range(2500):
        g = int(rng.randint(0, 12))
        base = 1000 + 6000 * g
        ls = int(rng.randint(1, 181))
        form = int(rng.randint(0, 5))
        alphabet = clean if rng.rand() < 0.5 else dirty
        seq = "".join(alphabet[int(x)] for x in rng.randint(0, len(alphabet), size=ls))
        qual = None if rng.rand() < 0.2 else rng.randint(0, 42, ls).astype(np.uint8).tobytes()
        if form == 0 or ls < 8:
            cig = "%dM" % ls
        elif form == 1:
            a = int(rng.randint(1, ls // 2)); cig = "%dS%dM" % (a, ls - a)
        elif form == 2:
            a = int(rng.randint(1, ls // 3 + 1)); c = int(rng.randint(1, ls // 3 + 1)); cig = "%dS%dM%dS" % (a, ls - a - c, c)
        elif form == 3:
            a = int(rng.randint(1, ls - 3)); cig = "%dM2I%dM" % (a, ls - a - 2) if ls - a - 2 > 0 else "%dM" % ls
        else:
            a = int(rng.randint(1, ls // 2)); cig = "%dM1D%dM" % (a, ls - a)
        pos0 = base + int(rng.randint(0, 600))
        flag = (16 if rng.rand() < 0.5 else 0) | (256 if rng.rand() < 0.1 else 0)
        n_tags = int(rng.randint(0, 7))
        aux = b"".join(tags[int(t)] for t in rng.randint(0, len(tags), size=n_tags))
        name = "r%d" % k + "x" * int(rng.randint(0, 36))
        recs.append(_rec_full(name, 0, pos0, flag, cig, -1, -1, seq, qual, aux))
    stream = np.frombuffer(b"".join(recs), dtype=np.uint8)
    roff, rlen, n_un, used = lib.bam_split(stream)
    idx = lib.Index(ann, device=0)
    orc, _, _, parsed = ob.run_bam(ob.OracleIndex(ann), ob.make_flags(**flags), stream, roff, rlen, np.array([0], dtype=np.int32))
    assert orc["n_rows"] > 2000
    for lanes in (0, 8):
        ctx = lib.Context(idx)
        ctx.set_param("bam_lanes", lanes)
        got, counters = ctx.project_bam_bundle(lib.make_config(**flags), stream, roff, rlen, np.array([0], dtype=np.int32))
        ctx.close()
        assert_streams_equal(got, orc["bam_stream"])
    idx.close()
