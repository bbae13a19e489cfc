"""br_bam_split_device (segments guess their first record, a verification pass against the real chain makes it exact) against
br_bam_split (host, the serial chain walk): offsets, lengths, unmapped counts and the bytes consumed, on streams with
unmapped records, records longer than a segment, a partial record at the end, payloads that look like record headers, tiny
inputs; a malformed record is refused like on the host."""
import ctypes as C

import numpy as np
import pytest
import torch

from bramble_amd import lib, synth
from bramble_amd.device import _DevArray
from tests import bamio

pytestmark = pytest.mark.gpu


def _ctx():
    idx = lib.Index({"refnames": ["chr1"], "transcripts": [{"id": "t", "ref_id": 0, "strand": "+", "exons": [[10, 50]]}]}, device=0)
    return idx, lib.Context(idx)


def host_split(stream):
    L = lib.lib()
    L.br_bam_split.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    cap = stream.size // 36 + 1
    off = np.zeros(cap, dtype=np.uint64); ln = np.zeros(cap, dtype=np.uint32)
    n, un, used = C.c_int64(), C.c_int64(), C.c_uint64()
    rc = L.br_bam_split(stream.ctypes.data, stream.size, cap, off.ctypes.data, ln.ctypes.data, C.byref(n), C.byref(un), C.byref(used))
    return rc, off[:n.value].copy(), ln[:n.value].copy(), un.value, used.value


def device_split(ctx, stream, n_ref):
    L = lib.lib()
    L.br_bam_split_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(lib.BrDeviceRecords), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    d = torch.from_numpy(stream.copy()).to("cuda:0") if stream.size else torch.zeros(1, dtype=torch.uint8, device="cuda:0")
    recs = lib.BrDeviceRecords(); un, used = C.c_int64(), C.c_uint64()
    rc = L.br_bam_split_device(ctx.h, C.c_void_p(d.data_ptr()), stream.size, n_ref, None, C.byref(recs), C.byref(un), C.byref(used))
    if rc:
        return rc, None, None, None, None
    n = int(recs.n_aln)
    off = torch.as_tensor(_DevArray(recs.rec_off, n, "<u8"), device="cuda:0").cpu().numpy().copy() if n else np.zeros(0, np.uint64)
    ln = torch.as_tensor(_DevArray(recs.rec_len, n, "<u4"), device="cuda:0").cpu().numpy().copy() if n else np.zeros(0, np.uint32)
    return rc, off.astype(np.uint64), ln.astype(np.uint32), un.value, used.value


def check(ctx, stream, n_ref=3):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    h = host_split(stream)
    d = device_split(ctx, stream, n_ref)
    assert d[0] == h[0]
    if h[0] == 0:
        assert np.array_equal(d[1], h[1]) and np.array_equal(d[2], h[2]) and d[3] == h[3] and d[4] == h[4]
    return h


def records_stream(n_pairs, seed, n_refs=3):
    ann = synth.Annotation("G", n_genes=300, n_refs=n_refs)
    b = ann.reads(n_pairs, "pe", with_records=1, seed=seed)
    stream, roff, rlen = synth.Annotation.frame_records(b)
    return stream


def test_short_read_streams_and_every_tail():
    idx, ctx = _ctx()
    stream = records_stream(4000, 3)
    h = check(ctx, stream)
    assert len(h[1]) > 7000 and h[4] == stream.size
    for cut in (1, 3, 4, 5, 35, 36, 37, 200, 32768, 40000):      # a partial record (or a partial length field) at the end
        check(ctx, stream[:stream.size - cut])
    for n in (0, 1, 3, 4, 36, 200):                                # less than a record
        check(ctx, stream[:n])
    ctx.close(); idx.close()


def test_unmapped_records_and_records_longer_than_a_segment():
    idx, ctx = _ctx()
    rng = np.random.RandomState(8)
    recs = []
    for k in range(3000):
        l_seq = int(rng.choice([50, 100, 151, 3000, 40000, 90000], p=[0.3, 0.3, 0.3, 0.05, 0.03, 0.02]))
        flag = 4 if k % 7 == 3 else (16 if k % 2 else 0)
        recs.append(bamio.bam_record(("r%05d" % (k // 2)).encode(), int(rng.randint(0, 3)), int(rng.randint(0, 10 ** 6)), [(l_seq << 4) | 0], l_seq, flag=flag,
                                     seq=rng.randint(0, 256, size=(l_seq + 1) // 2).astype(np.uint8).tobytes(), qual=rng.randint(0, 42, size=l_seq).astype(np.uint8).tobytes(),
                                     aux=b"NHC\x01"))
    stream = bamio.frame(recs)
    h = check(ctx, stream)
    assert h[3] == sum(1 for k in range(3000) if k % 7 == 3) and h[4] == stream.size
    check(ctx, stream[:stream.size - 12345])
    ctx.close(); idx.close()


def test_payloads_that_look_like_records_do_not_fool_the_result():
    """Qualities are free bytes: a run of them shaped like a record's fixed fields (and a chain of three) sits where a segment
    starts looking.  The guess takes the bait; the verification pass does not."""
    idx, ctx = _ctx()
    fake = bamio.frame([bamio.bam_record(b"x", 0, 5, [(4 << 4) | 0], 4, qual=b"\x05" * 4)]).tobytes()
    bait = (fake * 12)
    recs = []
    for k in range(600):
        l_seq = 400
        qual = bytearray(np.random.RandomState(k).randint(0, 42, size=l_seq).astype(np.uint8).tobytes())
        qual[7:7 + len(bait)] = bait[:l_seq - 7]
        recs.append(bamio.bam_record(("q%04d" % k).encode(), 1, 100 + k, [(l_seq << 4) | 0], l_seq, qual=bytes(qual), aux=b"NHC\x01"))
    stream = bamio.frame(recs)
    h = check(ctx, stream)
    assert len(h[1]) == 600
    ctx.close(); idx.close()


def test_a_malformed_record_is_refused_like_on_the_host():
    idx, ctx = _ctx()
    stream = records_stream(600, 5).copy()
    h = host_split(stream)
    k = len(h[1]) // 2
    at = int(h[1][k]) - 4
    bad = stream.copy(); bad[at:at + 4] = np.frombuffer((20).to_bytes(4, "little"), dtype=np.uint8)      # block_size < 32
    assert host_split(bad)[0] != 0
    assert device_split(ctx, bad, 3)[0] != 0
    bad = stream.copy(); bad[at + 4 + 8] = 0                                                               # l_read_name = 0
    assert host_split(bad)[0] != 0
    assert device_split(ctx, bad, 3)[0] != 0
    check(ctx, stream)                                                                                     # the context still works
    ctx.close(); idx.close()


def _bgzf(data, chunk=0xff00):
    import struct
    import zlib
    out = []
    for p in range(0, len(data), chunk):
        piece = data[p:p + chunk]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = c.compress(piece) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload
                   + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
    return b"".join(out) + bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def _read_all(L, raw, n_ref, header_bytes, piece_blocks, feed):
    """Drives br_bam_reader over the BGZF bytes `raw`, handing it `feed` more bytes of the file at a time.  Returns the
    bundles as host copies: (record bytes list, n_unmapped) per bundle."""
    L.br_bam_reader_new.argtypes = [C.c_int, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]
    L.br_bam_reader_next.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(lib.BrDeviceRecords), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.br_bam_reader_release.argtypes = [C.c_void_p, C.c_int64]
    L.br_bam_reader_set_piece_blocks.argtypes = [C.c_void_p, C.c_int64]
    L.br_bam_reader_free.argtypes = [C.c_void_p]
    rd = C.c_void_p()
    assert L.br_bam_reader_new(0, n_ref, header_bytes, C.byref(rd)) == 0
    assert L.br_bam_reader_set_piece_blocks(rd, piece_blocks) == 0
    arr = np.frombuffer(raw, dtype=np.uint8)
    pos, avail, bundles, rc = 0, min(feed, arr.size), [], 0
    while True:
        last = 1 if avail == arr.size else 0
        used, recs, bid, unm = C.c_uint64(), lib.BrDeviceRecords(), C.c_int64(), C.c_int64()
        rc = L.br_bam_reader_next(rd, arr.ctypes.data + pos, avail - pos, last, C.byref(used), C.byref(recs), C.byref(bid), C.byref(unm))
        if rc:
            break
        pos += used.value
        if bid.value >= 0:
            n = int(recs.n_aln)
            out = []
            if n:
                off = torch.as_tensor(_DevArray(recs.rec_off, n, "<u8"), device="cuda:0").cpu().numpy()
                ln = torch.as_tensor(_DevArray(recs.rec_len, n, "<u4"), device="cuda:0").cpu().numpy()
                end = int(off[-1]) + int(ln[-1])
                blob = torch.as_tensor(_DevArray(recs.blob, end, "|u1"), device="cuda:0").cpu().numpy()
                out = [blob[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, ln)]
            bundles.append((out, unm.value))
            assert L.br_bam_reader_release(rd, bid.value) == 0
        if last and pos == arr.size:
            break
        if last:
            assert used.value > 0, "no progress"
        elif used.value == 0 or avail - pos < (1 << 16):
            avail = min(arr.size, avail + feed)
    L.br_bam_reader_free(rd)
    return rc, bundles


@pytest.mark.parametrize("piece_blocks,feed", [(3, 150_000), (1, 70_000), (40, 1 << 30), (3072, 1 << 30)])
def test_reader_bundles_are_the_mapped_records_cut_at_name_changes(piece_blocks, feed):
    stream = records_stream(3000, 11)
    rng = np.random.RandomState(2)
    # some unmapped records in between (flag bit 2 set), like a real file has
    recs = bamio.split_stream(stream)
    mixed = []
    for k, r in enumerate(recs):
        if k % 11 == 5:
            u = bytearray(r); u[14] |= 4; mixed.append(bytes(u))
        mixed.append(r)
    stream = bamio.frame(mixed)
    header = b"BAM\x01" + (25).to_bytes(4, "little") + b"@HD\tVN:1.6\tSO:unsorted\n\0\0" [:25] + (3).to_bytes(4, "little") + b"".join(
        (5).to_bytes(4, "little") + b"chr%d\0" % k + (10 ** 8).to_bytes(4, "little") for k in range(3))
    raw = _bgzf(header + stream.tobytes(), chunk=20000 if piece_blocks < 10 else 0xff00)
    rc, bundles = _read_all(lib.lib(), raw, 3, len(header), piece_blocks, feed)
    assert rc == 0
    h = host_split(stream)
    want = [stream[int(o):int(o) + int(l)].tobytes() for o, l in zip(h[1], h[2])]
    got = [r for b, _ in bundles for r in b]
    assert got == want
    assert sum(u for _, u in bundles) == h[3]
    name = lambda r: r[32:32 + r[8]]
    for (a, _), (b, _) in zip([x for x in bundles if x[0]][:-1], [x for x in bundles if x[0]][1:]):
        assert name(a[-1]) != name(b[0])          # no read-name group is split between bundles
    if piece_blocks < 10:
        assert len(bundles) > 10


def test_reader_refuses_truncated_and_corrupt_files():
    stream = records_stream(500, 12)
    header = b"BAM\x01" + (0).to_bytes(4, "little") + (1).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"chr1\0" + (10 ** 8).to_bytes(4, "little")
    raw = _bgzf(header + stream.tobytes())
    L = lib.lib()
    assert _read_all(L, raw, 3, len(header), 3072, 1 << 30)[0] == 0
    assert _read_all(L, raw[:len(raw) - 28 - 100], 3, len(header), 3072, 1 << 30)[0] != 0                 # ends inside a block
    cut = _bgzf(header + stream.tobytes()[:stream.size - 77])
    assert _read_all(L, cut, 3, len(header), 3072, 1 << 30)[0] != 0                                          # ends inside a record
    bad = bytearray(raw); bad[200] ^= 0x40
    assert _read_all(L, bytes(bad), 3, len(header), 3072, 1 << 30)[0] != 0                                   # a flipped bit


def test_wrong_guesses_are_repaired():
    """The context parameter "split_spoil" (test hook): after the honest guesses, every k-th segment forgets its entry and others take
    an offset that starts no record (runs of up to four in a row): whatever the walk from there does -- stops at once as
    malformed, runs off as a huge record, or happens to meet the chain again -- the check / repair passes must end with the
    host's result."""
    idx, ctx = _ctx()
    stream = records_stream(6000, 21)
    rng = np.random.RandomState(6)
    recs = bamio.split_stream(stream)
    big = [bamio.bam_record(b"long%d" % k, 1, 500 + k, [(70000 << 4) | 0], 70000, seq=rng.randint(0, 256, size=35000).astype(np.uint8).tobytes(),
                            qual=rng.randint(0, 42, size=70000).astype(np.uint8).tobytes()) for k in range(3)]
    mixed = recs[:4000] + big[:1] + recs[4000:9000] + big[1:] + recs[9000:]
    stream2 = bamio.frame(mixed)
    for k in (1, 2, 3, 5, 17):
        ctx.set_param("split_spoil", k)
        check(ctx, stream)
        check(ctx, stream2)
        check(ctx, stream2[:stream2.size - 9999])
    ctx.set_param("split_spoil", 0)
    check(ctx, stream2)
    ctx.close(); idx.close()


class _PieceInfo(C.Structure):
    _fields_ = [("start_rel", C.c_uint64), ("end_rel", C.c_uint64), ("n_unmapped", C.c_int64), ("guessed", C.c_int32), ("at_end", C.c_int32)]


def _pieces(L, raw, n_ref, header_bytes, piece_blocks, guess, extra=2):
    """The piece-wise reader on the BGZF bytes `raw`: pieces of `piece_blocks` blocks, each with a known start (the end of
    the piece in front) or a guessed one (guess: every piece but the first).  Returns (bundles as lists of record bytes,
    unmapped total, infos, how often a piece asked for more blocks)."""
    L.br_bam_reader_new.argtypes = [C.c_int, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]
    L.br_bam_piece_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    L.br_bam_piece_process.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(lib.BrDeviceRecords), C.POINTER(C.c_int64), C.POINTER(_PieceInfo)]
    L.br_bam_reader_release.argtypes = [C.c_void_p, C.c_int64]
    L.br_bam_reader_free.argtypes = [C.c_void_p]
    arr = np.frombuffer(raw, dtype=np.uint8)
    blocks, consumed, total = lib.bgzf_scan(arr)
    assert consumed == arr.size
    bl = np.ascontiguousarray(blocks, dtype=lib.BGZF_BLOCK)
    nb = len(bl)
    rd = C.c_void_p()
    assert L.br_bam_reader_new(0, n_ref, header_bytes, C.byref(rd)) == 0
    bundles, infos, unm, more = [], [], 0, 0
    n_pieces = (nb + piece_blocks - 1) // piece_blocks
    for k in range(n_pieces):
        b0, b1 = k * piece_blocks, min(nb, (k + 1) * piece_blocks)
        ex = extra
        start = header_bytes if k == 0 else (-1 if guess else int(infos[-1].end_rel))
        while True:
            b1x = min(nb, b1 + ex)
            assert L.br_bam_piece_upload(rd, k & 1, arr.ctypes.data, arr.size, bl.ctypes.data, nb, b0, b1x) == 0
            recs, bid, info = lib.BrDeviceRecords(), C.c_int64(), _PieceInfo()
            rc = L.br_bam_piece_process(rd, k & 1, bl.ctypes.data, nb, b1, start, C.byref(recs), C.byref(bid), C.byref(info))
            if rc == 1:                       # BR_PIECE_MORE
                more += 1; ex *= 8
                assert ex < 100000
                continue
            assert rc == 0, rc
            if guess and k and int(info.start_rel) != int(infos[-1].end_rel):   # a wrong guess: once more, from the true start
                assert L.br_bam_reader_release(rd, bid.value) == 0
                start = int(infos[-1].end_rel)
                continue
            break
        n = int(recs.n_aln)
        out = []
        if n:
            off = torch.as_tensor(_DevArray(recs.rec_off, n, "<u8"), device="cuda:0").cpu().numpy()
            ln = torch.as_tensor(_DevArray(recs.rec_len, n, "<u4"), device="cuda:0").cpu().numpy()
            end = int(off[-1]) + int(ln[-1])
            blob = torch.as_tensor(_DevArray(recs.blob, end, "|u1"), device="cuda:0").cpu().numpy()
            out = [blob[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, ln)]
        bundles.append(out); infos.append(info); unm += int(info.n_unmapped)
        assert L.br_bam_reader_release(rd, bid.value) == 0
    L.br_bam_reader_free(rd)
    return bundles, unm, infos, more


@pytest.mark.parametrize("piece_blocks,chunk,guess", [(3, 20000, False), (3, 20000, True), (1, 9000, True), (7, 0xff00, True), (2, 700, True), (1000, 0xff00, False)])
def test_pieces_tile_the_mapped_records_whoever_made_them(piece_blocks, chunk, guess):
    """br_bam_piece_upload / br_bam_piece_process directly: the pieces' bundles, in piece order, are the file's mapped
    records, cut at read-name changes only -- with every start known (one reader in order) and with every start guessed
    and checked against the neighbour's END (readers on several devices); blocks smaller than a record (a piece must
    ask for more of them) and multi-mapper name groups across piece boundaries included."""
    stream = records_stream(2500, 31)
    recs = bamio.split_stream(stream)
    mixed = []
    for k, r in enumerate(recs):
        if k % 13 == 4:
            u = bytearray(r); u[14] |= 4; mixed.append(bytes(u))            # an unmapped record in between
        mixed.append(r)
        if k % 97 == 50:                                                    # a long name group: six more records of the same name
            mixed.extend([r] * 6)
    stream = bamio.frame(mixed)
    header = b"BAM\x01" + (0).to_bytes(4, "little") + (3).to_bytes(4, "little") + b"".join(
        (5).to_bytes(4, "little") + b"chr%d\0" % k + (10 ** 8).to_bytes(4, "little") for k in range(3))
    raw = _bgzf(header + stream.tobytes(), chunk=chunk)
    bundles, unm, infos, more = _pieces(lib.lib(), raw, 3, len(header), piece_blocks, guess)
    h = host_split(stream)
    want = [stream[int(o):int(o) + int(l)].tobytes() for o, l in zip(h[1], h[2])]
    got = [r for b in bundles for r in b]
    assert got == want
    assert unm == h[3]
    name = lambda r: r[32:32 + r[8]]
    full = [b for b in bundles if b]
    for a, b in zip(full[:-1], full[1:]):
        assert name(a[-1]) != name(b[0])
    for a, b in zip(infos[:-1], infos[1:]):
        assert int(b.start_rel) == int(a.end_rel)
    assert infos[-1].at_end == 1
    if chunk == 700:
        assert more > 0          # blocks of 700 bytes: a name group at a boundary needs more than two of them
