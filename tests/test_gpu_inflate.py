"""BGZF inflate on the device (k_inflate: one wave per block) against zlib: stored, fixed and dynamic DEFLATE blocks, several
per BGZF block, matches at every distance class (window / far back / overlapping runs), codes longer than the primary tables,
the library's own writers (host libdeflate framing, device k_deflate_dynamic) read back, and corrupt input refused.  The
block table comes from br_bgzf_scan (host), checked here too."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from bramble_amd import lib, synth

pytestmark = pytest.mark.gpu

HEAD = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0"
EOF_BLOCK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def frame(payload, data):
    return HEAD + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def bgzf(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, chunk=0xff00, flush_every=0):
    """BGZF blocks of `data` made with zlib (raw deflate); flush_every > 0: several DEFLATE blocks inside one BGZF block."""
    data = bytes(data)
    out = []
    for p in range(0, len(data), chunk):
        piece = data[p:p + chunk]
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        if flush_every:
            payload = b"".join(c.compress(piece[q:q + flush_every]) + c.flush(zlib.Z_FULL_FLUSH) for q in range(0, len(piece), flush_every)) + c.flush()
        else:
            payload = c.compress(piece) + c.flush()
        assert len(payload) + 26 <= 65536
        out.append(frame(payload, piece))
    return b"".join(out)


def _ctx():
    idx = lib.Index({"refnames": ["chr1"], "transcripts": [{"id": "t", "ref_id": 0, "strand": "+", "exons": [[10, 50]]}]}, device=0)
    return idx, lib.Context(idx)


def inflate_on_device(ctx, raw):
    raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    blocks, consumed, total = lib.bgzf_scan(raw)
    assert consumed == raw.size
    src = torch.from_numpy(raw.copy()).to("cuda:0")
    out = ctx.bgzf_inflate_device(src, blocks)
    assert out.numel() == total
    return out.cpu().numpy().tobytes(), blocks


def payloads():
    rng = np.random.RandomState(17)
    ann = synth.Annotation("S")
    recs = ann.reads(3000, "pe", with_records=1)
    stream, _, _ = synth.Annotation.frame_records(recs)
    yield "BAM records", stream.tobytes()
    yield "one byte", b"\x7f"
    yield "zeros (distance 1 runs of 258)", bytes(200_000)
    yield "random (stored or nearly)", rng.randint(0, 256, size=150_001).astype(np.uint8).tobytes()
    yield "text", b"@read%07d\tACGTACGTTTGACCA\t+\tIIIIHHHGGFF###\n" * 9000
    yield "far repeats (distances up to 32 KiB)", np.tile(rng.randint(0, 256, size=30000).astype(np.uint8), 6).tobytes()
    yield "repeats at the window's edge", np.tile(rng.randint(0, 256, size=6900).astype(np.uint8), 12).tobytes() + np.tile(rng.randint(0, 256, size=7000).astype(np.uint8), 12).tobytes()
    yield "short periods", b"".join(bytes(rng.randint(0, 256, size=p).astype(np.uint8)) * (2000 // p) for p in (1, 2, 3, 5, 7, 63, 64, 65, 257, 258, 259))
    yield "skewed (codes longer than ten bits)", np.concatenate([np.full(50000, 7, dtype=np.uint8), np.repeat(np.arange(256, dtype=np.uint8), rng.randint(1, 4, size=256))]).tobytes()
    # the lanes' token decode: code lengths 2 .. 15 side by side, many short matches between literals (near and far back),
    # one- and two-bit codes (dozens of tokens in one 64-bit buffer, overlapping matches among them), all of it mixed
    geo = np.minimum(rng.geometric(0.06, size=260_000) - 1, 255).astype(np.uint8)
    yield "geometric symbols (codes of 2 to 15 bits)", geo.tobytes()
    four = rng.randint(0, 4, size=300_000).astype(np.uint8) * 37 + 11
    yield "four letters (short matches everywhere)", four.tobytes()
    two = (rng.rand(200_000) < 0.9).astype(np.uint8) * 3 + 64
    yield "two letters, one of them rare", two.tobytes()
    mix = []
    for k in range(120):
        n = int(rng.randint(200, 9000))
        kind = k % 5
        if kind == 0: mix.append(np.minimum(rng.geometric(0.03 + 0.2 * rng.rand(), size=n) - 1, 255).astype(np.uint8))
        elif kind == 1: mix.append((rng.randint(0, int(rng.randint(2, 9)), size=n) * 29 + 3).astype(np.uint8))
        elif kind == 2: mix.append(np.tile(rng.randint(0, 256, size=int(rng.randint(1, 40))).astype(np.uint8), n // 8 + 1)[:n])
        elif kind == 3: mix.append(rng.randint(0, 256, size=n).astype(np.uint8))
        else: mix.append(np.concatenate(mix[-3:])[:n][::-1].copy() if len(mix) >= 3 else np.zeros(n, dtype=np.uint8))
    yield "a mixture of all of these", np.concatenate(mix).tobytes()
    yield "exactly one block", rng.randint(65, 70, size=0xff00).astype(np.uint8).tobytes()
    yield "one block + 1", rng.randint(65, 70, size=0xff00 + 1).astype(np.uint8).tobytes()


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_zlib_made_blocks_inflate_to_their_input(level, strategy):
    idx, ctx = _ctx()
    for name, data in payloads():
        got, blocks = inflate_on_device(ctx, bgzf(data, level, strategy) + EOF_BLOCK)
        assert got == data, (name, level, strategy)
        assert len(blocks) == (len(data) + 0xff00 - 1) // 0xff00
    ctx.close(); idx.close()


def test_several_deflate_blocks_per_bgzf_block_and_small_blocks():
    idx, ctx = _ctx()
    rng = np.random.RandomState(5)
    data = (b"GATTACA" * 3000 + rng.randint(0, 256, size=20000).astype(np.uint8).tobytes()) * 3
    for flush_every, level in ((1000, 6), (4096, 1), (300, 0), (7, 6)):
        chunk = 4000 if flush_every < 300 else 40000 if level == 0 else 0xff00     # (a flush costs five bytes, a stored block five more: keep the framed block under 64 KiB)
        got, _ = inflate_on_device(ctx, bgzf(data, level, chunk=chunk, flush_every=flush_every) + EOF_BLOCK)
        assert got == data, (flush_every, level)
    # many tiny BGZF blocks, empty blocks in between (stepped over by the scan)
    pieces = [data[p:p + 37] for p in range(0, 5000, 37)]
    raw = b"".join(bgzf(piece, 6) + (EOF_BLOCK if k % 5 == 0 else b"") for k, piece in enumerate(pieces))
    got, blocks = inflate_on_device(ctx, raw)
    assert got == b"".join(pieces) and len(blocks) == len(pieces)
    ctx.close(); idx.close()


def test_the_librarys_own_writers_read_back():
    """Host writer (libdeflate or zlib behind br_bgzf_write_file's framing) and the device deflate kernel: 3 MB of projected
    BAM records through each, then through k_inflate."""
    import ctypes as C
    import os
    import tempfile
    idx, ctx = _ctx()
    ann = synth.Annotation("S")
    recs = ann.reads(12000, "pe", with_records=1)
    stream, _, _ = synth.Annotation.frame_records(recs)
    data = stream.tobytes()
    L = lib.lib()
    L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
    with tempfile.TemporaryDirectory() as tmp:
        for level in (1, 6):
            path = os.path.join(tmp, "x%d.bgzf" % level)
            assert L.br_bgzf_write_file(path.encode(), stream.ctypes.data, stream.size, 4, level) == 0
            got, _ = inflate_on_device(ctx, open(path, "rb").read())
            assert got == data
    z = ctx.bgzf_deflate_device(torch.from_numpy(stream).to("cuda:0"))
    raw = z.cpu().numpy().copy()
    blocks, consumed, total = lib.bgzf_scan(raw)
    assert consumed == raw.size and total == len(data)
    out = ctx.bgzf_inflate_device(torch.from_numpy(raw).to("cuda:0"), blocks)
    assert out.cpu().numpy().tobytes() == data
    ctx.close(); idx.close()


def test_corrupt_blocks_are_refused():
    idx, ctx = _ctx()
    rng = np.random.RandomState(3)
    data = (b"ACGT" * 5000 + rng.randint(0, 256, size=9000).astype(np.uint8).tobytes()) * 2
    raw = bytearray(bgzf(data, 6))
    blocks, _, _ = lib.bgzf_scan(np.frombuffer(bytes(raw), dtype=np.uint8))
    good = ctx.bgzf_inflate_device(torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to("cuda:0"), blocks)
    assert good.cpu().numpy().tobytes() == data
    cases = []
    b0 = int(blocks[0]["src_off"])
    for where in (b0 + 1, b0 + 40, b0 + int(blocks[0]["clen"]) - 2):     # header bits, a symbol in the middle, near the end
        bad = bytearray(raw); bad[where] ^= 0x5a; cases.append(bytes(bad))
    bad = bytearray(raw); t = b0 + int(blocks[0]["clen"]); bad[t] ^= 1; cases.append(bytes(bad))                      # CRC32
    for c in cases:
        arr = np.frombuffer(c, dtype=np.uint8)
        bl, _, _ = lib.bgzf_scan(arr)
        with pytest.raises(lib.BrambleError):
            ctx.bgzf_inflate_device(torch.from_numpy(arr.copy()).to("cuda:0"), bl)
    # a wrong ISIZE in the table
    bl = blocks.copy(); bl["ulen"][0] -= 1
    with pytest.raises(lib.BrambleError):
        ctx.bgzf_inflate_device(torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to("cuda:0"), bl)
    # the context still works afterwards
    again = ctx.bgzf_inflate_device(torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to("cuda:0"), blocks)
    assert again.cpu().numpy().tobytes() == data
    ctx.close(); idx.close()


def test_random_streams_of_every_make():
    """The lane decoder against zlib on a few hundred random streams: random mixtures of symbol statistics (code lengths 1 to
    15, literal runs, short and long matches near and far, overlapping runs), random zlib parameters (level, strategy, memory
    level, window) and random BGZF block sizes -- so that token boundaries fall on every position of the 64-bit decode
    window, of the input ring (its wrap, its 512-byte refills) and of the 512-byte output pieces."""
    rng = np.random.RandomState(int(os.environ.get("INFLATE_FUZZ_SEED", 20261005)))   # (soak runs: another seed per run)
    idx, ctx = _ctx()

    def piece(n):
        kind = rng.randint(0, 8)
        if kind == 0: return np.minimum(rng.geometric(0.02 + 0.5 * rng.rand(), size=n) - 1, 255).astype(np.uint8)
        if kind == 1: return (rng.randint(0, rng.randint(2, 20), size=n) * rng.randint(1, 13) + rng.randint(0, 40)).astype(np.uint8)
        if kind == 2: return np.tile(rng.randint(0, 256, size=rng.randint(1, 300)).astype(np.uint8), n // 2 + 1)[:n]
        if kind == 3: return rng.randint(0, 256, size=n).astype(np.uint8)
        if kind == 4: return np.full(n, rng.randint(0, 256), dtype=np.uint8)
        if kind == 5: return np.where(rng.rand(n) < 0.02 + 0.3 * rng.rand(), rng.randint(0, 256, size=n), rng.randint(0, 256)).astype(np.uint8)
        if kind == 6:   # a sawtooth with noise: matches at one distance, broken up
            p = rng.randint(3, 2000)
            base = np.tile(rng.randint(0, 256, size=p).astype(np.uint8), n // p + 2)[:n].copy()
            hits = rng.rand(n) < 0.01 * rng.randint(1, 20)
            base[hits] = rng.randint(0, 256, size=int(hits.sum()))
            return base
        return np.frombuffer((b"@r%06d/1\tchr%d\t%d\t60\t100M\n" % (rng.randint(0, 10**6), rng.randint(1, 23), rng.randint(0, 10**8))) * (n // 30 + 1), dtype=np.uint8)[:n]

    made = []
    total = 0
    for k in range(240):
        parts, size = [], int(rng.randint(1, 400_000) if k % 7 else rng.randint(1, 300))
        while sum(len(p) for p in parts) < size:
            parts.append(piece(int(rng.randint(1, 40_000))))
        if len(parts) > 2 and rng.rand() < 0.5:   # something from far back, again
            parts.append(np.concatenate(parts)[:int(rng.randint(1, 70_000))])
        data = np.concatenate(parts)[:size].tobytes()
        level = int(rng.choice([0, 1, 2, 4, 6, 9]))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        mem, wbits = int(rng.randint(1, 10)), int(rng.randint(9, 16))
        chunk = int(rng.choice([0xff00, 0xff00, 40_000, 9_000, 513, 65_280]))
        out = []
        for p in range(0, len(data), chunk):
            seg = data[p:p + chunk]
            c = zlib.compressobj(level, zlib.DEFLATED, -wbits, mem, strategy)
            flush_at = int(rng.randint(1, len(seg) + 1)) if rng.rand() < 0.3 else len(seg)
            payload = c.compress(seg[:flush_at]) + (c.flush(zlib.Z_SYNC_FLUSH) if flush_at < len(seg) else b"") + c.compress(seg[flush_at:]) + c.flush()
            if len(payload) + 26 > 65536:          # (does not fit a BGZF block: stored halves)
                half = len(seg) // 2
                for s2 in (seg[:half], seg[half:]):
                    c2 = zlib.compressobj(0, zlib.DEFLATED, -15)
                    out.append(frame(c2.compress(s2) + c2.flush(), s2))
                continue
            out.append(frame(payload, seg))
        made.append((b"".join(out), data, (k, level, strategy, mem, wbits, chunk, size)))
        total += len(data)
    # all of them in one call (one block table), and a sample one by one
    raw = b"".join(m[0] for m in made)
    got, blocks = inflate_on_device(ctx, raw)
    want = b"".join(m[1] for m in made)
    if got != want:
        at = 0
        for r, d, what in made:
            assert got[at:at + len(d)] == d, what
            at += len(d)
    for r, d, what in made[::17]:
        g, _ = inflate_on_device(ctx, r)
        assert g == d, what
    assert total > 20_000_000
