"""BGZF deflate on the device (k_deflate_fixed / k_bgzf_compact): any inflater must reproduce the input, the block
framing must be valid BGZF (header, BSIZE, CRC32, ISIZE).  The compressed bytes themselves are not compared with
anything: they are not part of the parity contract."""
import gzip
import io
import struct
import zlib

import numpy as np
import pytest

from bramble_amd import lib, synth

pytestmark = pytest.mark.gpu


def _ctx():
    idx = lib.Index({"refnames": ["chr1"], "transcripts": [{"id": "t", "ref_id": 0, "strand": "+", "exons": [[10, 50]]}]}, device=0)
    return idx, lib.Context(idx)


def inflate_blocks(raw):
    """Walk BGZF blocks, inflate each payload with zlib, verify CRC32 / ISIZE; returns (data, n_blocks, max_block)."""
    raw = bytes(raw)
    p, out, nb, mx = 0, [], 0, 0
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04" and raw[p + 12:p + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        payload = raw[p + 18:p + bsize - 8]
        crc, isize = struct.unpack_from("<II", raw, p + bsize - 8)
        d = zlib.decompressobj(-15)
        data = d.decompress(payload)
        assert d.eof and d.unused_data == b""
        assert len(data) == isize and (zlib.crc32(data) & 0xffffffff) == crc
        out.append(data)
        nb += 1
        mx = max(mx, bsize)
        p += bsize
    return b"".join(out), nb, mx


def payloads():
    rng = np.random.RandomState(9)
    text = (b"@read%07d\tACGTACGTTTGACCA\t+\tIIIIHHHGGFF###\n" * 40000)
    yield "one byte", np.frombuffer(b"\x7f", dtype=np.uint8)
    yield "one symbol only", np.full(5000, 65, dtype=np.uint8)
    yield "two symbols", np.frombuffer(b"ab" * 3 + b"a" * 50 + b"b", dtype=np.uint8)
    yield "skewed (long codes for rare bytes)", np.concatenate([np.full(56000, 7, dtype=np.uint8), np.arange(256, dtype=np.uint8)])
    yield "three bytes", np.frombuffer(b"abc", dtype=np.uint8)
    yield "zeros", np.zeros(200_000, dtype=np.uint8)
    yield "random (incompressible, bytes >= 144 take 9 bits)", rng.randint(0, 256, size=300_001).astype(np.uint8)
    yield "high bytes", rng.randint(144, 256, size=57344 * 2).astype(np.uint8)
    yield "text", np.frombuffer(text, dtype=np.uint8)
    yield "block - 1", rng.randint(65, 70, size=57343).astype(np.uint8)
    yield "block", rng.randint(65, 70, size=57344).astype(np.uint8)
    yield "block + 1", rng.randint(65, 70, size=57345).astype(np.uint8)
    yield "long runs and far repeats", np.concatenate([np.tile(rng.randint(0, 256, size=40000).astype(np.uint8), 5),
                                                       np.repeat(rng.randint(0, 256, size=300).astype(np.uint8), 700)])


@pytest.mark.parametrize("dynamic", [1, 0])
def test_device_deflate_round_trips(dynamic):
    """dynamic = 1: k_deflate_dynamic (per-block Huffman codes, the default); 0: k_deflate_fixed."""
    import torch
    idx, ctx = _ctx()
    ctx.set_param("deflate_dynamic", dynamic)
    for label, data in payloads():
        src = torch.from_numpy(data.copy()).to("cuda:0")
        z = ctx.bgzf_deflate_device(src, 0).cpu().numpy().tobytes()
        got, nb, mx = inflate_blocks(z)
        assert got == data.tobytes(), label
        assert nb == (len(data) + 57343) // 57344 and mx <= 65536, label
        assert gzip.GzipFile(fileobj=io.BytesIO(z)).read() == data.tobytes(), label   # a second, independent reader
    ctx.close()
    idx.close()


def test_device_deflate_compresses_projected_stream():
    ann = synth.Annotation("G", n_genes=1500, n_refs=3)
    b = ann.reads(8000, "pe", with_records=1)
    stream, roff, rlen = synth.Annotation.frame_records(b)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    plain, c1 = ctx.project_bam_bundle(cfg, stream, roff, rlen, np.arange(3, dtype=np.int32))
    packed, c2 = ctx.project_bam_bundle(cfg, stream, roff, rlen, np.arange(3, dtype=np.int32), bgzf_on_device=True)
    assert c1 == c2
    got, nb, mx = inflate_blocks(packed.tobytes())
    assert got == plain.tobytes()
    assert len(packed) < 0.5 * len(plain)       # the stream repeats every read once per transcript
    ctx.close()
    idx.close()
