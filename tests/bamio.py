"""Test infrastructure: a minimal BAM / BGZF reader and writer in pure Python (zlib), independent of the
product's C++ BGZF layer, plus a GTF writer for synthetic annotations."""
import gzip
import struct
import zlib

import numpy as np


def bgzf_compress(data, block=0xff00, level=6):
    out = bytearray()
    data = bytes(data)
    for a in range(0, max(len(data), 1), block):
        chunk = data[a:a + block]
        if not chunk and a > 0:
            break
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = 18 + len(comp) + 8 - 1
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def write_bam(path, header_text, refs, record_stream, block=0xff00, level=6):
    """refs: [(name, length)]; record_stream: bytes of [block_size][record]..."""
    text = header_text.encode()
    h = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
    for name, ln in refs:
        nm = name.encode() + b"\0"
        h += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
    with open(path, "wb") as f:
        f.write(bgzf_compress(bytes(h) + bytes(record_stream), block=block, level=level))


def read_bam(path):
    """-> (header_text, [(name, length)], record stream as numpy uint8)"""
    with gzip.open(path, "rb") as f:   # BGZF is a series of gzip members
        data = f.read()
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].rstrip(b"\0").decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]
        p += 4
        name = data[p:p + l_name - 1].decode()
        p += l_name
        refs.append((name, struct.unpack_from("<i", data, p)[0]))
        p += 4
    return text, refs, np.frombuffer(data[p:], dtype=np.uint8)


def bgzf_block_sizes(path):
    """Sizes of the BGZF blocks of a file (framing check)."""
    raw = open(path, "rb").read()
    p, out = 0, []
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        out.append(bsize)
        p += bsize
    return out


def write_gtf(path, ann, order=None, with_transcript_lines=True, gz=False):
    """ann: dict with refnames + transcripts [{id, ref_id, strand, exons [[s,e) 1-based half-open]}]."""
    txs = ann["transcripts"]
    idx = list(range(len(txs))) if order is None else list(order)
    lines = ["# synthetic annotation"]
    for t in idx:
        tx = txs[t]
        ref = ann["refnames"][tx["ref_id"]]
        ex = sorted(tx["exons"])
        attr = 'gene_id "g_%s"; transcript_id "%s";' % (tx["id"], tx["id"])
        if with_transcript_lines:
            lines.append("\t".join([ref, "synth", "transcript", str(ex[0][0]), str(ex[-1][1] - 1), ".", tx["strand"], ".", attr]))
        for s, e in ex:
            lines.append("\t".join([ref, "synth", "exon", str(s), str(e - 1), ".", tx["strand"], ".", attr]))
    body = ("\n".join(lines) + "\n").encode()
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(body)
    else:
        with open(path, "wb") as f:
            f.write(body)


def guide_order(ann):
    """Independent restatement of gclib's gfo_cmpByLoc (gclib/gff.cpp:75-90) for transcripts of level 0:
    reference name (byte order), start, end, id."""
    txs = ann["transcripts"]

    def key(t):
        ex = sorted(txs[t]["exons"])
        return (ann["refnames"][txs[t]["ref_id"]].encode(), ex[0][0], ex[-1][1] - 1, txs[t]["id"].encode())
    return sorted(range(len(txs)), key=key)


def bam_record(name, ref_id, pos0, cigar, l_seq, flag=0, aux=b"", mapq=30, mate=(-1, -1, 0), spill=None, seq=None, qual=None):
    """One BAM record (from refID on, without block_size), assembled from the SAM/BAM specification, independent of the
    product and of the oracle.  cigar: BAM-packed uint32 ops.  A CIGAR of more than 65535 ops is stored the way the
    specification (4.2.2) and htslib's writer do: <l_seq>S<ref_len>N in the CIGAR field, the real ops in a CG:B,I tag
    behind the other tags (spill=True forces that form for a short CIGAR, spill=False forbids it)."""
    cigar = [int(w) for w in cigar]
    if spill is None:
        spill = len(cigar) > 65535
    nm = name + b"\0"
    field = cigar
    tail = b""
    if spill:
        reflen = sum(w >> 4 for w in cigar if (w & 0xF) in (0, 2, 3, 7, 8))
        field = [(l_seq << 4) | 4, (reflen << 4) | 3]
        tail = b"CGBI" + struct.pack("<I", len(cigar)) + np.asarray(cigar, dtype="<u4").tobytes()
    body = struct.pack("<iiBBHHHiiii", ref_id, pos0, len(nm), mapq, 4680, len(field), flag, l_seq, mate[0], mate[1], mate[2]) + nm
    body += np.asarray(field, dtype="<u4").tobytes()
    body += seq if seq is not None else bytes([0x12] * ((l_seq + 1) // 2))      # A C A C ...
    body += qual if qual is not None else bytes([(30 + k) % 40 for k in range(l_seq)])
    return body + aux + tail


def frame(records):
    """[block_size][record]... as numpy uint8 (the uncompressed alignment section)."""
    return np.frombuffer(b"".join(struct.pack("<I", len(r)) + r for r in records), dtype=np.uint8)


def split_stream(stream):
    """-> list of record bytes (without block_size) of an uncompressed alignment section"""
    data = bytes(stream)
    out, p = [], 0
    while p < len(data):
        n = struct.unpack_from("<I", data, p)[0]
        out.append(data[p + 4:p + 4 + n])
        p += 4 + n
    return out


def record_fields(rec):
    """the pieces of one BAM record: dict with n_cigar_field, cigar (the CIGAR field's words), aux (bytes), l_seq ..."""
    ref_id, pos, l_qname, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
    p = 32
    name = rec[p:p + l_qname - 1]
    p += l_qname
    cigar = list(np.frombuffer(rec[p:p + 4 * n_cig], dtype="<u4"))
    p += 4 * n_cig
    seq = rec[p:p + (l_seq + 1) // 2]
    p += (l_seq + 1) // 2
    qual = rec[p:p + l_seq]
    p += l_seq
    return {"ref_id": ref_id, "pos": pos, "name": name, "mapq": mapq, "bin": bin_, "flag": flag, "l_seq": l_seq, "mtid": mtid,
            "mpos": mpos, "tlen": tlen, "n_cigar_field": n_cig, "cigar": cigar, "seq": seq, "qual": qual, "aux": rec[p:]}
