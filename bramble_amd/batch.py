"""Struct-of-arrays alignment batches: the flat form of bramble-rs's
`GenomicAlignment` (bramble-rs/src/api.rs:73-126) that the C-ABI batch entry
points take (include/bramble_amd.h, `br_batch`).

A batch is a dict of numpy arrays:

  n_aln        int
  ref_id       int32[n]    0-based reference index (-1: skip the alignment)
  ref_start    int32[n]    1-based SAM POS
  flags        uint16[n]   SAM flag bits (0x1 paired, 0x10 reverse, 0x40/0x80 read1/2)
  xs, ts       int8[n]     first char of the XS / ts tag, 0 when absent
  cigar_off    uint64[n+1] offsets into `cigar`
  cigar        uint32[]    BAM-packed ops (len<<4|op)
  mate_ref_id  int32[n]    -1 when absent
  mate_start   int32[n]    1-based mate POS, 0 when absent
  name_off     uint64[n+1] offsets into `names`
  names        uint8[]     query names, concatenated (name-collated order)
  seq_off      uint64[n+1] or None; seqs uint8[] ASCII bases (only for -S)
  l_qseq       int32[n]    read length (BAM l_qseq)
"""
import numpy as np

CIGAR_ALPHABET = "MIDNSHP=XB,./;"

F_PAIRED, F_REVERSE, F_READ1, F_READ2, F_MUNMAP = 0x1, 0x10, 0x40, 0x80, 0x8


def parse_cigar(text):
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | CIGAR_ALPHABET.index(ch))
            num = ""
    return np.array(out, dtype=np.uint32)


def format_cigar(words):
    return "".join("%d%s" % (int(w) >> 4, CIGAR_ALPHABET[int(w) & 0xF]) for w in words)


def make_batch(records):
    """records: iterable of dicts with keys name, ref_id, ref_start, cigar (text or
    uint32 array) and optional flags, xs, ts, mate_ref_id, mate_start, seq, read_len."""
    records = list(records)
    n = len(records)
    b = {"n_aln": n}
    b["ref_id"] = np.array([r["ref_id"] for r in records], dtype=np.int32)
    b["ref_start"] = np.array([r["ref_start"] for r in records], dtype=np.int32)
    b["flags"] = np.array([r.get("flags", 0) for r in records], dtype=np.uint16)

    def tag(v):
        if v is None or v == 0:
            return 0
        return ord(v) if isinstance(v, str) else int(v)

    b["xs"] = np.array([tag(r.get("xs")) for r in records], dtype=np.int8)
    b["ts"] = np.array([tag(r.get("ts")) for r in records], dtype=np.int8)
    cigs = [parse_cigar(r["cigar"]) if isinstance(r["cigar"], str) else np.asarray(r["cigar"], dtype=np.uint32)
            for r in records]
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum([len(c) for c in cigs])
    b["cigar_off"] = off
    b["cigar"] = np.concatenate(cigs).astype(np.uint32) if n and int(off[-1]) else np.zeros(0, dtype=np.uint32)
    b["mate_ref_id"] = np.array([r.get("mate_ref_id", -1) for r in records], dtype=np.int32)
    b["mate_start"] = np.array([r.get("mate_start", 0) for r in records], dtype=np.int32)
    names = [r["name"].encode() for r in records]
    noff = np.zeros(n + 1, dtype=np.uint64)
    if n:
        noff[1:] = np.cumsum([len(x) for x in names])
    b["name_off"] = noff
    b["names"] = np.frombuffer(b"".join(names), dtype=np.uint8).copy()
    if any(r.get("seq") for r in records):
        seqs = [(r.get("seq") or "").encode() for r in records]
        soff = np.zeros(n + 1, dtype=np.uint64)
        soff[1:] = np.cumsum([len(x) for x in seqs])
        b["seq_off"] = soff
        b["seqs"] = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    else:
        b["seq_off"] = None
        b["seqs"] = None
    lq = []
    for r, c in zip(records, cigs):
        if "read_len" in r:
            lq.append(r["read_len"])
        elif r.get("seq"):
            lq.append(len(r["seq"]))
        else:  # query-consuming ops M,I,S,=,X
            lq.append(int(sum(int(w) >> 4 for w in c if (int(w) & 0xF) in (0, 1, 4, 7, 8))))
    b["l_qseq"] = np.array(lq, dtype=np.int32)
    return b


def annotation_from_gtf_like(refnames, transcripts):
    """transcripts: [{id, seqname, strand, exons_gtf_inclusive | exons_half_open}] ->
    annotation dict {refnames, transcripts:[{id, ref_id, strand, exons (1-based half-open)}]}.
    GTF ends are inclusive; the index uses [start, end+1) like the reference
    (src/bramble.cpp:164-165, bramble-rs/src/annotation.rs:52-64)."""
    rid = {n: i for i, n in enumerate(refnames)}
    out = []
    for t in transcripts:
        if "exons_half_open" in t:
            ex = [[s, e] for s, e in t["exons_half_open"]]
        else:
            ex = [[s, e + 1] for s, e in t["exons_gtf_inclusive"]]
        out.append({"id": t["id"], "ref_id": rid[t["seqname"]], "strand": t["strand"], "exons": ex})
    return {"refnames": list(refnames), "transcripts": out}
