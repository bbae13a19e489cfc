"""Seeded synthetic inputs: GENCODE-shaped annotation and
name-collated alignment batches.  ctypes binding of libbramble_synth.so
(bramble_amd/csrc/synth.cpp).  Input tooling for tests and bench.py -- not part
of the projection path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbramble_synth.so")
_LIB = None

SEED = 0xB4A3B1E


class ReadParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_templates", C.c_int64), ("mode", C.c_int32), ("read_len", C.c_int32),
                ("frag_mean", C.c_double), ("frag_sd", C.c_double), ("p_softclip", C.c_double),
                ("p_indel", C.c_double), ("p_junc_shift", C.c_double), ("p_intergenic", C.c_double),
                ("p_multimap", C.c_double), ("long_median", C.c_double), ("long_sigma", C.c_double),
                ("wobble", C.c_int32), ("p_wobble", C.c_double), ("p_skip_small", C.c_double),
                ("p_novel_small", C.c_double), ("p_clip", C.c_double), ("max_clip", C.c_int32),
                ("with_seq", C.c_int32), ("xs_tag", C.c_int32), ("with_records", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libbramble_synth.so is not built (run __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.synth_annotation_new.restype = C.c_void_p
        L.synth_annotation_new.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                           C.c_int32]
        L.synth_reads_new.restype = C.c_void_p
        L.synth_reads_new.argtypes = [C.c_void_p, C.POINTER(ReadParams)]
        for name in ("synth_annotation_free", "synth_reads_free"):
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("synth_annotation_n_tx", "synth_annotation_n_exons", "synth_reads_n"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("tx_ref", "tx_strand", "tx_gene", "tx_exon_off", "ex_start", "ex_end", "ref_len"):
            f = getattr(L, "synth_annotation_" + name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p]
        L.synth_annotation_ref_seq.restype = C.c_void_p
        L.synth_annotation_ref_seq.argtypes = [C.c_void_p, C.c_int32]
        for name in ("ref_id", "ref_start", "mate_ref_id", "mate_start", "l_qseq", "flags", "xs", "ts", "cigar_off",
                     "cigar", "name_off", "names", "src_tx", "seq_off", "seqs", "rec_off", "rec_blob"):
            f = getattr(L, "synth_reads_" + name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _copy(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Annotation:
    """size 'S': 1 reference, ~100 transcripts (BASELINE config 1); 'G': GENCODE-shaped."""

    def __init__(self, size="S", seed=SEED, with_genome=False, n_genes=None, n_refs=None):
        L = lib()
        if size == "S":
            nr, ng, iso, mex, cap = 1, 25, 4.0, 5.0, 12
        else:
            nr, ng, iso, mex, cap = 25, 60000, 4.2, 6.6, 400
        if n_genes is not None:
            ng = n_genes
        if n_refs is not None:
            nr = n_refs
        self.h = L.synth_annotation_new(seed, nr, ng, iso, mex, cap, 1 if with_genome else 0)
        self.n_refs = nr
        n = L.synth_annotation_n_tx(self.h)
        ne = L.synth_annotation_n_exons(self.h)
        self.flat = {
            "n_refs": nr,
            "tx_ref": _copy(L.synth_annotation_tx_ref(self.h), n, np.int32),
            "tx_strand": _copy(L.synth_annotation_tx_strand(self.h), n, np.int8),
            "tx_gene": _copy(L.synth_annotation_tx_gene(self.h), n, np.uint32),
            "tx_exon_off": _copy(L.synth_annotation_tx_exon_off(self.h), n + 1, np.uint64),
            "ex_start": _copy(L.synth_annotation_ex_start(self.h), ne, np.uint32),
            "ex_end": _copy(L.synth_annotation_ex_end(self.h), ne, np.uint32),
            "ref_len": _copy(L.synth_annotation_ref_len(self.h), nr, np.uint32),
            "ref_seqs": None,
        }
        if with_genome:
            self.flat["ref_seqs"] = [bytes(_copy(L.synth_annotation_ref_seq(self.h, r), int(self.flat["ref_len"][r]),
                                                 np.uint8)) for r in range(nr)]
        self.n_tx = n
        self.n_exons = ne

    def as_dict(self):
        """annotation dict {refnames, transcripts:[{id, ref_id, strand, exons}]} (small sizes only)."""
        f = self.flat
        txs = []
        for t in range(self.n_tx):
            a, b = int(f["tx_exon_off"][t]), int(f["tx_exon_off"][t + 1])
            txs.append({"id": "tx%d" % t, "ref_id": int(f["tx_ref"][t]), "strand": chr(f["tx_strand"][t]),
                        "exons": [[int(s), int(e)] for s, e in zip(f["ex_start"][a:b], f["ex_end"][a:b])]})
        d = {"refnames": ["ref%d" % r for r in range(self.n_refs)], "transcripts": txs}
        if f["ref_seqs"] is not None:
            d["ref_seqs"] = {r: s for r, s in enumerate(f["ref_seqs"])}
        return d

    def reads(self, n_templates, mode="pe", seed=None, xs_tag=False, **kw):
        """mode: 'se' (1x100), 'pe' (2x100), 'hifi', 'ont'.  Returns a batch dict (bramble_amd.batch layout)
        plus 'src_tx'."""
        L = lib()
        p = ReadParams()
        p.seed = (SEED ^ 0x51ED) if seed is None else seed
        p.n_templates = n_templates
        p.mode = {"se": 0, "pe": 1, "hifi": 2, "ont": 2}[mode]
        p.read_len, p.frag_mean, p.frag_sd = 100, 300.0, 50.0
        p.p_softclip, p.p_indel, p.p_junc_shift, p.p_intergenic, p.p_multimap = 0.05, 0.01, 0.02, 0.03, 0.03
        if mode == "ont":
            p.long_median, p.long_sigma, p.wobble, p.p_wobble = 900.0, 0.6, 40, 0.10
            p.p_skip_small, p.p_novel_small, p.p_clip, p.max_clip = 0.08, 0.02, 0.5, 300
        else:
            p.long_median, p.long_sigma, p.wobble, p.p_wobble = 2000.0, 0.5, 10, 0.10
            p.p_skip_small, p.p_novel_small, p.p_clip, p.max_clip = 0.08, 0.02, 0.2, 30
        p.xs_tag = 1 if xs_tag else 0
        for k, v in kw.items():
            setattr(p, k, v)
        h = L.synth_reads_new(self.h, C.byref(p))
        try:
            n = L.synth_reads_n(h)
            b = {"n_aln": n}
            for name, dt in (("ref_id", np.int32), ("ref_start", np.int32), ("mate_ref_id", np.int32),
                             ("mate_start", np.int32), ("l_qseq", np.int32), ("flags", np.uint16), ("xs", np.int8),
                             ("ts", np.int8), ("src_tx", np.uint32)):
                b[name] = _copy(getattr(L, "synth_reads_" + name)(h), n, dt)
            b["cigar_off"] = _copy(L.synth_reads_cigar_off(h), n + 1, np.uint64)
            b["cigar"] = _copy(L.synth_reads_cigar(h), int(b["cigar_off"][-1]) if n else 0, np.uint32)
            b["name_off"] = _copy(L.synth_reads_name_off(h), n + 1, np.uint64)
            b["names"] = _copy(L.synth_reads_names(h), int(b["name_off"][-1]) if n else 0, np.uint8)
            b["seq_off"] = None
            b["seqs"] = None
            if p.with_records:
                b["rec_off"] = _copy(L.synth_reads_rec_off(h), n + 1, np.uint64)
                b["rec_blob"] = _copy(L.synth_reads_rec_blob(h), int(b["rec_off"][-1]) if n else 0, np.uint8)
            if p.with_seq:
                b["seq_off"] = _copy(L.synth_reads_seq_off(h), n + 1, np.uint64)
                b["seqs"] = _copy(L.synth_reads_seqs(h), int(b["seq_off"][-1]) if n else 0, np.uint8)
        finally:
            L.synth_reads_free(h)
        return b

    @staticmethod
    def frame_records(b):
        """Records of a batch made with with_records=1 -> (stream uint8[] of [block_size][record]..., rec_off uint64[n]
        of each record's refID word, rec_len uint32[n]): an uncompressed BAM alignment section."""
        off = np.ascontiguousarray(b["rec_off"], dtype=np.uint64)
        blob = np.ascontiguousarray(b["rec_blob"], dtype=np.uint8)
        n = len(off) - 1
        out = np.empty(int(off[-1]) + 4 * n, dtype=np.uint8)
        L = lib()
        L.synth_frame_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.synth_frame_records.restype = None
        L.synth_frame_records(blob.ctypes.data, off.ctypes.data, n, out.ctypes.data)
        rec_len = np.diff(off).astype(np.uint32)
        rec_off = off[:-1] + np.uint64(4) * np.arange(1, n + 1, dtype=np.uint64)
        return out, rec_off, rec_len

    def __del__(self):
        try:
            lib().synth_annotation_free(self.h)
        except Exception:
            pass
