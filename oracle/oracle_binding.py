"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (bramble_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CIGAR_ALPHABET = "MIDNSHP=XB,./;"


def parse_cigar(text):
    """'2S8M' -> uint32 array of BAM-packed ops (len<<4|op); ',' './' ';' = override ops 10..13."""
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


class OrcFlags(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("lr", "lr_hq", "strict", "use_fasta", "fr", "rf", "has_max_clip", "has_max_junc_ins",
                 "has_max_junc_gap", "has_sim_thr", "has_max_error_exon")] + \
               [(n, C.c_uint32) for n in ("max_clip", "max_junc_ins", "max_junc_gap", "max_error_exon")] + \
               [("sim_thr", C.c_float)]


def make_flags(**kw):
    f = OrcFlags()
    for k, v in kw.items():
        if k in ("max_clip", "max_junc_ins", "max_junc_gap", "max_error_exon", "sim_thr"):
            setattr(f, "has_" + k, 1)
        setattr(f, k, v)
    return f


_P = C.POINTER


class OrcBatch(C.Structure):
    _fields_ = [("n_aln", C.c_int64), ("ref_id", _P(C.c_int32)), ("ref_start", _P(C.c_int32)),
                ("flags", _P(C.c_uint16)), ("xs", _P(C.c_int8)), ("ts", _P(C.c_int8)),
                ("cigar_off", _P(C.c_uint64)), ("cigar", _P(C.c_uint32)),
                ("mate_ref_id", _P(C.c_int32)), ("mate_start", _P(C.c_int32)),
                ("name_off", _P(C.c_uint64)), ("names", C.c_char_p),
                ("seq_off", _P(C.c_uint64)), ("seqs", C.c_char_p), ("l_qseq", _P(C.c_int32))]


class OrcMatches(C.Structure):
    _fields_ = [("n_aln", C.c_int64), ("n_matches", C.c_int64), ("aln_off", _P(C.c_uint64)),
                ("tid", _P(C.c_uint32)), ("fwpos", _P(C.c_uint32)), ("rcpos", _P(C.c_uint32)),
                ("strand", _P(C.c_int8)), ("similarity_score", _P(C.c_double)),
                ("total_coverage", _P(C.c_double)), ("total_operations", _P(C.c_double)),
                ("junc_hits", _P(C.c_int32)), ("ref_consumed", _P(C.c_int32)), ("clip_score", _P(C.c_int32)),
                ("ideal_off", _P(C.c_uint64)), ("ideal", _P(C.c_uint32)),
                ("out_off", _P(C.c_uint64)), ("out", _P(C.c_uint32)),
                ("n_exons", _P(C.c_int32)), ("mate_idx", _P(C.c_int32))]


class OrcRows(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("input_index", _P(C.c_int32)), ("tid", _P(C.c_uint32)),
                ("pos", _P(C.c_uint32)), ("strand", _P(C.c_int8)), ("cigar_off", _P(C.c_uint64)),
                ("cigar", _P(C.c_uint32)), ("similarity_score", _P(C.c_double)),
                ("clip_score", _P(C.c_int32)), ("junc_hits", _P(C.c_int32)), ("ref_consumed", _P(C.c_int32)),
                ("nh", _P(C.c_uint32)), ("hi", _P(C.c_uint32)), ("mapq", _P(C.c_uint32)),
                ("primary", _P(C.c_uint8)), ("is_paired", _P(C.c_uint8)), ("same_transcript", _P(C.c_uint8)),
                ("is_first", _P(C.c_uint8)), ("mate_tid", _P(C.c_int32)), ("mate_pos", _P(C.c_int32)),
                ("isize", _P(C.c_int32)), ("group", _P(C.c_uint32)),
                ("total_complete", C.c_uint64), ("total_unique", C.c_uint64),
                ("dropped_reads", C.c_uint64), ("total_processed", C.c_uint64)]


def build():
    """Compile liboracle.so from the sources in oracle/ (g++ only)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_NATIVE = False


def use_native():
    """bench.py's cpu_baseline leg only: rebuild the oracle ON THIS MACHINE with -march=native (the CPU
    figure is a host-tuned build; -ffp-contract=off stays) into oracle/_native/ and load that copy from now on.  The
    parity tests keep the portable liboracle.so, which is built in the container and travels to the GPU box -- a
    -march=native object from another machine could fault there.  Returns True when the native copy is in use."""
    global _LIB, _NATIVE
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native"], timeout=600)
    except Exception:
        return False
    _LIB = None
    _NATIVE = True
    return True


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_native", "liboracle.so") if _NATIVE else os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_index_new.restype = C.c_void_p
        L.orc_index_add_transcript.restype = C.c_int64
        L.orc_index_add_transcript.argtypes = [C.c_void_p, C.c_int32, C.c_char, C.c_char_p, C.c_void_p,
                                               C.c_int32, C.c_char_p, C.c_int64]
        L.orc_index_finish.argtypes = [C.c_void_p]
        L.orc_index_num_transcripts.restype = C.c_int64
        L.orc_index_num_transcripts.argtypes = [C.c_void_p]
        L.orc_index_transcript_len.restype = C.c_uint32
        L.orc_index_transcript_len.argtypes = [C.c_void_p, C.c_int64]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_run.restype = C.c_void_p
        L.orc_run.argtypes = [C.c_void_p, _P(OrcFlags), _P(OrcBatch), C.c_int32, C.c_int32]
        L.orc_result_rows.restype = _P(OrcRows)
        L.orc_result_rows.argtypes = [C.c_void_p]
        L.orc_result_matches.restype = _P(OrcMatches)
        L.orc_result_matches.argtypes = [C.c_void_p]
        L.orc_result_seconds.restype = C.c_double
        L.orc_result_seconds.argtypes = [C.c_void_p]
        L.orc_result_free.argtypes = [C.c_void_p]
        L.orc_merge_cigar.restype = C.c_int32
        L.orc_merge_cigar.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_segments.restype = C.c_int32
        L.orc_segments.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.orc_resolve_config.restype = C.c_int32
        L.orc_resolve_config.argtypes = [_P(OrcFlags), C.c_void_p, _P(C.c_float)]
        L.orc_ksw_align.restype = C.c_int32
        L.orc_ksw_align.argtypes = [C.c_char_p, C.c_char_p, _P(C.c_int32), _P(C.c_int32), C.c_void_p, C.c_int32]
        L.orc_bam_encode.restype = C.c_int64
        L.orc_bam_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, _P(C.c_void_p)]
        L.orc_bam_parse.restype = C.c_void_p
        L.orc_bam_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.orc_parsed_batch.restype = _P(OrcBatch)
        L.orc_parsed_batch.argtypes = [C.c_void_p]
        L.orc_parsed_free.argtypes = [C.c_void_p]
        L.orc_free_buffer.argtypes = [C.c_void_p]
        L.orc_primary_pick.restype = C.c_uint32
        L.orc_primary_pick.argtypes = [C.c_char_p, C.c_int64, C.c_uint32]
        _LIB = L
    return _LIB


def _ptr(a, ty):
    return a.ctypes.data_as(_P(ty))


def _arr(p, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True)


class OracleIndex:
    def __init__(self, annotation):
        """annotation: dict with refnames, transcripts [{id, ref_id, strand, exons [[s,e) 1-based half-open]}],
        optional ref_seqs {ref_id: str}."""
        L = lib()
        self.h = L.orc_index_new()
        seqs = annotation.get("ref_seqs") or {}
        enc = {k: (v.encode() if isinstance(v, str) else v) for k, v in seqs.items()}   # once per reference, not per transcript
        for t in annotation["transcripts"]:
            ex = np.asarray(t["exons"], dtype=np.uint32).reshape(-1)
            sb = enc.get(t["ref_id"])
            L.orc_index_add_transcript(self.h, t["ref_id"], t["strand"].encode(), t["id"].encode(),
                                       ex.ctypes.data, len(ex) // 2, sb, len(sb) if sb is not None else 0)
        L.orc_index_finish(self.h)

    def num_transcripts(self):
        return lib().orc_index_num_transcripts(self.h)

    def transcript_len(self, tid):
        return lib().orc_index_transcript_len(self.h, tid)

    def __del__(self):
        try:
            lib().orc_index_free(self.h)
        except Exception:
            pass


def run(index, flags, batch, n_threads=1, want_matches=True, bam_records=None):
    """batch: dict of numpy arrays in the shared SoA layout (see bramble_amd.batch).
    Returns (rows dict, matches dict or None, seconds)."""
    L = lib()
    keep = {}

    def get(name, dtype):
        a = np.ascontiguousarray(batch[name], dtype=dtype)
        keep[name] = a
        return a

    b = OrcBatch()
    n = int(batch["n_aln"])
    b.n_aln = n
    b.ref_id = _ptr(get("ref_id", np.int32), C.c_int32)
    b.ref_start = _ptr(get("ref_start", np.int32), C.c_int32)
    b.flags = _ptr(get("flags", np.uint16), C.c_uint16)
    b.xs = _ptr(get("xs", np.int8), C.c_int8)
    b.ts = _ptr(get("ts", np.int8), C.c_int8)
    b.cigar_off = _ptr(get("cigar_off", np.uint64), C.c_uint64)
    b.cigar = _ptr(get("cigar", np.uint32), C.c_uint32)
    b.mate_ref_id = _ptr(get("mate_ref_id", np.int32), C.c_int32)
    b.mate_start = _ptr(get("mate_start", np.int32), C.c_int32)
    b.name_off = _ptr(get("name_off", np.uint64), C.c_uint64)
    names = bytes(np.ascontiguousarray(batch["names"], dtype=np.uint8).tobytes())
    keep["names"] = names
    b.names = names
    if batch.get("seq_off") is not None:
        b.seq_off = _ptr(get("seq_off", np.uint64), C.c_uint64)
        seqs = bytes(np.ascontiguousarray(batch["seqs"], dtype=np.uint8).tobytes())
        keep["seqs"] = seqs
        b.seqs = seqs
    b.l_qseq = _ptr(get("l_qseq", np.int32), C.c_int32)
    return _run_struct(index, flags, C.byref(b), n, n_threads, want_matches, bam_records)


def run_bam(index, flags, blob, rec_off, rec_len, ref_map, n_threads=1, want_matches=False):
    """Reader side + projection + write_to_bam over raw mapped BAM records (blob uint8[], rec_off uint64[n]
    pointing at each record's refID word, rec_len uint32[n]).  Returns (rows incl. bam_stream, matches, seconds,
    parsed) where parsed holds the reader-side tables (mate pairing is reported through matches['mate_idx'])."""
    L = lib()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    rec_len = np.ascontiguousarray(rec_len, dtype=np.uint32)
    ref_map = np.ascontiguousarray(ref_map, dtype=np.int32)
    n = len(rec_len)
    ph = L.orc_bam_parse(blob.ctypes.data, rec_off.ctypes.data, rec_len.ctypes.data, n, ref_map.ctypes.data, len(ref_map))
    try:
        bp = L.orc_parsed_batch(ph)
        v = bp.contents
        parsed = {"n_aln": n}
        for name, dt in (("ref_id", np.int32), ("ref_start", np.int32), ("flags", np.uint16), ("xs", np.int8),
                         ("ts", np.int8), ("mate_ref_id", np.int32), ("mate_start", np.int32), ("l_qseq", np.int32)):
            parsed[name] = _arr(getattr(v, name), n, dt)
        parsed["cigar_off"] = _arr(v.cigar_off, n + 1, np.uint64)
        parsed["name_off"] = _arr(v.name_off, n + 1, np.uint64)
        rows, matches, secs = _run_struct(index, flags, bp, n, n_threads, want_matches, (blob, rec_off, rec_len))
    finally:
        L.orc_parsed_free(ph)
    return rows, matches, secs, parsed


def _run_struct(index, flags, bref, n, n_threads, want_matches, bam_records):
    L = lib()
    h = L.orc_run(index.h, flags if isinstance(flags, C._Pointer) else C.byref(flags), bref, n_threads, 1 if want_matches else 0)
    try:
        r = L.orc_result_rows(h).contents
        nr = r.n_rows
        rows = {"n_rows": nr}
        for name, dt in (("input_index", np.int32), ("tid", np.uint32), ("pos", np.uint32), ("strand", np.int8),
                         ("similarity_score", np.float64), ("clip_score", np.int32), ("junc_hits", np.int32),
                         ("ref_consumed", np.int32), ("nh", np.uint32), ("hi", np.uint32), ("mapq", np.uint32),
                         ("primary", np.uint8), ("is_paired", np.uint8), ("same_transcript", np.uint8),
                         ("is_first", np.uint8), ("mate_tid", np.int32), ("mate_pos", np.int32),
                         ("isize", np.int32), ("group", np.uint32)):
            rows[name] = _arr(getattr(r, name), nr, dt)
        rows["cigar_off"] = _arr(r.cigar_off, nr + 1, np.uint64)
        rows["cigar"] = _arr(r.cigar, int(rows["cigar_off"][-1]), np.uint32)
        for name in ("total_complete", "total_unique", "dropped_reads", "total_processed"):
            rows[name] = int(getattr(r, name))
        matches = None
        if want_matches:
            m = L.orc_result_matches(h).contents
            nm = m.n_matches
            matches = {"n_matches": nm, "aln_off": _arr(m.aln_off, n + 1, np.uint64)}
            for name, dt in (("tid", np.uint32), ("fwpos", np.uint32), ("rcpos", np.uint32), ("strand", np.int8),
                             ("similarity_score", np.float64), ("total_coverage", np.float64),
                             ("total_operations", np.float64), ("junc_hits", np.int32),
                             ("ref_consumed", np.int32), ("clip_score", np.int32)):
                matches[name] = _arr(getattr(m, name), nm, dt)
            matches["ideal_off"] = _arr(m.ideal_off, nm + 1, np.uint64)
            matches["ideal"] = _arr(m.ideal, int(matches["ideal_off"][-1]), np.uint32)
            matches["out_off"] = _arr(m.out_off, nm + 1, np.uint64)
            matches["out"] = _arr(m.out, int(matches["out_off"][-1]), np.uint32)
            matches["n_exons"] = _arr(m.n_exons, n, np.int32)
            matches["mate_idx"] = _arr(m.mate_idx, n, np.int32)
        secs = L.orc_result_seconds(h)
        if bam_records is not None:
            # write_to_bam over the rows: bam_records = (blob uint8[], rec_off uint64[n+1])
            blob = np.ascontiguousarray(bam_records[0], dtype=np.uint8)
            roff = np.ascontiguousarray(bam_records[1], dtype=np.uint64)
            outp = C.c_void_p()
            rlen = None
            if len(bam_records) > 2 and bam_records[2] is not None:
                rlen = np.ascontiguousarray(bam_records[2], dtype=np.uint32)
            nb = L.orc_bam_encode(h, blob.ctypes.data, roff.ctypes.data, rlen.ctypes.data if rlen is not None else None,
                                  n, 1 if (flags.lr or flags.lr_hq) else 0, C.byref(outp))
            buf = (C.c_char * max(nb, 1)).from_address(outp.value)
            rows["bam_stream"] = np.frombuffer(buf, dtype=np.uint8, count=nb).copy()
            L.orc_free_buffer(outp)
    finally:
        L.orc_result_free(h)
    return rows, matches, secs


def merge_cigar(real, ideal):
    real = np.ascontiguousarray(real, dtype=np.uint32)
    ideal = np.ascontiguousarray(ideal, dtype=np.uint32)
    out = np.zeros(len(real) + len(ideal) + 1, dtype=np.uint32)
    n = lib().orc_merge_cigar(real.ctypes.data, len(real), ideal.ctypes.data, len(ideal), out.ctypes.data)
    return out[:n]


def segments(ref_start, cigar):
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    out = np.zeros(2 * (len(cigar) + 1), dtype=np.uint32)
    n = lib().orc_segments(ref_start, cigar.ctypes.data, len(cigar), out.ctypes.data, len(cigar) + 1)
    if n < 0:
        return None
    return out[:2 * n].reshape(-1, 2)


def resolve_config(flags):
    out5 = np.zeros(5, dtype=np.uint32)
    thr = C.c_float()
    fil = lib().orc_resolve_config(C.byref(flags), out5.ctypes.data, C.byref(thr))
    return {"max_clip": int(out5[0]), "max_junc_ins": int(out5[1]), "max_junc_gap": int(out5[2]),
            "max_error_exon": int(out5[3]), "ignore_small_exons": bool(out5[4]),
            "similarity_threshold": float(thr.value), "filter_by_similarity": bool(fil)}


def ksw_align(tseq, qseq):
    cap = len(tseq) + len(qseq) + 4
    out = np.zeros(cap, dtype=np.uint32)
    score, mx = C.c_int32(), C.c_int32()
    n = lib().orc_ksw_align(tseq.encode(), qseq.encode(), C.byref(score), C.byref(mx), out.ctypes.data, cap)
    return out[:n], score.value, mx.value


def primary_pick(name, n_tied):
    b = name if isinstance(name, bytes) else name.encode()
    return lib().orc_primary_pick(b, len(b), n_tied)
