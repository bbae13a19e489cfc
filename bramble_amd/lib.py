"""ctypes binding of libbramble_amd.so (C ABI: include/bramble_amd.h).

The projection path is HIP only: if the shared library is missing, or no HIP
device is usable, every call raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbramble_amd.so")
_P = C.POINTER

K_SEGMENT, K_COUNT, K_EMIT, K_PAIR_COUNT, K_PAIR_EMIT, K_GATHER, K_SCAN, K_EMIT_AUX, K_KSW, K_BAM, K_PARSE, K_CODEC, K_EMIT_SIMPLE, K_PRIMARY, K_CIGAR_POOL, K_COUNT_WALK, K_EXPAND, K_GROUP_IDS, K_P1, K_P1_WALK, K_EMIT_WL, K_NAME_SEED, K_PAIR_MASK, K_PAIR_BIG, K_GROUP_DESC, K_EXPAND_ROWS, K_EMIT_ROWS_SIMPLE, K_EMIT_ROWS, K_BIG_EMIT, K_NUM = range(30)
KERNEL_NAMES = ["k_segment", "k_project<G,false,false,1>", "k_emit_dense<false,2>", "k_pair<false>", "k_pair<true>",
                "k_rows", "k_scan_*", "k_project<64,true>", "k_ksw", "k_bam_scan+k_bam_size+k_bam_encode",
                "k_rec_fields+k_group_off+k_rec_copy+k_mates+k_seq_*", "k_deflate_*+k_bgzf_compact",
                "k_emit_dense<false,1>", "k_primary", "(unused)", "k_project<G,false,false,2>", "k_expand", "k_group_ids",
                "k_project1<G,1>", "k_project1<G,2>", "k_emit_wl", "k_name_seed", "k_pair_mask", "k_big<0>+k_pair_big", "k_group_desc",
                "k_expand_rows", "k_emit_rows<1>", "k_emit_rows<2>", "k_big<1>"]


class BrambleError(RuntimeError):
    pass


class BrExon(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32)]


class BrTranscript(C.Structure):
    _fields_ = [("id", C.c_char_p), ("seqname", C.c_char_p), ("strand", C.c_char),
                ("exons", _P(BrExon)), ("n_exons", C.c_uint32)]


class BrFastaSeq(C.Structure):
    _fields_ = [("name", C.c_char_p), ("seq", C.c_char_p), ("len", C.c_uint64)]


class BrConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("lr", "lr_hq", "strict", "use_fasta", "fr", "rf", "has_max_clip", "has_max_junc_ins",
                 "has_max_junc_gap", "has_sim_thr", "has_max_error_exon")] + \
               [(n, C.c_uint32) for n in ("max_clip", "max_junc_ins", "max_junc_gap", "max_error_exon")] + \
               [("sim_thr", C.c_float), ("junc_miss_discount", C.c_double)]


class BrThresholds(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("max_clip", "max_junc_ins", "max_junc_gap", "max_error_exon")] + \
               [("ignore_small_exons", C.c_int32), ("filter_by_similarity", C.c_int32),
                ("similarity_threshold", C.c_float)]


class BrBatch(C.Structure):
    _fields_ = [("n_aln", C.c_int64), ("ref_id", C.c_void_p), ("ref_start", C.c_void_p), ("flags", C.c_void_p),
                ("xs", C.c_void_p), ("ts", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("mate_ref_id", C.c_void_p), ("mate_start", C.c_void_p), ("name_off", C.c_void_p),
                ("names", C.c_void_p), ("seq_off", C.c_void_p), ("seqs", C.c_void_p), ("l_qseq", C.c_void_p)]


_ROW_FIELDS = [("input_index", np.int32), ("transcript_id", np.uint32), ("pos", np.uint32), ("strand", np.int8),
               ("cigar_off", np.uint64), ("cigar", np.uint32), ("similarity_score", np.float64),
               ("clip_score", np.int32), ("junc_hits", np.int32), ("aligned_len", np.int32),
               ("nh", np.uint32), ("hi", np.uint32), ("mapq", np.uint32)]
_ROW_TAIL = [("is_paired", np.uint8), ("same_transcript_as_mate", np.uint8), ("is_first", np.uint8),
             ("mate_transcript_id", np.int32), ("mate_pos", np.int32), ("insert_size", np.int32),
             ("group", np.uint32)]
_COUNTERS = [("total_complete", C.c_uint64), ("total_unique", C.c_uint64), ("dropped_reads", C.c_uint64),
             ("total_processed", C.c_uint64)]


class BrRows(C.Structure):
    _fields_ = [("n_rows", C.c_int64)] + [(n, C.c_void_p) for n, _ in _ROW_FIELDS] + \
               [("is_primary", C.c_void_p)] + [(n, C.c_void_p) for n, _ in _ROW_TAIL] + _COUNTERS


class BrDeviceBatch(C.Structure):
    _fields_ = [("n_aln", C.c_int64), ("n_groups", C.c_int64), ("ref_id", C.c_void_p), ("ref_start", C.c_void_p),
                ("flags", C.c_void_p), ("xs", C.c_void_p), ("ts", C.c_void_p), ("cigar_off", C.c_void_p),
                ("cigar", C.c_void_p), ("mate_idx", C.c_void_p), ("group_off", C.c_void_p), ("l_qseq", C.c_void_p),
                ("seq_off", C.c_void_p), ("seqs", C.c_void_p), ("n_cigar_words", C.c_int64),
                ("max_n_cigar", C.c_int32), ("seq_src", C.c_void_p), ("max_soft_clip", C.c_int32),
                ("name_off", C.c_void_p), ("names", C.c_void_p)]


class BrDeviceRows(C.Structure):
    """Packed rows (ABI version 2): a = {tid, pos, meta, nh}, cigar = inline ops or pool offset, x = {input, junc_hits,
    aligned_len, hi}."""
    _fields_ = [("n_rows", C.c_int64), ("n_matches", C.c_int64), ("n_pool_words", C.c_int64),
                ("a", C.c_void_p), ("cigar", C.c_void_p), ("x", C.c_void_p), ("similarity_score", C.c_void_p),
                ("clip_score", C.c_void_p), ("pool", C.c_void_p), ("row_off", C.c_void_p)] + _COUNTERS


class BrDeviceWideRows(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cigar_words", C.c_int64)] + \
               [(n, C.c_void_p) for n, _ in _ROW_FIELDS] + [(n, C.c_void_p) for n, _ in _ROW_TAIL] + \
               [("is_primary", C.c_void_p)]


class BrAlignment(C.Structure):  # br_alignment = GenomicAlignment (bramble-rs/src/api.rs:73-126)
    _fields_ = [("query_name", C.c_char_p), ("ref_id", C.c_int32), ("ref_start", C.c_int64),
                ("is_reverse", C.c_uint8), ("is_paired", C.c_uint8), ("is_first_in_pair", C.c_uint8),
                ("mate_is_unmapped", C.c_uint8), ("xs_strand", C.c_char), ("ts_strand", C.c_char),
                ("hit_index", C.c_int32), ("mate_ref_id", C.c_int32), ("mate_ref_start", C.c_int64),
                ("cigar", C.c_void_p), ("n_cigar", C.c_uint32), ("sequence", C.c_char_p), ("sequence_len", C.c_uint32),
                ("read_len", C.c_uint32)]


class BrProjected(C.Structure):  # br_projected = ProjectedAlignment (api.rs:135-176) + mapq + rewritten CIGAR
    _fields_ = [("transcript_id", C.c_uint32), ("transcript_start", C.c_uint32), ("transcript_end", C.c_uint32),
                ("aligned_len", C.c_uint32), ("query_aligned_len", C.c_uint32), ("is_reverse", C.c_uint8),
                ("transcript_strand", C.c_char), ("similarity_score", C.c_double), ("nh", C.c_uint32), ("hi", C.c_uint32), ("is_primary", C.c_uint8),
                ("same_transcript_as_mate", C.c_uint8), ("is_paired_out", C.c_uint8), ("insert_size", C.c_int32),
                ("input_index", C.c_uint64), ("mapq", C.c_uint32), ("cigar", _P(C.c_uint32)), ("n_cigar", C.c_uint32)]


class BrHostRows(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_aln", C.c_int64), ("n_groups", C.c_int64), ("n_pool_words", C.c_int64),
                ("a", C.c_void_p), ("cigar", C.c_void_p), ("pool", C.c_void_p), ("row_off", C.c_void_p),
                ("mate_idx", C.c_void_p), ("x", C.c_void_p), ("similarity_score", C.c_void_p),
                ("clip_score", C.c_void_p)] + _COUNTERS


ROW_MINUS, ROW_PAIRED, ROW_SAME_TX, ROW_FIRST, ROW_PRIMARY = 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28


class BrDeviceRecords(C.Structure):
    _fields_ = [("blob", C.c_void_p), ("rec_off", C.c_void_p), ("n_aln", C.c_int64), ("rec_len", C.c_void_p)]


class BrBamBundle(C.Structure):
    _fields_ = [("blob", C.c_void_p), ("n_bytes", C.c_uint64), ("rec_off", C.c_void_p), ("rec_len", C.c_void_p),
                ("n_records", C.c_int64), ("ref_map", C.c_void_p), ("n_ref_map", C.c_int32), ("bgzf_on_device", C.c_int32)]


BGZF_BLOCK = np.dtype([("src_off", "<u8"), ("dst_off", "<u8"), ("clen", "<u4"), ("ulen", "<u4"), ("crc", "<u4"), ("pad", "<u4")])


def bgzf_scan(data, cap=None):
    """br_bgzf_scan over a numpy uint8 array of BGZF bytes: (block table, bytes consumed, inflated bytes)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    cap = int(cap) if cap is not None else data.size // 28 + 1
    blocks = np.zeros(cap, dtype=BGZF_BLOCK)
    n, consumed, total = C.c_int64(), C.c_uint64(), C.c_uint64()
    L = lib()
    L.br_bgzf_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p, _P(C.c_int64), _P(C.c_uint64), _P(C.c_uint64)]
    check(L.br_bgzf_scan(data.ctypes.data, data.size, cap, blocks.ctypes.data, C.byref(n), C.byref(consumed), C.byref(total)), "br_bgzf_scan")
    return blocks[:n.value], int(consumed.value), int(total.value)


class BrHostBam(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n_bytes", C.c_uint64), ("n_rows", C.c_int64)] + _COUNTERS


class BrDeviceBam(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n_bytes", C.c_uint64), ("row_off", C.c_void_p), ("n_rows", C.c_int64)]


# every symbol include/bramble_amd.h declares
EXPORTS = ["br_index_build", "br_index_build_flat", "br_index_free", "br_index_num_transcripts", "br_index_transcript_name",
           "br_index_transcript_len", "br_index_num_refs", "br_index_num_intervals", "br_index_device_bytes", "br_config_short_read",
           "br_config_long_read", "br_config_resolve", "br_batch_prepare", "br_batch_seq_source", "br_ctx_new", "br_ctx_free",
           "br_project_batch", "br_project_batch_device", "br_device_rows_expand", "br_batch_stage", "br_project_staged", "br_host_rows_wait", "br_project_batch_packed",
           "br_pin_host", "br_unpin_host", "br_project_group", "br_project_groups", "br_bam_encode_device", "br_project_bam_device", "br_project_bam_bundle", "br_bam_bundle_stage", "br_project_bam_staged", "br_bam_split", "br_annotation_load", "br_annotation_load_mt", "br_annotation_free",
           "br_annotation_num_transcripts", "br_annotation_transcripts", "br_annotation_num_refs", "br_annotation_refnames", "br_cli_main", "br_cli_exit_at_end", "br_device_warmup", "br_project_bam_staged_nowait", "br_host_bam_wait", "br_bgzf_scan", "br_bgzf_inflate_device", "br_bam_split_device", "br_bam_reader_new", "br_bam_reader_next", "br_bam_reader_set_piece_blocks", "br_bam_reader_release", "br_bam_reader_free", "br_bam_piece_upload", "br_bam_piece_process", "br_bam_reader_seconds", "br_bam_reader_upload_seconds", "br_project_bam_resident", "br_bgzf_write_file", "br_bgzf_read_file",
           "br_free_buffer", "br_bgzf_codec", "br_bgzf_deflate_device", "br_ctx_set_profiling",
           "br_ctx_set_param", "br_ctx_kernel_ms", "br_ctx_kernel_ms_sum", "br_ctx_collect_counters", "br_ctx_last_counters", "br_ctx_rescue_stats", "br_ctx_ksw_diag", "br_device_rows_detail", "br_ctx_ksw_pairs", "br_primary_pick", "br_row_mapq", "br_version", "br_strerror"]

_LIB = None


def bam_split(data, cap=None):
    """Host walk of the block_size chain: numpy uint8 alignment section -> (rec_off uint64[n], rec_len uint32[n],
    n_unmapped, consumed bytes); unmapped records are skipped."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    cap = int(cap if cap is not None else max(data.size // 36 + 1, 1))
    off = np.zeros(cap, dtype=np.uint64)
    ln = np.zeros(cap, dtype=np.uint32)
    n, un, used = C.c_int64(), C.c_int64(), C.c_uint64()
    check(lib().br_bam_split(data.ctypes.data, data.size, cap, off.ctypes.data, ln.ctypes.data, C.byref(n), C.byref(un),
                             C.byref(used)), "br_bam_split")
    return off[:n.value].copy(), ln[:n.value].copy(), un.value, used.value


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise BrambleError("libbramble_amd.so is not built (run __graft_entry__.build() or "
                               "`make -C bramble_amd/csrc`); there is no CPU fallback for the projection path")
        # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's).  Importing torch first makes libbramble_amd.so bind to that copy,
        # so torch tensors' device pointers and streams are valid for our launches; two
        # runtimes side by side leave the second one without a usable device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.br_index_build.argtypes = [_P(BrTranscript), C.c_size_t, _P(C.c_char_p), C.c_size_t, _P(BrFastaSeq),
                                     C.c_size_t, C.c_int, _P(C.c_void_p)]
        L.br_index_build_flat.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, _P(BrFastaSeq), C.c_int, _P(C.c_void_p)]
        L.br_index_free.argtypes = [C.c_void_p]
        L.br_index_num_transcripts.restype = C.c_size_t
        L.br_index_num_transcripts.argtypes = [C.c_void_p]
        L.br_index_transcript_name.restype = C.c_char_p
        L.br_index_transcript_name.argtypes = [C.c_void_p, C.c_uint32]
        L.br_index_transcript_len.restype = C.c_int64
        L.br_index_transcript_len.argtypes = [C.c_void_p, C.c_uint32]
        L.br_index_num_intervals.restype = C.c_size_t
        L.br_index_num_intervals.argtypes = [C.c_void_p]
        L.br_index_device_bytes.restype = C.c_size_t
        L.br_index_device_bytes.argtypes = [C.c_void_p]
        L.br_config_short_read.argtypes = [_P(BrConfig)]
        L.br_config_long_read.argtypes = [_P(BrConfig)]
        L.br_config_resolve.argtypes = [_P(BrConfig), _P(BrThresholds)]
        L.br_batch_prepare.argtypes = [_P(BrBatch), C.c_void_p, C.c_void_p, _P(C.c_int64)]
        L.br_batch_seq_source.argtypes = [_P(BrBatch), C.c_void_p, C.c_int64, C.c_void_p]
        L.br_ctx_new.argtypes = [C.c_void_p, _P(C.c_void_p)]
        L.br_ctx_free.argtypes = [C.c_void_p]
        L.br_project_batch.argtypes = [C.c_void_p, _P(BrConfig), _P(BrBatch), _P(BrRows)]
        L.br_project_batch_device.argtypes = [C.c_void_p, _P(BrConfig), _P(BrDeviceBatch), C.c_void_p,
                                              _P(BrDeviceRows)]
        L.br_project_group.argtypes = [C.c_void_p, _P(BrConfig), _P(BrAlignment), C.c_size_t, _P(_P(BrProjected)),
                                       _P(C.c_size_t)]
        L.br_project_groups.argtypes = L.br_project_group.argtypes
        L.br_batch_stage.argtypes = [C.c_void_p, _P(BrBatch), C.c_int]
        L.br_project_staged.argtypes = [C.c_void_p, _P(BrConfig), C.c_int, _P(BrHostRows)]
        L.br_host_rows_wait.argtypes = [C.c_void_p, C.c_int]
        L.br_project_batch_packed.argtypes = [C.c_void_p, _P(BrConfig), _P(BrBatch), _P(BrHostRows)]
        L.br_pin_host.argtypes = [C.c_void_p, C.c_size_t]
        L.br_unpin_host.argtypes = [C.c_void_p]
        L.br_row_mapq.restype = C.c_uint32
        L.br_row_mapq.argtypes = [C.c_uint32, C.c_int]
        L.br_device_rows_expand.argtypes = [C.c_void_p, C.c_void_p, _P(BrDeviceWideRows)]
        L.br_bam_encode_device.argtypes = [C.c_void_p, _P(BrConfig), _P(BrDeviceRecords), C.c_void_p, _P(BrDeviceBam)]
        L.br_project_bam_device.argtypes = [C.c_void_p, _P(BrConfig), _P(BrDeviceRecords), C.c_void_p, C.c_int32, C.c_void_p,
                                            _P(BrDeviceRows), _P(BrDeviceBam)]
        L.br_project_bam_bundle.argtypes = [C.c_void_p, _P(BrConfig), _P(BrBamBundle), _P(BrHostBam)]
        L.br_bam_split.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, _P(C.c_int64), _P(C.c_int64),
                                   _P(C.c_uint64)]
        L.br_index_num_refs.restype = C.c_size_t
        L.br_index_num_refs.argtypes = [C.c_void_p]
        L.br_bgzf_deflate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, _P(C.c_void_p), _P(C.c_uint64)]
        L.br_ctx_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.br_ctx_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.br_ctx_kernel_ms.argtypes = [C.c_void_p, C.c_int, _P(C.c_double), _P(C.c_int32)]
        L.br_ctx_collect_counters.argtypes = [C.c_void_p, _P(BrDeviceBatch), C.c_void_p]
        L.br_ctx_last_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.br_ctx_rescue_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.br_ctx_ksw_diag.argtypes = [C.c_void_p, C.c_void_p]
        L.br_device_rows_detail.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.br_ctx_ksw_pairs.argtypes = [C.c_void_p, C.c_int64, _P(C.c_char_p), _P(C.c_char_p), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_uint32]
        L.br_primary_pick.restype = C.c_uint32
        L.br_primary_pick.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        L.br_version.restype = C.c_char_p
        L.br_strerror.restype = C.c_char_p
        L.br_strerror.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        raise BrambleError("%s failed: %s (%d)" % (what, lib().br_strerror(rc).decode(), rc))


def make_config(**kw):
    """lr, lr_hq, strict, use_fasta, fr, rf + optional max_clip, max_junc_ins, max_junc_gap,
    max_error_exon, sim_thr overrides (bramble CLI flags, src/bramble.cpp:457-485)."""
    c = BrConfig()
    lib().br_config_short_read(C.byref(c))
    for k, v in kw.items():
        if k in ("max_clip", "max_junc_ins", "max_junc_gap", "max_error_exon", "sim_thr"):
            setattr(c, "has_" + k, 1)
        setattr(c, k, v)
    return c


def resolve_config(cfg):
    t = BrThresholds()
    check(lib().br_config_resolve(C.byref(cfg), C.byref(t)), "br_config_resolve")
    return {"max_clip": t.max_clip, "max_junc_ins": t.max_junc_ins, "max_junc_gap": t.max_junc_gap,
            "max_error_exon": t.max_error_exon, "ignore_small_exons": bool(t.ignore_small_exons),
            "filter_by_similarity": bool(t.filter_by_similarity),
            "similarity_threshold": float(t.similarity_threshold)}


class Index:
    """g2tTree replacement: flattened exon tables, uploaded to `device` (-1: host only)."""

    def __init__(self, annotation, device=0):
        L = lib()
        txs = annotation["transcripts"]
        refnames = [r.encode() for r in annotation["refnames"]]
        arr = (BrTranscript * max(len(txs), 1))()
        keep = []
        for i, t in enumerate(txs):
            ex = (BrExon * max(len(t["exons"]), 1))()
            for k, (s, e) in enumerate(t["exons"]):
                ex[k].start, ex[k].end = int(s), int(e)
            keep.append(ex)
            arr[i].id = t["id"].encode()
            arr[i].seqname = (t["seqname"] if "seqname" in t else annotation["refnames"][t["ref_id"]]).encode()
            arr[i].strand = t["strand"].encode()
            arr[i].exons = ex
            arr[i].n_exons = len(t["exons"])
        rn = (C.c_char_p * max(len(refnames), 1))(*refnames)
        fa, nfa = None, 0
        seqs = annotation.get("ref_seqs")
        if seqs:
            items = sorted(seqs.items())
            fa = (BrFastaSeq * len(items))()
            for i, (rid, s) in enumerate(items):
                sb = s.encode() if isinstance(s, str) else bytes(s)
                keep.append(sb)
                fa[i].name = annotation["refnames"][rid].encode()
                fa[i].seq = sb
                fa[i].len = len(sb)
            nfa = len(items)
        h = C.c_void_p()
        check(L.br_index_build(arr, len(txs), rn, len(refnames), fa, nfa, device, C.byref(h)), "br_index_build")
        self.h = h
        self.device = device

    @classmethod
    def from_flat(cls, flat, device=0):
        """flat: dict with n_refs, tx_ref int32[], tx_strand int8[], tx_exon_off uint64[], ex_start/ex_end
        uint32[] (half-open) and optional ref_seqs (list of bytes or None per reference)."""
        self = cls.__new__(cls)
        ref = np.ascontiguousarray(flat["tx_ref"], dtype=np.int32)
        strand = np.ascontiguousarray(flat["tx_strand"], dtype=np.int8)
        off = np.ascontiguousarray(flat["tx_exon_off"], dtype=np.uint64)
        es = np.ascontiguousarray(flat["ex_start"], dtype=np.uint32)
        ee = np.ascontiguousarray(flat["ex_end"], dtype=np.uint32)
        fa = None
        keep = []
        if flat.get("ref_seqs") is not None:
            fa = (BrFastaSeq * flat["n_refs"])()
            for r, sq in enumerate(flat["ref_seqs"]):
                if sq is None:
                    continue
                sb = sq if isinstance(sq, (bytes, bytearray)) else bytes(sq)
                keep.append(sb)
                fa[r].name = b"ref%d" % r
                fa[r].seq = sb
                fa[r].len = len(sb)
        h = C.c_void_p()
        check(lib().br_index_build_flat(len(ref), ref.ctypes.data, strand.ctypes.data, off.ctypes.data,
                                        es.ctypes.data, ee.ctypes.data, None, int(flat["n_refs"]), fa, device,
                                        C.byref(h)), "br_index_build_flat")
        self.h = h
        self.device = device
        return self

    def num_transcripts(self):
        return lib().br_index_num_transcripts(self.h)

    def transcript_name(self, tid):
        s = lib().br_index_transcript_name(self.h, tid)
        return s.decode() if s is not None else None

    def transcript_len(self, tid):
        v = lib().br_index_transcript_len(self.h, tid)
        return None if v < 0 else int(v)

    def num_intervals(self):
        return lib().br_index_num_intervals(self.h)

    def device_bytes(self):
        return lib().br_index_device_bytes(self.h)

    def close(self):
        if self.h:
            lib().br_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _batch_struct(batch, keep):
    b = BrBatch()
    b.n_aln = int(batch["n_aln"])

    def put(name, dtype):
        a = np.ascontiguousarray(batch[name], dtype=dtype)
        keep.append(a)
        setattr(b, name, a.ctypes.data)

    for name, dt in (("ref_id", np.int32), ("ref_start", np.int32), ("flags", np.uint16), ("xs", np.int8),
                     ("ts", np.int8), ("cigar_off", np.uint64), ("cigar", np.uint32), ("mate_ref_id", np.int32),
                     ("mate_start", np.int32), ("name_off", np.uint64), ("names", np.uint8), ("l_qseq", np.int32)):
        put(name, dt)
    if batch.get("seq_off") is not None:
        put("seq_off", np.uint64)
        put("seqs", np.uint8)
    return b


def prepare_batch(batch):
    """Host-side input contract (br_batch_prepare): returns (mate_idx int32[n], group_off uint32[g+1])."""
    keep = []
    b = _batch_struct(batch, keep)
    n = int(batch["n_aln"])
    mate = np.full(max(n, 1), -1, dtype=np.int32)
    goff = np.zeros(n + 1, dtype=np.uint32)
    ng = C.c_int64()
    check(lib().br_batch_prepare(C.byref(b), mate.ctypes.data, goff.ctypes.data, C.byref(ng)), "br_batch_prepare")
    return mate[:n], goff[:ng.value + 1].copy()


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def host_rows_to_numpy(r):
    """BrHostRows -> dict of numpy copies: a uint32 [n, 4] = {tid, pos, meta, nh}, cigar uint64 [n], pool uint32,
    row_off uint64 [n_aln + 1], mate_idx int32 [n_aln], optional x / similarity_score / clip_score."""
    n, na = int(r.n_rows), int(r.n_aln)
    out = {"n_rows": n, "n_aln": na, "n_groups": int(r.n_groups),
           "a": _view(r.a, 4 * n, np.uint32).reshape(n, 4), "cigar": _view(r.cigar, n, np.uint64),
           "pool": _view(r.pool, int(r.n_pool_words), np.uint32), "row_off": _view(r.row_off, na + 1, np.uint64),
           "mate_idx": _view(r.mate_idx, na, np.int32)}
    if r.x:
        out["x"] = _view(r.x, 4 * n, np.uint32).reshape(n, 4)
    if r.similarity_score:
        out["similarity_score"] = _view(r.similarity_score, n, np.float64)
        out["clip_score"] = _view(r.clip_score, n, np.int32)
    for name, _ in _COUNTERS:
        out[name] = int(getattr(r, name))
    return out


def unpack_host_rows(p, l_qseq, long_reads=False):
    """Packed host rows (host_rows_to_numpy, with the x array) -> the wide dict project_batch returns (minus `group`),
    derived on the host exactly as include/bramble_amd.h documents: HI / MAPQ / mate fields / insert size."""
    n = p["n_rows"]
    a, x = p["a"], p["x"]
    meta = a[:, 2]
    ncig = (meta & 0xffffff).astype(np.int64)
    w = {"n_rows": n, "transcript_id": a[:, 0].copy(), "pos": a[:, 1].copy(), "nh": a[:, 3].copy(),
         "strand": np.where(meta & ROW_MINUS, ord("-"), ord("+")).astype(np.int8),
         "is_paired": ((meta & ROW_PAIRED) != 0).astype(np.uint8),
         "same_transcript_as_mate": ((meta & ROW_SAME_TX) != 0).astype(np.uint8),
         "is_first": ((meta & ROW_FIRST) != 0).astype(np.uint8), "is_primary": ((meta & ROW_PRIMARY) != 0).astype(np.uint8),
         "input_index": x[:, 0].astype(np.int32), "junc_hits": x[:, 1].astype(np.int32),
         "aligned_len": x[:, 2].astype(np.int32), "hi": x[:, 3].copy()}
    nh = w["nh"]
    if long_reads:
        w["mapq"] = np.where(nh > 1, 0, 3).astype(np.uint32)
    else:
        w["mapq"] = np.select([nh == 1, nh == 2, (nh == 3) | (nh == 4)], [255, 3, 1], 0).astype(np.uint32)
    coff = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(ncig, out=coff[1:])
    cig = np.zeros(int(coff[-1]), dtype=np.uint32)
    c = p["cigar"]
    one = np.nonzero(ncig >= 1)[0]
    inl = ncig <= 2
    i1 = np.nonzero(inl & (ncig >= 1))[0]
    cig[coff[i1].astype(np.int64)] = (c[i1] & np.uint64(0xffffffff)).astype(np.uint32)
    i2 = np.nonzero(ncig == 2)[0]
    cig[coff[i2].astype(np.int64) + 1] = (c[i2] >> np.uint64(32)).astype(np.uint32)
    for r in np.nonzero(~inl)[0]:
        o = int(c[r])
        cig[int(coff[r]):int(coff[r + 1])] = p["pool"][o:o + int(ncig[r])]
    del one
    w["cigar_off"], w["cigar"] = coff, cig
    paired = w["is_paired"].astype(bool)
    first = w["is_first"].astype(bool)
    idx = np.arange(n)
    nb = np.where(first, idx + 1, idx - 1)
    nb = np.where(paired, nb, idx)
    mate_pos = np.where(paired, a[nb, 1].astype(np.int64), -1)
    same = w["same_transcript_as_mate"].astype(bool)
    mate_tid = np.where(paired, np.where(same, a[:, 0].astype(np.int64), a[nb, 0].astype(np.int64)), -1)
    lq = np.asarray(l_qseq, dtype=np.int64)[w["input_index"]] if n else np.zeros(0, np.int64)
    my = a[:, 1].astype(np.int64)
    isz = np.where(my <= mate_pos, (mate_pos + lq) - my, -((my + lq) - mate_pos))
    w["mate_transcript_id"] = mate_tid.astype(np.int32)
    w["mate_pos"] = mate_pos.astype(np.int32)
    w["insert_size"] = np.where(paired & same, isz, 0).astype(np.int32)
    w["similarity_score"] = p.get("similarity_score", np.zeros(n))
    w["clip_score"] = p.get("clip_score", np.zeros(n, np.int32))
    for k in ("total_complete", "total_unique", "dropped_reads", "total_processed"):
        w[k] = p[k]
    return w


class Context:
    """ProjectionContext: device scratch + stream-ordered pipeline for one index."""

    def __init__(self, index):
        h = C.c_void_p()
        check(lib().br_ctx_new(index.h, C.byref(h)), "br_ctx_new")
        self.h = h
        self.index = index

    def set_param(self, key, value):
        check(lib().br_ctx_set_param(self.h, key.encode(), int(value)), "br_ctx_set_param")

    def set_profiling(self, enabled):
        check(lib().br_ctx_set_profiling(self.h, 1 if enabled else 0), "br_ctx_set_profiling")

    def kernel_ms(self):
        out = {}
        for k in range(K_NUM):
            ms, ln = C.c_double(), C.c_int32()
            check(lib().br_ctx_kernel_ms(self.h, k, C.byref(ms), C.byref(ln)), "br_ctx_kernel_ms")
            out[KERNEL_NAMES[k]] = (ms.value, ln.value)
        return out

    def kernel_ms_sum(self):
        """{kernel name: (ms, launches)} summed over the calls since set_profiling(True)."""
        out = {}
        L = lib()
        L.br_ctx_kernel_ms_sum.argtypes = [C.c_void_p, C.c_int, _P(C.c_double), _P(C.c_int64)]
        for k in range(K_NUM):
            ms, ln = C.c_double(), C.c_int64()
            check(L.br_ctx_kernel_ms_sum(self.h, k, C.byref(ms), C.byref(ln)), "br_ctx_kernel_ms_sum")
            out[KERNEL_NAMES[k]] = (ms.value, ln.value)
        return out

    def project_batch(self, cfg, batch):
        """Host batch in, rows (dict of numpy arrays) out: convert_reads minus BAM writing."""
        keep = []
        b = _batch_struct(batch, keep)
        r = BrRows()
        check(lib().br_project_batch(self.h, C.byref(cfg), C.byref(b), C.byref(r)), "br_project_batch")
        n = r.n_rows
        rows = {"n_rows": n}
        for name, dt in _ROW_FIELDS + [("is_primary", np.uint8)] + _ROW_TAIL:
            if name == "cigar_off":
                rows[name] = _view(r.cigar_off, n + 1, dt)
            elif name == "cigar":
                continue
            else:
                rows[name] = _view(getattr(r, name), n, dt)
        rows["cigar"] = _view(r.cigar, int(rows["cigar_off"][-1]) if n else 0, np.uint32)
        for name, _ in _COUNTERS:
            rows[name] = int(getattr(r, name))
        return rows

    @staticmethod
    def _device_batch_struct(dev_batch):
        db = BrDeviceBatch()
        db.n_aln = dev_batch["n_aln"]
        db.n_groups = dev_batch["n_groups"]
        for name in ("ref_id", "ref_start", "flags", "xs", "ts", "cigar_off", "cigar", "mate_idx", "group_off",
                     "l_qseq"):
            setattr(db, name, dev_batch[name].data_ptr())
        db.n_cigar_words = dev_batch["n_cigar_words"]
        db.max_n_cigar = dev_batch["max_n_cigar"]
        if dev_batch.get("seqs") is not None:
            db.seq_off = dev_batch["seq_off"].data_ptr()
            db.seqs = dev_batch["seqs"].data_ptr()
            db.seq_src = dev_batch["seq_src"].data_ptr()
            db.max_soft_clip = dev_batch["max_soft_clip"]
        if dev_batch.get("names") is not None:
            db.name_off = dev_batch["name_off"].data_ptr()
            db.names = dev_batch["names"].data_ptr()
        return db

    def collect_counters(self, dev_batch, stream=0):
        """Exact algorithmic-bytes counters of the batch projected last."""
        db = self._device_batch_struct(dev_batch)
        check(lib().br_ctx_collect_counters(self.h, C.byref(db), C.c_void_p(stream)), "br_ctx_collect_counters")
        out = (C.c_uint64 * 8)()
        check(lib().br_ctx_last_counters(self.h, out), "br_ctx_last_counters")
        keys = ("B_in", "B_idx", "B_out", "n_cigar", "read_exons", "overlap_hits", "matches", "out_cigar_words")
        return dict(zip(keys, [int(v) for v in out]))

    def project_bam_device(self, cfg, blob, rec_off, rec_len, ref_map, stream=0):
        """Raw mapped BAM records resident in HBM -> (BrDeviceRows, BrDeviceBam): reader side, projection and
        record re-encoding, all on the device.  blob uint8 / rec_off int64 [n] / rec_len int32 [n] are torch CUDA
        tensors; ref_map is a host int32 array (input refID -> annotation reference index)."""
        recs = BrDeviceRecords()
        recs.blob = blob.data_ptr()
        recs.rec_off = rec_off.data_ptr()
        recs.n_aln = rec_len.numel()
        recs.rec_len = rec_len.data_ptr()
        rm = np.ascontiguousarray(ref_map, dtype=np.int32)
        rows, out = BrDeviceRows(), BrDeviceBam()
        check(lib().br_project_bam_device(self.h, C.byref(cfg), C.byref(recs), rm.ctypes.data, len(rm), C.c_void_p(stream),
                                          C.byref(rows), C.byref(out)), "br_project_bam_device")
        return rows, out

    def bgzf_deflate_device(self, src, stream=0):
        """src: torch CUDA uint8 tensor -> torch CUDA uint8 tensor view of the concatenated BGZF blocks (valid until the
        next call on this context)."""
        import torch
        from .device import _DevArray
        out, n = C.c_void_p(), C.c_uint64()
        check(lib().br_bgzf_deflate_device(self.h, C.c_void_p(src.data_ptr()), src.numel(), C.c_void_p(stream), C.byref(out),
                                           C.byref(n)), "br_bgzf_deflate_device")
        if n.value == 0:
            return torch.zeros(0, dtype=torch.uint8, device=src.device)
        return torch.as_tensor(_DevArray(out.value, n.value, "|u1"), device=src.device)

    def bgzf_inflate_device(self, src, blocks, stream=0):
        """src: torch CUDA uint8 tensor holding BGZF bytes, blocks: the table bgzf_scan made of the same bytes -> torch CUDA
        uint8 tensor view of the inflated stream (valid until the next call on this context)."""
        import torch
        from .device import _DevArray
        out, n = C.c_void_p(), C.c_uint64()
        blocks = np.ascontiguousarray(blocks, dtype=BGZF_BLOCK)
        L = lib()
        L.br_bgzf_inflate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, _P(C.c_void_p), _P(C.c_uint64)]
        check(L.br_bgzf_inflate_device(self.h, C.c_void_p(src.data_ptr()), src.numel(), C.c_void_p(blocks.ctypes.data), len(blocks),
                                       C.c_void_p(stream), C.byref(out), C.byref(n)), "br_bgzf_inflate_device")
        if n.value == 0:
            return torch.zeros(0, dtype=torch.uint8, device=src.device)
        return torch.as_tensor(_DevArray(out.value, n.value, "|u1"), device=src.device)

    def project_bam_bundle(self, cfg, blob, rec_off, rec_len, ref_map, bgzf_on_device=False):
        """Host form: numpy blob / rec_off (uint64) / rec_len (uint32) in, (stream uint8[], counters dict) out."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        rec_len = np.ascontiguousarray(rec_len, dtype=np.uint32)
        rm = np.ascontiguousarray(ref_map, dtype=np.int32)
        bb = BrBamBundle(blob.ctypes.data, blob.size, rec_off.ctypes.data, rec_len.ctypes.data, len(rec_len),
                         rm.ctypes.data, len(rm), 1 if bgzf_on_device else 0)
        out = BrHostBam()
        check(lib().br_project_bam_bundle(self.h, C.byref(cfg), C.byref(bb), C.byref(out)), "br_project_bam_bundle")
        n = int(out.n_bytes)
        data = np.ctypeslib.as_array(C.cast(out.data, _P(C.c_uint8)), shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        return data, {"n_rows": int(out.n_rows), "total_complete": int(out.total_complete),
                      "total_unique": int(out.total_unique), "dropped_reads": int(out.dropped_reads),
                      "total_processed": int(out.total_processed)}

    def bam_encode_device(self, cfg, blob, rec_off, stream=0):
        """Re-encode every row of the last project_batch_device call as BAM records.
        blob / rec_off: torch CUDA tensors (uint8 record bytes, int64 offsets n_aln+1)."""
        recs = BrDeviceRecords()
        recs.blob = blob.data_ptr()
        recs.rec_off = rec_off.data_ptr()
        recs.n_aln = rec_off.numel() - 1
        recs.rec_len = None
        out = BrDeviceBam()
        check(lib().br_bam_encode_device(self.h, C.byref(cfg), C.byref(recs), C.c_void_p(stream), C.byref(out)),
              "br_bam_encode_device")
        return out

    def ksw_pairs(self, pairs, cap=None):
        """Diagnostic: k_ksw alone on [(target, query)] -> (ok int32[n], max int32[n], [cigar uint32[] per pair])."""
        n = len(pairs)
        cap = int(cap or max([len(t) + len(q) + 4 for t, q in pairs] + [8]))
        ts = (C.c_char_p * max(n, 1))(*[t.encode() for t, _ in pairs])
        qs = (C.c_char_p * max(n, 1))(*[q.encode() for _, q in pairs])
        ok = np.zeros(max(n, 1), np.int32)
        mx = np.zeros(max(n, 1), np.int32)
        nc = np.zeros(max(n, 1), np.uint32)
        cig = np.zeros(max(n, 1) * cap, np.uint32)
        check(lib().br_ctx_ksw_pairs(self.h, n, ts, qs, ok.ctypes.data, mx.ctypes.data, nc.ctypes.data, cig.ctypes.data, cap),
              "br_ctx_ksw_pairs")
        return ok[:n], mx[:n], [cig[p * cap:p * cap + int(nc[p])].copy() for p in range(n)]

    def rescue_stats(self):
        out = (C.c_uint64 * 4)()
        check(lib().br_ctx_rescue_stats(self.h, out), "br_ctx_rescue_stats")
        return dict(zip(("problems", "dp_cells", "rescued", "seq_bytes"), [int(v) for v in out]))

    def rows_detail(self, stream=0):
        """Device pointer of the br_row_x array of the last call's rows (derived on first request)."""
        x = C.c_void_p()
        check(lib().br_device_rows_detail(self.h, C.c_void_p(stream), C.byref(x)), "br_device_rows_detail")
        return x.value

    def ksw_diag(self):
        out = (C.c_uint64 * 16)()
        check(lib().br_ctx_ksw_diag(self.h, out), "br_ctx_ksw_diag")
        v = [int(x) for x in out]
        return {"pieces": v[0], "per_shape": v[1:5], "leftover_before": v[5], "tape_bytes": v[6], "leftover_after": v[7],
                "rows_per_shape": v[8:12]}

    def project_batch_packed(self, cfg, batch):
        """Host batch in, packed host rows (dict of numpy arrays copied out of the context's pinned buffers) out."""
        keep = []
        b = _batch_struct(batch, keep)
        r = BrHostRows()
        check(lib().br_project_batch_packed(self.h, C.byref(cfg), C.byref(b), C.byref(r)), "br_project_batch_packed")
        return host_rows_to_numpy(r)

    def project_groups(self, cfg, alns):
        """br_project_groups: any number of name-collated groups in one call (same dicts as project_group)."""
        return self.project_group(cfg, alns, _many=True)

    def project_group(self, cfg, alns, _many=False):
        """project_group_with (bramble-rs/src/api.rs:285-290): alns = list of dicts with the GenomicAlignment fields
        (query_name, ref_id, ref_start, cigar [uint32 BAM-packed] and optional is_reverse, is_paired, is_first_in_pair,
        mate_is_unmapped, xs_strand, ts_strand, hit_index, mate_ref_id, mate_ref_start, sequence, read_len) -> list of
        dicts with the br_projected fields.  Raises BrambleError(BR_ERR_INVALID_ARG) when the query names differ."""
        n = len(alns)
        arr = (BrAlignment * max(n, 1))()
        keep = []
        for i, a in enumerate(alns):
            cg = np.ascontiguousarray(a["cigar"], dtype=np.uint32)
            keep.append(cg)
            x = arr[i]
            x.query_name = a["query_name"].encode()
            x.ref_id = int(a["ref_id"])
            x.ref_start = int(a["ref_start"])
            x.is_reverse = int(bool(a.get("is_reverse")))
            x.is_paired = int(bool(a.get("is_paired")))
            x.is_first_in_pair = int(bool(a.get("is_first_in_pair")))
            x.mate_is_unmapped = int(bool(a.get("mate_is_unmapped")))
            x.xs_strand = (a.get("xs_strand") or "\0").encode()
            x.ts_strand = (a.get("ts_strand") or "\0").encode()
            x.hit_index = int(a.get("hit_index", 0))
            x.mate_ref_id = int(a.get("mate_ref_id", -1))
            x.mate_ref_start = int(a.get("mate_ref_start", 0))
            x.cigar = cg.ctypes.data if len(cg) else None
            x.n_cigar = len(cg)
            sq = a.get("sequence")
            if sq:
                sb = sq.encode() if isinstance(sq, str) else bytes(sq)
                keep.append(sb)
                x.sequence = sb
                x.sequence_len = len(sb)
            x.read_len = int(a.get("read_len", 0))
        out, n_out = _P(BrProjected)(), C.c_size_t()
        fn = lib().br_project_groups if _many else lib().br_project_group
        check(fn(self.h, C.byref(cfg), arr, n, C.byref(out), C.byref(n_out)), "br_project_groups" if _many else "br_project_group")
        res = []
        for k in range(n_out.value):
            p = out[k]
            d = {f: getattr(p, f) for f, _ in BrProjected._fields_ if f != "cigar"}
            d["transcript_strand"] = p.transcript_strand.decode()
            d["cigar"] = np.array([p.cigar[j] for j in range(p.n_cigar)], dtype=np.uint32)
            res.append(d)
        return res

    def expand_rows(self, stream=0):
        """Wide (one array per field) view of the last projection call's rows: BrDeviceWideRows."""
        out = BrDeviceWideRows()
        check(lib().br_device_rows_expand(self.h, C.c_void_p(stream), C.byref(out)), "br_device_rows_expand")
        return out

    def project_batch_device(self, cfg, dev_batch, stream=0):
        """dev_batch: dict of torch CUDA tensors (see bramble_amd.device.upload_batch).  Returns the
        BrDeviceRows struct (packed rows; device pointers owned by the context)."""
        db = self._device_batch_struct(dev_batch)
        out = BrDeviceRows()
        check(lib().br_project_batch_device(self.h, C.byref(cfg), C.byref(db), C.c_void_p(stream), C.byref(out)),
              "br_project_batch_device")
        return out

    def close(self):
        if self.h:
            lib().br_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
