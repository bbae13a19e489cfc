"""torch plumbing for HBM-resident batches: device memory and streams only (the
compute is libbramble_amd.so).  A device batch is the dict that
`lib.Context.project_batch_device` takes."""
import numpy as np
import torch

from . import lib


def upload_batch(batch, device="cuda:0", with_names=True):
    """Host SoA batch -> torch tensors on `device`, plus the host-side input contract
    (read-name groups, mate index; br_batch_prepare)."""
    mate, goff = lib.prepare_batch(batch)
    n = int(batch["n_aln"])
    coff64 = np.asarray(batch["cigar_off"], dtype=np.uint64)
    if n and int(coff64[-1]) >= 2 ** 32 - n - 16:
        raise lib.BrambleError("batch exceeds 32-bit device offsets; split it")
    coff = coff64.astype(np.uint32)

    def t(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.size == 0:
            a = np.zeros(1, dtype=dtype)
        # torch has no uint16/uint32 arithmetic needs here: ship raw bytes under a same-width signed dtype
        view = {np.uint16: np.int16, np.uint32: np.int32, np.uint64: np.int64}.get(dtype, dtype)
        return torch.from_numpy(a.view(view)).to(device, non_blocking=False)

    d = {
        "n_aln": n, "n_groups": len(goff) - 1,
        "ref_id": t(batch["ref_id"], np.int32), "ref_start": t(batch["ref_start"], np.int32),
        "flags": t(batch["flags"], np.uint16), "xs": t(batch["xs"], np.int8), "ts": t(batch["ts"], np.int8),
        "cigar_off": t(coff, np.uint32), "cigar": t(batch["cigar"], np.uint32),
        "mate_idx": t(mate, np.int32), "group_off": t(goff, np.uint32), "l_qseq": t(batch["l_qseq"], np.int32),
        "n_cigar_words": int(coff64[-1]) if n else 0,
        "max_n_cigar": int(np.diff(coff64).max()) if n else 0,
    }
    if with_names:
        noff64 = np.asarray(batch["name_off"], dtype=np.uint64)
        if n and int(noff64[-1]) >= 2 ** 32 - 16:
            raise lib.BrambleError("read names exceed 32-bit device offsets; split the batch")
        d["name_off"] = t(noff64.astype(np.uint32), np.uint32)
        d["names"] = t(batch["names"], np.uint8)
    if batch.get("seq_off") is not None:
        soff64 = np.asarray(batch["seq_off"], dtype=np.uint64)
        if n and int(soff64[-1]) >= 2 ** 32 - 16:
            raise lib.BrambleError("batch sequences exceed 32-bit device offsets; split it")
        src = np.full(max(n, 1), -1, dtype=np.int32)
        keep = []
        bs = lib._batch_struct(batch, keep)
        import ctypes as C
        lib.check(lib.lib().br_batch_seq_source(C.byref(bs), goff.ctypes.data, len(goff) - 1, src.ctypes.data),
                  "br_batch_seq_source")
        cig = np.asarray(batch["cigar"], dtype=np.uint32)
        sc = cig[(cig & 0xF) == 4] >> 4
        d["seq_off"] = t(soff64.astype(np.uint32), np.uint32)
        d["seqs"] = t(batch["seqs"], np.uint8)
        d["seq_src"] = t(src[:n], np.int32)
        d["max_soft_clip"] = int(sc.max()) if sc.size else 0
    return d


class _DevArray:
    """Zero-copy view of context-owned device memory for torch (valid until the next projection call)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


_TYPESTR = {"int8": "|i1", "uint8": "|u1", "int32": "<i4", "int64": "<i8", "float64": "<f8"}


def packed_as_tensors(rows, n_aln, device="cuda:0", ctx=None, stream=0):
    """BrDeviceRows (packed) -> dict of torch tensors: a int32 [n, 4] = {tid, pos, meta, nh}, cigar int64 [n],
    x int32 [n, 4] = {input, junc_hits, aligned_len, hi}, pool int32, row_off int64 [n_aln + 1]."""
    n = int(rows.n_rows)
    out = {"n_rows": n, "row_off": torch.as_tensor(_DevArray(rows.row_off, n_aln + 1, "<i8"), device=device)}
    if n == 0:
        return out
    out["a"] = torch.as_tensor(_DevArray(rows.a, 4 * n, "<i4"), device=device).view(n, 4)
    if ctx is not None:   # the detail column is derived on request (br_device_rows_detail)
        out["x"] = torch.as_tensor(_DevArray(ctx.rows_detail(stream), 4 * n, "<i4"), device=device).view(n, 4)
    out["cigar"] = torch.as_tensor(_DevArray(rows.cigar, n, "<i8"), device=device)
    if int(rows.n_pool_words):
        out["pool"] = torch.as_tensor(_DevArray(rows.pool, int(rows.n_pool_words), "<i4"), device=device)
    if rows.similarity_score:
        out["similarity_score"] = torch.as_tensor(_DevArray(rows.similarity_score, n, "<f8"), device=device)
        out["clip_score"] = torch.as_tensor(_DevArray(rows.clip_score, n, "<i4"), device=device)
    return out


def rows_as_tensors(ctx, device="cuda:0", stream=0):
    """Wide view of the context's last projection call (br_device_rows_expand) -> dict of torch tensors (unsigned
    32/64-bit columns are viewed as signed)."""
    rows = ctx.expand_rows(stream)
    n = int(rows.n_rows)
    spec = {"input_index": "int32", "transcript_id": "int32", "pos": "int32", "strand": "int8",
            "similarity_score": "float64", "clip_score": "int32", "junc_hits": "int32", "aligned_len": "int32",
            "nh": "int32", "hi": "int32", "mapq": "int32", "is_paired": "uint8", "same_transcript_as_mate": "uint8",
            "is_first": "uint8", "mate_transcript_id": "int32", "mate_pos": "int32", "insert_size": "int32",
            "group": "int32", "is_primary": "uint8"}
    out = {"n_rows": n}
    if n == 0:
        return out
    for name, ty in spec.items():
        out[name] = torch.as_tensor(_DevArray(getattr(rows, name), n, _TYPESTR[ty]), device=device)
    out["cigar_off"] = torch.as_tensor(_DevArray(rows.cigar_off, n + 1, "<i8"), device=device)
    nw = int(rows.n_cigar_words)
    if nw:
        out["cigar"] = torch.as_tensor(_DevArray(rows.cigar, nw, "<i4"), device=device)
    return out


def upload_records(batch, device="cuda:0"):
    """Original BAM records of the batch (batch['rec_blob'], batch['rec_off']) -> (blob, rec_off) CUDA tensors."""
    blob = np.ascontiguousarray(batch["rec_blob"], dtype=np.uint8)
    off = np.ascontiguousarray(batch["rec_off"], dtype=np.uint64).view(np.int64)
    return torch.from_numpy(blob).to(device), torch.from_numpy(off).to(device)


def bam_stream_to_host(bam, device="cuda:0"):
    """BrDeviceBam -> numpy uint8 array with the uncompressed BAM record stream."""
    n = int(bam.n_bytes)
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    return torch.as_tensor(_DevArray(bam.data, n, "|u1"), device=device).cpu().numpy().copy()
