"""torch plumbing for HBM-resident batches: device memory and streams only (the
compute is libbramble_amd.so).  A device batch is the dict that
`lib.Context.project_batch_device` takes."""
import numpy as np
import torch

from . import lib


def upload_batch(batch, device="cuda:0"):
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
