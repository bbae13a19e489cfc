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
    return d
