"""Where the PCIe-inclusive step spends its time on the host: per call of the bench's pipelined loop
(br_batch_stage / br_project_staged / br_host_rows_wait), direct rows on or off (BRAMBLE_AMD_DIRECT_ROWS).
python3 profiles/pcie_phases.py [pairs]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bramble_amd import lib, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ann = synth.Annotation("G")
index = lib.Index.from_flat(ann.flat, device=0)
ctx = lib.Context(index)
cfg = lib.make_config()
batch = ann.reads(P, "pe", seed=11)
n_aln = int(batch["n_aln"])
pinned = {"n_aln": n_aln, "seq_off": None, "seqs": None}
for name, dt in (("ref_id", np.int32), ("ref_start", np.int32), ("flags", np.uint16), ("xs", np.int8), ("ts", np.int8), ("cigar_off", np.uint64),
                 ("cigar", np.uint32), ("mate_ref_id", np.int32), ("mate_start", np.int32), ("name_off", np.uint64), ("names", np.uint8), ("l_qseq", np.int32)):
    a = np.ascontiguousarray(batch[name], dtype=dt)
    view = {np.uint16: np.int16, np.uint32: np.int32, np.uint64: np.int64}.get(dt, dt)
    pinned[name] = torch.from_numpy(a.view(view)).pin_memory().numpy().view(dt)
keep = []
bs = lib._batch_struct(pinned, keep)
L = lib.lib()
res = [lib.BrHostRows(), lib.BrHostRows()]


def run(n_steps, log):
    t = time.perf_counter
    lib.check(L.br_batch_stage(ctx.h, C.byref(bs), 0), "stage")
    for k in range(n_steps):
        t0 = t()
        if k + 1 < n_steps:
            lib.check(L.br_batch_stage(ctx.h, C.byref(bs), (k + 1) % 2), "stage")
        t1 = t()
        lib.check(L.br_project_staged(ctx.h, C.byref(cfg), k % 2, C.byref(res[k % 2])), "project")
        t2 = t()
        if k >= 1:
            lib.check(L.br_host_rows_wait(ctx.h, (k - 1) % 2), "wait")
        t3 = t()
        if log:
            print("step %d: stage %.1f ms, project %.1f ms, wait for the rows of the step before %.1f ms" % (k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
    lib.check(L.br_host_rows_wait(ctx.h, (n_steps - 1) % 2), "wait")


# what bench.py does with the context before this leg (PCIE_PHASES_PRE=resident,profile,counters): which of it changes the leg?
import os
from bramble_amd import device as brdev  # noqa: E402
pre = os.environ.get("PCIE_PHASES_PRE", "").split(",")
if "resident" in pre or "profile" in pre or "counters" in pre:
    dbatch = brdev.upload_batch(batch, "cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    if "resident" in pre:
        for _ in range(3):
            ctx.project_batch_device(cfg, dbatch, stream)
    if "profile" in pre:
        ctx.set_profiling(True)
        for _ in range(3):
            ctx.project_batch_device(cfg, dbatch, stream)
            ctx.kernel_ms()
        ctx.set_profiling(False)
    if "counters" in pre:
        ctx.collect_counters(dbatch, stream)
    torch.cuda.synchronize()
run(3, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(6, True)
print("%.2f ms per step" % (1e3 * (time.perf_counter() - t0) / 6))
