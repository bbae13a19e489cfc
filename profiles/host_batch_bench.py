#!/usr/bin/env python3
"""PCIe-inclusive rate of br_project_batch (host SoA in, host rows out): profiles/host_batch_bench.py [pairs]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bramble_amd import lib, synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ann = synth.Annotation("G")
batch = ann.reads(pairs, "pe")
idx = lib.Index.from_flat(ann.flat, device=0)
ctx = lib.Context(idx)
cfg = lib.make_config()
keep = []
b = lib._batch_struct(batch, keep)
r = lib.BrRows()
L = lib.lib()
for it in range(3):
    t0 = time.perf_counter()
    rc = L.br_project_batch(ctx.h, C.byref(cfg), C.byref(b), C.byref(r))
    dt = time.perf_counter() - t0
    assert rc == 0
    print("call %d: %.3f s, %d alignments -> %d rows, %.1f M alignments/s" % (it, dt, batch["n_aln"], r.n_rows, batch["n_aln"] / dt / 1e6), flush=True)
