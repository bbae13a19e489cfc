import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from bramble_amd import lib, synth, device as brdev
mode, flags = sys.argv[1], ({"lr_hq": 1} if sys.argv[1] == "hifi" else {"lr": 1})
ann = synth.Annotation("G")
batch = ann.reads(300000, mode, with_records=1)
cfg = lib.make_config(**flags)
idx = lib.Index.from_flat(ann.flat, device=0)
for lanes in (0, 8, 0, 8):
    ctx = lib.Context(idx)
    ctx.set_param("bam_lanes", lanes)
    db = brdev.upload_batch(batch, "cuda:0")
    blob, roff = brdev.upload_records(batch, "cuda:0")
    rows = ctx.project_batch_device(cfg, db, 0)
    bam = ctx.bam_encode_device(cfg, blob, roff, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        bam = ctx.bam_encode_device(cfg, blob, roff, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(mode, "lanes", lanes, "rows", int(bam.n_rows), "bytes", int(bam.n_bytes), "ms", round(dt * 1e3, 3), "GB/s out", round(int(bam.n_bytes) / dt / 1e9, 1), flush=True)
    ctx.close()
