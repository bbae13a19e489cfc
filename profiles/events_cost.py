"""What the per-kernel hipEvents of the bench's timed region cost the step: the same context and batch, ten steps with the
context's profiling off, ten with it on (as bench.py times them), alternated.  python3 profiles/events_cost.py [pairs]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from bramble_amd import device as brdev  # noqa: E402
from bramble_amd import lib, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ann = synth.Annotation("G")
index = lib.Index.from_flat(ann.flat, device=0)
ctx = lib.Context(index)
cfg = lib.make_config()
b = ann.reads(P, "pe", seed=1)
db = brdev.upload_batch(b, "cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ctx.project_batch_device(cfg, db, st)
for rep in range(3):
    for on in (False, True):
        ctx.set_profiling(on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.project_batch_device(cfg, db, st)
            if on:
                ctx.kernel_ms()
        torch.cuda.synchronize()
        print("per-kernel events %s: %.3f ms per step" % ("on " if on else "off", 1e2 * (time.perf_counter() - t0)), flush=True)
