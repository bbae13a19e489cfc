"""What does running two half batches side by side buy?  Two contexts on one device, each with its own stream and its
own batch of P/2 read pairs, stepped by two host threads at once, against one context with P pairs (the bench's shape).
If the pair finishes its 2 x P/2 pairs clearly sooner than the single context its P, the step's kernels leave room beside
each other (the count pass waits on latency, the emit pass on bandwidth) and a step split into two halves on two
streams would pay.  python3 profiles/overlap_probe.py [pairs]"""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from bramble_amd import device as brdev  # noqa: E402
from bramble_amd import lib, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
STEPS = 6
ann = synth.Annotation("G")
index = lib.Index.from_flat(ann.flat, device=0)
cfg = lib.make_config()


def make(pairs, seed):
    ctx = lib.Context(index)
    b = ann.reads(pairs, "pe", seed=seed)
    return ctx, brdev.upload_batch(b, "cuda:0"), int(b["n_aln"]), torch.cuda.Stream()


def run(ctx, db, st, steps):
    for _ in range(steps):
        ctx.project_batch_device(cfg, db, st.cuda_stream)


one = make(P, 1)
halves = [make(P // 2, 2), make(P // 2, 3)]
for c in [one] + halves:
    run(c[0], c[1], c[3], 2)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); run(one[0], one[1], one[3], STEPS); torch.cuda.synchronize(); t_one = (time.perf_counter() - t0) / STEPS
    t0 = time.perf_counter(); run(halves[0][0], halves[0][1], halves[0][3], STEPS); torch.cuda.synchronize(); t_half = (time.perf_counter() - t0) / STEPS
    th = [threading.Thread(target=run, args=(h[0], h[1], h[3], STEPS)) for h in halves]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / STEPS
    print("one context, %d pairs: %.3f ms per step | one context, %d pairs: %.3f ms | two contexts side by side, %d pairs each: %.3f ms per step of both (%.2fx the single context's rate)"
          % (P, 1e3 * t_one, P // 2, 1e3 * t_half, P // 2, 1e3 * t_two, t_one * (halves[0][2] + halves[1][2]) / one[2] / t_two), flush=True)
