#!/usr/bin/env python3
"""Small batches: ms per device-resident step without the per-kernel events (bench.py keeps them on inside its timed
region; at 10 k alignments they are a third of the step), the path without host round trips against the ordinary one,
and br_project_group(s) host to host.   python3 profiles/small_batch_bench.py"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from bramble_amd import device as brdev  # noqa: E402
from bramble_amd import lib, synth  # noqa: E402

ann = synth.Annotation("G")
idx = lib.Index.from_flat(ann.flat, device=0)
cfg = lib.make_config()
out = {}
for pairs in (1, 32, 1000, 5000, 30000, 52000):
    batch = ann.reads(pairs, "pe", seed=1234 + pairs)
    db = brdev.upload_batch(batch, "cuda:0")
    for small in (1, 0):
        ctx = lib.Context(idx)
        ctx.set_param("small_batch", small)
        for _ in range(20):
            ctx.project_batch_device(cfg, db, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 300
        for _ in range(k):
            ctx.project_batch_device(cfg, db, 0)
        torch.cuda.synchronize()
        out["pairs=%d small_batch=%d" % (pairs, small)] = round((time.perf_counter() - t0) / k * 1e6, 1)
        ctx.close()
# the AoS entry, host to host: one pair per call, and 64 pairs per call
ctx = lib.Context(idx)
batch = ann.reads(64, "pe", seed=99)
from tests.test_gpu_group import _group_alignments  # noqa: E402
_, goff = lib.prepare_batch(batch)
groups = [_group_alignments(batch, int(goff[g]), int(goff[g + 1])) for g in range(len(goff) - 1)]
for _ in range(5):
    ctx.project_group(cfg, groups[0])
t0 = time.perf_counter()
k = 200
for i in range(k):
    ctx.project_group_raw(cfg, groups[i % len(groups)]) if hasattr(ctx, "project_group_raw") else ctx.project_group(cfg, groups[i % len(groups)])
out["br_project_group, one pair, us per call (incl. ctypes marshalling)"] = round((time.perf_counter() - t0) / k * 1e6, 1)
allg = [a for g in groups for a in g]
for _ in range(3):
    ctx.project_groups(cfg, allg)
t0 = time.perf_counter()
for i in range(50):
    ctx.project_groups(cfg, allg)
out["br_project_groups, 64 pairs, us per call (incl. ctypes marshalling)"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
print(json.dumps(out, indent=1))
