import sys, time, torch
sys.path.insert(0, ".")
from bramble_amd import device as brdev
from bramble_amd import lib, synth
ann = synth.Annotation("G"); index = lib.Index.from_flat(ann.flat, device=0); ctx = lib.Context(index); cfg = lib.make_config()
b = ann.reads(10_000_000, "pe", seed=1); db = brdev.upload_batch(b, "cuda:0"); st = torch.cuda.current_stream().cuda_stream
db2 = dict(db); db2["names"] = None; db2["name_off"] = None
for d in (db, db2):
    for _ in range(3): ctx.project_batch_device(cfg, d, st)
for rep in range(3):
    for tag, d in (("with names", db), ("without names (no seeds, no primary draw)", db2)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ctx.project_batch_device(cfg, d, st)
        torch.cuda.synchronize(); print("%s: %.3f ms per step" % (tag, 1e2 * (time.perf_counter() - t0)), flush=True)
