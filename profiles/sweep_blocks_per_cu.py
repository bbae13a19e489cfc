import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from bramble_amd import lib, synth, device as brdev
ann = synth.Annotation("G")
idx = lib.Index.from_flat(ann.flat, device=0)
batch = ann.reads(10_000_000, "pe")
db = brdev.upload_batch(batch, "cuda:0")
cfg = lib.make_config()
st = torch.cuda.current_stream().cuda_stream
for bpc in (4, 6, 8, 10, 12, 16, 24):
    ctx = lib.Context(idx)
    ctx.set_param("blocks_per_cu", bpc)
    for _ in range(2): ctx.project_batch_device(cfg, db, st)
    ctx.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc = 0.0
    for _ in range(5):
        ctx.project_batch_device(cfg, db, st)
        acc += ctx.kernel_ms()["k_project<G,false>"][0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(bpc, "step %.2f ms  count %.3f ms" % (dt * 1e3, acc / 5), flush=True)
    ctx.close()
