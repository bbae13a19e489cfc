"""k_inflate on a match-heavy stream: the projected records of N read pairs (5:1 under the device deflate), deflated on the
device and inflated again: python3 profiles/inflate_probe.py [pairs]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bramble_amd import device as brdev
from bramble_amd import lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ann = synth.Annotation("G")
batch = ann.reads(n, "pe", with_records=1)
stream_h, roff, rlen = synth.Annotation.frame_records(batch)
idx = lib.Index.from_flat(ann.flat, device=0)
ctx = lib.Context(idx)
blob = torch.from_numpy(stream_h).to("cuda:0")
off_d = torch.from_numpy(roff.view(np.int64)).to("cuda:0")
len_d = torch.from_numpy(rlen.view(np.int32)).to("cuda:0")
rows, bam = ctx.project_bam_device(lib.make_config(), blob, off_d, len_d, np.arange(ann.flat["n_refs"], dtype=np.int32), 0)
src = torch.as_tensor(brdev._DevArray(bam.data, int(bam.n_bytes), "|u1"), device="cuda:0").clone()
z = ctx.bgzf_deflate_device(src, 0).clone()
raw = z.cpu().numpy()
blocks, consumed, total = lib.bgzf_scan(raw)
assert consumed == raw.size and total == src.numel()
out = ctx.bgzf_inflate_device(z, blocks)
assert torch.equal(out, src)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = ctx.bgzf_inflate_device(z, blocks)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 3
print("projected records of %d pairs: %d bytes in %d blocks of %d compressed bytes (ratio %.2f): inflate %.1f ms = %.1f GB/s out" %
      (n, total, len(blocks), raw.size, total / raw.size, el * 1e3, total / el / 1e9))
