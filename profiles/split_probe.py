"""Time br_bam_split_device on the raw records of N read pairs (one call): python3 profiles/split_probe.py [pairs]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bramble_amd import lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
ann = synth.Annotation("G")
b = ann.reads(n, "pe", with_records=1)
stream, _, _ = synth.Annotation.frame_records(b)
idx = lib.Index.from_flat(ann.flat, device=0)
ctx = lib.Context(idx)
L = lib.lib()
L.br_bam_split_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(lib.BrDeviceRecords), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
d = torch.from_numpy(stream).to("cuda:0")
recs = lib.BrDeviceRecords(); un, used = C.c_int64(), C.c_uint64()
for k in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = L.br_bam_split_device(ctx.h, C.c_void_p(d.data_ptr()), stream.size, ann.flat["n_refs"], None, C.byref(recs), C.byref(un), C.byref(used))
    torch.cuda.synchronize()
    print("call %d: rc %d, %d records of %d bytes in %.2f ms" % (k, rc, recs.n_aln, stream.size, (time.perf_counter() - t0) * 1e3), flush=True)
