"""Upper bound of what the far matches cost k_inflate: the library built with -DINFLATE_FAKE_FAR serves a match that
reaches behind the LDS window from the window all the same (wrong bytes: every such block fails its CRC and the call
returns an error, which this probe ignores -- only the time is of interest).  .ab/fakefar.so against .ab/base.so.
python3 profiles/inflate_far_probe.py [pairs]"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    from bramble_amd import lib
    raw = np.fromfile(sys.argv[2], dtype=np.uint8)
    blocks, consumed, total = lib.bgzf_scan(raw)
    idx = lib.Index({"refnames": ["chr1"], "transcripts": [{"id": "t", "ref_id": 0, "strand": "+", "exons": [[10, 50]]}]}, device=0)
    ctx = lib.Context(idx)
    src = torch.from_numpy(raw).to("cuda:0")
    L = lib.lib()
    L.br_bgzf_inflate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    out, n = C.c_void_p(), C.c_uint64()
    bl = np.ascontiguousarray(blocks, dtype=lib.BGZF_BLOCK)
    ts = []
    for k in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.br_bgzf_inflate_device(ctx.h, C.c_void_p(src.data_ptr()), src.numel(), C.c_void_p(bl.ctypes.data), len(bl), None, C.byref(out), C.byref(n))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("rc %d, %d blocks, %d bytes out: %.1f ms (best of the last three)" % (rc, len(bl), total, 1e3 * min(ts[1:])))
    sys.exit(0)

sys.path.insert(0, ROOT)
from bramble_amd import lib, synth  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ann = synth.Annotation("G")
batch = ann.reads(n, "pe", with_records=1)
stream_h, roff, rlen = synth.Annotation.frame_records(batch)
L = lib.lib()
L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
tmp = tempfile.mkdtemp(prefix="bramble_far_")
path = os.path.join(tmp, "records.bgzf")
assert L.br_bgzf_write_file(path.encode(), stream_h.ctypes.data, stream_h.size, 16, 6) == 0
so = os.path.join(ROOT, "bramble_amd", "libbramble_amd.so")
keep = so + ".keep"
shutil.copy(so, keep)
try:
    for v in ("base", "fakefar", "base", "fakefar"):
        shutil.copy(os.path.join(ROOT, ".ab", v + ".so"), so + ".new")
        os.replace(so + ".new", so)   # (a rename: this process keeps the file it has mapped)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", path], capture_output=True, text=True)
        print(v, r.stdout.strip() or r.stderr[-300:], flush=True)
finally:
    os.replace(keep, so)
    shutil.rmtree(tmp, ignore_errors=True)
