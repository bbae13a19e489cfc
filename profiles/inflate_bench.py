import ctypes as C, time, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from bramble_amd import lib, synth
L = lib.lib()
L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
L.br_bgzf_read_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
L.br_free_buffer.argtypes = [C.c_void_p]
ann = synth.Annotation("G", n_genes=3000, n_refs=3)
b = ann.reads(3_000_000, "pe", with_records=1)
stream, _, _ = synth.Annotation.frame_records(b)
print("bytes", stream.size)
path = b"/tmp/inf_test.bgzf"
t0 = time.perf_counter(); assert L.br_bgzf_write_file(path, stream.ctypes.data, stream.size, 16, 6) == 0; print("write L6 16t %.2fs" % (time.perf_counter() - t0), os.path.getsize(path))
for th in (1, 4, 8, 16, 32):
    p, n = C.c_void_p(), C.c_uint64()
    t0 = time.perf_counter(); assert L.br_bgzf_read_file(path, th, C.byref(p), C.byref(n)) == 0; dt = time.perf_counter() - t0
    print("read %2d threads %.2fs  %.2f GB/s" % (th, dt, n.value / dt / 1e9)); L.br_free_buffer(p)
