"""BGZF layer of the command line (host only): round trips against Python's gzip / the test-side BGZF writer, on both
codecs (libdeflate bound at run time, zlib)."""
import ctypes as C
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from bramble_amd import lib
from tests import bamio


def _bind():
    L = lib.lib()
    L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
    L.br_bgzf_read_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.br_free_buffer.argtypes = [C.c_void_p]
    L.br_bgzf_codec.restype = C.c_char_p
    return L


def read_file(path, threads=4):
    L = _bind()
    p, n = C.c_void_p(), C.c_uint64()
    rc = L.br_bgzf_read_file(str(path).encode(), threads, C.byref(p), C.byref(n))
    if rc:
        raise lib.BrambleError("br_bgzf_read_file %d" % rc)
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n.value, 1),))[:n.value].copy()
    L.br_free_buffer(p)
    return out


def payload(n, seed=1):
    rng = np.random.RandomState(seed)
    # compressible text-like part, an incompressible part, a run
    a = rng.randint(65, 70, size=n // 2).astype(np.uint8)
    b = rng.randint(0, 256, size=n // 3).astype(np.uint8)
    c = np.zeros(n - len(a) - len(b), dtype=np.uint8)
    return np.concatenate([a, b, c])


@pytest.mark.parametrize("n", [0, 1, 0xff00 - 1, 0xff00, 0xff00 + 1, 1_000_003])
@pytest.mark.parametrize("level", [0, 1, 6])
def test_writer_output_is_valid_bgzf(tmp_path, n, level):
    L = _bind()
    data = payload(n)
    path = tmp_path / "w.bgzf"
    assert L.br_bgzf_write_file(str(path).encode(), data.ctypes.data, n, 4, level) == 0
    with gzip.open(path, "rb") as f:            # independent decoder
        assert f.read() == data.tobytes()
    sizes = bamio.bgzf_block_sizes(path)
    assert sizes[-1] == 28 and max(sizes) <= 65536 and len(sizes) == (n + 0xff00 - 1) // 0xff00 + 1
    assert np.array_equal(read_file(path), data)


def test_reader_accepts_foreign_block_sizes_and_detects_corruption(tmp_path):
    data = payload(300_000, seed=2)
    path = tmp_path / "r.bgzf"
    path.write_bytes(bamio.bgzf_compress(data.tobytes(), block=12345, level=9))
    assert np.array_equal(read_file(path, threads=3), data)
    raw = bytearray(path.read_bytes())
    raw[len(raw) // 2] ^= 0x5a
    bad = tmp_path / "bad.bgzf"
    bad.write_bytes(bytes(raw))
    with pytest.raises(lib.BrambleError):
        read_file(bad)
    with pytest.raises(lib.BrambleError):
        read_file(tmp_path / "missing.bgzf")


def test_zlib_codec_when_libdeflate_is_disabled(tmp_path):
    """The same round trip in a child process with BRAMBLE_AMD_NO_LIBDEFLATE=1 (the codec is chosen once per process)."""
    code = r'''
import sys, gzip, ctypes as C, numpy as np
sys.path.insert(0, %r)
from bramble_amd import lib
L = lib.lib()
L.br_bgzf_codec.restype = C.c_char_p
assert L.br_bgzf_codec() == b"zlib", L.br_bgzf_codec()
L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
d = (np.arange(200000) %% 251).astype(np.uint8)
assert L.br_bgzf_write_file(%r.encode(), d.ctypes.data, d.size, 2, 6) == 0
assert gzip.open(%r, "rb").read() == d.tobytes()
print("ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "z.bgzf"), str(tmp_path / "z.bgzf"))
    env = dict(os.environ, BRAMBLE_AMD_NO_LIBDEFLATE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def _zlib_bgzf(data, level=6):
    import struct
    import zlib
    out = []
    for p in range(0, len(data), 0xff00):
        piece = data[p:p + 0xff00]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = c.compress(piece) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload
                   + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
    return b"".join(out)


def test_scan_stops_at_a_partial_block_and_refuses_what_is_not_bgzf():
    data = np.random.RandomState(4).randint(65, 70, size=90000).astype(np.uint8).tobytes()
    raw = _zlib_bgzf(data, 6)
    n_full = len(lib.bgzf_scan(np.frombuffer(raw, dtype=np.uint8))[0])
    assert n_full == 2
    for cut in (1, 10, 17, 18, 30, 5000):
        blocks, consumed, total = lib.bgzf_scan(np.frombuffer(raw[:len(raw) - cut], dtype=np.uint8))
        assert len(blocks) == 1 and total == 0xff00 and raw[consumed:consumed + 4] == b"\x1f\x8b\x08\x04"
    blocks, consumed, total = lib.bgzf_scan(np.frombuffer(raw, dtype=np.uint8), cap=1)
    assert len(blocks) == 1 and total == 0xff00
    with pytest.raises(lib.BrambleError):
        lib.bgzf_scan(np.frombuffer(b"\x1f\x8b\x08\x00" + raw[4:], dtype=np.uint8))      # gzip without the extra field
    with pytest.raises(lib.BrambleError):
        lib.bgzf_scan(np.frombuffer(raw[:12] + b"XY" + raw[14:], dtype=np.uint8))         # no BC subfield


