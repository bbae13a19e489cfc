set -e
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C, struct
from bramble_amd import lib, synth
from tests import bamio
ann = synth.Annotation("G"); annd = ann.as_dict()
b = ann.reads(4000000, "pe", with_records=1)
stream, roff, rlen = synth.Annotation.frame_records(b)
bamio.write_gtf("/tmp/g.gtf", annd)
refs = [(r, 250_000_000) for r in annd["refnames"]]
text = ("@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)).encode()
hb = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
for name, ln in refs:
    nm = name.encode() + b"\0"; hb += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
whole = np.concatenate([np.frombuffer(bytes(hb), dtype=np.uint8), stream])
L = lib.lib(); L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
assert L.br_bgzf_write_file(b"/tmp/in.bam", whole.ctypes.data, whole.size, 16, 6) == 0
PY
for i in 1 2; do s=$(date +%s.%N); bramble_amd/bin/bramble /tmp/in.bam -G /tmp/g.gtf -o /tmp/out.bam -p 16 --device-deflate | grep -E "bundles|release|stage"; e=$(date +%s.%N); echo "outer wall $(echo "$e - $s" | bc) s"; done
