"""Error behaviour of the C ABI on a box without a GPU: every entry point returns a BR_ERR_* code (none aborts,
INTEGRATION.md section 4) -- the Python binding turns them into BrambleError with the br_strerror text."""
import ctypes as C

import numpy as np
import pytest

from bramble_amd import lib

ANN = {"refnames": ["chr1"], "transcripts": [{"id": "t1", "ref_id": 0, "strand": "+", "exons": [[100, 200], [300, 400]]}]}


def test_unknown_seqname_and_overlapping_exons_are_annotation_errors():
    L = lib.lib()
    ex = (lib.BrExon * 2)(lib.BrExon(100, 200), lib.BrExon(300, 400))
    tx = (lib.BrTranscript * 1)(lib.BrTranscript(b"t1", b"chrMissing", b"+", ex, 2))
    refs = (C.c_char_p * 1)(b"chr1")
    h = C.c_void_p()
    rc = L.br_index_build(tx, 1, refs, 1, None, 0, -1, C.byref(h))     # bramble-rs/src/g2t.rs:442-444
    assert rc == -4 and L.br_strerror(rc)
    bad = {"refnames": ["chr1"], "transcripts": [{"id": "t", "ref_id": 0, "strand": "+", "exons": [[100, 250], [200, 300]]}]}
    with pytest.raises(lib.BrambleError, match="annotation|exon"):
        lib.Index(bad, device=-1)


def test_transcript_without_exonic_bases_is_an_annotation_error():
    """The reference writes no @SQ line for a transcript of length 0 (src/bramble.cpp:588-596) while its tids count every
    guide: header and records would disagree, so the index refuses such a transcript instead of numbering around it."""
    for exons in ([[100, 100]], []):
        bad = {"refnames": ["chr1"], "transcripts": [{"id": "ok", "ref_id": 0, "strand": "+", "exons": [[10, 20]]},
                                                     {"id": "empty", "ref_id": 0, "strand": "+", "exons": exons}]}
        with pytest.raises(lib.BrambleError, match="annotation"):
            lib.Index(bad, device=-1)


def test_host_only_index_has_accessors_but_no_context():
    idx = lib.Index(ANN, device=-1)
    assert idx.num_transcripts() == 1 and idx.transcript_len(0) == 200 and idx.transcript_name(0) == "t1"
    assert idx.transcript_len(7) is None and idx.transcript_name(7) is None   # out of range (g2t.rs:320-342 return None)
    with pytest.raises(lib.BrambleError, match="device"):
        lib.Context(idx)                                                  # there is no CPU fallback
    idx.close()


def test_config_resolution_rejects_the_rust_only_discount():
    cfg = lib.make_config()
    cfg.junc_miss_discount = 0.5                                          # bramble-rs/src/api.rs:197-205: no C++ counterpart
    with pytest.raises(lib.BrambleError):
        lib.resolve_config(cfg)


def test_null_arguments_are_invalid_arg_not_crashes():
    L = lib.lib()
    assert L.br_index_build(None, 1, None, 0, None, 0, -1, None) == -1
    assert L.br_ctx_new(None, None) == -1
    assert L.br_project_batch(None, None, None, None) == -1
    assert L.br_project_bam_bundle(None, None, None, None) == -1
    assert L.br_batch_prepare(None, None, None, None) == -1
    for code in (0, -1, -2, -3, -4, -5, -6, -99):
        assert L.br_strerror(code)


def test_bam_split_partial_and_malformed_records():
    def rec(name, body_extra=b""):
        nm = name + b"\0"
        body = (0).to_bytes(4, "little") + (10).to_bytes(4, "little") + bytes([len(nm), 30]) + (4680).to_bytes(2, "little") + \
            (1).to_bytes(2, "little") + (0).to_bytes(2, "little") + (4).to_bytes(4, "little") + (0xffffffff).to_bytes(4, "little") * 2 + \
            (0).to_bytes(4, "little") + nm + ((4 << 4) | 0).to_bytes(4, "little") + bytes([0x12, 0x48]) + bytes([30] * 4) + body_extra
        return len(body).to_bytes(4, "little") + body
    good = np.frombuffer(rec(b"a") + rec(b"bb", b"NHC\x01") + rec(b"ccc"), dtype=np.uint8)
    off, ln, un, used = lib.bam_split(good)
    assert len(off) == 3 and used == good.size and un == 0
    off2, ln2, _, used2 = lib.bam_split(good[:-3])                         # the last record is incomplete: left for the next call
    assert len(off2) == 2 and used2 == int(off[2]) - 4
    bad = good.copy()
    bad[int(off[1]) + 8] = 200                                             # l_read_name runs past the record
    with pytest.raises(lib.BrambleError):
        lib.bam_split(bad)
    tiny = good.copy()
    tiny[0:4] = np.frombuffer((5).to_bytes(4, "little"), dtype=np.uint8)   # block_size < 32
    with pytest.raises(lib.BrambleError):
        lib.bam_split(tiny)


def test_cli_setup_errors_after_the_reader_started_exit_instead_of_hanging(tmp_path):
    """A setup error that comes after the reader thread has started (the guides are loaded beside it) must end the run with
    a non-zero code: the command once drained the queue of the reader that was NOT running and waited for ever.
    Both readers, plus the default; no GPU is needed to get as far as the guide loader."""
    import os
    import subprocess
    from tests import bamio
    BIN = os.path.join(os.path.dirname(lib.__file__), "bin", "bramble")
    if not os.path.exists(BIN):
        pytest.skip("command line not built")
    recs = [bamio.bam_record(b"r%d" % i, 0, 100 + i, [(50 << 4) | 0], 50) for i in range(200)]
    in_bam = str(tmp_path / "in.bam")
    bamio.write_bam(in_bam, "@HD\tVN:1.6\n", [("chr1", 100000)], bamio.frame(recs).tobytes())
    for extra in ([], ["--host-reader"], ["--device-reader"]):
        r = subprocess.run([BIN, in_bam, "-G", str(tmp_path / "missing.gtf"), "-o", str(tmp_path / "o.bam")] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "annotation" in r.stderr, (extra, r.stderr)
