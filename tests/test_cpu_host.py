"""CPU-side checks of the product's host logic (no GPU, no compute calls):
the C-ABI library loads and exports every symbol include/bramble_amd.h declares,
presets resolve like the reference's, the host-only index reports the reference's
known lengths, and the mate index / name grouping match the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from bramble_amd import lib, synth
from bramble_amd.batch import annotation_from_gtf_like, make_batch
from oracle import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "bramble_amd.h")).read()
    declared = set(re.findall(r"\b(br_[a-z_0-9]+)\s*\(", header))
    assert declared == set(lib.EXPORTS)
    L = C.CDLL(lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name


def test_presets_match_reference(golden):
    for case in golden["presets"]:
        got = lib.resolve_config(lib.make_config(**case["flags"]))
        for k in ("max_clip", "max_junc_ins", "max_junc_gap", "max_error_exon", "ignore_small_exons",
                  "filter_by_similarity"):
            assert got[k] == case[k], (case["id"], k)
        assert got["similarity_threshold"] == float(np.float32(case["similarity_threshold"]))


def test_overrides_resolve_like_oracle():
    for kw in ({"max_clip": 7}, {"lr": 1, "max_junc_gap": 3, "sim_thr": 0.8}, {"lr_hq": 1, "max_error_exon": 0},
               {"strict": 1, "max_error_exon": 12}, {"lr": 1, "lr_hq": 1}):
        assert lib.resolve_config(lib.make_config(**kw)) == ob.resolve_config(ob.make_flags(**kw)), kw


def test_host_only_index_lengths(golden):
    fx = golden["index_lengths"]
    ann = annotation_from_gtf_like(fx["refnames"], fx["transcripts"])
    idx = lib.Index(ann, device=-1)
    assert idx.num_transcripts() == 2
    for tid, t in enumerate(fx["transcripts"]):
        assert idx.transcript_name(tid) == t["id"]
        assert idx.transcript_len(tid) == t["length"]
    assert idx.transcript_name(2) is None and idx.transcript_len(2) is None
    assert idx.num_intervals() == 3


def test_index_rejects_bad_annotation():
    bad = {"refnames": ["chr1"], "transcripts": [{"id": "t", "ref_id": 0, "strand": "+",
                                                   "exons": [[100, 200], [150, 300]]}]}
    with pytest.raises(lib.BrambleError):
        lib.Index(bad, device=-1)
    unknown = {"refnames": ["chr1"], "transcripts": [{"id": "t", "seqname": "chrX", "strand": "+",
                                                       "exons": [[100, 200]]}]}
    with pytest.raises(lib.BrambleError):
        lib.Index(unknown, device=-1)


def test_projection_without_device_fails_loudly(golden):
    fx = golden["index_lengths"]
    idx = lib.Index(annotation_from_gtf_like(fx["refnames"], fx["transcripts"]), device=-1)
    with pytest.raises(lib.BrambleError):
        lib.Context(idx)  # host-only index: no CPU fallback exists


@pytest.mark.parametrize("mode", ["se", "pe"])
def test_prepare_matches_oracle_mate_index(mode):
    ann = synth.Annotation("S")
    b = ann.reads(3000, mode)
    mate, goff = lib.prepare_batch(b)
    oi = ob.OracleIndex(ann.as_dict())
    rows, matches, _ = ob.run(oi, ob.make_flags(), b)
    assert np.array_equal(mate, matches["mate_idx"])
    # groups = runs of equal names
    names = [bytes(b["names"][int(b["name_off"][i]):int(b["name_off"][i + 1])]) for i in range(b["n_aln"])]
    starts = [0] + [i for i in range(1, len(names)) if names[i] != names[i - 1]] + [len(names)]
    assert goff.tolist() == starts


def test_prepare_pairing_rules():
    # same name, mates point at each other; a third record with the same start as read 0 replaces it
    recs = [
        {"name": "q", "ref_id": 0, "ref_start": 100, "cigar": "50M", "flags": 0x41, "mate_ref_id": 0, "mate_start": 300},
        {"name": "q", "ref_id": 0, "ref_start": 100, "cigar": "50M", "flags": 0x41, "mate_ref_id": 0, "mate_start": 300},
        {"name": "q", "ref_id": 0, "ref_start": 300, "cigar": "50M", "flags": 0x81, "mate_ref_id": 0, "mate_start": 100},
        {"name": "q", "ref_id": 0, "ref_start": 500, "cigar": "50M", "flags": 0x81, "mate_ref_id": 1, "mate_start": 100},
        {"name": "z", "ref_id": 0, "ref_start": 300, "cigar": "50M", "flags": 0x0},
    ]
    b = make_batch(recs)
    mate, goff = lib.prepare_batch(b)
    assert mate.tolist() == [-1, 2, 1, -1, -1]
    assert goff.tolist() == [0, 4, 5]


def test_primary_tie_break_matches_libstdcxx():
    """br_primary_pick (murmur hash + lazy mt19937_64 + Lemire, restated for host and device)
    against the oracle's call of the real std::hash / std::mt19937_64 / uniform_int_distribution."""
    rng = np.random.RandomState(11)
    L = lib.lib()
    for it in range(5000):
        ln = int(rng.randint(0, 40))
        name = bytes(rng.randint(33, 126, size=ln).astype(np.uint8))
        n = int(rng.randint(1, 9)) if it % 2 else int(rng.randint(1, 10 ** 6))
        assert L.br_primary_pick(name, len(name), n) == ob.primary_pick(name, n), (name, n)


def test_header_is_plain_c(tmp_path):
    """include/bramble_amd.h is the drop-in boundary: it must compile as C99 (and C++11) on its own, no torch / HIP types."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "bramble_amd.h"\nint main(void) { br_config c; br_config_short_read(&c); return (int)br_index_num_refs(0); }\n')
    inc = os.path.join(root, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-x", "c++", "-I", inc, "-c", str(src), "-o", str(tmp_path / "b.o")])
    code = "\n".join(l for l in open(os.path.join(inc, "bramble_amd.h")) if l.lstrip().startswith("#include"))
    assert code.split() == ["#include", "<stddef.h>", "#include", "<stdint.h>"]      # nothing but the two C headers
