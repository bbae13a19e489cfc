"""br_project_group: the C form of project_group_with (bramble-rs/src/api.rs:285-464) -- one query name in, one
br_projected per emitted record out.  Checked against the reference-held fixture (K9 / K10), against
br_project_batch on the same alignments, against the oracle, and from a plain C program compiled against
include/bramble_amd.h."""
import os
import subprocess

import numpy as np
import pytest

from bramble_amd import lib, synth
from bramble_amd.batch import annotation_from_gtf_like, format_cigar, make_batch, parse_cigar
from oracle import oracle_binding as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_fixture_through_project_group(golden):
    fx = golden["projection"]
    ann = annotation_from_gtf_like(fx["refnames"], fx["transcripts"])
    idx = lib.Index(ann, device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    for rd in fx["reads"]:
        res = ctx.project_group(cfg, [{"query_name": rd["name"], "ref_id": rd["ref_id"], "ref_start": rd["ref_start"],
                                       "cigar": parse_cigar(rd["cigar"]), "read_len": rd["read_len"]}])
        assert len(res) == len(rd["expect"]), rd["id"]
        for p, exp in zip(res, rd["expect"]):
            assert idx.transcript_name(p["transcript_id"]) == exp["transcript"]
            assert p["transcript_start"] == exp["pos"] and p["transcript_strand"] == exp["strand"]
            # api.rs:453 <- evaluate.rs:1062: an untagged read's inferred strand is '.', which differs from either
            # transcript strand -- the Rust API returns true for K9 / K10; an XS tag that agrees gives false
            assert p["is_reverse"] == 1
            assert (p["nh"], p["hi"], p["mapq"]) == (exp["nh"], exp["hi"], exp["mapq"])
            assert format_cigar(p["cigar"]) == exp["out"]
            assert p["aligned_len"] == 100 and p["query_aligned_len"] == 100
            assert p["transcript_end"] == p["transcript_start"] + p["aligned_len"] - 1
            assert p["is_primary"] == 1 and p["input_index"] == 0 and p["is_paired_out"] == 0
    ctx.close()
    idx.close()


def _group_alignments(b, lo, hi):
    alns = []
    name = bytes(b["names"][int(b["name_off"][lo]):int(b["name_off"][lo + 1])]).decode()
    for i in range(lo, hi):
        f = int(b["flags"][i])
        c0, c1 = int(b["cigar_off"][i]), int(b["cigar_off"][i + 1])
        a = {"query_name": name, "ref_id": int(b["ref_id"][i]), "ref_start": int(b["ref_start"][i]),
             "cigar": b["cigar"][c0:c1], "is_reverse": bool(f & 0x10), "is_paired": bool(f & 0x1),
             "is_first_in_pair": bool(f & 0x40), "mate_is_unmapped": bool(f & 0x8),
             "xs_strand": chr(b["xs"][i]) if b["xs"][i] else None, "ts_strand": chr(b["ts"][i]) if b["ts"][i] else None,
             "mate_ref_id": int(b["mate_ref_id"][i]), "mate_ref_start": int(b["mate_start"][i]),
             "read_len": int(b["l_qseq"][i])}
        if b.get("seq_off") is not None:
            s0, s1 = int(b["seq_off"][i]), int(b["seq_off"][i + 1])
            if s1 > s0:
                a["sequence"] = bytes(b["seqs"][s0:s1])
        alns.append(a)
    return alns


def _check_groups(ann, b, flags, max_groups):
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config(**flags)
    orc, _, _ = ob.run(ob.OracleIndex(ann.as_dict()), ob.make_flags(**flags), b, want_matches=False)
    wide = ctx.project_batch(cfg, b)
    _, goff = lib.prepare_batch(b)
    ogroup = np.asarray(orc["group"])
    checked = rows_seen = 0
    # the groups with the most alignments first (multi-mappers, pairs), then a spread of the rest
    sizes = np.diff(goff.astype(np.int64))
    order = np.argsort(-sizes, kind="stable")[:max_groups // 2].tolist() + list(range(0, len(sizes), max(1, len(sizes) // (max_groups // 2))))
    for g in order:
        lo, hi = int(goff[g]), int(goff[g + 1])
        alns = _group_alignments(b, lo, hi)
        res = ctx.project_group(cfg, alns)
        sel = np.nonzero(ogroup == g)[0]
        assert len(res) == len(sel), (g, len(res), len(sel))
        for p, r in zip(res, sel):
            c0, c1 = int(orc["cigar_off"][r]), int(orc["cigar_off"][r + 1])
            assert p["transcript_id"] == orc["tid"][r] and p["transcript_start"] == orc["pos"][r]
            assert p["transcript_strand"] == chr(orc["strand"][r])
            a_in = alns[p["input_index"]]
            rs = a_in["xs_strand"] or ((a_in["ts_strand"] if not a_in["is_reverse"] else {"+": "-", "-": "+"}[a_in["ts_strand"]])
                                       if a_in["ts_strand"] else ".")
            assert p["is_reverse"] == int(p["transcript_strand"] != rs)
            assert p["aligned_len"] == max(int(orc["ref_consumed"][r]), 0)
            assert np.array_equal(p["cigar"], orc["cigar"][c0:c1])
            assert (p["nh"], p["hi"], p["mapq"]) == (orc["nh"][r], orc["hi"][r], orc["mapq"][r])
            assert p["is_primary"] == orc["primary"][r] and p["is_paired_out"] == orc["is_paired"][r]
            assert p["same_transcript_as_mate"] == orc["same_transcript"][r] and p["insert_size"] == orc["isize"][r]
            assert p["input_index"] == orc["input_index"][r] - lo
            assert abs(p["similarity_score"] - orc["similarity_score"][r]) <= 1e-6
            q = sum(int(w) >> 4 for w in p["cigar"] if (int(w) & 0xF) in (0, 1, 7, 8, 10, 12))
            assert p["query_aligned_len"] == q
            # and the batch entry point agrees field by field
            assert wide["transcript_id"][r] == p["transcript_id"] and wide["pos"][r] == p["transcript_start"]
            assert wide["nh"][r] == p["nh"] and wide["hi"][r] == p["hi"] and wide["is_primary"][r] == p["is_primary"]
        checked += 1
        rows_seen += len(res)
    ctx.close()
    idx.close()
    return checked, rows_seen


def test_project_group_paired_multimappers_equal_batch_and_oracle():
    ann = synth.Annotation("G", n_genes=900, n_refs=2)
    b = ann.reads(1500, "pe", p_multimap=0.4)
    checked, rows_seen = _check_groups(ann, b, {}, 120)
    assert checked >= 100 and rows_seen > 300


def test_project_group_passes_the_sequence_to_the_clip_rescue():
    """use_fasta through the AoS entry point: GenomicAlignment::sequence (api.rs:91-95) must reach the -S rescue."""
    ann = synth.Annotation("G", n_genes=600, n_refs=2, with_genome=True)
    b = ann.reads(400, "ont", with_seq=1)
    flags = {"lr": 1, "use_fasta": 1}
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config(**flags)
    orc, _, _ = ob.run(ob.OracleIndex(ann.as_dict()), ob.make_flags(**flags), b, want_matches=False)
    assert int(np.asarray(orc["clip_score"]).max()) > 0, "the workload must contain accepted rescues"
    _, goff = lib.prepare_batch(b)
    ogroup = np.asarray(orc["group"])
    rescued = 0
    for g in range(len(goff) - 1):
        lo, hi = int(goff[g]), int(goff[g + 1])
        sel = np.nonzero(ogroup == g)[0]
        if not len(sel) or int(np.asarray(orc["clip_score"])[sel].max()) == 0:
            continue
        res = ctx.project_group(cfg, _group_alignments(b, lo, hi))
        assert len(res) == len(sel)
        for p, r in zip(res, sel):
            c0, c1 = int(orc["cigar_off"][r]), int(orc["cigar_off"][r + 1])
            assert p["transcript_id"] == orc["tid"][r] and p["transcript_start"] == orc["pos"][r]
            assert np.array_equal(p["cigar"], orc["cigar"][c0:c1])
            assert abs(p["similarity_score"] - orc["similarity_score"][r]) <= 1e-6
        rescued += 1
        if rescued >= 40:
            break
    assert rescued >= 10
    ctx.close()
    idx.close()


def test_project_groups_many_names_in_one_call_equal_the_per_group_calls():
    """br_project_groups: a run of name-collated groups in one call = the concatenation of the per-group calls (input_index
    counted over the whole call); alignments with ref_id < 0 are skipped (api.rs:316-318) without shifting input_index."""
    ann = synth.Annotation("G", n_genes=900, n_refs=2)
    b = ann.reads(600, "pe", p_multimap=0.3)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    _, goff = lib.prepare_batch(b)
    n_groups = min(len(goff) - 1, 200)
    many, per_group = [], []
    for g in range(n_groups):
        lo, hi = int(goff[g]), int(goff[g + 1])
        alns = _group_alignments(b, lo, hi)
        base = len(many)
        # an unplaced record in front of every third group
        if g % 3 == 0:
            many.append(dict(alns[0], ref_id=-1))
            base += 1
        many.extend(alns)
        for p in ctx.project_group(cfg, alns):
            q = dict(p)
            q["input_index"] += base
            per_group.append(q)
    res = ctx.project_groups(cfg, many)
    assert len(res) == len(per_group) and len(res) > 200
    for p, q in zip(res, per_group):
        for k in ("transcript_id", "transcript_start", "transcript_end", "aligned_len", "query_aligned_len", "is_reverse",
                  "transcript_strand", "nh", "hi", "is_primary", "same_transcript_as_mate", "is_paired_out", "insert_size",
                  "input_index", "mapq"):
            assert p[k] == q[k], k
        assert np.array_equal(p["cigar"], q["cigar"])
    # a call that holds only unplaced records returns nothing (api.rs:392-394)
    assert ctx.project_groups(cfg, [dict(many[1], ref_id=-1)]) == []
    ctx.close()
    idx.close()


def test_project_group_strand_tags_decide_is_reverse(golden):
    """is_reverse = transcript strand != inferred read strand (evaluate.rs:1062 with infer_strand, api.rs:470-489)."""
    fx = golden["projection"]
    ann = annotation_from_gtf_like(fx["refnames"], fx["transcripts"])
    idx = lib.Index(ann, device=0)
    ctx = lib.Context(idx)
    rd = fx["reads"][0]   # K9: tx1 on '+'
    base = {"query_name": rd["name"], "ref_id": rd["ref_id"], "ref_start": rd["ref_start"], "cigar": parse_cigar(rd["cigar"]),
            "read_len": rd["read_len"]}
    for extra, want_n, want_rev in (({}, 1, 1), ({"xs_strand": "+"}, 1, 0), ({"ts_strand": "+"}, 1, 0),
                                    ({"ts_strand": "-", "is_reverse": True}, 1, 0), ({"xs_strand": "-"}, 0, None)):
        res = ctx.project_group(lib.make_config(), [dict(base, **extra)])
        assert len(res) == want_n, extra
        if want_n:
            assert res[0]["transcript_strand"] == "+" and res[0]["is_reverse"] == want_rev, extra
    ctx.close()
    idx.close()


def test_project_group_refuses_mixed_query_names():
    ann = synth.Annotation("S")
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    a = {"query_name": "a", "ref_id": 0, "ref_start": 100, "cigar": parse_cigar("50M")}
    b = dict(a, query_name="b")
    with pytest.raises(lib.BrambleError, match="invalid argument"):
        ctx.project_group(lib.make_config(), [a, b])
    assert ctx.project_group(lib.make_config(), []) == []
    ctx.close()
    idx.close()


def test_plain_c_program_against_the_header(tmp_path, golden):
    exe = str(tmp_path / "group_main")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "group_main.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "bramble_amd"), "-lbramble_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "bramble_amd")])
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
    exp = {rd["name"]: rd["expect"][0] for rd in golden["projection"]["reads"]}
    assert out[0] == "unspliced tx1 50 149 100 100 + nh=1 hi=1 mapq=255 primary=1 cigar=" + exp["unspliced"]["out"]
    assert out[1] == "spliced tx2 51 150 100 100 + nh=1 hi=1 mapq=255 primary=1 cigar=" + exp["spliced"]["out"]
    assert out[2] == "mixed-names rc=-1"


def test_project_group_at_a_dense_locus_takes_the_ordinary_pipeline():
    """A read with more than 32 matches per alignment does not fit the small path's tables: the call falls back to the
    ordinary pipeline inside the same entry point (and fetches its rows instead of finding them in pinned memory).  Same
    records as br_project_batch and the oracle."""
    txs = []
    for t in range(150):
        ex = [[1000, 1200 + (t % 4)], [2000 + 5 * (t % 7), 2300], [3000, 3100 + t]]
        txs.append({"id": "iso%d" % t, "ref_id": 0, "strand": "+" if t % 3 else "-", "exons": ex[:2] if t % 5 == 0 else ex})
    ann = {"refnames": ["chr1"], "transcripts": txs}
    recs = [{"name": "dense", "ref_id": 0, "ref_start": 1050, "cigar": "80M"},
            {"name": "dense", "ref_id": 0, "ref_start": 1120, "cigar": "80M3S"}]
    b = make_batch(recs)
    idx = lib.Index(ann, device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    orc, _, _ = ob.run(ob.OracleIndex(ann), ob.make_flags(), b, want_matches=False)
    assert orc["n_rows"] > 100
    res = ctx.project_group(cfg, _group_alignments(b, 0, 2))
    assert len(res) == orc["n_rows"]
    for p, r in zip(res, range(orc["n_rows"])):
        c0, c1 = int(orc["cigar_off"][r]), int(orc["cigar_off"][r + 1])
        assert (p["transcript_id"], p["transcript_start"], p["nh"], p["hi"], p["input_index"]) == \
               (orc["tid"][r], orc["pos"][r], orc["nh"][r], orc["hi"][r], orc["input_index"][r])
        assert np.array_equal(p["cigar"], orc["cigar"][c0:c1]) and p["is_primary"] == orc["primary"][r]
    # and a small call right after it goes the short way again
    one = ctx.project_group(cfg, [_group_alignments(b, 0, 1)[0]])
    assert len(one) > 32
    ctx.close()
    idx.close()
