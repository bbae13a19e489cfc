"""The oracle against every known answer the reference's own tests hold
(fixture: tests/golden/reference_known_answers.json)."""
import numpy as np
import pytest

from bramble_amd.batch import annotation_from_gtf_like, make_batch
from oracle import oracle_binding as ob


def test_merge_cigar_known_answers(golden):
    for case in golden["merge_cigar"]:
        out = ob.merge_cigar(ob.parse_cigar(case["real"]), ob.parse_cigar(case["ideal"]))
        assert ob.format_cigar(out) == case["out"], case["id"]


def test_presets(golden):
    for case in golden["presets"]:
        got = ob.resolve_config(ob.make_flags(**case["flags"]))
        for k in ("max_clip", "max_junc_ins", "max_junc_gap", "max_error_exon", "ignore_small_exons",
                  "filter_by_similarity"):
            assert got[k] == case[k], (case["id"], k)
        # threshold is a float widened to double (include/evaluate.h:281)
        assert got["similarity_threshold"] == float(np.float32(case["similarity_threshold"])), case["id"]


def test_projection_fixture(golden):
    fx = golden["projection"]
    ann = annotation_from_gtf_like(fx["refnames"], fx["transcripts"])
    names = [t["id"] for t in ann["transcripts"]]
    idx = ob.OracleIndex(ann)
    for rd in fx["reads"]:
        segs = ob.segments(rd["ref_start"], ob.parse_cigar(rd["cigar"]))
        assert segs.tolist() == rd["segs"], rd["id"]
        batch = make_batch([{"name": rd["name"], "ref_id": rd["ref_id"], "ref_start": rd["ref_start"],
                             "cigar": rd["cigar"], "read_len": rd["read_len"]}])
        rows, matches, _ = ob.run(idx, ob.make_flags(), batch)
        assert rows["n_rows"] == len(rd["expect"]), rd["id"]
        for k, exp in enumerate(rd["expect"]):
            assert names[rows["tid"][k]] == exp["transcript"]
            assert rows["pos"][k] == exp["pos"]
            assert chr(rows["strand"][k]) == exp["strand"]
            assert rows["nh"][k] == exp["nh"] and rows["hi"][k] == exp["hi"] and rows["mapq"][k] == exp["mapq"]
            assert rows["junc_hits"][k] == exp["junc_hits"]
            c0, c1 = int(rows["cigar_off"][k]), int(rows["cigar_off"][k + 1])
            assert ob.format_cigar(rows["cigar"][c0:c1]) == exp["out"]
            i0, i1 = int(matches["ideal_off"][k]), int(matches["ideal_off"][k + 1])
            assert ob.format_cigar(matches["ideal"][i0:i1]) == exp["ideal"]


def test_index_lengths(golden):
    fx = golden["index_lengths"]
    ann = annotation_from_gtf_like(fx["refnames"], fx["transcripts"])
    idx = ob.OracleIndex(ann)
    assert idx.num_transcripts() == len(fx["transcripts"])
    for tid, t in enumerate(fx["transcripts"]):
        assert idx.transcript_len(tid) == t["length"]


def test_hand_derived_cases():
    """The oracle against expectations worked out by hand from the reference's rules
    (tests/golden/hand_derived.json): '-' strand, pairing, NH > 1, INS / GAP exons, quirks."""
    from tests import hand_cases
    for case in hand_cases.load():
        idx = ob.OracleIndex(hand_cases.annotation(case))
        rows, _, _ = ob.run(idx, ob.make_flags(**case["flags"]), hand_cases.batch(case), want_matches=False)
        hand_cases.check(case, rows, "tid")
