"""Shared loader/checker for tests/golden/hand_derived.json."""
import json
import os

from bramble_amd.batch import format_cigar, make_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    with open(os.path.join(ROOT, "tests", "golden", "hand_derived.json")) as f:
        return json.load(f)["cases"]


def annotation(case):
    ann = {"refnames": ["chr1"], "transcripts": [{"id": t["id"], "ref_id": 0, "strand": t["strand"], "exons": t["exons"]}
                                                  for t in case["transcripts"]]}
    if case.get("genome"):          # -S cases: chr1's sequence (1-based position p is genome[p - 1])
        ann["ref_seqs"] = {0: case["genome"]}
    return ann


def batch(case):
    return make_batch([dict(r, ref_id=0) for r in case["reads"]])


def check(case, rows, key_tid, key_refc=None):
    """rows: dict of arrays (oracle or product naming differs only in the transcript-id key)."""
    names = [t["id"] for t in case["transcripts"]]
    exp = case["rows"]
    assert rows["n_rows"] == len(exp), (case["id"], rows["n_rows"], len(exp))
    for k, e in enumerate(exp):
        tag = (case["id"], k)
        assert rows["input_index"][k] == e["read"], tag
        assert names[rows[key_tid][k]] == e["transcript"], tag
        assert rows["pos"][k] == e["pos"], tag
        assert chr(rows["strand"][k]) == e["strand"], tag
        c0, c1 = int(rows["cigar_off"][k]), int(rows["cigar_off"][k + 1])
        assert format_cigar(rows["cigar"][c0:c1]) == e["cigar"], tag
        for f in ("nh", "hi", "mapq", "junc_hits", "is_paired"):
            assert rows[f][k] == e[f], tag + (f,)
        if "clip_score" in e:
            assert rows["clip_score"][k] == e["clip_score"], tag + ("clip_score",)
        if e.get("is_paired"):
            same = rows.get("same_transcript_as_mate", rows.get("same_transcript"))
            mtid = rows.get("mate_transcript_id", rows.get("mate_tid"))
            isz = rows.get("insert_size", rows.get("isize"))
            assert same[k] == e["same_transcript"], tag
            assert names[mtid[k]] == e["mate_transcript"], tag
            assert rows["mate_pos"][k] == e["mate_pos"] and isz[k] == e["isize"], tag
        if "similarity" in e:
            assert abs(rows["similarity_score"][k] - e["similarity"]) <= 1e-6, tag
