"""Shared checker: product rows (HIP path, through the C ABI) against oracle rows.
Bit-exact for every integer / index / CIGAR field; similarity within 1e-6 as
BASELINE.json's north_star states (in practice the doubles are bit-identical:
both sides are built with -ffp-contract=off)."""
import numpy as np

SIM_TOL = 1e-6

_EXACT = [("input_index", "input_index"), ("transcript_id", "tid"), ("pos", "pos"), ("strand", "strand"),
          ("clip_score", "clip_score"), ("junc_hits", "junc_hits"), ("aligned_len", "ref_consumed"),
          ("nh", "nh"), ("hi", "hi"), ("mapq", "mapq"), ("is_paired", "is_paired"),
          ("same_transcript_as_mate", "same_transcript"), ("is_first", "is_first"),
          ("mate_transcript_id", "mate_tid"), ("mate_pos", "mate_pos"), ("insert_size", "isize"),
          ("group", "group")]


def assert_rows_equal(prod, orc, check_primary=True):
    assert prod["n_rows"] == orc["n_rows"], (prod["n_rows"], orc["n_rows"])
    for pk, ok in _EXACT:
        a, b = np.asarray(prod[pk]), np.asarray(orc[ok])
        if not np.array_equal(a.astype(np.int64), b.astype(np.int64)):
            bad = np.nonzero(a.astype(np.int64) != b.astype(np.int64))[0]
            raise AssertionError("%s differs at %d rows, first %d: %r vs %r" % (pk, len(bad), bad[0], a[bad[0]], b[bad[0]]))
    assert np.array_equal(prod["cigar_off"], orc["cigar_off"]), "cigar offsets differ"
    assert np.array_equal(prod["cigar"], orc["cigar"]), "rewritten CIGARs differ"
    if prod["n_rows"]:
        assert np.max(np.abs(prod["similarity_score"] - orc["similarity_score"])) <= SIM_TOL
    if check_primary:
        assert np.array_equal(prod["is_primary"], orc["primary"]), "primary flags differ"
    for k in ("total_complete", "total_unique", "dropped_reads", "total_processed"):
        assert prod[k] == orc[k], (k, prod[k], orc[k])
