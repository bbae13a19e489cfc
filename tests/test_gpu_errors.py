"""Error behaviour of the device entry points: codes, not aborts; and a context stays usable after an error."""
import numpy as np
import pytest

from bramble_amd import lib, synth
from bramble_amd.batch import make_batch

pytestmark = pytest.mark.gpu

ANN = {"refnames": ["chr1"], "transcripts": [{"id": "t1", "ref_id": 0, "strand": "+", "exons": [[100, 400]]}]}


def test_errors_leave_the_context_usable():
    idx = lib.Index(ANN, device=0)
    ctx = lib.Context(idx)
    good = make_batch([{"name": "r1", "ref_id": 0, "ref_start": 150, "cigar": "50M", "read_len": 50}])
    # -S without sequences in the batch / without sequences in the index
    with pytest.raises(lib.BrambleError):
        ctx.project_batch(lib.make_config(lr=1, use_fasta=1), good)
    # the Rust-only discount knob
    cfg = lib.make_config()
    cfg.junc_miss_discount = 0.25
    with pytest.raises(lib.BrambleError):
        ctx.project_batch(cfg, good)
    # unknown tuning key / value
    with pytest.raises(lib.BrambleError):
        ctx.set_param("no_such_key", 1)
    with pytest.raises(lib.BrambleError):
        ctx.set_param("group_lanes", 7)
    # still works
    rows = ctx.project_batch(lib.make_config(), good)
    assert rows["n_rows"] == 1 and rows["pos"][0] == 50
    ctx.close()
    idx.close()


def test_alignments_the_reference_would_reject_give_no_records():
    """Reference ids outside the annotation and negative ids (bramble-rs/src/api.rs:316-318): zero records, no error.
    A CIGAR without reference bases is NOT rejected by the C++ path: setupCoordinates closes a zero-length exon
    [pos, pos) after its `end++` (gclib/GSam.cpp:283-288), which still overlaps the guide exon -- the read comes out
    with its all-clip CIGAR (the oracle agrees; kept as the reference behaves)."""
    idx = lib.Index(ANN, device=0)
    ctx = lib.Context(idx)
    recs = [{"name": "a", "ref_id": -1, "ref_start": 150, "cigar": "50M", "read_len": 50},
            {"name": "b", "ref_id": 5, "ref_start": 150, "cigar": "50M", "read_len": 50},
            {"name": "c", "ref_id": 0, "ref_start": 150, "cigar": "50S", "read_len": 50},
            {"name": "d", "ref_id": 0, "ref_start": 150, "cigar": "20M", "read_len": 20}]
    rows = ctx.project_batch(lib.make_config(), make_batch(recs))
    assert rows["n_rows"] == 2 and list(rows["input_index"]) == [2, 3]
    assert rows["cigar"][0] == (50 << 4 | 4) and rows["pos"][0] == 50 and rows["pos"][1] == 50
    ctx.close()
    idx.close()


def test_bundle_with_wrong_record_offsets_is_rejected_on_the_host():
    idx = lib.Index(ANN, device=0)
    ctx = lib.Context(idx)
    blob = np.zeros(100, dtype=np.uint8)
    with pytest.raises(lib.BrambleError):     # the record would end past the blob
        ctx.project_bam_bundle(lib.make_config(), blob, np.array([4], np.uint64), np.array([400], np.uint32), np.array([0], np.int32))
    ctx.close()
    idx.close()
