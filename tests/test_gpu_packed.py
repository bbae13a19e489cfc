"""Packed rows (ABI version 2) and the flat-batch staging path.

The packed table is what the row stage writes; the wide one-array-per-field view the other GPU tests compare
with the oracle is derived from it on the device.  Here the packed HOST rows are unpacked on the host exactly as
include/bramble_amd.h documents (HI / MAPQ / mate fields / insert size from the adjacent row) and must give the
oracle's rows; the input contract the device computes for a staged flat batch (read-name groups, mate index) must
equal br_batch_prepare's; and two batches in flight through the two staging slots must not disturb each other."""
import ctypes as C

import numpy as np
import pytest

from bramble_amd import lib, synth
from oracle import oracle_binding as ob
from tests.parity import assert_rows_equal

pytestmark = pytest.mark.gpu


def _oracle(ann, flags, batch):
    orc, _, _ = ob.run(ob.OracleIndex(ann.as_dict()), ob.make_flags(**flags), batch, want_matches=False)
    return orc


def _group_of_rows(batch, input_index):
    _, goff = lib.prepare_batch(batch)
    return (np.searchsorted(goff, np.asarray(input_index, dtype=np.int64), side="right") - 1).astype(np.uint32)


@pytest.mark.parametrize("mode,flags,kw", [
    ("pe", {}, {"p_multimap": 0.2}),
    ("se", {"fr": 1}, {}),
    ("hifi", {"lr_hq": 1, "sim_thr": 0.9}, {}),
    ("ont", {"lr": 1}, {}),
])
def test_packed_host_rows_unpack_to_the_oracle_rows(mode, flags, kw):
    ann = synth.Annotation("G", n_genes=1200, n_refs=3)
    b = ann.reads(6000, mode, **kw)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    ctx.set_param("host_detail", 1)
    cfg = lib.make_config(**flags)
    p = ctx.project_batch_packed(cfg, b)
    assert p["n_rows"] > 1000
    w = lib.unpack_host_rows(p, b["l_qseq"], long_reads=bool(flags.get("lr") or flags.get("lr_hq")))
    w["group"] = _group_of_rows(b, w["input_index"])
    assert_rows_equal(w, _oracle(ann, flags, b))
    # the documented derivations: HI = rank inside the read name's rows, NH = their count, row_off / BR_ROW_FIRST / mate_idx
    # give the input alignment
    ro = p["row_off"].astype(np.int64)
    lead = np.repeat(np.arange(p["n_aln"]), np.diff(ro))
    mate = p["mate_idx"][lead]
    derived_input = np.where(w["is_first"] == 1, lead, mate)
    assert np.array_equal(derived_input, w["input_index"])
    g = w["group"].astype(np.int64)
    first_row_of_group = np.r_[0, np.nonzero(np.diff(g))[0] + 1]
    counts = np.diff(np.r_[first_row_of_group, len(g)])
    assert np.array_equal(np.repeat(counts, counts), w["nh"])
    assert np.array_equal(np.arange(len(g)) - np.repeat(first_row_of_group, counts) + 1, w["hi"])
    # the same call without the detail array downloads 24 bytes per row and no x
    ctx.set_param("host_detail", 0)
    p2 = ctx.project_batch_packed(cfg, b)
    assert "x" not in p2 and np.array_equal(p2["a"], p["a"]) and np.array_equal(p2["cigar"], p["cigar"])
    ctx.close()
    idx.close()


def test_device_input_contract_equals_host_prepare():
    """Read-name groups and the mate index of process_pairs, computed on the device for a staged flat batch
    (k_soa_fields / k_group_off / k_mates*), against br_batch_prepare on the host -- with multi-mapping reads (several
    alignments per name, mates to be matched by position) and a name group beyond the one-lane limit (96)."""
    ann = synth.Annotation("G", n_genes=800, n_refs=2)
    b = ann.reads(5000, "pe", p_multimap=0.5)
    # one name group of 260 alignments: paired records whose mates are the record 130 places on
    n0 = int(b["n_aln"])
    big = 260
    extra = {k: np.zeros(big, dtype=b[k].dtype) for k in ("ref_id", "ref_start", "mate_ref_id", "mate_start", "l_qseq", "flags", "xs", "ts")}
    for k in range(big):
        extra["ref_id"][k] = 0
        extra["mate_ref_id"][k] = 0 if k % 7 else 1            # every seventh: mate on another reference (not eligible)
        extra["ref_start"][k] = 1000 + 10 * (k % 130) + (5 if k >= 130 else 0)
        extra["mate_start"][k] = 1000 + 10 * (k % 130) + (0 if k >= 130 else 5)
        extra["flags"][k] = 0x1 | (0x40 if k < 130 else 0x80)
        extra["l_qseq"][k] = 100
    for k, v in extra.items():
        b[k] = np.concatenate([b[k], v])
    b["cigar"] = np.concatenate([b["cigar"], np.full(big, (100 << 4), dtype=np.uint32)])
    b["cigar_off"] = np.concatenate([b["cigar_off"], b["cigar_off"][-1] + np.arange(1, big + 1, dtype=np.uint64)])
    nm = np.frombuffer(b"bigname", dtype=np.uint8)
    b["names"] = np.concatenate([b["names"], np.tile(nm, big)])
    b["name_off"] = np.concatenate([b["name_off"], b["name_off"][-1] + len(nm) * np.arange(1, big + 1, dtype=np.uint64)])
    b["n_aln"] = n0 + big
    b["src_tx"] = np.concatenate([b["src_tx"], np.zeros(big, dtype=b["src_tx"].dtype)])
    mate, goff = lib.prepare_batch(b)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    p = ctx.project_batch_packed(lib.make_config(), b)
    assert p["n_groups"] == len(goff) - 1
    assert np.array_equal(p["mate_idx"], mate)
    assert (mate[n0:] >= 0).sum() > 150
    ctx.close()
    idx.close()


def test_two_batches_in_flight_through_the_staging_slots():
    ann = synth.Annotation("G", n_genes=1000, n_refs=2)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    batches = [ann.reads(4000 + 500 * k, "pe", seed=100 + k) for k in range(4)]
    want = [ctx.project_batch_packed(cfg, b) for b in batches]
    L = lib.lib()
    keeps, structs = [], []
    for b in batches:
        keep = []
        structs.append(lib._batch_struct(b, keep))
        keeps.append(keep)
    got = []
    res = [lib.BrHostRows(), lib.BrHostRows()]
    lib.check(L.br_batch_stage(ctx.h, C.byref(structs[0]), 0), "stage")
    for k in range(len(batches)):
        if k + 1 < len(batches):
            lib.check(L.br_batch_stage(ctx.h, C.byref(structs[k + 1]), (k + 1) % 2), "stage")
        lib.check(L.br_project_staged(ctx.h, C.byref(cfg), k % 2, C.byref(res[k % 2])), "project")
        if k >= 1:
            lib.check(L.br_host_rows_wait(ctx.h, (k - 1) % 2), "wait")
            got.append(lib.host_rows_to_numpy(res[(k - 1) % 2]))
    lib.check(L.br_host_rows_wait(ctx.h, (len(batches) - 1) % 2), "wait")
    got.append(lib.host_rows_to_numpy(res[(len(batches) - 1) % 2]))
    for g, w in zip(got, want):
        assert g["n_rows"] == w["n_rows"] and g["n_rows"] > 0
        for key in ("a", "cigar", "pool", "row_off", "mate_idx"):
            assert np.array_equal(g[key], w[key]), key
    ctx.close()
    idx.close()


def test_device_detail_column_on_request():
    """br_device_rows.x is NULL after a projection call; br_device_rows_detail derives the br_row_x array from what the
    call left in HBM, and it agrees with the wide view (which is compared with the oracle everywhere else)."""
    import torch
    from bramble_amd import device as brdev
    ann = synth.Annotation("G", n_genes=800, n_refs=2)
    b = ann.reads(5000, "pe", p_multimap=0.2)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    db = brdev.upload_batch(b, "cuda:0")
    rows = ctx.project_batch_device(cfg, db, 0)
    assert not rows.x and rows.n_rows > 1000
    p = brdev.packed_as_tensors(rows, int(b["n_aln"]), ctx=ctx)
    w = brdev.rows_as_tensors(ctx)
    x = p["x"].cpu().numpy()
    assert np.array_equal(x[:, 0], w["input_index"].cpu().numpy())
    assert np.array_equal(x[:, 1], w["junc_hits"].cpu().numpy())
    assert np.array_equal(x[:, 2], w["aligned_len"].cpu().numpy())
    assert np.array_equal(x[:, 3], w["hi"].cpu().numpy())
    torch.cuda.synchronize()
    ctx.close()
    idx.close()
