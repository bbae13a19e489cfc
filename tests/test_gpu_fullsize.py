"""BASELINE.json's full size (configs[1]: 10M read pairs vs the GENCODE-shaped annotation) through
size-independent properties -- the oracle cannot run 20M alignments in test time:

  * every rewritten CIGAR consumes exactly the read's l_qseq query bases (merge preserves the query);
  * NH of a read name = number of records emitted for it, HI runs 1..NH inside the name;
  * idempotence: projecting the same batch twice gives the same checksums;
  * shard additivity ("checksum of checksums"): the two halves of the batch, cut at a read-name boundary,
    give exactly the rows of the whole batch (counts and per-column checksums add up);
  * with the paired flag cleared every match is emitted: n_rows == n_matches.
"""
import numpy as np
import pytest
import torch

from bramble_amd import device as brdev
from bramble_amd import lib, shard, synth

pytestmark = pytest.mark.gpu

PAIRS = 10_000_000


def _checksums(t):
    cs = {"n_rows": t["n_rows"]}
    for k in ("transcript_id", "pos", "nh", "hi", "mapq", "junc_hits", "aligned_len", "insert_size", "mate_pos"):
        cs[k] = int(t[k].to(torch.int64).sum().item())
    cs["strand"] = int(t["strand"].to(torch.int64).sum().item())
    w = t["cigar"].to(torch.int64)
    cs["cigar"] = int(((w & 0xFFFFFFFF) * 1315423911 % 2147483647).sum().item())
    return cs


@pytest.fixture(scope="module")
def setup():
    ann = synth.Annotation("G")
    idx = lib.Index.from_flat(ann.flat, device=0)
    ctx = lib.Context(idx)
    batch = ann.reads(PAIRS, "pe")
    return ann, idx, ctx, batch


def _check_row_properties(t, batch):
    """size-independent properties of one projected batch (t = the wide device view of its rows)"""
    n = t["n_rows"]
    # query length of every rewritten CIGAR == l_qseq of its input record
    w = t["cigar"].to(torch.int64) & 0xFFFFFFFF
    op = w & 0xF
    ln = w >> 4
    consumes = (op == 0) | (op == 1) | (op == 4) | (op == 7) | (op == 8)
    qcum = torch.cumsum(torch.where(consumes, ln, torch.zeros_like(ln)), 0)
    qcum = torch.cat([torch.zeros(1, dtype=torch.int64, device=qcum.device), qcum])
    off = t["cigar_off"]
    qlen = qcum[off[1:]] - qcum[off[:-1]]
    lq = torch.from_numpy(batch["l_qseq"].astype(np.int64)).cuda()
    assert bool((qlen == lq[t["input_index"].to(torch.int64)]).all())

    # NH = records per read name; HI = 1..NH
    g = t["group"].to(torch.int64)
    ng = int(g.max().item()) + 1
    per_group = torch.bincount(g, minlength=ng)
    assert bool((t["nh"].to(torch.int64) == per_group[g]).all())
    first_row = torch.cumsum(per_group, 0) - per_group
    assert bool((t["hi"].to(torch.int64) == torch.arange(n, device=g.device) - first_row[g] + 1).all())
    assert bool((g[1:] >= g[:-1]).all())
    # a name with exactly one record is what total_unique counts (src/core.cpp:314-315); names with none are dropped
    return int((per_group == 1).sum().item()), int((per_group > 0).sum().item())


def test_full_size_properties(setup):
    ann, idx, ctx, batch = setup
    cfg = lib.make_config()
    db = brdev.upload_batch(batch, "cuda:0")
    rows = ctx.project_batch_device(cfg, db, 0)
    t = brdev.rows_as_tensors(ctx)
    n = t["n_rows"]
    assert n > 5 * PAIRS and rows.total_processed == batch["n_aln"]
    _check_row_properties(t, batch)

    cs1 = _checksums(t)
    # idempotence
    rows2 = ctx.project_batch_device(cfg, db, 0)
    assert _checksums(brdev.rows_as_tensors(ctx)) == cs1
    del t, rows, rows2

    # shard additivity
    starts = shard.group_starts(batch) if batch["n_aln"] < 2_000_000 else None
    if starts is None:  # names are "r<i>", mates adjacent: cheap group starts
        noff = batch["name_off"].astype(np.int64)
        ln_ = np.diff(noff)
        names = batch["names"]
        same = np.zeros(batch["n_aln"], dtype=bool)
        cand = np.nonzero(ln_[1:] == ln_[:-1])[0] + 1
        # compare fixed-width chunks only where lengths match
        for L in np.unique(ln_[cand]):
            idxs = cand[ln_[cand] == L]
            a = names[(noff[idxs][:, None] + np.arange(L)[None, :])]
            b = names[(noff[idxs - 1][:, None] + np.arange(L)[None, :])]
            same[idxs] = (a == b).all(axis=1)
        starts = np.concatenate([np.nonzero(~same)[0], [batch["n_aln"]]]).astype(np.int64)
    total = {}
    for r in range(2):
        sub, lo = shard.shard_batch(batch, r, 2, starts=starts)
        rs = ctx.project_batch_device(cfg, brdev.upload_batch(sub, "cuda:0"), 0)
        cs = _checksums(brdev.rows_as_tensors(ctx))
        for k, v in cs.items():
            total[k] = total.get(k, 0) + v
    assert total == cs1


def test_unpaired_emits_every_match(setup):
    ann, idx, ctx, batch = setup
    b2 = dict(batch)
    b2["flags"] = (batch["flags"] & ~np.uint16(0x1)).astype(np.uint16)
    rows = ctx.project_batch_device(lib.make_config(), brdev.upload_batch(b2, "cuda:0"), 0)
    assert rows.n_rows == rows.n_matches and rows.n_matches > 5 * PAIRS


def test_full_size_bam_bundle_stream_properties():
    """The records-in / records-out path at full size (20.9 M raw records -> 108.8 M projected records, 24 GB):
      * the output is a well-formed chain: every record's block_size equals the distance to the next row offset;
      * the fixed fields written into the stream equal the row table (refID = transcript id, pos, mapq, flag bits);
      * l_seq of every output record equals its input record's, and the rewritten n_cigar_op equals the row's;
      * the bytes of the first and of the last 3000 read names' records equal the oracle's (the tail lies beyond 20 GB);
      * the rows equal those of the flat-batch path on the same alignments (checksums);
      * BGZF deflate on the device: the first and the last blocks and a sample in between inflate (zlib) to the
        stream bytes they cover, with correct CRC32 / ISIZE, and all block sizes chain up to the compressed length."""
    import struct
    import zlib
    ann = synth.Annotation("G")
    idx = lib.Index.from_flat(ann.flat, device=0)
    ctx = lib.Context(idx)
    batch = ann.reads(PAIRS, "pe", with_records=1)
    stream_h, roff, rlen = synth.Annotation.frame_records(batch)
    cfg = lib.make_config()
    blob = torch.from_numpy(stream_h).cuda()
    off_d = torch.from_numpy(roff.view(np.int64)).cuda()
    len_d = torch.from_numpy(rlen.view(np.int32)).cuda()
    rows, bam = ctx.project_bam_device(cfg, blob, off_d, len_d, np.arange(ann.flat["n_refs"], dtype=np.int32), 0)
    t = brdev.rows_as_tensors(ctx)
    n = t["n_rows"]
    assert n > 5 * PAIRS and int(bam.n_rows) == n
    out = torch.as_tensor(brdev._DevArray(bam.data, int(bam.n_bytes), "|u1"), device="cuda:0")
    row_off = torch.as_tensor(brdev._DevArray(bam.row_off, n + 1, "<i8"), device="cuda:0")
    assert int(row_off[-1].item()) == int(bam.n_bytes) and int(row_off[0].item()) == 0

    def u32_at(base, rel):
        idx8 = base + rel
        v = out[idx8].to(torch.int64) | (out[idx8 + 1].to(torch.int64) << 8) | (out[idx8 + 2].to(torch.int64) << 16) | \
            (out[idx8 + 3].to(torch.int64) << 24)
        return v
    starts = row_off[:-1]
    assert bool((u32_at(starts, 0) == (row_off[1:] - starts - 4)).all())            # block_size chain
    assert bool((u32_at(starts, 4) == t["transcript_id"].to(torch.int64)).all())    # refID
    assert bool((u32_at(starts, 8) == t["pos"].to(torch.int64)).all())              # pos
    w3 = u32_at(starts, 12)
    assert bool((((w3 >> 8) & 0xff) == (t["mapq"].to(torch.int64) & 0xff)).all())
    w4 = u32_at(starts, 16)
    ncig_rows = (t["cigar_off"][1:] - t["cigar_off"][:-1])
    assert bool(((w4 & 0xffff) == ncig_rows).all())
    flag = w4 >> 16
    assert bool((((flag & 0x100) == 0) == (t["is_primary"] != 0)).all())
    assert bool((((flag & 0x1) != 0) == (t["is_paired"] != 0)).all())
    lq_in = torch.from_numpy(batch["l_qseq"].astype(np.int64)).cuda()
    assert bool((u32_at(starts, 20) == lq_in[t["input_index"].to(torch.int64)]).all())
    cs_bundle = _checksums(t)
    del w3, w4, flag

    # bytes: the records of the first and of the last 3000 read names equal the oracle's stream of those records (read
    # names are independent of each other; the tail of the stream lies beyond 20 GB: 64-bit offsets everywhere)
    from oracle import oracle_binding as ob
    gs = _name_group_starts(batch)
    oi = ob.OracleIndex(ann.as_dict())
    ref_map = np.arange(ann.flat["n_refs"], dtype=np.int32)
    n_aln = int(batch["n_aln"])
    for a0, a1 in ((0, int(gs[3000])), (int(gs[-3001]), n_aln)):
        orc = ob.run_bam(oi, ob.make_flags(), stream_h, roff[a0:a1], rlen[a0:a1], ref_map)[0]
        exp = orc["bam_stream"]
        assert orc["n_rows"] > 10000
        got = (out[:len(exp)] if a0 == 0 else out[int(bam.n_bytes) - len(exp):]).cpu().numpy()
        assert np.array_equal(got, exp), "BAM stream differs from the oracle's in the %s" % ("head" if a0 == 0 else "tail")
    assert int(bam.n_bytes) > 20 * 1000 ** 3

    # the flat-batch path on the same alignments gives the same rows
    db = brdev.upload_batch(batch, "cuda:0")
    rows2 = ctx.project_batch_device(cfg, db, 0)
    assert _checksums(brdev.rows_as_tensors(ctx)) == cs_bundle
    del db

    # device deflate of the whole stream; inflate a sample of blocks on the host
    rows, bam = ctx.project_bam_device(cfg, blob, off_d, len_d, np.arange(ann.flat["n_refs"], dtype=np.int32), 0)
    out = torch.as_tensor(brdev._DevArray(bam.data, int(bam.n_bytes), "|u1"), device="cuda:0")
    z = ctx.bgzf_deflate_device(out, 0)
    nblk = (int(bam.n_bytes) + 57343) // 57344
    assert z.numel() < 0.4 * int(bam.n_bytes)
    zh = z.cpu().numpy()
    p, b = 0, 0
    sample = set([0, 1, nblk - 1, nblk - 2] + list(range(7, nblk, max(nblk // 300, 1))))
    while p < len(zh):
        bsize = int(zh[p + 16]) | (int(zh[p + 17]) << 8)
        bsize += 1
        if b in sample:
            raw = zh[p:p + bsize].tobytes()
            assert raw[:4] == b"\x1f\x8b\x08\x04"
            crc, isize = struct.unpack_from("<II", raw, bsize - 8)
            data = zlib.decompressobj(-15).decompress(raw[18:bsize - 8])
            lo = b * 57344
            assert len(data) == isize == min(57344, int(bam.n_bytes) - lo)
            assert (zlib.crc32(data) & 0xffffffff) == crc
            assert data == out[lo:lo + isize].cpu().numpy().tobytes()
        p += bsize
        b += 1
    assert b == nblk and p == len(zh)
    ctx.close()
    idx.close()


# ---- configs[2] (1 M ONT-like reads, --lr -S) and configs[4] (5 M HiFi-like reads, --lr-hq --strict
# --similarity-threshold 0.95) at BASELINE.json's full sizes ----------------------------------------------------

def _name_group_starts(batch):
    """Alignment indices where a new read name starts (+ n_aln), vectorised (shard.group_starts is a Python loop)."""
    n = int(batch["n_aln"])
    noff = batch["name_off"].astype(np.int64)
    ln_ = np.diff(noff)
    names = batch["names"]
    same = np.zeros(n, dtype=bool)
    cand = np.nonzero(ln_[1:] == ln_[:-1])[0] + 1
    for L in np.unique(ln_[cand]):
        idxs = cand[ln_[cand] == L]
        a = names[(noff[idxs][:, None] + np.arange(L)[None, :])]
        b = names[(noff[idxs - 1][:, None] + np.arange(L)[None, :])]
        same[idxs] = (a == b).all(axis=1)
    return np.concatenate([np.nonzero(~same)[0], [n]]).astype(np.int64)


def _long_read_properties(ctx, cfg, batch, min_rows, sim_thr=None, need_rescues=False):
    """The size-independent properties of the module docstring for a long-read batch, plus: every emitted similarity
    score is that of a match that passed the filter (score > 0 <=> similarity > (double)threshold_f32, evaluate.cpp:
    843-865 gives passing matches x^2 * (junc_hits + 1) with x > 0), and rescued clips are present when asked for."""
    db = brdev.upload_batch(batch, "cuda:0")
    rows = ctx.project_batch_device(cfg, db, 0)
    t = brdev.rows_as_tensors(ctx)
    n = t["n_rows"]
    assert n > min_rows and rows.total_processed == batch["n_aln"]
    w = t["cigar"].to(torch.int64) & 0xFFFFFFFF
    op = w & 0xF
    ln = w >> 4
    assert bool((op <= 8).all())   # no private override op (10..13) survives the merge
    consumes = (op == 0) | (op == 1) | (op == 4) | (op == 7) | (op == 8)
    qcum = torch.cat([torch.zeros(1, dtype=torch.int64, device=w.device), torch.cumsum(torch.where(consumes, ln, torch.zeros_like(ln)), 0)])
    off = t["cigar_off"]
    lq = torch.from_numpy(batch["l_qseq"].astype(np.int64)).cuda()
    assert bool(((qcum[off[1:]] - qcum[off[:-1]]) == lq[t["input_index"].to(torch.int64)]).all())
    g = t["group"].to(torch.int64)
    per_group = torch.bincount(g, minlength=int(g.max().item()) + 1)
    assert bool((t["nh"].to(torch.int64) == per_group[g]).all())
    first_row = torch.cumsum(per_group, 0) - per_group
    assert bool((t["hi"].to(torch.int64) == torch.arange(n, device=g.device) - first_row[g] + 1).all())
    # long-read MAPQ (get_mapq, src/core.cpp:46-58): 3 when unique, 0 otherwise; one primary record per read name
    assert bool((t["mapq"].to(torch.int64) == torch.where(t["nh"].to(torch.int64) > 1, 0, 3)).all())
    assert int(t["is_primary"].to(torch.int64).sum().item()) == int((per_group > 0).sum().item())
    sim = t["similarity_score"]
    assert bool((sim > 0).all()) and bool(torch.isfinite(sim).all())
    if need_rescues:
        assert int((t["clip_score"] != 0).sum().item()) > n // 100
    cs1 = _checksums(t)
    cs1["sim"] = float(sim.sum().item())
    rows2 = ctx.project_batch_device(cfg, db, 0)
    t2 = brdev.rows_as_tensors(ctx)
    cs2 = _checksums(t2)
    cs2["sim"] = float(t2["similarity_score"].sum().item())
    assert cs2 == cs1   # idempotence, doubles included (same order of summation: bit-equal)
    del t, t2, rows, rows2, db, w, op, ln, qcum
    # shard additivity: the two halves, cut at a read-name boundary, give exactly the rows of the whole
    starts = _name_group_starts(batch)
    total = {}
    for r in range(2):
        sub, lo = shard.shard_batch(batch, r, 2, starts=starts)
        ctx.project_batch_device(cfg, brdev.upload_batch(sub, "cuda:0"), 0)
        ts = brdev.rows_as_tensors(ctx)
        cs = _checksums(ts)
        for k, v in cs.items():
            total[k] = total.get(k, 0) + v
        del ts
    cs1.pop("sim")
    assert total == cs1
    return starts


def _oracle_head(ann, batch, starts, flags, n_groups, with_seq):
    """The first n_groups read names against the oracle (cut at a name boundary)."""
    from oracle import oracle_binding as ob
    from tests.parity import assert_rows_equal
    cut = int(starts[n_groups])
    sub = {"n_aln": cut}
    for k in ("ref_id", "ref_start", "flags", "xs", "ts", "mate_ref_id", "mate_start", "l_qseq"):
        sub[k] = batch[k][:cut]
    for ok_, pk in (("cigar_off", "cigar"), ("name_off", "names")) + ((("seq_off", "seqs"),) if with_seq else ()):
        sub[ok_] = batch[ok_][:cut + 1]
        sub[pk] = batch[pk][:int(sub[ok_][-1])]
    if not with_seq:
        sub["seq_off"] = None
        sub["seqs"] = None
    return sub


def test_full_size_config2_ont_with_clip_rescue():
    """configs[2]: 1 M ONT-like reads, --lr -S (clip rescue: k_project_fa + k_ksw at full size)."""
    from oracle import oracle_binding as ob
    from tests.parity import assert_rows_equal
    ann = synth.Annotation("G", n_genes=6000, n_refs=5, with_genome=True)
    idx = lib.Index.from_flat(ann.flat, device=0)
    ctx = lib.Context(idx)
    batch = ann.reads(1_000_000, "ont", with_seq=1)
    flags = {"lr": 1, "use_fasta": 1}
    cfg = lib.make_config(**flags)
    starts = _long_read_properties(ctx, cfg, batch, 1_000_000, need_rescues=True)
    st = ctx.rescue_stats()
    assert st["problems"] > 100_000 and st["rescued"] > 10_000
    # oracle comparison on a 20 k-read head
    sub = _oracle_head(ann, batch, starts, flags, 20000, True)
    prod = ctx.project_batch(cfg, sub)
    annd = ann.as_dict()
    orc, _, _ = ob.run(ob.OracleIndex(annd), ob.make_flags(**flags), sub, want_matches=False)
    assert_rows_equal(prod, orc)
    ctx.close()
    idx.close()


def test_full_size_config4_hifi_similarity_filter():
    """configs[4] on one GPU: 5 M HiFi-like reads, --lr-hq --strict --similarity-threshold 0.95 vs the GENCODE-shaped
    annotation (on 8 GPUs the same batch is sharded by read name: the additivity property below is that sharding)."""
    from oracle import oracle_binding as ob
    from tests.parity import assert_rows_equal
    ann = synth.Annotation("G")
    idx = lib.Index.from_flat(ann.flat, device=0)
    ctx = lib.Context(idx)
    batch = ann.reads(5_000_000, "hifi")
    flags = {"lr_hq": 1, "strict": 1, "sim_thr": 0.95}
    cfg = lib.make_config(**flags)
    starts = _long_read_properties(ctx, cfg, batch, 5_000_000)
    sub = _oracle_head(ann, batch, starts, flags, 20000, False)
    prod = ctx.project_batch(cfg, sub)
    f = ann.flat
    oi = ob.OracleIndex.__new__(ob.OracleIndex)
    L = ob.lib()
    oi.h = L.orc_index_new()
    exs = np.stack([f["ex_start"], f["ex_end"]], axis=1).astype(np.uint32)
    off = f["tx_exon_off"].astype(np.int64)
    for tx in range(len(f["tx_ref"])):
        e = np.ascontiguousarray(exs[off[tx]:off[tx + 1]]).reshape(-1)
        L.orc_index_add_transcript(oi.h, int(f["tx_ref"][tx]), bytes([int(f["tx_strand"][tx])]), b"", e.ctypes.data, len(e) // 2, None, 0)
    L.orc_index_finish(oi.h)
    orc, _, _ = ob.run(oi, ob.make_flags(**flags), sub, want_matches=False)
    assert orc["n_rows"] > 10000
    assert_rows_equal(prod, orc)
    # every emitted score belongs to a match above the threshold: the oracle's rows are exactly the filtered set, and
    # the raw similarity behind a score x^2 (junc_hits + 1) is thr + x (1 - thr) > thr
    thr = float(np.float32(0.95))
    x = np.sqrt(prod["similarity_score"] / (prod["junc_hits"].astype(np.float64) + 1.0))
    assert (thr + x * (1.0 - thr) > thr).all()
    ctx.close()
    idx.close()


def test_config3_eight_shards_of_12_5m_pairs():
    """BASELINE.json configs[3]: 100 M paired-end short reads, input batch-sharded across 8 x MI355X.  The driver owns the
    8-GPU node; here the eight 12.5 M-pair shards -- each rank's own seeded batch, exactly what `bench.py --gpus 8
    --pairs 12500000` gives rank r -- go through the ONE card one after the other, every shard at full size: query-length
    conservation, NH = records per read name, HI = 1..NH, and the counters of src/bramble.cpp:729-736 add up over the
    shards the way the reference's workers add theirs (threads.cpp:114-162: no shared state but those sums)."""
    ann = synth.Annotation("G")
    idx = lib.Index.from_flat(ann.flat, device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config()
    pairs = 12_500_000
    tot = {"processed": 0, "complete": 0, "unique": 0, "dropped": 0, "groups": 0, "n_aln": 0, "rows": 0}
    per_shard_rows = []
    for rank in range(8):
        batch = ann.reads(pairs, "pe", seed=(synth.SEED ^ 0x51ED) + 7919 * rank)     # bench.py's shard of rank `rank`
        db = brdev.upload_batch(batch, "cuda:0")
        rows = ctx.project_batch_device(cfg, db, 0)
        t = brdev.rows_as_tensors(ctx)
        assert t["n_rows"] == rows.n_rows and rows.n_rows > 5 * pairs and rows.total_processed == batch["n_aln"]
        uniq, named = _check_row_properties(t, batch)
        n_groups = int(db["n_groups"]) if isinstance(db, dict) and "n_groups" in db else pairs
        assert rows.total_complete == rows.n_rows and rows.total_unique == uniq
        # dropped = names none of whose alignments matched anything; a name can also match and still emit nothing (mates.cpp:153)
        assert 0 < rows.dropped_reads <= n_groups - named
        tot["processed"] += rows.total_processed; tot["complete"] += rows.total_complete; tot["unique"] += rows.total_unique
        tot["dropped"] += rows.dropped_reads; tot["groups"] += n_groups; tot["n_aln"] += int(batch["n_aln"]); tot["rows"] += int(rows.n_rows)
        per_shard_rows.append(int(rows.n_rows))
        del t, rows, db, batch
        torch.cuda.empty_cache()
    assert tot["processed"] == tot["n_aln"] and tot["complete"] == tot["rows"] and tot["n_aln"] > 2 * 8 * pairs
    assert tot["unique"] + tot["dropped"] < tot["groups"]
    assert len(set(per_shard_rows)) == 8        # eight different shards, not one batch eight times
    ctx.close()
    idx.close()
