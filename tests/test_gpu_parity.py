"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs, and against the reference's known answers."""
import numpy as np
import pytest

from bramble_amd import lib, synth
from bramble_amd.batch import annotation_from_gtf_like, make_batch
from oracle import oracle_binding as ob
from tests.parity import assert_rows_equal

pytestmark = pytest.mark.gpu


def run_both(ann_dict, batch, group_lanes=64, **flags):
    idx = lib.Index(ann_dict, device=0)
    ctx = lib.Context(idx)
    ctx.set_param("group_lanes", group_lanes)
    prod = ctx.project_batch(lib.make_config(**flags), batch)
    oi = ob.OracleIndex(ann_dict)
    orc, _, _ = ob.run(oi, ob.make_flags(**flags), batch, want_matches=False)
    # batches of up to 65536 alignments take the path without host round trips (run_device_small) by default: the
    # ordinary pipeline must give the same rows on the same input
    if batch["n_aln"] <= 65536:
        ctx.set_param("small_batch", 0)
        assert_rows_equal(ctx.project_batch(lib.make_config(**flags), batch), orc)
    ctx.close()
    idx.close()
    return prod, orc


def test_reference_known_answers(golden):
    fx = golden["projection"]
    ann = annotation_from_gtf_like(fx["refnames"], fx["transcripts"])
    names = [t["id"] for t in ann["transcripts"]]
    idx = lib.Index(ann, device=0)
    ctx = lib.Context(idx)
    for rd in fx["reads"]:
        batch = make_batch([{"name": rd["name"], "ref_id": rd["ref_id"], "ref_start": rd["ref_start"],
                             "cigar": rd["cigar"], "read_len": rd["read_len"]}])
        rows = ctx.project_batch(lib.make_config(), batch)
        assert rows["n_rows"] == len(rd["expect"])
        for k, exp in enumerate(rd["expect"]):
            assert names[rows["transcript_id"][k]] == exp["transcript"]
            assert rows["pos"][k] == exp["pos"] and chr(rows["strand"][k]) == exp["strand"]
            assert rows["nh"][k] == exp["nh"] and rows["hi"][k] == exp["hi"] and rows["mapq"][k] == exp["mapq"]
            assert rows["junc_hits"][k] == exp["junc_hits"]
            c0, c1 = int(rows["cigar_off"][k]), int(rows["cigar_off"][k + 1])
            assert ob.format_cigar(rows["cigar"][c0:c1]) == exp["out"]


def test_hand_derived_cases():
    """The HIP path against the hand-derived expectations (independent of the oracle)."""
    from tests import hand_cases
    for case in hand_cases.load():
        idx = lib.Index(hand_cases.annotation(case), device=0)
        ctx = lib.Context(idx)
        rows = ctx.project_batch(lib.make_config(**case["flags"]), hand_cases.batch(case))
        hand_cases.check(case, rows, "transcript_id")
        ctx.close()
        idx.close()


@pytest.mark.parametrize("group_lanes", [64, 16])
@pytest.mark.parametrize("mode,flags", [("se", {}), ("pe", {}), ("pe", {"strict": 1}), ("pe", {"fr": 1}),
                                        ("pe", {"rf": 1}), ("pe", {"max_clip": 2, "max_junc_ins": 3, "max_junc_gap": 3})])
def test_small_annotation_short_reads(mode, flags, group_lanes):
    ann = synth.Annotation("S")
    b = ann.reads(10000, mode)
    prod, orc = run_both(ann.as_dict(), b, group_lanes=group_lanes, **flags)
    assert orc["n_rows"] > 1000
    assert_rows_equal(prod, orc)


def test_config1_10k_single_end(golden):
    """BASELINE.json configs[0]: 10k single-end short reads, 1 chr / ~100 transcripts."""
    ann = synth.Annotation("S")
    b = ann.reads(10000, "se")
    prod, orc = run_both(ann.as_dict(), b)
    assert_rows_equal(prod, orc)


def test_stranded_xs_tags():
    ann = synth.Annotation("S")
    b = ann.reads(5000, "pe", xs_tag=True)
    prod, orc = run_both(ann.as_dict(), b)
    assert_rows_equal(prod, orc)


def test_all_matches_visible_when_unpaired():
    """Clearing the paired flag emits every match of every alignment: compares the full
    evaluate-level table (tid, pos, rewritten CIGAR) rather than the pair-filtered rows."""
    ann = synth.Annotation("G", n_genes=3000, n_refs=3)
    b = ann.reads(20000, "pe")
    b["flags"] = (b["flags"] & ~np.uint16(0x1)).astype(np.uint16)
    prod, orc = run_both(ann.as_dict(), b)
    assert_rows_equal(prod, orc)


@pytest.mark.parametrize("group_lanes", [64, 32, 16, 8])
def test_gencode_shaped_paired(group_lanes):
    ann = synth.Annotation("G", n_genes=4000, n_refs=4)
    b = ann.reads(40000, "pe")
    prod, orc = run_both(ann.as_dict(), b, group_lanes=group_lanes)
    assert orc["n_rows"] > 100000
    assert_rows_equal(prod, orc)


@pytest.mark.parametrize("mode,flags", [("hifi", {"lr_hq": 1}), ("hifi", {"lr": 1}), ("ont", {"lr": 1}),
                                        ("hifi", {"lr_hq": 1, "strict": 1, "sim_thr": 0.95}),
                                        ("ont", {"lr": 1, "max_error_exon": 0}), ("pe", {"max_error_exon": 30})])
def test_long_read_presets_without_fasta(mode, flags):
    ann = synth.Annotation("G", n_genes=3000, n_refs=3)
    b = ann.reads(6000, mode)
    prod, orc = run_both(ann.as_dict(), b, **flags)
    assert orc["n_rows"] > 1000
    assert_rows_equal(prod, orc)


@pytest.mark.parametrize("flags", [{"lr": 1, "use_fasta": 1}, {"lr_hq": 1, "use_fasta": 1},
                                   {"lr": 1, "use_fasta": 1, "max_clip": 10, "sim_thr": 0.8}])
def test_soft_clip_rescue_with_genome(flags):
    """-S: ksw2 clip rescue on device (k_project_fa + k_ksw) against the oracle's scalar restatement.
    The ksw2 piece of the oracle is 'parity unpinned' (no reference vector exists for it)."""
    ann = synth.Annotation("G", n_genes=300, n_refs=2, with_genome=True)
    b = ann.reads(3000, "ont", with_seq=1)
    prod, orc = run_both(ann.as_dict(), b, **flags)
    assert (orc["clip_score"] != 0).sum() > 500      # rescues do happen
    assert_rows_equal(prod, orc)


def test_fasta_flag_is_inert_for_short_reads():
    ann = synth.Annotation("G", n_genes=300, n_refs=2, with_genome=True)
    b = ann.reads(3000, "pe")
    prod, orc = run_both(ann.as_dict(), b, use_fasta=1)
    assert_rows_equal(prod, orc)


@pytest.mark.parametrize("mode,flags,kw", [("pe", {}, {"xs_tag": True}), ("pe", {}, {}), ("se", {"strict": 1}, {}),
                                           ("hifi", {"lr_hq": 1}, {}), ("ont", {"lr": 1}, {})])
def test_bam_reencode_matches_oracle(mode, flags, kw):
    """k_bam_scan / k_bam_size / k_bam_tasks against the oracle's restatement of write_to_bam
    (update_cigar, NH/HI/AS tags, XS/ts deletion, reverse_complement_bam, set_mate_info): the whole
    uncompressed BAM record stream must be byte-identical."""
    import torch
    from bramble_amd import device as brdev
    ann = synth.Annotation("G", n_genes=1500, n_refs=3)
    b = ann.reads(8000, mode, with_records=1, **kw)
    idx = lib.Index(ann.as_dict(), device=0)
    ctx = lib.Context(idx)
    cfg = lib.make_config(**flags)
    db = brdev.upload_batch(b, "cuda:0")
    rows = ctx.project_batch_device(cfg, db, 0)
    blob, roff = brdev.upload_records(b, "cuda:0")
    bam = ctx.bam_encode_device(cfg, blob, roff, 0)
    got = brdev.bam_stream_to_host(bam)
    orc, _, _ = ob.run(ob.OracleIndex(ann.as_dict()), ob.make_flags(**flags), b, want_matches=False,
                       bam_records=(b["rec_blob"], b["rec_off"]))
    assert bam.n_rows == orc["n_rows"] and orc["n_rows"] > 5000
    exp = orc["bam_stream"]
    assert len(got) == len(exp)
    if not np.array_equal(got, exp):
        bad = int(np.nonzero(got != exp)[0][0])
        raise AssertionError("BAM streams differ first at byte %d of %d" % (bad, len(exp)))
    ctx.close()
    idx.close()


def test_dense_locus_more_than_64_candidates():
    """150 isoforms share an exon: reads there have > 64 candidate rows (the group kernel's
    two-sweep path) next to reads that take the dense per-match path."""
    txs = []
    for t in range(150):
        strand = "+" if t % 3 else "-"
        ex = [[1000, 1200 + (t % 4)], [2000 + 5 * (t % 7), 2300], [3000, 3100 + t]]
        if t % 5 == 0:
            ex = ex[:2]
        txs.append({"id": "iso%d" % t, "ref_id": 0, "strand": strand, "exons": ex})
    txs.append({"id": "solo", "ref_id": 0, "strand": "+", "exons": [[10000, 10500]]})
    ann = {"refnames": ["chr1"], "transcripts": txs}
    recs = []
    rng = np.random.RandomState(7)
    for i in range(400):
        kind = i % 4
        if kind == 0:
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": 1000 + int(rng.randint(0, 120)), "cigar": "80M"})
        elif kind == 1:
            st = 1100 + int(rng.randint(0, 50))
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": st,
                         "cigar": "%dM%dN60M" % (1200 - st, 2000 - 1200)})
        elif kind == 2:
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": 10000 + int(rng.randint(0, 400)), "cigar": "90M"})
        else:
            st = 2200 + int(rng.randint(0, 60))
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": st, "cigar": "%dM%dN50M" % (2300 - st, 700)})
    b = make_batch(recs)
    for g in (64, 8):
        prod, orc = run_both(ann, b, group_lanes=g)
        assert orc["nh"].max() > 64
        assert_rows_equal(prod, orc)


def test_dense_locus_with_clip_rescue():
    """-S at a locus where 150 isoforms share exons: long reads with soft clips there have > 64 candidate rows (the
    rescue's emit pass for them is k_project_fa<3>) next to reads with few candidates (work list + k_emit_dense<.., FA>);
    clips that are the neighbouring exon's bases, foreign clips, both sides."""
    rng = np.random.RandomState(19)
    genome = "".join("ACGT"[k] for k in rng.randint(0, 4, 12000))
    txs = []
    for t in range(150):
        strand = "+" if t % 3 else "-"
        ex = [[1000, 1200 + (t % 4)], [2000 + 5 * (t % 7), 2300], [3000, 3100 + t]]
        if t % 5 == 0:
            ex = ex[:2]
        txs.append({"id": "iso%d" % t, "ref_id": 0, "strand": strand, "exons": ex})
    txs.append({"id": "duo", "ref_id": 0, "strand": "+", "exons": [[9000, 9400], [10000, 10500]]})
    ann = {"refnames": ["chr1"], "transcripts": txs, "ref_seqs": {0: genome}}

    def sub(a, b_):     # 1-based half-open genome slice
        return genome[a - 1:b_ - 1]

    def foreign(n):
        return "".join("ACGT"[k] for k in rng.randint(0, 4, n))

    recs = []
    for i in range(240):
        kind = i % 6
        if kind == 0:      # second exon of the dense locus, left clip = the end of the first exon (rescuable for many isoforms)
            st, cl = 2000 + 5 * int(rng.randint(0, 7)), int(rng.randint(8, 60))
            seq = sub(1200 - cl, 1200) + sub(st, st + 90)
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": st, "cigar": "%dS90M" % cl, "seq": seq})
        elif kind == 1:    # first exon, right clip = the start of the second exon
            cl = int(rng.randint(8, 60))
            st = 1100 + int(rng.randint(0, 40))
            seq = sub(st, 1200) + sub(2000, 2000 + cl)
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": st, "cigar": "%dM%dS" % (1200 - st, cl), "seq": seq})
        elif kind == 2:    # foreign clip at the dense locus
            cl = int(rng.randint(8, 40))
            seq = foreign(cl) + sub(2035, 2035 + 80)
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": 2035, "cigar": "%dS80M" % cl, "seq": seq})
        elif kind == 3:    # spliced through the dense locus with a clip on either side
            cl, cr = int(rng.randint(5, 30)), int(rng.randint(5, 30))
            seq = foreign(cl) + sub(1150, 1200) + sub(2000, 2300) + sub(3000, 3040) + foreign(cr)
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": 1150, "cigar": "%dS50M800N300M700N40M%dS" % (cl, cr), "seq": seq})
        elif kind == 4:    # the two-exon transcript: few candidates, rescuable left clip
            cl = int(rng.randint(8, 80))
            seq = sub(9400 - cl, 9400) + sub(10000, 10100)
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": 10000, "cigar": "%dS100M" % cl, "seq": seq})
        else:              # ... and a rescuable right clip
            cl = int(rng.randint(8, 80))
            seq = sub(9300, 9400) + sub(10000, 10000 + cl)
            recs.append({"name": "r%d" % i, "ref_id": 0, "ref_start": 9300, "cigar": "100M%dS" % cl, "seq": seq})
    b = make_batch(recs)
    for flags in ({"lr": 1, "use_fasta": 1}, {"lr_hq": 1, "use_fasta": 1}):
        prod, orc = run_both(ann, b, group_lanes=8, **flags)
        assert orc["nh"].max() > 64 and (orc["clip_score"] != 0).sum() > 100
        assert_rows_equal(prod, orc)


def test_empty_and_degenerate_inputs():
    ann = synth.Annotation("S").as_dict()
    idx = lib.Index(ann, device=0)
    ctx = lib.Context(idx)
    empty = make_batch([])
    rows = ctx.project_batch(lib.make_config(), empty)
    assert rows["n_rows"] == 0 and rows["total_processed"] == 0
    recs = [
        {"name": "a", "ref_id": -1, "ref_start": 100, "cigar": "50M"},          # no reference
        {"name": "b", "ref_id": 7, "ref_start": 100, "cigar": "50M"},           # reference without a tree
        {"name": "c", "ref_id": 0, "ref_start": 1, "cigar": "20S"},             # no aligned base
        {"name": "d", "ref_id": 0, "ref_start": 2000000000, "cigar": "50M"},    # far beyond every exon
        {"name": "e", "ref_id": 0, "ref_start": 5000, "cigar": "10N50M"},       # leading intron (uLTRA quirk)
        {"name": "f", "ref_id": 0, "ref_start": 5000, "cigar": "20M100N5I100N20M"},  # insertion inside an intron
        {"name": "g", "ref_id": 0, "ref_start": 5000, "cigar": "3H5S40M2D10M5S3H"},
    ]
    b = make_batch(recs)
    prod = ctx.project_batch(lib.make_config(), b)
    orc, _, _ = ob.run(ob.OracleIndex(ann), ob.make_flags(), b, want_matches=False)
    assert_rows_equal(prod, orc)
    ctx.close()
    idx.close()


def test_one_context_across_presets_and_batch_sizes():
    """A context is reused for batches of different presets and sizes: the clip / similarity score columns are written by
    long-read runs and must read as zero again in the short-read runs that follow (they are zero-filled lazily)."""
    ann = synth.Annotation("G", n_genes=800, n_refs=3)
    annd = ann.as_dict()
    idx = lib.Index(annd, device=0)
    ctx = lib.Context(idx)
    oi = ob.OracleIndex(annd)
    plan = [("pe", {}, 1500), ("hifi", {"lr_hq": 1}, 2500), ("pe", {}, 6000), ("ont", {"lr": 1}, 1200), ("se", {"strict": 1}, 9000),
            ("pe", {"lr": 1}, 2000), ("pe", {}, 500)]
    for mode, flags, n in plan:
        b = ann.reads(n, mode, seed=n)
        prod = ctx.project_batch(lib.make_config(**flags), b)
        orc, _, _ = ob.run(oi, ob.make_flags(**flags), b, want_matches=False)
        assert_rows_equal(prod, orc)
        if not flags.get("lr") and not flags.get("lr_hq"):
            assert not prod["similarity_score"].any() and not prod["clip_score"].any()
    ctx.close()
    idx.close()


def test_bucket_table_ranges_at_bin_edges():
    """The count pass takes its candidate rows straight from the 1 kb bucket tables ([t_lo[bin(qstart)],
    t_hi[bin(qend) + 1]), exact search only above 64 rows).  Crafted loci against the oracle: exons that span many bins
    (the range must reach back to a row that starts far to the left), nested and abutting exons around multiples of
    1024, a locus with > 64 transcripts in one bin, reads in the last bin of a reference, beyond every exon, and on a
    reference without transcripts."""
    txs = []
    # a 60 kb exon with short exons nested inside its span (other transcripts), on both strands
    txs.append({"id": "long+", "ref_id": 0, "strand": "+", "exons": [[1000, 61000], [70000, 70200]]})
    txs.append({"id": "long-", "ref_id": 0, "strand": "-", "exons": [[500, 62000]]})
    for k in range(40):
        s = 1024 * (k + 2) - 3 * (k % 5)            # starts hugging bin edges
        txs.append({"id": "in%d" % k, "ref_id": 0, "strand": "+-"[k & 1], "exons": [[s, s + 90 + k], [s + 400, s + 520]]})
    # 90 isoforms sharing a first exon inside one bin (more candidates than the mask has bits), second exons differ
    for k in range(90):
        txs.append({"id": "iso%d" % k, "ref_id": 1, "strand": "+", "exons": [[5000, 5150], [5300 + 2 * k, 5400 + 2 * k]]})
    # last bin of reference 1 and an abutting pair across a bin edge
    txs.append({"id": "tail", "ref_id": 1, "strand": "-", "exons": [[204700, 204801]]})
    txs.append({"id": "abutA", "ref_id": 1, "strand": "+", "exons": [[8000, 8192]]})
    txs.append({"id": "abutB", "ref_id": 1, "strand": "+", "exons": [[8192, 8300]]})
    ann = {"refnames": ["r0", "r1", "empty"], "transcripts": txs}   # exons are 1-based half-open here
    reads = []

    def rd(ref, start, cigar, **kw):
        reads.append(dict(name="q%d" % len(reads), ref_id=ref, ref_start=start, cigar=cigar, read_len=100, **kw))
    for p in (1000, 1023, 1024, 1025, 2040, 30000, 30719, 30720, 60900, 60999, 61000, 61500, 69999, 70100):
        rd(0, p, "100M")
    for k in range(0, 40, 3):
        s = 1024 * (k + 2) - 3 * (k % 5)
        rd(0, s, "90M%dN10M" % (400 - 90))          # spliced onto the nested transcripts
        rd(0, s + 5, "50M")
    for p in (4990, 5000, 5049, 5100):
        rd(1, p, "100M")
    rd(1, 5050, "100M%dN50M" % 170)                 # junction into iso10's second exon
    for p in (204690, 204700, 204750, 204800, 300000):
        rd(1, p, "50M")
    for p in (8100, 8150, 8191, 8192, 8200):
        rd(1, p, "60M")
    rd(2, 100, "100M")
    rd(-1, 0, "100M")
    batch = make_batch(reads)
    for flags in ({}, {"strict": 1}, {"lr": 1}):
        for gl in (8, 64):
            prod, orc = run_both(ann, batch, group_lanes=gl, **flags)
            assert orc["n_rows"] > 100
            assert_rows_equal(prod, orc)


@pytest.mark.parametrize("key", ["count_split", "emit_split"])
def test_single_kernel_forms_of_the_split_passes(key):
    """count_split = 0 / emit_split = 0: the count pass as one kernel (exon walk inline), the emit work list in one
    launch -- the forms the long-read presets use; same rows as the oracle on short reads, and on long reads under a
    short-read preset (every alignment takes the second count kernel when the split is on)."""
    ann = synth.Annotation("G", n_genes=3000, n_refs=3)
    for mode, n in (("pe", 30000), ("hifi", 4000)):
        b = ann.reads(n, mode)
        idx = lib.Index(ann.as_dict(), device=0)
        oi = ob.OracleIndex(ann.as_dict())
        orc, _, _ = ob.run(oi, ob.make_flags(), b, want_matches=False)
        for v in (0, 1):
            ctx = lib.Context(idx)
            ctx.set_param(key, v)
            prod = ctx.project_batch(lib.make_config(), b)
            assert_rows_equal(prod, orc)
            ctx.close()
        idx.close()


def test_single_pass_variant_equals_oracle():
    """"single_pass" = 1: count and emit in one sweep (k_project1 places every alignment's matches from wave-owned pages,
    writes the simple class at once and lists the general class for k_emit_wl).  The variant lost its A/B (DESIGN
    section 10) and is off by default; it stays selectable, so it stays bit-exact: short-read presets, both group
    widths, a dense locus (> 64 candidate rows, > 32 survivors), small first-call capacities that overflow and retry."""
    cases = []
    s_ann = synth.Annotation("S")
    for mode, flags in (("se", {}), ("pe", {}), ("pe", {"strict": 1}), ("pe", {"fr": 1}), ("pe", {"max_clip": 2, "max_junc_ins": 3, "max_junc_gap": 3}),
                        ("pe", {"max_error_exon": 30})):
        cases.append((s_ann.as_dict(), s_ann.reads(10000, mode), flags))
    g_ann = synth.Annotation("G", n_genes=4000, n_refs=4)
    cases.append((g_ann.as_dict(), g_ann.reads(40000, "pe"), {}))
    cases.append((g_ann.as_dict(), g_ann.reads(3000, "hifi"), {}))      # long CIGARs under the short-read preset: every alignment walks
    b_un = g_ann.reads(20000, "pe")
    b_un["flags"] = (b_un["flags"] & ~np.uint16(0x1)).astype(np.uint16)  # every match emitted
    cases.append((g_ann.as_dict(), b_un, {}))
    txs = []
    for t in range(150):
        ex = [[1000, 1200 + (t % 4)], [2000 + 5 * (t % 7), 2300], [3000, 3100 + t]]
        txs.append({"id": "iso%d" % t, "ref_id": 0, "strand": "+" if t % 3 else "-", "exons": ex[:2] if t % 5 == 0 else ex})
    txs.append({"id": "solo", "ref_id": 0, "strand": "+", "exons": [[10000, 10500]]})
    recs = []
    rng = np.random.RandomState(11)
    for i in range(600):
        st = 1100 + int(rng.randint(0, 50))
        recs.append([{"name": "r%d" % i, "ref_id": 0, "ref_start": 1000 + int(rng.randint(0, 120)), "cigar": "80M"},
                     {"name": "r%d" % i, "ref_id": 0, "ref_start": st, "cigar": "%dM%dN60M" % (1200 - st, 800)},
                     {"name": "r%d" % i, "ref_id": 0, "ref_start": 10000 + int(rng.randint(0, 400)), "cigar": "3S87M"}][i % 3])
    cases.append(({"refnames": ["chr1"], "transcripts": txs}, make_batch(recs), {}))
    for ann, b, flags in cases:
        orc, _, _ = ob.run(ob.OracleIndex(ann), ob.make_flags(**flags), b, want_matches=False)
        idx = lib.Index(ann, device=0)
        for gl in (8, 16):
            ctx = lib.Context(idx)
            ctx.set_param("group_lanes", gl)
            ctx.set_param("single_pass", 1)
            for _ in range(2):   # the second call starts from the first one's high-water marks
                assert_rows_equal(ctx.project_batch(lib.make_config(**flags), b), orc)
            ctx.close()
        idx.close()


def test_large_batches_launched_from_the_last_calls_counts():
    """Batches beyond `small_n` alignments on a context that has projected before are launched without the three host round
    trips of the ordinary pipeline: tables as earlier calls left them, emit / row grids from the LAST call's counts scaled to the
    batch (+15 %), one check at the end, the ordinary pipeline again when a table or a grid fell short.  Same rows as the
    oracle whatever the order of batches: equal ones (prediction exact), a smaller one, one with far more matches per alignment
    than predicted (falls back and grows the tables), and the long-read preset in between (other kernels, other tables)."""
    ann = synth.Annotation("G", n_genes=4000, n_refs=4)
    annd = ann.as_dict()
    oi = ob.OracleIndex(annd)
    batches = {"a": ann.reads(40000, "pe"), "b": ann.reads(34000, "pe", seed=77), "c": ann.reads(45000, "pe", seed=78, p_multimap=0.5),
               "h": ann.reads(70000, "hifi", seed=79)}
    for k in ("a", "b", "c"):
        assert batches[k]["n_aln"] > 65536
    want = {}
    for k, flags in (("a", {}), ("b", {}), ("c", {}), ("h", {"lr_hq": 1})):
        want[k], _, _ = ob.run(oi, ob.make_flags(**flags), batches[k], want_matches=False)
    idx = lib.Index(annd, device=0)
    for spec in (1, 0):
        ctx = lib.Context(idx)
        ctx.set_param("speculate", spec)
        for k in ("a", "a", "b", "c", "a", "h", "h", "c", "b"):
            flags = {"lr_hq": 1} if k == "h" else {}
            assert_rows_equal(ctx.project_batch(lib.make_config(**flags), batches[k]), want[k])
        ctx.close()
    idx.close()
