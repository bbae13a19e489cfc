"""The command line end to end on the GPU box (scope table rows f-1 / f-3 / f-4): BGZF/BAM in, GTF (and FASTA)
in, BAM out -- header layout of src/bramble.cpp:513-623 and a record stream byte-identical to the oracle's."""
import os
import struct
import subprocess

import numpy as np
import pytest

from bramble_amd import lib, synth
from oracle import oracle_binding as ob
from tests import bamio
from tests.test_gpu_bam_bundle import framed_stream

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bramble_amd", "bin", "bramble")


def remap_refs(stream, perm):
    """Rewrite refID / next_refID of every record: new = perm[old]."""
    s = bytearray(stream.tobytes())
    p = 0
    while p < len(s):
        bs = struct.unpack_from("<I", s, p)[0]
        for at in (p + 4, p + 4 + 20):
            v = struct.unpack_from("<i", s, at)[0]
            if v >= 0:
                struct.pack_into("<i", s, at, int(perm[v]))
        p += 4 + bs
    return np.frombuffer(bytes(s), dtype=np.uint8)


def run_cli(tmp_path, annd, stream_in_bam_ids, bam_refs, flags_cli, flags_cfg, in_header, fasta=None, bundle=2500):
    order = bamio.guide_order(annd)
    rng = np.random.RandomState(5)
    gtf = str(tmp_path / "guides.gtf")
    bamio.write_gtf(gtf, annd, order=rng.permutation(len(annd["transcripts"])))
    in_bam, out_bam = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    bamio.write_bam(in_bam, in_header, bam_refs, stream_in_bam_ids.tobytes(), block=40000)
    cmd = [BIN, in_bam, "-G", gtf, "-o", out_bam, "-p", "4", "--bundle-size", str(bundle), "--compression-level", "1"] + flags_cli
    if fasta:
        cmd += ["-S", fasta]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    text, refs, got = bamio.read_bam(out_bam)
    # oracle: transcripts in guide order (tid = @SQ position), records with annotation reference ids
    sorted_ann = dict(annd)
    sorted_ann["transcripts"] = [annd["transcripts"][t] for t in order]
    name_to_ann = {n: i for i, n in enumerate(annd["refnames"])}
    ref_map = np.array([name_to_ann.get(n, -1) for n, _ in bam_refs], dtype=np.int32)
    roff, rlen, n_un, used = lib.bam_split(stream_in_bam_ids)
    orc, _, _, _ = ob.run_bam(ob.OracleIndex(sorted_ann), ob.make_flags(**flags_cfg), stream_in_bam_ids, roff, rlen, ref_map)
    return text, refs, got, orc, sorted_ann, r.stdout, gtf, n_un, len(rlen)


def tx_len(tx):
    return sum(e - s for s, e in tx["exons"])


@pytest.mark.parametrize("reader", ["--host-reader", "--device-reader"])
def test_cli_short_reads_header_and_records(tmp_path, reader):
    ann = synth.Annotation("G", n_genes=1200, n_refs=5)
    annd = ann.as_dict()
    b = ann.reads(8000, "pe", with_records=1, xs_tag=True)
    stream = framed_stream(b, unmapped_every=101)
    # the BAM lists the references in another order and has one the annotation lacks
    perm = np.array([3, 0, 4, 1, 2])
    bam_refs = [None] * 6
    for old, new in enumerate(perm):
        bam_refs[new] = (annd["refnames"][old], 1000000)
    bam_refs[5] = ("chrUn_extra", 500)
    in_header = "@HD\tVN:1.6\tSO:unsorted\tGO:query\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in bam_refs) + \
        "@RG\tID:rg1\tSM:s\n@PG\tID:aligner\tPN:aligner\tVN:2.1\n@PG\tID:sorter\tPN:sorter\tPP:aligner\n@CO\tuser comment\n"
    text, refs, got, orc, sorted_ann, out, gtf, n_un, n_rec = run_cli(tmp_path, annd, remap_refs(stream, perm), bam_refs, [reader], {}, in_header)
    # header: @HD first, one @SQ per transcript in guide order, the other input lines, the chained @PG, the @CO
    lines = text.rstrip("\n").split("\n")
    assert lines[0] == "@HD\tVN:1.6\tSO:unsorted\tGO:query"
    ntx = len(sorted_ann["transcripts"])
    assert lines[1:1 + ntx] == ["@SQ\tSN:%s\tLN:%d" % (t["id"], tx_len(t)) for t in sorted_ann["transcripts"]]
    rest = lines[1 + ntx:]
    assert rest[:4] == ["@RG\tID:rg1\tSM:s", "@PG\tID:aligner\tPN:aligner\tVN:2.1", "@PG\tID:sorter\tPN:sorter\tPP:aligner", "@CO\tuser comment"]
    assert rest[4].startswith("@PG\tID:bramble\tPN:bramble\tPP:sorter\tVN:") and "\tCL:" in rest[4] and "-G" in rest[4]
    assert rest[5] == "@CO\tGenerated from GTF: " + gtf and len(rest) == 6
    assert refs == [(t["id"], tx_len(t)) for t in sorted_ann["transcripts"]]
    # records
    assert orc["n_rows"] > 5000
    assert len(got) == len(orc["bam_stream"]) and np.array_equal(got, orc["bam_stream"])
    # final report (src/bramble.cpp:727-736)
    assert "# input alignments:   %d" % (n_rec + n_un) in out and "# unmapped reads:     %d" % n_un in out
    assert "# total alignments:   %d" % orc["total_complete"] in out and "# unique alignments:  %d" % orc["total_unique"] in out
    assert "# dropped alignments: %d" % orc["dropped_reads"] in out


@pytest.mark.parametrize("reader", ["--host-reader", "--device-reader"])
def test_cli_long_reads_with_genome(tmp_path, reader):
    ann = synth.Annotation("G", n_genes=300, n_refs=2, with_genome=True)
    annd = ann.as_dict()
    b = ann.reads(3000, "ont", with_seq=1, with_records=1)
    stream = framed_stream(b)
    fasta = str(tmp_path / "genome.fa")
    with open(fasta, "w") as f:
        for rid, name in enumerate(annd["refnames"]):
            seq = annd["ref_seqs"][rid]
            if isinstance(seq, (bytes, bytearray)):
                seq = bytes(seq).decode()
            f.write(">%s some description\n" % name)
            for a in range(0, len(seq), 70):
                f.write(seq[a:a + 70] + "\n")
    bam_refs = [(n, len(annd["ref_seqs"][i])) for i, n in enumerate(annd["refnames"])]
    in_header = "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in bam_refs)   # no @HD, no @PG
    text, refs, got, orc, sorted_ann, out, gtf, _, _ = run_cli(tmp_path, annd, stream, bam_refs, ["--lr", reader], {"lr": 1, "use_fasta": 1},
                                                             in_header, fasta=fasta, bundle=700)
    lines = text.rstrip("\n").split("\n")
    assert lines[0].startswith("@SQ\tSN:") and lines[-2].startswith("@PG\tID:bramble\tPN:bramble\tVN:") and lines[-1].startswith("@CO\t")
    assert (orc["clip_score"] != 0).sum() > 300
    assert len(got) == len(orc["bam_stream"]) and np.array_equal(got, orc["bam_stream"])


def test_cli_result_does_not_depend_on_bundle_size(tmp_path):
    ann = synth.Annotation("G", n_genes=600, n_refs=3)
    annd = ann.as_dict()
    b = ann.reads(5000, "pe", with_records=1)
    stream = framed_stream(b)
    bam_refs = [(n, 1000000) for n in annd["refnames"]]
    gtf = str(tmp_path / "g.gtf")
    bamio.write_gtf(gtf, annd)
    in_bam = str(tmp_path / "in.bam")
    bamio.write_bam(in_bam, "@HD\tVN:1.6\n", bam_refs, stream.tobytes())
    outs = []
    for k, (bundle, extra) in enumerate(((1, ["--host-deflate", "--host-reader"]), (777, ["--compression-level", "1"]), (10 ** 7, []), (900, ["--device-deflate"]),
                                         (1200, ["--device-reader"]), (40, ["--device-reader", "--compression-level", "1"]), (10 ** 7, ["--device-reader"]))):
        out_bam = str(tmp_path / ("o%d.bam" % k))
        r = subprocess.run([BIN, in_bam, "-G", gtf, "-o", out_bam, "-p", "2", "--bundle-size", str(bundle), "--quiet",
                            "--strict", "--max-soft-clip", "3"] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(bamio.read_bam(out_bam)[2])
        sizes = bamio.bgzf_block_sizes(out_bam)
        assert max(sizes) <= 65536 and sizes[-1] == 28
    assert len(outs[0]) > 100000
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.array_equal(outs[0], outs[3])   # BGZF blocks made on the device hold the same stream
    for k in (4, 5, 6):                       # inflate, record split and bundle cuts on the device (br_bam_reader): the same stream
        assert np.array_equal(outs[0], outs[k]), k


def test_cli_device_readers_on_several_devices_give_the_single_device_stream(tmp_path):
    """--devices a,b,... with the device reader: the file's BGZF blocks are cut into pieces, piece k is inflated, split and
    projected by device k mod N (one br_bam_reader per device, nothing one host thread wide).  A piece's first block starts
    anywhere -- inside a record, inside a multi-mapper's read-name group (p_multimap = 0.4: groups of up to eight records;
    with two or three blocks per piece nearly every boundary falls inside both) -- so a piece guesses its first record, cuts
    itself off at read-name changes by the rule its neighbour uses too, and the guess is checked against the neighbour's
    chain.  The GPU box has one card: the same device is listed two and three times (separate readers, contexts, index
    replicas).  The record stream and the report must be those of the single-device run with the host reader; with the
    BRAMBLE_AMD_PIECE_SPOIL hook every second / every guessed start is treated as wrong and the piece is processed again from
    the start its neighbour found: the same stream again."""
    ann = synth.Annotation("G", n_genes=900, n_refs=3)
    annd = ann.as_dict()
    b = ann.reads(15000, "pe", with_records=1, p_multimap=0.4)
    stream = framed_stream(b, unmapped_every=37)
    gtf, in_bam = str(tmp_path / "g.gtf"), str(tmp_path / "in.bam")
    bamio.write_gtf(gtf, annd)
    bamio.write_bam(in_bam, "@HD\tVN:1.6\n", [(n, 1000000) for n in annd["refnames"]], stream.tobytes())
    assert len(bamio.bgzf_block_sizes(in_bam)) > 60
    outs, reports = [], []
    cases = ((["--device", "0", "--host-reader"], "900", {}), (["--devices", "0,0", "--device-reader"], "900", {}),
             (["--devices", "0,0,0", "--device-reader"], "400", {}), (["--devices", "0,0", "--device-reader"], "1400", {"BRAMBLE_AMD_PIECE_SPOIL": "2"}),
             (["--devices", "0,0,0", "--device-reader", "--host-deflate"], "700", {"BRAMBLE_AMD_PIECE_SPOIL": "1"}), (["--device", "0", "--device-reader"], "400", {}))
    for k, (extra, bundle, env) in enumerate(cases):
        out_bam = str(tmp_path / ("o%d.bam" % k))
        e = dict(os.environ); e.update(env); e["BRAMBLE_AMD_TIMING"] = "1"
        r = subprocess.run([BIN, in_bam, "-G", gtf, "-o", out_bam, "-p", "4", "--bundle-size", bundle] + extra,
                           capture_output=True, text=True, timeout=900, env=e)
        assert r.returncode == 0, r.stderr + r.stdout
        outs.append(bamio.read_bam(out_bam))
        reports.append([l for l in r.stdout.splitlines() if l.startswith("# ")])
        if env:
            assert "processed again from the true start" in r.stderr and " 0 processed again" not in r.stderr, r.stderr
    assert len(outs[0][2]) > 500000
    for k, o in enumerate(outs[1:]):
        assert o[1] == outs[0][1] and np.array_equal(o[2], outs[0][2]), k + 1
        assert reports[k + 1] == reports[0], (k + 1, reports[k + 1], reports[0])


def test_cli_reads_stdin_and_writes_stdout(tmp_path):
    """"-" as in.bam / out.bam: a pipeline stage like `aligner | samtools view -b | bramble - -G g.gtf -o - | ...`."""
    ann = synth.Annotation("G", n_genes=300, n_refs=2)
    annd = ann.as_dict()
    b = ann.reads(2000, "pe", with_records=1)
    stream = framed_stream(b)
    gtf, in_bam, ref_bam = str(tmp_path / "g.gtf"), str(tmp_path / "in.bam"), str(tmp_path / "ref.bam")
    bamio.write_gtf(gtf, annd)
    bamio.write_bam(in_bam, "@HD\tVN:1.6\n", [(n, 1000000) for n in annd["refnames"]], stream.tobytes())
    r = subprocess.run([BIN, in_bam, "-G", gtf, "-o", ref_bam, "--quiet"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr
    piped = subprocess.run([BIN, "-", "-G", gtf, "-o", "-"], input=open(in_bam, "rb").read(), capture_output=True, timeout=600)
    assert piped.returncode == 0, piped.stderr
    out_path = tmp_path / "piped.bam"
    out_path.write_bytes(piped.stdout)
    t1, refs1, s1 = bamio.read_bam(ref_bam)
    t2, refs2, s2 = bamio.read_bam(str(out_path))
    assert refs1 == refs2 and np.array_equal(s1, s2) and len(s1) > 100000


def test_cli_empty_and_all_unmapped_inputs(tmp_path):
    gtf = str(tmp_path / "g.gtf")
    open(gtf, "w").write('chr1\tx\texon\t10\t500\t.\t+\t.\tgene_id "g"; transcript_id "t1";\n')
    refs = [("chr1", 10000)]
    ann = {"refnames": ["chr1"], "transcripts": [{"id": "t1", "ref_id": 0, "strand": "+", "exons": [[10, 501]]}]}
    from tests.test_gpu_bam_bundle import _rec_full
    unmapped = b"".join(_rec_full("u%d" % i, -1, -1, 4, "", -1, -1, "ACGT", bytes([20] * 4), b"") for i in range(10))
    for label, stream in (("empty", b""), ("unmapped", unmapped)):
        in_bam, out_bam = str(tmp_path / (label + ".bam")), str(tmp_path / (label + ".out.bam"))
        bamio.write_bam(in_bam, "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:10000\n", refs, stream)
        for reader in ("--host-reader", "--device-reader"):
            r = subprocess.run([BIN, in_bam, "-G", gtf, "-o", out_bam, reader], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr
            text, orefs, recs = bamio.read_bam(out_bam)
            assert orefs == [("t1", 491)] and recs.size == 0
            assert "# input alignments:   %d" % (0 if label == "empty" else 10) in r.stdout, reader
            assert "# unmapped reads:     %d" % (0 if label == "empty" else 10) in r.stdout, reader


def test_cli_errors(tmp_path):
    r = subprocess.run([BIN, "--version"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("version: ")
    r = subprocess.run([BIN, str(tmp_path / "missing.bam"), "-G", str(tmp_path / "x.gtf"), "-o", str(tmp_path / "o.bam")],
                       capture_output=True, text=True)
    assert r.returncode != 0
    notbam = tmp_path / "n.bam"
    notbam.write_bytes(b"this is not a bam file")
    gtf = tmp_path / "g.gtf"
    gtf.write_text('chr1\tx\texon\t10\t50\t.\t+\t.\tgene_id "g"; transcript_id "t";\n')
    r = subprocess.run([BIN, str(notbam), "-G", str(gtf), "-o", str(tmp_path / "o.bam")], capture_output=True, text=True)
    assert r.returncode != 0 and "BGZF" in r.stderr


def test_cli_several_device_workers_give_the_single_device_stream(tmp_path):
    """--devices a,b,...: one reader deals bundles to one worker per listed device (index replica + context + host
    threads each, as src/threads.cpp:114-162 feeds N workers from one reader), the writer restores bundle order.  The GPU
    box has one card, so the same device is listed two and three times: every worker is a separate context / replica,
    which is all the path knows about a device.  The record stream must be byte-identical to the single-worker run."""
    ann = synth.Annotation("G", n_genes=900, n_refs=3)
    annd = ann.as_dict()
    b = ann.reads(12000, "pe", with_records=1, p_multimap=0.1)
    stream = framed_stream(b)
    gtf, in_bam = str(tmp_path / "g.gtf"), str(tmp_path / "in.bam")
    bamio.write_gtf(gtf, annd)
    bamio.write_bam(in_bam, "@HD\tVN:1.6\n", [(n, 1000000) for n in annd["refnames"]], stream.tobytes())
    outs, reports = [], []
    for k, extra in enumerate((["--device", "0"], ["--devices", "0,0"], ["--devices", "0,0,0", "--host-deflate"])):
        out_bam = str(tmp_path / ("o%d.bam" % k))
        r = subprocess.run([BIN, in_bam, "-G", gtf, "-o", out_bam, "-p", "4", "--bundle-size", "900"] + extra,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr + r.stdout
        assert not os.path.exists(out_bam + ".tmp-bramble")
        outs.append(bamio.read_bam(out_bam))
        reports.append([l for l in r.stdout.splitlines() if l.startswith("# ")])
        assert ("on %d device worker(s)" % (k + 1)) in r.stdout
    assert len(outs[0][2]) > 500000
    for o in outs[1:]:
        assert o[1] == outs[0][1] and np.array_equal(o[2], outs[0][2])
    assert reports[1] == reports[0] and reports[2] == reports[0]


def test_cli_failed_run_leaves_no_output_file(tmp_path):
    """A run that fails mid-stream (here: a BGZF block whose payload is corrupt, after several good bundles) must exit
    non-zero and leave neither the output nor a complete-looking temporary behind."""
    ann = synth.Annotation("G", n_genes=300, n_refs=2)
    annd = ann.as_dict()
    b = ann.reads(6000, "pe", with_records=1)
    stream = framed_stream(b)
    gtf, in_bam, out_bam = str(tmp_path / "g.gtf"), str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    bamio.write_gtf(gtf, annd)
    bamio.write_bam(in_bam, "@HD\tVN:1.6\n", [(n, 1000000) for n in annd["refnames"]], stream.tobytes(), block=30000)
    raw = bytearray(open(in_bam, "rb").read())
    sizes = bamio.bgzf_block_sizes(in_bam)
    at = sum(sizes[:len(sizes) * 2 // 3]) + 30      # inside the deflate payload of a block two thirds in
    for k in range(8):
        raw[at + k] ^= 0x5a
    open(in_bam, "wb").write(bytes(raw))
    for extra in (["--devices", "0,0"], ["--device-reader"]):     # the host reader (two workers); the device reader
        r = subprocess.run([BIN, in_bam, "-G", gtf, "-o", out_bam, "-p", "2", "--bundle-size", "500"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "error" in r.stderr, extra
        assert not os.path.exists(out_bam) and not os.path.exists(out_bam + ".tmp-bramble")


@pytest.mark.parametrize("reader", [[], ["--host-reader"], ["--device-reader"], ["--device-reader", "--devices", "0,0"]])
def test_cli_setup_errors_after_the_reader_started_end_the_run(tmp_path, reader):
    """A setup step that fails while the reader is already at work (the guide file missing or malformed, the output not
    writable) must end the run with a non-zero exit, not leave it waiting for a queue nobody will finish."""
    ann = synth.Annotation("G", n_genes=300, n_refs=2)
    annd = ann.as_dict()
    b = ann.reads(30000, "pe", with_records=1)
    gtf, in_bam = str(tmp_path / "g.gtf"), str(tmp_path / "in.bam")
    bamio.write_gtf(gtf, annd)
    bamio.write_bam(in_bam, "@HD\tVN:1.6\n", [(n, 1000000) for n in annd["refnames"]], framed_stream(b).tobytes(), block=30000)
    bad_gtf = str(tmp_path / "bad.gtf")
    open(bad_gtf, "w").write("this is\tnot\ta guide file\n")
    cases = [(str(tmp_path / "missing.gtf"), str(tmp_path / "o1.bam")),
             (bad_gtf, str(tmp_path / "o2.bam")),
             (gtf, str(tmp_path / "no_such_dir" / "o3.bam"))]
    for g, o in cases:
        r = subprocess.run([BIN, in_bam, "-G", g, "-o", o, "-p", "2", "--bundle-size", "500"] + reader, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "error" in r.stderr, (g, o, r.stderr)
        assert not os.path.exists(o)
