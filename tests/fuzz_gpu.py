#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the oracle (not collected by pytest; run on a GPU box:
`python tests/fuzz_gpu.py [rounds] [seed]`).  Each round draws an annotation size, a read mode, a flag combination
(incl. the --max-* / --similarity-threshold overrides) and random synth perturbation rates, then compares
  * rows of br_project_batch with the oracle's, and
  * the BAM stream of br_project_bam_bundle (records in / records out) with the oracle's write_to_bam stream.
Prints the first diverging configuration and exits non-zero."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bramble_amd import lib, synth  # noqa: E402
from oracle import oracle_binding as ob  # noqa: E402
from tests.parity import assert_rows_equal  # noqa: E402


def run(rounds, seed, verbose=True, read_counts=(500, 3000, 9000), gene_counts=(30, 300, 1500)):
    """Returns None when every round agrees, else the description of the first diverging configuration."""
    rng = np.random.RandomState(seed)
    for it in range(rounds):
        mode = ["pe", "se", "hifi", "ont"][rng.randint(0, 4)]
        long_mode = mode in ("hifi", "ont")
        flags = {}
        if long_mode:
            flags[["lr", "lr_hq"][rng.randint(0, 2)]] = 1
        elif rng.rand() < 0.25:
            flags[["lr", "lr_hq"][rng.randint(0, 2)]] = 1     # long-read presets on short reads
        if rng.rand() < 0.3:
            flags["strict"] = 1
        if not long_mode and rng.rand() < 0.3:
            flags[["fr", "rf"][rng.randint(0, 2)]] = 1
        for key, hi in (("max_clip", 40), ("max_junc_ins", 30), ("max_junc_gap", 30), ("max_error_exon", 60)):
            if rng.rand() < 0.25:
                flags[key] = int(rng.randint(0, hi))
        if rng.rand() < 0.3:
            flags["sim_thr"] = float(np.float32(rng.uniform(0.5, 0.99)))
        with_genome = long_mode and rng.rand() < 0.5
        if with_genome:
            flags["use_fasta"] = 1
        n_genes = int(rng.choice(list(gene_counts)))
        n_refs = int(rng.randint(1, 5))
        kw = {}
        if not long_mode:
            kw = {"p_softclip": float(rng.uniform(0, 0.3)), "p_indel": float(rng.uniform(0, 0.1)),
                  "p_junc_shift": float(rng.uniform(0, 0.2)), "p_intergenic": float(rng.uniform(0, 0.1)),
                  "p_multimap": float(rng.uniform(0, 0.3))}
            xs = rng.rand() < 0.5
        else:
            kw = {"p_wobble": float(rng.uniform(0, 0.4)), "p_skip_small": float(rng.uniform(0, 0.3)),
                  "p_novel_small": float(rng.uniform(0, 0.1)), "p_clip": float(rng.uniform(0, 0.8))}
        xs = (not long_mode) and locals().get("xs", False)
        n_reads = int(rng.choice(list(read_counts)))
        desc = dict(it=it, mode=mode, flags=flags, n_genes=n_genes, n_refs=n_refs, n_reads=n_reads, kw=kw, seed=seed)
        ann = synth.Annotation("G", n_genes=n_genes, n_refs=n_refs, with_genome=with_genome, seed=int(rng.randint(1, 1 << 30)))
        annd = ann.as_dict()
        b = ann.reads(n_reads, mode, with_records=1, with_seq=1 if with_genome else 0, seed=int(rng.randint(1, 1 << 30)), xs_tag=bool(xs), **kw)
        idx = lib.Index(annd, device=0)
        ctx = lib.Context(idx)
        ctx.set_param("small_batch", it & 1)        # the path without host round trips on odd rounds, the ordinary one on even rounds
        desc["small_batch"] = it & 1
        ctx.set_param("group_lanes", int(rng.choice([8, 16, 32, 64])))
        # direct rows (pairing before the emit pass) is the default of the ordinary pipeline; one round in four takes the
        # match-table path it replaced (still the path of the similarity-filter presets and of -S)
        desc["direct_rows"] = int(rng.rand() >= 0.25)
        ctx.set_param("direct_rows", desc["direct_rows"])
        cfg = lib.make_config(**flags)
        oi = ob.OracleIndex(annd)
        try:
            prod = ctx.project_batch(cfg, b)
            orc, _, _ = ob.run(oi, ob.make_flags(**flags), b, want_matches=False, bam_records=(b["rec_blob"], b["rec_off"]))
            assert_rows_equal(prod, orc)
            stream, roff, rlen = synth.Annotation.frame_records(b)
            got, counters = ctx.project_bam_bundle(cfg, stream, roff, rlen, np.arange(n_refs, dtype=np.int32))
            # the records are the input here (their SEQ carries injected N codes the flat table lacks): oracle from records
            orc2, _, _, _ = ob.run_bam(oi, ob.make_flags(**flags), stream, roff, rlen, np.arange(n_refs, dtype=np.int32))
            if not np.array_equal(got, orc2["bam_stream"]):
                raise AssertionError("BAM streams differ (%d vs %d bytes)" % (len(got), len(orc2["bam_stream"])))
        except Exception as e:  # noqa: BLE001
            return "%s %s" % (desc, repr(e)[:500])
        finally:
            ctx.close()
            idx.close()
        if verbose:
            print("ok", it, mode, flags, "rows", orc["n_rows"], flush=True)
    return None


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(rounds, seed)
    if bad:
        print("DIVERGENCE", bad)
        sys.exit(1)
    print("fuzz ok: %d rounds" % rounds)


if __name__ == "__main__":
    main()
