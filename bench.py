#!/usr/bin/env python3
"""Headline benchmark: projected alignments/sec against a GENCODE-scale annotation.

One "step" = one pass of the projection hot path (k_segment -> count pass -> k_pair_mask ->
scan -> k_group_desc -> k_expand_rows -> k_emit_rows: "direct rows") over
one synthetic name-collated batch.  At N=1
the workload is BASELINE.json configs[1] (paired-end short reads vs a GENCODE-shaped
annotation, 1 x MI355X); N>1 shards read-name groups across ranks (one process per GPU, a
private index replica each, no collective on the data path) -- weak scaling.

  python bench.py --gpus N --steps K --warmup W [--pairs P]

Rank 0 prints ONE JSON line (contract in the task statement):
  value / ms_per_step   device-resident: the batch is already in HBM when the timed region starts and the packed
                        rows stay in HBM (said in config.workload);
  pcie_inclusive        the same batch from pinned host arrays to packed rows in pinned host memory through the
                        staged C ABI (br_batch_stage / br_project_staged / br_host_rows_wait): batch k+1 uploads and
                        batch k-1 downloads while batch k is projected;
  roofline              whole path: the SURVEY 8d algorithmic bytes of one step / ms_per_step (never above 1), with
                        a per-kernel table (each kernel's own share of the formula's terms over its own hipEvent
                        time) and the dominant kernel by single-kernel device time;
  cpu_baseline          the CPU oracle (restatement of the reference algorithm) on a bounded sample on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def kernel_bytes(cn, n_aln, n_rows):
    """Per-kernel share of the SURVEY 8d formula B = B_in + B_idx + B_out: what each kernel must
    move at least.  B_in = 24 n + 4 ncig; the index term splits into the search part (8 ceil(log2 N) per read exon and
    strand: the bucket tables replace it, nobody is charged for it) and 40 B per overlap hit; B_out = 4 n + sum over
    matches (24 + 4 n_out)."""
    hits40 = 40 * cn["overlap_hits"]
    match_rec = 24 * cn["matches"] + 4 * cn["out_cigar_words"]
    return {
        "segment": cn["B_in"],                             # reads every input field and CIGAR word once
        "count": cn["B_in"] + hits40,                      # heads again + one 40-byte row per overlap hit
        "emit": 4 * n_aln + 40 * cn["matches"] + match_rec,  # one row re-read + one match record written per match
        "rows": match_rec + 24 * n_rows,                   # match records read, 24 B per emitted record written
        # direct rows: the pairing reads a 4-byte transcript id per survivor and writes 4 B per alignment (records of the
        # leader); the emit kernels re-read one 40-byte row per EMITTED record and write its 24 bytes + its CIGAR words
        # (the formula's per-match CIGAR words stand in for the emitted ones: an upper bound by the dropped 9 %)
        "pairing": 4 * cn["matches"] + 4 * n_aln,
        "emit_rows": 4 * n_aln + 40 * n_rows + 24 * n_rows + 4 * cn["out_cigar_words"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("BENCH_PAIRS", 10_000_000)),
                    help="read pairs per GPU per step (BASELINE configs[1]: 10M)")
    ap.add_argument("--group-lanes", type=int, default=int(os.environ.get("BRAMBLE_AMD_GROUP_LANES", 0)))
    ap.add_argument("--cpu-sample", type=int, default=400_000, help="alignments in the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # One command, N ranks: like the reference's one reader feeding N workers from a single process
        # (src/threads.cpp:114-162, src/bramble.cpp:675-690), `python bench.py --gpus N` starts its own ranks.  This
        # process has not touched HIP (torch is not even imported yet): the ranks are fresh children, one per GPU.
        raise SystemExit(spawn_ranks(args.gpus))

    import ctypes as C
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d, or without a launcher" % (args.gpus, world, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL (backend "nccl") in production; BENCH_DIST_BACKEND=gloo rehearses the N>1 path on a box
        # with fewer GPUs than ranks (ranks then share devices)
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        ndev = max(torch.cuda.device_count(), 1)
        try:
            if backend == "nccl":
                dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world,
                                            device_id=torch.device("cuda", local_rank % ndev))
                # the first collective is where RCCL really connects the ranks: do it here, so that a failure names its rank
                probe = torch.ones(1, device="cuda:%d" % (local_rank % ndev))
                dist_mod.all_reduce(probe)
                torch.cuda.synchronize()
            else:
                dist_mod.init_process_group(backend=backend, rank=rank, world_size=world)
        except Exception as e:  # noqa: BLE001 -- say which rank / device before the non-zero exit
            sys.stderr.write("bench.py: rank %d of %d (local rank %d, device cuda:%d of %d visible, backend %s, "
                             "HSA_ENABLE_IPC_MODE_LEGACY=%s) could not join the process group: %s: %s\n"
                             % (rank, world, local_rank, local_rank % ndev, ndev, backend,
                                os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), type(e).__name__, e))
            sys.stderr.flush()
            raise SystemExit(3)
        dist = dist_mod
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    from bramble_amd import device as brdev
    from bramble_amd import lib, synth

    t0 = time.time()
    ann = synth.Annotation("G")
    index = lib.Index.from_flat(ann.flat, device=local_rank)
    ctx = lib.Context(index)
    if args.group_lanes:
        ctx.set_param("group_lanes", args.group_lanes)
    cfg = lib.make_config()  # short-read defaults, unstranded (XS absent): both strands tried
    # this rank's shard of read-name groups: its own seeded batch (weak scaling)
    batch = ann.reads(args.pairs, "pe", seed=(synth.SEED ^ 0x51ED) + 7919 * rank)
    n_aln = int(batch["n_aln"])
    dbatch = brdev.upload_batch(batch, dev)
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max_sum(elapsed, count):
        if dist is None:
            return elapsed, float(count)
        rdev = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cnt = torch.tensor([count], dtype=torch.float64, device=rdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        return float(t.item()), float(cnt.item())

    def gather_all(x):
        """every rank's value of x (rank order): what the max above hides"""
        if dist is None:
            return [float(x)]
        rdev = dev if dist.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([x], dtype=torch.float64, device=rdev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        return [float(p.item()) for p in parts]

    # ---- leg 1 (value): device-resident ----
    rows = None
    for _ in range(args.warmup):
        rows = ctx.project_batch_device(cfg, dbatch, stream)
    ctx.set_profiling(True)
    kernel_ms = {}
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        rows = ctx.project_batch_device(cfg, dbatch, stream)
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms = {k: [ms, ln] for k, (ms, ln) in ctx.kernel_ms_sum().items()}   # every launch of the timed region, summed by the context
    ctx.set_profiling(False)
    per_rank_ms = [1e3 * e / args.steps for e in gather_all(elapsed)]
    per_rank_aln = gather_all(n_aln)
    elapsed, total_aln = reduce_max_sum(elapsed, n_aln)
    n_rows, n_matches = int(rows.n_rows), int(rows.n_matches)
    counters = ctx.collect_counters(dbatch, stream) if rank == 0 else None

    # ---- leg 2: PCIe-inclusive (pinned host arrays in, packed rows in pinned host memory out) ----
    pcie = None
    if not args.no_pcie:
        pinned = {"n_aln": n_aln, "seq_off": None, "seqs": None}
        for name, dt in (("ref_id", np.int32), ("ref_start", np.int32), ("flags", np.uint16), ("xs", np.int8),
                         ("ts", np.int8), ("cigar_off", np.uint64), ("cigar", np.uint32), ("mate_ref_id", np.int32),
                         ("mate_start", np.int32), ("name_off", np.uint64), ("names", np.uint8), ("l_qseq", np.int32)):
            a = np.ascontiguousarray(batch[name], dtype=dt)
            view = {np.uint16: np.int16, np.uint32: np.int32, np.uint64: np.int64}.get(dt, dt)
            pinned[name] = torch.from_numpy(a.view(view)).pin_memory().numpy().view(dt)
        keep = []
        bs = lib._batch_struct(pinned, keep)
        h2d_bytes = sum(int(pinned[k].nbytes) for k in pinned if isinstance(pinned[k], np.ndarray))
        L = lib.lib()
        res = [lib.BrHostRows(), lib.BrHostRows()]

        trace = os.environ.get("BENCH_PCIE_TRACE") is not None   # per-call host times of the loop on stderr

        def run_pipelined(n_steps):
            lib.check(L.br_batch_stage(ctx.h, C.byref(bs), 0), "br_batch_stage")
            for k in range(n_steps):
                ta = time.perf_counter()
                if k + 1 < n_steps:
                    lib.check(L.br_batch_stage(ctx.h, C.byref(bs), (k + 1) % 2), "br_batch_stage")
                tb = time.perf_counter()
                lib.check(L.br_project_staged(ctx.h, C.byref(cfg), k % 2, C.byref(res[k % 2])), "br_project_staged")
                tc = time.perf_counter()
                if k >= 1:
                    lib.check(L.br_host_rows_wait(ctx.h, (k - 1) % 2), "br_host_rows_wait")
                if trace:
                    sys.stderr.write("pcie step %d: stage %.1f ms, project %.1f ms, wait %.1f ms\n"
                                     % (k, 1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (time.perf_counter() - tc)))
            lib.check(L.br_host_rows_wait(ctx.h, (n_steps - 1) % 2), "br_host_rows_wait")

        run_pipelined(max(args.warmup, 2))
        barrier()
        t1 = time.perf_counter()
        run_pipelined(args.steps)
        barrier()
        p_el = time.perf_counter() - t1
        p_per_rank_ms = [1e3 * e / args.steps for e in gather_all(p_el)]   # N ranks share one host's PCIe root and memory: what the max hides
        p_el, p_total = reduce_max_sum(p_el, n_aln)
        r = res[(args.steps - 1) % 2]
        d2h_bytes = 24 * int(r.n_rows) + 4 * int(r.n_pool_words) + 12 * n_aln + 8
        assert int(r.n_rows) == n_rows
        pcie = {"value": p_total * args.steps / p_el, "unit": "alignments/s", "ms_per_step": 1e3 * p_el / args.steps,
                "per_rank_ms_per_step": [round(x, 3) for x in p_per_rank_ms],
                "h2d_bytes_per_step": h2d_bytes, "d2h_bytes_per_step": d2h_bytes,
                "what": "br_batch_stage / br_project_staged / br_host_rows_wait: pinned host SoA in -> packed rows (24 B per "
                        "record + row_off + mate_idx) in pinned host memory; read-name groups and mate index computed on the "
                        "device; uploads, projection and downloads of consecutive batches overlap"}

    if rank == 0:
        alg_bytes = counters["B_in"] + counters["B_idx"] + counters["B_out"]
        step_s = elapsed / args.steps
        per_ms = {k: v[0] / args.steps for k, v in kernel_ms.items() if v[1]}
        # PMC traffic (profiles/pmc_traffic.json: separate --pmc passes of this same command, per launch)
        pmc = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("pairs") == args.pairs:
                    pmc = tj.get("kernels", {})
            except Exception:
                pmc = {}
        kb = kernel_bytes(counters, n_aln, n_rows)

        def entry(ms, nbytes, names):
            tr = [pmc[x]["hbm_bytes_corrected"] for x in names if x in pmc]
            e = {"ms": round(ms, 4), "alg_bytes": nbytes, "achieved": nbytes / (ms * 1e-3) / 1e9 if ms else None,
                 "traffic": sum(tr) if len(tr) == len(names) and tr else None}
            e["frac"] = e["achieved"] / HBM_PEAK_GBS if e["achieved"] is not None else None
            return e

        g = per_ms.get
        table = {
            "k_segment": entry(g("k_segment", 0.0), kb["segment"], ["k_segment"]),
            "count (k_project<G,false,false,1|2>)": entry(g("k_project<G,false,false,1>", 0.0) + g("k_project<G,false,false,2>", 0.0),
                                                          kb["count"],
                                                          ["k_project<G,false,false,1>", "k_project<G,false,false,2>"]),
        }
        if g("k_emit_rows<2>", 0.0) or g("k_emit_rows<1>", 0.0):
            # direct rows (the default): pairing on the survivor sets, then the emit kernels write the packed rows
            pair_names = ["k_group_ids", "k_pair_mask", "k_big<0>+k_pair_big", "k_name_seed", "k_group_desc"]
            emit_names = ["k_expand_rows", "k_emit_rows<1>", "k_emit_rows<2>", "k_big<1>"]
            table["pairing (k_group_ids + k_pair_mask + k_big<0> + k_pair_big + k_name_seed + k_group_desc)"] = entry(
                sum(g(x, 0.0) for x in pair_names), kb["pairing"], pair_names)
            table["emit (k_expand_rows + k_emit_rows<1|2> + k_big<1>)"] = entry(sum(g(x, 0.0) for x in emit_names), kb["emit_rows"], emit_names)
        else:
            table["emit (k_expand + k_emit_dense<false,1|2> + k_project<64,true>)"] = entry(
                g("k_expand", 0.0) + g("k_emit_dense<false,1>", 0.0) + g("k_emit_dense<false,2>", 0.0) + g("k_project<64,true>", 0.0),
                kb["emit"], ["k_expand", "k_emit_dense<false,1>", "k_emit_dense<false,2>", "k_project<64,true>"])
            table["rows (k_group_ids + k_pair<false|true> + k_primary + k_rows)"] = entry(
                g("k_group_ids", 0.0) + g("k_pair<false>", 0.0) + g("k_pair<true>", 0.0) + g("k_rows", 0.0) + g("k_primary", 0.0),
                kb["rows"], ["k_group_ids", "k_pair<false>", "k_pair<true>", "k_rows", "k_primary"])
        # dominant = the single kernel with the most device time per step (k_scan_* is a group of small launches)
        single = {k: v for k, v in per_ms.items() if k != "k_scan_*"}
        dom = max(single, key=lambda k: single[k])
        whole_tr = [pmc[k]["hbm_bytes_corrected"] for k in pmc]
        out = {
            "metric": "projected alignments/sec vs GENCODE-scale annotation",
            "value": total_aln * args.steps / elapsed,
            "unit": "alignments/s",
            "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
            "dist_backend": (dist.get_backend() if dist is not None else None),
            "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
            "per_rank_alignments_per_step": [int(x) for x in per_rank_aln],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * step_s,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": ("device-resident (inputs already in HBM, packed rows left in HBM): %dM paired-end 2x100 short reads "
                             "(name-collated, unstranded) vs GENCODE-shaped synthetic annotation (%d transcripts, %d transcript-exon "
                             "rows, 25 refs); BASELINE.json configs[1]" % (args.pairs // 1_000_000, ann.n_tx, ann.n_exons))
                if args.pairs >= 1_000_000 else
                "device-resident: %d paired-end 2x100 short reads vs GENCODE-shaped synthetic annotation (%d transcripts)" % (args.pairs, ann.n_tx),
                "alignments_per_gpu_per_step": n_aln,
                "pairs_per_gpu_per_step": args.pairs,
                "sharding": "read-name groups per rank, index replicated, no collective",
                "group_lanes": args.group_lanes or 8,
                "seed": hex(synth.SEED),
            },
            "projected_records_per_step": n_rows,
            "matches_per_step": n_matches,
            "setup_seconds": round(setup_s, 1),
            "kernel_ms_per_step": {k: round(v, 4) for k, v in per_ms.items()},
            "pcie_inclusive": pcie,
            "roofline": {
                "bound": "hbm",
                "kernel": "whole path (%d timed kernel groups per step: k_scan_* is three launches, k_pair_mask and "
                          "k_big<0>+k_pair_big two each)" % int(sum(v[1] for v in kernel_ms.values()) / args.steps),
                "achieved": alg_bytes / step_s / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                "traffic": sum(whole_tr) if whole_tr else None,
                "algorithmic_bytes_per_alignment": alg_bytes / n_aln,
                "algorithmic_bytes_per_launch": alg_bytes,
                "counters": counters,
                "dominant_kernel": {"name": dom, "ms_per_launch": round(single[dom], 4),
                                    "share_of_step": round(single[dom] / (1e3 * step_s), 3)},
                "per_kernel": table,
                "note": "achieved = the SURVEY 8d algorithmic bytes of one step over the whole step (every launch of the path); "
                        "per_kernel charges each stage only the formula terms it must move itself (DESIGN.md section 4) over its "
                        "own hipEvent time (k_big<0> + k_pair_big, k_name_seed, k_group_desc and k_big<1> run on the context's second "
                        "and third streams beside k_pair_mask, the scan and the emit kernels: overlapped kernels stretch each other, "
                        "so the stage times sum to more than ms_per_step and the pairing / emit fractions are lower bounds); traffic "
                        "is NOT measured by this run: it is the corrected rocprofv3 FETCH_SIZE + WRITE_SIZE per launch that the builder "
                        "collected for this workload and committed as profiles/pmc_traffic.json (null when that file is for another size)",
            },
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N=1 only (the other ranks would wait for it)
            out["cpu_baseline"] = cpu_baseline(ann, batch, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a child (fresh processes: nothing here has initialised the GPU), pass the ranks' output through, and
    return the launcher's exit code -- non-zero when any rank failed, in which case no JSON line counts."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()          # rank 0's single JSON line: printed last, once the launcher has ended cleanly
        else:
            sys.stdout.write(ln)
            sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks ended without a result line", file=sys.stderr)
        return 1
    if rc == 0:
        print(line, flush=True)
    return rc


def cpu_baseline(ann, batch, sample_aln):
    """The CPU oracle (restatement of the reference algorithm; NOT the bramble binary, which
    cannot be built here) on the first `sample_aln` alignments of the same batch, cut at a
    read-name boundary: -p 1 and -p nproc, bundle-per-thread like src/threads.cpp."""
    import numpy as np
    from oracle import oracle_binding as ob
    native = ob.use_native()   # -O3 -march=native -ffp-contract=off, compiled on this machine
    n = int(batch["n_aln"])
    noff = batch["name_off"]
    names = batch["names"]

    def name(i):
        return bytes(names[int(noff[i]):int(noff[i + 1])])

    def head(cut):
        cut = min(cut, n)
        while cut < n and cut > 0 and name(cut) == name(cut - 1):
            cut += 1
        sub = {"n_aln": cut}
        for k in ("ref_id", "ref_start", "flags", "xs", "ts", "mate_ref_id", "mate_start", "l_qseq"):
            sub[k] = batch[k][:cut]
        sub["cigar_off"] = batch["cigar_off"][:cut + 1]
        sub["cigar"] = batch["cigar"][:int(sub["cigar_off"][-1])]
        sub["name_off"] = batch["name_off"][:cut + 1]
        sub["names"] = batch["names"][:int(sub["name_off"][-1])]
        sub["seq_off"] = None
        sub["seqs"] = None
        return sub

    f = ann.flat
    oi = ob.OracleIndex.__new__(ob.OracleIndex)
    L = ob.lib()
    oi.h = L.orc_index_new()
    import numpy as np  # noqa: F811
    exs = np.stack([f["ex_start"], f["ex_end"]], axis=1).astype(np.uint32)
    off = f["tx_exon_off"].astype(np.int64)
    for t in range(len(f["tx_ref"])):
        e = np.ascontiguousarray(exs[off[t]:off[t + 1]]).reshape(-1)
        L.orc_index_add_transcript(oi.h, int(f["tx_ref"][t]), bytes([int(f["tx_strand"][t])]), b"", e.ctypes.data,
                                   len(e) // 2, None, 0)
    L.orc_index_finish(oi.h)
    # the GPU box gives one GPU a 16-core CPU share
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("BENCH_CPU_THREADS", 16))))
    # the reference cuts bundles of >= 100000 alignments (src/bramble.cpp:362): give every worker two.  Both legs run the
    # SAME sample (about 13 s on one thread, 1 s on sixteen)
    subn = head(max(sample_aln, 2 * 100000 * cores))
    _, _, s1 = ob.run(oi, ob.make_flags(), subn, n_threads=1, want_matches=False)
    _, _, sn = ob.run(oi, ob.make_flags(), subn, n_threads=cores, want_matches=False)
    return {
        "value": subn["n_aln"] / sn,
        "unit": "alignments/s",
        "cores": cores,
        "kind": "port",
        "build": "g++ -O3 -march=native -ffp-contract=off on this host" if native else "g++ -O3 -ffp-contract=off (portable build: the native rebuild failed)",
        "sample": "first %d alignments of the same batch (cut at a read-name boundary) on %d threads and on 1 thread, "
                  "bundles of >=100000 alignments per worker like src/threads.cpp; CPU oracle = restatement of the "
                  "reference algorithm (not the bramble binary)" % (subn["n_aln"], cores),
        "value_1_thread": subn["n_aln"] / s1,
        "cores_1_thread": 1,
    }


if __name__ == "__main__":
    main()
