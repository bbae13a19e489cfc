#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (not the bench contract line; numbers for BASELINE.md):

  python bench_extra.py c5 [--reads N]   HiFi-like --lr-hq --strict --similarity-threshold 0.95 (configs[4], 1 GPU)
  python bench_extra.py c3 [--reads N]   ONT-like --lr -S with a synthetic genome: clip rescue incl. k_ksw GCUPS (configs[2])
  python bench_extra.py bam [--reads N]  re-encode the rows of configs[1] as BAM records (device part)
  python bench_extra.py bundle [--reads N]  raw BAM records resident in HBM -> projected BAM records (br_project_bam_device)
  python bench_extra.py cli [--reads N] [--threads T]  the command line file to file (BGZF inflate, device path, BGZF deflate)
  python bench_extra.py small            small calls: us per device-resident step at 1 .. 52 000 pairs (without the per-kernel
                                         events bench.py keeps on), the path without host round trips against the ordinary one,
                                         and br_project_group / br_project_groups host to host from plain C (profiles/group_latency.c)

Same protocol as bench.py: inputs resident in HBM, warmup, hipEvent kernel times, one JSON line.
"""
import argparse
import json
import os

import numpy as np
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c3", "c5", "bam", "bundle", "cli", "small", "inflate"])
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import torch
    from bramble_amd import device as brdev
    from bramble_amd import lib, synth
    if args.config == "small":
        import subprocess
        root = os.path.dirname(os.path.abspath(__file__))
        ann = synth.Annotation("G")
        idx = lib.Index.from_flat(ann.flat, device=0)
        cfg = lib.make_config()
        out = {"config": "small", "unit": "us per call", "device_resident_step": {}}
        for pairs in (1, 32, 1000, 5000, 30000, 52000):
            batch = ann.reads(pairs, "pe", seed=1234 + pairs)
            db = brdev.upload_batch(batch, "cuda:0")
            row = {"alignments": int(batch["n_aln"])}
            for small in (1, 0):
                ctx = lib.Context(idx)
                ctx.set_param("small_batch", small)
                for _ in range(20):
                    ctx.project_batch_device(cfg, db, 0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k = 300
                for _ in range(k):
                    ctx.project_batch_device(cfg, db, 0)
                torch.cuda.synchronize()
                row["no_host_round_trips" if small else "ordinary_pipeline"] = round((time.perf_counter() - t0) / k * 1e6, 1)
                ctx.close()
            out["device_resident_step"]["pairs=%d" % pairs] = row
        exe = "/tmp/group_latency"
        try:
            subprocess.check_call(["gcc", "-O2", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "profiles", "group_latency.c"),
                                   "-o", exe, "-L", os.path.join(root, "bramble_amd"), "-lbramble_amd", "-Wl,-rpath," + os.path.join(root, "bramble_amd")])
            runs = [json.loads(subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]) for _ in range(3)]
            out["host_to_host_from_c"] = {k: [r[k] for r in runs] for k in runs[0]}
        except Exception as e:   # no compiler on the box: the device-resident numbers stand alone
            out["host_to_host_from_c"] = "not measured: %s" % e
        print(json.dumps(out))
        return
    if args.config == "bam":
        # re-encode the rows of configs[1] as BAM records: HBM-bound byte streaming
        n = args.reads or 10_000_000
        ann = synth.Annotation("G")
        batch = ann.reads(n, "pe", with_records=1)
        cfg = lib.make_config()
        idx = lib.Index.from_flat(ann.flat, device=0)
        ctx = lib.Context(idx)
        pass
        if os.environ.get("BAM_LANES"):
            ctx.set_param("bam_lanes", int(os.environ["BAM_LANES"]))
        db = brdev.upload_batch(batch, "cuda:0")
        blob, roff = brdev.upload_records(batch, "cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        rows = ctx.project_batch_device(cfg, db, stream)
        n_rows = int(rows.n_rows)
        bam = ctx.bam_encode_device(cfg, blob, roff, stream)
        ctx.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kms = {}
        for _ in range(args.steps):
            bam = ctx.bam_encode_device(cfg, blob, roff, stream)
            for k, (ms, ln) in ctx.kernel_ms().items():
                kms[k] = kms.get(k, 0.0) + ms
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        out_bytes = int(bam.n_bytes)
        in_bytes_per_row = out_bytes  # every output byte is copied from (or computed next to) one input byte
        k_ms = kms.get("k_bam_scan+k_bam_size+k_bam_encode", 0.0) / args.steps
        print(json.dumps({"config": "bam", "workload": "re-encode the %d rows of %d paired-end alignments as BAM records" % (n_rows, batch["n_aln"]),
                          "rows_per_s": n_rows / el, "ms_per_step": el * 1e3, "output_bytes": out_bytes,
                          "input_record_bytes": int(len(batch["rec_blob"])), "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in kms.items() if v},
                          "bam_kernels_GBps_read_plus_write": (out_bytes + in_bytes_per_row) / (k_ms * 1e-3) / 1e9 if k_ms else None,
                          "hbm_peak_GBps": 8000.0}))
        return
    if args.config == "bundle":
        n = args.reads or 10_000_000
        ann = synth.Annotation("G")
        batch = ann.reads(n, "pe", with_records=1)
        stream_h, roff, rlen = synth.Annotation.frame_records(batch)
        cfg = lib.make_config()
        idx = lib.Index.from_flat(ann.flat, device=0)
        ctx = lib.Context(idx)
        blob = torch.from_numpy(stream_h).to("cuda:0")
        off_d = torch.from_numpy(roff.view(np.int64)).to("cuda:0")
        len_d = torch.from_numpy(rlen.view(np.int32)).to("cuda:0")
        ref_map = np.arange(ann.flat["n_refs"], dtype=np.int32)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(args.warmup):
            rows, bam = ctx.project_bam_device(cfg, blob, off_d, len_d, ref_map, st)
        ctx.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kms = {}
        for _ in range(args.steps):
            rows, bam = ctx.project_bam_device(cfg, blob, off_d, len_d, ref_map, st)
            for k, (ms, ln) in ctx.kernel_ms().items():
                kms[k] = kms.get(k, 0.0) + ms
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        # optional last stage: BGZF deflate of the projected stream on the device
        if os.environ.get("DEFLATE_DYNAMIC") is not None:
            ctx.set_param("deflate_dynamic", int(os.environ["DEFLATE_DYNAMIC"]))
        z = ctx.bgzf_deflate_device(torch.as_tensor(brdev._DevArray(bam.data, int(bam.n_bytes), "|u1"), device="cuda:0"), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        zk = {}
        for _ in range(args.steps):
            z = ctx.bgzf_deflate_device(torch.as_tensor(brdev._DevArray(bam.data, int(bam.n_bytes), "|u1"), device="cuda:0"), st)
            for k, (ms, ln) in ctx.kernel_ms().items():
                zk[k] = zk.get(k, 0.0) + ms
        torch.cuda.synchronize()
        zel = (time.perf_counter() - t0) / args.steps
        deflate = {"ms_per_step": zel * 1e3, "compressed_bytes": int(z.numel()), "ratio": int(bam.n_bytes) / max(int(z.numel()), 1),
                   "GBps_in": int(bam.n_bytes) / zel / 1e9, "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in zk.items() if v}}
        print(json.dumps({"config": "bundle", "device_deflate": deflate, "workload": "%d raw paired-end BAM records resident in HBM -> %d projected BAM records (reader side, projection and re-encoding on the device)" % (len(rlen), int(bam.n_rows)),
                          "alignments_per_s": len(rlen) / el, "ms_per_step": el * 1e3, "input_bytes": int(stream_h.size),
                          "output_bytes": int(bam.n_bytes), "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in kms.items() if v}}))
        return
    if args.config == "inflate":
        # BGZF inflate on the device against the host codec: the raw records of N read pairs, compressed by the library's writer
        import ctypes as C
        import tempfile
        n = args.reads or 10_000_000
        ann = synth.Annotation("G")
        batch = ann.reads(n, "pe", with_records=1)
        stream_h, roff, rlen = synth.Annotation.frame_records(batch)
        L = lib.lib()
        L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.br_bgzf_read_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.br_free_buffer.argtypes = [C.c_void_p]
        tmp = tempfile.mkdtemp(prefix="bramble_inflate_")
        path = os.path.join(tmp, "records.bgzf")
        assert L.br_bgzf_write_file(path.encode(), stream_h.ctypes.data, stream_h.size, args.threads, 6) == 0
        raw = np.fromfile(path, dtype=np.uint8)
        t0 = time.perf_counter()
        blocks, consumed, total = lib.bgzf_scan(raw)
        scan_s = time.perf_counter() - t0
        assert consumed == raw.size and total == stream_h.size
        idx = lib.Index.from_flat(ann.flat, device=0)
        ctx = lib.Context(idx)
        src = torch.from_numpy(raw).to("cuda:0")
        out = ctx.bgzf_inflate_device(src, blocks)
        assert torch.equal(out.cpu(), torch.from_numpy(stream_h))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = ctx.bgzf_inflate_device(src, blocks)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        host = {}
        for th in (1, args.threads):
            p, nb = C.c_void_p(), C.c_uint64()
            t0 = time.perf_counter()
            assert L.br_bgzf_read_file(path.encode(), th, C.byref(p), C.byref(nb)) == 0
            host["threads=%d" % th] = round(time.perf_counter() - t0, 3)
            L.br_free_buffer(p)
        print(json.dumps({"config": "inflate", "workload": "%d BAM records, %d bytes in %d BGZF blocks of %d compressed bytes (host writer, level 6)" % (len(rlen), total, len(blocks), raw.size),
                          "device_ms": el * 1e3, "device_GBps_out": total / el / 1e9, "device_GBps_in": raw.size / el / 1e9, "block_scan_host_s": round(scan_s, 3),
                          "host_whole_file_s (br_bgzf_read_file, includes its buffer growth)": host, "kernel_ms": {k: round(v[0], 3) for k, v in ctx.kernel_ms().items() if v[0]}}))
        return
    if args.config == "cli":
        pass
        import subprocess
        import tempfile
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from tests import bamio
        n = args.reads or 2_000_000
        ann = synth.Annotation("G")
        annd = ann.as_dict()
        batch = ann.reads(n, "pe", with_records=1)
        stream_h, roff, rlen = synth.Annotation.frame_records(batch)
        tmp = os.environ.get("CLI_TMP") or tempfile.mkdtemp(prefix="bramble_cli_")   # CLI_TMP: keep guides.gtf / in.bam where a profiler run finds them
        os.makedirs(tmp, exist_ok=True)
        gtf, in_bam = os.path.join(tmp, "guides.gtf"), os.path.join(tmp, "in.bam")
        bamio.write_gtf(gtf, annd)
        refs = [(r, 250_000_000) for r in annd["refnames"]]
        t0 = time.perf_counter()
        # input preparation (not measured): BAM header + the framed records, BGZF-compressed with the library's writer
        import ctypes as C
        import struct
        text = ("@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)).encode()
        hb = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
        for name, ln in refs:
            nm = name.encode() + b"\0"
            hb += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
        whole = np.concatenate([np.frombuffer(bytes(hb), dtype=np.uint8), stream_h])
        L = lib.lib()
        L.br_bgzf_write_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        assert L.br_bgzf_write_file(in_bam.encode(), whole.ctypes.data, whole.size, args.threads, 6) == 0
        del whole
        prep = time.perf_counter() - t0
        exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bramble_amd", "bin", "bramble")
        res = {}
        # CLI_LEVELS entries: a host codec level or "device", optionally "@NAME=VALUE[@...]" environment for that run (A/B on one input)
        for entry in os.environ.get("CLI_LEVELS", "1,6,device").split(","):
            parts = entry.split("@")
            level = parts[0] if parts[0] == "device" else int(parts[0])
            run_env = dict(os.environ)
            run_env.update(dict(kv.split("=", 1) for kv in parts[1:] if not kv.startswith("ARG=")))
            entry_args = [kv[4:] for kv in parts[1:] if kv.startswith("ARG=")]   # "@ARG=--host-reader": an extra command-line argument for that run
            os.sync()   # the previous run's output is on its way to the disk: not this run's business
            time.sleep(float(os.environ.get("CLI_GAP_S", "0")))   # ... and the driver is still taking the previous process apart
            out_bam = os.path.join(tmp, "out%s.bam" % level)
            if os.path.exists(out_bam):
                os.remove(out_bam)
            t0 = time.perf_counter()
            e0 = time.time()
            import resource
            ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
            codec = ["--device-deflate"] if level == "device" else ["--compression-level", str(level)]
            extra = os.environ.get("CLI_EXTRA", "").split()
            r = subprocess.run([exe, in_bam, "-G", gtf, "-o", out_bam, "-p", str(args.threads)] + codec + extra + entry_args,
                               capture_output=True, text=True, env=run_env)
            wall = time.perf_counter() - t0
            e1 = time.time()
            if os.environ.get("BRAMBLE_AMD_TIMING"):
                print("%s: spawn at %.3f, child gone at %.3f\n%s" % (entry, e0, e1, r.stderr), file=sys.stderr)
            if r.returncode != 0:
                print(r.stderr, file=sys.stderr)
                sys.exit(1)
            tail = [l for l in r.stdout.splitlines() if "bundles" in l or "stage busy" in l or "release of" in l]
            key = "level%s" % entry
            while key in res:
                key += "'"
            ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            import xxhash
            hx = xxhash.xxh64()
            with open(out_bam, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    hx.update(blk)
            rec_hash = None
            if os.environ.get("CLI_VERIFY"):   # the inflated record stream (whatever the bundle cuts and the BGZF framing were)
                L.br_bgzf_read_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
                L.br_free_buffer.argtypes = [C.c_void_p]
                pp, nn = C.c_void_p(), C.c_uint64()
                assert L.br_bgzf_read_file(out_bam.encode(), args.threads, C.byref(pp), C.byref(nn)) == 0
                view = (C.c_uint8 * nn.value).from_address(pp.value)
                hr = xxhash.xxh64()
                mv = memoryview(view)
                for q in range(0, nn.value, 1 << 26):
                    hr.update(mv[q:q + (1 << 26)])
                rec_hash = hr.hexdigest()
                del mv, view
                L.br_free_buffer(pp)
            res[key] = {"out_xxh64": hx.hexdigest(), "inflated_xxh64": rec_hash, "wall_s": round(wall, 2), "user_s": round(ru1.ru_utime - ru0.ru_utime, 2), "sys_s": round(ru1.ru_stime - ru0.ru_stime, 2),
                        "max_rss_gb": round(ru1.ru_maxrss / 1048576.0, 2), "alignments_per_s": len(rlen) / wall, "out_bam_bytes": os.path.getsize(out_bam),
                                       "report": " | ".join(tail)}
        print(json.dumps({"config": "cli", "workload": "%d paired-end alignments, BAM file -> BAM file, -p %d, GENCODE-shaped GTF (%d transcripts)" % (len(rlen), args.threads, len(annd["transcripts"])),
                          "in_bam_bytes": os.path.getsize(in_bam), "uncompressed_in_bytes": int(stream_h.size), "results": res,
                          "input_prep_s": round(prep, 1)}))
        return
    if args.config == "c5":
        n = args.reads or 1_000_000
        ann = synth.Annotation("G")
        batch = ann.reads(n, "hifi")
        cfg = lib.make_config(lr_hq=1, strict=1, sim_thr=0.95)
        label = "%d HiFi-like reads (median 2 kb) --lr-hq --strict --similarity-threshold 0.95 vs GENCODE-shaped annotation" % n
    else:
        n = args.reads or 200_000
        ann = synth.Annotation("G", n_genes=6000, n_refs=5, with_genome=True)
        batch = ann.reads(n, "ont", with_seq=1)
        cfg = lib.make_config(lr=1, use_fasta=1)
        label = "%d ONT-like reads (median 900 bp, clips <= 300) --lr -S vs a 6000-gene annotation with synthetic genome" % n
    idx = lib.Index.from_flat(ann.flat, device=0)
    ctx = lib.Context(idx)
    if os.environ.get("KSW_FAST") is not None:
        ctx.set_param("ksw_fast", int(os.environ["KSW_FAST"]))
    if os.environ.get("KSW_TAPE_MB") is not None:
        ctx.set_param("ksw_tape_mb", int(os.environ["KSW_TAPE_MB"]))
    db = brdev.upload_batch(batch, "cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        rows = ctx.project_batch_device(cfg, db, stream)
    ctx.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = {}
    for _ in range(args.steps):
        rows = ctx.project_batch_device(cfg, db, stream)
        for k, (ms, ln) in ctx.kernel_ms().items():
            kms[k] = kms.get(k, 0.0) + ms
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"config": args.config, "workload": label, "alignments_per_step": int(batch["n_aln"]),
           "value": batch["n_aln"] * args.steps / el, "unit": "alignments/s", "ms_per_step": 1e3 * el / args.steps,
           "projected_records": int(rows.n_rows), "matches": int(rows.n_matches),
           "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in kms.items() if v}}
    if args.config == "c3":
        st = ctx.rescue_stats()
        ksw_ms = kms.get("k_ksw", 0.0) / args.steps
        out["rescue"] = st
        out["ksw_routing"] = ctx.ksw_diag()
        out["k_ksw_GCUPS"] = st["dp_cells"] / (ksw_ms * 1e-3) / 1e9 if ksw_ms else None
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
