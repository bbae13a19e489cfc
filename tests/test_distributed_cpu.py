"""N>1 path: world_size-2 gloo.  Each rank takes its shard of read-name groups
(bramble_amd.shard), projects it independently and the concatenation over ranks must equal
the unsharded result -- the property that makes the path collective-free.  On a box without
a GPU the oracle stands in for the per-rank device (test_sharded_equals_unsharded); on the GPU
box the ranks call the HIP path through the C ABI (test_sharded_hip_ranks_equal_unsharded_oracle,
marked gpu: both ranks share the one card, each with its own index replica and context)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, use_hip=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bramble_amd import shard, synth
    ann = synth.Annotation("S")
    batch = ann.reads(3000, "pe")
    sub, lo = shard.shard_batch(batch, rank, world)
    if use_hip:
        from bramble_amd import lib
        ndev = max(torch.cuda.device_count(), 1)
        idx = lib.Index(ann.as_dict(), device=rank % ndev)   # this rank's index replica
        ctx = lib.Context(idx)
        w = ctx.project_batch(lib.make_config(), sub)
        rows = {"n_rows": w["n_rows"], "tid": w["transcript_id"], "pos": w["pos"], "nh": w["nh"], "input_index": w["input_index"],
                "cigar": w["cigar"], "total_unique": w["total_unique"], "dropped_reads": w["dropped_reads"]}
        ctx.close()
        idx.close()
    else:
        from oracle import oracle_binding as ob
        rows, _, _ = ob.run(ob.OracleIndex(ann.as_dict()), ob.make_flags(), sub, want_matches=False)
    # the only cross-rank traffic: counters (5 integers summed) -- never the data path
    cnt = torch.tensor([rows["n_rows"], rows["total_unique"], rows["dropped_reads"], sub["n_aln"]], dtype=torch.int64)
    dist.all_reduce(cnt)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, tid=rows["tid"], pos=rows["pos"], nh=rows["nh"],
             input_index=rows["input_index"] + lo, cigar=rows["cigar"], total=cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _check_sharded(tmp_path, use_hip):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), use_hip), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from bramble_amd import synth
    from oracle import oracle_binding as ob
    ann = synth.Annotation("S")
    batch = ann.reads(3000, "pe")
    full, _, _ = ob.run(ob.OracleIndex(ann.as_dict()), ob.make_flags(), batch, want_matches=False)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for key, fk in (("tid", "tid"), ("pos", "pos"), ("nh", "nh"), ("input_index", "input_index"), ("cigar", "cigar")):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), full[fk]), key
    assert parts[0]["total"].tolist() == [full["n_rows"], full["total_unique"], full["dropped_reads"], batch["n_aln"]]


def test_sharded_equals_unsharded(tmp_path):
    _check_sharded(tmp_path, False)


@pytest.mark.gpu
def test_sharded_hip_ranks_equal_unsharded_oracle(tmp_path):
    _check_sharded(tmp_path, True)


def test_shards_never_split_a_name_group():
    sys.path.insert(0, ROOT)
    from bramble_amd import shard, synth
    batch = synth.Annotation("S").reads(500, "pe")
    starts = shard.group_starts(batch)
    for world in (2, 3, 8):
        prev = 0
        for r in range(world):
            lo, hi = shard.shard_bounds(starts, r, world)
            assert lo == prev and lo in starts and hi in starts
            prev = hi
        assert prev == batch["n_aln"]


def _run_bench(extra, env_extra=None, timeout=600):
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_without_launcher_fails_loudly_when_a_rank_fails():
    """`python bench.py --gpus 2` starts its own ranks (no external torchrun); on a box without a GPU every rank
    fails, and the command must end non-zero without a result line instead of hanging or printing a number."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = _run_bench(["--gpus", "2", "--pairs", "1000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pcie"],
                   {"BENCH_DIST_BACKEND": "gloo"})
    assert r.returncode != 0
    assert '"metric"' not in r.stdout


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """one command, N ranks.  Two ranks share the one card of the test box (gloo for the barrier and
    the reductions); the line must say n_gpus 2 = n_ranks_seen 2, carry both ranks' times, and value = both ranks'
    alignments over the slowest rank's time."""
    import json
    r = _run_bench(["--gpus", "2", "--pairs", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                   {"BENCH_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and len(d["per_rank_ms_per_step"]) == 2
    assert d["scaling"] == "weak" and d["pcie_inclusive"] is not None
    total = sum(d["per_rank_alignments_per_step"])
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert abs(d["ms_per_step"] - max(d["per_rank_ms_per_step"])) <= 1e-3 * d["ms_per_step"] + 1e-3
