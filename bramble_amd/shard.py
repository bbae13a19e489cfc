"""Batch sharding for multi-GPU runs: contiguous ranges of read-name
groups per rank, never splitting a group (mates and multi-mappers stay together;
the reference cuts bundles at name boundaries too, src/bramble.cpp:396-398).
There is no collective on the data path: each rank projects its shard against its
own index replica; results concatenate in rank order."""
import numpy as np


def group_starts(batch):
    """Indices where a new read name starts (plus n_aln)."""
    n = int(batch["n_aln"])
    off = np.asarray(batch["name_off"], dtype=np.int64)
    names = np.asarray(batch["names"], dtype=np.uint8)
    starts = [0]
    for i in range(1, n):
        a = names[off[i - 1]:off[i]]
        b = names[off[i]:off[i + 1]]
        if len(a) != len(b) or not np.array_equal(a, b):
            starts.append(i)
    starts.append(n)
    return np.array(starts, dtype=np.int64)


def shard_bounds(starts, rank, world):
    """[lo, hi) alignment range of `rank`: groups dealt in contiguous, near-equal chunks."""
    n_groups = len(starts) - 1
    g0 = (n_groups * rank) // world
    g1 = (n_groups * (rank + 1)) // world
    return int(starts[g0]), int(starts[g1])


def shard_batch(batch, rank, world, starts=None):
    if starts is None:
        starts = group_starts(batch)
    lo, hi = shard_bounds(starts, rank, world)
    sub = {"n_aln": hi - lo}
    for k in ("ref_id", "ref_start", "flags", "xs", "ts", "mate_ref_id", "mate_start", "l_qseq"):
        sub[k] = np.asarray(batch[k])[lo:hi]
    for off_key, pool_key in (("cigar_off", "cigar"), ("name_off", "names"), ("seq_off", "seqs")):
        if batch.get(off_key) is None:
            sub[off_key] = None
            sub[pool_key] = None
            continue
        off = np.asarray(batch[off_key], dtype=np.uint64)
        sub[off_key] = off[lo:hi + 1] - off[lo]
        sub[pool_key] = np.asarray(batch[pool_key])[int(off[lo]):int(off[hi])]
    return sub, lo
