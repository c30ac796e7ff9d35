"""Multi-GPU decomposition of the path: one process per GPU, torch.distributed for the exchange
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* query sharding (north-star choice): DB + graph replicated, queries split into contiguous blocks, one all-gather of the
  fixed-size top-k blocks per batch (k*12 B per query: latency bound, link bandwidth irrelevant);
* DB sharding (the reference's scripts/multiple_search.sh:71-107 idea): every rank answers all queries on its shard,
  all-gather, k-way merge under the (distance, id) order.
"""
import numpy as np


def shard_bounds(n, rank, world):
    """contiguous block [lo, hi) of rank; sizes differ by at most one"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_topk(ids, dist):
    """ids: (nq_local, k) int64 tensor, dist: (nq_local, k) float32 tensor, identical shapes on every rank.
    Returns the concatenation over ranks in rank order (every rank gets it)."""
    import torch
    import torch.distributed as td
    world = td.get_world_size()
    all_ids = torch.empty((world * ids.shape[0], ids.shape[1]), dtype=ids.dtype, device=ids.device)
    all_dist = torch.empty((world * dist.shape[0], dist.shape[1]), dtype=dist.dtype, device=dist.device)
    td.all_gather_into_tensor(all_ids, ids.contiguous())
    td.all_gather_into_tensor(all_dist, dist.contiguous())
    return all_ids, all_dist


def merge_topk_shards(ids_shards, dist_shards, knbn):
    """k-way merge of per-DB-shard answers for the same queries: (S, nq, k) -> (nq, knbn) under (distance, id)."""
    ids = np.concatenate(ids_shards, axis=1).astype(np.uint64)
    dist = np.concatenate(dist_shards, axis=1).astype(np.float32)
    order = np.lexsort((ids, dist), axis=1)[:, :knbn]
    return np.take_along_axis(ids, order, axis=1), np.take_along_axis(dist, order, axis=1)
