"""Multi-GPU decomposition of the path: one process per GPU, torch.distributed for the exchange
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* query sharding (north-star choice): DB + graph replicated, queries split into contiguous blocks, one all-gather of the
  fixed-size top-k blocks per batch (k*12 B per query: latency bound, link bandwidth irrelevant);
* DB sharding (the reference's scripts/multiple_search.sh:71-107 idea): every rank answers all queries on its shard,
  all-gather, k-way merge under the (distance, id) order.
"""
import os
import socket
import subprocess
import sys

import numpy as np


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ensure_launched(n_procs, script, argv, env=None):
    """`python bench.py --gpus N` must start N ranks by itself. When this process is not already a rank of a torch.distributed.run job
    (no RANK in the environment) and n_procs > 1, re-exec `script argv` as N local ranks - one process per GPU, rendezvous on
    127.0.0.1 - wait for them and return their exit code. Returns None when the caller IS a rank (or n_procs <= 1) and should just
    carry on. The conceptual ancestor is the per-shard loop of scripts/multiple_search.sh:71-107."""
    if n_procs <= 1 or "RANK" in os.environ:
        return None
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_procs)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_procs), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=e)


def rank_env():
    """(rank, world, local_rank) of this process under torch.distributed.run; (0, 1, 0) otherwise"""
    rank = int(os.environ.get("RANK", "0"))
    return rank, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", str(rank)))


class TopkExchange:
    """Per-rank answer block of one query batch - ids (i64) and distances (f32) of nq x knbn neighbours - laid out in ONE byte buffer
    so that the exchange is ONE all-gather per batch (nq*knbn*12 B per rank, latency bound). `ids` / `dist` are views into
    the send buffer: the search writes its results straight into them."""

    def __init__(self, nq, knbn, world, device):
        import torch
        self.nq, self.knbn, self.world = nq, knbn, world
        self.nb_ids = nq * knbn * 8
        self.block = self.nb_ids + nq * knbn * 4
        self.send = torch.zeros(self.block, dtype=torch.uint8, device=device)
        self.ids = self.send[: self.nb_ids].view(torch.int64).view(nq, knbn)
        self.dist = self.send[self.nb_ids:].view(torch.float32).view(nq, knbn)
        self.recv = torch.zeros(world * self.block, dtype=torch.uint8, device=device) if world > 1 else self.send

    def exchange(self):
        """the ONE collective of a step (RCCL all-gather of the packed blocks; a no-op for a single rank)"""
        if self.world > 1:
            import torch.distributed as td
            td.all_gather_into_tensor(self.recv, self.send)

    def gathered(self):
        """(all_ids, all_dist) of shape (world*nq, knbn) in rank order, unpacked from the receive buffer"""
        import torch
        blocks = self.recv.view(self.world, self.block)
        ids = blocks[:, : self.nb_ids].contiguous().view(torch.int64).view(self.world * self.nq, self.knbn)
        dist = blocks[:, self.nb_ids:].contiguous().view(torch.float32).view(self.world * self.nq, self.knbn)
        return ids, dist


class LibExchange:
    """The same one-collective exchange through the LIBRARY's communicator (gs_comm_*: RCCL loaded by the C-ABI library, pack kernel -> one
    ncclAllGather of fixed-size blocks -> unpack kernel) - what a host that is not Python calls, and what `bench.py --gpus N` times. Shards may be
    unequal (`nq_max` = the largest). torch only carries the 128-byte unique id from rank 0 to the others once, at set-up."""

    def __init__(self, ctx, nq_local, nq_max, knbn, world, rank, device):
        import torch
        import torch.distributed as td
        import gsearch_amd as G
        self.ctx, self.nq, self.nq_max, self.knbn, self.world, self.rank = ctx, nq_local, nq_max, knbn, world, rank
        uid = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(G.Comm.unique_id()), dtype=torch.uint8))
        if world > 1:
            td.broadcast(uid, src=0)
        self.comm = G.Comm(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))
        self.ids = torch.zeros((max(nq_local, 1), knbn), dtype=torch.int64, device=device)[:nq_local]
        self.dist = torch.zeros((max(nq_local, 1), knbn), dtype=torch.float32, device=device)[:nq_local]
        self.all_ids = torch.zeros((world * nq_max, knbn), dtype=torch.int64, device=device)
        self.all_dist = torch.zeros((world * nq_max, knbn), dtype=torch.float32, device=device)
        self.counts = None

    def exchange(self):
        """queued on the library's stream, no host round trip per step (round 6): the next step's kernels follow it in stream order"""
        self.comm.allgatherv_topk_async_dev(self.ids.data_ptr(), self.dist.data_ptr(), self.nq, self.nq_max, self.knbn, self.all_ids.data_ptr(), self.all_dist.data_ptr())
        self.counts = None

    def wait(self):
        if self.counts is None:
            self.counts = self.comm.wait()
        return self.counts

    def gathered(self):
        tot = int(self.wait().sum())
        return self.all_ids[:tot], self.all_dist[:tot]

    def ranks_seen(self):
        return self.comm.size()


def allgather_topk_blocks(ids, dist, nq_max):
    """unequal shards over ANY torch.distributed backend (the gloo tests; a host with its own transport does the same with gs_topk_pack / gs_topk_unpack):
    this rank's (nq_local, k) answers -> (compact ids, compact distances, counts per rank), rank order, through ONE all_gather of fixed-size blocks"""
    import torch
    import torch.distributed as td
    import gsearch_amd as G
    world = td.get_world_size()
    knbn = ids.shape[1]
    block = torch.from_numpy(G.topk_pack(ids, dist, nq_max))
    recv = torch.empty(world * block.numel(), dtype=torch.uint8)
    td.all_gather_into_tensor(recv, block)
    return G.topk_unpack(recv.numpy(), world, nq_max, knbn)


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
