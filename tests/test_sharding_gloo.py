"""world_size-2 gloo test of the N>1 path: query sharding + all-gather equals the single-process answer, and the
DB-sharded alternative merges to the exact global top-k. The searcher here is the CPU oracle (the test is about the
decomposition, not the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as td
    import helpers as H
    import oracle_lib as O
    from gsearch_amd import sharding as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    db = H.synth_sig_db(10, 25, 128, 3)
    q = H.queries_from(db, 24, 4, frac=0.2)
    knbn = 7
    # --- query sharding, DB replicated
    ix = O.Index(np.float32, 128, 8, 32, seed=2)
    ix.parallel_insert(db, batch=4)
    lo, hi = S.shard_bounds(len(q), rank, world)
    ids, dist, _, _ = ix.parallel_search(q[lo:hi], knbn, 64)
    all_ids, all_dist = S.allgather_topk(torch.from_numpy(ids.view(np.int64)), torch.from_numpy(dist))
    # --- DB sharding, queries replicated
    dlo, dhi = S.shard_bounds(len(db), rank, world)
    bi, bd = O.bruteforce_topk(db[dlo:dhi], q, knbn)
    bi = bi + np.uint64(dlo)
    g_ids, g_dist = S.allgather_topk(torch.from_numpy(bi.view(np.int64)), torch.from_numpy(bd))
    if rank == 0:
        fids, fdist, _, _ = ix.parallel_search(q, knbn, 64)
        ok1 = np.array_equal(all_ids.numpy().view(np.uint64), fids) and np.array_equal(all_dist.numpy(), fdist)
        nq = len(q)
        sh_i = [g_ids.numpy().view(np.uint64)[r * nq:(r + 1) * nq] for r in range(world)]
        sh_d = [g_dist.numpy()[r * nq:(r + 1) * nq] for r in range(world)]
        mi, md = S.merge_topk_shards(sh_i, sh_d, knbn)
        ei, ed = O.bruteforce_topk(db, q, knbn)
        ok2 = np.array_equal(mi, ei) and np.array_equal(md, ed)
        open(os.path.join(out_dir, "result.txt"), "w").write("%d %d" % (ok1, ok2))
    td.barrier()
    td.destroy_process_group()


def _worker_blocks(rank, world, port, out_dir, nq_total):
    """unequal shards through the C-ABI block layout (gs_topk_pack / gs_topk_unpack - host functions, no device needed) and ONE all-gather"""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as td
    import gsearch_amd as G
    from gsearch_amd import sharding as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    knbn = 5
    lo, hi = S.shard_bounds(nq_total, rank, world)
    nq_max = max(S.shard_bounds(nq_total, r, world)[1] - S.shard_bounds(nq_total, r, world)[0] for r in range(world))
    gq = np.arange(lo, hi, dtype=np.uint64).reshape(-1, 1)
    j = np.arange(knbn, dtype=np.uint64).reshape(1, -1)
    ids = gq * np.uint64(1000) + j                                  # a known function of the GLOBAL query id
    dist = ((gq * np.uint64(8) + j).astype(np.float32) / np.float32(64.0)).astype(np.float32)
    if hi - lo == 0:
        ids, dist = np.zeros((0, knbn), np.uint64), np.zeros((0, knbn), np.float32)
    all_ids, all_dist, counts = S.allgather_topk_blocks(ids, dist, nq_max)
    allq = np.arange(nq_total, dtype=np.uint64).reshape(-1, 1)
    ok = (np.array_equal(all_ids, allq * np.uint64(1000) + j) and np.array_equal(all_dist, ((allq * np.uint64(8) + j).astype(np.float32) / np.float32(64.0)))
          and [int(c) for c in counts] == [S.shard_bounds(nq_total, r, world)[1] - S.shard_bounds(nq_total, r, world)[0] for r in range(world)])
    open(os.path.join(out_dir, "blocks_%d.txt" % rank), "w").write("%d" % ok)
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("world,nq_total", [(2, 7), (3, 10), (3, 2), (3, 9)])
def test_unequal_shards_through_the_block_layout(tmp_path, world, nq_total):
    """the exchange of UNEQUAL query shards (sizes differ by one; with 2 queries on 3 ranks one rank holds none) at world 2 and 3:
    every rank ends with the same compact, rank-ordered answer and the per-rank counts"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_blocks, args=(world, port, str(tmp_path), nq_total), nprocs=world, join=True)
    assert [open(os.path.join(str(tmp_path), "blocks_%d.txt" % r)).read() for r in range(world)] == ["1"] * world


def test_block_layout_rejects_foreign_blocks():
    import gsearch_amd as G
    ids = np.arange(6, dtype=np.uint64).reshape(2, 3); dist = np.ones((2, 3), np.float32)
    b = G.topk_pack(ids, dist, 4)
    assert len(b) == G.topk_block_bytes(4, 3) and len(b) % 16 == 0
    i2, d2, c = G.topk_unpack(np.concatenate([b, b]), 2, 4, 3)
    assert np.array_equal(i2, np.concatenate([ids, ids])) and np.array_equal(d2, np.concatenate([dist, dist])) and list(c) == [2, 2]
    with pytest.raises(G.GsError):
        G.topk_unpack(np.concatenate([b, b]), 2, 4, 4)             # another knbn
    bad = b.copy(); bad[12] ^= 0xFF
    with pytest.raises(G.GsError):
        G.topk_unpack(bad, 1, 4, 3)                                 # not a block
    with pytest.raises(G.GsError):
        G.topk_pack(np.zeros((5, 3), np.uint64), np.zeros((5, 3), np.float32), 4)   # more rows than nq_max


def test_query_and_db_sharding_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), "result.txt")).read() == "1 1"


def test_shard_bounds_cover_everything():
    from gsearch_amd import sharding as S
    for n in (0, 1, 7, 10000):
        for w in (1, 2, 3, 8):
            b = [S.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
