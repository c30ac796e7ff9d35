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
