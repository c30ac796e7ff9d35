"""gs_comm_* (the C-ABI form of the path's one exchange step) with MORE than one RCCL rank: one process per GPU, unique id handed over
through a file, every rank searches its own block of the query batch on its replica and all-gathers the top-k; the gathered answer must
equal the single-rank answer for the whole batch. Needs >= 2 visible GPUs (skipped on the 1-GPU gpurun boxes; runs on a multi-GPU node)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import gsearch_amd as G
    import helpers as H
    rank, world, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    ctx = G.Context(rank)
    if rank == 0:
        uid = G.Comm.unique_id()
        open(idfile + ".tmp", "wb").write(bytes(uid)); os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile): break
            time.sleep(0.1)
        uid = open(idfile, "rb").read()
    comm = G.Comm(ctx, world, rank, uid)
    assert comm.rank == rank and comm.n_ranks == world
    db = H.synth_sig_db(12, 40, 200, 3, jlo=0.05, jhi=0.9)
    q = H.queries_from(db, 8 * world, 5, frac=0.2)
    hn = G.Hnsw.new(8, 10000, 16, 40, G.DistHamming(ctx), seed=3, insert_batch=64, ctx=ctx)
    hn.set_extend_candidates(True)
    hn.parallel_insert(db)                                  # DB + graph replicated per GPU (deterministic build: identical replicas)
    nq, knbn = 8, 10
    ids, dist, cnt, ev = hn.search_arrays(q[rank * nq:(rank + 1) * nq], knbn, 60)
    d_ids, d_dist = ctx.alloc(ids.nbytes), ctx.alloc(dist.nbytes)
    d_aids, d_adist = ctx.alloc(ids.nbytes * world), ctx.alloc(dist.nbytes * world)
    ctx.upload(d_ids, ids); ctx.upload(d_dist, dist)
    comm.allgather_topk_dev(d_ids, d_dist, nq, knbn, d_aids, d_adist)
    all_ids = ctx.download(d_aids, (world * nq, knbn), np.uint64); all_dist = ctx.download(d_adist, (world * nq, knbn), np.float32)
    want_ids, want_dist, _, _ = hn.search_arrays(q, knbn, 60)            # the single-rank answer for the whole batch
    ok = bool(np.array_equal(all_ids, want_ids) and np.array_equal(all_dist, want_dist))
    # unequal shards (round 5): the same batch minus its last three queries, contiguous blocks that differ by one, through gs_comm_allgatherv_topk_dev
    from gsearch_amd import sharding as S
    nqt = nq * world - 3
    lo, hi = S.shard_bounds(nqt, rank, world)
    nq_max = max(S.shard_bounds(nqt, r, world)[1] - S.shard_bounds(nqt, r, world)[0] for r in range(world))
    ids2, dist2, _, _ = hn.search_arrays(q[lo:hi], knbn, 60)
    ctx.upload(d_ids, ids2); ctx.upload(d_dist, dist2)
    counts = comm.allgatherv_topk_dev(d_ids, d_dist, hi - lo, nq_max, knbn, d_aids, d_adist)
    g_ids = ctx.download(d_aids, (nqt, knbn), np.uint64); g_dist = ctx.download(d_adist, (nqt, knbn), np.float32)
    ok = ok and bool(np.array_equal(g_ids, want_ids[:nqt]) and np.array_equal(g_dist, want_dist[:nqt]) and int(counts.sum()) == nqt and comm.size() == world)
    open(outfile, "w").write("ok" if ok else "MISMATCH")
    comm.close()
""")


@pytest.mark.gpu
def test_comm_allgather_two_or_more_ranks(tmp_path):
    import torch
    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs >= 2 GPUs: RCCL refuses two ranks on one device (the 1-GPU boxes run test_comm_allgather_single_rank)")
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT % {"root": ROOT})
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(tmp_path / "uid"), str(tmp_path / ("out%d" % r))], env=env) for r in range(world)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0] * world, rcs
    assert [(tmp_path / ("out%d" % r)).read_text() for r in range(world)] == ["ok"] * world
