"""hnsw_rs' DataId / PointId on the Hnsw surface: parallel_insert(&[(&Vec<Sig>, usize)]) takes the caller's ids
(/root/reference/src/dna/dnasketch.rs:426-435) and Neighbour{d_id, distance, p_id} hands them back (/root/reference/src/answer.rs:42-57 uses
d_id as an index into its seqdict). The oracle numbers nodes 0.. in insertion order; the device index must return the caller's id of the
same node, the PointId (layer, rank in layer) of the same node, and keep both through every dump format."""
import os

import numpy as np
import pytest

import helpers as H
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _build(G, db, ids, split, M=8, efc=40, scale=1.0, seed=21):
    hn = G.Hnsw.new(M, 100000, 16, efc, G.DistHamming(), seed=seed, insert_batch=64)
    hn.modify_level_scale(scale); hn.set_extend_candidates(True); hn.set_keeping_pruned(False)
    hn.parallel_insert(db[:split])                                      # first call: implicit ids 0..split-1
    hn.parallel_insert([(db[i], int(ids[i])) for i in range(split, len(db))])     # then (vector, id) pairs like the reference
    return hn


@pytest.mark.parametrize("mode", ["gather", "dense"])
def test_caller_ids_and_point_ids(gpu_ctx, monkeypatch, tmp_path, mode):
    import gsearch_amd as G
    monkeypatch.setenv("GS_DIST_MODE", mode)
    m, M, efc = 128, 8, 40
    db = H.synth_sig_db(8, 60, m, 5, jlo=0.2, jhi=0.95)
    n, split = len(db), 100
    rng = np.random.default_rng(1)
    ids = np.arange(n, dtype=np.uint64)
    ids[split:] = (10_000_000_000 + 7 * rng.permutation(n - split)).astype(np.uint64)          # beyond 2^32, not monotone
    oix = O.Index(np.float32, m, M, efc, seed=21)
    oix.parallel_insert(db[:split], batch=64)                           # (two calls like the device index: a call's batches start at its first point)
    oix.parallel_insert(db[split:], batch=64)
    og = oix.export()
    hn = _build(G, db, ids, split)
    assert np.array_equal(hn.get_ids(), ids) and np.array_equal(hn.get_ids(split - 2, 5), ids[split - 2:split + 3])
    q = H.queries_from(db, 40, 9, frac=0.2)
    want = oix.parallel_search(q, 10, 60)
    gi, gd, gc, ge, pl, pr = hn.search_arrays_pid(q, 10, 60)
    assert np.array_equal(gd, want[1]) and np.array_equal(gc, want[2]) and np.array_equal(ge, want[3])
    assert np.array_equal(gi, ids[want[0].astype(np.int64)])                                    # d_id = the caller's id of the oracle's node
    lv = og["levels"]
    rank = np.zeros(n, np.int32)
    for L in np.unique(lv):
        rank[lv == L] = np.arange(int((lv == L).sum()))
    assert (lv > 0).any()                                                                       # upper layers exist: PointIds are not all (0, i)
    assert np.array_equal(pl, lv[want[0].astype(np.int64)]) and np.array_equal(pr, rank[want[0].astype(np.int64)])
    nb = hn.parallel_search(q[:2], 10, 60)
    assert nb[0][0].d_id == int(gi[0, 0]) and nb[0][0].p_id == (int(pl[0, 0]), int(pr[0, 0]))
    bi, bd = hn.bruteforce_search(q[:8], 5)
    oi, od = O.bruteforce_topk(db, q[:8], 5)
    assert np.array_equal(bd, od) and np.array_equal(bi, ids[oi.astype(np.int64)])
    # own dump format: ids travel
    p = str(tmp_path / "ix.bin")
    hn.file_dump(p)
    h2 = G.Hnsw.load(p)
    assert np.array_equal(h2.get_ids(), ids)
    for a, b in zip(h2.search_arrays(q, 10, 60), (gi, gd, gc, ge)):
        assert np.array_equal(a, b)
    # import / export carry them through set_ids
    h3 = G.Hnsw.new(M, 100000, 16, efc, G.DistHamming(), seed=21, insert_batch=64)
    h3.import_graph(db, hn.export_graph())
    assert np.array_equal(h3.get_ids(), np.arange(n, dtype=np.uint64))
    h3.set_ids(ids)
    assert np.array_equal(h3.search_arrays(q, 10, 60)[0], gi)
    with pytest.raises(G.GsError):
        h3.set_ids(ids[:-1])


def test_hnswrs_dump_keeps_caller_ids(gpu_ctx, tmp_path):
    """hnsw_rs' own dump holds the DataId of every point: ids that are not 0..n-1 survive dump + load (a single-layer graph, so the node
    order of the reloaded index - the order of the data file - is the insertion order and even ties come back in the same order)"""
    import gsearch_amd as G
    m = 96
    db = H.synth_sig_db(6, 50, m, 11, jlo=0.2, jhi=0.95)
    n = len(db)
    ids = (5_000_000_000 + 3 * np.random.default_rng(2).permutation(n)).astype(np.uint64)
    hn = G.Hnsw.new(8, 100000, 16, 40, G.DistHamming(), seed=3, insert_batch=64)
    hn.modify_level_scale(0.25); hn.set_extend_candidates(True)
    hn.parallel_insert(db, ids=ids)
    assert hn.export_graph()["n_upper"] == 0
    q = H.queries_from(db, 30, 4, frac=0.2)
    want = hn.search_arrays(q, 10, 60)
    assert set(np.unique(want[0])) <= set(ids.tolist())
    base = str(tmp_path / "hnswdump")
    hn.file_dump_hnswrs(base)
    h2 = G.Hnsw.load_hnswrs(base, hint=hn)
    assert np.array_equal(h2.get_ids(), ids)
    for a, b in zip(h2.search_arrays(q, 10, 60), want):
        assert np.array_equal(a, b)
    # and an index with gsearch's own ids (0..n-1) still reloads with implicit ids
    h3 = G.Hnsw.new(8, 100000, 16, 40, G.DistHamming(), seed=3, insert_batch=64)
    h3.modify_level_scale(0.25); h3.set_extend_candidates(True)
    h3.parallel_insert(db)
    h3.file_dump_hnswrs(base + "2")
    h4 = G.Hnsw.load_hnswrs(base + "2", hint=h3)
    assert np.array_equal(h4.get_ids(), np.arange(n, dtype=np.uint64))
    assert np.array_equal(h4.search_arrays(q, 10, 60)[0], h3.search_arrays(q, 10, 60)[0])
