"""The match-join's heavy blocks (gs_join.hip "clusters"): request batches that hold many near-identical queries - many isolates of a few
species against a database that holds those species hundreds of times over, the workload gsearch is run on (/root/reference/README.md:134) -
must give the same DistHamming counts (anndists DistHamming::eval, /root/reference/src/dna/dnasketch.rs:72) as the oracle for EVERY
(query, node) pair, whichever way a pair's counter was produced: match by match (phase 0 + main pass + queued expansion of shared
entries) or overwritten by the compare tile kernel over its block."""
import os

import numpy as np
import pytest

import helpers as H
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _family_db(rng, n_roots, per, m, dtype, universe):
    """rows of `n_roots` families; values from a small universe so that unrelated rows agree by chance (~m / universe slots per pair)"""
    def rnd(shape):
        v = rng.integers(0, universe, shape)
        return v.astype(np.float32) if np.dtype(dtype) == np.float32 else v.astype(dtype)
    roots = rnd((n_roots, m))
    db = np.repeat(roots, per, axis=0)
    J = rng.uniform(0.25, 0.95, (len(db), 1))
    mk = rng.random(db.shape) > J
    db[mk] = rnd(db.shape)[mk]
    return np.ascontiguousarray(db[rng.permutation(len(db))]), roots


def _isolated_graph(n, M):
    return dict(levels=np.zeros(n, np.uint8), entry=0, deg0=np.zeros(n, np.uint32), nbr0=np.zeros((n, 2 * M), np.uint32), cnt0=np.zeros((n, 2 * M), np.uint32),
                upidx=np.full(n, -1, np.int32), n_upper=0)


def _oracle_counts(q, db, m):
    d = O.hamming_qxc(q, db, nthreads=os.cpu_count())
    return np.rint(d.astype(np.float64) * m).astype(np.uint16)


@pytest.mark.parametrize("dtype,m,universe", [(np.float32, 800, 1600), (np.uint32, 800, 1 << 30), (np.uint64, 768, 1500), (np.float32, 1100, 1 << 22)])
def test_count_matrix_of_redundant_batches(gpu_ctx, monkeypatch, capfd, dtype, m, universe):
    import gsearch_amd as G
    rng = np.random.default_rng(m + universe % 97)
    db, roots = _family_db(rng, 12, 700, m, dtype, universe)              # 8400 nodes
    n = len(db)
    # 640 queries: 5 families x ~110 isolates, two pairs of twins, 60 unrelated rows, NaN / -0 / +0 slots for f32
    fam = rng.integers(0, 5, 550)
    q = roots[fam].copy()
    Jq = rng.uniform(0.2, 0.98, (len(q), 1))
    mk = rng.random(q.shape) > Jq
    fresh = rng.integers(0, universe, q.shape)
    q[mk] = (fresh.astype(np.float32) if np.dtype(dtype) == np.float32 else fresh.astype(dtype))[mk]
    un = rng.integers(0, universe, (60, m))
    q = np.concatenate([q, db[[5, 5, 77, 77]], un.astype(np.float32) if np.dtype(dtype) == np.float32 else un.astype(dtype), db[rng.integers(0, n, 26)]])
    if np.dtype(dtype) == np.float32:
        q[3, :40] = np.nan; q[4, 40:60] = -0.0; db[9, 40:60] = 0.0; db[10, :8] = np.nan
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    want = _oracle_counts(q, db, m)
    hn = G.Hnsw.new(8, n, 16, 16, G.DistHamming(), dtype=dtype, seed=1)
    hn.import_graph(db, _isolated_graph(n, 8))
    monkeypatch.setenv("GS_JOIN_VERBOSE", "1")
    capfd.readouterr()
    hn.search_stats(reset=True)
    got = hn.count_matrix(q)
    err = capfd.readouterr().err
    assert "clusters" in err and " 0 clusters" not in err, err           # the heavy-block path ran and found the families
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
    if universe <= 2000 and np.dtype(dtype).itemsize == 4:
        assert hn.search_stats()["join_shared_expansions"] > 0            # chance matches on shared entries were expanded over cluster members
    for env in ({"GS_JOIN_CLUSTER": "0"}, {"GS_DENSE_IMPL": "tile"}):     # the plain join and the compare tile kernel: the same matrix
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(hn.count_matrix(q), want), env
        for k in env:
            monkeypatch.delenv(k)
    hn.close()


@pytest.mark.parametrize("dtype,m", [(np.float32, 800), (np.uint32, 1000), (np.float32, 802)])
def test_count_matrix_on_a_skewed_database_of_small_clusters(gpu_ctx, monkeypatch, capfd, dtype, m):
    """round 6, skewed databases: families of 40 .. 1500 genomes with 2 .. 16 isolates each in the batch (clusters by related PAIRS, gs_join.hip) - the regime in which most
    nodes belong to a cluster and most blocks are a few query rows high. Exercises the in-place own-cluster test of k_match_join (>= 10 % of the nodes clustered), shared
    table entries for every cluster (GS_JOIN_DEDUP_MINQ default 2), and k_hamming_thin (blocks of <= 16 query rows; m = 802: rows not 16-byte aligned, the tile kernel's
    THIN path takes them). The count matrix must equal the oracle's for every pair, and stay the same with each of the three switched off."""
    import gsearch_amd as G
    rng = np.random.default_rng(m)
    universe = 1500
    def rnd(shape):
        v = rng.integers(0, universe, shape)
        return v.astype(np.float32) if np.dtype(dtype) == np.float32 else v.astype(dtype)
    sizes = [1500, 1300, 1000, 800] + list(rng.integers(80, 260, 34))       # (the cluster path wants >= 8192 nodes and >= 256 queries)
    isolates = [16, 9, 5, 3] + list(rng.integers(2, 9, 33)) + [40]        # the last family: a block 40 rows high (not thin)
    roots = rnd((len(sizes), m))
    db = np.repeat(roots, sizes, axis=0)
    mk = rng.random(db.shape) > rng.uniform(0.3, 0.95, (len(db), 1))
    db[mk] = rnd(db.shape)[mk]
    q = np.repeat(roots, isolates, axis=0)
    mk = rng.random(q.shape) > rng.uniform(0.3, 0.98, (len(q), 1))
    q[mk] = rnd(q.shape)[mk]
    q = np.concatenate([q, rnd((60, m)), db[[3, 3, 2000]]])              # unrelated rows, twins of a node
    assert len(db) >= 8192 and len(q) >= 256
    if np.dtype(dtype) == np.float32:
        q[1, :30] = np.nan; q[2, 30:50] = -0.0; db[7, 30:50] = 0.0; db[8, :6] = np.nan
    db = np.ascontiguousarray(db[rng.permutation(len(db))]); q = np.ascontiguousarray(q[rng.permutation(len(q))])
    n = len(db)
    want = _oracle_counts(q, db, m)
    hn = G.Hnsw.new(8, n, 16, 16, G.DistHamming(), dtype=dtype, seed=1)
    hn.import_graph(db, _isolated_graph(n, 8))
    monkeypatch.setenv("GS_JOIN_VERBOSE", "1")
    monkeypatch.setenv("GS_JOIN_CLUSTER_MIN", "0")
    capfd.readouterr()
    got = hn.count_matrix(q)
    err = capfd.readouterr().err
    assert "clusters" in err and " 0 clusters" not in err, err
    ncl = int(err.split(" clusters,")[0].split()[-1])
    assert ncl >= 20, err                                                  # the small families became clusters (pair rule), not only the 40-isolate one
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
    for env in ({"GS_JOIN_INPLACE": "0"}, {"GS_BLOCKS_THIN_OFF": "1"}, {"GS_JOIN_DEDUP_MINQ": "12"}, {"GS_JOIN_INPLACE": "1", "GS_JOIN_DEDUP_MINQ": "5"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(hn.count_matrix(q), want), env
        for k in env:
            monkeypatch.delenv(k)
    hn.close()


def test_heavy_blocks_with_everything_related(gpu_ctx, monkeypatch, capfd):
    """degenerate batches: every query related to every node (one giant component), and a value band so narrow that chance matches alone
    make pairs heavy (random merges of unrelated components) - any labelling must give the oracle's counts"""
    import gsearch_amd as G
    rng = np.random.default_rng(8)
    m = 768
    monkeypatch.setenv("GS_JOIN_VERBOSE", "1")
    for universe, n_roots in ((1 << 20, 1), (300, 6)):
        db, roots = _family_db(rng, n_roots, 8400 // n_roots, m, np.float32, universe)
        q = db[rng.integers(0, len(db), 300)].copy()
        mk = rng.random(q.shape) < 0.3
        q[mk] = rng.integers(0, universe, q.shape).astype(np.float32)[mk]
        hn = G.Hnsw.new(8, len(db), 16, 16, G.DistHamming(), seed=1)
        hn.import_graph(db, _isolated_graph(len(db), 8))
        got = hn.count_matrix(q)
        assert np.array_equal(got, _oracle_counts(q, db, m)), universe
        hn.close()


@pytest.mark.parametrize("cluster", ["2", "0"])
def test_search_of_redundant_batch_matches_oracle(gpu_ctx, monkeypatch, cluster):
    """the whole dense search (match-join with heavy blocks -> dense traversal) on a redundant request against a built graph: ids, distances,
    counts and evaluation counts == oracle.parallel_search; GS_JOIN_CLUSTER=2 forces the heavy-block path on this small shape, 0 disables it"""
    import gsearch_amd as G
    monkeypatch.setenv("GS_DIST_MODE", "dense")
    monkeypatch.setenv("GS_JOIN_CLUSTER", cluster)
    m = 256
    db = H.synth_sig_db(6, 900, m, 17, jlo=0.3, jhi=0.97)               # 5400 nodes, six families
    oix = O.Index(np.float32, m, 8, 40, seed=12)
    oix.parallel_insert(db, batch=200)
    hn = G.Hnsw.new(8, 100000, 16, 40, G.DistHamming(), seed=12, insert_batch=200)
    hn.set_extend_candidates(True)
    hn.parallel_insert(db)
    rng = np.random.default_rng(3)
    base = db[rng.integers(0, 40, 700)].copy()                          # 700 queries drawn from 40 nodes: many twins and siblings
    mk = rng.random(base.shape) < 0.15
    base[mk] = rng.random(base.shape, dtype=np.float32)[mk]
    got, want = hn.search_arrays(base, 10, 80), oix.parallel_search(base, 10, 80, nthreads=os.cpu_count())
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_count_matrix_differential_run_over_random_redundant_and_skewed_shapes():
    """tools/join_fuzz.py (round 6): the count matrix of a request batch against the oracle over random element types, sketch sizes (aligned rows or not), family-size laws,
    isolates per family, value universes and join knobs (in-place own-cluster test on / off, shared-entry threshold, thin blocks off, pair bar, cluster rule). 14 cases, bounded."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-u", os.path.join(root, "tools", "join_fuzz.py"), "14", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "14 cases, 0 mismatches" in out.stdout, out.stdout[-1500:]
