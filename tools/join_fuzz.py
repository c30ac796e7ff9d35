#!/usr/bin/env python3
"""Differential run of the count matrix of a request batch (gs_index_count_matrix: match-join with heavy blocks, or the compare tile kernel) against the oracle's DistHamming
over random REDUNDANT / SKEWED shapes: element type, sketch size (aligned rows or not), family-size law (uniform .. a few huge species), isolates per family in the batch
(clusters of 2 .. hundreds of queries: thin and full blocks, shared entries, the in-place own-cluster test), value universe (chance matches rare .. everywhere), NaN / signed
zeros for f32, and the join's knobs flipped at random. Every (query, node) counter must equal the oracle's.  usage: join_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsearch_amd as G
import oracle_lib as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    dtype = [np.float32, np.float32, np.uint32, np.uint64][int(rng.integers(0, 4))]
    m = int(rng.choice([768, 800, 802, 1000, 1101, 1400]))       # (the heavy-block path wants m >= 16 x 48 slots)
    n_roots = int(rng.choice([6, 40, 150]))
    alpha = float(rng.choice([1.0, 2.5, 4.0]))                   # family = floor(n_roots u^alpha): 1 uniform, 4 a few huge species
    n = int(rng.choice([8192, 9000, 14000, 24000]))
    universe = int(rng.choice([300, 2 * m, 1 << 22]))
    def rnd(shape):
        v = rng.integers(0, universe, shape)
        return v.astype(np.float32) if dtype == np.float32 else v.astype(dtype)
    roots = rnd((n_roots, m))
    fam = np.minimum((n_roots * rng.random(n) ** alpha).astype(np.int64), n_roots - 1)
    db = roots[fam].copy()
    mk = rng.random(db.shape) > rng.uniform(0.2, 0.98, (n, 1))
    db[mk] = rnd(db.shape)[mk]
    nq = int(rng.choice([256, 400, 900, 1500]))
    qlaw = float(rng.choice([1.0, alpha, 6.0]))
    qf = np.minimum((n_roots * rng.random(nq) ** qlaw).astype(np.int64), n_roots - 1)
    q = roots[qf].copy()
    mk = rng.random(q.shape) > rng.uniform(0.2, 0.99, (nq, 1))
    q[mk] = rnd(q.shape)[mk]
    n_un = int(rng.integers(0, nq // 4))
    q[:n_un] = rnd((n_un, m))                                    # unrelated rows
    if n > 100 and nq > 8: q[-4:] = db[[1, 1, 50, 99]]           # twins of nodes
    if dtype == np.float32 and rng.random() < 0.6:
        q[2, :17] = np.nan; q[3, 20:33] = -0.0; db[5, 20:33] = 0.0; db[6, :9] = np.nan
    db = np.ascontiguousarray(db[rng.permutation(n)]); q = np.ascontiguousarray(q[rng.permutation(nq)])
    knobs = {}
    if rng.random() < 0.5: knobs["GS_JOIN_CLUSTER_MIN"] = "0"
    if rng.random() < 0.3: knobs["GS_JOIN_CLUSTER"] = "2"
    if rng.random() < 0.4: knobs["GS_JOIN_INPLACE"] = str(int(rng.integers(0, 2)))
    if rng.random() < 0.4: knobs["GS_JOIN_DEDUP_MINQ"] = str(int(rng.choice([2, 3, 5, 12, 40])))
    if rng.random() < 0.2: knobs["GS_BLOCKS_THIN_OFF"] = "1"
    if rng.random() < 0.3: knobs["GS_JOIN_CLUSTER_MINPAIRS"] = str(int(rng.choice([1, 32, 4096])))
    if rng.random() < 0.15: knobs["GS_JOIN_CHUNK_MAJOR"] = str(int(rng.integers(0, 2)))
    desc = "case %d %s m=%d n=%d roots=%d alpha=%.1f nq=%d qlaw=%.1f universe=%d knobs %s" % (case, np.dtype(dtype).name, m, n, n_roots, alpha, nq, qlaw, universe,
                                                                                                " ".join("%s=%s" % kv for kv in knobs.items()) or "-")
    t0 = time.perf_counter()
    want = np.rint(O.hamming_qxc(q, db, nthreads=os.cpu_count()).astype(np.float64) * m).astype(np.uint16)
    for k, v in knobs.items(): os.environ[k] = v
    try:
        hn = G.Hnsw.new(8, n, 16, 16, G.DistHamming(), dtype=dtype, seed=1)
        hn.import_graph(db, dict(levels=np.zeros(n, np.uint8), entry=0, deg0=np.zeros(n, np.uint32), nbr0=np.zeros((n, 16), np.uint32), cnt0=np.zeros((n, 16), np.uint32),
                                 upidx=np.full(n, -1, np.int32), n_upper=0))
        got = hn.count_matrix(q)
        st = hn.search_stats()
        hn.close()
    finally:
        for k in knobs: os.environ.pop(k, None)
    nbad = int((got != want).sum())
    if nbad:
        bad += 1
        b = np.argwhere(got != want)[:3]
        print(desc + ": %d WRONG counters, e.g. %s got %s want %s" % (nbad, b.tolist(), [int(got[tuple(x)]) for x in b], [int(want[tuple(x)]) for x in b]), flush=True)
    else:
        print(desc + ": ok (%.1f s, atomics %.2e, expansions %s)" % (time.perf_counter() - t0, st.get("join_atomics", 0), st.get("join_shared_expansions", 0)), flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
