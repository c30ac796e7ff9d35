#!/usr/bin/env python3
"""Differential run of parallel_insert + parallel_search against the oracle over random index shapes: element type, sketch size, M, ef_construction, level scale, family
structure (including one dominant family and exact duplicates), insert batch, several insert calls, knbn / ef, every distance strategy. The device-built graph must equal the
oracle's (levels, degrees, neighbour ids and counts) and every strategy's answers the oracle's search of that graph.  usage: hnsw_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsearch_amd as G
import oracle_lib as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    dtype = [np.float32, np.uint32, np.uint64, np.uint16][int(rng.integers(0, 4))]
    m = int(rng.choice([32, 64, 96, 200, 256, 777]))
    M = int(rng.choice([4, 8, 12, 24, 48]))
    efc = int(rng.choice([M + 1, 2 * M, 2 * M + 1, 40, 100, 200]))
    scale = float(rng.choice([0.25, 0.5, 1.0]))
    ib = int(rng.choice([1, 16, 64, 256]))
    n_roots = int(rng.choice([1, 3, 20, 100]))
    big = bool(os.environ.get("FUZZ_BIG"))
    n = int(rng.choice([50, 300, 1500, 5000] if not big else [9000, 20000]))
    universe = int(rng.choice([4, 50, 2 * m, 1 << 20] if not big else [2 * m, 8 * m, 1 << 20]))
    def rnd(shape):
        v = rng.integers(0, universe, shape)
        return v.astype(np.float32) if dtype == np.float32 else v.astype(dtype)
    roots = rnd((n_roots, m))
    fam = np.minimum((n_roots * rng.random(n) ** float(rng.choice([1.0, 3.0]))).astype(np.int64), n_roots - 1)
    db = roots[fam].copy()
    J = rng.uniform(0.05, 1.0, (n, 1))
    mk = rng.random(db.shape) > J
    db[mk] = rnd(db.shape)[mk]
    if rng.random() < 0.3 and n > 10:
        db[n // 2: n // 2 + 5] = db[0]                         # exact duplicates
    nq = int(rng.choice([1, 7, 64, 300] if not big else [1500, 3000]))
    q = db[rng.integers(0, n, nq)].copy()
    mq = rng.random(q.shape) < 0.2
    q[mq] = rnd(q.shape)[mq]
    knbn = int(rng.choice([1, 5, 10, 50])); ef = int(rng.choice([knbn, 2 * knbn + 3, 100, 500]))
    ext = bool(rng.random() < 0.8)
    # FUZZ_KNOBS=1: the paths larger indexes take, forced on small ones - no dense pair cache (sparse rows, short lists, no level bitmaps), insert batches joined one by one /
    # in small groups, the split / global visited bitmap, the sorted-array layer-0 search of inserts, the sequential traversal without the order-free phase
    knobs = {}
    if os.environ.get("FUZZ_KNOBS"):
        if rng.random() < 0.5: knobs["GS_PAIR_CACHE_GB"] = "0"
        if rng.random() < 0.3: knobs["GS_SPARSE_L"] = str(int(rng.choice([8, 64, 512])))
        if rng.random() < 0.2: knobs["GS_SPARSE_BITMAP_GB"] = "0"
        if rng.random() < 0.3: knobs["GS_INSERT_GROUP"] = str(int(rng.choice([1, 2, 5])))
        if rng.random() < 0.3: knobs["GS_DENSE_VIS"] = str(rng.choice(["split", "global"])); knobs["GS_SPLIT_W"] = "1024"
        if rng.random() < 0.2: knobs["GS_PLAN_PREPASS"] = "0"
        if rng.random() < 0.2: knobs["GS_DENSE_PHASE2"] = "0"
        if rng.random() < 0.2: knobs["GS_JOIN_CLUSTER"] = "2"
        if rng.random() < 0.5: knobs["GS_DIST_MODE"] = "dense"        # (for the INSERTS: the searches below set their own)
    for kk, vv in knobs.items():
        os.environ[kk] = vv
    t0 = time.perf_counter()
    oix = O.Index(dtype, m, M, efc, scale_modify=scale, extend_candidates=ext, seed=case + 3)
    hn = G.Hnsw.new(M, max(n, 1024), 16, efc, G.DistHamming(), dtype=dtype, seed=case + 3, insert_batch=ib)
    hn.modify_level_scale(scale); hn.set_extend_candidates(ext)
    cuts = sorted(set([0, n] + [int(c) // ib * ib for c in rng.integers(0, n, int(rng.integers(0, 3)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            oix.parallel_insert(db[a:b], batch=ib); hn.parallel_insert(db[a:b])
    og, g = oix.export(), hn.export_graph()
    why = []
    if not (np.array_equal(g["levels"], og["levels"]) and np.array_equal(g["deg0"], og["deg0"])):
        why.append("graph levels / degrees")
    else:
        live = np.arange(g["nbr0"].shape[1])[None, :] < og["deg0"][:, None]
        if not (np.array_equal(np.where(live, g["nbr0"], 0), np.where(live, og["nbr0"], 0)) and np.array_equal(np.where(live, g["cnt0"], 0), np.where(live, og["cnt0"], 0))):
            why.append("graph neighbours")
    want = oix.parallel_search(q, knbn, ef, nthreads=os.cpu_count())
    for mode in ("auto", "dense", "gather"):
        if mode == "auto":
            os.environ.pop("GS_DIST_MODE", None)
        else:
            os.environ["GS_DIST_MODE"] = mode
        got = hn.search_arrays(q, knbn, ef)
        for name, x, y in zip(("ids", "distances", "counts", "evaluations"), got, want):
            xv = x.view(np.uint32) if x.dtype == np.float32 else x
            yv = y.view(np.uint32) if y.dtype == np.float32 else y
            if not np.array_equal(xv, yv):
                why.append("%s search %s" % (mode, name))
    os.environ.pop("GS_DIST_MODE", None)
    for kk in knobs:
        os.environ.pop(kk, None)
    hn.close()
    bad += bool(why)
    print("case %2d %s m=%d M=%d efc=%d scale=%.2f ib=%d n=%d roots=%d universe=%d ext=%d nq=%d knbn=%d ef=%d calls=%d: %s (%.1f s)"
          % (case, np.dtype(dtype).name, m, M, efc, scale, ib, n, n_roots, universe, ext, nq, knbn, ef, len(cuts) - 1, "ok" if not why else "MISMATCH " + "; ".join(why), time.perf_counter() - t0)
          + (" knobs " + " ".join("%s=%s" % kv for kv in sorted(knobs.items())) if knobs else ""), flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
