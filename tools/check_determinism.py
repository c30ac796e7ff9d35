#!/usr/bin/env python3
"""build the same synthetic-genome DB twice, compare graphs; search in the three evaluation modes, compare everything"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsearch_amd as G
from gsearch_amd import _lib
N, NQ, L, k, m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 1024, 1_000_000, 21, 18000
ctx = G.Context(0); lib = ctx.L
prm = G.SeqSketcherParams(k, m, "optdens")
words = (L + 31) // 32; gb = words * 8
def sketch(first, n):
    d_seq, d_sig = ctx.alloc(n * gb + 64), ctx.alloc(n * m * 4)
    rs = np.arange(n, dtype=np.uint64) * np.uint64(words * 32)
    d_rs, d_rl, d_go = ctx.alloc(8 * n), ctx.alloc(8 * n), ctx.alloc(8 * (n + 1))
    ctx.upload(d_rs, rs); ctx.upload(d_rl, np.full(n, L, np.uint64)); ctx.upload(d_go, np.arange(n + 1, dtype=np.uint64))
    _lib.check(lib.gs_synth_dna_family_dev(ctx.h, 7, first, n, L, max(N // 100, 1), 0.001, 0.08, d_seq))
    _lib.check(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, n * gb + 64, d_rs, d_rl, n, d_go, n, d_sig))
    out = ctx.download(d_sig, (n, m), np.float32)
    for p in (d_seq, d_sig, d_rs, d_rl, d_go): ctx.free(p)
    return out
db = np.concatenate([sketch(g0, min(4000, N - g0)) for g0 in range(0, N, 4000)])
q = sketch(10**9, NQ)
graphs = []
for rep in range(2):
    hn = G.Hnsw.new(128, 1_500_000, 16, 1600, G.DistHamming(ctx), seed=3, insert_batch=256, ctx=ctx)
    hn.modify_level_scale(0.25); hn.set_extend_candidates(True); hn.set_keeping_pruned(False)
    for g0 in range(0, N, 8192): hn.parallel_insert(db[g0:g0 + 8192])
    graphs.append(hn.export_graph())
    if rep == 0: hn0 = hn
same = all(np.array_equal(graphs[0][k2], graphs[1][k2]) for k2 in ("deg0", "levels")) and all(
    np.array_equal(graphs[0]["nbr0"][i, :graphs[0]["deg0"][i]], graphs[1]["nbr0"][i, :graphs[1]["deg0"][i]]) for i in range(N))
print("two builds identical:", same, " mean deg0 %.1f max %d" % (graphs[0]["deg0"].mean(), graphs[0]["deg0"].max()))
res = {}
for mode, legacy in (("gather", ""), ("dense", "1"), ("dense", "")):
    os.environ["GS_DIST_MODE"] = mode
    if legacy: os.environ["GS_DENSE_LEGACY"] = "1"
    else: os.environ.pop("GS_DENSE_LEGACY", None)
    res[mode + legacy] = hn0.search_arrays(q, 50, 5000)
base = res["gather"]
for kx, r in res.items():
    print(kx, "ids", np.array_equal(r[0], base[0]), "dist", np.array_equal(r[1], base[1]), "evals", np.array_equal(r[3], base[3]), "mean evals %.2f" % r[3].mean())
