#!/usr/bin/env python3
"""A/B of traversal variants on ONE box: build a sketch-level synthetic DB on the device once, then run the same request under each
environment variant given on the command line ("K=V,K=V" ...; "" = defaults), check that ids / distances / evaluation counts are
identical to the first variant and print the kernel times (HIP events per family).
usage: trav_ab.py [--n 300000] [--nq 10000] [--reps 2] "GS_DENSE_PIPE=2" "GS_DENSE_PIPE=4" ..."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsearch_amd as G
from gsearch_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=300000)
ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--m", type=int, default=18000)
ap.add_argument("--M", type=int, default=128)
ap.add_argument("--efc", type=int, default=1600)
ap.add_argument("--ef", type=int, default=5000)
ap.add_argument("--knbn", type=int, default=50)
ap.add_argument("--per-root", type=int, default=100)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--genomes", action="store_true", help="database and queries sketched from synthetic genomes like bench.py (representative match statistics)")
ap.add_argument("--genome-len", type=int, default=5_000_000)
ap.add_argument("--query-roots", type=int, default=0, help="draw the queries from the first R roots only (0 = all): the redundant regime, many isolates of a few species")
ap.add_argument("variants", nargs="*", default=[""])
a = ap.parse_args()

ctx = G.Context(0)
L = ctx.L
n_roots = max(a.n // a.per_root, 1)
hn = G.Hnsw.new(a.M, 1_500_000, 16, a.efc, G.DistHamming(ctx), seed=1, insert_batch=256, ctx=ctx)
hn.modify_level_scale(0.25); hn.set_extend_candidates(True); hn.set_keeping_pruned(False)
hn._ensure(a.m)
import ctypes as C
chunk = 8192 if a.genomes else 25600
d_rows = ctx.alloc(chunk * a.m * 4)
d_q = ctx.alloc(a.nq * a.m * 4)
t0 = time.perf_counter()
if a.genomes:
    Lg = a.genome_len; words = (Lg + 31) // 32; gb = words * 8
    prm = G.SeqSketcherParams(21, a.m, "optdens")
    nrec = max(chunk, a.nq)
    d_seq = ctx.alloc(nrec * gb + 64)
    d_rs, d_rl, d_go = ctx.alloc(8 * nrec), ctx.alloc(8 * nrec), ctx.alloc(8 * (nrec + 1))
    ctx.upload(d_rs, np.arange(nrec, dtype=np.uint64) * np.uint64(words * 32)); ctx.upload(d_rl, np.full(nrec, Lg, np.uint64)); ctx.upload(d_go, np.arange(nrec + 1, dtype=np.uint64))
    def sk(first, n, d_out, roots=n_roots):
        _lib.check(L.gs_synth_dna_family_dev(ctx.h, 2024, first, n, Lg, roots, 0.001, 0.08, d_seq))
        _lib.check(L.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, n * gb + 64, d_rs, d_rl, n, d_go, n, d_out))
    for r0 in range(0, a.n, chunk):
        nr = min(chunk, a.n - r0)
        sk(r0, nr, d_rows)
        _lib.check(L.gs_index_parallel_insert_dev(hn.h, d_rows, nr))
    sk(1_000_000_000, a.nq, d_q, a.query_roots or n_roots)
    for p_ in (d_seq, d_rs, d_rl, d_go):
        ctx.free(p_)
else:
    for r0 in range(0, a.n, chunk):
        nr = min(chunk, a.n - r0)
        _lib.check(L.gs_synth_sigs_dev(ctx.h, 3, a.m, 99, r0, nr, n_roots, 0.3, 0.99, d_rows))
        _lib.check(L.gs_index_parallel_insert_dev(hn.h, d_rows, nr))
    _lib.check(L.gs_synth_sigs_dev(ctx.h, 3, a.m, 99, 10_000_000, a.nq, a.query_roots or n_roots, 0.3, 0.99, d_q))
ctx.sync()
print("built %d nodes in %.1fs" % (a.n, time.perf_counter() - t0), flush=True)
ctx.free(d_rows)
q = ctx.download(d_q, (a.nq, a.m), np.float32)
base = None
for v in a.variants:
    saved = {}
    for kv in [x for x in v.split(",") if x]:
        k, val = kv.split("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = val
    best = None
    for rep in range(a.reps + 1):                       # first repetition warms up
        ctx.profile(True)
        for f in range(4):
            ctx.profile_read(f, reset=True)
        hn.search_stats(reset=True)
        t = time.perf_counter()
        res = hn.search_arrays(q, a.knbn, a.ef)
        dt = time.perf_counter() - t
        srch = ctx.profile_read(2, reset=True); join = ctx.profile_read(1, reset=True)
        ctx.profile(False)
        st = hn.search_stats(reset=True)
        if rep and (best is None or srch[0] < best[0][0]):
            best = (srch, join, dt, st)
    srch, join, dt, st = best
    same = "-" if base is None else ("True" if all(np.array_equal(x, y) for x, y in zip(res, base)) else
                                     "False[ids %s dist %s count %s evals %s; queries differing in ids: %d]" % (np.array_equal(res[0], base[0]), np.array_equal(res[1], base[1]), np.array_equal(res[2], base[2]),
                                                                                         np.array_equal(res[3], base[3]), int((res[0] != base[0]).any(axis=1).sum())))
    if base is None:
        base = res
    print("%-40s traversal %8.2f ms (%d launches)  join %8.2f ms  call %8.1f ms  pops/q %.0f (before dmax=tau %.0f, order-free %.0f) acc/q %.0f wg %d  same_as_first=%s" %
          (v or "(defaults)", srch[0], srch[1], join[0], dt * 1e3, st["pops"] / a.nq, st["pops_phase1"] / a.nq, st["pops_phase2"] / a.nq, st["accepting_pops"] / a.nq, st["wg_in_flight"], same), flush=True)
    for k, old in saved.items():
        if old is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = old
print("evals/query mean %.0f" % base[3].mean())
if os.environ.get("GS_AB_GRAPH_STATS"):
    g = hn.export_graph()
    d = g["deg0"]
    print("deg0: mean %.1f median %d p10 %d p90 %d max %d; frac full %.3f" % (d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max(), (d == 2 * a.M).mean()))
    # match statistics of the first query against the whole DB (exhaustive): how many nodes share >= 1, 2, 3 slots
    D = G.DistHamming(ctx).eval_qxc(q[:4], hn.get_data())
    cnt = np.rint(D * a.m).astype(np.int64)
    for i in range(4):
        mt = a.m - cnt[i]
        print("query %d: nodes with 0 matches %.3f, 1: %.3f, 2: %.3f, >=3: %.4f" % (i, (mt == 0).mean(), (mt == 1).mean(), (mt == 2).mean(), (mt >= 3).mean()))
