#!/usr/bin/env python3
"""Build-time scaling probe (round 5): `tohnsw` of N synthetic 5 Mbp genomes (generated, sketched and inserted on the device, like
bench.py's setup) with a wall-clock stamp every `--report` genomes, then one request of --nq queries. Environment variables pass through
(GS_PAIR_CACHE_GB=0 forces the sparse pair rows from the start; GS_SPARSE_ROWS=0 is the round-4 behaviour beyond the dense cache).
usage: build_scale.py --n 1000000 [--per-root 100] [--nq 10000]"""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsearch_amd as G
from gsearch_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--m", type=int, default=18000)
ap.add_argument("--M", type=int, default=128)
ap.add_argument("--efc", type=int, default=1600)
ap.add_argument("--ef", type=int, default=5000)
ap.add_argument("--knbn", type=int, default=50)
ap.add_argument("--per-root", type=int, default=100)
ap.add_argument("--genome-len", type=int, default=5_000_000)
ap.add_argument("--report", type=int, default=131072)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--variants", nargs="*", default=[""], help='environment variants of the final request ("GS_DENSE_VIS=global" "GS_SPLIT_PER_CU=1" ...)')
a = ap.parse_args()

ctx = G.Context(0)
L = ctx.L
n_roots = max(a.n // a.per_root, 1)
hn = G.Hnsw.new(a.M, 1_500_000, 16, a.efc, G.DistHamming(ctx), seed=1, insert_batch=256, ctx=ctx)
hn.modify_level_scale(0.25); hn.set_extend_candidates(True); hn.set_keeping_pruned(False)
hn._ensure(a.m)
chunk = 8192
Lg = a.genome_len; words = (Lg + 31) // 32; gb = words * 8
prm = G.SeqSketcherParams(21, a.m, "optdens")
nrec = max(chunk, a.nq)
d_rows = ctx.alloc(chunk * a.m * 4); d_q = ctx.alloc(a.nq * a.m * 4)
d_seq = ctx.alloc(nrec * gb + 64)
d_rs, d_rl, d_go = ctx.alloc(8 * nrec), ctx.alloc(8 * nrec), ctx.alloc(8 * (nrec + 1))
ctx.upload(d_rs, np.arange(nrec, dtype=np.uint64) * np.uint64(words * 32)); ctx.upload(d_rl, np.full(nrec, Lg, np.uint64)); ctx.upload(d_go, np.arange(nrec + 1, dtype=np.uint64))


def sk(first, n, d_out, roots=n_roots):
    _lib.check(L.gs_synth_dna_family_dev(ctx.h, 2024, first, n, Lg, roots, 0.001, 0.08, d_seq))
    _lib.check(L.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, n * gb + 64, d_rs, d_rl, n, d_go, n, d_out))


t0 = time.perf_counter(); last = t0; nxt = a.report
t_sk = 0.0
for r0 in range(0, a.n, chunk):
    nr = min(chunk, a.n - r0)
    ts = time.perf_counter()
    sk(r0, nr, d_rows); ctx.sync()
    t_sk += time.perf_counter() - ts
    _lib.check(L.gs_index_parallel_insert_dev(hn.h, d_rows, nr))
    if r0 + nr >= nxt or r0 + nr == a.n:
        ctx.sync()
        now = time.perf_counter()
        fr, tot = 0, 0
        print("nodes %8d  elapsed %7.1f s  (+%6.1f s, %.2f ms per 256-point batch; generate+sketch so far %.1f s)" % (r0 + nr, now - t0, now - last, (now - last) / max(1, (r0 + nr - (nxt - a.report)) / 256) * 1e3, t_sk), flush=True)
        last = now; nxt += a.report
ctx.sync()
print("built %d nodes in %.1f s (generate + sketch %.1f s of it)" % (a.n, time.perf_counter() - t0, t_sk), flush=True)
sk(1_000_000_000, a.nq, d_q)
ctx.sync()
ctx.free(d_rows)
for p_ in (d_seq, d_rs, d_rl, d_go):
    ctx.free(p_)
q = ctx.download(d_q, (a.nq, a.m), np.float32)
base = None
for v in a.variants:
  saved = {}
  for kv in [x for x in v.split(",") if x]:
      k_, val = kv.split("="); saved[k_] = os.environ.get(k_); os.environ[k_] = val
  print("request variant [%s]" % (v or "defaults"), flush=True)
  for rep in range(a.reps):
    ctx.profile(True)
    for f in range(4):
        ctx.profile_read(f, reset=True)
    hn.search_stats(reset=True)
    t = time.perf_counter()
    res = hn.search_arrays(q, a.knbn, a.ef)
    dt = time.perf_counter() - t
    srch = ctx.profile_read(2, reset=True); joink = ctx.profile_read(1, reset=True)
    ctx.profile(False)
    st = hn.search_stats(reset=True)
    same = "-" if base is None else str(all(np.array_equal(x, y) for x, y in zip(res, base)))
    print("request of %d queries at %d nodes: call %.1f ms, count-matrix kernels %.1f ms in %d launches, traversal kernel %.2f ms; pops/q %.0f (phase 1 %.0f) evals/q %.0f wg %d  same answers as the first variant: %s" %
          (a.nq, a.n, dt * 1e3, joink[0], joink[1], srch[0], st["pops"] / a.nq, st["pops_phase1"] / a.nq, res[3].mean(), st["wg_in_flight"], same), flush=True)
  if base is None:
      base = res
  for k_, old in saved.items():
      if old is None: os.environ.pop(k_, None)
      else: os.environ[k_] = old
