#!/usr/bin/env python3
"""exploration: build a sketch-level synthetic DB on the device, time parallel_insert and parallel_search."""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsearch_amd as G
from gsearch_amd import _lib
from gsearch_amd.api import _p

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--nq", type=int, default=512)
ap.add_argument("--m", type=int, default=18000)
ap.add_argument("--M", type=int, default=128)
ap.add_argument("--efc", type=int, default=1600)
ap.add_argument("--ef", type=int, default=5000)
ap.add_argument("--knbn", type=int, default=50)
ap.add_argument("--per-root", type=int, default=100)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--scale", type=float, default=0.25)
ap.add_argument("--chunk", type=int, default=25600)
ap.add_argument("--brute", type=int, default=0)
ap.add_argument("--dtype", default="f32", choices=["f32", "u32", "u64"])
a = ap.parse_args()

ctx = G.Context(0)
L = ctx.L
DT = {"f32": np.float32, "u32": np.uint32, "u64": np.uint64}[a.dtype]
KIND = {"f32": 3, "u32": 1, "u64": 2}[a.dtype]
ESZ = np.dtype(DT).itemsize
n_roots = max(a.n // a.per_root, 1)
hn = G.Hnsw.new(a.M, 1_500_000, 16, a.efc, G.DistHamming(ctx), dtype=DT, seed=1, insert_batch=a.batch, ctx=ctx)
hn.modify_level_scale(a.scale); hn.set_extend_candidates(True); hn.set_keeping_pruned(False)
hn._ensure(a.m)
d_rows = ctx.alloc(a.chunk * a.m * ESZ)
t_build = 0.0
ctx.profile(True)
for r0 in range(0, a.n, a.chunk):
    nr = min(a.chunk, a.n - r0)
    _lib.check(L.gs_synth_sigs_dev(ctx.h, KIND, a.m, 99, r0, nr, n_roots, 0.3, 0.99, d_rows)); ctx.sync()
    t = time.perf_counter()
    _lib.check(L.gs_index_parallel_insert_dev(hn.h, d_rows, nr))
    dt = time.perf_counter() - t; t_build += dt
    print("inserted %d..%d in %.2fs  (%.0f pts/s) evals/pt so far %.0f" % (r0, r0 + nr, dt, nr / dt, hn.insert_evals() / (r0 + nr)), flush=True)
ms, nl = ctx.profile_read(3)
print("build: %.2fs total, plan kernels %.2fs over %d launches; insert evals %.3g -> %.1f GB/s algorithmic" % (t_build, ms / 1e3, nl, hn.insert_evals(), hn.insert_evals() * a.m * ESZ / (ms / 1e3) / 1e9 if ms else 0))
# queries: fresh members of random roots (rows beyond the DB range share the same roots)
d_q = ctx.alloc(a.nq * a.m * ESZ)
_lib.check(L.gs_synth_sigs_dev(ctx.h, KIND, a.m, 99, 10_000_000, a.nq, n_roots, 0.3, 0.99, d_q)); ctx.sync()
q = ctx.download(d_q, (a.nq, a.m), DT)
for ef in [a.ef]:
    ctx.profile_read(2)
    t = time.perf_counter()
    ids, dist, cnt, ev = hn.search_arrays(q, a.knbn, ef)
    dt = time.perf_counter() - t
    ms, nl = ctx.profile_read(2)
    print("search ef=%d: %.3fs wall, kernel %.3fs; %.1f q/s; evals/query mean %.0f max %d; algorithmic %.1f GB/s" % (ef, dt, ms / 1e3, a.nq / (ms / 1e3), ev.mean(), ev.max(), ev.sum() * a.m * ESZ / (ms / 1e3) / 1e9))
    print("dist[0][:8]", dist[0][:8], "n<0.99:", (dist < 0.99).sum(1).mean())
if a.brute:
    t = time.perf_counter(); bi, bd = hn.bruteforce_search(q[:a.brute], a.knbn); dt = time.perf_counter() - t
    rec = np.mean([(dist[i] <= bd[i, -1]).mean() for i in range(a.brute)])
    print("brute force %d queries %.2fs; tie-aware recall@%d = %.4f; exact id match %.4f" % (a.brute, dt, a.knbn, rec, (ids[:a.brute] == bi).mean()))
