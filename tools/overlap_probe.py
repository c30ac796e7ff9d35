#!/usr/bin/env python3
"""Do the sketch kernel (VALU-bound) and the match-join (memory-side-atomics-bound) overlap when they run on two streams?
Two contexts on device 0: A sketches batches of genomes, B joins query batches against a genome-level DB (search with ef = knbn = 1, so
the traversal is a greedy descent). Prints the wall time of R rounds of each alone and of both at once.
usage: overlap_probe.py [--n 100000] [--rounds 8] [env K=V,K=V]"""
import argparse, ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsearch_amd as G
from gsearch_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--nq", type=int, default=3276)
ap.add_argument("--ns", type=int, default=2500)
ap.add_argument("--m", type=int, default=18000)
ap.add_argument("--rounds", type=int, default=8)
a = ap.parse_args()
Lg = 5_000_000; words = (Lg + 31) // 32; gb = words * 8
prm = G.SeqSketcherParams(21, a.m, "optdens")

def genome_bufs(ctx, nrec):
    d_seq = ctx.alloc(nrec * gb + 64)
    d_rs, d_rl, d_go = ctx.alloc(8 * nrec), ctx.alloc(8 * nrec), ctx.alloc(8 * (nrec + 1))
    ctx.upload(d_rs, np.arange(nrec, dtype=np.uint64) * np.uint64(words * 32)); ctx.upload(d_rl, np.full(nrec, Lg, np.uint64)); ctx.upload(d_go, np.arange(nrec + 1, dtype=np.uint64))
    return d_seq, d_rs, d_rl, d_go

cB = G.Context(0); LB = cB.L
n_roots = max(a.n // 100, 1)
hn = G.Hnsw.new(16, 1_500_000, 16, 32, G.DistHamming(cB), seed=1, insert_batch=256, ctx=cB)
hn._ensure(a.m)
chunk = 4096
bufs = genome_bufs(cB, max(chunk, a.nq))
d_rows = cB.alloc(chunk * a.m * 4); d_q = cB.alloc(a.nq * a.m * 4)
def sk(ctx, b, first, n, d_out):
    _lib.check(ctx.L.gs_synth_dna_family_dev(ctx.h, 2024, first, n, Lg, n_roots, 0.001, 0.08, b[0]))
    _lib.check(ctx.L.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), b[0], n * gb + 64, b[1], b[2], n, b[3], n, d_out))
t0 = time.perf_counter()
for r0 in range(0, a.n, chunk):
    nr = min(chunk, a.n - r0)
    sk(cB, bufs, r0, nr, d_rows)
    _lib.check(LB.gs_index_parallel_insert_dev(hn.h, d_rows, nr))
sk(cB, bufs, 1_000_000_000, a.nq, d_q)
cB.sync()
print("DB of %d nodes in %.1fs" % (a.n, time.perf_counter() - t0), flush=True)
q = cB.download(d_q, (a.nq, a.m), np.float32)
os.environ["GS_DIST_MODE"] = "dense"

cA = G.Context(0)
bA = genome_bufs(cA, a.ns)
d_sigA = cA.alloc(a.ns * a.m * 4)
_lib.check(cA.L.gs_synth_dna_family_dev(cA.h, 77, 5_000_000, a.ns, Lg, n_roots, 0.001, 0.08, bA[0]))
cA.sync()

def run_a():
    for _ in range(a.rounds):
        _lib.check(cA.L.gs_sketch_batch_dev(cA.h, C.byref(prm.c), bA[0], a.ns * gb + 64, bA[1], bA[2], a.ns, bA[3], a.ns, d_sigA))
    cA.sync()
def run_b():
    for _ in range(a.rounds):
        hn.search_arrays(q, 1, 1)
    cB.sync()
def wall(fs):
    th = [threading.Thread(target=f) for f in fs]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    return (time.perf_counter() - t) * 1e3
run_a(); run_b()
for rep in range(2):
    ta, tb, tab = wall([run_a]), wall([run_b]), wall([run_a, run_b])
    print("rep %d: %d rounds | sketch alone %.1f ms (%.1f/round) | join alone %.1f ms (%.1f/round) | both at once %.1f ms = %.2f of the sum" %
          (rep, a.rounds, ta, ta / a.rounds, tb, tb / a.rounds, tab, tab / (ta + tb)), flush=True)
