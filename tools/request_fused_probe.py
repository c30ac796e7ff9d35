#!/usr/bin/env python3
"""`request` as one call (gs_index_sketch_and_search_dev: sketch of batch b + 1 beside the count matrix of batch b) against the two separate calls, on the
bench's data (300 k x 5 Mbp OptDens DB, 10 000 queries per request).  usage: request_fused_probe.py [n] [nq] [reps]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsearch_amd as G
from gsearch_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k, m, L, knbn, ef = 21, 18000, 5_000_000, 50, 5000
ctx = G.Context(0); lib = ctx.L; chk = _lib.check
words = (L + 31) // 32; gb = words * 8
prm = G.SeqSketcherParams(k, m, "optdens")
chunk = 8192; nrec = max(chunk, nq)
d_seq = ctx.alloc(nrec * gb + 64); d_rows = ctx.alloc(chunk * m * 4)
d_rs, d_rl, d_go = ctx.alloc(8 * nrec), ctx.alloc(8 * nrec), ctx.alloc(8 * (nrec + 1))
ctx.upload(d_rs, np.arange(nrec, dtype=np.uint64) * np.uint64(words * 32)); ctx.upload(d_rl, np.full(nrec, L, np.uint64)); ctx.upload(d_go, np.arange(nrec + 1, dtype=np.uint64))
hn = G.Hnsw.new(128, n, 16, 1600, G.DistHamming(ctx), seed=1, insert_batch=256, ctx=ctx)
hn.modify_level_scale(0.25); hn.set_extend_candidates(True); hn._ensure(m)
n_roots = max(n // 100, 1)
t0 = time.perf_counter()
for r0 in range(0, n, chunk):
    nr = min(chunk, n - r0)
    chk(lib.gs_synth_dna_family_dev(ctx.h, 2024, r0, nr, L, n_roots, 0.001, 0.08, d_seq))
    chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, nr * gb + 64, d_rs, d_rl, nr, d_go, nr, d_rows))
    chk(lib.gs_index_parallel_insert_dev(hn.h, d_rows, nr))
ctx.sync(); print("built %d nodes in %.1fs" % (n, time.perf_counter() - t0), flush=True)
chk(lib.gs_synth_dna_family_dev(ctx.h, 2024, 1_000_000_000, nq, L, n_roots, 0.001, 0.08, d_seq))
d_qsig = ctx.alloc(nq * m * 4)
outs = [[ctx.alloc(8 * nq * knbn), ctx.alloc(4 * nq * knbn), ctx.alloc(4 * nq), ctx.alloc(8 * nq)] for _ in range(2)]
def separate():
    chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, nq * gb + 64, d_rs, d_rl, nq, d_go, nq, d_qsig))
    chk(lib.gs_index_parallel_search_dev(hn.h, d_qsig, nq, knbn, ef, *outs[0]))
def fused():
    hn.sketch_and_search_dev(prm, d_seq, nq * gb + 64, d_rs, d_rl, nq, d_go, nq, knbn, ef, *outs[1], d_sig=d_qsig)
# further arguments: environment variants of the one-call form ("GS_REQUEST_PIPELINE=3,GS_PIPE_SKETCH_LDS=102400" ...), each timed and compared with the two calls
variants = sys.argv[4:] or [""]
shapes = ((nq, knbn), (nq, knbn), (nq,), (nq,)); dts = (np.uint64, np.uint32, np.uint32, np.uint64)
def timeit(fn):
    best = None
    for r in range(reps):
        ctx.sync(); t0 = time.perf_counter(); fn(); ctx.sync(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best
b = timeit(separate)
print("two calls: %.1f ms per %d-query request (%.0f genomes/s)" % (b * 1e3, nq, nq / b), flush=True)
ref = [ctx.download(a, s_, d) for a, s_, d in zip(outs[0], shapes, dts)]
for v in variants:
    saved = {}
    for kv in [x for x in v.split(",") if x]:
        k_, val = kv.split("="); saved[k_] = os.environ.get(k_); os.environ[k_] = val
    for a in outs[1]:
        lib.gs_dev_memset(ctx.h, a, 0, 4 * nq)
    b = timeit(fused)
    same = all(np.array_equal(ctx.download(a, s_, d), r_) for a, s_, d, r_ in zip(outs[1], shapes, dts, ref))
    print("one call [%s]: %.1f ms per %d-query request (%.0f genomes/s)  same answers: %s" % (v or "defaults", b * 1e3, nq, nq / b, same), flush=True)
    for k_, old in saved.items():
        if old is None: os.environ.pop(k_, None)
        else: os.environ[k_] = old
b = timeit(separate)
print("two calls: %.1f ms per %d-query request (%.0f genomes/s)" % (b * 1e3, nq, nq / b), flush=True)
