#!/usr/bin/env python3
"""The reference's worker pattern on ONE context: T host threads (its --nbthreads sketcher clones, dnasketch.rs:252,305,322) each sketch their own
small batches through gs_sketch_batch (host pointers, synchronous). With per-thread worker contexts the calls run side by side on the GPU;
GS_THREAD_CONTEXTS=0 queues them on the one context lock.  usage: thread_ctx_probe.py [threads] [calls per thread] [genomes per call] [genome len]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsearch_amd as G

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ng = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2_000_000
rng = np.random.default_rng(0)
sk = G.OptDensHashSketch.new(G.SeqSketcherParams(21, 18000, "optdens"))
packed = [rng.integers(0, 256, L // 4 + 64, dtype=np.uint8) for _ in range(ng)]         # random packed 2-bit genomes
seq = np.concatenate(packed)
stride = (L // 4 + 64) * 4
rs = np.arange(ng, dtype=np.uint64) * np.uint64(stride)
rl = np.full(ng, L, np.uint64)
go = np.arange(ng + 1, dtype=np.uint64)
want = sk.sketch_packed(seq, rs, rl, go)
bad = []


def work():
    for _ in range(calls):
        if not np.array_equal(sk.sketch_packed(seq, rs, rl, go).view(np.uint32), want.view(np.uint32)):
            bad.append(1)


for rep in range(2):
    th = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("GS_THREAD_CONTEXTS=%s rep %d: %d threads x %d calls x %d genomes of %.1f Mbp in %.2f s -> %.0f genomes/s%s"
          % (os.environ.get("GS_THREAD_CONTEXTS", "1"), rep, T, calls, ng, L / 1e6, dt, T * calls * ng / dt, "  MISMATCH" if bad else ""), flush=True)
