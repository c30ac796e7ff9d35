#!/usr/bin/env python3
"""k-mers/s of gs_sketch_batch_dev for any DNA algorithm on synthetic genomes generated in HBM.
usage: sketch_rate.py <algo: optdens|revoptdens|prob|super|super2|hll> [n_genomes] [len] [k] [m] [dna|aa] [record_len]
(record_len: split every genome into records of that many symbols - k-mers never span records, dnasketch.rs:348-363)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsearch_amd as G
from gsearch_amd.api import check, default_context

algo = sys.argv[1] if len(sys.argv) > 1 else "optdens"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
k = int(sys.argv[4]) if len(sys.argv) > 4 else 21
m = int(sys.argv[5]) if len(sys.argv) > 5 else 18000
data = sys.argv[6] if len(sys.argv) > 6 else "dna"
rec = int(sys.argv[7]) if len(sys.argv) > 7 else 0
ctx = default_context(); lib = ctx.L
prm = G.SeqSketcherParams(k, m, algo, data)
if data == "dna":
    words = (L + 31) // 32
    gbytes = words * 8
    d_seq = ctx.alloc(n * gbytes + 64)
    check(lib.gs_synth_dna_dev(ctx.h, 7, 0, n, L, d_seq))
    gstart = np.arange(n, dtype=np.uint64) * np.uint64(words * 32)
else:
    gbytes = L
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    host = aa[np.random.default_rng(3).integers(0, 20, n * L + 64)]
    d_seq = ctx.alloc(n * L + 64)
    ctx.upload(d_seq, host)
    gstart = np.arange(n, dtype=np.uint64) * np.uint64(L)
if rec:      # records of `rec` symbols (DNA: starts stay on 32-base boundaries)
    per = (L + rec - 1) // rec
    off = np.arange(per, dtype=np.uint64) * np.uint64(rec)
    rs = (gstart[:, None] + off[None, :]).reshape(-1)
    rl = np.minimum(np.uint64(rec), np.uint64(L) - np.tile(off, n)).astype(np.uint64)
    go = np.arange(n + 1, dtype=np.uint64) * np.uint64(per)
else:
    rs, rl, go = gstart, np.full(n, L, np.uint64), np.arange(n + 1, dtype=np.uint64)
nrec = len(rs)
d_rs, d_rl, d_go = ctx.alloc(8 * nrec), ctx.alloc(8 * nrec), ctx.alloc(8 * (n + 1))
ctx.upload(d_rs, rs); ctx.upload(d_rl, rl); ctx.upload(d_go, go)
d_sig = ctx.alloc(n * m * prm.sig_dtype().itemsize)
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    check(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, n * gbytes + 64, d_rs, d_rl, nrec, d_go, n, d_sig))
    ctx.sync(); dt = time.perf_counter() - t0
    print("%s %s rec=%d k=%d m=%d: %d genomes x %.1f Mbp in %.1f ms -> %.3e k-mers/s, %.0f genomes/s" % (algo, data, rec, k, m, n, L / 1e6, dt * 1e3, n * (L - k + 1) / dt, n / dt))
