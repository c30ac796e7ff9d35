#!/usr/bin/env python3
"""BASELINE configs[4]: AA path, --aa k=7 s=24000 SuperMinHash2 (u64 sketches), synthetic proteomes of 1.5 M residues,
DistHamming on u64. Sketch throughput on the GPU (inputs generated on the host in chunks, timing = kernel only via HIP events),
parity of a sample vs the oracle, and the dense DistHamming rate on u64 signatures."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsearch_amd as G
from gsearch_amd import _lib
from gsearch_amd.api import _p
import helpers as H
import oracle_lib as O

NP_TOTAL, CH, L, k, m = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 500, 1_500_000, 7, 24000
ctx = G.Context(0); lib = ctx.L
prm = G.SeqSketcherParams(k, m, "super2", "aa")
rng = np.random.default_rng(5)
base = H.AA20[rng.integers(0, 20, CH * L + 64)].copy()          # one chunk of proteomes; re-used with a per-chunk rotation
rs = np.arange(CH, dtype=np.uint64) * np.uint64(L); rl = np.full(CH, L, np.uint64); goff = np.arange(CH + 1, dtype=np.uint64)
d_seq, d_sig = ctx.alloc(CH * L + 64), ctx.alloc(CH * m * 8)
d_rs, d_rl, d_goff = ctx.alloc(rs.nbytes), ctx.alloc(rl.nbytes), ctx.alloc(goff.nbytes)
ctx.upload(d_rs, rs); ctx.upload(d_rl, rl); ctx.upload(d_goff, goff)
ctx.profile(True)
tot_ms, done, parity = 0.0, 0, True
t0 = time.perf_counter()
for c in range(0, NP_TOTAL, CH):
    buf = np.roll(base, 7919 * (c // CH))                          # different proteomes per chunk, same cost
    ctx.upload(d_seq, buf)
    _lib.check(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, CH * L + 64, d_rs, d_rl, CH, d_goff, CH, d_sig))
    if c == 0:
        sig = ctx.download(d_sig, (CH, m), np.uint64)
        ref = O.sketch_batch(O.params(k, m, "super2", "aa"), buf, rs[:4], rl[:4], np.arange(5, dtype=np.uint64), nthreads=4)
        parity = bool(np.array_equal(sig[:4], ref))
        t = time.perf_counter(); O.sketch_batch(O.params(k, m, "super2", "aa"), buf, rs[:64], rl[:64], np.arange(65, dtype=np.uint64), nthreads=os.cpu_count()); cpu_s = time.perf_counter() - t
    done += CH
ms, n = ctx.profile_read(0)
wall = time.perf_counter() - t0
kmers = float(L - k + 1) * done
out = {"config": "C5: AA k=7 s=24000 super2 (u64), %d proteomes x %.1f M residues" % (done, L / 1e6), "sketch_kernel_ms_total": ms, "launches": n,
       "kmers_per_sec_kernel": kmers / (ms * 1e-3), "proteomes_per_sec_kernel": done / (ms * 1e-3), "wall_s_incl_host_generation_and_pcie": wall,
       "bit_exact_vs_oracle_sample": parity, "cpu_oracle_kmers_per_sec": float(L - k + 1) * 64 / cpu_s, "cores": os.cpu_count()}
# DistHamming on u64 signatures: dense tile kernel rate, 2048 x 20000
nq, nc = 2048, 20000
dq, dc, do = ctx.alloc(nq * m * 8), ctx.alloc(nc * m * 8), ctx.alloc(nq * nc * 4)
_lib.check(lib.gs_synth_sigs_dev(ctx.h, 2, m, 1, 0, nq, 100, 0.3, 0.9, dq)); _lib.check(lib.gs_synth_sigs_dev(ctx.h, 2, m, 1, 10**6, nc, 100, 0.3, 0.9, dc)); ctx.sync()
ctx.timer_start(); _lib.check(lib.gs_hamming_qxc_dev(ctx.h, 2, m, dq, nq, dc, nc, do)); t_ms = ctx.timer_stop()
ctx.timer_start(); _lib.check(lib.gs_hamming_qxc_dev(ctx.h, 2, m, dq, nq, dc, nc, do)); t_ms = ctx.timer_stop()
out.update({"hamming_u64_evals_per_sec": nq * nc / (t_ms * 1e-3), "hamming_u64_algorithmic_GBps": nq * nc * m * 8 / (t_ms * 1e-3) / 1e9})
print(json.dumps(out))
