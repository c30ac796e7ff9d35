#!/usr/bin/env python3
"""The match-join on a SKEWED database in isolation: rows of power-law family sizes (gs_synth_sigs_skew_dev), an isolated-node graph, one dense-mode
search = the count matrix of the batch + a one-node traversal.  usage: skew_probe.py [n] [nq] [m] [n_roots] [alpha]   (GS_JOIN_VERBOSE=1 GS_JOIN_TIMES=1 for the stages)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["GS_DIST_MODE"] = "dense"
import gsearch_amd as G
from gsearch_amd.api import check, default_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
m = int(sys.argv[3]) if len(sys.argv) > 3 else 18000
n_roots = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
alpha = float(sys.argv[5]) if len(sys.argv) > 5 else 3.5
ctx = default_context(); lib = ctx.L
d = ctx.alloc(n * m * 4)
check(lib.gs_synth_sigs_skew_dev(ctx.h, G._lib.KIND_F32, m, 515, 0, n, n_roots, 0.3, 0.99, alpha, d))
db = ctx.download(d, (n, m), np.float32)
check(lib.gs_synth_sigs_skew_dev(ctx.h, G._lib.KIND_F32, m, 515, 7_000_000, nq, n_roots, 0.3, 0.99, alpha, d))
q = ctx.download(d, (nq, m), np.float32)
ctx.free(d)
M = 8
hn = G.Hnsw.new(M, n, 16, 16, G.DistHamming(ctx), seed=1, ctx=ctx)
g = dict(levels=np.zeros(n, np.uint8), entry=0, deg0=np.zeros(n, np.uint32), nbr0=np.zeros((n, 2 * M), np.uint32), cnt0=np.zeros((n, 2 * M), np.uint32),
         upidx=np.full(n, -1, np.int32), n_upper=0)
hn.import_graph(db, g)
for rep in range(3):
    ctx.profile(True); ctx.profile_read(1, reset=True)
    t0 = time.perf_counter()
    ids, dist, cnt, ev = hn.search_arrays(q, 1, 1)
    dt = time.perf_counter() - t0
    ms, nl = ctx.profile_read(1, reset=True); ctx.profile(False)
    st = hn.search_stats(reset=True)
    print("rep %d: count-matrix kernels %.2f ms (%d launches), call %.1f ms, join atomics %.3e, shared-entry expansions %s" % (rep, ms, nl, dt * 1e3, st.get("join_atomics", 0), st.get("join_shared_expansions")), flush=True)
exp = (q[:8, None, :] != db[None, :1, :]).sum(-1)[:, 0] / np.float32(m)
assert np.allclose(dist[:8, 0], exp), (dist[:8, 0], exp)
print("ok")
