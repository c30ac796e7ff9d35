#!/usr/bin/env python3
"""Times the match-join (dense DistHamming counts of a query batch against every node) in isolation: an index with a trivial graph
(every node isolated) is imported, so a dense-mode search is the join plus a one-node traversal.  usage: join_probe.py [n] [nq] [m]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["GS_DIST_MODE"] = "dense"
import gsearch_amd as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
m = int(sys.argv[3]) if len(sys.argv) > 3 else 18000
rng = np.random.default_rng(1)
U = 2 * m                                   # value universe: unrelated rows agree in ~0.5 slots, as OptDens sketches of unrelated genomes do
fam = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # members per root (0: unrelated rows; 100: the family structure of the bench DB)
if fam:
    roots = rng.integers(0, U, (n // fam, m), dtype=np.int32).astype(np.float32)
    db = np.repeat(roots, fam, axis=0)
    for r0 in range(0, n, 4096):                           # members keep a root slot with probability J ~ U[0.3, 0.9]
        blk = db[r0:r0 + 4096]
        J = rng.uniform(0.3, 0.9, (len(blk), 1)).astype(np.float32)
        mk = rng.random(blk.shape, dtype=np.float32) > J
        blk[mk] = rng.integers(0, U, int(mk.sum())).astype(np.float32)
    q = db[rng.integers(0, n, nq)].copy()
    mask = rng.random(q.shape, dtype=np.float32) < 0.3
    q[mask] = rng.integers(0, U, int(mask.sum())).astype(np.float32)
else:
    db = rng.integers(0, U, (n, m), dtype=np.int32).astype(np.float32)
    q = db[rng.integers(0, n, nq)].copy()
    mask = rng.random(q.shape) < 0.5
    q[mask] = rng.integers(0, U, int(mask.sum())).astype(np.float32)
M = 8
hn = G.Hnsw.new(M, n, 16, 16, G.DistHamming(), seed=1)
g = dict(levels=np.zeros(n, np.uint8), entry=0, deg0=np.zeros(n, np.uint32), nbr0=np.zeros((n, 2 * M), np.uint32), cnt0=np.zeros((n, 2 * M), np.uint32),
         upidx=np.full(n, -1, np.int32), n_upper=0)
hn.import_graph(db, g)
ctx = hn.ctx
for rep in range(3):
    ctx.profile(True); ctx.profile_read(1, reset=True)
    t0 = time.perf_counter()
    ids, dist, cnt, ev = hn.search_arrays(q, 1, 1)
    dt = time.perf_counter() - t0
    ms, nl = ctx.profile_read(1, reset=True); ctx.profile(False)
    print("rep %d: join kernel %.2f ms (%d launches), call %.1f ms, column stream %.1f GB/s" % (rep, ms, nl, dt * 1e3, n * m * 4.0 / (ms * 1e-3) / 1e9 if ms else 0))
exp = (q[:8, None, :] != db[None, :1, :]).sum(-1)[:, 0] / np.float32(m)     # entry node 0 is the only node reached
assert np.allclose(dist[:8, 0], exp), (dist[:8, 0], exp)
print("ok")
