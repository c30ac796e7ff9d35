#!/usr/bin/env python3
"""Does anything else run on the device while k_inflate is resident? One thread inflates N gzip members (gs_gunzip_batch, context B) while the main
thread keeps sketching small batches of resident genomes on context A and records when each call returns.
usage: inflate_corun.py [n_members] [lds|global]"""
import os, sys, threading, time, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
if len(sys.argv) > 2: os.environ["GS_INFLATE_WINDOW"] = sys.argv[2]
import gsearch_amd as G
L = 2_000_000
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", np.uint8)
seq = acgt[rng.integers(0, 4, L)]
lines = np.concatenate([np.resize(seq, ((L + 79) // 80, 80)), np.full(((L + 79) // 80, 1), 10, np.uint8)], axis=1)
t = b">g\n" + lines.tobytes()
c = zlib.compressobj(6, zlib.DEFLATED, 31); member = c.compress(t) + c.flush()
cA, cB = G.Context(0), G.Context(0)
sk = G.OptDensHashSketch.new(G.SeqSketcherParams(21, 18000, "optdens"), ctx=cA)
genomes = [[bytes(seq[:1_000_000])] for _ in range(64)]
sk.sketch_genomes(genomes)                                   # warm
G.gunzip_batch(cB, [member] * 8)                             # warm
done = {}
def infl():
    t0 = time.perf_counter(); res = G.gunzip_batch(cB, [member] * n); done["t"] = (t0, time.perf_counter()); done["ok"] = all(s == 0 for s, _ in res)
th = threading.Thread(target=infl); th.start()
stamps = []
t_end = time.perf_counter() + 3.0
while th.is_alive() and time.perf_counter() < t_end:
    a = time.perf_counter(); sk.sketch_genomes(genomes); stamps.append((a, time.perf_counter()))
th.join()
t0, t1 = done["t"]
print("inflate of %d members (%s): %.3f s, ok=%s" % (n, os.environ.get("GS_INFLATE_WINDOW", "auto"), t1 - t0, done["ok"]))
durs = [b - a for a, b in stamps if a >= t0 and b <= t1]
print("sketch calls of 64 x 1 Mbp that started and finished inside it: %d, median %.1f ms, max %.1f ms" % (len(durs), 1e3 * (np.median(durs) if durs else 0), 1e3 * (max(durs) if durs else 0)))
a0 = time.perf_counter(); sk.sketch_genomes(genomes); print("the same call alone: %.1f ms" % (1e3 * (time.perf_counter() - a0)))
