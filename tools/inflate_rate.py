#!/usr/bin/env python3
"""Device inflate alone: n single-member gzip buffers of L-bp FASTA through gs_gunzip_batch (H2D of the members, k_inflate, k_crc32_chunks,
D2H of the text). Run under tools/kstats.sh for the kernel times.  usage: inflate_rate.py [n_members] [genome_len] [gzip_level]"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsearch_amd as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
lvl = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", np.uint8)
texts, members = [], []
for i in range(min(n, 8)):
    seq = acgt[rng.integers(0, 4, L)]
    lines = np.concatenate([np.resize(seq, ((L + 79) // 80, 80)), np.full(((L + 79) // 80, 1), 10, np.uint8)], axis=1)
    t = b">genome%d synthetic\n" % i + lines.tobytes()
    c = zlib.compressobj(lvl, zlib.DEFLATED, 31)
    texts.append(t); members.append(c.compress(t) + c.flush())
print("member: %.2f MB -> %.2f MB text" % (len(members[0]) / 1e6, len(texts[0]) / 1e6), flush=True)
ctx = G.default_context()
batch = [members[i % len(members)] for i in range(n)]
for rep in range(3):
    t0 = time.perf_counter()
    res = G.gunzip_batch(ctx, batch)
    dt = time.perf_counter() - t0
    ok = all(st == 0 for st, _ in res)
    same = all(res[i][1] == texts[i % len(texts)] for i in range(0, n, max(1, n // 16)))
    print("rep %d: %d members in %.3fs -> %.0f members/s, %.2f GB/s of text (copies included); status ok=%s, bytes equal=%s" % (rep, n, dt, n / dt, n * len(texts[0]) / dt / 1e9, ok, same), flush=True)
