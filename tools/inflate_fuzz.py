#!/usr/bin/env python3
"""Differential run of the device DEFLATE decoder (gs_inflate.hip through gs_gunzip_batch) against zlib over random gzip members: text kinds (DNA FASTA, low-entropy, random bytes,
long runs, tiny / empty), every compression level and strategy (fixed / Huffman-only / RLE / filtered: stored, fixed and dynamic blocks), window sizes, and each of the kernel's forms
(GS_INFLATE_WINDOW). A member must come back byte-identical with status 0; a damaged copy of it must NOT come back with status 0 and different bytes.
usage: inflate_fuzz.py [rounds] [seed]"""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsearch_amd as G
from gsearch_amd.api import default_context

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = default_context()


def text(kind, n):
    if kind == 0:                                                   # FASTA-like DNA with headers and line breaks
        body = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]
        t = bytearray(body.tobytes())
        for p in range(0, len(t), 81):
            t[p:p + 1] = b"\n"
        return b">seq%d some description\n" % n + bytes(t)
    if kind == 1:                                                   # two symbols, long matches
        return bytes(np.frombuffer(b"AT", np.uint8)[(rng.random(n) < 0.02).astype(np.int64)])
    if kind == 2:                                                   # incompressible
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 3:                                                   # one byte many times, then a motif repeated (distance-1 and short-distance copies, 258-byte matches)
        return b"N" * (n // 2) + (b"ACGTTGCA" * (n // 16 + 1))[: n - n // 2]
    if kind == 4:                                                   # a long text pasted twice (matches at the far end of the window)
        half = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, max(n // 2, 1))].tobytes()
        return half + half
    return b""


def gz(data, level, strategy, wbits):
    c = zlib.compressobj(level, zlib.DEFLATED, 16 + wbits, 9 if rng.random() < 0.5 else 1, strategy)
    return c.compress(data) + c.flush()


bad = 0
for rnd in range(rounds):
    members, texts = [], []
    for i in range(int(rng.integers(20, 200))):
        kind = int(rng.integers(0, 6))
        n = int(rng.choice([0, 1, 2, 17, 300, 5000, 70000, 300000, 2_000_000]))
        t = text(kind, n)
        level = int(rng.integers(0, 10))
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED][int(rng.integers(0, 5))]
        wbits = int(rng.choice([9, 12, 15]))
        members.append(gz(t, level, strategy, wbits)); texts.append(t)
    for form in ("pipe", "lds", "global"):
        os.environ["GS_INFLATE_WINDOW"] = form
        res = G.gunzip_batch(ctx, members)
        wrong = [i for i, ((st, out), t) in enumerate(zip(res, texts)) if st != 0 or out != t]
        # damage: flip one bit somewhere in the deflate stream of a third of the members
        dm, idx = [], []
        for i, mbr in enumerate(members):
            if len(mbr) > 30 and rng.random() < 0.33:
                b = bytearray(mbr); p = int(rng.integers(10, len(b) - 8)); b[p] ^= 1 << int(rng.integers(0, 8))
                dm.append(bytes(b)); idx.append(i)
        silent = []
        if dm:
            rd = G.gunzip_batch(ctx, dm, out_caps=[len(texts[i]) for i in idx])
            silent = [idx[j] for j, (st, out) in enumerate(rd) if st == 0 and out != texts[idx[j]]]
        bad += len(wrong) + len(silent)
        print("round %d form %-6s: %d members (%.1f MB of text), wrong %s, damaged %d of which silently wrong %s" %
              (rnd, form, len(members), sum(map(len, texts)) / 1e6, wrong[:8], len(dm), silent[:8]), flush=True)
os.environ.pop("GS_INFLATE_WINDOW", None)
print("%d rounds, %d failures" % (rounds, bad))
sys.exit(1 if bad else 0)
