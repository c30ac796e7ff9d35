#!/usr/bin/env python3
"""Differential run of a sketcher (default ProbMinHash3a: the tiered form and its fallbacks) against the oracle over random shapes: k, sketch size, genome sizes around the
thresholds of the kernels' forms, records, N runs, repeats of random multiplicity, poly-A runs. usage: prob_fuzz.py [cases] [seed] [algo: prob|optdens|revoptdens|super|super2|hll|any]
(GS_PROB_VERBOSE=1 to see the fallbacks of prob)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsearch_amd as G
import helpers as H
import oracle_lib as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ALGO = sys.argv[3] if len(sys.argv) > 3 else "prob"
bad = 0
for case in range(cases):
    algo = ALGO if ALGO != "any" else str(rng.choice(["prob", "optdens", "revoptdens", "super", "super2", "hll"]))
    data = "aa" if rng.random() < 0.2 else "dna"
    if data == "dna":
        k = int(rng.choice([8, 11, 12, 14, 16, 17, 19, 21, 21, 21, 24, 32]))
    else:
        k = int(rng.choice([5, 6, 7, 8]))
    m = int(rng.choice([64, 200, 500, 1000, 2000, 4096]))
    genomes = []
    for g in range(int(rng.integers(2, 6))):
        n = int(rng.choice([64 * m + 50, 70 * m, 120 * m, 300 * m, 1000 * m, 30 * m] if not os.environ.get("FUZZ_BIG") else [600_000, 900_000, 1_500_000, 2_200_000, 70 * m])) + int(rng.integers(0, 1000))
        n = min(n, 2_500_000)
        if data == "dna":
            a = H.dna_ascii(H.rand_dna(rng, n))
        else:
            a = H.aa_ascii(rng.integers(0, 20, n))
        recs = [a]
        r = rng.random()
        if r < 0.25:                                   # a repeat of random length and multiplicity somewhere
            ln, mult, at = int(rng.integers(k + 1, 3000)), int(rng.integers(2, 80)), int(rng.integers(0, max(1, n - 4000)))
            recs = [a[:at] + a[at:at + ln] * mult + a[at:]]
        elif r < 0.4:                                  # records
            cuts = sorted(set(int(x) for x in rng.integers(1, n - 1, int(rng.integers(1, 40)))))
            recs = [a[i:j] for i, j in zip([0] + cuts, cuts + [n])]
        elif r < 0.5 and data == "dna":                # N runs and lower case
            at = int(rng.integers(0, n - 2000))
            recs = [a[:at] + b"NNNNNNNNNNnn" + a[at:at + 1500].lower() + a[at + 1500:]]
        elif r < 0.58:                                 # one k-mer very often
            at = int(rng.integers(0, n - 10))
            recs = [a[:at] + (b"A" if data == "dna" else b"L") * int(rng.integers(300, 90000)) + a[at:]]
        elif r < 0.65:                                 # a part of the genome twice
            recs = [a + a[: n // int(rng.integers(2, 6))]]
        genomes.append(recs)
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != case:
        continue
    sk = G.sketcher_for(G.SeqSketcherParams(k, m, algo, data))
    t0 = time.perf_counter()
    got = sk.sketch_genomes(genomes)
    t1 = time.perf_counter()
    flat = [r for g in genomes for r in g]
    goff = np.cumsum([0] + [len(g) for g in genomes]).astype(np.uint64)
    seq, rs, rl = (O.pack_dna(flat) if data == "dna" else O.filter_aa(flat))
    ref = O.sketch_batch(O.params(k, m, algo, data), seq, rs, rl, goff, nthreads=os.cpu_count())
    ok = got.dtype == ref.dtype and np.array_equal(got.view(np.uint8), ref.view(np.uint8))
    bad += not ok
    print("case %2d %s %s k=%d m=%d genomes=%s: %s (device %.0f ms)" % (case, algo, data, k, m, [sum(len(r) for r in g) for g in genomes], "ok" if ok else "MISMATCH rows %s" % np.nonzero((got.view(np.uint8).reshape(len(got), -1) != ref.view(np.uint8).reshape(len(ref), -1)).any(axis=1))[0].tolist(), (t1 - t0) * 1e3), flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
