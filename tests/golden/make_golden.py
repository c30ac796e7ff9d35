#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz from the CPU oracle (oracle/gs_oracle.c).

The reference repository has no fixtures of its own for this path and cannot be built here, so these vectors
pin THIS repository's SPEC.md arithmetic: they guard the oracle against silent changes and give the GPU tests a second,
run-independent target. Inputs are stored alongside the outputs (ASCII genomes), so the file is data only.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    rng = np.random.default_rng(20240928)
    out = {}
    root = H.rand_dna(rng, 2000)
    dna = [H.dna_ascii(root), H.dna_ascii(H.mutate(rng, root, 0.02)), H.dna_ascii(H.rand_dna(rng, 2000))]
    # one genome with N's / lower case / several records, one too short for any k-mer
    multi = [dna[0][:700] + b"NNNN" + dna[0][700:900].lower(), b"ACGT", dna[1][900:1800]]
    genomes_dna = [[g] for g in dna] + [multi, [b"ACGTN"]]
    aaroot = rng.integers(0, 20, 1500)
    aa = [H.aa_ascii(aaroot), H.aa_ascii(H.mutate(rng, aaroot, 0.05, 20)), H.aa_ascii(rng.integers(0, 20, 1500))]
    genomes_aa = [[g] for g in aa] + [[aa[0][:500] + b"*X", aa[1][400:1400]]]

    def flat(genomes):
        recs = [r for g in genomes for r in g]
        goff = np.cumsum([0] + [len(g) for g in genomes]).astype(np.uint64)
        return recs, goff

    for name, genomes in (("dna", genomes_dna), ("aa", genomes_aa)):
        recs, goff = flat(genomes)
        out["%s_records" % name] = np.array([np.frombuffer(r, np.uint8).tobytes().hex() for r in recs])
        out["%s_goff" % name] = goff
    cases = [("dna", k, m, a) for k in (14, 16, 21) for m in (64, 1024) for a in ("optdens", "revoptdens", "super", "super2", "prob")]
    cases += [("aa", 7, m, a) for m in (64, 1024) for a in ("optdens", "super2", "prob")]
    for data, k, m, algo in cases:
        genomes = genomes_dna if data == "dna" else genomes_aa
        recs, goff = flat(genomes)
        seq, rs, rl = O.pack_dna(recs) if data == "dna" else O.filter_aa(recs)
        sig = O.sketch_batch(O.params(k, m, algo, data), seq, rs, rl, goff)
        out["sig_%s_k%d_m%d_%s" % (data, k, m, algo)] = sig
    # DistHamming + HNSW on a small sketch-level database
    db = H.synth_sig_db(8, 12, 96, 5, jlo=0.1, jhi=0.9)
    q = H.queries_from(db, 10, 6, frac=0.2)
    out["hnsw_db"], out["hnsw_q"] = db, q
    out["hamming_qxdb"] = O.hamming_qxc(q, db)
    ix = O.Index(np.float32, 96, 6, 24, scale_modify=1.0, seed=11)
    ix.parallel_insert(db, batch=8)
    g = ix.export()
    for key in ("levels", "deg0", "nbr0", "cnt0", "upidx", "degU", "nbrU", "cntU"):
        out["hnsw_graph_" + key] = g[key]
    out["hnsw_graph_entry"] = np.array([g["entry"]])
    ids, dist, cnt, ev = ix.parallel_search(q, 5, 30)
    out["hnsw_ids"], out["hnsw_dist"], out["hnsw_evals"] = ids, dist, ev
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_v1.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
