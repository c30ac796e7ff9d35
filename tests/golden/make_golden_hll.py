#!/usr/bin/env python3
"""Generates tests/golden/golden_v2_hll.npz: SetSketch (--algo hll, SPEC 3.4) signatures of the golden_v1 genomes, from the CPU oracle.
Same status as golden_v1 (make_golden.py): pins THIS repository's SPEC arithmetic; the reference holds no vectors for this path.
Run from the repo root:  python tests/golden/make_golden_hll.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    z = np.load(os.path.join(HERE, "golden_v1.npz"))
    out = {}
    for data, k in (("dna", 14), ("dna", 16), ("dna", 21), ("aa", 7)):
        recs = [bytes.fromhex(h) for h in z["%s_records" % data]]
        goff = z["%s_goff" % data]
        seq, rs, rl = O.pack_dna(recs) if data == "dna" else O.filter_aa(recs)
        for m in (64, 1024):
            out["sig_%s_k%d_m%d_hll" % (data, k, m)] = O.sketch_batch(O.params(k, m, "hll", data), seq, rs, rl, goff)
    np.savez_compressed(os.path.join(HERE, "golden_v2_hll.npz"), **out)
    print("wrote golden_v2_hll.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
