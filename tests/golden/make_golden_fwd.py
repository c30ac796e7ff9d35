#!/usr/bin/env python3
"""Generates tests/golden/golden_v3_fwd.npz: signatures of the golden_v1 DNA genomes under the NON-canonical k-mer closure of bindash-rs
(GS_DATA_DNA_FWD, /root/reference/src/bin/bindash.rs:346-354: k <= 14), from the CPU oracle. Same status as golden_v1 (make_golden.py): pins THIS
repository's SPEC arithmetic; the reference holds no vectors for this path. The key encodes the data type as `dnafwd`.
Run from the repo root:  python tests/golden/make_golden_fwd.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    z = np.load(os.path.join(HERE, "golden_v1.npz"))
    recs = [bytes.fromhex(h) for h in z["dna_records"]]
    goff = z["dna_goff"]
    seq, rs, rl = O.pack_dna(recs)
    out = {}
    for k in (12, 14):
        for m in (64, 1024):
            for algo in ("optdens", "revoptdens"):          # the two sketchers bindash-rs builds (bindash.rs:182-226)
                out["sig_dnafwd_k%d_m%d_%s" % (k, m, algo)] = O.sketch_batch(O.params(k, m, algo, "dna_fwd"), seq, rs, rl, goff)
    np.savez_compressed(os.path.join(HERE, "golden_v3_fwd.npz"), **out)
    print("wrote golden_v3_fwd.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
