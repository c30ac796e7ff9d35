"""Committed golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them (CPU), and the HIP path must reproduce them through the C ABI (GPU)."""
import os

import numpy as np
import pytest

import oracle_lib as O

G_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")


G2_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v2_hll.npz")
G3_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v3_fwd.npz")


def _load():
    z = np.load(G_PATH)
    genomes = {}
    for name in ("dna", "aa"):
        recs = [bytes.fromhex(h) for h in z["%s_records" % name]]
        goff = z["%s_goff" % name]
        genomes[name] = [recs[int(goff[i]):int(goff[i + 1])] for i in range(len(goff) - 1)]
    cases = []
    for key in z.files:
        if key.startswith("sig_"):
            _, data, k, m, algo = key.split("_")
            cases.append((data, int(k[1:]), int(m[1:]), algo, z[key]))
    z2 = np.load(G2_PATH)                       # SetSketch signatures of the same genomes (make_golden_hll.py)
    for key in z2.files:
        _, data, k, m, algo = key.split("_")
        cases.append((data, int(k[1:]), int(m[1:]), algo, z2[key]))
    z3 = np.load(G3_PATH)                       # forward-only k-mers (bindash.rs:346-354) of the same DNA genomes (make_golden_fwd.py)
    genomes["dnafwd"] = genomes["dna"]
    for key in z3.files:
        _, data, k, m, algo = key.split("_")
        cases.append((data, int(k[1:]), int(m[1:]), algo, z3[key]))
    return z, genomes, cases


DATA_OF = {"dna": "dna", "aa": "aa", "dnafwd": "dna_fwd"}


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_oracle_reproduces_golden_sketches():
    z, genomes, cases = _load()
    assert len(cases) == 36 + 8 + 8
    for data, k, m, algo, want in cases:
        recs = [r for g in genomes[data] for r in g]
        goff = np.cumsum([0] + [len(g) for g in genomes[data]]).astype(np.uint64)
        seq, rs, rl = O.pack_dna(recs) if data != "aa" else O.filter_aa(recs)
        got = O.sketch_batch(O.params(k, m, algo, DATA_OF[data]), seq, rs, rl, goff)
        assert got.dtype == want.dtype and np.array_equal(_bits(got), _bits(want)), (data, k, m, algo)


def test_oracle_reproduces_golden_hnsw():
    z, _, _ = _load()
    db, q = z["hnsw_db"], z["hnsw_q"]
    assert np.array_equal(O.hamming_qxc(q, db), z["hamming_qxdb"])
    ix = O.Index(np.float32, 96, 6, 24, scale_modify=1.0, seed=11)
    ix.parallel_insert(db, batch=8)
    g = ix.export()
    for key in ("levels", "deg0", "nbr0", "cnt0", "upidx", "degU", "nbrU", "cntU"):
        assert np.array_equal(g[key], z["hnsw_graph_" + key]), key
    ids, dist, cnt, ev = ix.parallel_search(q, 5, 30)
    assert np.array_equal(ids, z["hnsw_ids"]) and np.array_equal(dist, z["hnsw_dist"]) and np.array_equal(ev, z["hnsw_evals"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_sketches(gpu_ctx):
    import gsearch_amd as G
    z, genomes, cases = _load()
    for data, k, m, algo, want in cases:
        got = G.sketcher_for(G.SeqSketcherParams(k, m, algo, DATA_OF[data])).sketch_genomes(genomes[data])
        assert got.dtype == want.dtype and np.array_equal(_bits(got), _bits(want)), (data, k, m, algo)


@pytest.mark.gpu
def test_gpu_reproduces_golden_hnsw(gpu_ctx):
    import gsearch_amd as G
    z, _, _ = _load()
    db, q = z["hnsw_db"], z["hnsw_q"]
    assert np.array_equal(G.DistHamming().eval_qxc(q, db), z["hamming_qxdb"])
    hn = G.Hnsw.new(6, 1000, 16, 24, G.DistHamming(), seed=11, insert_batch=8)
    hn.set_extend_candidates(True)
    hn.parallel_insert(db)
    g = hn.export_graph()
    assert np.array_equal(g["deg0"], z["hnsw_graph_deg0"]) and g["entry"] == int(z["hnsw_graph_entry"][0])
    for i in range(len(db)):
        d = int(g["deg0"][i])
        assert np.array_equal(g["nbr0"][i, :d], z["hnsw_graph_nbr0"][i, :d])
    ids, dist, cnt, ev = hn.search_arrays(q, 5, 30)
    assert np.array_equal(ids, z["hnsw_ids"]) and np.array_equal(dist, z["hnsw_dist"]) and np.array_equal(ev, z["hnsw_evals"])
