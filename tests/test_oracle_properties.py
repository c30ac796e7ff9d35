"""CPU tests of the oracle: a pure-Python restatement for small cases, the input contract of the reference's
readers, and algebraic properties that hold for ANY correct implementation of the path."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O

ALGOS = ["optdens", "revoptdens", "super", "super2", "prob"]
SET_ALGOS = ALGOS + ["hll"]


def py_kmers_dna(s, k):
    """kmer_hash_fn closure of dnasketch.rs:164-169 in plain Python on the filtered sequence."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    c = [code[ch] for ch in s.upper() if ch in code]
    out = []
    for i in range(len(c) - k + 1):
        w = c[i:i + k]
        fwd = 0
        for x in w:
            fwd = fwd * 4 + x
        rc = 0
        for x in reversed(w):
            rc = rc * 4 + (3 - x)
        out.append(min(fwd, rc) & (4 ** k - 1))
    return out


def py_kmers_aa(s, k):
    alpha = "ACDEFGHIKLMNPQRSTVWY"
    c = [alpha.index(ch) for ch in s.upper() if ch in alpha]
    out = []
    for i in range(len(c) - k + 1):
        v = 0
        for x in c[i:i + k]:
            v = v * 32 + x
        out.append(v & (32 ** k - 1))
    return out


@pytest.mark.parametrize("k", [1, 2, 7, 14, 16, 21, 31, 32])
def test_kmers_match_pure_python(k):
    rng = np.random.default_rng(k)
    s = H.dna_ascii(H.rand_dna(rng, 300)).decode()
    s = s[:50] + "NNnn" + s[50:120].lower() + "xyz-" + s[120:]
    seq, rs, rl = O.pack_dna([s.encode()])
    got = O.kmers(O.params(k, 64, "optdens"), seq, rs[0], rl[0]).tolist()
    assert got == py_kmers_dna(s, k)
    assert int(rl[0]) == 300            # non-ACGT dropped, lower case kept (dnafiles.rs:41,150-151)


@pytest.mark.parametrize("k", [1, 5, 6, 7, 12])
def test_aa_kmers_match_pure_python(k):
    rng = np.random.default_rng(k)
    s = H.aa_ascii(rng.integers(0, 20, 200)).decode()
    s = s[:30] + "*BXZ" + s[30:90].lower() + s[90:]
    seq, rs, rl = O.filter_aa([s.encode()])
    got = O.kmers(O.params(k, 64, "optdens", "aa"), seq, rs[0], rl[0]).tolist()
    assert got == py_kmers_aa(s, k)
    assert int(rl[0]) == 200


def test_parameter_dispatch_follows_reference_tables():
    # (data, algo, k) -> Sig of dnasketch.rs:499-642 / aasketch.rs:455-550
    assert O.sig_dtype(O.params(21, 100, "optdens")) == np.float32
    assert O.sig_dtype(O.params(21, 100, "super")) == np.float32
    assert O.sig_dtype(O.params(14, 100, "prob")) == np.uint32
    assert O.sig_dtype(O.params(16, 100, "prob")) == np.uint32
    assert O.sig_dtype(O.params(17, 100, "prob")) == np.uint64
    assert O.sig_dtype(O.params(16, 100, "super2")) == np.uint32
    assert O.sig_dtype(O.params(21, 100, "super2")) == np.uint64
    assert O.sig_dtype(O.params(6, 100, "super2", "aa")) == np.uint32
    assert O.sig_dtype(O.params(7, 100, "super2", "aa")) == np.uint64
    assert O.sig_dtype(O.params(21, 100, "hll")) == np.uint16 and O.sig_dtype(O.params(7, 100, "hll", "aa")) == np.uint16    # dnasketch.rs:562-584
    for bad in (O.params(15, 100, "optdens"), O.params(33, 100, "optdens"), O.params(13, 100, "optdens", "aa")):
        with pytest.raises(ValueError):
            O.sketch_batch(bad, np.zeros(8, np.uint8), np.zeros(1, np.uint64), np.zeros(1, np.uint64), np.array([0, 1], np.uint64))


def _sketch(algo, k, m, genomes, data="dna"):
    recs = [r for g in genomes for r in g]
    goff = np.cumsum([0] + [len(g) for g in genomes]).astype(np.uint64)
    seq, rs, rl = (O.pack_dna(recs) if data != "aa" else O.filter_aa(recs))
    return O.sketch_batch(O.params(k, m, algo, data), seq, rs, rl, goff)


@pytest.mark.parametrize("algo", SET_ALGOS)
def test_identical_and_reverse_complement_invariance(algo):
    rng = np.random.default_rng(3)
    g = H.dna_ascii(H.rand_dna(rng, 20000))
    S = _sketch(algo, 21, 512, [[g], [g], [H.revcomp_ascii(g)], [g[:9000], g[9000 - 20:]], [g[9000 - 20:], g[:9000]]])
    assert np.array_equal(S[0], S[1])
    assert np.array_equal(S[0], S[2])                 # canonical k-mers: strand does not matter
    # records overlapping by k-1 bases carry the same k-mer set except the two duplicated across the cut;
    # record ORDER never matters (one signature per file, dnasketch.rs:357-359)
    assert np.array_equal(S[3], S[4])
    if algo != "prob":                                # prob is multiset sensitive, the others are set sketches
        assert np.array_equal(S[0], S[3])


@pytest.mark.parametrize("k", [12, 14, 16, 21])
def test_forward_only_kmers_are_the_windows_as_read(k):
    """GS_DATA_DNA_FWD = the k <= 14 closure of bindash.rs:346-354 (`kmer.get_compressed_value() & mask`): the value is the 2-bit window itself
    (pure-Python restatement, SPEC 1.2), the canonical value is min(window, reverse complement of the window)."""
    rng = np.random.default_rng(k)
    g = H.dna_ascii(H.rand_dna(rng, 3000))
    seq, rs, rl = O.pack_dna([g])
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    c = [code[b] for b in g]
    fwd = [sum(c[i + j] << (2 * (k - 1 - j)) for j in range(k)) for i in range(len(c) - k + 1)]
    rcv = [sum((3 - c[i + k - 1 - j]) << (2 * (k - 1 - j)) for j in range(k)) for i in range(len(c) - k + 1)]
    assert O.kmers(O.params(k, 64, "optdens", "dna_fwd"), seq, rs[0], rl[0]).tolist() == fwd
    assert O.kmers(O.params(k, 64, "optdens", "dna"), seq, rs[0], rl[0]).tolist() == [min(a, b) for a, b in zip(fwd, rcv)]


@pytest.mark.parametrize("algo", SET_ALGOS)
def test_forward_only_sketch_is_strand_specific(algo):
    """the non-canonical closure must NOT be strand invariant (a sketcher that canonicalised anyway would pass every other property), must still
    ignore record order, and must differ from the canonical sketch of the same genome"""
    rng = np.random.default_rng(5)
    g = H.dna_ascii(H.rand_dna(rng, 20000))
    genomes = [[g], [H.revcomp_ascii(g)], [g[:9000], g[9000 - 11:]], [g[9000 - 11:], g[:9000]]]
    F = _sketch(algo, 12, 512, genomes, "dna_fwd")
    Cn = _sketch(algo, 12, 512, genomes[:1], "dna")
    assert not np.array_equal(F[0], F[1])
    assert np.array_equal(F[2], F[3]) and np.array_equal(F[0], F[2])
    assert not np.array_equal(F[0], Cn[0])
    assert F.dtype == Cn.dtype                        # same (k -> Kmer::Val, Sig) table as canonical DNA


@pytest.mark.parametrize("algo", ALGOS)
def test_jaccard_estimate_within_3_sigma(algo):
    rng = np.random.default_rng(11)
    k, m = 21, 4000
    fam = H.family(rng, 80000, [0.005, 0.02, 0.06])
    genomes = [[H.dna_ascii(g)] for g in fam]
    S = _sketch(algo, k, m, genomes)
    seq, rs, rl = O.pack_dna([g[0] for g in genomes])
    sets = [set(O.kmers(O.params(k, m, "optdens"), seq, rs[i], rl[i]).tolist()) for i in range(len(genomes))]
    for i in range(1, len(genomes)):
        J = len(sets[0] & sets[i]) / len(sets[0] | sets[i])
        est = 1.0 - O.hamming_qxc(S[:1], S[i:i + 1])[0, 0]
        sigma = np.sqrt(J * (1 - J) / m)
        assert abs(est - J) < 3 * sigma + 2e-3, (algo, i, J, est)


@pytest.mark.parametrize("algo", ["optdens", "super", "super2"])
def test_min_sketch_mergeability(algo):
    """a set sketch of A u B is the slot-wise min of the sketches of A and B when no bin is empty
    (what lets one genome be split over several workgroups)."""
    rng = np.random.default_rng(5)
    a, b = H.dna_ascii(H.rand_dna(rng, 30000)), H.dna_ascii(H.rand_dna(rng, 30000))
    S = _sketch(algo, 21, 256, [[a], [b], [a, b]])
    assert np.array_equal(np.minimum(S[0], S[1]), S[2])


def test_hll_registers_merge_by_max_and_track_jaccard():
    """SetSketch (SPEC 3.4): the registers of A u B are the slot-wise MAX of those of A and of B - exactly, for any split, which is what
    lets the reference sketch 10 M-base blocks independently and merge (dnasketch.rs:553) - and the fraction of equal registers follows
    the Jaccard index (b = 1.001: chance collisions of different maxima are rare)."""
    rng = np.random.default_rng(15)
    a, b = H.dna_ascii(H.rand_dna(rng, 40000)), H.dna_ascii(H.rand_dna(rng, 25000))
    S = _sketch("hll", 21, 256, [[a], [b], [a, b], [b, a], [a[:100], a[80:]]])
    assert S.dtype == np.uint16
    assert np.array_equal(np.maximum(S[0], S[1]), S[2]) and np.array_equal(S[2], S[3])
    assert np.array_equal(S[0], S[4])                                     # an overlap of k-1 keeps the k-mer set
    assert (S[0] > 0).all() and S[0].max() < 65535
    k, m = 21, 2000
    fam = H.family(rng, 60000, [0.005, 0.03])
    G3 = _sketch("hll", k, m, [[H.dna_ascii(g)] for g in fam])
    seq, rs, rl = O.pack_dna([H.dna_ascii(g) for g in fam])
    sets = [set(O.kmers(O.params(k, m, "optdens"), seq, rs[i], rl[i]).tolist()) for i in range(3)]
    est = []
    for i in (1, 2):
        J = len(sets[0] & sets[i]) / len(sets[0] | sets[i])
        e = 1.0 - O.hamming_qxc(G3[:1], G3[i:i + 1])[0, 0]
        est.append(e)
        assert abs(e - J) < 4 * np.sqrt(J * (1 - J) / m) + 0.02, (i, J, e)
    assert est[0] > est[1]
    # tiny input: registers hit by no point stay 0; no k-mer at all: all 0
    T = _sketch("hll", 21, 4096, [[H.dna_ascii(H.rand_dna(rng, 60))], [b"ACG"]])
    assert (T[1] == 0).all() and 0 < (T[0] > 0).sum() <= 4096


def test_prob_signature_is_subset_of_kmers_and_weight_sensitive():
    rng = np.random.default_rng(6)
    g = H.dna_ascii(H.rand_dna(rng, 5000))
    rep = g[:200] * 40
    S = _sketch("prob", 21, 512, [[g], [g, rep]])
    seq, rs, rl = O.pack_dna([g, rep])
    ks = set(O.kmers(O.params(21, 512, "prob"), seq, rs[0], rl[0]).tolist()) | set(O.kmers(O.params(21, 512, "prob"), seq, rs[1], rl[1]).tolist())
    assert set(S[0].tolist()) <= ks and set(S[1].tolist()) <= ks
    repk = set(O.kmers(O.params(21, 512, "prob"), seq, rs[1], rl[1]).tolist())
    # heavy k-mers (multiplicity 40) win far more slots in the weighted sketch
    assert np.isin(S[1], list(repk)).mean() > 5 * np.isin(S[0], list(repk)).mean() + 0.2


def test_densification_fills_every_bin_and_is_consistent():
    rng = np.random.default_rng(7)
    g = H.dna_ascii(H.rand_dna(rng, 400))
    for algo in ("optdens", "revoptdens"):
        S = _sketch(algo, 21, 4096, [[g], [g + b"ACGTACGTAACCGGTTACGTAGCTAGCTAGGATCGATCGA"], [b"ACG"]])
        assert ((S >= 0) & (S <= 1)).all()
        assert len(set(S[0].tolist())) <= 400            # only ~380 distinct values spread over 4096 bins
        assert (S[0] == S[1]).mean() > 0.5                # densification preserves similarity
        assert (S[2] == 1.0).all()                         # no k-mer at all


def test_hamming_semantics():
    a = np.array([[0.0, np.nan, 1.0, -0.0]], dtype=np.float32)
    b = np.array([[-0.0, np.nan, 1.0, 0.0]], dtype=np.float32)
    assert O.hamming_qxc(a, b)[0, 0] == np.float32(0.25)       # float `!=`: NaN != NaN, -0 == +0
    u = np.arange(12, dtype=np.uint64).reshape(2, 6)
    assert O.hamming_qxc(u[:1], u[1:])[0, 0] == 1.0
    assert O.hamming_qxc(u[:1], u[:1])[0, 0] == 0.0
    x = np.array([[1, 2, 3]], dtype=np.uint32)
    y = np.array([[1, 5, 3]], dtype=np.uint32)
    assert O.hamming_qxc(x, y)[0, 0] == np.float32(1) / np.float32(3)


@pytest.mark.parametrize("B", [1, 8])
def test_hnsw_search_is_consistent_with_bruteforce(B):
    db = H.synth_sig_db(12, 30, 200, 9, jlo=0.2, jhi=0.9)
    ix = O.Index(np.float32, 200, 12, 64, seed=5)
    ix.parallel_insert(db, batch=B)
    q = H.queries_from(db, 40, 10, frac=0.2)
    ids, dist, cnt, ev = ix.parallel_search(q, 10, 400)
    bi, bd = O.bruteforce_topk(db, q, 10)
    assert (cnt == 10).all()
    assert (np.diff(dist, axis=1) >= 0).all()                 # ascending
    rec = np.mean([(dist[i] <= bd[i, -1]).mean() for i in range(len(q))])
    assert rec >= 0.99
    # every reported distance is the true distance of the reported id
    for i in range(5):
        true = O.hamming_qxc(q[i:i + 1], db[ids[i].astype(np.int64)])[0]
        assert np.array_equal(true, dist[i])
    # deterministic
    ids2, dist2, _, ev2 = ix.parallel_search(q, 10, 400, nthreads=4)
    assert np.array_equal(ids, ids2) and np.array_equal(ev, ev2)


def test_hnsw_export_invariants():
    db = H.synth_sig_db(10, 20, 100, 2, jlo=0.05, jhi=0.95)
    M = 6
    ix = O.Index(np.float32, 100, M, 40, seed=9)
    ix.parallel_insert(db, batch=4)
    g = ix.export()
    n = len(db)
    assert (g["deg0"] <= 2 * M).all() and (g["degU"] <= M).all()
    assert g["levels"][g["entry"]] == g["levels"].max()
    for i in range(n):
        d = int(g["deg0"][i])
        nb = g["nbr0"][i, :d]
        assert len(set(nb.tolist())) == d and i not in nb
        keys = g["cnt0"][i, :d].astype(np.uint64) << np.uint64(32) | nb.astype(np.uint64)
        assert (np.diff(keys.astype(np.int64)) > 0).all()     # sorted by (count, id)
        true = np.rint(O.hamming_qxc(db[i:i + 1], db[nb.astype(np.int64)])[0] * 100).astype(np.uint32) if d else np.zeros(0, np.uint32)
        assert np.array_equal(true, g["cnt0"][i, :d])


def test_index_import_view_searches_like_a_copy():
    """go_index_import_view keeps a pointer to the caller's rows (the 300 000-node GPU tests hand the oracle 21.6 GB this way): same answers as the
    copying import, and a borrowed index refuses to grow"""
    db = H.synth_sig_db(5, 40, 96, 3)
    a = O.Index(np.float32, 96, 6, 30, seed=2)
    a.parallel_insert(db, batch=16)
    g = a.export()
    b, c = O.Index(np.float32, 96, 6, 30, seed=2), O.Index(np.float32, 96, 6, 30, seed=2)
    b.import_graph(db, g)
    c.import_graph(db, g, view=True)
    q = H.queries_from(db, 12, 5)
    for x, y, z in zip(a.parallel_search(q, 5, 40), b.parallel_search(q, 5, 40), c.parallel_search(q, 5, 40)):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert O.lib().go_index_insert(c.h, db.ctypes.data_as(O.C.c_void_p), 1, 1) != 0


def test_bgzf_helper_writes_valid_gzip():
    import gzip
    data = bytes(np.random.default_rng(0).integers(65, 70, 300_000).astype(np.uint8))
    blob = H.bgzf_bytes(data, block=40_000)
    assert gzip.decompress(blob) == data and blob.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
