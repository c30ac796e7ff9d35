"""Known-answer tests that pin the CPU oracle to PUBLISHED material (the reference itself has no tests):
 * SplitMix64 / xoshiro256++ reference vectors (Vigna's reference code; the xoshiro vector is the one the
   rand_xoshiro crate tests against), fxhash constants;
 * the distance -> ANI table of the reference README (README.md:231-242) against reformat.rs:80-86.
"""
import pytest
import ctypes as C

import numpy as np

import oracle_lib as O


def _hooks():
    L = O.lib()
    L.go_test_splitmix.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
    L.go_test_xoshiro.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.go_test_seeded.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
    L.go_test_fx.argtypes = [C.c_uint64, C.c_int, C.c_int]
    L.go_test_fx.restype = C.c_uint64
    L.go_test_uint.argtypes = [C.c_uint64, C.c_uint64]
    L.go_test_uint.restype = C.c_uint64
    L.go_test_u64f.argtypes = [C.c_uint64]
    L.go_test_u64f.restype = C.c_double
    L.go_test_u32f.argtypes = [C.c_uint64]
    L.go_test_u32f.restype = C.c_float
    return L


def test_splitmix64_reference_vector():
    out = np.zeros(5, np.uint64)
    _hooks().go_test_splitmix(1234567, 5, out.ctypes.data_as(C.c_void_p))
    assert out.tolist() == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431, 16408922859458223821]


def test_xoshiro256plusplus_reference_vector():
    st = np.array([1, 2, 3, 4], np.uint64)
    out = np.zeros(10, np.uint64)
    _hooks().go_test_xoshiro(st.ctypes.data_as(C.c_void_p), 10, out.ctypes.data_as(C.c_void_p))
    assert out.tolist() == [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205, 9973669472204895162,
                            14011001112246962877, 12406186145184390807, 15849039046786891736, 10450023813501588000]


def test_seed_from_u64_is_splitmix_then_xoshiro():
    L = _hooks()
    sm = np.zeros(4, np.uint64)
    L.go_test_splitmix(42, 4, sm.ctypes.data_as(C.c_void_p))
    a, b = np.zeros(6, np.uint64), np.zeros(6, np.uint64)
    L.go_test_xoshiro(sm.ctypes.data_as(C.c_void_p), 6, a.ctypes.data_as(C.c_void_p))
    L.go_test_seeded(42, 6, b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a, b)


def test_fxhash_constants():
    L = _hooks()
    assert L.go_test_fx(1, 64, 64) == 0x517cc1b727220a95
    assert L.go_test_fx(3, 64, 64) == (3 * 0x517cc1b727220a95) % 2 ** 64
    assert L.go_test_fx(1, 32, 32) == 0x9e3779b9
    # u64 through FxHasher32 = two 32-bit writes, low word first
    h = (1 * 0x9e3779b9) % 2 ** 32
    h = ((((h << 5) | (h >> 27)) & 0xFFFFFFFF) ^ 2) * 0x9e3779b9 % 2 ** 32
    assert L.go_test_fx((2 << 32) | 1, 32, 64) == h


def test_uniform_samplers_follow_rand08():
    L = _hooks()
    first = np.zeros(1, np.uint64)
    for seed in (0, 1, 2 ** 63 + 12345):
        L.go_test_seeded(seed, 1, first.ctypes.data_as(C.c_void_p))
        x = int(first[0])
        assert L.go_test_u64f(seed) == (x >> 12) * 2.0 ** -52
        assert L.go_test_u32f(seed) == np.float32((x >> 41) * 2.0 ** -23)
        for n in (1, 7, 18000, 2 ** 31 + 11):
            assert L.go_test_uint(seed, n) == (x * n) >> 64      # rejection zone misses w.p. n/2^64


README_ANI_TABLE = [  # README.md:231-242, `reformat 16 1`: displayed distance (3 digits) -> ANI
    (5.40e-01, 97.1126), (8.22e-01, 92.5276), (8.71e-01, 90.7837), (8.76e-01, 90.5424), (8.78e-01, 90.4745),
    (8.79e-01, 90.4108), (8.79e-01, 90.398), (8.82e-01, 90.2678), (8.83e-01, 90.2361), (8.86e-01, 90.1098)]


def test_ani_matches_readme_table():
    # the table prints the distance rounded to 3 digits while ANI was computed from the unrounded value:
    # the published ANI must lie between the ANI of the two rounding bounds
    for d, ani in README_ANI_TABLE:
        lo, hi = O.ani(d + 0.0005, 16, 1), O.ani(d - 0.0005, 16, 1)
        assert lo - 1e-4 <= ani <= hi + 1e-4, (d, ani, lo, hi)
    # closed forms of reformat.rs:80-86
    for d in (0.0, 0.3, 0.54, 0.99):
        j = 1.0 - d
        assert abs(O.ani(d, 21, 1) - (1.0 + np.log(2 * j / (1 + j)) / 21) * 100.0) < 1e-9
        assert abs(O.ani(d, 21, 2) - (2 * j / (1 + j)) ** (1.0 / 21) * 100.0) < 1e-9


def test_spec_ln_against_libm():
    """SPEC 2 LN: the + - * / only logarithm the SetSketch registers are computed with agrees with libm to a few ulp"""
    import math
    L = O.lib()
    assert L.go_test_ln(1.0) == 0.0
    xs = [2.0, 0.5, math.e, 1.4142135623730951, 1.4142135623730954, 1e-300, 1e300, 2.0 ** -52, 1 - 2.0 ** -53, 3.7e-9, 0.999, 1.001]
    rng = np.random.default_rng(4)
    xs += list(np.exp(rng.uniform(-40, 40, 2000)))
    for x in xs:
        got, ref = L.go_test_ln(float(x)), math.log(float(x))
        assert abs(got - ref) <= 4e-16 * max(1.0, abs(ref)), (x, got, ref)
    # log_b steps of the register formula: 1/LN(1.001) and one register value by hand: x = 1e-8 -> 1 - ln(1e-8)/ln(1.001) = 18430.9...
    assert int(1.0 - L.go_test_ln(1e-8) / L.go_test_ln(1.001)) == 18430


@pytest.mark.parametrize("k,m,data", [(12, 64, "dna_fwd"), (14, 96, "dna_fwd"), (7, 50, "dna_fwd"), (12, 64, "dna"), (21, 80, "dna"), (16, 64, "dna")])
def test_oracle_sketch_equals_an_independent_python_restatement(k, m, data):
    """golden_v3_fwd.npz is generated by the oracle itself, so it cannot catch a forward-only closure that the oracle and the kernels share
    but that differs from bindash.rs:346-354. tests/pyref.py restates SPEC.md 1.1 / 2 / 3.1 in pure Python without sharing a line with the oracle: the
    optdens signatures of small multi-record genomes (N runs, lower case, a record shorter than k, few enough k-mers that densification runs) must agree
    bit for bit - for the forward-only closure, for the canonical one, and the two closures must differ on the same genome."""
    import numpy as np
    import helpers as H
    import oracle_lib as O
    import pyref
    rng = np.random.default_rng(k * 7 + m)
    a = H.dna_ascii(H.rand_dna(rng, 900))
    genomes = [[a[:400] + b"NNnn" + a[400:600].lower(), b"ACGT", a[600:]], [a[100:160]], [H.revcomp_ascii(a[:400])]]
    recs = [r for g in genomes for r in g]
    goff = np.cumsum([0] + [len(g) for g in genomes]).astype(np.uint64)
    seq, rs, rl = O.pack_dna(recs)
    got = O.sketch_batch(O.params(k, m, "optdens", data), seq, rs, rl, goff).view(np.uint32)
    for gi, g in enumerate(genomes):
        want = np.array(pyref.optdens(g, k, m, forward_only=(data == "dna_fwd")), dtype=np.uint32)
        assert np.array_equal(got[gi], want), gi
    other = O.sketch_batch(O.params(k, m, "optdens", "dna" if data == "dna_fwd" else "dna_fwd"), seq, rs, rl, goff).view(np.uint32)
    assert not np.array_equal(other[0], got[0])
