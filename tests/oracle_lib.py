"""ctypes binding of the CPU oracle (oracle/libgs_oracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (gsearch_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libgs_oracle.so")

ALGO = {"prob": 0, "super": 1, "super2": 2, "hll": 3, "optdens": 4, "revoptdens": 5}
DATA = {"dna": 0, "aa": 1, "dna_fwd": 2}
KIND_U16, KIND_U32, KIND_U64, KIND_F32 = 0, 1, 2, 3
KIND_DTYPE = {KIND_U16: np.uint16, KIND_U32: np.uint32, KIND_U64: np.uint64, KIND_F32: np.float32}


class Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sketch_size", C.c_uint32), ("algo", C.c_uint32), ("data_t", C.c_uint32)]


def build():
    src = os.path.join(_ORACLE_DIR, "gs_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "CC=gcc"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, u64p = C.c_void_p, C.c_void_p
        L.go_check_params.argtypes = [C.POINTER(Params)]
        L.go_sig_kind.argtypes = [C.POINTER(Params)]
        L.go_sig_elem_bytes.argtypes = [C.POINTER(Params)]
        L.go_sig_elem_bytes.restype = C.c_size_t
        L.go_value_bits.argtypes = [C.POINTER(Params)]
        L.go_pack_dna.argtypes = [vp, C.c_uint64, vp, C.c_uint64]
        L.go_pack_dna.restype = C.c_uint64
        L.go_filter_aa.argtypes = [vp, C.c_uint64, vp]
        L.go_filter_aa.restype = C.c_uint64
        L.go_kmers.argtypes = [C.POINTER(Params), vp, C.c_uint64, C.c_uint64, vp, C.c_uint64]
        L.go_kmers.restype = C.c_uint64
        L.go_test_ln.argtypes = [C.c_double]
        L.go_test_ln.restype = C.c_double
        L.go_sketch_batch.argtypes = [C.POINTER(Params), vp, u64p, u64p, u64p, C.c_uint64, vp, C.c_int]
        L.go_hamming_count.argtypes = [C.c_int, C.c_uint32, vp, vp]
        L.go_hamming_count.restype = C.c_uint32
        L.go_hamming.argtypes = [C.c_int, C.c_uint32, vp, vp]
        L.go_hamming.restype = C.c_float
        L.go_hamming_qxc.argtypes = [C.c_int, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_int]
        L.go_hamming_qxc.restype = None
        L.go_hamming_pairs.argtypes = [C.c_int, C.c_uint32, vp, vp, vp, vp, C.c_uint64, vp]
        L.go_hamming_pairs.restype = None
        L.go_ani.argtypes = [C.c_double, C.c_int, C.c_int]
        L.go_ani.restype = C.c_double
        L.go_index_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int,
                                      C.c_int, C.c_uint64]
        L.go_index_create.restype = C.c_void_p
        L.go_index_destroy.argtypes = [C.c_void_p]
        L.go_index_destroy.restype = None
        L.go_index_insert.argtypes = [C.c_void_p, vp, C.c_uint64, C.c_uint32]
        L.go_index_nb_point.argtypes = [C.c_void_p]
        L.go_index_nb_point.restype = C.c_uint64
        L.go_index_total_evals.argtypes = [C.c_void_p]
        L.go_index_total_evals.restype = C.c_uint64
        L.go_index_search.argtypes = [C.c_void_p, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_int]
        L.go_bruteforce_topk.argtypes = [C.c_int, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_int]
        L.go_index_export.argtypes = [C.c_void_p] + [vp] * 10
        L.go_index_import.argtypes = [C.c_void_p, vp, C.c_uint64, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
        L.go_index_import_view.argtypes = L.go_index_import.argtypes
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def params(k, m, algo, data="dna"):
    return Params(k, m, ALGO[algo] if isinstance(algo, str) else algo, DATA[data] if isinstance(data, str) else data)


def sig_kind(p):
    return lib().go_sig_kind(C.byref(p))


def sig_dtype(p):
    return KIND_DTYPE[sig_kind(p)]


def kind_of_dtype(dt):
    dt = np.dtype(dt)
    return {np.dtype(np.uint16): KIND_U16, np.dtype(np.uint32): KIND_U32, np.dtype(np.uint64): KIND_U64,
            np.dtype(np.float32): KIND_F32}[dt]


def pack_dna(records):
    """records: list of ASCII bytes objects (one genome's records, or a flat list). Returns
    (packed uint8 array, rec_start uint64[], rec_len uint64[]); every record starts 4-base aligned."""
    total = sum(len(r) for r in records) + 4 * len(records) + 64
    packed = np.zeros(total // 4 + 16, dtype=np.uint8)
    starts, lens = [], []
    off = 0
    for r in records:
        a = np.frombuffer(r, dtype=np.uint8)
        n = lib().go_pack_dna(_p(a) if len(a) else None, len(a), _p(packed), off)
        starts.append(off)
        lens.append(n)
        off += (n + 3) // 4 * 4
    return packed, np.array(starts, dtype=np.uint64), np.array(lens, dtype=np.uint64)


def filter_aa(records):
    outs, starts, lens = [], [], []
    off = 0
    for r in records:
        a = np.frombuffer(r, dtype=np.uint8)
        o = np.zeros(max(len(a), 1), dtype=np.uint8)
        n = lib().go_filter_aa(_p(a) if len(a) else None, len(a), _p(o))
        outs.append(o[:n])
        starts.append(off)
        lens.append(n)
        off += n
    seq = np.concatenate(outs) if outs else np.zeros(0, np.uint8)
    if len(seq) == 0:
        seq = np.zeros(1, np.uint8)
    return seq, np.array(starts, dtype=np.uint64), np.array(lens, dtype=np.uint64)


def kmers(p, seq, start, length):
    cap = max(int(length), 1)
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().go_kmers(C.byref(p), _p(seq), int(start), int(length), _p(out), cap)
    return out[:n]


def sketch_batch(p, seq, rec_start, rec_len, genome_rec_off, nthreads=1):
    rc = lib().go_check_params(C.byref(p))
    if rc:
        raise ValueError("invalid sketch parameters (%d)" % rc)
    ng = len(genome_rec_off) - 1
    out = np.zeros((ng, p.sketch_size), dtype=sig_dtype(p))
    genome_rec_off = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
    rc = lib().go_sketch_batch(C.byref(p), _p(seq), _p(rec_start), _p(rec_len), _p(genome_rec_off), ng, _p(out), nthreads)
    assert rc == 0
    return out


def hamming_qxc(Q, Cm, nthreads=1):
    Q = np.ascontiguousarray(Q)
    Cm = np.ascontiguousarray(Cm)
    out = np.zeros((Q.shape[0], Cm.shape[0]), dtype=np.float32)
    lib().go_hamming_qxc(kind_of_dtype(Q.dtype), Q.shape[1], _p(Q), Q.shape[0], _p(Cm), Cm.shape[0], _p(out), nthreads)
    return out


def hamming_pairs(A, B, ia, ib):
    A = np.ascontiguousarray(A)
    B = np.ascontiguousarray(B)
    ia = np.ascontiguousarray(ia, dtype=np.uint64)
    ib = np.ascontiguousarray(ib, dtype=np.uint64)
    out = np.zeros(len(ia), dtype=np.float32)
    lib().go_hamming_pairs(kind_of_dtype(A.dtype), A.shape[1], _p(A), _p(B), _p(ia), _p(ib), len(ia), _p(out))
    return out


def ani(dist, k, model=1):
    return lib().go_ani(float(dist), int(k), int(model))


class Index:
    """Oracle twin of hnsw_rs::Hnsw<Sig, DistHamming> as gsearch drives it (SPEC 5)."""

    def __init__(self, dtype, m, max_nb_conn, ef_construction, max_layer=16, scale_modify=1.0,
                 extend_candidates=True, keep_pruned=False, seed=0):
        self.dtype = np.dtype(dtype)
        self.m, self.M, self.max_layer = m, max_nb_conn, max_layer
        self.h = lib().go_index_create(kind_of_dtype(dtype), m, max_nb_conn, ef_construction, max_layer,
                                       scale_modify, int(extend_candidates), int(keep_pruned), seed)

    def __del__(self):
        if getattr(self, "h", None):
            lib().go_index_destroy(self.h)
            self.h = None

    def parallel_insert(self, sigs, batch=1):
        sigs = np.ascontiguousarray(sigs, dtype=self.dtype)
        assert sigs.shape[1] == self.m
        assert lib().go_index_insert(self.h, _p(sigs), sigs.shape[0], batch) == 0

    def nb_point(self):
        return lib().go_index_nb_point(self.h)

    def import_graph(self, sigs, g, view=False):
        """view=True: keep a pointer to `sigs` instead of a copy (search only; the array is held alive by this object)"""
        sigs = np.ascontiguousarray(sigs, dtype=self.dtype)
        a = {k: np.ascontiguousarray(v) for k, v in g.items() if isinstance(v, np.ndarray)}
        U = int(g["n_upper"])
        if view:
            self._rows = sigs
        rc = (lib().go_index_import_view if view else lib().go_index_import)(self.h, _p(sigs), sigs.shape[0], _p(a["levels"]), int(g["entry"]), _p(a["deg0"]), _p(a["nbr0"]),
                                   _p(a["cnt0"]), _p(a["upidx"]), _p(a["degU"]) if U else None, _p(a["nbrU"]) if U else None,
                                   _p(a["cntU"]) if U else None)
        assert rc == 0

    def total_evals(self):
        return lib().go_index_total_evals(self.h)

    def parallel_search(self, queries, knbn, ef, nthreads=1):
        q = np.ascontiguousarray(queries, dtype=self.dtype)
        nq = q.shape[0]
        ids = np.zeros((nq, knbn), dtype=np.uint64)
        dist = np.zeros((nq, knbn), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        ev = np.zeros(nq, dtype=np.uint64)
        assert lib().go_index_search(self.h, _p(q), nq, knbn, ef, _p(ids), _p(dist), _p(cnt), _p(ev), nthreads) == 0
        return ids, dist, cnt, ev

    def export(self):
        n = self.nb_point()
        M, ML = self.M, self.max_layer
        levels = np.zeros(n, np.uint8)
        entry = np.zeros(1, np.int64)
        deg0 = np.zeros(n, np.uint32)
        nbr0 = np.zeros((n, 2 * M), np.uint32)
        cnt0 = np.zeros((n, 2 * M), np.uint32)
        upidx = np.zeros(n, np.int32)
        nup = np.zeros(1, np.uint64)
        lib().go_index_export(self.h, _p(levels), _p(entry), None, None, None, None, _p(nup), None, None, None)
        U = int(nup[0])
        degU = np.zeros((max(U, 1), ML), np.uint32)
        nbrU = np.zeros((max(U, 1), ML, M), np.uint32)
        cntU = np.zeros((max(U, 1), ML, M), np.uint32)
        lib().go_index_export(self.h, _p(levels), _p(entry), _p(deg0), _p(nbr0), _p(cnt0), _p(upidx), _p(nup),
                              _p(degU), _p(nbrU), _p(cntU))
        return dict(levels=levels, entry=int(entry[0]), deg0=deg0, nbr0=nbr0, cnt0=cnt0, upidx=upidx, n_upper=U,
                    degU=degU[:U], nbrU=nbrU[:U], cntU=cntU[:U])


def bruteforce_topk(db, queries, knbn, nthreads=1):
    db = np.ascontiguousarray(db)
    q = np.ascontiguousarray(queries, dtype=db.dtype)
    ids = np.zeros((q.shape[0], knbn), dtype=np.uint64)
    dist = np.zeros((q.shape[0], knbn), dtype=np.float32)
    lib().go_bruteforce_topk(kind_of_dtype(db.dtype), db.shape[1], _p(db), db.shape[0], _p(q), q.shape[0], knbn,
                             _p(ids), _p(dist), nthreads)
    return ids, dist
