"""Host-side mirror of the reference's operator interface for the sketch-and-query hot path.

Names and argument meaning follow the Rust traits gsearch calls (paths relative to /root/reference):

* ``SeqSketcherParams`` / ``*HashSketch.new(params)`` / ``sketch_compressedkmer`` / ``sketch_compressedkmer_seqs``
  — kmerutils::sketching::setsketchert::SeqSketcherT, called at src/dna/dnasketch.rs:336,357,
  src/dna/dnarequest.rs:272,287, src/aa/aasketch.rs:313,329.
* ``DistHamming.eval`` — anndists::dist::DistHamming (src/dna/dnasketch.rs:72, src/bin/bindash.rs:93-99).
* ``Hnsw.new / modify_level_scale / set_extend_candidates / set_keeping_pruned / parallel_insert /
  parallel_search / get_nb_point`` — hnsw_rs::Hnsw (src/dna/dnasketch.rs:139-141,159-160,435; src/dna/dnarequest.rs:353).

Everything here is a thin ctypes shell over the C ABI (include/gsearch_amd.h); the arithmetic runs in the
HIP library. There is no CPU fallback.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import ALGO, DATA, KIND_F32, KIND_U16, KIND_U32, KIND_U64, GsError, IndexParams, SketchParams, check

KIND_DTYPE = {KIND_U16: np.dtype(np.uint16), KIND_U32: np.dtype(np.uint32), KIND_U64: np.dtype(np.uint64),
              KIND_F32: np.dtype(np.float32)}
DTYPE_KIND = {v: k for k, v in KIND_DTYPE.items()}

Neighbour = namedtuple("Neighbour", ["d_id", "distance", "p_id"], defaults=[None])   # hnsw_rs::Neighbour; gsearch reads d_id and distance (answer.rs:42,55-57); p_id = (layer, rank in layer)

# At interpreter exit the HIP runtime may already be torn down when Python finalises leftover objects: destroying device
# objects then would call into a dead runtime. After this flag is set __del__ becomes a no-op (the OS reclaims everything).
_exiting = [False]
import atexit  # noqa: E402
atexit.register(lambda: _exiting.__setitem__(0, True))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One per (process, GPU): owns the HIP stream all calls are enqueued on."""

    def __init__(self, device_id=0, stream=None):
        self.L = _lib.load()
        h = C.c_void_p()
        check(self.L.gs_ctx_create(C.byref(h), device_id, stream))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.gs_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if not _exiting[0]:
                self.close()
        except Exception:
            pass

    def sync(self):
        check(self.L.gs_ctx_sync(self.h))

    def release_scratch(self):
        """give back the device scratch the context keeps between calls (it grows on demand)"""
        check(self.L.gs_ctx_release_scratch(self.h))

    def device_info(self):
        ncu, hbm, name = C.c_int(), C.c_uint64(), C.create_string_buffer(128)
        check(self.L.gs_ctx_device_info(self.h, C.byref(ncu), C.byref(hbm), name, 128))
        return {"n_cu": ncu.value, "hbm_bytes": hbm.value, "name": name.value.decode()}

    def last_sketch_info(self):
        """form of the slot-min sketch kernel the last sketch call launched (include/gsearch_amd.h gs_ctx_last_sketch_info)"""
        out = np.zeros(4, np.uint32)
        check(self.L.gs_ctx_last_sketch_info(self.h, _p(out)))
        return {"filtered": bool(out[0]), "table_in_lds": bool(out[1]), "workgroups_per_genome": int(out[2]), "launches": int(out[3])}

    # stopwatch / per-family kernel timers (HIP events on the context's stream)
    def timer_start(self):
        check(self.L.gs_ctx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        check(self.L.gs_ctx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, enable=True):
        check(self.L.gs_ctx_profile(self.h, int(enable)))

    def profile_read(self, family, reset=True):
        ms, n = C.c_double(), C.c_uint64()
        check(self.L.gs_ctx_profile_read(self.h, family, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    # device memory
    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self.L.gs_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        check(self.L.gs_dev_free(self.h, ptr))

    def upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self.L.gs_dev_upload(self.h, ptr, _p(arr), arr.nbytes))

    def download(self, ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        check(self.L.gs_dev_download(self.h, _p(out), ptr, out.nbytes))
        return out

    def memset(self, ptr, byte, nbytes):
        check(self.L.gs_dev_memset(self.h, ptr, byte, nbytes))


class Comm:
    """RCCL communicator of the path's one exchange step (include/gsearch_amd.h gs_comm_*): all-gather of the per-rank top-k blocks."""

    def __init__(self, ctx, n_ranks, rank, unique_id):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(self.L.gs_comm_create(ctx.h, int(n_ranks), int(rank), buf, C.byref(h)))
        self.h, self.n_ranks, self.rank = h, int(n_ranks), int(rank)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(_lib.load().gs_comm_unique_id(buf))
        return buf.raw

    def allgather_topk_dev(self, ids_dev, dist_dev, nq_local, knbn, all_ids_dev, all_dist_dev):
        check(self.L.gs_comm_allgather_topk_dev(self.h, ids_dev, dist_dev, nq_local, knbn, all_ids_dev, all_dist_dev))

    def allgatherv_topk_dev(self, ids_dev, dist_dev, nq_local, nq_max, knbn, all_ids_dev, all_dist_dev):
        """unequal shards: this rank's nq_local (<= nq_max, may be 0) rows -> compact concatenation in rank order; returns every rank's count"""
        counts = np.zeros(self.n_ranks, dtype=np.uint64)
        check(self.L.gs_comm_allgatherv_topk_dev(self.h, ids_dev, dist_dev, int(nq_local), int(nq_max), int(knbn), all_ids_dev, all_dist_dev, _p(counts)))
        return counts

    def allgatherv_topk_async_dev(self, ids_dev, dist_dev, nq_local, nq_max, knbn, all_ids_dev, all_dist_dev, counts_dev=None):
        """the exchange queued on the context's stream, no host round trip (gs_comm_allgatherv_topk_async_dev); wait() is the synchronising half"""
        check(self.L.gs_comm_allgatherv_topk_async_dev(self.h, ids_dev, dist_dev, int(nq_local), int(nq_max), int(knbn), all_ids_dev, all_dist_dev, counts_dev))

    def wait(self):
        """waits for the context's stream, checks the last exchange's block shapes, returns every rank's count (gs_comm_wait)"""
        counts = np.zeros(self.n_ranks, dtype=np.uint64)
        check(self.L.gs_comm_wait(self.h, _p(counts)))
        return counts

    def size(self):
        return int(self.L.gs_comm_size(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.gs_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if not _exiting[0]:
                self.close()
        except Exception:
            pass


def topk_block_bytes(nq_max, knbn):
    return int(_lib.load().gs_topk_block_bytes(int(nq_max), int(knbn)))


def topk_pack(ids, dist, nq_max):
    """host form of the exchange's block layout (gs_topk_pack): (nq_local, knbn) ids / distances -> one fixed-size block (uint8 array)"""
    ids = np.ascontiguousarray(ids, dtype=np.uint64); dist = np.ascontiguousarray(dist, dtype=np.float32)
    nq, knbn = ids.shape
    out = np.zeros(topk_block_bytes(nq_max, knbn), dtype=np.uint8)
    check(_lib.load().gs_topk_pack(_p(ids) if nq else None, _p(dist) if nq else None, nq, int(nq_max), knbn, _p(out)))
    return out


def topk_unpack(blocks, n_ranks, nq_max, knbn):
    """n_ranks blocks back to back -> (compact ids, compact distances, counts per rank), rank order (gs_topk_unpack)"""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    counts = np.zeros(n_ranks, dtype=np.uint64)
    check(_lib.load().gs_topk_unpack(_p(blocks), int(n_ranks), int(nq_max), int(knbn), None, None, _p(counts)))
    tot = int(counts.sum())
    ids, dist = np.zeros((tot, knbn), np.uint64), np.zeros((tot, knbn), np.float32)
    check(_lib.load().gs_topk_unpack(_p(blocks), int(n_ranks), int(nq_max), int(knbn), _p(ids), _p(dist), _p(counts)))
    return ids, dist, counts


def topk_merge_dev(ctx, ids_dev, dist_dev, n_shards, nq, knbn_in, knbn_out, out_ids_dev, out_dist_dev, id_offset=None):
    """DB-sharded alternative: merge the answers of n_shards shards for the same nq queries on the device (gs_topk_merge_dev)"""
    off = None if id_offset is None else np.ascontiguousarray(id_offset, dtype=np.uint64)
    check(ctx.L.gs_topk_merge_dev(ctx.h, ids_dev, dist_dev, int(n_shards), int(nq), int(knbn_in), _p(off), int(knbn_out), out_ids_dev, out_dist_dev))


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ----------------------------------------------------------------------------------------------------------
class SeqSketcherParams:
    """kmerutils::sketcharg::SeqSketcherParams::new(kmer_size, sketch_size, algo, data_t) (gsearch.rs:258-263)."""

    def __init__(self, kmer_size, sketch_size, algo, data_t="dna"):
        self.c = SketchParams(int(kmer_size), int(sketch_size), ALGO[algo] if isinstance(algo, str) else int(algo),
                              DATA[data_t] if isinstance(data_t, str) else int(data_t))
        check(_lib.load().gs_check_params(C.byref(self.c)))

    def get_kmer_size(self):
        return self.c.k

    def get_sketch_size(self):
        return self.c.sketch_size

    def sig_kind(self):
        return _lib.load().gs_sig_kind(C.byref(self.c))

    def sig_dtype(self):
        return KIND_DTYPE[self.sig_kind()]


def pack_dna_records(records):
    """ASCII records -> (packed 2-bit buffer, rec_start, rec_len); each record starts on a byte boundary.
    Same filtering as Sequence::encode_and_add (dnafiles.rs:70-71): non-ACGT dropped, case folded."""
    L = _lib.load()
    total = sum(len(r) for r in records) + 4 * len(records)
    packed = np.zeros(total // 4 + 64, dtype=np.uint8)
    starts = np.zeros(len(records), dtype=np.uint64)
    lens = np.zeros(len(records), dtype=np.uint64)
    off = 0
    for i, r in enumerate(records):
        a = np.frombuffer(r, dtype=np.uint8)
        n = L.gs_pack_dna(_p(a) if len(a) else None, len(a), _p(packed), off)
        starts[i], lens[i] = off, n
        off += (n + 3) // 4 * 4
    return packed[: (off + 3) // 4 + 8], starts, lens


def filter_aa_records(records):
    L = _lib.load()
    outs = []
    starts = np.zeros(len(records), dtype=np.uint64)
    lens = np.zeros(len(records), dtype=np.uint64)
    off = 0
    for i, r in enumerate(records):
        a = np.frombuffer(r, dtype=np.uint8)
        o = np.zeros(max(len(a), 1), dtype=np.uint8)
        n = L.gs_filter_aa(_p(a) if len(a) else None, len(a), _p(o))
        outs.append(o[:n])
        starts[i], lens[i] = off, n
        off += n
    seq = np.concatenate(outs) if outs else np.zeros(0, np.uint8)
    return np.concatenate([seq, np.zeros(8, np.uint8)]), starts, lens


class _SeqSketcher:
    """Common shell of the SeqSketcherT implementations. A genome is a list of records (ASCII bytes)."""
    ALGO_NAME = None

    def __init__(self, params, ctx=None):
        if self.ALGO_NAME is not None and params.c.algo != ALGO[self.ALGO_NAME]:
            raise GsError(_lib.GS_ERR_INVALID, "params.algo does not match %s" % type(self).__name__)
        self.params = params
        self.ctx = ctx or default_context()

    @classmethod
    def new(cls, params, ctx=None):
        return cls(params, ctx)

    def sig_dtype(self):
        return self.params.sig_dtype()

    def sketch_packed(self, seq, rec_start, rec_len, genome_rec_off):
        """Lowest level: already packed input (the layout of include/gsearch_amd.h gs_sketch_batch)."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        rec_start = np.ascontiguousarray(rec_start, dtype=np.uint64)
        rec_len = np.ascontiguousarray(rec_len, dtype=np.uint64)
        goff = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
        ng = len(goff) - 1
        out = np.zeros((ng, self.params.c.sketch_size), dtype=self.sig_dtype())
        check(self.ctx.L.gs_sketch_batch(self.ctx.h, C.byref(self.params.c), _p(seq), seq.nbytes, _p(rec_start), _p(rec_len),
                                         len(rec_start), _p(goff), ng, _p(out)))
        return out

    def _pack(self, records):
        if self.params.c.data_t != DATA["aa"]:
            return pack_dna_records(records)
        return filter_aa_records(records)

    def sketch_compressedkmer_seqs(self, vseq):
        """All sequences are ONE genome -> exactly one signature (assert at dnasketch.rs:359)."""
        seq, rs, rl = self._pack(list(vseq))
        return [row for row in self.sketch_packed(seq, rs, rl, np.array([0, len(rs)], dtype=np.uint64))]

    def sketch_compressedkmer(self, vseq):
        """One signature per input sequence, in input order (assert at dnasketch.rs:338)."""
        seq, rs, rl = self._pack(list(vseq))
        return [row for row in self.sketch_packed(seq, rs, rl, np.arange(len(rs) + 1, dtype=np.uint64))]

    def sketch_files(self, paths, block=False, pio=0, threads=0):
        """the reader side of sketchandstore_dir_compressedkmer (dnasketch.rs:240-300) for a list of FASTA files (plain / gz / bz2 / xz):
        host threads read + decode + scan groups of `pio` files while the previous group crosses PCIe and the one before is packed and
        sketched. -> ((n_files, m) signatures, records kept per file, symbols sketched per file, stats dict)"""
        paths = [str(x).encode() for x in paths]
        n = len(paths)
        arr = (C.c_char_p * max(n, 1))(*paths)
        out = np.zeros((n, self.params.c.sketch_size), dtype=self.sig_dtype())
        nrec, nsym, st = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64), np.zeros(6, np.float64)
        check(self.ctx.L.gs_sketch_files_ex(self.ctx.h, C.byref(self.params.c), arr, n, int(bool(block)), int(pio), int(threads), _p(out), _p(nrec), _p(nsym), _p(st), len(st)))
        return out, nrec[:n], nsym[:n], {"host_read_decode_scan_s": st[0], "pcie_wait_s": st[1], "device_s": st[2], "wall_s": st[3],
                                         "gz_members_inflated_on_device": int(st[4]), "gz_members_handed_back_to_host": int(st[5])}

    def sketch_genomes(self, genomes):
        """Batch form used by the drivers: genomes = list of lists of records -> (n_genomes, m) array."""
        recs, goff = [], [0]
        for g in genomes:
            recs.extend(g)
            goff.append(len(recs))
        seq, rs, rl = self._pack(recs)
        return self.sketch_packed(seq, rs, rl, np.array(goff, dtype=np.uint64))


def is_fasta_file(path, data_t="dna"):
    """files.rs:117-146 is_fasta_dna_file / is_fasta_aa_file"""
    return bool(_lib.load().gs_is_fasta_file(str(path).encode(), DATA[data_t] if isinstance(data_t, str) else int(data_t)))


def read_fasta_file(path):
    """file_to_buffer + needletail's transparent gz / bz2 / xz decoding (files.rs:220-250): the decompressed text as bytes"""
    L = _lib.load()
    p, n = C.c_void_p(), C.c_uint64()
    check(L.gs_read_fasta_file(str(path).encode(), C.byref(p), C.byref(n)))
    try:
        return C.string_at(p, n.value)
    finally:
        L.gs_host_free(p)


def gunzip_batch(ctx, members, out_caps=None):
    """gzip members (bytes objects, one single-member file each) inflated on the device (gs_inflate.hip): list of (status, text bytes).
    status 0 = the deflate data, ISIZE and CRC-32 all check; see gs_gunzip_batch in include/gsearch_amd.h for the others."""
    n = len(members)
    if n == 0:
        return []
    ins = [np.frombuffer(m, dtype=np.uint8) if len(m) else np.zeros(1, np.uint8) for m in members]
    caps = [int(c) for c in out_caps] if out_caps is not None else [
        (int.from_bytes(m[-4:], "little") if len(m) >= 18 else 0) for m in members]
    outs = [np.empty(max(c, 1) + 64, dtype=np.uint8) for c in caps]
    pin = (C.c_void_p * n)(*[a.ctypes.data for a in ins])
    pout = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    lin = (C.c_uint64 * n)(*[len(m) for m in members])
    cap = (C.c_uint64 * n)(*caps)
    lout = (C.c_uint64 * n)()
    st = (C.c_int * n)()
    check(ctx.L.gs_gunzip_batch(ctx.h, pin, lin, n, pout, cap, lout, st))
    return [(int(st[i]), outs[i][:lout[i]].tobytes()) for i in range(n)]


def list_fasta_files(directory, data_t="dna"):
    """recursive directory walk of process_dir (files.rs:148-215): accepted files in name order"""
    L = _lib.load()
    dt = DATA[data_t] if isinstance(data_t, str) else int(data_t)
    n, nb = C.c_uint64(), C.c_uint64()
    check(L.gs_list_fasta_files(str(directory).encode(), dt, None, 0, C.byref(n), C.byref(nb)))
    buf = C.create_string_buffer(max(nb.value, 1))
    check(L.gs_list_fasta_files(str(directory).encode(), dt, buf, nb.value, C.byref(n), C.byref(nb)))
    return [x.decode() for x in buf.raw[:nb.value].split(b"\0") if x]


def fasta_scan(text, skip_capsid=True):
    """record boundaries of a FASTA text (bytes): list of (id, seq_begin, seq_end) — the reader side of dnafiles.rs:43-193"""
    L = _lib.load()
    buf = np.frombuffer(text, dtype=np.uint8)
    n = C.c_uint64()
    check(L.gs_fasta_scan(_p(buf) if len(buf) else None, len(buf), int(skip_capsid), 0, None, None, None, None, C.byref(n)))
    nr = n.value
    sb, se, ib = np.zeros(nr, np.uint64), np.zeros(nr, np.uint64), np.zeros(nr, np.uint64)
    il = np.zeros(nr, np.uint32)
    check(L.gs_fasta_scan(_p(buf) if len(buf) else None, len(buf), int(skip_capsid), nr, _p(sb), _p(se), _p(ib), _p(il), C.byref(n)))
    return [(bytes(text[int(ib[i]):int(ib[i]) + int(il[i])]).decode("ascii", "replace"), int(sb[i]), int(se[i])) for i in range(nr)]


def sketch_fasta_files(sketcher, files, skip_capsid=True):
    """files: list of FASTA texts (bytes), one genome each -> ((n_files, m) signatures, (rec_start, rec_len, packed)).
    Record splitting on the host, filtering + 2-bit packing + sketching on the device; one signature per file, k-mers never
    span records (by-sequence mode, dnasketch.rs:348-363)."""
    ctx, L = sketcher.ctx, sketcher.ctx.L
    text = b"".join(files)
    offs = np.cumsum([0] + [len(f) for f in files])
    sb, se, goff = [], [], [0]
    for fi, f in enumerate(files):
        recs = fasta_scan(f, skip_capsid)
        for _, b, e in recs:
            sb.append(int(offs[fi]) + b)
            se.append(int(offs[fi]) + e)
        goff.append(len(sb))
    nrec = len(sb)
    sb, se = np.array(sb, dtype=np.uint64), np.array(se, dtype=np.uint64)
    tbuf = np.frombuffer(text, dtype=np.uint8)
    d_text = ctx.alloc(len(tbuf) + 64)
    pbytes = len(tbuf) // 4 + 8 * nrec + 128
    d_packed = ctx.alloc(pbytes)
    try:
        ctx.upload(d_text, tbuf)
        ctx.memset(d_packed, 0, pbytes)
        rs, rl = np.zeros(max(nrec, 1), np.uint64), np.zeros(max(nrec, 1), np.uint64)
        check(L.gs_pack_fasta_dev(ctx.h, d_text, len(tbuf), _p(sb), _p(se), nrec, d_packed, _p(rs), _p(rl)))
        m = sketcher.params.c.sketch_size
        ng = len(files)
        goff = np.array(goff, dtype=np.uint64)
        d_rs, d_rl, d_go = ctx.alloc(8 * max(nrec, 1)), ctx.alloc(8 * max(nrec, 1)), ctx.alloc(8 * (ng + 1))
        d_sig = ctx.alloc(ng * m * sketcher.sig_dtype().itemsize)
        try:
            ctx.upload(d_rs, rs); ctx.upload(d_rl, rl); ctx.upload(d_go, goff)
            check(L.gs_sketch_batch_dev(ctx.h, C.byref(sketcher.params.c), d_packed, pbytes // 8 * 8, d_rs, d_rl, nrec, d_go, ng, d_sig))
            out = ctx.download(d_sig, (ng, m), sketcher.sig_dtype())
        finally:
            for p_ in (d_rs, d_rl, d_go, d_sig):
                ctx.free(p_)
        return out, (rs[:nrec], rl[:nrec], ctx.download(d_packed, (pbytes,), np.uint8))
    finally:
        ctx.free(d_text)
        ctx.free(d_packed)


class OptDensHashSketch(_SeqSketcher):
    ALGO_NAME = "optdens"


class RevOptDensHashSketch(_SeqSketcher):
    ALGO_NAME = "revoptdens"


class ProbHash3aSketch(_SeqSketcher):
    ALGO_NAME = "prob"


class SuperHashSketch(_SeqSketcher):
    ALGO_NAME = "super"


class SuperHash2Sketch(_SeqSketcher):
    ALGO_NAME = "super2"


class HyperLogLogSketch(_SeqSketcher):
    """kmerutils HyperLogLogSketch<Kmer, u16>: SetSketch registers with SetSketchParams::default() + set_m(sketch_size)
    (dnasketch.rs:541-574, aasketch.rs:481-500)"""
    ALGO_NAME = "hll"


def sketcher_for(params, ctx=None):
    """(algo) dispatch of dna_process_tohnsw (dnasketch.rs:493-644)."""
    table = {ALGO["optdens"]: OptDensHashSketch, ALGO["revoptdens"]: RevOptDensHashSketch, ALGO["prob"]: ProbHash3aSketch,
             ALGO["super"]: SuperHashSketch, ALGO["super2"]: SuperHash2Sketch, ALGO["hll"]: HyperLogLogSketch}
    return table[params.c.algo](params, ctx)


# ----------------------------------------------------------------------------------------------------------
class DistHamming:
    """anndists::dist::DistHamming — eval(a, b) = count(a[i] != b[i]) / len, f32."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def eval(self, va, vb):
        va = np.ascontiguousarray(va)
        vb = np.ascontiguousarray(vb, dtype=va.dtype)
        return float(self.eval_qxc(va[None, :], vb[None, :])[0, 0])

    def eval_qxc(self, Q, Cm):
        Q = np.ascontiguousarray(Q)
        Cm = np.ascontiguousarray(Cm, dtype=Q.dtype)
        if Q.shape[1] != Cm.shape[1]:
            raise GsError(_lib.GS_ERR_INVALID, "signature lengths differ")
        out = np.zeros((Q.shape[0], Cm.shape[0]), dtype=np.float32)
        check(self.ctx.L.gs_hamming_qxc(self.ctx.h, DTYPE_KIND[Q.dtype], Q.shape[1], _p(Q), Q.shape[0], _p(Cm), Cm.shape[0], _p(out)))
        return out

    def eval_pairs(self, A, B, ia, ib):
        A = np.ascontiguousarray(A)
        B = np.ascontiguousarray(B, dtype=A.dtype)
        ia = np.ascontiguousarray(ia, dtype=np.uint64)
        ib = np.ascontiguousarray(ib, dtype=np.uint64)
        out = np.zeros(len(ia), dtype=np.float32)
        check(self.ctx.L.gs_hamming_pairs(self.ctx.h, DTYPE_KIND[A.dtype], A.shape[1], _p(A), A.shape[0], _p(B), B.shape[0],
                                          _p(ia), _p(ib), len(ia), _p(out)))
        return out


def ani(distance, kmer_size, model=1):
    """reformat.rs:80-86 calculate_ani."""
    return _lib.load().gs_ani(float(distance), int(kmer_size), int(model))


def bindash_sketch_params(kmer_size, sketch_size, dens=0):
    """The sketcher bindash-rs builds for (kmer_size, dens): OptDens (dens = 0) or RevOptDens (dens = 1) over f32 (bindash.rs:182-226), with the
    k-mer closure of its three branches - k <= 14: the forward window, NOT canonical (bindash.rs:346-354); k = 16 and 17..32: canonical
    (bindash.rs:366-377, 388-397)."""
    if dens not in (0, 1):
        raise ValueError("Only densification = 0 or 1 are supported!")          # bindash.rs:227-229
    return SeqSketcherParams(kmer_size, sketch_size, "optdens" if dens == 0 else "revoptdens", "dna_fwd" if kmer_size <= 14 else "dna")


def bindash_distance(hamming_distance, kmer_size):
    """bindash.rs:93-99 compute_distance: j = 1 - d ; 1 - (2j/(1+j))^(1/k), with the f32 powf the reference uses."""
    j = np.float32(1.0) - np.float32(hamming_distance)
    frac = np.float32(2.0) * j / (np.float32(1.0) + j)
    return float(1.0 - float(np.power(frac, np.float32(1.0) / np.float32(kmer_size), dtype=np.float32)))


class ReqAnswer:
    """src/answer.rs:18-76 — text record of one request; only neighbours with distance < threshold are written
    (out_threshold = 0.99, dnarequest.rs:83). `seqdict` = list of (path, fasta_id, length) indexed by d_id."""

    def __init__(self, rank, req_item, neighbours):
        self.rank, self.req_item, self.neighbours = rank, req_item, neighbours

    def dump(self, seqdict, threshold, out):
        if not any(n.distance <= threshold for n in self.neighbours):
            return 0
        path, fasta_id, length = self.req_item
        out.write("\n%d\t%s\tfasta_id:\t%s\tlength:\t%d" % (self.rank, path, fasta_id, length))
        nb_match = 0
        for n in self.neighbours:
            if n.distance < threshold:
                nb_match += 1
                dpath, dfid, dlen = seqdict[n.d_id]
                out.write("\nquery_id:\t%s\tdistance:\t%s\tanswer_fasta_path\t%s\t%s \t answer_seq_len:\t %d"
                          % (path, _rust_5e(n.distance), dpath, dfid, dlen))
        return nb_match


def _rust_5e(x):
    """Rust's {:.5E}: mantissa with 5 decimals, exponent without padding or plus sign (6.07500E-1)"""
    mant, exp = ("%.5E" % x).split("E")
    return "%sE%d" % (mant, int(exp))


# ----------------------------------------------------------------------------------------------------------
class Hnsw:
    """hnsw_rs::Hnsw<Sig, DistHamming> as gsearch uses it."""

    def __init__(self, max_nb_connection, max_elements, max_layer, ef_construction, dist_f=None, dtype=np.float32,
                 sketch_size=None, seed=0, insert_batch=0, ctx=None):
        self.ctx = ctx or (dist_f.ctx if dist_f is not None else default_context())
        self.prm = IndexParams(DTYPE_KIND[np.dtype(dtype)], int(sketch_size or 0), int(max_nb_connection), int(max_elements),
                               int(max_layer), int(ef_construction), 1.0, 0, 0, int(seed), int(insert_batch))
        self.dtype = np.dtype(dtype)
        self.h = None

    @classmethod
    def new(cls, max_nb_connection, max_elements, max_layer, ef_construction, dist_f=None, **kw):
        return cls(max_nb_connection, max_elements, max_layer, ef_construction, dist_f, **kw)

    # setters are only legal before the first point, like the reference's use (dnasketch.rs:141,159-160)
    def modify_level_scale(self, scale_modification):
        self._frozen_check()
        self.prm.scale_modify = float(scale_modification)

    def set_extend_candidates(self, flag):
        self._frozen_check()
        self.prm.extend_candidates = int(bool(flag))

    def set_keeping_pruned(self, flag):
        self._frozen_check()
        self.prm.keep_pruned = int(bool(flag))

    def _frozen_check(self):
        if self.h is not None:
            raise GsError(_lib.GS_ERR_STATE, "index parameters are frozen once the index holds points")

    def _ensure(self, m):
        if self.h is None:
            if self.prm.m == 0:
                self.prm.m = int(m)
            h = C.c_void_p()
            check(self.ctx.L.gs_index_create(self.ctx.h, C.byref(self.prm), C.byref(h)))
            self.h = h
        if int(m) != self.prm.m:
            raise GsError(_lib.GS_ERR_INVALID, "signature length %d != index length %d" % (m, self.prm.m))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.gs_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if not _exiting[0]:
                self.close()
        except Exception:
            pass

    def get_nb_point(self):
        return 0 if self.h is None else self.ctx.L.gs_index_nb_point(self.h)

    def parallel_insert(self, datas, ids=None):
        """datas: (n, m) array (ids: optional DataIds, default nb_point..), or a list of (vector, id) pairs like the reference's
        parallel_insert(&[(&Vec<Sig>, usize)]) (dnasketch.rs:426-435); searches return the ids as d_id"""
        if isinstance(datas, (list, tuple)) and len(datas) and isinstance(datas[0], tuple):
            ids = [did for _, did in datas]
            datas = np.stack([d for d, _ in datas])
        datas = np.ascontiguousarray(datas, dtype=self.dtype)
        self._ensure(datas.shape[1])
        if ids is None:
            check(self.ctx.L.gs_index_parallel_insert(self.h, _p(datas), datas.shape[0]))
        else:
            ids = np.ascontiguousarray(ids, dtype=np.uint64)
            if len(ids) != datas.shape[0]:
                raise GsError(_lib.GS_ERR_INVALID, "one id per vector")
            check(self.ctx.L.gs_index_parallel_insert_ids(self.h, _p(datas), _p(ids), datas.shape[0]))

    def set_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        check(self.ctx.L.gs_index_set_ids(self.h, _p(ids), len(ids)))

    def get_ids(self, first=0, n=None):
        n = self.get_nb_point() - first if n is None else n
        out = np.zeros(n, dtype=np.uint64)
        check(self.ctx.L.gs_index_get_ids(self.h, first, n, _p(out)))
        return out

    def search_arrays_pid(self, datas, knbn, ef):
        """search_arrays plus hnsw_rs' PointId of every neighbour: (ids, dist, cnt, evals, pid_layer, pid_rank)"""
        datas = np.ascontiguousarray(datas, dtype=self.dtype)
        if self.h is None:
            raise GsError(_lib.GS_ERR_STATE, "search on an empty index")
        nq = datas.shape[0]
        ids, dist = np.zeros((nq, knbn), np.uint64), np.zeros((nq, knbn), np.float32)
        cnt, ev = np.zeros(nq, np.uint32), np.zeros(nq, np.uint64)
        pl, pr = np.zeros((nq, knbn), np.uint8), np.zeros((nq, knbn), np.int32)
        check(self.ctx.L.gs_index_parallel_search_pid(self.h, _p(datas), nq, knbn, ef, _p(ids), _p(dist), _p(cnt), _p(ev), _p(pl), _p(pr)))
        return ids, dist, cnt, ev, pl, pr

    def search_arrays(self, datas, knbn, ef):
        datas = np.ascontiguousarray(datas, dtype=self.dtype)
        if self.h is None:
            raise GsError(_lib.GS_ERR_STATE, "search on an empty index")
        nq = datas.shape[0]
        ids = np.zeros((nq, knbn), dtype=np.uint64)
        dist = np.zeros((nq, knbn), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        ev = np.zeros(nq, dtype=np.uint64)
        check(self.ctx.L.gs_index_parallel_search(self.h, _p(datas), nq, knbn, ef, _p(ids), _p(dist), _p(cnt), _p(ev)))
        return ids, dist, cnt, ev

    def parallel_search(self, datas, knbn, ef):
        """-> Vec<Vec<Neighbour>>, ascending distance (dnarequest.rs:353)."""
        ids, dist, cnt, _, pl, pr = self.search_arrays_pid(datas, knbn, ef)
        return [[Neighbour(int(ids[i, j]), float(dist[i, j]), (int(pl[i, j]), int(pr[i, j]))) for j in range(int(cnt[i]))] for i in range(len(ids))]

    def count_matrix(self, datas):
        """mismatch counts of every query against every node (nq x nb_point, uint16) from the dense producer the search would use"""
        datas = np.ascontiguousarray(datas, dtype=self.dtype)
        out = np.zeros((datas.shape[0], self.get_nb_point()), dtype=np.uint16)
        check(self.ctx.L.gs_index_count_matrix(self.h, _p(datas), datas.shape[0], _p(out)))
        return out

    def sketch_and_search_dev(self, params, d_seq, seq_bytes, d_rec_start, d_rec_len, n_rec, d_genome_rec_off, n_genomes, knbn, ef, d_ids, d_dist, d_count=None,
                              d_evals=None, d_sig=None):
        """sketch_and_request on device-resident genomes (gs_index_sketch_and_search_dev): every d_* is a device pointer (int)"""
        check(self.ctx.L.gs_index_sketch_and_search_dev(self.h, C.byref(params.c), d_seq, seq_bytes, d_rec_start, d_rec_len, n_rec, d_genome_rec_off, n_genomes, d_sig,
                                                        knbn, ef, d_ids, d_dist, d_count, d_evals))

    def bruteforce_search(self, datas, knbn):
        datas = np.ascontiguousarray(datas, dtype=self.dtype)
        ids = np.zeros((datas.shape[0], knbn), dtype=np.uint64)
        dist = np.zeros((datas.shape[0], knbn), dtype=np.float32)
        check(self.ctx.L.gs_index_bruteforce_search(self.h, _p(datas), datas.shape[0], knbn, _p(ids), _p(dist)))
        return ids, dist

    # graph import/export in the library's dense layout (role of HnswIo::load_hnsw / file_dump)
    def import_graph(self, sigs, g):
        sigs = np.ascontiguousarray(sigs, dtype=self.dtype)
        self._ensure(sigs.shape[1])
        a = {k: np.ascontiguousarray(v) for k, v in g.items() if isinstance(v, np.ndarray)}
        U = int(g["n_upper"])
        check(self.ctx.L.gs_index_import(self.h, _p(sigs), sigs.shape[0], _p(a["levels"]), int(g["entry"]), _p(a["deg0"]),
                                         _p(a["nbr0"]), _p(a["cnt0"]), _p(a["upidx"]), U,
                                         _p(a["degU"]) if U else None, _p(a["nbrU"]) if U else None, _p(a["cntU"]) if U else None))

    def export_graph(self):
        n = self.get_nb_point()
        M, ML = self.prm.max_nb_conn, self.prm.max_layer
        levels = np.zeros(n, np.uint8)
        entry = np.zeros(1, np.int64)
        nup = np.zeros(1, np.uint64)
        check(self.ctx.L.gs_index_export(self.h, None, _p(entry), None, None, None, None, _p(nup), None, None, None))
        U = int(nup[0])
        deg0, nbr0, cnt0 = np.zeros(n, np.uint32), np.zeros((n, 2 * M), np.uint32), np.zeros((n, 2 * M), np.uint32)
        upidx = np.zeros(n, np.int32)
        degU, nbrU, cntU = np.zeros((max(U, 1), ML), np.uint32), np.zeros((max(U, 1), ML, M), np.uint32), np.zeros((max(U, 1), ML, M), np.uint32)
        check(self.ctx.L.gs_index_export(self.h, _p(levels), _p(entry), _p(deg0), _p(nbr0), _p(cnt0), _p(upidx), _p(nup),
                                         _p(degU), _p(nbrU), _p(cntU)))
        return dict(levels=levels, entry=int(entry[0]), deg0=deg0, nbr0=nbr0, cnt0=cnt0, upidx=upidx, n_upper=U,
                    degU=degU[:U], nbrU=nbrU[:U], cntU=cntU[:U])

    def get_data(self, first=0, n=None):
        n = self.get_nb_point() - first if n is None else n
        out = np.zeros((n, self.prm.m), dtype=self.dtype)
        check(self.ctx.L.gs_index_get_data(self.h, first, n, _p(out)))
        return out

    def file_dump(self, path):
        """Hnsw::file_dump counterpart (own format, dumpload.rs:31)"""
        check(self.ctx.L.gs_index_save(self.h, str(path).encode()))

    @classmethod
    def load(cls, path, ctx=None):
        """HnswIo::load_hnsw counterpart (reloadhnsw.rs:41-51)"""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(ctx.L.gs_index_load(ctx.h, str(path).encode(), C.byref(h)))
        self = cls.__new__(cls)
        self.ctx, self.h = ctx, h
        self.prm = IndexParams()
        check(ctx.L.gs_index_get_params(h, C.byref(self.prm)))
        self.dtype = KIND_DTYPE[self.prm.kind]
        return self

    def search_stats(self, reset=False):
        """device-side work counters since the last reset (include/gsearch_amd.h gs_index_search_stats)"""
        if self.h is None:
            return {}
        out = np.zeros(8, dtype=np.uint64)
        check(self.ctx.L.gs_index_search_stats(self.h, _p(out), int(reset)))
        pops = int(out[1])
        return {"join_atomics": int(out[0]), "pops": pops, "accepting_pops": int(out[2]), "wg_in_flight": int(out[3]),
                "adj_bytes": pops * int(out[4]), "pops_phase1": int(out[5]), "pops_phase2": int(out[6]), "join_shared_expansions": int(out[7])}

    def file_dump_hnswrs(self, basename, truncate_255=False):
        """Hnsw::file_dump(dir, "hnswdump") in hnsw_rs' own format: <basename>.hnsw.graph + <basename>.hnsw.data (dumpload.rs:26-31).
        truncate_255: the format keeps neighbour counts in one byte; cut longer layer-0 lists to their 255 closest entries instead of refusing"""
        check(self.ctx.L.gs_index_dump_hnswrs_ex(self.h, str(basename).encode(), 1 if truncate_255 else 0))

    @classmethod
    def load_hnswrs(cls, basename, hint=None, ctx=None):
        """HnswIo::load_hnsw counterpart for hnsw_rs dumps (reloadhnsw.rs:41-51); `hint`: an Hnsw whose parameters (capacity, level scale,
        flags, seed, insert batch) later insertions should use"""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(ctx.L.gs_index_load_hnswrs(ctx.h, str(basename).encode(), C.byref(hint.prm) if hint is not None else None, C.byref(h)))
        self = cls.__new__(cls)
        self.ctx, self.h = ctx, h
        self.prm = IndexParams()
        check(ctx.L.gs_index_get_params(h, C.byref(self.prm)))
        self.dtype = KIND_DTYPE[self.prm.kind]
        return self

    def insert_evals(self):
        return 0 if self.h is None else self.ctx.L.gs_index_insert_evals(self.h)
