"""ctypes binding of libgsearch_amd.so (the C ABI declared in include/gsearch_amd.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("GS_LIB_PATH") or os.path.join(_HERE, "libgsearch_amd.so")      # GS_LIB_PATH: A/B builds (tools/ only)

GS_OK, GS_ERR_INVALID, GS_ERR_HIP, GS_ERR_UNSUPPORTED, GS_ERR_STATE, GS_ERR_IO = 0, -1, -2, -3, -4, -5
ALGO = {"prob": 0, "super": 1, "super2": 2, "hll": 3, "optdens": 4, "revoptdens": 5}
DATA = {"dna": 0, "aa": 1, "dna_fwd": 2}   # dna_fwd: forward window, no reverse-complement minimum (bindash.rs:346-354, k <= 14)
KIND_U16, KIND_U32, KIND_U64, KIND_F32 = 0, 1, 2, 3


class GsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gsearch_amd error %d: %s" % (code, msg))
        self.code = code


class SketchParams(C.Structure):
    """kmerutils::sketcharg::SeqSketcherParams {kmer_size, sketch_size, algo, data_t}"""
    _fields_ = [("k", C.c_uint32), ("sketch_size", C.c_uint32), ("algo", C.c_uint32), ("data_t", C.c_uint32)]


class IndexParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("m", C.c_uint32), ("max_nb_conn", C.c_uint32), ("capacity", C.c_uint64),
                ("max_layer", C.c_uint32), ("ef_construction", C.c_uint32), ("scale_modify", C.c_double),
                ("extend_candidates", C.c_int), ("keep_pruned", C.c_int), ("seed", C.c_uint64),
                ("insert_batch", C.c_uint32)]


# every symbol include/gsearch_amd.h declares: name -> (restype, argtypes)
_vp, _u64, _u32, _i = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
_PP = C.POINTER(SketchParams)
SYMBOLS = {
    "gs_last_error": (C.c_char_p, []),
    "gs_version": (C.c_char_p, []),
    "gs_ctx_create": (_i, [C.POINTER(_vp), _i, _vp]),
    "gs_ctx_destroy": (None, [_vp]),
    "gs_ctx_sync": (_i, [_vp]),
    "gs_ctx_release_scratch": (_i, [_vp]),
    "gs_ctx_stream": (_vp, [_vp]),
    "gs_ctx_device_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_u64), C.c_char_p, C.c_size_t]),
    "gs_ctx_last_sketch_info": (_i, [_vp, _vp]),
    "gs_ctx_timer_start": (_i, [_vp]),
    "gs_ctx_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "gs_ctx_profile": (_i, [_vp, _i]),
    "gs_ctx_profile_read": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_u64), _i]),
    "gs_dev_alloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "gs_dev_free": (_i, [_vp, _vp]),
    "gs_dev_upload": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "gs_dev_download": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "gs_dev_memset": (_i, [_vp, _vp, _i, C.c_size_t]),
    "gs_check_params": (_i, [_PP]),
    "gs_sig_kind": (_i, [_PP]),
    "gs_sig_elem_bytes": (C.c_size_t, [_PP]),
    "gs_value_bits": (_i, [_PP]),
    "gs_sketch_batch": (_i, [_vp, _PP, _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp]),
    "gs_sketch_batch_dev": (_i, [_vp, _PP, _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp]),
    "gs_fasta_scan": (_i, [_vp, _u64, _i, _u64, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "gs_pack_fasta_dev": (_i, [_vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp]),
    "gs_filter_aa_dev": (_i, [_vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp]),
    "gs_is_fasta_file": (_i, [C.c_char_p, _i]),
    "gs_read_fasta_file": (_i, [C.c_char_p, C.POINTER(_vp), C.POINTER(_u64)]),
    "gs_host_free": (None, [_vp]),
    "gs_list_fasta_files": (_i, [C.c_char_p, _i, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "gs_sketch_files": (_i, [_vp, _PP, C.POINTER(C.c_char_p), _u64, _i, _u32, _u32, _vp, _vp, _vp, _vp]),
    "gs_sketch_files_ex": (_i, [_vp, _PP, C.POINTER(C.c_char_p), _u64, _i, _u32, _u32, _vp, _vp, _vp, _vp, _u32]),
    "gs_gunzip_batch": (_i, [_vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "gs_pack_dna": (_u64, [_vp, _u64, _vp, _u64]),
    "gs_filter_aa": (_u64, [_vp, _u64, _vp]),
    "gs_hamming_qxc": (_i, [_vp, _i, _u32, _vp, _u64, _vp, _u64, _vp]),
    "gs_hamming_qxc_dev": (_i, [_vp, _i, _u32, _vp, _u64, _vp, _u64, _vp]),
    "gs_hamming_pairs": (_i, [_vp, _i, _u32, _vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp]),
    "gs_ani": (C.c_double, [C.c_double, _i, _i]),
    "gs_index_create": (_i, [_vp, C.POINTER(IndexParams), C.POINTER(_vp)]),
    "gs_index_destroy": (None, [_vp]),
    "gs_index_nb_point": (_u64, [_vp]),
    "gs_index_get_params": (_i, [_vp, C.POINTER(IndexParams)]),
    "gs_index_parallel_insert": (_i, [_vp, _vp, _u64]),
    "gs_index_parallel_insert_dev": (_i, [_vp, _vp, _u64]),
    "gs_index_parallel_insert_ids": (_i, [_vp, _vp, _vp, _u64]),
    "gs_index_parallel_insert_ids_dev": (_i, [_vp, _vp, _vp, _u64]),
    "gs_index_sketch_and_search_dev": (_i, [_vp, _PP, _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "gs_index_set_ids": (_i, [_vp, _vp, _u64]),
    "gs_index_get_ids": (_i, [_vp, _u64, _u64, _vp]),
    "gs_index_parallel_search_pid": (_i, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gs_index_parallel_search_pid_dev": (_i, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gs_index_parallel_search": (_i, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "gs_index_parallel_search_dev": (_i, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "gs_index_count_matrix": (_i, [_vp, _vp, _u64, _vp]),
    "gs_index_bruteforce_search": (_i, [_vp, _vp, _u64, _u32, _vp, _vp]),
    "gs_index_import": (_i, [_vp, _vp, _u64, _vp, C.c_int64, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp]),
    "gs_index_export": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gs_index_get_data": (_i, [_vp, _u64, _u64, _vp]),
    "gs_index_save": (_i, [_vp, C.c_char_p]),
    "gs_index_load": (_i, [_vp, C.c_char_p, C.POINTER(_vp)]),
    "gs_index_dump_hnswrs": (_i, [_vp, C.c_char_p]),
    "gs_index_dump_hnswrs_ex": (_i, [_vp, C.c_char_p, _u32]),
    "gs_index_load_hnswrs": (_i, [_vp, C.c_char_p, C.POINTER(IndexParams), C.POINTER(_vp)]),
    "gs_index_insert_evals": (_u64, [_vp]),
    "gs_index_search_stats": (_i, [_vp, _vp, _i]),
    "gs_comm_unique_id": (_i, [_vp]),
    "gs_comm_create": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "gs_comm_destroy": (None, [_vp]),
    "gs_comm_rank": (_i, [_vp]),
    "gs_comm_size": (_i, [_vp]),
    "gs_comm_allgather_topk_dev": (_i, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    "gs_comm_allgatherv_topk_dev": (_i, [_vp, _vp, _vp, _u64, _u64, _u32, _vp, _vp, _vp]),
    "gs_comm_allgatherv_topk_async_dev": (_i, [_vp, _vp, _vp, _u64, _u64, _u32, _vp, _vp, _vp]),
    "gs_comm_wait": (_i, [_vp, _vp]),
    "gs_index_release_build_scratch": (_i, [_vp]),
    "gs_topk_block_bytes": (_u64, [_u64, _u32]),
    "gs_topk_pack": (_i, [_vp, _vp, _u64, _u64, _u32, _vp]),
    "gs_topk_unpack": (_i, [_vp, _i, _u64, _u32, _vp, _vp, _vp]),
    "gs_topk_merge_dev": (_i, [_vp, _vp, _vp, _u32, _u64, _u32, _vp, _u32, _vp, _vp]),
    "gs_synth_dna_dev": (_i, [_vp, _u64, _u64, _u64, _u64, _vp]),
    "gs_synth_aa_dev": (_i, [_vp, _u64, _u64, _u64, _u64, _vp]),
    "gs_synth_dna_family_dev": (_i, [_vp, _u64, _u64, _u64, _u64, _u64, C.c_double, C.c_double, _vp]),
    "gs_synth_sigs_dev": (_i, [_vp, _i, _u32, _u64, _u64, _u64, _u64, C.c_double, C.c_double, _vp]),
    "gs_synth_sigs_skew_dev": (_i, [_vp, _i, _u32, _u64, _u64, _u64, _u64, C.c_double, C.c_double, C.c_double, _vp]),
    "gs_synth_dna_family_skew_dev": (_i, [_vp, _u64, _u64, _u64, _u64, _u64, C.c_double, C.c_double, C.c_double, _vp]),
}

_lib = None


def load():
    """Load the HIP shared library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("gsearch_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C gsearch_amd/csrc`. There is no CPU fallback." % SO_PATH)
        L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)          # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != GS_OK:
        raise GsError(rc, load().gs_last_error().decode("utf-8", "replace"))
    return rc
