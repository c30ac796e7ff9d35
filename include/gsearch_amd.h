/*
 * gsearch_amd.h — C ABI of the MI355X-native sketch-and-query hot path of gsearch.
 *
 * The reference (Rust, /root/reference) has no FFI of its own for this path: the extension points are
 * generic traits resolved at compile time. Each entry point below replaces one *batch-level* call the
 * reference makes into those traits (paths relative to /root/reference); INTEGRATION.md shows the
 * `extern "C"` block a Rust maintainer would add to bind them.
 *
 * Conventions: every function returns 0 on success and a negative GS_ERR_* code on failure; nothing
 * throws or aborts across the boundary; gs_last_error() gives a thread-local message. All pointers
 * are HOST pointers unless the name ends in `_dev`. Plain C types only.
 * Arithmetic is normative in SPEC.md.
 */
#ifndef GSEARCH_AMD_H
#define GSEARCH_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------- */
enum {
    GS_OK = 0,
    GS_ERR_INVALID = -1,      /* bad argument / unsupported parameter combination */
    GS_ERR_HIP = -2,          /* HIP runtime error (no device, OOM, launch failure) */
    GS_ERR_UNSUPPORTED = -3,  /* valid in the reference but not implemented on the device yet */
    GS_ERR_STATE = -4,        /* object used in the wrong state (e.g. search on an empty index) */
    GS_ERR_IO = -5
};
/* kmerutils::sketcharg::SketchAlgo / DataType as parsed at src/bin/gsearch.rs:181-196,258-263 */
enum { GS_ALGO_PROB3A = 0, GS_ALGO_SUPER = 1, GS_ALGO_SUPER2 = 2, GS_ALGO_HLL = 3, GS_ALGO_OPTDENS = 4, GS_ALGO_REVOPTDENS = 5 };
/* GS_DATA_DNA_FWD: DNA whose k-mer value is the forward window itself, WITHOUT the reverse-complement minimum - the closure bindash-rs passes
 * for k <= 14 (`kmer.get_compressed_value() & mask`, src/bin/bindash.rs:346-354; its k = 16 and k > 16 closures, :366-377,:388-397, and every
 * closure of gsearch itself, dnasketch.rs:164-169, are canonical = GS_DATA_DNA). Accepted for every k and algo; same (k -> Kmer::Val, Sig) table as DNA. */
enum { GS_DATA_DNA = 0, GS_DATA_AA = 1, GS_DATA_DNA_FWD = 2 };
/* signature element type (table SURVEY 2.2; src/dna/dnasketch.rs:499-642, src/aa/aasketch.rs:455-550) */
enum { GS_KIND_U16 = 0, GS_KIND_U32 = 1, GS_KIND_U64 = 2, GS_KIND_F32 = 3 };

const char *gs_last_error(void);
const char *gs_version(void);

/* ---------------------------------------------------------------------------------------------- */
/* Context: one per (process, GPU). Owns a HIP stream; every call on a context is enqueued on it.   */
/* Every entry point that takes a context (or an index made on one) may be called concurrently from */
/* several host threads. The synchronous host-pointer calls gs_sketch_batch, gs_hamming_qxc and      */
/* gs_hamming_pairs run side by side: every calling thread but the first gets a worker stream +      */
/* scratch of its own on the same device (GS_THREAD_CONTEXTS=0: off). Everything else - the `_dev`   */
/* calls, whose ordering on the context's stream the caller relies on, and the calls on an index -   */
/* queues on the context's one stream and scratch pool behind its lock.                              */
/* `stream` may be NULL (the context creates its own) or an existing hipStream_t to adopt.          */
typedef struct gs_ctx gs_ctx;
int   gs_ctx_create(gs_ctx **out, int device_id, void *stream);
void  gs_ctx_destroy(gs_ctx *);
int   gs_ctx_sync(gs_ctx *);
int   gs_ctx_release_scratch(gs_ctx *);   /* free the device scratch the context keeps between calls (it grows on demand) */
void *gs_ctx_stream(gs_ctx *);                         /* the hipStream_t kernels are launched on */
int   gs_ctx_device_info(gs_ctx *, int *n_cu, uint64_t *hbm_bytes, char *name, size_t name_cap);
/* HIP-event stopwatch on the context's stream (bench.py measures kernels with it) */
int   gs_ctx_timer_start(gs_ctx *);
int   gs_ctx_timer_stop(gs_ctx *, float *elapsed_ms);
/* duration (ms) and launch count of the kernels of one family since the last reset, measured with
 * HIP events around every launch when profiling is enabled (gs_ctx_profile(ctx,1)).
 * family: 0 = sketch main kernel, 1 = hamming q x c, 2 = index search kernel, 3 = index insert kernels */
int   gs_ctx_profile(gs_ctx *, int enable);
int   gs_ctx_profile_read(gs_ctx *, int family, double *total_ms, uint64_t *launches, int reset);

/* which form of the slot-min sketch kernel (optdens / revoptdens / super / super2 level 0) the LAST sketch call on this context launched:
 * out[0] = 1 when the early-rejection ("filtered") emitter ran, out[1] = 1 when the slot table lived in LDS, out[2] = workgroups per
 * genome, out[3] = launches. Tests use it to prove that a parity case exercised the instantiation the bench times. */
int   gs_ctx_last_sketch_info(gs_ctx *, uint32_t out[4]);

/* plain device-memory helpers so that hosts without a HIP binding can keep data resident in HBM */
int   gs_dev_alloc(gs_ctx *, size_t bytes, void **dev_ptr);
int   gs_dev_free(gs_ctx *, void *dev_ptr);
int   gs_dev_upload(gs_ctx *, void *dst_dev, const void *src_host, size_t bytes);
int   gs_dev_download(gs_ctx *, void *dst_host, const void *src_dev, size_t bytes);
int   gs_dev_memset(gs_ctx *, void *dst_dev, int byte, size_t bytes);

/* ---------------------------------------------------------------------------------------------- */
/* Sketching: replaces SeqSketcherT::sketch_compressedkmer_seqs / sketch_compressedkmer            */
/*   call sites: src/dna/dnasketch.rs:336,357  src/dna/dnarequest.rs:272,287                        */
/*               src/aa/aasketch.rs:313,329    src/aa/aarequest.rs:268,283  src/bin/bindash.rs:81   */
/* mirrors kmerutils::sketcharg::SeqSketcherParams{kmer_size, sketch_size, algo, data_t}            */
typedef struct { uint32_t k, sketch_size, algo, data_t; } gs_sketch_params;

int    gs_check_params(const gs_sketch_params *);       /* k=15, k>32 (DNA) / k>12 (AA) -> error */
int    gs_sig_kind(const gs_sketch_params *);           /* GS_KIND_* */
size_t gs_sig_elem_bytes(const gs_sketch_params *);
int    gs_value_bits(const gs_sketch_params *);         /* width of Kmer::Val: 32 or 64 */

/*
 * One signature per genome, in input order (asserts at dnasketch.rs:338,359).
 *   seq        DNA: 2-bit packed, base i in byte i>>2 at bits [6-2(i&3), 7-2(i&3)] (SPEC 1.1);
 *              AA : one ASCII letter per residue, already alphabet-filtered (aafiles.rs:11-28).
 *   seq_bytes  size of seq; for the _dev variant the allocation must extend to a multiple of 8 bytes.
 *   record r   = bases/residues [rec_start[r], rec_start[r]+rec_len[r]); k-mers never span records.
 *   genome g   = records [genome_rec_off[g], genome_rec_off[g+1]).  `--block` mode = one record/genome.
 *   sig_out    n_genomes x sketch_size elements of gs_sig_kind(), caller owned.
 * Thread-safe: callable concurrently from many host threads on ONE context, like the reference's &self sketcher cloned into
 * --nbthreads workers (dnasketch.rs:252,305,322): gs_sketch_batch calls of different threads overlap on the device (worker streams; 16 threads
 * x 4 genomes of 2 Mbp per call: 34.7 k genomes/s against 19.2 k queued, profiles/r04_thread_contexts.log); gs_sketch_batch_dev queues on the context's stream.
 */
int gs_sketch_batch(gs_ctx *, const gs_sketch_params *, const void *seq, uint64_t seq_bytes,
                    const uint64_t *rec_start, const uint64_t *rec_len, uint64_t n_rec,
                    const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out);
/* same, every pointer in device memory, asynchronous on the context's stream */
int gs_sketch_batch_dev(gs_ctx *, const gs_sketch_params *, const void *seq_dev, uint64_t seq_bytes,
                        const uint64_t *rec_start_dev, const uint64_t *rec_len_dev, uint64_t n_rec,
                        const uint64_t *genome_rec_off_dev, uint64_t n_genomes, void *sig_out_dev);
/* ---- FASTA ingest: the reader side of src/dna/dnafiles.rs:43-193 for already-decompressed text ---- */
/* host: record boundaries. Record r = sequence text bytes [seq_begin[r], seq_end[r]) (newlines included) and the header's first
 * word [id_begin[r], +id_len[r]). Records whose header LINE (needletail id(): description included) contains "capsid" are skipped
 * when skip_capsid != 0 (dnafiles.rs:62-67).
 * Arrays may be NULL / cap 0 to count only; *n_rec_out = number of records kept. */
int gs_fasta_scan(const char *buf, uint64_t n, int skip_capsid, uint64_t cap, uint64_t *seq_begin, uint64_t *seq_end,
                  uint64_t *id_begin, uint32_t *id_len, uint64_t *n_rec_out);
/* device: filter + case-fold + 2-bit pack the text of n_rec records (Sequence::encode_and_add, dnafiles.rs:70-71,148-149; every
 * non-ACGT byte, newlines included, is dropped). text_dev: raw text; seq_begin/seq_end: HOST offsets from gs_fasta_scan;
 * packed_dev: ZEROED device buffer >= n_bytes/4 + 8*n_rec + 64 bytes; rec_start_out/rec_len_out: HOST, ready for gs_sketch_batch_dev. */
int gs_pack_fasta_dev(gs_ctx *, const void *text_dev, uint64_t n_bytes, const uint64_t *seq_begin, const uint64_t *seq_end,
                      uint64_t n_rec, void *packed_dev, uint64_t *rec_start_out, uint64_t *rec_len_out);
/* amino acids: drop everything outside the 20-letter alphabet (filter_out_non_aa, src/aa/aafiles.rs:11-28; newlines of the raw text go
 * with it) from the text of n_rec records. out_dev: >= n_bytes bytes; rec_start_out / rec_len_out: HOST, residue coordinates. */
int gs_filter_aa_dev(gs_ctx *, const void *text_dev, uint64_t n_bytes, const uint64_t *seq_begin, const uint64_t *seq_end,
                     uint64_t n_rec, void *out_dev, uint64_t *rec_start_out, uint64_t *rec_len_out);
/* ---- files: the reader side of sketchandstore_dir_compressedkmer, src/dna/dnasketch.rs:240-300 ---- */
/* src/utils/files.rs:117-146: 1 when the name carries a FASTA suffix gsearch accepts for data_t (fna fa fasta / faa, also .gz .xz .bz2) */
int  gs_is_fasta_file(const char *path, int data_t);
/* whole file in memory, gzip (multi-member) / bzip2 / xz decompressed by magic bytes like needletail (files.rs:220-250); release the
 * malloc-ed *text_out with gs_host_free */
int  gs_read_fasta_file(const char *path, void **text_out, uint64_t *n_out);
void gs_host_free(void *);
/* files.rs:148-215,345-455: accepted files under dir, recursively, name order. paths_buf NULL to size (*n_out files, *bytes_out bytes),
 * then a buffer that receives the NUL-terminated paths back to back */
int  gs_list_fasta_files(const char *dir, int data_t, char *paths_buf, uint64_t cap_bytes, uint64_t *n_out, uint64_t *bytes_out);
/* One signature per file, input order. Groups of `pio` files (--pio, files.rs:258-341; 0 -> 32) are read + decompressed + scanned by
 * n_threads host threads (0 -> the CPUs the process may use: affinity mask and cgroup quota) while the previous group crosses PCIe from pinned memory on a copy stream and the one before is
 * filtered / 2-bit packed / sketched on the context's stream. block_mode 0: k-mers never span records (process_file_by_sequence,
 * dnafiles.rs:43-107); 1: --block, records concatenated (process_file_in_one_block, dnafiles.rs:200-262); `capsid` records skipped.
 * sig_out: HOST n_files x sketch_size. Optional per-file n_records_out / n_symbols_out (HOST) and stats_out[4] = {host read+decode+scan
 * seconds summed over threads, seconds waited for PCIe, seconds in device pack + sketch, wall seconds}. */
int  gs_sketch_files(gs_ctx *, const gs_sketch_params *, const char *const *paths, uint64_t n_files, int block_mode, uint32_t pio,
                     uint32_t n_threads, void *sig_out, uint64_t *n_records_out, uint64_t *n_symbols_out, double *stats_out);
/* the same call with a SIZED statistics array: the first min(stats_cap, GS_SKETCH_FILES_STATS) entries of {the four above, .gz members inflated by
 * the device kernel, members the device path handed back to the host decoders (multi-member files, a trailer / CRC-32 that does not check, no room)}
 * are written - a later library may know more entries, a caller never receives more than it made room for. */
enum { GS_SKETCH_FILES_STATS = 6 };
int  gs_sketch_files_ex(gs_ctx *, const gs_sketch_params *, const char *const *paths, uint64_t n_files, int block_mode, uint32_t pio,
                        uint32_t n_threads, void *sig_out, uint64_t *n_records_out, uint64_t *n_symbols_out, double *stats_out, uint32_t stats_cap);
/* gzip members inflated ON the device (gs_inflate.hip; the .gz path of gs_sketch_files, exposed for parity tests against zlib - the
 * reference reads .gz through needletail's flate2 reader, files.rs:258-341). in[i]/in_len[i]: HOST bytes of one single-member gzip file;
 * out[i]/out_cap[i]: HOST buffers for the text; out_len[i]: bytes produced; status[i]: 0 = ok (deflate data, ISIZE and CRC-32 all
 * check), 1..8 = malformed deflate data, 100 = header not taken (not gzip, reserved flags), 101 = bytes after the member (multi-member
 * file: host path), 102 = ISIZE mismatch, 103 = CRC mismatch, 104 = out_cap below the member's ISIZE. */
int  gs_gunzip_batch(gs_ctx *, const uint8_t *const *in, const uint64_t *in_len, uint64_t n, uint8_t *const *out, const uint64_t *out_cap,
                     uint64_t *out_len, int *status);
/* ASCII helpers for hosts that do not pack themselves (Sequence::encode_and_add, dnafiles.rs:70-71) */
uint64_t gs_pack_dna(const uint8_t *ascii, uint64_t n, uint8_t *packed_zeroed, uint64_t base_off);
uint64_t gs_filter_aa(const uint8_t *ascii, uint64_t n, uint8_t *out);

/* ---------------------------------------------------------------------------------------------- */
/* DistHamming::eval, batched (anndists; bound at dnasketch.rs:72,139; direct use bindash.rs:93-99) */
/* dist = (f32)count(a[i] != b[i]) / (f32)m                                                         */
int gs_hamming_qxc(gs_ctx *, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc,
                   float *dist_out /* nq x nc */);
int gs_hamming_qxc_dev(gs_ctx *, int kind, uint32_t m, const void *Q_dev, uint64_t nq, const void *C_dev,
                       uint64_t nc, float *dist_out_dev);
int gs_hamming_pairs(gs_ctx *, int kind, uint32_t m, const void *A, uint64_t na, const void *B, uint64_t nb,
                     const uint64_t *ia, const uint64_t *ib, uint64_t npairs, float *dist_out);
/* reformat.rs:80-86 calculate_ani (model 1 Poisson, 2 binomial), host arithmetic in f64 */
double gs_ani(double distance, int kmer_size, int model);

/* ---------------------------------------------------------------------------------------------- */
/* Hnsw<Sig, DistHamming> (hnsw_rs) as gsearch drives it:                                           */
/*   new/modify_level_scale/set_extend_candidates/set_keeping_pruned  dnasketch.rs:139-141,159-160  */
/*   parallel_insert dnasketch.rs:435, aasketch.rs:407;  parallel_search dnarequest.rs:353, aarequest.rs:344 */
typedef struct gs_index gs_index;
typedef struct {
    int      kind;               /* GS_KIND_* of Sig */
    uint32_t m;                  /* signature length */
    uint32_t max_nb_conn;        /* M <= 255 (gsearch.rs:268) */
    uint64_t capacity;           /* hnsw_params.capacity, 1_500_000 in gsearch (gsearch.rs:269); rows are pre-allocated */
    uint32_t max_layer;          /* 16 (dnasketch.rs:139) */
    uint32_t ef_construction;
    double   scale_modify;       /* modify_level_scale factor (dnasketch.rs:141) */
    int      extend_candidates;  /* true in gsearch (dnasketch.rs:159) */
    int      keep_pruned;        /* false in gsearch (dnasketch.rs:160) */
    uint64_t seed;               /* level generator seed (SPEC 5; upstream uses OS entropy) */
    uint32_t insert_batch;       /* B of SPEC 5 parallel_insert; 0 -> default */
} gs_index_params;

int      gs_index_create(gs_ctx *, const gs_index_params *, gs_index **out);
void     gs_index_destroy(gs_index *);
uint64_t gs_index_nb_point(const gs_index *);
int      gs_index_get_params(const gs_index *, gs_index_params *out);
/* parallel_insert(&[(&Vec<Sig>, usize)]) for ids that continue nb_point.. in input order, gsearch's own case (dnasketch.rs:429-433) */
int      gs_index_parallel_insert(gs_index *, const void *sigs, uint64_t n);
int      gs_index_parallel_insert_dev(gs_index *, const void *sigs_dev, uint64_t n);
/* parallel_insert(&[(&Vec<Sig>, usize)]) with the caller's DataIds (HOST array in both forms; hnsw_rs takes any usize, dnasketch.rs:426-435): searches
 * return them as d_id, both dump formats store them, and gs_index_set_ids / gs_index_get_ids carry them across gs_index_import / _export.
 * Inside the library nodes are numbered in insertion order; that number breaks distance ties in answers ((distance, insertion order) ascending)
 * and is the id whenever no ids were given. */
int      gs_index_parallel_insert_ids(gs_index *, const void *sigs, const uint64_t *ids, uint64_t n);
int      gs_index_parallel_insert_ids_dev(gs_index *, const void *sigs_dev, const uint64_t *ids, uint64_t n);
int      gs_index_set_ids(gs_index *, const uint64_t *ids, uint64_t n /* == nb_point */);
int      gs_index_get_ids(gs_index *, uint64_t first, uint64_t n, uint64_t *ids_out);
/* parallel_search(&[Vec<Sig>], knbn, ef) -> per query min(knbn, found) Neighbour{d_id, distance}, ascending by (distance, node number): the
 * node number is the insertion order, which IS d_id unless the caller gave its own ids (gs_index_parallel_insert_ids / gs_index_set_ids) - ties
 * are then still broken by insertion order, not by the caller's id - or the index came from gs_index_load_hnswrs of a dump with ids other than
 * 0..n-1, whose nodes are renumbered in data-file order (layer-major). Unused tail slots: id = UINT64_MAX, distance = +inf.
 * evals_out (optional): number of DistHamming evaluations spent per query.
 * ef: gsearch asks for 5000 (gsearch.rs:893). Up to ~6700 (at max_nb_conn <= 128) either traversal serves; beyond that, up to 65535, the call takes the
 * dense strategy (count matrix + look-up traversal) whatever the cost model says, and is GS_ERR_UNSUPPORTED where that is not possible
 * (m > 65535, GS_DIST_MODE=gather). */
int      gs_index_parallel_search(gs_index *, const void *queries, uint64_t nq, uint32_t knbn, uint32_t ef,
                                  uint64_t *ids_out, float *dist_out, uint32_t *count_out, uint64_t *evals_out);
int      gs_index_parallel_search_dev(gs_index *, const void *queries_dev, uint64_t nq, uint32_t knbn, uint32_t ef,
                                      uint64_t *ids_out_dev, float *dist_out_dev, uint32_t *count_out_dev,
                                      uint64_t *evals_out_dev);
/* the same search, also returning hnsw_rs' PointId of every neighbour (Neighbour.p_id, answer.rs:42): pid_layer_out[nq x knbn] = the layer the
 * point is filed under (its level), pid_rank_out[nq x knbn] = its rank among the points of that layer in insertion order; 0xFF / -1 in unused slots */
int      gs_index_parallel_search_pid(gs_index *, const void *queries, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids_out, float *dist_out,
                                      uint32_t *count_out, uint64_t *evals_out, uint8_t *pid_layer_out, int32_t *pid_rank_out);
int      gs_index_parallel_search_pid_dev(gs_index *, const void *queries_dev, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids_out_dev,
                                          float *dist_out_dev, uint32_t *count_out_dev, uint64_t *evals_out_dev, uint8_t *pid_layer_out_dev,
                                          int32_t *pid_rank_out_dev);
/* sketch_and_request (sketch the request genomes, then ONE parallel_search: sketch_and_request_dir_compressedkmer, dnarequest.rs:240-360) as one call on
 * device-resident genomes (layout of gs_sketch_batch_dev). Same answers as gs_sketch_batch_dev + gs_index_parallel_search_dev; with the dense strategy the
 * sketch of the next <= 3276 genomes runs on a second stream beside the count matrix of the previous ones. sig_out_dev (optional): the n_genomes signatures. */
int      gs_index_sketch_and_search_dev(gs_index *, const gs_sketch_params *, const void *seq_dev, uint64_t seq_bytes, const uint64_t *rec_start_dev,
                                        const uint64_t *rec_len_dev, uint64_t n_rec, const uint64_t *genome_rec_off_dev, uint64_t n_genomes, void *sig_out_dev,
                                        uint32_t knbn, uint32_t ef, uint64_t *ids_out_dev, float *dist_out_dev, uint32_t *count_out_dev, uint64_t *evals_out_dev);
/* exact top-k by exhaustive DistHamming (recall ground truth; also what bindash.rs:120-157 computes) */
int      gs_index_bruteforce_search(gs_index *, const void *queries, uint64_t nq, uint32_t knbn,
                                    uint64_t *ids_out, float *dist_out);
/* the dense producer on its own: DistHamming of every query against EVERY node as 16-bit mismatch counts (count / m = the distance), the
 * matrix the dense traversal looks its distances up in - match-join (with heavy blocks through the compare tile kernel) or compare tile kernel
 * as the search would choose. counts_out: HOST nq x nb_point, row-major. Needs m <= 65535. (bindash.rs:120-157 computes the same all-pairs) */
int      gs_index_count_matrix(gs_index *, const void *queries, uint64_t nq, uint16_t *counts_out);
/* Graph import / export (the role of hnswio::HnswIo::load_hnsw / Hnsw::file_dump, reloadhnsw.rs:41-51,
 * dumpload.rs:31, in this library's own dense layout): levels[n], entry id, layer 0: deg0[n], nbr0[n*2M],
 * cnt0[n*2M] (mismatch counts to the owner); upper layers: upidx[n] (-1 for level-0 nodes) and for the
 * n_upper nodes of level >= 1: degU[U*max_layer], nbrU[U*max_layer*M], cntU[...] (row l-1 = layer l). */
int      gs_index_import(gs_index *, const void *sigs, uint64_t n, const uint8_t *levels, int64_t entry,
                         const uint32_t *deg0, const uint32_t *nbr0, const uint32_t *cnt0, const int32_t *upidx,
                         uint64_t n_upper, const uint32_t *degU, const uint32_t *nbrU, const uint32_t *cntU);
int      gs_index_export(gs_index *, uint8_t *levels, int64_t *entry, uint32_t *deg0, uint32_t *nbr0,
                         uint32_t *cnt0, int32_t *upidx, uint64_t *n_upper, uint32_t *degU, uint32_t *nbrU,
                         uint32_t *cntU);
int      gs_index_get_data(gs_index *, uint64_t first, uint64_t n, void *sigs_out);
int      gs_index_save(gs_index *, const char *path);
int      gs_index_load(gs_ctx *, const char *path, gs_index **out);
/* hnsw_rs' own dump (Hnsw::file_dump / HnswIo::load_hnsw: dumpload.rs:26-31, reloadhnsw.rs:13-51): <basename>.hnsw.graph and
 * <basename>.hnsw.data, format 3. The byte layout is restated from the un-vendored crate as recalled (gs_hnswio.hip header lists every
 * recalled constant). dump needs max_layer = 16 and lists of at most 255 neighbours; load takes capacity / scale_modify / flags / seed /
 * insert_batch for later insertions from `hint` (may be NULL) - the dump itself holds only max_nb_connection, ef and the element type. */
int      gs_index_dump_hnswrs(gs_index *, const char *basename);
/* flags = GS_DUMP_TRUNCATE_255: neighbour counts are ONE byte in this format while layer 0 holds up to 2 * max_nb_conn = 256..510 ids; lists
 * longer than 255 are cut to their 255 closest entries (lossy for those nodes; without the flag such an index is refused - gs_index_save is lossless) */
enum { GS_DUMP_TRUNCATE_255 = 1 };
int      gs_index_dump_hnswrs_ex(gs_index *, const char *basename, uint32_t flags);
int      gs_index_load_hnswrs(gs_ctx *, const char *basename, const gs_index_params *hint, gs_index **out);
uint64_t gs_index_insert_evals(const gs_index *);      /* DistHamming evaluations spent by inserts so far */
/* after the last parallel_insert of a build (the replicas of a multi-GPU request, a server that only answers): gives the insert-time pair cache back - up to 55 % of
 * the device, 90 GB at 300 k nodes - and keeps everything a search needs. Later inserts still work (bit-identical graph): pairs among the nodes inserted before the
 * call are evaluated from their rows instead of looked up. */
int      gs_index_release_build_scratch(gs_index *);
/* device-side work counters of the searches and dense-mode inserts since the last reset (bench.py prices kernels with them):
 * out[0] memory-side atomics sent by the match-join, out[1] candidates popped by the dense traversal, out[2] pops that accepted
 * at least one neighbour, out[3] traversal workgroups in flight (last launch), out[4] bytes of adjacency a pop loads, out[5] / out[6] pops before / after the traversal
 * became order-free, out[7] chance matches on shared table entries the match-join expanded over a cluster's members (heavy blocks) */
int      gs_index_search_stats(gs_index *, uint64_t out[8], int reset);

/* ---------------------------------------------------------------------------------------------- */
/* Multi-GPU: one process per GPU, query batches sharded, DB + graph replicated. The path has ONE exchange step - the
 * all-gather of the per-rank top-k blocks - and this is it, over RCCL / xGMI, for hosts that are not Python (the reference's host is
 * Rust; conceptual ancestor: the per-shard loop of scripts/multiple_search.sh:71-107). Bootstrap like NCCL: one rank calls
 * gs_comm_unique_id and hands the 128 bytes to the others by its own means (file, socket, MPI); every rank then calls gs_comm_create. */
typedef struct gs_comm gs_comm;
int  gs_comm_unique_id(void *id_out_128);
int  gs_comm_create(gs_ctx *, int n_ranks, int rank, const void *id_128, gs_comm **out);
void gs_comm_destroy(gs_comm *);
int  gs_comm_rank(const gs_comm *);
int  gs_comm_size(const gs_comm *);
/* ids_dev / dist_dev: this rank's nq_local x knbn block (device memory); all_*_dev: n_ranks x nq_local x knbn, rank order.
 * One ncclAllGather of the packed block (12 bytes per neighbour). Same nq_local and knbn on every rank. */
int  gs_comm_allgather_topk_dev(gs_comm *, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint32_t knbn,
                                uint64_t *all_ids_dev, float *all_dist_dev);
/* the same exchange for UNEQUAL shards (a contiguous sharding of a batch hands out blocks that differ by one query; a rank may hold none): every rank
 * passes its own nq_local and the same nq_max >= all of them; all_*_dev (room for n_ranks x nq_max rows) receive the COMPACT concatenation in rank order,
 * counts_out (HOST, n_ranks, optional) every rank's count. Still ONE ncclAllGather - of fixed-size blocks, gs_topk_block_bytes() each. */
int  gs_comm_allgatherv_topk_dev(gs_comm *, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint64_t nq_max, uint32_t knbn,
                                 uint64_t *all_ids_dev, float *all_dist_dev, uint64_t *counts_out);
/* the same exchange WITHOUT the host round trip: the pack kernel, the ncclAllGather and the unpack kernel are queued on the context's stream and the call returns;
 * work queued afterwards on that stream (the next step's sketch, gs_topk_merge_dev) sees the gathered answers in order. counts_dev (DEVICE, n_ranks + 1 words,
 * optional): every rank's count, then a word that is non-zero when a rank sent a block of another shape. gs_comm_wait: waits for the stream, fails with
 * GS_ERR_INVALID on such a block, hands the counts of the LAST exchange to the host (counts_out: HOST, n_ranks, optional). */
int  gs_comm_allgatherv_topk_async_dev(gs_comm *, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint64_t nq_max, uint32_t knbn,
                                       uint64_t *all_ids_dev, float *all_dist_dev, uint64_t *counts_dev);
int  gs_comm_wait(gs_comm *, uint64_t *counts_out);
/* the block layout itself, on the HOST (no device needed), for hosts that move the blocks by their own means (MPI, sockets): header {u64 nq_local, u32 knbn,
 * u32 magic}, nq_max x knbn ids, nq_max x knbn distances. unpack: n_ranks blocks back to back -> compact rows in rank order + counts. */
uint64_t gs_topk_block_bytes(uint64_t nq_max, uint32_t knbn);
int  gs_topk_pack(const uint64_t *ids, const float *dist, uint64_t nq_local, uint64_t nq_max, uint32_t knbn, void *block_out);
int  gs_topk_unpack(const void *blocks, int n_ranks, uint64_t nq_max, uint32_t knbn, uint64_t *all_ids, float *all_dist, uint64_t *counts_out);
/* DB-sharded alternative (the per-shard loop + merge of scripts/multiple_search.sh:71-107: the database split over the GPUs, every rank answers ALL queries
 * on its shard, the gathered answers are merged): ids_dev / dist_dev = n_shards x nq x knbn_in (shard-major, as gathered), id_offset (HOST, optional) is
 * added to the ids of each shard; out_*_dev: nq x knbn_out, the best under (distance, id). */
int  gs_topk_merge_dev(gs_ctx *, const uint64_t *ids_dev, const float *dist_dev, uint32_t n_shards, uint64_t nq, uint32_t knbn_in, const uint64_t *id_offset,
                       uint32_t knbn_out, uint64_t *out_ids_dev, float *out_dist_dev);

/* ---------------------------------------------------------------------------------------------- */
/* Synthetic inputs generated in HBM (bench / tests): counter-based, reproducible on the host.      */
/* DNA genome g of length L: packed word w (32 bases, 8 bytes little endian as stored) =             */
/*   splitmix64 finaliser of (seed*0x9e3779b97f4a7c15 + g*0xbf58476d1ce4e5b9 + w)  (see gs_synth.hip) */
int gs_synth_dna_dev(gs_ctx *, uint64_t seed, uint64_t first_genome, uint64_t n_genomes, uint64_t len_bases,
                     void *seq_dev /* n_genomes * ceil(len/32)*8 bytes */);
/* proteome g of length L (residues of the 20-letter alphabet, one byte each; ceil(L/8)*8 bytes per proteome, back to back): residue j of
 * word w = "ACDEFGHIKLMNPQRSTVWY"[byte j of the synth word of (seed ^ 0xAA5EED, g, w) mod 20] */
int gs_synth_aa_dev(gs_ctx *, uint64_t seed, uint64_t first_proteome, uint64_t n_proteomes, uint64_t len_residues, void *seq_dev);
/* related genomes: genome g = root genome number hash(seed,g) mod n_roots with iid substitutions at rate
 * mu(g) ~ U[mu_lo, mu_hi] (Jaccard to the root ~ p/(2-p), p=(1-mu)^k). Same layout as gs_synth_dna_dev. */
int gs_synth_dna_family_dev(gs_ctx *, uint64_t seed, uint64_t first_genome, uint64_t n_genomes, uint64_t len_bases,
                            uint64_t n_roots, double mu_lo, double mu_hi, void *seq_dev);
/* sketch-level database: n_roots random root signatures; row r belongs to root hash(seed,r) mod n_roots
 * and keeps each root slot with probability J(r) ~ U[j_lo, j_hi], else draws its own value. */
int gs_synth_sigs_dev(gs_ctx *, int kind, uint32_t m, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                      uint64_t n_roots, double j_lo, double j_hi, void *sigs_dev /* n_rows x m */);
/* the same two generators with SKEWED family sizes (the regime of NCBI / GTDB prokaryotes, /root/reference/README.md:134: a few species with 10^4 genomes,
 * a long tail of singletons): member g belongs to root floor(n_roots * u(g)^alpha), u uniform in [0,1) - root 0 holds a fraction n_roots^(-1/alpha) of
 * everything (alpha = 3.5, n_roots = 3000: 10 %), the sizes fall off as a power law. alpha = 1 is uniform. */
int gs_synth_dna_family_skew_dev(gs_ctx *, uint64_t seed, uint64_t first_genome, uint64_t n_genomes, uint64_t len_bases,
                                 uint64_t n_roots, double mu_lo, double mu_hi, double alpha, void *seq_dev);
int gs_synth_sigs_skew_dev(gs_ctx *, int kind, uint32_t m, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                           uint64_t n_roots, double j_lo, double j_hi, double alpha, void *sigs_dev /* n_rows x m */);

#ifdef __cplusplus
}
#endif
#endif
