/*
 * gs_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of SPEC.md, i.e. of the algorithms gsearch reaches through the crates
 * kmerutils / probminhash / hnsw_rs / anndists / fxhash (none of which is vendored in
 * /root/reference, none pinned by a Cargo.lock, no Rust toolchain in this image).
 *
 *   >>> PARITY UNPINNED: the reference holds no test, golden vector or fixture for this path and
 *   >>> cannot be built here; this oracle is pinned only by SPEC.md, by algebraic properties that
 *   >>> hold for any correct implementation, and by the README's distance->ANI table.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (gsearch_amd/) never links, imports or calls it.
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* SketchAlgo / DataType as in kmerutils::sketcharg (used at src/bin/gsearch.rs:181-196,258-263) */
enum { GO_ALGO_PROB3A = 0, GO_ALGO_SUPER = 1, GO_ALGO_SUPER2 = 2, GO_ALGO_HLL = 3, GO_ALGO_OPTDENS = 4, GO_ALGO_REVOPTDENS = 5 };
enum { GO_DATA_DNA = 0, GO_DATA_AA = 1, GO_DATA_DNA_FWD = 2 /* forward window, no reverse-complement minimum: bindash.rs:346-354 */ };
enum { GO_KIND_U16 = 0, GO_KIND_U32 = 1, GO_KIND_U64 = 2, GO_KIND_F32 = 3 };

typedef struct { uint32_t k, sketch_size, algo, data_t; } go_params;

int      go_check_params(const go_params *p);          /* 0 ok, <0 invalid (k=15, k>32, ...) */
int      go_sig_kind(const go_params *p);              /* GO_KIND_* per SURVEY 2.2 / SPEC 3 */
size_t   go_sig_elem_bytes(const go_params *p);
int      go_value_bits(const go_params *p);            /* 32 or 64: Kmer::Val width */

/* ---- sequence helpers (SPEC 1) ---- */
/* ASCII -> 2-bit packed at base coordinate base_off (must be pre-zeroed buffer); returns bases kept */
uint64_t go_pack_dna(const uint8_t *ascii, uint64_t n, uint8_t *packed, uint64_t base_off);
/* ASCII -> filtered upper-case AA letters; returns residues kept */
uint64_t go_filter_aa(const uint8_t *ascii, uint64_t n, uint8_t *out);
/* enumerate emitted values of one record (tests); returns count, writes up to cap */
uint64_t go_kmers(const go_params *p, const uint8_t *seq, uint64_t start, uint64_t len, uint64_t *out, uint64_t cap);

/* ---- sketching (SPEC 3) == SeqSketcherT::sketch_compressedkmer_seqs for each genome ---- */
int go_sketch_batch(const go_params *p, const uint8_t *seq,
                    const uint64_t *rec_start, const uint64_t *rec_len,
                    const uint64_t *genome_rec_off, uint64_t n_genomes,
                    void *sig_out, int nthreads);

/* ---- DistHamming (SPEC 4) ---- */
uint32_t go_hamming_count(int kind, uint32_t m, const void *a, const void *b);
float    go_hamming(int kind, uint32_t m, const void *a, const void *b);
void     go_hamming_qxc(int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc,
                        float *out, int nthreads);
void     go_hamming_pairs(int kind, uint32_t m, const void *A, const void *B, const uint64_t *ia,
                          const uint64_t *ib, uint64_t npairs, float *out);
double   go_ani(double dist, int k, int model);        /* reformat.rs:80-86 */

/* ---- HNSW (SPEC 5) ---- */
typedef struct go_index go_index;
go_index *go_index_create(int kind, uint32_t m, uint32_t max_nb_conn, uint32_t ef_construction,
                          uint32_t max_layer, double scale_modify, int extend_candidates,
                          int keep_pruned, uint64_t seed);
void     go_index_destroy(go_index *);
/* parallel_insert: appends n points (ids continue from nb_point); batch = B of SPEC 5 */
int      go_index_insert(go_index *, const void *sigs, uint64_t n, uint32_t batch);
uint64_t go_index_nb_point(const go_index *);
/* parallel_search; ids/dist: nq x knbn, count: nq; evals (optional): distance evaluations per query */
int      go_index_search(const go_index *, const void *queries, uint64_t nq, uint32_t knbn, uint32_t ef,
                         uint64_t *ids_out, float *dist_out, uint32_t *count_out, uint64_t *evals_out,
                         int nthreads);
/* exact brute-force top-k under (count,id) order (recall ground truth) */
int      go_bruteforce_topk(int kind, uint32_t m, const void *db, uint64_t n, const void *queries, uint64_t nq,
                            uint32_t knbn, uint64_t *ids_out, float *dist_out, int nthreads);
/* graph export: level per node, entry point, and dense per-layer adjacency.
 * layer-0: deg0[n], nbr0[n*2M], cnt0[n*2M]; upper: upidx[n] (-1 if level 0) and for U upper nodes
 * degU[U*max_layer], nbrU[U*max_layer*M], cntU[...] (row l-1 = layer l). Pass NULL to query sizes. */
int      go_index_export(const go_index *, uint8_t *levels, int64_t *entry, uint32_t *deg0, uint32_t *nbr0,
                         uint32_t *cnt0, int32_t *upidx, uint64_t *n_upper, uint32_t *degU, uint32_t *nbrU,
                         uint32_t *cntU);
int      go_index_import(go_index *, const void *sigs, uint64_t n, const uint8_t *levels, int64_t entry, const uint32_t *deg0,
                         const uint32_t *nbr0, const uint32_t *cnt0, const int32_t *upidx, const uint32_t *degU,
                         const uint32_t *nbrU, const uint32_t *cntU);
/* search-only view: keeps a pointer to the caller's rows instead of copying them */
int      go_index_import_view(go_index *, const void *sigs, uint64_t n, const uint8_t *levels, int64_t entry, const uint32_t *deg0,
                         const uint32_t *nbr0, const uint32_t *cnt0, const int32_t *upidx, const uint32_t *degU,
                         const uint32_t *nbrU, const uint32_t *cntU);
uint64_t go_index_total_evals(const go_index *);       /* distance evaluations spent in insert so far */

/* ---- test hooks (PRNG restatement vs published reference vectors) ---- */
void     go_test_splitmix(uint64_t seed, uint32_t n, uint64_t *out);
void     go_test_xoshiro(const uint64_t *state4, uint32_t n, uint64_t *out);
void     go_test_seeded(uint64_t seed, uint32_t n, uint64_t *out);
uint64_t go_test_fx(uint64_t v, int hasher_bits, int value_bits);
uint64_t go_test_uint(uint64_t seed, uint64_t n);
double   go_test_u64f(uint64_t seed);
float    go_test_u32f(uint64_t seed);
double   go_test_ln(double x);   /* SPEC 2 LN */

#ifdef __cplusplus
}
#endif
#endif
