/*
 * gs_oracle.c — CPU ORACLE (test infrastructure, NOT product code). See gs_oracle.h.
 *
 *   >>> PARITY UNPINNED (no reference tests / golden vectors / buildable reference) <<<
 *
 * Plain C99 restatement of SPEC.md. Each block cites the reference call site it stands in for
 * (paths relative to /root/reference) and the crate whose published algorithm it restates.
 * Written for clarity, not speed; OpenMP gives the "rayon-like" decomposition of the reference:
 * one task per genome (src/dna/dnasketch.rs:325-366), par_iter over queries
 * (src/dna/dnarequest.rs:353), par_iter over points (src/dna/dnasketch.rs:435).
 */
#include "gs_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* SPEC 2: hashing and RNG                                                                     */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* fxhash 0.2: FxHasher64 over one integer write */
static inline uint64_t fx64(uint64_t v) { return v * 0x517cc1b727220a95ULL; }
/* fxhash 0.2: FxHasher32; a u64 is two u32 writes, low word first */
static inline uint64_t fx32(uint64_t v, int bits)
{
    uint32_t h = 0;
    h = (rotl32(h, 5) ^ (uint32_t)v) * 0x9e3779b9u;
    if (bits == 64) h = (rotl32(h, 5) ^ (uint32_t)(v >> 32)) * 0x9e3779b9u;
    return (uint64_t)h;
}
/* SPEC 2 table: element hash by algorithm (super2 row: src/dna/dnasketch.rs:579-592, src/aa/aasketch.rs:508-517) */
static inline uint64_t elem_hash(uint32_t algo, int vbits, uint64_t v)
{
    switch (algo) {
    case GO_ALGO_PROB3A: return v;
    case GO_ALGO_HLL:    return fx64(v);
    case GO_ALGO_SUPER2: return vbits == 32 ? fx32(v, 32) : fx64(v);
    default:             return fx64(v);
    }
}

typedef struct { uint64_t s[4]; } rng_t;
static inline uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline void rng_seed(rng_t *g, uint64_t seed)
{
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) g->s[i] = splitmix64(&x);
}
static inline uint64_t rng_next64(rng_t *g)
{
    uint64_t *s = g->s;
    uint64_t r = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl64(s[3], 45);
    return r;
}
static inline uint32_t rng_next32(rng_t *g) { return (uint32_t)(rng_next64(g) >> 32); }
/* 23 random mantissa bits; the f32 value is r23 * 2^-23 */
static inline uint32_t rng_r23(rng_t *g) { return rng_next32(g) >> 9; }
static inline double   rng_u64f(rng_t *g) { return (double)(rng_next64(g) >> 12) * 0x1.0p-52; }
static inline uint64_t rng_uint(rng_t *g, uint64_t n)
{
    uint64_t rej = (0 - n) % n;               /* 2^64 mod n */
    uint64_t zone = ~(uint64_t)0 - rej;
    for (;;) {
        unsigned __int128 p = (unsigned __int128)rng_next64(g) * n;
        if ((uint64_t)p <= zone) return (uint64_t)(p >> 64);
    }
}

/* test hooks: expose the PRNG restatement so tests can pin it against the published reference vectors
 * of SplitMix64 / xoshiro256++ / fxhash (tests/test_oracle_kat.py) */
void go_test_splitmix(uint64_t seed, uint32_t n, uint64_t *out) { uint64_t x = seed; for (uint32_t i = 0; i < n; i++) out[i] = splitmix64(&x); }
void go_test_xoshiro(const uint64_t *state4, uint32_t n, uint64_t *out)
{
    rng_t g; memcpy(g.s, state4, 32);
    for (uint32_t i = 0; i < n; i++) out[i] = rng_next64(&g);
}
void go_test_seeded(uint64_t seed, uint32_t n, uint64_t *out) { rng_t g; rng_seed(&g, seed); for (uint32_t i = 0; i < n; i++) out[i] = rng_next64(&g); }
uint64_t go_test_fx(uint64_t v, int hasher_bits, int value_bits) { return hasher_bits == 64 ? fx64(v) : fx32(v, value_bits); }
uint64_t go_test_uint(uint64_t seed, uint64_t n) { rng_t g; rng_seed(&g, seed); return rng_uint(&g, n); }
double go_test_u64f(uint64_t seed) { rng_t g; rng_seed(&g, seed); return rng_u64f(&g); }
float go_test_u32f(uint64_t seed) { rng_t g; rng_seed(&g, seed); return (float)rng_r23(&g) * 0x1.0p-23f; }

/* ------------------------------------------------------------------------------------------ */
/* SPEC 1: parameters, sequences, k-mers                                                       */
/* ------------------------------------------------------------------------------------------ */
int go_check_params(const go_params *p)
{
    if (!p || p->sketch_size < 2) return -1;
    if (p->algo > GO_ALGO_REVOPTDENS) return -2;
    if (p->algo == GO_ALGO_HLL && p->sketch_size > 65535u * 16u) return -1;
    if (p->data_t == GO_DATA_DNA || p->data_t == GO_DATA_DNA_FWD) {
        if (p->k < 1 || p->k > 32 || p->k == 15) return -3;   /* dnarequest.rs:451-454, README.md:676 */
    } else if (p->data_t == GO_DATA_AA) {
        if (p->k < 1 || p->k > 12) return -3;                 /* aasketch.rs:457-466 */
    } else return -4;
    return 0;
}
int go_value_bits(const go_params *p)
{
    if (p->data_t != GO_DATA_AA) return (p->k <= 14 || p->k == 16) ? 32 : 64;  /* dnasketch.rs:499-518 */
    return p->k <= 6 ? 32 : 64;                                                  /* aasketch.rs:455-467 */
}
int go_sig_kind(const go_params *p)
{
    int vb = go_value_bits(p);
    switch (p->algo) {
    case GO_ALGO_PROB3A: return vb == 32 ? GO_KIND_U32 : GO_KIND_U64;   /* dnarequest.rs:419,430,441 */
    case GO_ALGO_SUPER2: return vb == 32 ? GO_KIND_U32 : GO_KIND_U64;   /* dnasketch.rs:579-599 */
    case GO_ALGO_HLL:    return GO_KIND_U16;
    default:             return GO_KIND_F32;                            /* dnasketch.rs:520-540,600-642 */
    }
}
static size_t kind_bytes(int kind) { return kind == GO_KIND_U16 ? 2 : (kind == GO_KIND_U64 ? 8 : 4); }
size_t go_sig_elem_bytes(const go_params *p) { return kind_bytes(go_sig_kind(p)); }

static inline int dna_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}
/* dnafiles.rs:70-71,148-149 -> Sequence::encode_and_add: non-ACGT dropped, case folded */
uint64_t go_pack_dna(const uint8_t *ascii, uint64_t n, uint8_t *packed, uint64_t base_off)
{
    uint64_t w = base_off;
    for (uint64_t i = 0; i < n; i++) {
        int c = dna_code(ascii[i]);
        if (c < 0) continue;
        packed[w >> 2] |= (uint8_t)(c << (6 - 2 * (w & 3)));
        w++;
    }
    return w - base_off;
}
static const char AA_ALPHABET[] = "ACDEFGHIKLMNPQRSTVWY";
static inline int aa_code(uint8_t c)
{
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
    for (int i = 0; i < 20; i++) if (AA_ALPHABET[i] == (char)c) return i;
    return -1;
}
/* aafiles.rs:11-28 filter_out_non_aa */
uint64_t go_filter_aa(const uint8_t *ascii, uint64_t n, uint8_t *out)
{
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (aa_code(ascii[i]) < 0) continue;
        uint8_t c = ascii[i];
        if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
        out[w++] = c;
    }
    return w;
}
static inline int dna_at(const uint8_t *seq, uint64_t i) { return (seq[i >> 2] >> (6 - 2 * (i & 3))) & 3; }

/* generic emitter: calls f(ctx, v) for every emitted value of one record.
 * DNA: kmer_hash_fn closure of dnasketch.rs:164-169; AA: aasketch.rs:156-160 */
typedef void (*emit_fn)(void *ctx, uint64_t v);
static void for_each_kmer(const go_params *p, const uint8_t *seq, uint64_t start, uint64_t len, emit_fn f, void *ctx)
{
    uint32_t k = p->k;
    if (len < k) return;
    if (p->data_t == GO_DATA_DNA_FWD) {
        /* bindash.rs:346-354 (k <= 14): `kmer.get_compressed_value() & mask` - the window as read, no reverse complement */
        uint64_t mask = (k == 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
        uint64_t fwd = 0;
        for (uint64_t i = 0; i < len; i++) {
            fwd = ((fwd << 2) | (uint64_t)dna_at(seq, start + i)) & mask;
            if (i + 1 >= k) f(ctx, fwd);
        }
    } else if (p->data_t == GO_DATA_DNA) {
        uint64_t mask = (k == 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
        uint64_t fwd = 0, rc = 0;
        for (uint64_t i = 0; i < len; i++) {
            uint64_t c = (uint64_t)dna_at(seq, start + i);
            fwd = ((fwd << 2) | c) & mask;
            rc = (rc >> 2) | ((3 - c) << (2 * (k - 1)));
            if (i + 1 >= k) f(ctx, (fwd < rc ? fwd : rc) & mask);
        }
    } else {
        uint64_t mask = ((uint64_t)1 << (5 * k)) - 1;
        uint64_t val = 0;
        for (uint64_t i = 0; i < len; i++) {
            int c = aa_code(seq[start + i]);
            if (c < 0) c = 0; /* contract: input already filtered */
            val = ((val << 5) | (uint64_t)c) & mask;
            if (i + 1 >= k) f(ctx, val);
        }
    }
}
typedef struct { uint64_t *out, n, cap; } collect_t;
static void collect_emit(void *ctx, uint64_t v)
{
    collect_t *c = (collect_t *)ctx;
    if (c->n < c->cap) c->out[c->n] = v;
    c->n++;
}
uint64_t go_kmers(const go_params *p, const uint8_t *seq, uint64_t start, uint64_t len, uint64_t *out, uint64_t cap)
{
    collect_t c = { out, 0, cap };
    for_each_kmer(p, seq, start, len, collect_emit, &c);
    return c.n;
}

/* ------------------------------------------------------------------------------------------ */
/* SPEC 3.1: optdens / revoptdens  (probminhash::densminhash::{OptDensMinHash,RevOptDensMinHash}) */
/* ------------------------------------------------------------------------------------------ */
#define EMPTY32 0xFFFFFFFFu
typedef struct { const go_params *p; int vbits; uint32_t m; uint32_t *slot; } oph_t;
static void oph_emit(void *ctx, uint64_t v)
{
    oph_t *o = (oph_t *)ctx;
    rng_t g;
    rng_seed(&g, elem_hash(o->p->algo, o->vbits, v));
    uint32_t r = rng_r23(&g);
    uint64_t b = rng_uint(&g, o->m);
    if (r < o->slot[b]) o->slot[b] = r;
}
static void oph_finish(oph_t *o, float *sig)
{
    uint32_t m = o->m;
    uint32_t nfilled = 0;
    for (uint32_t b = 0; b < m; b++) nfilled += (o->slot[b] != EMPTY32);
    if (nfilled == 0) { for (uint32_t b = 0; b < m; b++) sig[b] = 1.0f; return; }
    uint32_t *dens = (uint32_t *)malloc(sizeof(uint32_t) * m);
    memcpy(dens, o->slot, sizeof(uint32_t) * m);
    if (nfilled < m) {
        if (o->p->algo == GO_ALGO_OPTDENS) {
            for (uint32_t b = 0; b < m; b++) {
                if (o->slot[b] != EMPTY32) continue;
                rng_t g; rng_seed(&g, (uint64_t)b);
                for (;;) { uint64_t j = rng_uint(&g, m); if (o->slot[j] != EMPTY32) { dens[b] = o->slot[j]; break; } }
            }
        } else {
            uint32_t nempty = m - nfilled;
            for (uint64_t t = 0; nempty > 0; t++) {
                /* winners of this round: smallest proposing j per still-empty target */
                uint32_t *win = (uint32_t *)malloc(sizeof(uint32_t) * m);
                for (uint32_t b = 0; b < m; b++) win[b] = EMPTY32;
                for (uint32_t j = 0; j < m; j++) {
                    if (o->slot[j] == EMPTY32) continue;
                    rng_t g; rng_seed(&g, ((uint64_t)j << 20) + t);
                    uint64_t i = rng_uint(&g, m);
                    if (dens[i] == EMPTY32 && win[i] == EMPTY32) win[i] = j;   /* j ascending -> first = smallest */
                }
                for (uint32_t b = 0; b < m; b++) if (win[b] != EMPTY32) { dens[b] = o->slot[win[b]]; nempty--; }
                free(win);
            }
        }
    }
    for (uint32_t b = 0; b < m; b++) sig[b] = (float)dens[b] * 0x1.0p-23f;
    free(dens);
}

/* ------------------------------------------------------------------------------------------ */
/* SPEC 3.2: super / super2  (probminhash::superminhasher::SuperMinHash, superminhasher2::SuperMinHash2) */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const go_params *p; int vbits; uint32_t m; int wide;  /* wide: r is 64 bit (super2/u64) */
    uint32_t *lvl; uint64_t *rr; uint8_t *filled;
    int64_t *q; uint32_t *perm; uint32_t *hist; uint32_t a; int64_t item;
} smh_t;
static void smh_init(smh_t *s, const go_params *p, int vbits)
{
    uint32_t m = p->sketch_size;
    s->p = p; s->vbits = vbits; s->m = m;
    s->wide = (p->algo == GO_ALGO_SUPER2 && vbits == 64);
    s->lvl = (uint32_t *)malloc(4 * m); s->rr = (uint64_t *)malloc(8 * m); s->filled = (uint8_t *)calloc(m, 1);
    s->q = (int64_t *)malloc(8 * m); s->perm = (uint32_t *)malloc(4 * m); s->hist = (uint32_t *)calloc(m, 4);
    for (uint32_t i = 0; i < m; i++) { s->lvl[i] = m - 1; s->rr[i] = ~(uint64_t)0; s->q[i] = -1; }
    s->hist[m - 1] = m; s->a = m - 1; s->item = 0;
}
static void smh_free(smh_t *s) { free(s->lvl); free(s->rr); free(s->filled); free(s->q); free(s->perm); free(s->hist); }
static void smh_emit(void *ctx, uint64_t v)
{
    smh_t *s = (smh_t *)ctx;
    uint32_t m = s->m;
    rng_t g;
    rng_seed(&g, elem_hash(s->p->algo, s->vbits, v));
    int64_t it = s->item++;
    for (uint32_t j = 0; j <= s->a; j++) {
        uint64_t r;
        if (s->p->algo == GO_ALGO_SUPER) r = rng_r23(&g);
        else r = s->wide ? rng_next64(&g) : (uint64_t)rng_next32(&g);
        uint32_t t = j + (uint32_t)rng_uint(&g, (uint64_t)(m - j));
        if (s->q[j] != it) { s->q[j] = it; s->perm[j] = j; }
        if (s->q[t] != it) { s->q[t] = it; s->perm[t] = t; }
        uint32_t tmp = s->perm[j]; s->perm[j] = s->perm[t]; s->perm[t] = tmp;
        uint32_t sl = s->perm[j];
        int better = !s->filled[sl] || j < s->lvl[sl] || (j == s->lvl[sl] && r < s->rr[sl]);
        if (better) {
            uint32_t jp = s->lvl[sl];
            s->filled[sl] = 1; s->lvl[sl] = j; s->rr[sl] = r;
            if (j < jp) {
                s->hist[jp]--; s->hist[j]++;
                while (s->hist[s->a] == 0) s->a--;
            }
        }
    }
}
static void smh_finish(smh_t *s, void *sig)
{
    for (uint32_t b = 0; b < s->m; b++) {
        if (s->p->algo == GO_ALGO_SUPER) {
            float v = s->filled[b] ? ((float)s->lvl[b] + (float)(uint32_t)s->rr[b] * 0x1.0p-23f) : INFINITY;
            ((float *)sig)[b] = v;
        } else if (s->wide) {
            ((uint64_t *)sig)[b] = s->filled[b] ? s->rr[b] : ~(uint64_t)0;
        } else {
            ((uint32_t *)sig)[b] = s->filled[b] ? (uint32_t)s->rr[b] : 0xFFFFFFFFu;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SPEC 3.3: prob  (probminhash::probminhasher::ProbMinHash3a + ExpRestricted01 + MaxValueTracker) */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t *v; uint64_t n, cap; } vec64_t;
static void vec_emit(void *ctx, uint64_t v)
{
    vec64_t *c = (vec64_t *)ctx;
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 1024; c->v = (uint64_t *)realloc(c->v, 8 * c->cap); }
    c->v[c->n++] = v;
}
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y);
}
typedef struct { double lambda, c1, c2, c3; } texp_t;
static void texp_init(texp_t *t, uint32_t m)
{
    t->lambda = log((double)m / (double)(m - 1));
    t->c1 = expm1(t->lambda) / t->lambda;
    t->c2 = log(2.0 / (1.0 + exp(-t->lambda))) / t->lambda;
    t->c3 = (1.0 - exp(-t->lambda)) / t->lambda;
}
static double em1_spec(double z)
{
    double t = 1.0 + z / 6.0;
    t = 1.0 + (z / 5.0) * t;
    t = 1.0 + (z / 4.0) * t;
    t = 1.0 + (z / 3.0) * t;
    t = 1.0 + (z / 2.0) * t;
    return z * t;
}
static double texp_sample(const texp_t *t, rng_t *g)
{
    double x = t->c1 * rng_u64f(g);
    if (x < 1.0) return x;
    for (;;) {
        x = rng_u64f(g);
        if (x < t->c2) return x;
        double y = 0.5 * rng_u64f(g);
        if (y > 1.0 - x) { x = 1.0 - x; y = 1.0 - y; }
        if (x <= t->c3 * (1.0 - y)) return x;
        if (y * t->c1 <= 1.0 - x) return x;
        if ((y * t->c1) * t->lambda <= em1_spec(t->lambda * (1.0 - x))) return x;
    }
}
/* max tree over m leaves (MaxValueTracker) */
typedef struct { uint32_t m; double *t; } maxtree_t;
static void mt_init(maxtree_t *mt, uint32_t m)
{
    mt->m = m; mt->t = (double *)malloc(sizeof(double) * 2 * m);
    for (uint32_t i = 0; i < 2 * m; i++) mt->t[i] = INFINITY;
}
static void mt_update(maxtree_t *mt, uint32_t b, double h)
{
    uint32_t i = mt->m + b;
    mt->t[i] = h;
    for (i >>= 1; i >= 1; i >>= 1) {
        double l = mt->t[2 * i], r = mt->t[2 * i + 1];
        double mx = l > r ? l : r;
        if (mt->t[i] == mx) break;
        mt->t[i] = mx;
    }
}
static void prob_sketch(const go_params *p, int vbits, vec64_t *vals, void *sig)
{
    uint32_t m = p->sketch_size;
    uint64_t *sg = (uint64_t *)calloc(m, 8);
    if (vals->n) {
        qsort(vals->v, vals->n, 8, cmp_u64);
        texp_t te; texp_init(&te, m);
        maxtree_t mt; mt_init(&mt, m);
        double *q = mt.t + m;
        for (uint64_t i = 0; i < vals->n;) {
            uint64_t j = i; while (j < vals->n && vals->v[j] == vals->v[i]) j++;
            uint64_t v = vals->v[i]; double w = (double)(j - i); i = j;
            double winv = 1.0 / w;
            rng_t g; rng_seed(&g, elem_hash(p->algo, vbits, v));
            for (uint64_t it = 1;; it++) {
                double base = winv * (double)(it - 1);
                if (base > mt.t[1]) break;
                double x = texp_sample(&te, &g);
                double h = base + winv * x;
                uint32_t b = (uint32_t)rng_uint(&g, m);
                if (h < q[b] || (h == q[b] && v < sg[b])) { sg[b] = v; mt_update(&mt, b, h); }
            }
        }
        free(mt.t);
    }
    if (vbits == 32) for (uint32_t b = 0; b < m; b++) ((uint32_t *)sig)[b] = (uint32_t)sg[b];
    else memcpy(sig, sg, 8 * (size_t)m);
    free(sg);
}


/* ------------------------------------------------------------------------------------------ */
/* SPEC 2 LN / TEXP and SPEC 3.4: hll = SetSketch1 with u16 registers                           */
/* (probminhash::setsketcher::SetSketcher behind kmerutils' HyperLogLogSketch<Kmer,u16>;        */
/*  parameters from SetSketchParams::default() + set_m: dnasketch.rs:541-574)                   */
/* ------------------------------------------------------------------------------------------ */
static double spec_ln(double x)
{
    uint64_t bits; memcpy(&bits, &x, 8);
    int64_t e = (int64_t)((bits >> 52) & 0x7FF) - 1023;
    uint64_t mb = (bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double t; memcpy(&t, &mb, 8);
    if (t > 1.4142135623730951) { t = t * 0.5; e += 1; }
    const double s = (t - 1.0) / (t + 1.0), z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0; p = p * z + 1.0 / 19.0; p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0; p = p * z + 1.0 / 9.0; p = p * z + 1.0 / 7.0; p = p * z + 1.0 / 5.0; p = p * z + 1.0 / 3.0; p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}
double go_test_ln(double x) { return spec_ln(x); }
#define HLL_B 1.001
#define HLL_A 20.0
#define HLL_Q 65534u
static inline uint32_t hll_k(double x, double inv_lnb)
{
    if (!(x > 0.0)) return HLL_Q + 1;
    const double y = 1.0 - spec_ln(x) * inv_lnb;
    if (y < 0.0) return 0;
    if (y >= (double)(HLL_Q + 1)) return HLL_Q + 1;
    return (uint32_t)y;
}
typedef struct {
    const go_params *p; uint32_t m; double inv_lnb;
    uint32_t *K; uint32_t klow, nmod;            /* klow: lower bound of the registers (pruning only), refreshed every m modifications */
    int64_t *q; uint32_t *perm; int64_t item;
} hll_t;
static void hll_emit(void *ctx, uint64_t v)
{
    hll_t *h = (hll_t *)ctx;
    const uint32_t m = h->m;
    rng_t g;
    rng_seed(&g, elem_hash(GO_ALGO_HLL, 64, v));
    const int64_t it = h->item++;
    double x = 0.0;
    for (uint32_t j = 0; j < m; j++) {
        const double u = rng_u64f(&g);
        const double te = -spec_ln(1.0 - u);
        const double den = HLL_A * (double)(m - j);
        x = x + te / den;
        const uint32_t k = hll_k(x, h->inv_lnb);
        if (k <= h->klow) break;
        const uint32_t t = j + (uint32_t)rng_uint(&g, (uint64_t)(m - j));
        if (h->q[j] != it) { h->q[j] = it; h->perm[j] = j; }
        if (h->q[t] != it) { h->q[t] = it; h->perm[t] = t; }
        const uint32_t tmp = h->perm[j]; h->perm[j] = h->perm[t]; h->perm[t] = tmp;
        const uint32_t sl = h->perm[j];
        if (k > h->K[sl]) {
            h->K[sl] = k;
            if (++h->nmod >= m) {                 /* the paper's lower-bound refresh */
                uint32_t lo = h->K[0];
                for (uint32_t i = 1; i < m; i++) if (h->K[i] < lo) lo = h->K[i];
                h->klow = lo; h->nmod = 0;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* sketch driver == SeqSketcherT::sketch_compressedkmer_seqs, one call per genome              */
/* (dnasketch.rs:336,357; dnarequest.rs:272,287; aasketch.rs:313,329; aarequest.rs:268,283)    */
/* ------------------------------------------------------------------------------------------ */
static void sketch_one(const go_params *p, const uint8_t *seq, const uint64_t *rs, const uint64_t *rl, uint64_t nrec, void *sig)
{
    int vbits = go_value_bits(p);
    uint32_t m = p->sketch_size;
    if (p->algo == GO_ALGO_OPTDENS || p->algo == GO_ALGO_REVOPTDENS) {
        oph_t o = { p, vbits, m, (uint32_t *)malloc(4 * (size_t)m) };
        for (uint32_t b = 0; b < m; b++) o.slot[b] = EMPTY32;
        for (uint64_t r = 0; r < nrec; r++) for_each_kmer(p, seq, rs[r], rl[r], oph_emit, &o);
        oph_finish(&o, (float *)sig);
        free(o.slot);
    } else if (p->algo == GO_ALGO_SUPER || p->algo == GO_ALGO_SUPER2) {
        smh_t s; smh_init(&s, p, vbits);
        for (uint64_t r = 0; r < nrec; r++) for_each_kmer(p, seq, rs[r], rl[r], smh_emit, &s);
        smh_finish(&s, sig);
        smh_free(&s);
    } else if (p->algo == GO_ALGO_HLL) {
        hll_t h; memset(&h, 0, sizeof h);
        h.p = p; h.m = m; h.inv_lnb = 1.0 / spec_ln(HLL_B);
        h.K = (uint32_t *)calloc(m, 4); h.q = (int64_t *)malloc(8 * (size_t)m); h.perm = (uint32_t *)malloc(4 * (size_t)m);
        for (uint32_t i = 0; i < m; i++) h.q[i] = -1;
        for (uint64_t r = 0; r < nrec; r++) for_each_kmer(p, seq, rs[r], rl[r], hll_emit, &h);
        for (uint32_t i = 0; i < m; i++) ((uint16_t *)sig)[i] = (uint16_t)h.K[i];
        free(h.K); free(h.q); free(h.perm);
    } else {
        vec64_t vals = { 0, 0, 0 };
        for (uint64_t r = 0; r < nrec; r++) for_each_kmer(p, seq, rs[r], rl[r], vec_emit, &vals);
        prob_sketch(p, vbits, &vals, sig);
        free(vals.v);
    }
}
int go_sketch_batch(const go_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len,
                    const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out, int nthreads)
{
    int rc = go_check_params(p);
    if (rc) return rc;
    size_t row = go_sig_elem_bytes(p) * (size_t)p->sketch_size;
    if (nthreads < 1) nthreads = 1;
    int64_t g;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (g = 0; g < (int64_t)n_genomes; g++) {
        uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1];
        sketch_one(p, seq, rec_start + r0, rec_len + r0, r1 - r0, (uint8_t *)sig_out + row * (size_t)g);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* SPEC 4: DistHamming (anndists::dist::DistHamming::eval; bindash.rs:93-99)                    */
/* ------------------------------------------------------------------------------------------ */
uint32_t go_hamming_count(int kind, uint32_t m, const void *a, const void *b)
{
    uint32_t c = 0;
    switch (kind) {
    case GO_KIND_U16: for (uint32_t i = 0; i < m; i++) c += ((const uint16_t *)a)[i] != ((const uint16_t *)b)[i]; break;
    case GO_KIND_U32: for (uint32_t i = 0; i < m; i++) c += ((const uint32_t *)a)[i] != ((const uint32_t *)b)[i]; break;
    case GO_KIND_U64: for (uint32_t i = 0; i < m; i++) c += ((const uint64_t *)a)[i] != ((const uint64_t *)b)[i]; break;
    default:          for (uint32_t i = 0; i < m; i++) c += ((const float *)a)[i] != ((const float *)b)[i]; break;
    }
    return c;
}
float go_hamming(int kind, uint32_t m, const void *a, const void *b)
{
    return (float)go_hamming_count(kind, m, a, b) / (float)m;
}
void go_hamming_qxc(int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out, int nthreads)
{
    size_t row = kind_bytes(kind) * (size_t)m;
    if (nthreads < 1) nthreads = 1;
    int64_t t;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (t = 0; t < (int64_t)(nq * nc); t++) {
        uint64_t i = (uint64_t)t / nc, j = (uint64_t)t % nc;
        out[t] = go_hamming(kind, m, (const uint8_t *)Q + row * i, (const uint8_t *)C + row * j);
    }
}
void go_hamming_pairs(int kind, uint32_t m, const void *A, const void *B, const uint64_t *ia, const uint64_t *ib,
                      uint64_t npairs, float *out)
{
    size_t row = kind_bytes(kind) * (size_t)m;
    for (uint64_t t = 0; t < npairs; t++)
        out[t] = go_hamming(kind, m, (const uint8_t *)A + row * ia[t], (const uint8_t *)B + row * ib[t]);
}
/* reformat.rs:80-86 calculate_ani */
double go_ani(double distance, int k, int model)
{
    double f = (1.0 - distance) * 2.0 / (1.0 - distance + 1.0);
    if (model == 1) return (1.0 + log(f) / (double)k) * 100.0;
    return pow(f, 1.0 / (double)k) * 100.0;
}

/* ------------------------------------------------------------------------------------------ */
/* SPEC 5: HNSW (hnsw_rs::hnsw::Hnsw as driven by dnasketch.rs:139-160,435 / dnarequest.rs:353)  */
/* ------------------------------------------------------------------------------------------ */
typedef uint64_t key_t_;                      /* (count << 32) | id : the total order (c,id) */
#define KEY(c, id) (((uint64_t)(c) << 32) | (uint64_t)(id))
#define KCNT(k) ((uint32_t)((k) >> 32))
#define KID(k) ((uint32_t)(k))

typedef struct { key_t_ *a; uint32_t n, cap; } heap_t;
static void heap_reserve(heap_t *h, uint32_t need)
{
    if (need > h->cap) { h->cap = need * 2 + 16; h->a = (key_t_ *)realloc(h->a, 8 * (size_t)h->cap); }
}
static void minheap_push(heap_t *h, key_t_ k)
{
    heap_reserve(h, h->n + 1);
    uint32_t i = h->n++;
    while (i > 0) { uint32_t p = (i - 1) / 2; if (h->a[p] <= k) break; h->a[i] = h->a[p]; i = p; }
    h->a[i] = k;
}
static key_t_ minheap_pop(heap_t *h)
{
    key_t_ top = h->a[0], k = h->a[--h->n];
    uint32_t i = 0;
    for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && h->a[c + 1] < h->a[c]) c++;
        if (h->a[c] >= k) break;
        h->a[i] = h->a[c]; i = c;
    }
    if (h->n) h->a[i] = k;
    return top;
}
static void maxheap_push(heap_t *h, key_t_ k)
{
    heap_reserve(h, h->n + 1);
    uint32_t i = h->n++;
    while (i > 0) { uint32_t p = (i - 1) / 2; if (h->a[p] >= k) break; h->a[i] = h->a[p]; i = p; }
    h->a[i] = k;
}
static key_t_ maxheap_pop(heap_t *h)
{
    key_t_ top = h->a[0], k = h->a[--h->n];
    uint32_t i = 0;
    for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && h->a[c + 1] > h->a[c]) c++;
        if (h->a[c] <= k) break;
        h->a[i] = h->a[c]; i = c;
    }
    if (h->n) h->a[i] = k;
    return top;
}
static int cmp_key(const void *a, const void *b) { return cmp_u64(a, b); }

typedef struct {
    uint8_t level;
    uint32_t *deg;      /* [level+1] */
    key_t_ **nbr;       /* [level+1][cap(L)] sorted ascending by (count to owner, id) */
} node_t;

struct go_index {
    int kind; uint32_t m; size_t row;
    uint32_t M, efc, max_layer; double scale; int extend, keep_pruned; uint64_t seed;
    uint64_t n, cap; uint8_t *data; node_t *nodes;
    int64_t entry; int top;
    uint64_t evals;
    int borrowed;            /* data points into the caller's rows (go_index_import_view): searched, never grown or freed */
};
static inline uint32_t layer_cap(const go_index *ix, int L) { return L == 0 ? 2 * ix->M : ix->M; }
static inline const void *rowp(const go_index *ix, uint64_t id) { return ix->data + ix->row * id; }

go_index *go_index_create(int kind, uint32_t m, uint32_t max_nb_conn, uint32_t efc, uint32_t max_layer,
                          double scale_modify, int extend, int keep_pruned, uint64_t seed)
{
    go_index *ix = (go_index *)calloc(1, sizeof(*ix));
    ix->kind = kind; ix->m = m; ix->row = kind_bytes(kind) * (size_t)m;
    ix->M = max_nb_conn; ix->efc = efc; ix->max_layer = max_layer;
    ix->scale = scale_modify / log((double)max_nb_conn);       /* dnasketch.rs:141 modify_level_scale */
    ix->extend = extend; ix->keep_pruned = keep_pruned; ix->seed = seed;
    ix->entry = -1; ix->top = -1;
    return ix;
}
void go_index_destroy(go_index *ix)
{
    if (!ix) return;
    for (uint64_t i = 0; i < ix->n; i++) {
        for (int L = 0; L <= ix->nodes[i].level; L++) free(ix->nodes[i].nbr[L]);
        free(ix->nodes[i].nbr); free(ix->nodes[i].deg);
    }
    free(ix->nodes); if (!ix->borrowed) free(ix->data); free(ix);
}
uint64_t go_index_nb_point(const go_index *ix) { return ix->n; }
uint64_t go_index_total_evals(const go_index *ix) { return ix->evals; }

static int gen_level(const go_index *ix, uint64_t id)
{
    rng_t g; rng_seed(&g, ix->seed ^ fx64(id));
    double u = rng_u64f(&g);
    if (u == 0.0) return (int)rng_uint(&g, ix->max_layer);
    int l = (int)floor(-log(u) * ix->scale);
    if (l >= (int)ix->max_layer) l = (int)rng_uint(&g, ix->max_layer);
    return l;
}

typedef struct { uint32_t *stamp; uint32_t epoch; uint64_t cap; heap_t C, R; uint64_t evals; } scratch_t;
static void scratch_prepare(scratch_t *s, uint64_t n)
{
    if (n > s->cap) { s->stamp = (uint32_t *)realloc(s->stamp, 4 * n); memset(s->stamp + s->cap, 0, 4 * (n - s->cap)); s->cap = n; }
    if (++s->epoch == 0) { memset(s->stamp, 0, 4 * s->cap); s->epoch = 1; }
}
static void scratch_free(scratch_t *s) { free(s->stamp); free(s->C.a); free(s->R.a); }

static inline uint32_t dcount(const go_index *ix, scratch_t *s, const void *q, uint32_t id)
{
    s->evals++;
    return go_hamming_count(ix->kind, ix->m, q, rowp(ix, id));
}

/* Malkov alg. 2 as hnsw_rs::search_layer; result sorted ascending into out (<= ef), returns length */
static uint32_t search_layer(const go_index *ix, scratch_t *s, const void *q, uint32_t ep, uint32_t ep_cnt,
                             uint32_t ef, int L, key_t_ *out)
{
    scratch_prepare(s, ix->n);
    s->C.n = 0; s->R.n = 0;
    s->stamp[ep] = s->epoch;
    minheap_push(&s->C, KEY(ep_cnt, ep));
    maxheap_push(&s->R, KEY(ep_cnt, ep));
    while (s->C.n) {
        key_t_ c = minheap_pop(&s->C);
        if (KCNT(c) > KCNT(s->R.a[0])) break;
        const node_t *nd = &ix->nodes[KID(c)];
        for (uint32_t t = 0; t < nd->deg[L]; t++) {
            uint32_t e = KID(nd->nbr[L][t]);
            if (s->stamp[e] == s->epoch) continue;
            s->stamp[e] = s->epoch;
            uint32_t ce = dcount(ix, s, q, e);
            if (ce < KCNT(s->R.a[0]) || s->R.n < ef) {
                minheap_push(&s->C, KEY(ce, e));
                maxheap_push(&s->R, KEY(ce, e));
                if (s->R.n > ef) maxheap_pop(&s->R);
            }
        }
    }
    uint32_t n = s->R.n;
    memcpy(out, s->R.a, 8 * (size_t)n);
    qsort(out, n, 8, cmp_key);
    return n;
}
/* greedy walk on one layer (hnsw_rs::search upper-layer loop): scan all neighbours, move to the best
 * strictly closer one, repeat */
static void greedy_layer(const go_index *ix, scratch_t *s, const void *q, uint32_t *ep, uint32_t *ep_cnt, int L)
{
    int changed = 1;
    while (changed) {
        changed = 0;
        const node_t *nd = &ix->nodes[*ep];
        uint32_t best = *ep, bc = *ep_cnt;
        for (uint32_t t = 0; t < nd->deg[L]; t++) {
            uint32_t e = KID(nd->nbr[L][t]);
            uint32_t ce = dcount(ix, s, q, e);
            if (ce < bc) { bc = ce; best = e; changed = 1; }
        }
        *ep = best; *ep_cnt = bc;
    }
}

/* Malkov alg. 4 as hnsw_rs::select_neighbours. W: candidates sorted ascending (n items). x_row: the new
 * point's data. Returns accepted count, accepted (ascending) in acc. */
static uint32_t select_neighbours(const go_index *ix, scratch_t *s, const void *x_row, int64_t x_id,
                                  const key_t_ *W, uint32_t n, uint32_t deg, int extend, int L, key_t_ *acc)
{
    heap_t cand = { 0, 0, 0 };
    if (n <= deg && !extend) { memcpy(acc, W, 8 * (size_t)n); return n; }
    for (uint32_t i = 0; i < n; i++) minheap_push(&cand, W[i]);
    if (n <= deg && extend) {
        /* neighbours (layer L) of candidates, not themselves candidates (nor x) */
        scratch_prepare(s, ix->n);
        for (uint32_t i = 0; i < n; i++) s->stamp[KID(W[i])] = s->epoch;
        if (x_id >= 0 && (uint64_t)x_id < ix->n) s->stamp[x_id] = s->epoch;
        for (uint32_t i = 0; i < n; i++) {
            if (KID(W[i]) >= ix->n) continue;          /* batch-mate: not linked yet, no neighbours */
            const node_t *nd = &ix->nodes[KID(W[i])];
            if (L > nd->level) continue;
            for (uint32_t t = 0; t < nd->deg[L]; t++) {
                uint32_t e = KID(nd->nbr[L][t]);
                if (s->stamp[e] == s->epoch) continue;
                s->stamp[e] = s->epoch;
                minheap_push(&cand, KEY(dcount(ix, s, x_row, e), e));
            }
        }
    }
    uint32_t na = 0;
    heap_t disc = { 0, 0, 0 };
    while (cand.n && na < deg) {
        key_t_ e = minheap_pop(&cand);
        int ok = 1;
        for (uint32_t j = 0; j < na && ok; j++) {
            uint32_t ces = dcount(ix, s, rowp(ix, KID(e)), KID(acc[j]));
            if (ces <= KCNT(e)) ok = 0;
        }
        if (ok) acc[na++] = e;
        else if (ix->keep_pruned) minheap_push(&disc, e);
    }
    if (ix->keep_pruned) {
        while (disc.n && na < deg) acc[na++] = minheap_pop(&disc);
        qsort(acc, na, 8, cmp_key);
    }
    free(cand.a); free(disc.a);
    return na;
}

static void node_alloc(go_index *ix, node_t *nd, int level)
{
    nd->level = (uint8_t)level;
    nd->deg = (uint32_t *)calloc((size_t)level + 1, 4);
    nd->nbr = (key_t_ **)calloc((size_t)level + 1, sizeof(key_t_ *));
    for (int L = 0; L <= level; L++) nd->nbr[L] = (key_t_ *)malloc(8 * (size_t)(layer_cap(ix, L) + 1));
}
/* sorted insert of (cnt,id) into owner's layer-L list, dedup by id, keep the cap smallest */
static void link_add(go_index *ix, uint32_t owner, int L, uint32_t id, uint32_t cnt)
{
    node_t *nd = &ix->nodes[owner];
    uint32_t cap = layer_cap(ix, L), d = nd->deg[L];
    key_t_ *a = nd->nbr[L];
    for (uint32_t i = 0; i < d; i++) if (KID(a[i]) == id) return;
    key_t_ k = KEY(cnt, id);
    uint32_t pos = d;
    while (pos > 0 && a[pos - 1] > k) { a[pos] = a[pos - 1]; pos--; }
    a[pos] = k; d++;
    if (d > cap) d = cap;
    nd->deg[L] = d;
}

typedef struct { uint32_t n[16]; key_t_ *sel[16]; } plan_t;   /* selected neighbours per layer of one new point */

/* phase 1 of SPEC 5 parallel_insert for point id (level lv) against the graph frozen at n0 points,
 * batch = ids [b0,b1) */
static void plan_point(const go_index *ix, scratch_t *s, uint64_t id, int lv, uint64_t n0, int64_t entry0, int top0,
                       uint64_t b0, uint64_t b1, const uint8_t *levels, plan_t *pl)
{
    const void *x = rowp(ix, id);
    uint32_t efc = ix->efc;
    key_t_ *W = (key_t_ *)malloc(8 * (size_t)(efc + (b1 - b0) + 1));
    uint32_t ep = 0, ep_cnt = 0;
    int have_graph = (n0 > 0);
    /* the searches below must only see the frozen graph: ids < n0. New nodes are appended after
     * phase 1, so ix->nodes[0..n0) is exactly the frozen graph. */
    if (have_graph) {
        ep = (uint32_t)entry0; ep_cnt = dcount(ix, s, x, ep);
        for (int L = top0; L > lv; L--) greedy_layer(ix, s, x, &ep, &ep_cnt, L);
    }
    for (int L = lv; L >= 0; L--) {
        uint32_t nW = 0;
        if (have_graph && L <= top0) {
            nW = search_layer(ix, s, x, ep, ep_cnt, efc, L, W);
            ep = KID(W[0]); ep_cnt = KCNT(W[0]);
        }
        /* batch-mates of sufficient level are candidates too */
        uint32_t nG = nW;
        for (uint64_t q = b0; q < b1; q++) {
            if (q == id || levels[q - b0] < L) continue;
            W[nW++] = KEY(dcount(ix, s, x, (uint32_t)q), (uint32_t)q);
        }
        if (nW > nG) { qsort(W, nW, 8, cmp_key); if (nW > efc) nW = efc; }
        uint32_t deg = layer_cap(ix, L);
        pl->sel[L] = (key_t_ *)malloc(8 * (size_t)(deg + 1));
        pl->n[L] = nW ? select_neighbours(ix, s, x, (int64_t)id, W, nW, deg, L == 0 && ix->extend, L, pl->sel[L]) : 0;
    }
    free(W);
}

int go_index_insert(go_index *ix, const void *sigs, uint64_t n, uint32_t batch)
{
    if (!ix || (!sigs && n) || ix->borrowed) return -1;
    if (batch < 1) batch = 1;
    if (ix->n + n > ix->cap) {
        ix->cap = ix->n + n;
        ix->data = (uint8_t *)realloc(ix->data, ix->row * ix->cap);
        ix->nodes = (node_t *)realloc(ix->nodes, sizeof(node_t) * ix->cap);
    }
    memcpy(ix->data + ix->row * ix->n, sigs, ix->row * n);
    uint64_t first = ix->n, end = ix->n + n;
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    scratch_t *scr = (scratch_t *)calloc((size_t)nth, sizeof(scratch_t));
    for (uint64_t b0 = first; b0 < end; b0 += batch) {
        uint64_t b1 = b0 + batch < end ? b0 + batch : end, nb = b1 - b0;
        uint8_t *levels = (uint8_t *)malloc(nb);
        for (uint64_t i = 0; i < nb; i++) levels[i] = (uint8_t)gen_level(ix, b0 + i);
        plan_t *plans = (plan_t *)calloc(nb, sizeof(plan_t));
        uint64_t n0 = ix->n; int64_t entry0 = ix->entry; int top0 = ix->top;
        int64_t i;
        /* scratch needs stamps for ids up to b1 (batch-mates are marked visited in select) */
        for (int t = 0; t < nth; t++) scratch_prepare(&scr[t], b1);
#pragma omp parallel for schedule(dynamic, 1) if (nb > 1)
        for (i = 0; i < (int64_t)nb; i++) {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            /* ix->n stays n0 during phase 1: scratch sized for b1 above */
            plan_point(ix, &scr[t], b0 + (uint64_t)i, levels[i], n0, entry0, top0, b0, b1, levels, &plans[i]);
        }
        /* phase 2: create nodes, then links (order-free: every list = cap smallest of the union) */
        for (uint64_t i2 = 0; i2 < nb; i2++) node_alloc(ix, &ix->nodes[b0 + i2], levels[i2]);
        ix->n = b1;
        for (uint64_t i2 = 0; i2 < nb; i2++) {
            uint32_t id = (uint32_t)(b0 + i2);
            for (int L = 0; L <= levels[i2]; L++) {
                for (uint32_t j = 0; j < plans[i2].n[L]; j++) {
                    key_t_ e = plans[i2].sel[L][j];
                    link_add(ix, id, L, KID(e), KCNT(e));
                    link_add(ix, KID(e), L, id, KCNT(e));
                }
                free(plans[i2].sel[L]);
            }
            if ((int)levels[i2] > ix->top) { ix->top = levels[i2]; ix->entry = id; }
        }
        free(plans); free(levels);
    }
    for (int t = 0; t < nth; t++) { ix->evals += scr[t].evals; scratch_free(&scr[t]); }
    free(scr);
    return 0;
}

/* hnsw_rs::Hnsw::search (SPEC 5) */
static uint32_t search_one(const go_index *ix, scratch_t *s, const void *q, uint32_t knbn, uint32_t ef,
                           key_t_ *buf, uint64_t *ids, float *dist)
{
    if (ix->n == 0) return 0;
    uint32_t ep = (uint32_t)ix->entry, ep_cnt = dcount(ix, s, q, ep);
    for (int L = ix->top; L >= 1; L--) greedy_layer(ix, s, q, &ep, &ep_cnt, L);
    uint32_t efs = ef > knbn ? ef : knbn;
    uint32_t n = search_layer(ix, s, q, ep, ep_cnt, efs, 0, buf);
    if (n > knbn) n = knbn;
    for (uint32_t i = 0; i < n; i++) { ids[i] = KID(buf[i]); dist[i] = (float)KCNT(buf[i]) / (float)ix->m; }
    return n;
}
int go_index_search(const go_index *ix, const void *queries, uint64_t nq, uint32_t knbn, uint32_t ef,
                    uint64_t *ids_out, float *dist_out, uint32_t *count_out, uint64_t *evals_out, int nthreads)
{
    if (!ix) return -1;
    if (nthreads < 1) nthreads = 1;
    uint32_t efs = ef > knbn ? ef : knbn;
#pragma omp parallel num_threads(nthreads)
    {
        scratch_t s; memset(&s, 0, sizeof(s));
        key_t_ *buf = (key_t_ *)malloc(8 * (size_t)(efs + 1));
        int64_t i;
#pragma omp for schedule(dynamic, 1)
        for (i = 0; i < (int64_t)nq; i++) {
            uint64_t e0 = s.evals;
            for (uint32_t j = 0; j < knbn; j++) { ids_out[(uint64_t)i * knbn + j] = ~(uint64_t)0; dist_out[(uint64_t)i * knbn + j] = INFINITY; }
            uint32_t n = search_one(ix, &s, (const uint8_t *)queries + ix->row * (uint64_t)i, knbn, ef, buf,
                                    ids_out + (uint64_t)i * knbn, dist_out + (uint64_t)i * knbn);
            if (count_out) count_out[i] = n;
            if (evals_out) evals_out[i] = s.evals - e0;
        }
        free(buf); scratch_free(&s);
    }
    return 0;
}

int go_bruteforce_topk(int kind, uint32_t m, const void *db, uint64_t n, const void *queries, uint64_t nq,
                       uint32_t knbn, uint64_t *ids_out, float *dist_out, int nthreads)
{
    size_t row = kind_bytes(kind) * (size_t)m;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        key_t_ *keys = (key_t_ *)malloc(8 * (size_t)(n ? n : 1));
        int64_t i;
#pragma omp for schedule(dynamic, 1)
        for (i = 0; i < (int64_t)nq; i++) {
            const void *q = (const uint8_t *)queries + row * (uint64_t)i;
            for (uint64_t j = 0; j < n; j++) keys[j] = KEY(go_hamming_count(kind, m, q, (const uint8_t *)db + row * j), j);
            qsort(keys, n, 8, cmp_key);
            for (uint32_t j = 0; j < knbn; j++) {
                if (j < n) { ids_out[(uint64_t)i * knbn + j] = KID(keys[j]); dist_out[(uint64_t)i * knbn + j] = (float)KCNT(keys[j]) / (float)m; }
                else { ids_out[(uint64_t)i * knbn + j] = ~(uint64_t)0; dist_out[(uint64_t)i * knbn + j] = INFINITY; }
            }
        }
        free(keys);
    }
    return 0;
}

/* load a graph in the export layout (used to hand a device-built graph to the CPU baseline / parity check) */
static int index_import(go_index *ix, const void *sigs, uint64_t n, const uint8_t *levels, int64_t entry, const uint32_t *deg0,
                        const uint32_t *nbr0, const uint32_t *cnt0, const int32_t *upidx, const uint32_t *degU, const uint32_t *nbrU,
                        const uint32_t *cntU, int view)
{
    if (!ix || ix->n != 0 || n == 0) return -1;
    uint32_t M = ix->M, ML = ix->max_layer;
    ix->cap = n;
    ix->nodes = (node_t *)malloc(sizeof(node_t) * n);
    if (view) { ix->data = (uint8_t *)(uintptr_t)sigs; ix->borrowed = 1; }
    else { ix->data = (uint8_t *)malloc(ix->row * n); memcpy(ix->data, sigs, ix->row * n); }
    ix->top = 0;
    for (uint64_t i = 0; i < n; i++) {
        node_t *nd = &ix->nodes[i];
        node_alloc(ix, nd, levels[i]);
        nd->deg[0] = deg0[i];
        for (uint32_t t = 0; t < deg0[i]; t++) nd->nbr[0][t] = KEY(cnt0[i * 2 * M + t], nbr0[i * 2 * M + t]);
        for (int L = 1; L <= levels[i]; L++) {
            uint64_t u = (uint64_t)upidx[i];
            nd->deg[L] = degU[u * ML + (uint32_t)(L - 1)];
            for (uint32_t t = 0; t < nd->deg[L]; t++)
                nd->nbr[L][t] = KEY(cntU[(u * ML + (uint32_t)(L - 1)) * M + t], nbrU[(u * ML + (uint32_t)(L - 1)) * M + t]);
        }
        if (levels[i] > ix->top) ix->top = levels[i];
    }
    ix->n = n; ix->entry = entry;
    return 0;
}
int go_index_import(go_index *ix, const void *sigs, uint64_t n, const uint8_t *levels, int64_t entry, const uint32_t *deg0,
                    const uint32_t *nbr0, const uint32_t *cnt0, const int32_t *upidx, const uint32_t *degU, const uint32_t *nbrU,
                    const uint32_t *cntU)
{
    return index_import(ix, sigs, n, levels, entry, deg0, nbr0, cnt0, upidx, degU, nbrU, cntU, 0);
}
/* same, but the index keeps a pointer to the caller's rows instead of a copy (a 300 k x 18000 f32 database is 21.6 GB): search only,
 * the rows must outlive the index */
int go_index_import_view(go_index *ix, const void *sigs, uint64_t n, const uint8_t *levels, int64_t entry, const uint32_t *deg0,
                         const uint32_t *nbr0, const uint32_t *cnt0, const int32_t *upidx, const uint32_t *degU, const uint32_t *nbrU,
                         const uint32_t *cntU)
{
    return index_import(ix, sigs, n, levels, entry, deg0, nbr0, cnt0, upidx, degU, nbrU, cntU, 1);
}

int go_index_export(const go_index *ix, uint8_t *levels, int64_t *entry, uint32_t *deg0, uint32_t *nbr0, uint32_t *cnt0,
                    int32_t *upidx, uint64_t *n_upper, uint32_t *degU, uint32_t *nbrU, uint32_t *cntU)
{
    uint64_t U = 0;
    uint32_t M = ix->M, ML = ix->max_layer;
    if (entry) *entry = ix->entry;
    for (uint64_t i = 0; i < ix->n; i++) {
        const node_t *nd = &ix->nodes[i];
        if (levels) levels[i] = nd->level;
        if (deg0) deg0[i] = nd->deg[0];
        for (uint32_t t = 0; t < nd->deg[0]; t++) {
            if (nbr0) nbr0[i * 2 * M + t] = KID(nd->nbr[0][t]);
            if (cnt0) cnt0[i * 2 * M + t] = KCNT(nd->nbr[0][t]);
        }
        if (nd->level > 0) {
            if (upidx) upidx[i] = (int32_t)U;
            for (int L = 1; L <= nd->level; L++) {
                if (degU) degU[U * ML + (uint32_t)(L - 1)] = nd->deg[L];
                for (uint32_t t = 0; t < nd->deg[L]; t++) {
                    if (nbrU) nbrU[(U * ML + (uint32_t)(L - 1)) * M + t] = KID(nd->nbr[L][t]);
                    if (cntU) cntU[(U * ML + (uint32_t)(L - 1)) * M + t] = KCNT(nd->nbr[L][t]);
                }
            }
            U++;
        } else if (upidx) upidx[i] = -1;
    }
    if (n_upper) *n_upper = U;
    return 0;
}
