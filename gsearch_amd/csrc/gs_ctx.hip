// gs_ctx.hip — context, device-memory helpers, parameter tables, host-side helpers and the
// synthetic-input generators of the C ABI (include/gsearch_amd.h).
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include "gs_internal.hpp"
#include "gs_spec.hpp"

namespace gs {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gs

namespace gs {
// ---- worker contexts: life cycle ---------------------------------------------------------------------------------------------
// A worker belongs to (parent context, host thread). It is given back (a) when its thread exits - a thread_local registry whose destructor
// runs at thread exit, checked against the set of live parents so that a parent destroyed first is not touched -, (b) when the table is full:
// the least recently used idle worker is evicted for the newcomer (a thread pool that keeps respawning therefore recycles 64 workers instead
// of leaking them and then silently sharing the parent), and (c) its scratch whenever a worker's call fails on the device: on_worker_failed()
// releases the pools of every idle worker and the caller repeats the call on the parent, where it queues as all calls did before round 4.
static std::mutex &reg_mu() { static std::mutex *m = new std::mutex(); return *m; }                      // leaked on purpose: used from thread-exit destructors
static std::vector<gs_ctx *> &live_parents() { static std::vector<gs_ctx *> *v = new std::vector<gs_ctx *>(); return *v; }
static void destroy_worker(gs_ctx *w) { gs_ctx_destroy(w); }        // (parent stays set: gs_ctx_destroy then leaves the registry alone - its lock may be held here)
struct ThreadWorkers {
    std::vector<gs_ctx *> parents;
    ~ThreadWorkers()
    {
        const std::thread::id me = std::this_thread::get_id();
        std::lock_guard<std::mutex> rk(reg_mu());
        for (gs_ctx *c : parents) {
            bool live = false;
            for (gs_ctx *x : live_parents()) live |= (x == c);
            if (!live) continue;
            gs_ctx *w = nullptr;
            {
                std::lock_guard<std::mutex> lk(c->workers_mu);
                auto it = c->workers.find(me);
                if (it != c->workers.end()) {
                    w = it->second; c->workers.erase(it);
                    if (w->pins > 0) { w->orphan = true; w = nullptr; }      // a snapshot of the table (profile read, failure sweep) still holds it: its unpin destroys it
                }
            }
            if (w) { std::lock_guard<std::recursive_mutex> wl(w->mu); }      // (nobody else can hold it: only this thread used it; taken for the memory order)
            if (w) destroy_worker(w);
        }
    }
};
static thread_local ThreadWorkers t_workers;
void register_parent(gs_ctx *c) { std::lock_guard<std::mutex> rk(reg_mu()); live_parents().push_back(c); }
void unregister_parent(gs_ctx *c)
{
    std::lock_guard<std::mutex> rk(reg_mu());
    auto &v = live_parents();
    for (size_t i = 0; i < v.size(); i++) if (v[i] == c) { v[i] = v.back(); v.pop_back(); break; }
}
gs_ctx *worker_ctx(gs_ctx *c)
{
    if (!c || c->parent) return c;
    static const bool off = getenv("GS_THREAD_CONTEXTS") && !atoi(getenv("GS_THREAD_CONTEXTS"));
    if (off) return c;
    static const size_t cap = getenv("GS_THREAD_CONTEXTS_MAX") ? (size_t)std::max(1, atoi(getenv("GS_THREAD_CONTEXTS_MAX"))) : 64;
    const std::thread::id me = std::this_thread::get_id();
    gs_ctx *evicted = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->workers_mu);
        if (c->owner_thread == std::thread::id{}) c->owner_thread = me;
        if (c->owner_thread == me) return c;
        auto it = c->workers.find(me);
        if (it != c->workers.end()) { it->second->last_use = ++c->use_tick; it->second->pins++; return it->second; }
        if (c->workers.size() >= cap) {
            // full: evict the least recently used worker that is not pinned (nobody between worker_ctx() and worker_unpin(), no snapshot holding it), else share
            // the main context for this call
            auto victim = c->workers.end();
            for (auto jt = c->workers.begin(); jt != c->workers.end(); ++jt)
                if (jt->second->pins == 0 && (victim == c->workers.end() || jt->second->last_use < victim->second->last_use)) victim = jt;
            if (victim == c->workers.end()) return c;
            evicted = victim->second;
            c->workers.erase(victim);
        }
    }
    if (evicted) destroy_worker(evicted);
    gs_ctx *w = nullptr;
    if (gs_ctx_create(&w, c->device, nullptr) != GS_OK) return c;
    unregister_parent(w);                                          // a worker is not a parent
    w->parent = c;
    w->last_sketch[0] = 0;
    w->profile = c->profile;
    {
        std::lock_guard<std::mutex> lk(c->workers_mu);
        w->last_use = ++c->use_tick;
        w->pins = 1;
        c->workers[me] = w;
    }
    bool known = false;
    for (gs_ctx *x : t_workers.parents) known |= (x == c);
    if (!known) t_workers.parents.push_back(c);
    return w;
}
// after a call on worker w of parent c: what gs_ctx_last_sketch_info(parent) reports is the last sketch call of ANY thread
void worker_done(gs_ctx *c, gs_ctx *w)
{
    if (!c || !w || w == c) return;
    { std::lock_guard<std::mutex> lk(c->workers_mu); for (int i = 0; i < 4; i++) c->last_sketch[i] = w->last_sketch[i]; }      // (w is still pinned by this thread)
    worker_unpin(c, w);
}
void worker_unpin(gs_ctx *c, gs_ctx *w)
{
    bool destroy = false;
    { std::lock_guard<std::mutex> lk(c->workers_mu); destroy = --w->pins == 0 && w->orphan; }
    if (destroy) { { std::lock_guard<std::recursive_mutex> wl(w->mu); } destroy_worker(w); }
}
// the workers of c, each pinned: the caller unpins every one of them when it is done
static std::vector<gs_ctx *> pinned_workers(gs_ctx *c)
{
    std::vector<gs_ctx *> ws;
    std::lock_guard<std::mutex> lk(c->workers_mu);
    for (auto &w : c->workers) { w.second->pins++; ws.push_back(w.second); }
    return ws;
}
// a call on a worker failed on the device (typically: its scratch did not fit beside the other workers' pools): give every idle worker's pool back
void on_worker_failed(gs_ctx *c)
{
    (void)hipGetLastError();
    std::vector<gs_ctx *> ws = pinned_workers(c);
    for (gs_ctx *w : ws) {
        if (w->mu.try_lock()) {
            (void)hipSetDevice(w->device);
            (void)hipStreamSynchronize(w->stream);
            delete (ScratchPool *)w->scratch_pool; w->scratch_pool = nullptr;
            w->mu.unlock();
        }
        worker_unpin(c, w);
    }
    (void)hipGetLastError();
}
}  // namespace gs

extern "C" {

const char *gs_last_error(void) { return gs::g_err; }
const char *gs_version(void) { return "gsearch_amd 0.1 (gfx950)"; }

int gs_ctx_create(gs_ctx **out, int device_id, void *stream)
{
    GS_REQUIRE(out, GS_ERR_INVALID, "gs_ctx_create: out is NULL");
    int ndev = 0;
    GS_HIP_CHECK(hipGetDeviceCount(&ndev));
    GS_REQUIRE(ndev > 0 && device_id >= 0 && device_id < ndev, GS_ERR_HIP,
               "gs_ctx_create: device %d not available (%d HIP devices visible)", device_id, ndev);
    GS_HIP_CHECK(hipSetDevice(device_id));
    gs_ctx *c = new gs_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    GS_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    c->n_cu = prop.multiProcessorCount;
    c->hbm_bytes = prop.totalGlobalMem;
    {
        // clockRate / memoryClockRate are kHz, memoryBusWidth bits. The reference is what an MI355X REPORTS (256 CUs, 2 400 000 kHz, 2 000 000 kHz x 8192 bits - its
        // 8 TB/s are four transfers per reported memory clock; only the ratio matters here). Unreported or implausible values leave 1.0.
        const double compute = (double)prop.multiProcessorCount * (double)prop.clockRate * 1e3, ref_compute = 256.0 * 2.4e9;
        const double bw = (double)prop.memoryClockRate * 1e3 * (double)prop.memoryBusWidth / 8.0, ref_bw = 2.0e9 * 1024.0;
        if (compute > 0.05 * ref_compute && compute < 20.0 * ref_compute) c->rel_compute = compute / ref_compute;
        if (bw > 0.05 * ref_bw && bw < 20.0 * ref_bw) c->rel_hbm = bw / ref_bw;
    }
    snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { GS_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    GS_HIP_CHECK(hipEventCreate(&c->t0));
    GS_HIP_CHECK(hipEventCreate(&c->t1));
    gs::register_parent(c);
    *out = c;
    return GS_OK;
}
void gs_ctx_destroy(gs_ctx *c)
{
    if (!c) return;
    if (!c->parent) gs::unregister_parent(c);          // from here on no exiting thread touches this context's worker table
    if (c->child) { gs_ctx_destroy(c->child); c->child = nullptr; }
    for (auto &w : c->workers) gs_ctx_destroy(w.second);
    c->workers.clear();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->prof) for (auto &p : s.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
    delete (gs::ScratchPool *)c->scratch_pool;
    delete (gs::PinnedPool *)c->pinned_pool;
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
int gs_ctx_release_scratch(gs_ctx *c)
{
    GS_REQUIRE(c, GS_ERR_INVALID, "null context");
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    delete (gs::ScratchPool *)c->scratch_pool;
    c->scratch_pool = nullptr;
    delete (gs::PinnedPool *)c->pinned_pool;
    c->pinned_pool = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->workers_mu);
        for (auto &w : c->workers) { const int rc = gs_ctx_release_scratch(w.second); if (rc) return rc; }
    }
    if (c->child) return gs_ctx_release_scratch(c->child);
    return GS_OK;
}
int gs_ctx_sync(gs_ctx *c)
{
    GS_REQUIRE(c, GS_ERR_INVALID, "null context");
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}
void *gs_ctx_stream(gs_ctx *c) { return c ? (void *)c->stream : nullptr; }
int gs_ctx_device_info(gs_ctx *c, int *n_cu, uint64_t *hbm, char *name, size_t cap)
{
    GS_REQUIRE(c, GS_ERR_INVALID, "null context");
    if (n_cu) *n_cu = c->n_cu;
    if (hbm) *hbm = c->hbm_bytes;
    if (name && cap) { strncpy(name, c->name, cap - 1); name[cap - 1] = 0; }
    return GS_OK;
}
int gs_ctx_last_sketch_info(gs_ctx *c, uint32_t out[4])
{
    GS_REQUIRE(c && out, GS_ERR_INVALID, "gs_ctx_last_sketch_info: null argument");
    GS_CTX_LOCK(c);
    std::lock_guard<std::mutex> lk(c->workers_mu);                 // worker threads publish their last call here (gs::worker_done)
    for (int i = 0; i < 4; i++) out[i] = c->last_sketch[i];
    return GS_OK;
}
int gs_ctx_timer_start(gs_ctx *c)
{
    GS_REQUIRE(c, GS_ERR_INVALID, "null context");
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipEventRecord(c->t0, c->stream));
    return GS_OK;
}
int gs_ctx_timer_stop(gs_ctx *c, float *ms)
{
    GS_REQUIRE(c && ms, GS_ERR_INVALID, "null argument");
    GS_HIP_CHECK(hipEventRecord(c->t1, c->stream));
    GS_HIP_CHECK(hipEventSynchronize(c->t1));
    GS_HIP_CHECK(hipEventElapsedTime(ms, c->t0, c->t1));
    return GS_OK;
}
int gs_ctx_profile(gs_ctx *c, int enable)
{
    GS_REQUIRE(c, GS_ERR_INVALID, "null context");
    c->profile = enable != 0;
    std::lock_guard<std::mutex> lk(c->workers_mu);
    for (auto &w : c->workers) w.second->profile = c->profile;      // kernels of worker threads are counted too (gs_ctx_profile_read merges them)
    return GS_OK;
}
int gs_ctx_profile_read(gs_ctx *c, int family, double *total_ms, uint64_t *launches, int reset)
{
    GS_REQUIRE(c && family >= 0 && family < gs::FAM_COUNT, GS_ERR_INVALID, "bad family");
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    gs::ProfSlot &s = c->prof[family];
    for (auto &p : s.pending) {
        float ms = 0;
        GS_HIP_CHECK(hipEventSynchronize(p.second));
        GS_HIP_CHECK(hipEventElapsedTime(&ms, p.first, p.second));
        s.total_ms += ms; s.launches++;
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    s.pending.clear();
    double tot = s.total_ms; uint64_t nl = s.launches;
    if (reset) { s.total_ms = 0; s.launches = 0; }
    if (!c->parent) {                                               // the launches of this context's worker threads belong to the same totals
        std::vector<gs_ctx *> ws = gs::pinned_workers(c);
        int wrc = GS_OK;
        for (gs_ctx *w : ws) {
            double t = 0; uint64_t n = 0;
            if (wrc == GS_OK && (wrc = gs_ctx_profile_read(w, family, &t, &n, reset)) == GS_OK) { tot += t; nl += n; }
            gs::worker_unpin(c, w);
        }
        if (wrc) return wrc;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = nl;
    return GS_OK;
}

int gs_dev_alloc(gs_ctx *c, size_t bytes, void **p)
{
    GS_REQUIRE(c && p, GS_ERR_INVALID, "null argument");
    GS_HIP_CHECK(hipSetDevice(c->device));
    GS_HIP_CHECK(hipMalloc(p, bytes ? bytes : 16));
    return GS_OK;
}
int gs_dev_free(gs_ctx *c, void *p)
{
    GS_REQUIRE(c, GS_ERR_INVALID, "null context");
    if (p) { GS_HIP_CHECK(hipStreamSynchronize(c->stream)); GS_HIP_CHECK(hipFree(p)); }
    return GS_OK;
}
int gs_dev_upload(gs_ctx *c, void *dst, const void *src, size_t bytes)
{
    GS_REQUIRE(c && (bytes == 0 || (dst && src)), GS_ERR_INVALID, "null argument");
    GS_CTX_LOCK(c);
    if (bytes) {
        GS_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return GS_OK;
}
int gs_dev_download(gs_ctx *c, void *dst, const void *src, size_t bytes)
{
    GS_REQUIRE(c && (bytes == 0 || (dst && src)), GS_ERR_INVALID, "null argument");
    GS_CTX_LOCK(c);
    if (bytes) {
        GS_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return GS_OK;
}
int gs_dev_memset(gs_ctx *c, void *dst, int byte, size_t bytes)
{
    GS_REQUIRE(c && (bytes == 0 || dst), GS_ERR_INVALID, "null argument");
    GS_CTX_LOCK(c);
    if (bytes) GS_HIP_CHECK(hipMemsetAsync(dst, byte, bytes, c->stream));
    return GS_OK;
}

/* ---- parameter tables: (algo,k) dispatch of dnasketch.rs:493-644 / aasketch.rs:449-552 ---------- */
int gs_check_params(const gs_sketch_params *p)
{
    GS_REQUIRE(p, GS_ERR_INVALID, "null params");
    GS_REQUIRE(p->sketch_size >= 2, GS_ERR_INVALID, "sketch_size must be >= 2");
    GS_REQUIRE(p->algo <= GS_ALGO_REVOPTDENS, GS_ERR_INVALID, "unknown sketch algo %u", p->algo);
    if (p->data_t == GS_DATA_DNA || p->data_t == GS_DATA_DNA_FWD) {
        GS_REQUIRE(p->k >= 1 && p->k <= 32, GS_ERR_INVALID, "DNA kmer size must be in 1..32");
        GS_REQUIRE(p->k != 15, GS_ERR_INVALID, "kmer size 15 is rejected (dnarequest.rs:451-454)");
    } else if (p->data_t == GS_DATA_AA) {
        GS_REQUIRE(p->k >= 1 && p->k <= 12, GS_ERR_INVALID, "kmer for Amino Acids must be less or equal to 12");
    } else {
        GS_REQUIRE(false, GS_ERR_INVALID, "unknown data type %u", p->data_t);
    }
    return GS_OK;
}
int gs_value_bits(const gs_sketch_params *p)
{
    if (p->data_t != GS_DATA_AA) return (p->k <= 14 || p->k == 16) ? 32 : 64;
    return p->k <= 6 ? 32 : 64;
}
int gs_sig_kind(const gs_sketch_params *p)
{
    int vb = gs_value_bits(p);
    switch (p->algo) {
    case GS_ALGO_PROB3A: return vb == 32 ? GS_KIND_U32 : GS_KIND_U64;
    case GS_ALGO_SUPER2: return vb == 32 ? GS_KIND_U32 : GS_KIND_U64;
    case GS_ALGO_HLL: return GS_KIND_U16;
    default: return GS_KIND_F32;
    }
}
size_t gs_sig_elem_bytes(const gs_sketch_params *p) { return gs::kind_bytes(gs_sig_kind(p)); }

/* ---- host helpers ----------------------------------------------------------------------------- */
uint64_t gs_pack_dna(const uint8_t *ascii, uint64_t n, uint8_t *packed, uint64_t base_off)
{
    static const int8_t lut[256] = {
#define X -1
        X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,
        X,0,X,1,X,X,X,2,X,X,X,X,X,X,X,X, X,X,X,X,3,X,X,X,X,X,X,X,X,X,X,X, X,0,X,1,X,X,X,2,X,X,X,X,X,X,X,X, X,X,X,X,3,X,X,X,X,X,X,X,X,X,X,X,
        X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,
        X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X
#undef X
    };
    uint64_t w = base_off;
    for (uint64_t i = 0; i < n; i++) {
        int c = lut[ascii[i]];
        if (c < 0) continue;
        packed[w >> 2] |= (uint8_t)(c << (6 - 2 * (w & 3)));
        w++;
    }
    return w - base_off;
}
static inline int aa_valid(uint8_t c)
{
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
    switch (c) {
    case 'A': case 'C': case 'D': case 'E': case 'F': case 'G': case 'H': case 'I': case 'K': case 'L':
    case 'M': case 'N': case 'P': case 'Q': case 'R': case 'S': case 'T': case 'V': case 'W': case 'Y': return 1;
    default: return 0;
    }
}
uint64_t gs_filter_aa(const uint8_t *ascii, uint64_t n, uint8_t *out)
{
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = ascii[i];
        if (!aa_valid(c)) continue;
        if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
        out[w++] = c;
    }
    return w;
}
double gs_ani(double distance, int k, int model)
{
    double f = (1.0 - distance) * 2.0 / (1.0 - distance + 1.0);
    if (model == 1) return (1.0 + log(f) / (double)k) * 100.0;
    return pow(f, 1.0 / (double)k) * 100.0;
}

}  // extern "C"

/* ---- synthetic inputs --------------------------------------------------------------------------- */
namespace gs {
__host__ __device__ inline uint64_t synth_word(uint64_t seed, uint64_t g, uint64_t w)
{
    return splitmix_mix(seed * 0x9e3779b97f4a7c15ULL + g * 0xbf58476d1ce4e5b9ULL + w);
}
__global__ void k_synth_dna(uint64_t seed, uint64_t g0, uint64_t ng, uint64_t words_per, uint64_t len, uint64_t *out)
{
    uint64_t total = ng * words_per;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t g = i / words_per, w = i % words_per;
        uint64_t x = synth_word(seed, g0 + g, w);
        uint64_t nb = len - w * 32;                     // bases in this word
        if (nb < 32) {                                  // zero the bits past the end: bytes are stored little endian,
            uint64_t be = __builtin_bswap64(x);         // base j of the word sits at bits 62-2j of the byte-swapped value
            be &= ~(uint64_t)0 << (64 - 2 * nb);
            x = __builtin_bswap64(be);
        }
        out[i] = x;
    }
}
// proteome g, 8 residues per word: residue j of word w = "ACDEFGHIKLMNPQRSTVWY"[byte j of synth_word(seed ^ 0xAA5EED, g, w) % 20]
__global__ void k_synth_aa(uint64_t seed, uint64_t g0, uint64_t ng, uint64_t words_per, uint64_t *out)
{
    const uint64_t total = ng * words_per;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g = i / words_per, w = i % words_per;
        const uint64_t x = synth_word(seed ^ 0xAA5EEDULL, g0 + g, w);
        uint64_t o = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t r = (uint32_t)((x >> (8 * j)) & 0xFF) % 20u;
            const char *A = "ACDEFGHIKLMNPQRSTVWY";
            o |= (uint64_t)(uint8_t)A[r] << (8 * j);
        }
        out[i] = o;
    }
}
// genome g = root genome (hash(seed,g) mod n_roots) with iid substitutions at rate mu(g) ~ U[mu_lo, mu_hi]:
// 16 random bits per base (12 decide, 4 pick one of the three other bases); roots are never emitted themselves.
// root of member x: uniform (alpha == 0: hash mod n_roots) or skewed (floor(n_roots u^alpha): power-law family sizes)
__device__ __forceinline__ uint64_t synth_root(uint64_t seed, uint64_t x, uint64_t n_roots, double alpha)
{
    const uint64_t h = splitmix_mix(seed * 31 + x * 0xA24BAED4963EE407ULL + 3);
    if (alpha == 0.0) return h % n_roots;
    const uint64_t r = (uint64_t)((double)n_roots * pow((double)(h >> 11) * 0x1.0p-53, alpha));
    return r < n_roots ? r : n_roots - 1;
}
__global__ void k_synth_family(uint64_t seed, uint64_t g0, uint64_t ng, uint64_t words_per, uint64_t len, uint64_t n_roots, double mu_lo,
                               double mu_hi, double alpha, uint64_t *out)
{
    uint64_t total = ng * words_per;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t g = g0 + i / words_per, w = i % words_per;
        uint64_t root = synth_root(seed, g, n_roots, alpha);
        double mu = mu_lo + (mu_hi - mu_lo) * ((double)(splitmix_mix(seed + 0x1234567ULL * (g + 1)) >> 11) * 0x1.0p-53);
        uint32_t thr = (uint32_t)(mu * 4096.0 + 0.5);
        uint64_t be = __builtin_bswap64(synth_word(seed ^ 0x5DEECE66DULL, root, w));      // base j at bits 62-2j
        uint64_t ctr = seed * 0xD6E8FEB86659FD93ULL + g * 0xCA5A826395121157ULL + w * 8;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint64_t rnd = splitmix_mix(ctr + (uint64_t)r + 1);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                uint32_t bits = (uint32_t)(rnd >> (16 * t)) & 0xFFFF;
                if ((bits & 0xFFF) < thr) {
                    int j = r * 4 + t, sh = 62 - 2 * j;
                    uint64_t b = (be >> sh) & 3;
                    b = (b + 1 + (((bits >> 12) * 3) >> 4)) & 3;
                    be = (be & ~((uint64_t)3 << sh)) | (b << sh);
                }
            }
        }
        uint64_t nb = len - w * 32;
        if (nb < 32) be &= ~(uint64_t)0 << (64 - 2 * nb);
        out[i] = __builtin_bswap64(be);
    }
}
// row r, slot s. root = hash(seed, r) mod n_roots. root value = f(seed, root, s); member keeps it with prob J(r).
template <int KIND>
__global__ void k_synth_sigs(uint32_t m, uint64_t seed, uint64_t r0, uint64_t nrows, uint64_t n_roots, double jlo, double jhi, double alpha, void *out)
{
    uint64_t total = nrows * (uint64_t)m;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = r0 + i / m, s = i % m;
        uint64_t root = synth_root(seed, r, n_roots, alpha);
        double J = jlo + (jhi - jlo) * ((double)(splitmix_mix(seed + 0x1234567ULL * (r + 1)) >> 11) * 0x1.0p-53);
        uint64_t rootv = splitmix_mix(seed ^ (root * 0xD1342543DE82EF95ULL + s * 0x2545F4914F6CDD1DULL + 1));
        uint64_t coin = splitmix_mix(seed ^ (r * 0x9E3779B97F4A7C15ULL + s * 0xC2B2AE3D27D4EB4FULL + 7));
        uint64_t ownv = splitmix_mix(coin + 0x5851F42D4C957F2DULL);
        bool keep = ((double)(coin >> 11) * 0x1.0p-53) < J;
        uint64_t v = keep ? rootv : ownv;
        if (KIND == GS_KIND_F32) ((float *)out)[i] = (float)(uint32_t)(v >> 41) * 0x1.0p-23f;
        else if (KIND == GS_KIND_U32) ((uint32_t *)out)[i] = (uint32_t)(v >> 32);
        else if (KIND == GS_KIND_U16) ((uint16_t *)out)[i] = (uint16_t)(v >> 48);       // (SetSketch-like registers: unrelated rows agree in 1 of 65 536 slots)
        else ((uint64_t *)out)[i] = v;
    }
}
}  // namespace gs

extern "C" {
int gs_synth_dna_dev(gs_ctx *c, uint64_t seed, uint64_t g0, uint64_t ng, uint64_t len, void *seq_dev)
{
    GS_REQUIRE(c && seq_dev, GS_ERR_INVALID, "null argument");
    GS_CTX_LOCK(c);
    uint64_t wp = (len + 31) / 32;
    if (ng * wp == 0) return GS_OK;
    hipLaunchKernelGGL(gs::k_synth_dna, dim3(c->n_cu * 8), dim3(256), 0, c->stream, seed, g0, ng, wp, len, (uint64_t *)seq_dev);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
int gs_synth_aa_dev(gs_ctx *c, uint64_t seed, uint64_t g0, uint64_t ng, uint64_t len, void *seq_dev)
{
    GS_REQUIRE(c && seq_dev, GS_ERR_INVALID, "null argument");
    GS_CTX_LOCK(c);
    const uint64_t wp = (len + 7) / 8;
    if (ng * wp == 0) return GS_OK;
    hipLaunchKernelGGL(gs::k_synth_aa, dim3(c->n_cu * 8), dim3(256), 0, c->stream, seed, g0, ng, wp, (uint64_t *)seq_dev);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
int gs_synth_dna_family_skew_dev(gs_ctx *c, uint64_t seed, uint64_t g0, uint64_t ng, uint64_t len, uint64_t n_roots, double mu_lo, double mu_hi, double alpha,
                                 void *seq_dev)
{
    GS_REQUIRE(c && seq_dev && n_roots > 0 && mu_lo >= 0 && mu_hi >= mu_lo && mu_hi < 1.0 && alpha >= 0.0, GS_ERR_INVALID, "bad argument");
    GS_CTX_LOCK(c);
    uint64_t wp = (len + 31) / 32;
    if (ng * wp == 0) return GS_OK;
    hipLaunchKernelGGL(gs::k_synth_family, dim3(c->n_cu * 8), dim3(256), 0, c->stream, seed, g0, ng, wp, len, n_roots, mu_lo, mu_hi, alpha, (uint64_t *)seq_dev);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
int gs_synth_dna_family_dev(gs_ctx *c, uint64_t seed, uint64_t g0, uint64_t ng, uint64_t len, uint64_t n_roots, double mu_lo, double mu_hi,
                            void *seq_dev)
{
    return gs_synth_dna_family_skew_dev(c, seed, g0, ng, len, n_roots, mu_lo, mu_hi, 0.0, seq_dev);      // (alpha = 0: hash mod n_roots, the uniform assignment)
}
int gs_synth_sigs_skew_dev(gs_ctx *c, int kind, uint32_t m, uint64_t seed, uint64_t r0, uint64_t nrows, uint64_t n_roots,
                           double jlo, double jhi, double alpha, void *out)
{
    GS_REQUIRE(c && out && n_roots > 0 && alpha >= 0.0, GS_ERR_INVALID, "bad argument");
    GS_CTX_LOCK(c);
    if (nrows == 0) return GS_OK;
    dim3 g(c->n_cu * 8), b(256);
    if (kind == GS_KIND_F32) hipLaunchKernelGGL(gs::k_synth_sigs<GS_KIND_F32>, g, b, 0, c->stream, m, seed, r0, nrows, n_roots, jlo, jhi, alpha, out);
    else if (kind == GS_KIND_U32) hipLaunchKernelGGL(gs::k_synth_sigs<GS_KIND_U32>, g, b, 0, c->stream, m, seed, r0, nrows, n_roots, jlo, jhi, alpha, out);
    else if (kind == GS_KIND_U64) hipLaunchKernelGGL(gs::k_synth_sigs<GS_KIND_U64>, g, b, 0, c->stream, m, seed, r0, nrows, n_roots, jlo, jhi, alpha, out);
    else if (kind == GS_KIND_U16) hipLaunchKernelGGL(gs::k_synth_sigs<GS_KIND_U16>, g, b, 0, c->stream, m, seed, r0, nrows, n_roots, jlo, jhi, alpha, out);
    else GS_REQUIRE(false, GS_ERR_INVALID, "unsupported kind %d", kind);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
int gs_synth_sigs_dev(gs_ctx *c, int kind, uint32_t m, uint64_t seed, uint64_t r0, uint64_t nrows, uint64_t n_roots,
                      double jlo, double jhi, void *out)
{
    return gs_synth_sigs_skew_dev(c, kind, m, seed, r0, nrows, n_roots, jlo, jhi, 0.0, out);
}
}
