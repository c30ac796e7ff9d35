// gs_files.hip — the reader side of the path on the host: what gsearch does between a directory of FASTA files
// and the sketcher call, for a LIST of files, with the per-byte work on the device.
//   /root/reference/src/utils/files.rs:117-146   is_fasta_dna_file / is_fasta_aa_file            -> gs_is_fasta_file
//   files.rs:220-250 file_to_buffer + needletail's transparent decompression (gz / bz2 / xz)       -> gs_read_fasta_file
//   files.rs:148-215,345-455 recursive directory walk                                              -> gs_list_fasta_files
//   files.rs:258-341 process_files_group (`--pio` files read together, parsed in parallel)
//   src/dna/dnafiles.rs:43-360, src/aa/aafiles.rs:30-300 (by-sequence and --block readers)
//   src/dna/dnasketch.rs:240-300 (reader thread -> sketcher)                                        -> gs_sketch_files
// gs_sketch_files pipelines groups of files: host threads read + decompress + find record boundaries of group g+1 while the raw text
// of group g crosses PCIe from a pinned staging buffer on a copy stream and group g-1 is filtered / 2-bit packed and sketched on the
// context's stream. One signature per file.
#include <dirent.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/stat.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <string>
#include <thread>
#include <vector>
#include "gs_internal.hpp"
#include <condition_variable>
#include <deque>
#include "gs_inflate.hpp"

namespace gs {
int ingest_records_dev(gs_ctx *c, bool aa, bool contiguous, const void *text_dev, uint64_t n_bytes, const uint64_t *seq_begin, const uint64_t *seq_end,
                       uint64_t n_rec, void *out_dev, uint64_t out_base0, uint64_t *rec_start_out, uint64_t *rec_len_out, uint64_t *out_end);

// growable byte buffer WITHOUT value initialisation (std::vector::resize zero-fills: one wasted pass over every file)
struct Bytes {
    uint8_t *p = nullptr; size_t n = 0, cap = 0;
    Bytes() = default;
    Bytes(const Bytes &) = delete;
    Bytes &operator=(const Bytes &) = delete;
    Bytes(Bytes &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ~Bytes() { free(p); }
    bool reserve(size_t c) { if (c <= cap) return true; void *q = realloc(p, c); if (!q) return false; p = (uint8_t *)q; cap = c; return true; }
    bool resize(size_t c) { if (!reserve(c)) return false; n = c; return true; }
    size_t size() const { return n; }
    uint8_t *data() { return p; }
    uint8_t operator[](size_t i) const { return p[i]; }
    const uint8_t *data() const { return p; }
    bool empty() const { return n == 0; }
    void swap(Bytes &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
    void release() { free(p); p = nullptr; n = cap = 0; }
};

static bool ends_with(const std::string &s, const char *suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && !s.compare(s.size() - n, n, suf);
}

// ---- decompression by magic bytes, like needletail's parse_fastx_reader -----------------------------------------------------------
// gzip goes through libdeflate when the image has it (whole-buffer decoder, ~3x zlib's inflate on FASTA text: the gz ingest rate is bound
// by exactly this loop); it ships without headers like libbz2 / liblzma, so its three entry points are bound by hand. zlib stays as the
// fallback and as the arbiter of anything libdeflate does not accept (its error codes carry no detail).
struct LibDeflate {
    typedef void *(*fn_alloc)(void);
    typedef void (*fn_free)(void *);
    typedef int (*fn_gz)(void *d, const void *in, size_t in_n, void *out, size_t out_avail, size_t *in_used, size_t *out_used);
    fn_alloc alloc = nullptr; fn_free release = nullptr; fn_gz gunzip = nullptr;
    LibDeflate()
    {
        const char *e = getenv("GS_GZIP_IMPL");
        if (e && !strcmp(e, "zlib")) return;
        void *h = nullptr;
        for (const char *nm : {"libdeflate.so.0", "libdeflate.so"}) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) return;
        alloc = (fn_alloc)dlsym(h, "libdeflate_alloc_decompressor"); release = (fn_free)dlsym(h, "libdeflate_free_decompressor");
        gunzip = (fn_gz)dlsym(h, "libdeflate_gzip_decompress_ex");
        if (!alloc || !release || !gunzip) { alloc = nullptr; release = nullptr; gunzip = nullptr; }
    }
};
static const LibDeflate &libdeflate() { static const LibDeflate L; return L; }
struct DeflateHandle {                 // one decompressor per host thread
    void *d = nullptr;
    ~DeflateHandle() { if (d) libdeflate().release(d); }
    void *get() { if (!d && libdeflate().alloc) d = libdeflate().alloc(); return d; }
};
static thread_local DeflateHandle t_deflate;
enum { LD_OK = 0, LD_BAD_DATA = 1, LD_SHORT_OUTPUT = 2, LD_NO_SPACE = 3 };
// every member of a gzip buffer, back to back, into `out` (grown as needed). 1 = done, 0 = let zlib take (and judge) the input
static int gunzip_libdeflate(const uint8_t *in, size_t n, Bytes &out)
{
    void *d = t_deflate.get();
    if (!d) return 0;
    // a single member says how long its text is in its last four bytes (ISIZE, mod 2^32): an exact first guess for the common case
    size_t guess = std::max<size_t>(n * 4, 1 << 16);
    if (n >= 18) { const size_t isize = (size_t)in[n - 4] | (size_t)in[n - 3] << 8 | (size_t)in[n - 2] << 16 | (size_t)in[n - 1] << 24; if (isize > guess && isize < n * 64) guess = isize + 64; }
    if (!out.resize(guess)) return 0;
    size_t consumed = 0, produced = 0;
    while (consumed < n) {
        if (n - consumed < 18 || in[consumed] != 0x1f || in[consumed + 1] != 0x8b) return 0;       // trailing bytes that are not a member
        size_t iu = 0, ou = 0;
        const int rc = libdeflate().gunzip(d, in + consumed, n - consumed, out.data() + produced, out.size() - produced, &iu, &ou);
        if (rc == LD_NO_SPACE) { if (!out.resize(out.size() * 2)) return 0; continue; }
        if (rc != LD_OK || iu == 0) return 0;
        consumed += iu; produced += ou;
    }
    out.resize(produced);
    return 1;
}
static int inflate_gzip(const uint8_t *in, size_t n, Bytes &out)
{
    if (gunzip_libdeflate(in, n, out) == 1) return GS_OK;
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) { set_error("zlib inflateInit2 failed"); return GS_ERR_IO; }       // 32: gzip or zlib header
    if (!out.resize(std::max<size_t>(n * 4, 1 << 16))) { inflateEnd(&zs); set_error("out of host memory inflating %zu bytes of gzip", n); return GS_ERR_IO; }
    zs.next_in = (Bytef *)in; zs.avail_in = (uInt)std::min<size_t>(n, 1u << 30);
    size_t consumed = 0, produced = 0;
    for (;;) {
        if (produced == out.size() && !out.resize(out.size() * 2)) { inflateEnd(&zs); set_error("out of host memory inflating %zu bytes of gzip", n); return GS_ERR_IO; }
        zs.next_out = out.data() + produced; zs.avail_out = (uInt)std::min<size_t>(out.size() - produced, 1u << 30);
        const uInt in0 = zs.avail_in, out0 = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        consumed += in0 - zs.avail_in; produced += out0 - zs.avail_out;
        if (zs.avail_in == 0 && consumed < n) zs.avail_in = (uInt)std::min<size_t>(n - consumed, 1u << 30), zs.next_in = (Bytef *)in + consumed;
        if (rc == Z_STREAM_END) {
            if (consumed >= n) break;
            if (inflateReset(&zs) != Z_OK) break;                  // multi-member gzip (bgzip, concatenated files)
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { inflateEnd(&zs); set_error("gzip stream is corrupt (zlib %d)", rc); return GS_ERR_IO; }
        if (rc == Z_BUF_ERROR && zs.avail_in == 0 && consumed >= n) { inflateEnd(&zs); set_error("gzip stream is truncated"); return GS_ERR_IO; }
    }
    inflateEnd(&zs);
    out.resize(produced);
    return GS_OK;
}
// single-member gzip straight into a caller-owned buffer of `cap` bytes (the size its ISIZE trailer promises). Returns 1 when the whole
// input was one member that fitted, 0 when the general path must take over (more members, more output than promised, or an error - which
// the general path then reports)
static int inflate_gzip_into(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *produced_out)
{
    if (void *d = t_deflate.get()) {
        size_t iu = 0, ou = 0;
        if (libdeflate().gunzip(d, in, n, out, cap, &iu, &ou) == LD_OK && iu == n) { *produced_out = ou; return 1; }
        return 0;
    }
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (n > (1u << 30) || cap > (1u << 30) || inflateInit2(&zs, 15 + 32) != Z_OK) return 0;
    zs.next_in = (Bytef *)in; zs.avail_in = (uInt)n; zs.next_out = out; zs.avail_out = (uInt)cap;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.avail_in == 0;
    *produced_out = cap - zs.avail_out;
    inflateEnd(&zs);
    return ok ? 1 : 0;
}
// bz2 and xz: the image ships the shared libraries without headers, so the two stable one-shot entry points are bound by hand
typedef int (*fn_bz2)(char *dest, unsigned int *destLen, char *source, unsigned int sourceLen, int small, int verbosity);
typedef int (*fn_xz)(uint64_t *memlimit, uint32_t flags, const void *allocator, const uint8_t *in, size_t *in_pos, size_t in_size, uint8_t *out, size_t *out_pos, size_t out_size);
static int decompress_grow(const char *what, const uint8_t *in, size_t n, Bytes &out)
{
    static std::atomic<void *> h_bz2{nullptr}, h_xz{nullptr};
    const bool bz = !strcmp(what, "bz2");
    std::atomic<void *> &slot = bz ? h_bz2 : h_xz;
    void *h = slot.load();
    if (!h) {
        const char *names_bz[] = {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}, *names_xz[] = {"liblzma.so.5", "liblzma.so"};
        for (const char *nm : (bz ? std::vector<const char *>(names_bz, names_bz + 3) : std::vector<const char *>(names_xz, names_xz + 2))) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        GS_REQUIRE(h, GS_ERR_UNSUPPORTED, "%s input: the decompression library is not installed (%s)", what, dlerror());
        slot.store(h);
    }
    size_t cap = std::max<size_t>(n * 6, 1 << 16);
    for (int attempt = 0; attempt < 12; attempt++, cap *= 2) {
        GS_REQUIRE(out.resize(cap), GS_ERR_IO, "out of host memory decompressing %s input (%zu bytes wanted)", what, cap);
        if (bz) {
            fn_bz2 f = (fn_bz2)dlsym(h, "BZ2_bzBuffToBuffDecompress");
            GS_REQUIRE(f, GS_ERR_UNSUPPORTED, "libbz2 lacks BZ2_bzBuffToBuffDecompress");
            GS_REQUIRE(n < (1ull << 32) && cap < (1ull << 32), GS_ERR_UNSUPPORTED, "bz2 members beyond 4 GB are not supported");
            unsigned int dl = (unsigned int)cap;
            const int rc = f((char *)out.data(), &dl, (char *)in, (unsigned int)n, 0, 0);
            if (rc == 0) { out.resize(dl); return GS_OK; }
            GS_REQUIRE(rc == -8 /* BZ_OUTBUFF_FULL */, GS_ERR_IO, "bz2 stream is corrupt (libbz2 %d)", rc);
        } else {
            fn_xz f = (fn_xz)dlsym(h, "lzma_stream_buffer_decode");
            GS_REQUIRE(f, GS_ERR_UNSUPPORTED, "liblzma lacks lzma_stream_buffer_decode");
            uint64_t memlimit = UINT64_MAX; size_t ip = 0, op = 0;
            const int rc = f(&memlimit, 0, nullptr, in, &ip, n, out.data(), &op, cap);
            if (rc == 0) { out.resize(op); return GS_OK; }
            GS_REQUIRE(rc == 10 /* LZMA_BUF_ERROR */, GS_ERR_IO, "xz stream is corrupt (liblzma %d)", rc);
        }
    }
    GS_REQUIRE(false, GS_ERR_IO, "%s stream expands beyond every reasonable size", what);
}
static int read_whole_file(const char *path, Bytes &raw)
{
    FILE *f = fopen(path, "rb");
    GS_REQUIRE(f, GS_ERR_IO, "cannot open %s", path);
    struct stat st; size_t want = (fstat(fileno(f), &st) == 0 && st.st_size > 0) ? (size_t)st.st_size : (size_t)10000000;   // files.rs:233 fallback
    // one byte more than the file holds: the read that reaches EOF then comes back short and the buffer is never doubled (a doubling
    // realloc-copies the whole file and faults twice its pages in just to learn that nothing follows)
    if (!raw.resize(want + 1)) { fclose(f); GS_REQUIRE(false, GS_ERR_IO, "out of host memory reading %s", path); }
    size_t got = 0;
    for (;;) {
        const size_t r = fread(raw.data() + got, 1, raw.size() - got, f);
        got += r;
        if (r == 0 || got < raw.size()) { if (r == 0 || feof(f) || ferror(f)) break; continue; }
        if (!raw.resize(raw.size() * 2)) { fclose(f); GS_REQUIRE(false, GS_ERR_IO, "out of host memory reading %s", path); }
    }
    fclose(f);
    raw.resize(got);
    return GS_OK;
}
static int read_fasta(const char *path, Bytes &text)
{
    Bytes raw;
    int rc = read_whole_file(path, raw); if (rc) return rc;
    if (raw.size() >= 2 && raw[0] == 0x1f && raw[1] == 0x8b) return inflate_gzip(raw.data(), raw.size(), text);
    if (raw.size() >= 3 && raw[0] == 'B' && raw[1] == 'Z' && raw[2] == 'h') return decompress_grow("bz2", raw.data(), raw.size(), text);
    if (raw.size() >= 6 && !memcmp(raw.data(), "\xfd" "7zXZ\0", 6)) return decompress_grow("xz", raw.data(), raw.size(), text);
    text.swap(raw);
    return GS_OK;
}

// CPUs this process may really use: hardware_concurrency capped by the affinity mask and the cgroup quota (a container that lists 256
// threads with cpu.max = 16 CPUs runs 256 host threads no faster than 16, only with more contention)
static uint32_t usable_cpus()
{
    double eff = (double)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) eff = std::min(eff, (double)CPU_COUNT(&set));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0}; double per = 0;
        if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) eff = std::min(eff, atof(q) / per);
        fclose(f);
    }
    return (uint32_t)std::max(1.0, eff + 0.999);
}
struct FileBlob {       // one file after the host stage
    Bytes text;                           // decompressed text of a compressed file (plain files are read straight into the pinned buffer)
    uint8_t *dst = nullptr; size_t dst_cap = 0, dst_len = 0;    // where a plain file goes (pinned) and how much of it was filled
    uint8_t *src = nullptr; size_t src_cap = 0;                 // .gz: where the compressed bytes are read to (a per-slot buffer: no per-file allocation)
    // .gz inflated ON the device (gs_inflate.hip): the member goes to the pinned compressed buffer (src), its text will be produced at
    // g_off of the group's device-text region; gz_dev is set by the host stage when the member is one the device path takes
    bool want_dev = false, gz_dev = false; size_t gz_hdr = 0, gz_len = 0; uint64_t g_off = 0, g_cap = 0;
    // BGZF (bgzip: a .gz file of independent <= 64 KB members, each carrying its compressed size in a 'BC' extra field): the device inflates every
    // member with its own wavefront. want_bgzf: a candidate (its text size is only known once the members have been walked: g_cap is set by the
    // host stage, g_off when the group is staged); blocks: the members with data
    struct GzBlock { uint32_t in_off, in_len, out_off, isize, trailer; };
    bool want_bgzf = false; std::vector<GzBlock> blocks;
    std::vector<uint64_t> sb, se; int rc = GS_OK; std::string err; double read_s = 0;
};
static bool has_compressed_suffix(const char *path)
{
    const std::string f(path);
    return ends_with(f, ".gz") || ends_with(f, ".bz2") || ends_with(f, ".xz");
}
// record boundaries of a FASTA text, one pass when the guess of the record count holds
static int scan_records(const uint8_t *text, size_t n, std::vector<uint64_t> &sb, std::vector<uint64_t> &se)
{
    uint64_t nr = 0, cap = 1024;
    for (;;) {
        sb.resize(cap); se.resize(cap);
        const int rc = gs_fasta_scan((const char *)text, n, 1, cap, sb.data(), se.data(), nullptr, nullptr, &nr);
        if (rc) return rc;
        if (nr <= cap) break;
        cap = nr;
    }
    sb.resize(nr); se.resize(nr);
    return GS_OK;
}
// BGZF signature of a gzip member header: FEXTRA set and a 'B','C' subfield of two bytes (SAM/BAM spec 4.1); returns the member's total size
// (BSIZE + 1) and its header length, or 0
static size_t bgzf_member(const uint8_t *p, size_t n, size_t *hdr_len)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0) || !(p[3] & 4)) return 0;
    const size_t xlen = (size_t)p[10] | (size_t)p[11] << 8;
    if (12 + xlen > n) return 0;
    size_t bsize = 0;
    for (size_t q = 12; q + 4 <= 12 + xlen;) {
        const size_t slen = (size_t)p[q + 2] | (size_t)p[q + 3] << 8;
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) bsize = ((size_t)p[q + 4] | (size_t)p[q + 5] << 8) + 1;
        q += 4 + slen;
    }
    if (!bsize || (p[3] & (8 | 16 | 2))) return 0;              // (bgzip writes no name / comment / header CRC; such members go the host way)
    *hdr_len = 12 + xlen;
    return bsize >= *hdr_len + 8 && bsize <= n ? bsize : 0;
}
// a whole file as BGZF members: fills b->blocks / b->g_cap, false when the file is not exactly a sequence of BGZF members
static bool bgzf_walk(const uint8_t *p, size_t n, FileBlob *b)
{
    b->blocks.clear();
    uint64_t total = 0;
    size_t pos = 0;
    while (pos < n) {
        size_t hdr = 0;
        const size_t bs = bgzf_member(p + pos, n - pos, &hdr);
        if (!bs) return false;
        const uint8_t *t = p + pos + bs - 4;
        const uint32_t isize = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        if (isize > 65536 || total + isize >= ((uint64_t)1 << 30)) return false;
        if (isize) b->blocks.push_back({(uint32_t)(pos + hdr), (uint32_t)(bs - hdr), (uint32_t)total, isize, (uint32_t)(pos + bs - 8)});
        total += isize;
        pos += bs;
    }
    b->g_cap = total;
    return true;
}
// Host stage of one file. A file without a compression suffix is read straight into its place in the pinned staging buffer (b->dst,
// sized from stat): no intermediate allocation - hundreds of threads faulting fresh pages in contend on the process's mmap lock, which
// is what limited the group size - and no second copy. Compressed files (by suffix, or by magic bytes after all) decompress into b->text.
static void host_stage(const char *path, FileBlob *b)
{
    const auto t0 = std::chrono::steady_clock::now();
    b->rc = GS_OK; b->dst_len = 0; b->text.n = 0;
    const uint8_t *text = nullptr; size_t n = 0;
    bool direct = b->dst != nullptr;
    b->gz_dev = false;
    if (b->want_dev && b->src) {                                   // single-member .gz for the device: read the member, check its header, done
        size_t got = 0;
        FILE *f = fopen(path, "rb");
        if (f) {
            for (;;) { const size_t r = fread(b->src + got, 1, b->src_cap - got, f); got += r; if (r == 0 || got == b->src_cap) break; }
            const bool more = got == b->src_cap && fgetc(f) != EOF;
            fclose(f);
            const size_t h = more ? 0 : gzip_header_len(b->src, got);
            if (h && !b->want_bgzf) {
                const uint8_t *t = b->src + got - 4;
                const uint64_t isize = (uint64_t)t[0] | (uint64_t)t[1] << 8 | (uint64_t)t[2] << 16 | (uint64_t)t[3] << 24;
                if (isize == b->g_cap) {
                    b->gz_dev = true; b->gz_hdr = h; b->gz_len = got;
                    b->blocks.assign(1, {(uint32_t)h, (uint32_t)(got - h), 0u, (uint32_t)isize, (uint32_t)(got - 8)});
                }
            } else if (h && b->want_bgzf && bgzf_walk(b->src, got, b)) { b->gz_dev = true; b->gz_hdr = h; b->gz_len = got; }
        }
        if (b->gz_dev) { b->read_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); return; }
        direct = false;                                             // the file changed under us or is not plain gzip: the general path below
    } else if (direct && has_compressed_suffix(path)) {                   // .gz sized from its trailer: inflate straight into the pinned buffer
        size_t produced = 0, got = 0;
        direct = false;
        if (b->src) {
            FILE *f = fopen(path, "rb");
            if (f) {
                for (;;) { const size_t r = fread(b->src + got, 1, b->src_cap - got, f); got += r; if (r == 0 || got == b->src_cap) break; }
                const bool more = got == b->src_cap && fgetc(f) != EOF;
                fclose(f);
                direct = !more && got >= 2 && b->src[0] == 0x1f && b->src[1] == 0x8b && inflate_gzip_into(b->src, got, b->dst, b->dst_cap, &produced) == 1;
            }
        }
        if (direct) { b->dst_len = produced; text = b->dst; n = produced; }
    } else if (direct) {
        FILE *f = fopen(path, "rb");
        if (!f) { set_error("cannot open %s", path); b->rc = GS_ERR_IO; }
        else {
            size_t got = 0;
            for (;;) { const size_t r = fread(b->dst + got, 1, b->dst_cap - got, f); got += r; if (r == 0 || got == b->dst_cap) break; }
            const bool more = got == b->dst_cap && fgetc(f) != EOF;               // the file grew since it was sized: take the general path
            fclose(f);
            const uint8_t *d = b->dst;
            const bool packed = (got >= 2 && d[0] == 0x1f && d[1] == 0x8b) || (got >= 3 && d[0] == 'B' && d[1] == 'Z' && d[2] == 'h') ||
                                (got >= 6 && !memcmp(d, "\xfd" "7zXZ\0", 6));
            if (more || packed) direct = false;
            else { b->dst_len = got; text = d; n = got; }
        }
    }
    if (!direct && b->rc == GS_OK) {
        b->rc = read_fasta(path, b->text);
        text = b->text.data(); n = b->text.size();
    }
    if (b->rc == GS_OK) b->rc = scan_records(text, n, b->sb, b->se);
    if (b->rc) b->err = gs_last_error();                          // thread-local: carry it to the caller's thread
    b->read_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace gs

extern "C" {

/* files.rs:117-146: 1 when `path` carries one of the FASTA suffixes gsearch accepts for the data type (DNA: fna fa fasta, also .gz .xz .bz2;
 * AA: faa, also .gz .xz .bz2) */
int gs_is_fasta_file(const char *path, int data_t)
{
    if (!path) return 0;
    const std::string f(path);
    if (data_t == GS_DATA_AA) return gs::ends_with(f, "faa.gz") || gs::ends_with(f, "faa") || gs::ends_with(f, "faa.xz") || gs::ends_with(f, "faa.bz2");
    static const char *suf[] = {"fna.gz", "fa.gz", "fa.xz", "fna.xz", "fasta.xz", "fa.bz2", "fna.bz2", "fasta.bz2", "fasta.gz", "fna", "fa", "fasta"};
    for (const char *s : suf) if (gs::ends_with(f, s)) return 1;
    return 0;
}

/* read a FASTA file into memory, decompressing gzip (multi-member), bzip2 or xz by magic bytes (needletail::parse_fastx_reader does the
 * same behind files.rs:220-250 file_to_buffer). *text_out is malloc-ed: release with gs_host_free. */
int gs_read_fasta_file(const char *path, void **text_out, uint64_t *n_out)
{
    GS_REQUIRE(path && text_out && n_out, GS_ERR_INVALID, "null argument");
    gs::Bytes text;
    int rc = gs::read_fasta(path, text); if (rc) return rc;
    void *p = malloc(text.size() + 1);
    GS_REQUIRE(p, GS_ERR_IO, "out of host memory");
    memcpy(p, text.data(), text.size());
    *text_out = p; *n_out = text.size();
    return GS_OK;
}
void gs_host_free(void *p) { free(p); }

/* files.rs:148-215 / 345-455: every accepted file under `dir`, recursively, directory entries in name order (read_dir order is
 * unspecified upstream). Two calls: paths_buf NULL / cap 0 to learn *n_out and *bytes_out, then a buffer that receives the
 * NUL-terminated paths back to back. */
int gs_list_fasta_files(const char *dir, int data_t, char *paths_buf, uint64_t cap_bytes, uint64_t *n_out, uint64_t *bytes_out)
{
    GS_REQUIRE(dir && n_out && bytes_out, GS_ERR_INVALID, "null argument");
    std::vector<std::string> found, stack{std::string(dir)};
    while (!stack.empty()) {
        const std::string d = stack.back(); stack.pop_back();
        DIR *h = opendir(d.c_str());
        GS_REQUIRE(h, GS_ERR_IO, "directory %s does not exist or is not readable", d.c_str());
        std::vector<std::string> files, dirs;
        while (struct dirent *e = readdir(h)) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            const std::string p = d + "/" + e->d_name;
            struct stat st;
            if (stat(p.c_str(), &st)) continue;
            if (S_ISDIR(st.st_mode)) dirs.push_back(p);
            else if (gs_is_fasta_file(p.c_str(), data_t)) files.push_back(p);
        }
        closedir(h);
        std::sort(files.begin(), files.end()); std::sort(dirs.rbegin(), dirs.rend());
        found.insert(found.end(), files.begin(), files.end());
        stack.insert(stack.end(), dirs.begin(), dirs.end());
    }
    uint64_t bytes = 0;
    for (auto &f : found) bytes += f.size() + 1;
    *n_out = found.size(); *bytes_out = bytes;
    if (paths_buf && cap_bytes >= bytes) { char *w = paths_buf; for (auto &f : found) { memcpy(w, f.c_str(), f.size() + 1); w += f.size() + 1; } }
    return GS_OK;
}

/*
 * One signature per file, in input order, for n_files FASTA files (plain, .gz, .bz2, .xz):
 *   block_mode 0  by sequence: every record is its own sequence, k-mers never span records (process_file_by_sequence)
 *   block_mode 1  --block: the records of a file are concatenated, k-mers span the joins (process_file_in_one_block)
 * `capsid` records are skipped in both (dnafiles.rs:62-67,245). pio = files per group (`--pio`, files.rs:258-341; 0 -> 64),
 * n_threads = host threads that read / decompress / scan (0 -> hardware concurrency).
 * sig_out: HOST, n_files x sketch_size elements. n_records_out / n_symbols_out (optional, HOST, per file): records kept and bases /
 * residues that reached the sketcher. stats_out (optional, 6 doubles, see include/gsearch_amd.h): host seconds spent reading+decompressing+scanning (summed over
 * threads), seconds the caller waited for PCIe copies, seconds in device pack + sketch, wall seconds of the call.
 */
// The .gz files of a call are dealt between two pipelines as they go (see gs_sketch_files): the host pipeline claims them from the front of
// the list, the device pipeline from the back, group by group, until the two meet.
// The device pipeline takes a group only as large as lets both finish together. Host pipeline: rate ra (files/s, measured once 256 files are
// done, a prior before). Device pipeline: a start-up latency (reading, copying and inflating its first group: nothing is done before) and a
// rate rb behind it. With pa, pb files claimed but unfinished and R unclaimed: lat + (pb + k) / rb = (R - k + pa) / ra.
struct GzDeal {
    std::mutex m; uint64_t n_gz = 0, front = 0, back = 0, done_front = 0, done_back = 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    static constexpr double LAT = 0.45;
    double rate(bool back_side) const
    {
        const uint64_t d = back_side ? done_back : done_front;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - (back_side ? LAT : 0.0);
        return d >= 256 && el > 0.05 ? (double)d / el : (back_side ? 3500.0 : 2300.0);
    }
    // members the device pipeline should take now out of `avail`
    double device_share(uint64_t avail) const
    {
        const double ra = rate(false), rb = rate(true);
        const double pa = (double)(front - done_front), pb = (double)(back - done_back);
        return (((double)avail + pa) / ra - pb / rb - (done_back ? 0.0 : LAT)) / (1.0 / rb + 1.0 / ra);
    }
};

// dev_gzip: single-member .gz files are inflated on the device (gs_inflate.hip); members it does not take or that fail their trailer check are
// collected in `redo` (file indices) with zero records, for the caller to run through the host decoders
static int sketch_files_impl(gs_ctx *c, const gs_sketch_params *p, const char *const *paths, uint64_t n_files, int block_mode, uint32_t pio, uint32_t n_threads,
                             void *sig_out, uint64_t *n_records_out, uint64_t *n_symbols_out, double *stats_out, bool dev_gzip, std::vector<uint64_t> *redo,
                             const uint64_t *out_index /* row of file f in the outputs; NULL: f */, int lookahead /* groups read ahead; 0: default */,
                             GzDeal *deal /* optional */, bool from_back, uint64_t n_free /* leading files of the list that need no claim */)
{
    int rc = gs_check_params(p);
    if (rc) return rc;
    GS_REQUIRE(c && (n_files == 0 || (paths && sig_out)), GS_ERR_INVALID, "null argument");
    if (n_files == 0) return GS_OK;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const auto t_call = std::chrono::steady_clock::now();
    const bool aa = p->data_t == GS_DATA_AA;
    struct WaitMode { gs_ctx *c; bool prev; ~WaitMode() { c->wait_sleeping = prev; } } wait_mode{c, c->wait_sleeping};
    c->wait_sleeping = dev_gzip;     // the device pipeline waits for hundreds of milliseconds at a time: asleep; the host pipeline: spinning
    if (pio == 0) {                  // small groups overlap best on the host path (measured: 32 files per group 3200 genomes/s, 64: 2400, 256: 900);
        pio = 32;                    // the device inflates one member per wave: it wants thousands of members per group
        if (dev_gzip) {
            uint64_t ngz = 0;
            for (uint64_t f = 0; f < n_files; f++) ngz += gs::ends_with(paths[f], ".gz");
            // (beyond 4 per CU the launcher takes the window-less form of k_inflate. Six members per CU: the host pipeline's pack / sketch launches run
            // beside them almost unhindered - 41 ms against 34 alone for 64 genomes, 105 ms beside twelve per CU: tools/inflate_corun.py)
            if (2 * ngz > n_files) { const char *e = getenv("GS_GZIP_GROUP"); pio = (uint32_t)std::max(32, e ? atoi(e) : 6 * c->n_cu); }
        }
    }
    if (n_threads == 0) n_threads = gs::usable_cpus();
    const uint64_t n_groups = (n_files + pio - 1) / pio;
    std::vector<uint64_t> claimed_gz(n_groups, 0);
    uint64_t n_groups_eff = n_groups;                              // fewer when the other pipeline took the rest of the .gz files
    const size_t esz = gs_sig_elem_bytes(p), m = p->sketch_size;
    std::vector<std::vector<gs::FileBlob>> blobs(n_groups);
    // host stage: ONE team of n_threads workers takes the files of the started groups in order (files.rs:327 par_iter over a group). A team per
    // group - the round-2 form - put (look-ahead x n_threads) threads on the cores at once: deeper look-ahead only made them contend.
    struct HostTeam {
        std::mutex m; std::condition_variable work, done;
        std::deque<std::pair<uint64_t, uint64_t>> q;          // (group, file of the group)
        std::vector<uint64_t> left;                           // per group: files not finished yet
        bool stop = false; std::vector<std::thread> th;
    } team;
    team.left.assign(n_groups, 0);
    // LA groups are being read ahead while group g copies and g-1 is on the device: LA + 2 pinned staging buffers
    constexpr int LA_MAX = 14;
    int LA = 2;                        // (measured: 2 -> 3100-3400 genomes/s plain, 980 gz; 6 -> 2400 / 850; 12 -> 1650 / 840: more host threads only contend)
    if (lookahead > 0) LA = std::min(LA_MAX, lookahead);
    if (getenv("GS_INGEST_LOOKAHEAD")) LA = std::max(1, std::min(LA_MAX, atoi(getenv("GS_INGEST_LOOKAHEAD"))));
    const int NSLOT = LA + 2;
    // pinned staging buffers live in the context (gs::PinnedPool): they outlast the call, gs_ctx_release_scratch / gs_ctx_destroy free them
    gs::PinnedPool *pool = gs::pinned_pool(c);
    static_assert(LA_MAX + 2 <= 16, "text slots 0-15, compressed slots 16-31 of the pinned pool");
    void *pinned[LA_MAX + 2] = {}; size_t pinned_cap[LA_MAX + 2] = {};
    for (int i = 0; i < NSLOT; i++) { pinned[i] = pool->p[i]; pinned_cap[i] = pool->cap[i]; }
    std::vector<uint64_t> plain_total(n_groups, 0);
    gs::Bytes cbuf[LA_MAX + 2];                               // per slot: the compressed bytes of the group's .gz files
    void *cpin[LA_MAX + 2] = {}; size_t cpin_cap[LA_MAX + 2] = {};       // per slot, pinned: the members the DEVICE inflates
    for (int i = 0; i < NSLOT; i++) { cpin[i] = pool->p[16 + i]; cpin_cap[i] = pool->cap[16 + i]; }
    std::vector<uint64_t> dev_ctot(n_groups, 0), dev_gtot(n_groups, 0);    // per group: bytes of those members / of their texts
    // A device group is also bounded in BYTES (a group of 6 x CUs eukaryote-sized or highly compressible members would ask for tens of GB): its
    // texts live twice on the device (two parities) next to the packed output, its compressed bytes twice there and NSLOT times in pinned memory.
    // Members beyond the budget stay with this pipeline's host decoders.
    uint64_t dev_text_budget = (uint64_t)8 << 30, dev_comp_budget = (uint64_t)2 << 30;
    if (dev_gzip) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) dev_text_budget = std::max<uint64_t>((uint64_t)256 << 20, std::min<uint64_t>((uint64_t)24 << 30, (uint64_t)fr / 8));
        dev_comp_budget = std::max<uint64_t>((uint64_t)64 << 20, std::min<uint64_t>((uint64_t)4 << 30, dev_text_budget / 2));
        if (const char *e = getenv("GS_GZIP_GROUP_MB")) { dev_text_budget = (uint64_t)std::max(1, atoi(e)) << 20; dev_comp_budget = std::max<uint64_t>(dev_text_budget / 2, (uint64_t)1 << 20); }
    }
    int start_rc = GS_OK;
    // host stage of a group: its files are spread over n_threads threads (files.rs:327 par_iter over the group)
    auto start_group = [&](uint64_t g) {
        const uint64_t f0 = g * pio;
        uint64_t f1 = std::min<uint64_t>(n_files, f0 + pio);
        if (deal) {                                                // take what is left of the .gz files for this group; a short group is the last one
            const uint64_t cnt = f1 - std::max(f0, std::min(n_free, f1));
            std::lock_guard<std::mutex> lk(deal->m);
            const uint64_t avail = deal->n_gz - deal->front - deal->back;
            uint64_t take = std::min(cnt, avail);
            if (from_back) {
                const double k = deal->device_share(avail);
                take = k < 128 ? 0 : std::min<uint64_t>(take, (uint64_t)k);          // a launch costs the same ~0.4 s for 100 members as for 1000
            }
            (from_back ? deal->back : deal->front) += take;
            claimed_gz[g] = take;
            if (take < cnt) { f1 -= cnt - take; n_groups_eff = std::min<uint64_t>(n_groups_eff, f1 > f0 ? g + 1 : g); }
        }
        blobs[g].resize(f1 - f0);
        if (f1 == f0) return;
        {   // size the plain files and give each its place in this group's pinned buffer (its previous user, group g-4, is long done)
            const int sl = (int)(g % NSLOT);
            std::vector<uint64_t> off(f1 - f0, 0), cap(f1 - f0, 0), coff(f1 - f0, 0), ccap(f1 - f0, 0), doff(f1 - f0, 0), dcap(f1 - f0, 0), goff(f1 - f0, 0);
            std::vector<uint8_t> isbg(f1 - f0, 0);
            uint64_t tot = 0, ctot = 0, dtot = 0, gtot = 0, gtot_bgzf = 0;
            auto fz2_is_bgzf = [](const char *path) {               // the first member's header carries the 'BC' extra field
                uint8_t h[64]; size_t hl = 0;
                FILE *fz = fopen(path, "rb");
                const size_t got = fz ? fread(h, 1, sizeof h, fz) : 0;
                if (fz) fclose(fz);
                if (got < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
                const size_t xlen = (size_t)h[10] | (size_t)h[11] << 8;
                (void)hl;
                for (size_t q = 12; q + 6 <= std::min<size_t>(12 + xlen, got);) {
                    const size_t slen = (size_t)h[q + 2] | (size_t)h[q + 3] << 8;
                    if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2) return true;
                    q += 4 + slen;
                }
                return false;
            };
            for (uint64_t f = f0; f < f1; f++) {
                struct stat st;
                if (stat(paths[f], &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) continue;
                uint64_t want = 0;
                if (!gs::has_compressed_suffix(paths[f])) want = (uint64_t)st.st_size;
                else if (gs::ends_with(paths[f], ".gz") && st.st_size >= 18) {            // gzip: the last four bytes are the text size (mod 2^32)
                    FILE *fz = fopen(paths[f], "rb");
                    uint8_t t4[4];
                    if (fz && fseek(fz, -4, SEEK_END) == 0 && fread(t4, 1, 4, fz) == 4) want = (uint64_t)t4[0] | (uint64_t)t4[1] << 8 | (uint64_t)t4[2] << 16 | (uint64_t)t4[3] << 24;
                    if (fz) fclose(fz);
                    if (want < (uint64_t)st.st_size / 2 || want > (uint64_t)st.st_size * 64 || want >= (1u << 30)) want = 0;       // not a plausible single member
                    if (!want && dev_gzip && fz2_is_bgzf(paths[f]) && (uint64_t)st.st_size < ((uint64_t)1 << 30) && dtot + (uint64_t)st.st_size < ((uint64_t)12 << 30) &&
                        gtot_bgzf + 4 * (uint64_t)st.st_size <= dev_text_budget && dtot + (uint64_t)st.st_size <= dev_comp_budget) {
                        // bgzip: the members are found by the host stage, the text gets its place when the group is staged (4x the file as a budget guess)
                        doff[f - f0] = dtot; dcap[f - f0] = (uint64_t)st.st_size; dtot += ((uint64_t)st.st_size + 63) / 64 * 64;
                        gtot_bgzf += 4 * (uint64_t)st.st_size; isbg[f - f0] = 1;
                        continue;
                    }
                    if (want && dev_gzip && dtot + (uint64_t)st.st_size < ((uint64_t)12 << 30) &&          // (k_inflate indexes the compressed words of a launch with 32 bits)
                        gtot + want <= dev_text_budget && dtot + (uint64_t)st.st_size <= dev_comp_budget) {   // the device's: member -> pinned compressed buffer, text -> the group's device-text region
                        doff[f - f0] = dtot; dcap[f - f0] = (uint64_t)st.st_size; dtot += ((uint64_t)st.st_size + 63) / 64 * 64;
                        goff[f - f0] = gtot; cap[f - f0] = want; gtot += (want + 63) / 64 * 64;
                        continue;
                    }
                    if (want) { coff[f - f0] = ctot; ccap[f - f0] = (uint64_t)st.st_size; ctot += ((uint64_t)st.st_size + 63) / 64 * 64; }
                }
                if (want) { off[f - f0] = tot; cap[f - f0] = want; tot += (want + 63) / 64 * 64; }
            }
            plain_total[g] = tot;
            if (tot + 64 > pinned_cap[sl]) {
                pinned[sl] = pool->ensure(sl, (tot + 64) * 5 / 4);
                pinned_cap[sl] = pool->cap[sl];
                if (!pinned[sl]) { pinned_cap[sl] = 0; start_rc = GS_ERR_HIP; gs::set_error("hipHostMalloc of %zu bytes failed", (size_t)((tot + 64) * 5 / 4)); }
            }
            dev_ctot[g] = dtot; dev_gtot[g] = gtot; (void)gtot_bgzf;
            if (dtot + 64 > cpin_cap[sl]) {
                cpin[sl] = pool->ensure(16 + sl, (dtot + 64) * 5 / 4);
                cpin_cap[sl] = pool->cap[16 + sl];
                if (!cpin[sl]) {          // no pinned room for the members: they stay with this pipeline's host decoders (host_stage's general path)
                    cpin_cap[sl] = 0; (void)hipGetLastError();
                    dev_ctot[g] = dev_gtot[g] = 0;
                    for (uint64_t f = f0; f < f1; f++) if (dcap[f - f0]) { dcap[f - f0] = 0; cap[f - f0] = 0; }
                }
            }
            const bool have_c = ctot == 0 || cbuf[sl].reserve(ctot + 64);
            for (uint64_t f = f0; f < f1; f++) {
                gs::FileBlob &fb = blobs[g][f - f0];
                fb.want_dev = dcap[f - f0] != 0 && cpin[sl] != nullptr;
                fb.want_bgzf = fb.want_dev && isbg[f - f0];
                if (fb.want_dev) {
                    fb.dst = nullptr; fb.dst_cap = 0;
                    fb.src = (uint8_t *)cpin[sl] + doff[f - f0]; fb.src_cap = dcap[f - f0]; fb.g_off = goff[f - f0]; fb.g_cap = cap[f - f0];
                    continue;
                }
                fb.dst = (cap[f - f0] && pinned[sl] && !dcap[f - f0]) ? (uint8_t *)pinned[sl] + off[f - f0] : nullptr; fb.dst_cap = fb.dst ? cap[f - f0] : 0;
                fb.src = (ccap[f - f0] && have_c) ? cbuf[sl].data() + coff[f - f0] : nullptr; fb.src_cap = ccap[f - f0];
            }
        }
        {
            std::lock_guard<std::mutex> lk(team.m);
            team.left[g] = f1 - f0;
            for (uint64_t f = f0; f < f1; f++) team.q.emplace_back(g, f - f0);
        }
        team.work.notify_all();
    };
    auto wait_group = [&](uint64_t g) { std::unique_lock<std::mutex> lk(team.m); team.done.wait(lk, [&] { return team.left[g] == 0; }); };
    hipStream_t copy_stream = nullptr, inflate_stream = nullptr;
    GS_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    if (dev_gzip) {
        // GS_INFLATE_FREE_CUS=n (measurement aid, default 0): a CU mask keeps n CUs out of the inflate stream's reach. Tried in round 4 against the
        // theory that the HOST pipeline's pack / sketch launches queue behind the resident inflate workgroups: 8192 x 5 Mbp gzip -6 files ran at
        // 3484 / 3480 / 2320 / 2856 genomes/s with 0 / 16 / 32 / 64 CUs kept free (profiles/r04_ingest_cumask.log) - residency is not what couples
        // the two pipelines; the inflate kernel simply loses the CUs it is denied.
        hipError_t se = hipErrorUnknown;
        const char *fe = getenv("GS_INFLATE_FREE_CUS");
        const int free_cus = fe ? atoi(fe) : 0;
        if (free_cus > 0 && free_cus < c->n_cu) {
            std::vector<uint32_t> cumask((c->n_cu + 31) / 32, 0xFFFFFFFFu);
            for (int i = 0; i < free_cus; i++) cumask[i / 32] &= ~(1u << (i % 32));
            if (c->n_cu % 32) cumask.back() &= (1u << (c->n_cu % 32)) - 1;
            se = hipExtStreamCreateWithCUMask(&inflate_stream, (uint32_t)cumask.size(), cumask.data());
            if (se != hipSuccess) { (void)hipGetLastError(); inflate_stream = nullptr; }
        }
        if (se != hipSuccess && hipStreamCreateWithFlags(&inflate_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(copy_stream); gs::set_error("hipStreamCreate failed"); return GS_ERR_HIP; }
    }
    hipEvent_t ev[2] = {nullptr, nullptr}, iev[2] = {nullptr, nullptr};
    // the device's .gz members of group g are inflated on their own stream as soon as they are on the device - under the crc / scan / pack / sketch
    // of group g - 1; per buffer parity: descriptors, the files they belong to, device copies, pinned results
    std::vector<gs::InflateStream> ist[2]; std::vector<uint64_t> iwho[2]; std::vector<const uint8_t *> itrail[2]; gs::DevBuf ids[2], idr[2]; bool ilaunched[2] = {false, false};
    gs::DevBuf dtext[2], dcomp[2], dout, drs, drl, dgo, dsig;
    std::vector<uint8_t> rows_tmp;
    double read_s = 0, copy_wait_s = 0, dev_s = 0;
    double dsub[4] = {0, 0, 0, 0};           // GS_INGEST_TIMES: inside the device stage - wait for the inflate / crc / record scan / pack + sketch + results
    // per staged group: text bytes (H2D part, then the device-inflated texts from gbase on), record ranges per file, then flattened
    struct Staged { uint64_t bytes = 0, gbase = 0; std::vector<std::vector<uint64_t>> fsb, fse; std::vector<uint64_t> sb, se, frec; } staged[2];
    auto cleanup = [&]() {
        { std::lock_guard<std::mutex> lk(team.m); team.stop = true; }      // the workers finish what is queued (it points into buffers freed below), then leave
        team.work.notify_all();
        for (auto &x : team.th) x.join();
        team.th.clear();
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);      // an H2D copy out of a pinned buffer may still be in flight on an error path
        (void)hipStreamSynchronize(c->stream);
        if (inflate_stream) (void)hipStreamSynchronize(inflate_stream);
        for (int i = 0; i < 2; i++) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (iev[i]) (void)hipEventDestroy(iev[i]); }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (inflate_stream) (void)hipStreamDestroy(inflate_stream);
    };
#define GS_FILES_FAIL(code) do { const int rc_ = (code); cleanup(); return rc_; } while (0)
    for (int i = 0; i < 2; i++) if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | (dev_gzip ? hipEventBlockingSync : 0)) != hipSuccess ||
                                    hipEventCreateWithFlags(&iev[i], hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { gs::set_error("hipEventCreate failed"); GS_FILES_FAIL(GS_ERR_HIP); }
    for (uint32_t t = 0; t < std::max<uint32_t>(1, (uint32_t)std::min<uint64_t>(n_threads, n_files)); t++)
        team.th.emplace_back([&]() {
            for (;;) {
                std::pair<uint64_t, uint64_t> job;
                {
                    std::unique_lock<std::mutex> lk(team.m);
                    team.work.wait(lk, [&] { return team.stop || !team.q.empty(); });
                    if (team.q.empty()) return;
                    job = team.q.front(); team.q.pop_front();
                }
                gs::host_stage(paths[job.first * pio + job.second], &blobs[job.first][job.second]);
                std::lock_guard<std::mutex> lk(team.m);
                if (--team.left[job.first] == 0) team.done.notify_all();
            }
        });
    for (int g = 0; g < LA && (uint64_t)g < n_groups_eff; g++) start_group((uint64_t)g);
    // stage group g: wait for its host tasks, lay the texts of its files end to end in pinned memory, start the H2D copy
    auto stage = [&](uint64_t g) -> int {
        wait_group(g);
        if (start_rc) return start_rc;
        const int b = (int)(g & 1), sl = (int)(g % NSLOT);
        Staged &S = staged[b];
        S.fsb.assign(blobs[g].size(), {}); S.fse.assign(blobs[g].size(), {});
        // plain files sit where start_group put them; the texts of decompressed files are appended behind them
        uint64_t total = plain_total[g];
        std::vector<uint64_t> base(blobs[g].size());
        for (size_t f = 0; f < blobs[g].size(); f++) {
            auto &fb = blobs[g][f];
            if (fb.rc) { gs::set_error("%s", fb.err.c_str()); return fb.rc; }
            read_s += fb.read_s;
            if (fb.gz_dev) continue;                                 // its text does not exist yet
            if (fb.dst_len || fb.text.empty()) base[f] = fb.dst ? (uint64_t)(fb.dst - (uint8_t *)pinned[sl]) : 0;
            else { base[f] = total; total += (fb.text.size() + 63) / 64 * 64; }
            for (size_t r = 0; r < fb.sb.size(); r++) { S.fsb[f].push_back(base[f] + fb.sb[r]); S.fse[f].push_back(base[f] + fb.se[r]); }
        }
        S.gbase = (total + 63) / 64 * 64;
        for (size_t f = 0; f < blobs[g].size(); f++) {              // bgzip files: their text size is known now
            auto &fb = blobs[g][f];
            if (!fb.gz_dev || !fb.want_bgzf) continue;
            if (dev_gtot[g] + fb.g_cap > 2 * dev_text_budget) { fb.gz_dev = false; if (redo) redo->push_back(g * pio + f); continue; }      // (far beyond the guess: host decoders)
            fb.g_off = dev_gtot[g]; dev_gtot[g] += (fb.g_cap + 63) / 64 * 64;
        }
        S.bytes = dev_gtot[g] ? S.gbase + dev_gtot[g] : total;
        if (total + 64 > pinned_cap[sl]) {                          // decompressed texts do not fit behind the plain files: grow, keep what is there
            void *np = nullptr; const size_t ncap = (total + 64) * 5 / 4;
            GS_HIP_CHECK(hipHostMalloc(&np, ncap, hipHostMallocDefault));
            if (pinned[sl]) { memcpy(np, pinned[sl], plain_total[g]); (void)hipHostFree(pinned[sl]); }
            pinned[sl] = np; pinned_cap[sl] = ncap;
            pool->p[sl] = np; pool->cap[sl] = ncap;
        }
        {   // a team of threads copies the decompressed texts in (one thread moves ~8 GB/s)
            std::atomic<size_t> next{0};
            auto copier = [&]() { for (;;) { const size_t f = next.fetch_add(1); if (f >= blobs[g].size()) break; auto &fb = blobs[g][f];
                                             if (!fb.dst_len && !fb.text.empty()) memcpy((uint8_t *)pinned[sl] + base[f], fb.text.data(), fb.text.size());
                                             fb.text.release(); } };
            size_t n_ext = 0;
            for (auto &fb : blobs[g]) n_ext += (!fb.dst_len && !fb.text.empty());
            std::vector<std::thread> team;
            const uint32_t nt = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(n_threads, 32), n_ext);
            for (uint32_t t = 1; t < nt; t++) team.emplace_back(copier);
            copier();
            for (auto &x : team) x.join();
        }
        int rc2;
        if (dev_gtot[g] && (dtext[b].ensure(S.bytes + 64) != GS_OK || dcomp[b].ensure(dev_ctot[g] + 64) != GS_OK)) {
            // the device has no room for this group's inflated texts: its members are handed to the host decoders (redo) instead of failing the call
            (void)hipGetLastError();
            for (uint64_t f = 0; f < blobs[g].size(); f++) if (blobs[g][f].gz_dev) { blobs[g][f].gz_dev = false; if (redo) redo->push_back(g * pio + f); }
            dev_gtot[g] = dev_ctot[g] = 0;
            S.bytes = total;
            GS_REQUIRE(redo, GS_ERR_HIP, "device memory exhausted while staging gzip members");
        }
        if ((rc2 = dtext[b].ensure(S.bytes + 64))) return rc2;
        if (total) GS_HIP_CHECK(hipMemcpyAsync(dtext[b].p, pinned[sl], total, hipMemcpyHostToDevice, copy_stream));
        if (dev_ctot[g]) {
            if ((rc2 = dcomp[b].ensure(dev_ctot[g] + 64))) return rc2;
            // in pieces: one multi-GB copy command holds the copy engine while the OTHER pipeline's 300 MB text copies queue behind it
            static const uint64_t piece = [] { const char *e = getenv("GS_COPY_CHUNK_MB"); return (uint64_t)std::max(1, e ? atoi(e) : 64) << 20; }();
            for (uint64_t o = 0; o < dev_ctot[g]; o += piece)
                GS_HIP_CHECK(hipMemcpyAsync((uint8_t *)dcomp[b].p + o, (const uint8_t *)cpin[sl] + o, std::min<uint64_t>(piece, dev_ctot[g] - o), hipMemcpyHostToDevice, copy_stream));
        }
        GS_HIP_CHECK(hipEventRecord(ev[b], copy_stream));
        return GS_OK;
    };
    // start the inflate of group g's members (behind its H2D copies), on the inflate stream
    auto inflate_launch = [&](uint64_t g) -> int {
        const int b = (int)(g & 1), sl = (int)(g % NSLOT);
        ilaunched[b] = false;
        if (!dev_gtot[g]) return GS_OK;
        Staged &S = staged[b];
        ist[b].clear(); iwho[b].clear(); itrail[b].clear();
        for (uint64_t f = 0; f < blobs[g].size(); f++) {
            const auto &fb = blobs[g][f];
            if (!fb.gz_dev) continue;
            const uint64_t cbase = (uint64_t)(fb.src - (uint8_t *)cpin[sl]);
            for (const auto &blk : fb.blocks) {                        // one wavefront per member: a single-member file has one, a bgzip file one per <= 64 KB
                ist[b].push_back({cbase + blk.in_off, blk.in_len, S.gbase + fb.g_off + blk.out_off, blk.isize});
                iwho[b].push_back(f);
                itrail[b].push_back(fb.src + blk.trailer);
            }
        }
        const size_t ns = ist[b].size();
        if (!ns) return GS_OK;
        int rc2;
        uint8_t *pin = (uint8_t *)pool->ensure(32 + b, (sizeof(gs::InflateStream) + sizeof(gs::InflateResult)) * ns + 64);
        if (!pin) { gs::set_error("hipHostMalloc failed"); return GS_ERR_HIP; }
        memcpy(pin, ist[b].data(), sizeof(gs::InflateStream) * ns);
        if ((rc2 = ids[b].ensure(sizeof(gs::InflateStream) * ns)) || (rc2 = idr[b].ensure(sizeof(gs::InflateResult) * ns))) return rc2;
        GS_HIP_CHECK(hipStreamWaitEvent(inflate_stream, ev[b], 0));
        if ((rc2 = gs::inflate_streams_launch(c, inflate_stream, dcomp[b].p, (const gs::InflateStream *)pin, (uint32_t)ns, dtext[b].p, ids[b].p, idr[b].p,
                                              (gs::InflateResult *)(pin + sizeof(gs::InflateStream) * ns)))) return rc2;
        GS_HIP_CHECK(hipEventRecord(iev[b], inflate_stream));
        ilaunched[b] = true;
        return GS_OK;
    };
    // device stage of group g (its text is, or soon will be, in dtext[g & 1])
    auto device_stage = [&](uint64_t g) -> int {
        const int b = (int)(g & 1);
        Staged &S = staged[b];
        const uint64_t nf = blobs[g].size(), f0 = g * pio;
        auto t0 = std::chrono::steady_clock::now();
        GS_HIP_CHECK(hipEventSynchronize(ev[b]));
        copy_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        auto tl = t0;
        auto sub = [&](int k) { const auto now = std::chrono::steady_clock::now(); dsub[k] += std::chrono::duration<double>(now - tl).count(); tl = now; };
        int rc2;
        if (dev_gtot[g]) {          // inflate the group's .gz members on the device, check them against their trailers, find their records
            std::vector<gs::InflateStream> &st = ist[b]; std::vector<uint64_t> who = iwho[b];
            std::vector<gs::InflateResult> res(st.size());
            if (ilaunched[b]) {
                GS_HIP_CHECK(hipEventSynchronize(iev[b]));
                memcpy(res.data(), (const uint8_t *)pool->p[32 + b] + sizeof(gs::InflateStream) * st.size(), sizeof(gs::InflateResult) * st.size());
                ilaunched[b] = false;
            }
            sub(0);
            // every member: inflated completely, to its ISIZE, and its CRC-32 checks; a file is good when all of its members are
            std::vector<uint64_t> toff(st.size()), tlen(st.size());
            std::vector<uint8_t> bad(nf, 0);
            for (size_t k = 0; k < st.size(); k++) {
                const bool ok = res[k].status == 0 && res[k].in_used + 8 == st[k].in_len && res[k].out_len == st[k].out_cap;
                toff[k] = st[k].out_off; tlen[k] = ok ? res[k].out_len : 0;
                if (!ok) {
                    if (!bad[who[k]] && getenv("GS_INGEST_VERBOSE"))
                        fprintf(stderr, "[GS_INGEST] %s: member %zu not taken (status %u, consumed %llu of %llu - 8, produced %llu of %llu)\n", paths[f0 + who[k]], k, res[k].status,
                                (unsigned long long)res[k].in_used, (unsigned long long)st[k].in_len, (unsigned long long)res[k].out_len, (unsigned long long)st[k].out_cap);
                    bad[who[k]] = 1;
                }
            }
            std::vector<uint32_t> crc(st.size());
            if ((rc2 = gs::crc32_texts_dev(c, dtext[b].p, toff.data(), tlen.data(), (uint32_t)st.size(), crc.data()))) return rc2;
            sub(1);
            for (size_t k = 0; k < st.size(); k++) {
                const uint8_t *t = itrail[b][k];
                const uint32_t want = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                if (tlen[k] && crc[k] != want) {
                    if (!bad[who[k]] && getenv("GS_INGEST_VERBOSE")) fprintf(stderr, "[GS_INGEST] %s: member %zu CRC-32 %08x, trailer says %08x\n", paths[f0 + who[k]], k, crc[k], want);
                    bad[who[k]] = 1;
                }
            }
            std::vector<uint64_t> files, foff, flen;                 // the device's files of this group, their texts (a bgzip file: its members end to end)
            for (uint64_t f = 0; f < nf; f++) {
                const auto &fb = blobs[g][f];
                if (!fb.gz_dev) continue;
                files.push_back(f); foff.push_back(S.gbase + fb.g_off); flen.push_back(bad[f] ? 0 : fb.g_cap);
                if (bad[f] && redo) redo->push_back(f0 + f);
            }
            std::vector<std::vector<uint64_t>> gsb, gse;
            if ((rc2 = gs::fasta_scan_dev(c, dtext[b].p, foff.data(), flen.data(), (uint32_t)files.size(), gsb, gse))) return rc2;
            for (size_t k = 0; k < files.size(); k++) { S.fsb[files[k]] = std::move(gsb[k]); S.fse[files[k]] = std::move(gse[k]); }
            sub(2);
        }
        S.sb.clear(); S.se.clear(); S.frec.assign(1, 0);
        for (uint64_t f = 0; f < nf; f++) {
            S.sb.insert(S.sb.end(), S.fsb[f].begin(), S.fsb[f].end()); S.se.insert(S.se.end(), S.fse[f].begin(), S.fse[f].end());
            S.frec.push_back(S.sb.size());
        }
        const uint64_t nrec = S.sb.size();
        const size_t out_bytes = (aa ? S.bytes : S.bytes / 4) + 8 * (nrec + nf) + 128;
        if ((rc2 = dout.ensure(out_bytes))) return rc2;
        GS_HIP_CHECK(hipMemsetAsync(dout.p, 0, out_bytes, c->stream));
        std::vector<uint64_t> rs(std::max<uint64_t>(nrec, 1)), rl(std::max<uint64_t>(nrec, 1)), goff(nf + 1), rs2, rl2;
        if (!block_mode) {
            if ((rc2 = gs::ingest_records_dev(c, aa, false, dtext[b].p, S.bytes, S.sb.data(), S.se.data(), nrec, dout.p, 0, rs.data(), rl.data(), nullptr))) return rc2;
            for (uint64_t f = 0; f <= nf; f++) goff[f] = S.frec[f];
            rs2.assign(rs.begin(), rs.begin() + nrec); rl2.assign(rl.begin(), rl.begin() + nrec);
        } else {   // one record per file: its records concatenated without gaps
            uint64_t pos = 0;
            for (uint64_t f = 0; f < nf; f++) {
                const uint64_t r0 = S.frec[f], r1 = S.frec[f + 1];
                if (!aa) pos = (pos + 31) / 32 * 32;
                uint64_t end = pos;
                if (r1 > r0 && (rc2 = gs::ingest_records_dev(c, aa, true, dtext[b].p, S.bytes, S.sb.data() + r0, S.se.data() + r0, r1 - r0, dout.p, pos, rs.data() + r0, rl.data() + r0, &end))) return rc2;
                rs2.push_back(pos); rl2.push_back(end - pos); goff[f] = f;
                pos = end;
            }
            goff[nf] = nf;
        }
        const uint64_t nr2 = rs2.size();
        if ((rc2 = drs.ensure(8 * (nr2 + 1))) || (rc2 = drl.ensure(8 * (nr2 + 1))) || (rc2 = dgo.ensure(8 * (nf + 1))) || (rc2 = dsig.ensure(nf * m * esz))) return rc2;
        if (nr2) {
            GS_HIP_CHECK(hipMemcpyAsync(drs.p, rs2.data(), 8 * nr2, hipMemcpyHostToDevice, c->stream));
            GS_HIP_CHECK(hipMemcpyAsync(drl.p, rl2.data(), 8 * nr2, hipMemcpyHostToDevice, c->stream));
        }
        GS_HIP_CHECK(hipMemcpyAsync(dgo.p, goff.data(), 8 * (nf + 1), hipMemcpyHostToDevice, c->stream));
        if ((rc2 = gs_sketch_batch_dev(c, p, dout.p, out_bytes / 8 * 8, drs.as<uint64_t>(), drl.as<uint64_t>(), nr2, dgo.as<uint64_t>(), nf, dsig.p))) return rc2;
        if (out_index) {
            rows_tmp.resize(nf * m * esz);
            GS_HIP_CHECK(hipMemcpyAsync(rows_tmp.data(), dsig.p, nf * m * esz, hipMemcpyDeviceToHost, c->stream));
        } else GS_HIP_CHECK(hipMemcpyAsync((uint8_t *)sig_out + f0 * m * esz, dsig.p, nf * m * esz, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(gs::stream_wait(c));
        sub(3);
        for (uint64_t f = 0; f < nf; f++) {
            uint64_t sym = 0;
            if (!block_mode) for (uint64_t r = S.frec[f]; r < S.frec[f + 1]; r++) sym += rl[r]; else sym = rl2[f];
            const uint64_t row = out_index ? out_index[f0 + f] : f0 + f;
            if (out_index) memcpy((uint8_t *)sig_out + row * m * esz, rows_tmp.data() + f * m * esz, m * esz);
            if (n_records_out) n_records_out[row] = S.frec[f + 1] - S.frec[f];
            if (n_symbols_out) n_symbols_out[row] = sym;
        }
        dev_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (deal) { std::lock_guard<std::mutex> lk(deal->m); (from_back ? deal->done_back : deal->done_front) += claimed_gz[g]; }
        return GS_OK;
    };
    double tsec[5] = {0, 0, 0, 0, 0};      // GS_INGEST_TIMES=1: wall seconds in stage / inflate launch / start_group / device stage / cleanup
    auto tick = [&](int k, std::chrono::steady_clock::time_point t0) { tsec[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    for (uint64_t g = 0; g < n_groups_eff; g++) {
        auto t0 = std::chrono::steady_clock::now();
        if ((rc = stage(g))) GS_FILES_FAIL(rc);                    // H2D of group g starts ...
        tick(0, t0); t0 = std::chrono::steady_clock::now();
        if ((rc = inflate_launch(g))) GS_FILES_FAIL(rc);           // ... its .gz members inflate behind it ...
        tick(1, t0); t0 = std::chrono::steady_clock::now();
        if (g + LA < n_groups_eff) start_group(g + LA);            // ... the host threads move on to group g+LA ...
        tick(2, t0); t0 = std::chrono::steady_clock::now();
        if (g >= 1 && (rc = device_stage(g - 1))) GS_FILES_FAIL(rc);   // ... while the device packs and sketches group g-1
        tick(3, t0);
    }
    { auto t0 = std::chrono::steady_clock::now(); if (n_groups_eff && (rc = device_stage(n_groups_eff - 1))) GS_FILES_FAIL(rc); tick(3, t0); }
    { auto t0 = std::chrono::steady_clock::now(); cleanup(); tick(4, t0); }
    if (getenv("GS_INGEST_TIMES"))
        fprintf(stderr, "[GS_INGEST_TIMES] %s pipeline, %llu groups: stage %.3f s, inflate launch %.3f, start_group %.3f, device stage %.3f, cleanup %.3f\n", dev_gzip ? "device" : "host",
                (unsigned long long)n_groups_eff, tsec[0], tsec[1], tsec[2], tsec[3], tsec[4]),
        fprintf(stderr, "[GS_INGEST_TIMES]   device stage: copy wait %.3f s, inflate wait %.3f, crc %.3f, record scan %.3f, pack + sketch + results %.3f\n", copy_wait_s, dsub[0], dsub[1], dsub[2], dsub[3]);
#undef GS_FILES_FAIL
    if (stats_out) {
        stats_out[0] = read_s; stats_out[1] = copy_wait_s; stats_out[2] = dev_s;
        stats_out[3] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count();
    }
    return GS_OK;
}

static int sketch_files_all(gs_ctx *c, const gs_sketch_params *p, const char *const *paths, uint64_t n_files, int block_mode, uint32_t pio, uint32_t n_threads,
                            void *sig_out, uint64_t *n_records_out, uint64_t *n_symbols_out, double *stats_out /* 6 doubles or NULL */)
{
    GS_REQUIRE(c && (n_files == 0 || paths), GS_ERR_INVALID, "null argument");
    GS_CTX_LOCK(c);
    const auto t_call = std::chrono::steady_clock::now();
    const char *e = getenv("GS_GZIP_DEVICE");                 // 0: every .gz through the host decoders (libdeflate / zlib), as in round 2
    const bool dev_gzip = e ? atoi(e) != 0 : true;
    // Two pipelines side by side. The device inflates a thousand members per launch (one wave each) in ~0.4 s whatever their number,
    // with next to no host work; 16 host cores inflate ~2500 members/s through libdeflate and then only need the device for pack + sketch
    // (~10 % of its time). Each pipeline runs on its own context (stream, scratch, pinned buffers) from its own thread; the .gz files are
    // dealt as they go (GzDeal): the host pipeline - which also takes every other kind of file - claims them from the front of the list in
    // groups of 64, the device pipeline from the back in groups of 4 x CUs, until the two meet.
    std::vector<uint64_t> ia, ib;                             // the host pipeline's list / the device pipeline's list (file indices)
    GzDeal deal;
    uint64_t n_free = 0;
    if (dev_gzip) {
        for (uint64_t f = 0; f < n_files; f++) if (!gs::ends_with(paths[f], ".gz")) ia.push_back(f);
        n_free = ia.size();
        for (uint64_t f = n_files; f-- > 0;) if (gs::ends_with(paths[f], ".gz")) ib.push_back(f);
        deal.n_gz = ib.size();
        const char *only = getenv("GS_GZIP_DEVICE_ONLY");                       // measurement aid: 1 = every .gz to the device pipeline, the host pipeline idle
        if (deal.n_gz >= 64 && !(only && atoi(only))) ia.insert(ia.end(), ib.rbegin(), ib.rend());     // a handful of files: all of them to the device
    }
    const bool dealing = dev_gzip && deal.n_gz >= 64 && ia.size() > n_free;
    double st_a[4] = {0, 0, 0, 0}, st_b[4] = {0, 0, 0, 0};
    if (ib.empty()) {
        if (stats_out) stats_out[4] = stats_out[5] = 0.0;
        int rc = sketch_files_impl(c, p, paths, n_files, block_mode, pio, n_threads, sig_out, n_records_out, n_symbols_out, stats_out, false, nullptr, nullptr, 0, nullptr, false, 0);
        return rc;
    }
    std::vector<const char *> pa(ia.size()), pb(ib.size());
    for (size_t i = 0; i < ia.size(); i++) pa[i] = paths[ia[i]];
    for (size_t i = 0; i < ib.size(); i++) pb[i] = paths[ib[i]];
    int rc_a = GS_OK; std::string err_a;
    std::thread host_pipe;
    if (!ia.empty()) {
        if (!c->child) { int rc = gs_ctx_create(&c->child, c->device, nullptr); if (rc) return rc; }
        host_pipe = std::thread([&]() {
            rc_a = sketch_files_impl(c->child, p, pa.data(), pa.size(), block_mode, pio ? pio : 64, n_threads, sig_out, n_records_out, n_symbols_out, st_a, false, nullptr, ia.data(), 12, dealing ? &deal : nullptr, false, n_free);
            if (rc_a) err_a = gs_last_error();
        });
    }
    std::vector<uint64_t> redo;
    // the device pipeline's host side only reads files: a few threads are plenty, the cores belong to the host pipeline's decoders
    int rc = sketch_files_impl(c, p, pb.data(), pb.size(), block_mode, pio, ia.empty() ? n_threads : std::max(2u, std::min(8u, n_threads ? n_threads : 8u)), sig_out, n_records_out,
                               n_symbols_out, st_b, true, &redo, ib.data(), 1, dealing ? &deal : nullptr, true, 0);
    if (host_pipe.joinable()) host_pipe.join();
    if (rc) return rc;
    if (rc_a) { gs::set_error("%s", err_a.c_str()); return rc_a; }
    if (!redo.empty()) {
        // members the device path handed back (multi-member files, a trailer that does not check): the host decoders, results to their rows
        std::vector<const char *> rp(redo.size()); std::vector<uint64_t> ri(redo.size());
        for (size_t i = 0; i < redo.size(); i++) { ri[i] = ib[redo[i]]; rp[i] = paths[ri[i]]; }
        double st2[4] = {0, 0, 0, 0};
        if ((rc = sketch_files_impl(c, p, rp.data(), rp.size(), block_mode, 32, n_threads, sig_out, n_records_out, n_symbols_out, st2, false, nullptr, ri.data(), 0, nullptr, false, 0))) return rc;
        for (int i = 0; i < 3; i++) st_b[i] += st2[i];
    }
    if (stats_out) {
        for (int i = 0; i < 3; i++) stats_out[i] = st_a[i] + st_b[i];
        stats_out[3] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count();
        const uint64_t dev_members = dealing ? deal.back : ib.size();
        stats_out[4] = (double)(dev_members - std::min<uint64_t>(dev_members, redo.size()));      // .gz members inflated by the device kernel
        stats_out[5] = (double)redo.size();                // members it handed back to the host decoders (multi-member, trailer / CRC mismatch, no room)
    }
    return GS_OK;
}
// the round-1 contract: FOUR doubles (the array grew to six under the same symbol - a caller that passes double[4] was written out of bounds)
int gs_sketch_files(gs_ctx *c, const gs_sketch_params *p, const char *const *paths, uint64_t n_files, int block_mode, uint32_t pio, uint32_t n_threads,
                    void *sig_out, uint64_t *n_records_out, uint64_t *n_symbols_out, double *stats_out)
{
    return gs_sketch_files_ex(c, p, paths, n_files, block_mode, pio, n_threads, sig_out, n_records_out, n_symbols_out, stats_out, stats_out ? 4 : 0);
}
int gs_sketch_files_ex(gs_ctx *c, const gs_sketch_params *p, const char *const *paths, uint64_t n_files, int block_mode, uint32_t pio, uint32_t n_threads,
                       void *sig_out, uint64_t *n_records_out, uint64_t *n_symbols_out, double *stats_out, uint32_t stats_cap)
{
    GS_REQUIRE(stats_out || stats_cap == 0, GS_ERR_INVALID, "gs_sketch_files_ex: stats_cap without stats_out");
    double st[GS_SKETCH_FILES_STATS] = {0, 0, 0, 0, 0, 0};
    const int rc = sketch_files_all(c, p, paths, n_files, block_mode, pio, n_threads, sig_out, n_records_out, n_symbols_out, st);
    for (uint32_t i = 0; i < stats_cap && i < (uint32_t)GS_SKETCH_FILES_STATS; i++) stats_out[i] = st[i];
    return rc;
}

}  // extern "C"
