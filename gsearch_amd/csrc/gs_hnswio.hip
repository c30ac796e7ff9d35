// gs_hnswio.hip — reading and writing the dump of hnsw_rs (`Hnsw::file_dump` / `HnswIo::load_hnsw`), the on-disk form of a gsearch
// database: <dir>/hnswdump.hnsw.graph + <dir>/hnswdump.hnsw.data (/root/reference/src/utils/dumpload.rs:26-31,
// src/utils/reloadhnsw.rs:13-51 - file names and `load_description(..).t_name` are [REF]). SURVEY 8f row f3.
//
// The byte layout itself lives in the un-vendored crate hnsw_rs 0.3 (`hnswio.rs`, Cargo.toml:115) and is restated here from the
// crate's published source AS RECALLED - it could not be checked against the crate or against an upstream dump in this container.
// Every constant that is recall rather than [REF] is in `namespace fmt` below, so a maintainer holding the crate can align it in one
// place. All integers native endian (`to_ne_bytes`), `usize` = 8 bytes.
//
//   graph file:  Description | nb_layer:u8 | for each layer L = 0..nb_layer-1: MAGICLAYER:u32, nb_point:u64, that many Point records
//                | entry point: origin_id:u64, layer:u8, rank:i32
//   Description: MAGICDESCR_3:u32, dumpmode:u8 (1 = full), max_nb_connection:u8, nb_layer:u8 (= 16), ef:u64, nb_point:u64,
//                dimension:u64, len:u64 + distance type name, len:u64 + T type name ("f32", "u32", "u64", "u16")
//   Point:       MAGICPOINT:u32, origin_id:u64, layer:u8, rank_in_layer:i32, then for l = 0..15: count:u8 and count x
//                (origin_id:u64, layer:u8, rank:i32, distance:f32)   - a point is filed under ITS top layer, rank = position there
//   data file:   MAGICDATAP:u32, dimension:u64, then per point in graph order: MAGICDATAP:u32, origin_id:u64, nbytes:u64, raw T values
#include <math.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>
#include "gs_internal.hpp"

namespace gs { namespace fmt {
constexpr uint32_t MAGICDESCR_3 = 0x002a6771u;      // [RECALL] description, format 3 (raw data vectors, mmap-able)
constexpr uint32_t MAGICDESCR_2 = 0x002a677fu;      // [RECALL] format 2 (bincode-encoded vectors): recognised and refused
constexpr uint32_t MAGICPOINT = 0x000a678fu;        // [RECALL]
constexpr uint32_t MAGICLAYER = 0x000a676fu;        // [RECALL]
constexpr uint32_t MAGICDATAP = 0xa67f0000u;        // [RECALL]
constexpr uint8_t NB_LAYER_MAX = 16;                // [REF] Hnsw::new(.., 16, ..) dnasketch.rs:139
static const char *DISTNAME = "anndists::dist::distances::DistHamming";   // [RECALL] std::any::type_name of the distance
}  // namespace fmt

struct Wr {
    FILE *f; bool ok = true;
    template <class T> void put(T v) { ok = ok && fwrite(&v, sizeof(T), 1, f) == 1; }
    void bytes(const void *p, size_t n) { ok = ok && (n == 0 || fwrite(p, 1, n, f) == n); }
    void str(const char *s) { const uint64_t n = strlen(s); put<uint64_t>(n); bytes(s, n); }
};
struct Rd {
    FILE *f; bool ok = true;
    template <class T> T get() { T v{}; ok = ok && fread(&v, sizeof(T), 1, f) == 1; return v; }
    void bytes(void *p, size_t n) { ok = ok && (n == 0 || fread(p, 1, n, f) == n); }
    std::string str() { const uint64_t n = get<uint64_t>(); if (!ok || n > 4096) { ok = false; return std::string(); } std::string s(n, '\0'); bytes(&s[0], n); return s; }
};
static const char *tname_of(int kind) { return kind == GS_KIND_F32 ? "f32" : kind == GS_KIND_U32 ? "u32" : kind == GS_KIND_U64 ? "u64" : "u16"; }
static int kind_of_tname(const std::string &t) { return t == "f32" ? GS_KIND_F32 : t == "u32" ? GS_KIND_U32 : t == "u64" ? GS_KIND_U64 : t == "u16" ? GS_KIND_U16 : -1; }
}  // namespace gs

extern "C" {

/* Hnsw::file_dump(dir, "hnswdump") (dumpload.rs:26-31): writes <basename>.hnsw.graph and <basename>.hnsw.data in hnsw_rs' format 3
 * (layout above; recalled, see the file header). max_layer must be 16 and no neighbour list may exceed 255 entries (one-byte count). */
int gs_index_dump_hnswrs(gs_index *ix, const char *basename) { return gs_index_dump_hnswrs_ex(ix, basename, 0); }

/* flags: GS_DUMP_TRUNCATE_255 - the format stores every neighbour count in ONE byte (`len() as u8` upstream), while layer 0 holds up to
 * 2 * max_nb_connection = 256 .. 510 entries: a full list cannot be written faithfully (upstream, 256 would wrap to 0 and desynchronise the
 * reader). Without the flag such an index is refused (use gs_index_save, which is lossless); with it, lists longer than 255 are cut to their
 * 255 closest entries - a file upstream can read, at the price that the reloaded graph lacks the farthest link(s) of those nodes. */
int gs_index_dump_hnswrs_ex(gs_index *ix, const char *basename, uint32_t flags)
{
    GS_REQUIRE(ix && basename, GS_ERR_INVALID, "null argument");
    GS_REQUIRE((flags & ~(uint32_t)GS_DUMP_TRUNCATE_255) == 0, GS_ERR_INVALID, "unknown dump flags %x", flags);
    gs_index_params prm;
    int rc = gs_index_get_params(ix, &prm); if (rc) return rc;
    const uint64_t n = gs_index_nb_point(ix);
    GS_REQUIRE(n > 0, GS_ERR_STATE, "nothing to dump");
    GS_REQUIRE(prm.max_layer == gs::fmt::NB_LAYER_MAX, GS_ERR_UNSUPPORTED, "hnsw_rs dumps always carry %d layers", (int)gs::fmt::NB_LAYER_MAX);
    const uint32_t M = prm.max_nb_conn, ML = prm.max_layer, m = prm.m;
    const size_t esz = gs::kind_bytes(prm.kind);
    std::vector<uint8_t> lv(n); std::vector<uint32_t> d0(n), n0(n * 2 * M), c0(n * 2 * M); std::vector<int32_t> up(n);
    int64_t entry = -1; uint64_t U = 0;
    if ((rc = gs_index_export(ix, nullptr, &entry, nullptr, nullptr, nullptr, nullptr, &U, nullptr, nullptr, nullptr))) return rc;
    std::vector<uint32_t> dU(std::max<uint64_t>(U, 1) * ML), nU(std::max<uint64_t>(U, 1) * ML * M), cU(std::max<uint64_t>(U, 1) * ML * M);
    if ((rc = gs_index_export(ix, lv.data(), &entry, d0.data(), n0.data(), c0.data(), up.data(), &U, dU.data(), nU.data(), cU.data()))) return rc;
    std::vector<uint64_t> oid(n);
    if ((rc = gs_index_get_ids(ix, 0, n, oid.data()))) return rc;
    // PointId = (top layer, rank among the points of that layer in id order)
    std::vector<int32_t> rank(n); std::vector<std::vector<uint32_t>> by_layer(ML);
    for (uint64_t i = 0; i < n; i++) { rank[i] = (int32_t)by_layer[lv[i]].size(); by_layer[lv[i]].push_back((uint32_t)i); }
    if (flags & GS_DUMP_TRUNCATE_255) {
        // keep the 255 CLOSEST: lists built here are (count, id)-ascending, lists that came in through gs_index_import / a loaded dump need not be
        std::vector<uint64_t> keys;
        for (uint64_t i = 0; i < n; i++) if (d0[i] > 255) {
            keys.resize(d0[i]);
            for (uint32_t t = 0; t < d0[i]; t++) keys[t] = ((uint64_t)c0[i * 2 * M + t] << 32) | n0[i * 2 * M + t];
            std::sort(keys.begin(), keys.end());
            for (uint32_t t = 0; t < 255; t++) { n0[i * 2 * M + t] = (uint32_t)keys[t]; c0[i * 2 * M + t] = (uint32_t)(keys[t] >> 32); }
            d0[i] = 255;
        }
    }
    else for (uint64_t i = 0; i < n; i++)
        GS_REQUIRE(d0[i] <= 255, GS_ERR_UNSUPPORTED, "node %llu has %u layer-0 neighbours: hnsw_rs stores the count in one byte (dump with GS_DUMP_TRUNCATE_255, or use gs_index_save)", (unsigned long long)i, d0[i]);
    const std::string gname = std::string(basename) + ".hnsw.graph", dname = std::string(basename) + ".hnsw.data";
    FILE *fg = fopen(gname.c_str(), "wb"), *fd = fopen(dname.c_str(), "wb");
    if (!fg || !fd) { if (fg) fclose(fg); if (fd) fclose(fd); gs::set_error("cannot open %s / %s for writing", gname.c_str(), dname.c_str()); return GS_ERR_IO; }
    gs::Wr g{fg}, d{fd};
    g.put<uint32_t>(gs::fmt::MAGICDESCR_3); g.put<uint8_t>(1); g.put<uint8_t>((uint8_t)M); g.put<uint8_t>(gs::fmt::NB_LAYER_MAX);
    g.put<uint64_t>(prm.ef_construction); g.put<uint64_t>(n); g.put<uint64_t>(m); g.str(gs::fmt::DISTNAME); g.str(gs::tname_of(prm.kind));
    d.put<uint32_t>(gs::fmt::MAGICDATAP); d.put<uint64_t>(m);
    g.put<uint8_t>(gs::fmt::NB_LAYER_MAX);
    const uint64_t CH = 2048;
    std::vector<uint8_t> rows(CH * esz * m);
    const float fm = (float)m;
    for (uint32_t L = 0; L < ML && g.ok && d.ok; L++) {
        g.put<uint32_t>(gs::fmt::MAGICLAYER); g.put<uint64_t>(by_layer[L].size());
        for (size_t j0 = 0; j0 < by_layer[L].size() && g.ok && d.ok; j0 += CH) {
            const size_t j1 = std::min(by_layer[L].size(), j0 + CH);
            // data rows of this stretch of points (ids ascend inside a layer, usually contiguously: fetch the covering range)
            const uint64_t lo = by_layer[L][j0], hi = by_layer[L][j1 - 1];
            const bool contiguous = hi - lo + 1 == j1 - j0;
            if (contiguous) { if ((rc = gs_index_get_data(ix, lo, hi - lo + 1, rows.data()))) { fclose(fg); fclose(fd); return rc; } }
            for (size_t j = j0; j < j1; j++) {
                const uint32_t id = by_layer[L][j];
                if (!contiguous && (rc = gs_index_get_data(ix, id, 1, rows.data() + (j - j0) * esz * m))) { fclose(fg); fclose(fd); return rc; }
                g.put<uint32_t>(gs::fmt::MAGICPOINT); g.put<uint64_t>(oid[id]); g.put<uint8_t>((uint8_t)L); g.put<int32_t>((int32_t)j);
                for (uint32_t l = 0; l < gs::fmt::NB_LAYER_MAX; l++) {
                    uint32_t deg = 0; const uint32_t *nb = nullptr, *cn = nullptr;
                    if (l == 0) { deg = d0[id]; nb = &n0[(uint64_t)id * 2 * M]; cn = &c0[(uint64_t)id * 2 * M]; }
                    else if (l <= L) { const uint64_t u = (uint64_t)up[id]; deg = dU[u * ML + (l - 1)]; nb = &nU[(u * ML + (l - 1)) * M]; cn = &cU[(u * ML + (l - 1)) * M]; }
                    g.put<uint8_t>((uint8_t)deg);
                    for (uint32_t t = 0; t < deg; t++) { g.put<uint64_t>(oid[nb[t]]); g.put<uint8_t>(lv[nb[t]]); g.put<int32_t>(rank[nb[t]]); g.put<float>((float)cn[t] / fm); }
                }
                d.put<uint32_t>(gs::fmt::MAGICDATAP); d.put<uint64_t>(oid[id]); d.put<uint64_t>(esz * m); d.bytes(rows.data() + (j - j0) * esz * m, esz * m);
            }
        }
    }
    g.put<uint64_t>(oid[entry]); g.put<uint8_t>(lv[entry]); g.put<int32_t>(rank[entry]);
    const bool ok = g.ok && d.ok;
    fclose(fg); fclose(fd);
    GS_REQUIRE(ok, GS_ERR_IO, "short write to %s", gname.c_str());
    return GS_OK;
}

/* HnswIo::load_hnsw (reloadhnsw.rs:41-51): reads <basename>.hnsw.graph / .hnsw.data into a new index on ctx. The dump does not hold
 * what Hnsw::new was given beyond max_nb_connection and ef: `hint` (optional) supplies capacity, scale_modify, extend_candidates,
 * keep_pruned, seed and insert_batch for later insertions (gsearch's `add` re-reads them from parameters.json, gsearch.rs:717-741).
 * DataIds 0..n-1 (gsearch's dictionary ranks, dnasketch.rs:429-433) become the node numbers; any other distinct ids are kept as caller ids. */
int gs_index_load_hnswrs(gs_ctx *c, const char *basename, const gs_index_params *hint, gs_index **out)
{
    GS_REQUIRE(c && basename && out, GS_ERR_INVALID, "null argument");
    const std::string gname = std::string(basename) + ".hnsw.graph", dname = std::string(basename) + ".hnsw.data";
    FILE *fg = fopen(gname.c_str(), "rb"), *fd = fopen(dname.c_str(), "rb");
    if (!fg || !fd) { if (fg) fclose(fg); if (fd) fclose(fd); gs::set_error("cannot open %s / %s", gname.c_str(), dname.c_str()); return GS_ERR_IO; }
    gs::Rd g{fg}, d{fd};
    struct Closer { FILE *a, *b; ~Closer() { fclose(a); fclose(b); } } closer{fg, fd};
    const uint32_t magic = g.get<uint32_t>();
    GS_REQUIRE(magic != gs::fmt::MAGICDESCR_2, GS_ERR_UNSUPPORTED, "%s is a format-2 dump (bincode vectors); re-dump it with hnsw_rs >= 0.2", gname.c_str());
    GS_REQUIRE(g.ok && magic == gs::fmt::MAGICDESCR_3, GS_ERR_IO, "%s is not an hnsw_rs graph dump (magic %08x)", gname.c_str(), magic);
    const uint8_t dumpmode = g.get<uint8_t>(), M8 = g.get<uint8_t>(), nbl = g.get<uint8_t>();
    const uint64_t ef = g.get<uint64_t>(), n = g.get<uint64_t>(), dim = g.get<uint64_t>();
    const std::string distname = g.str(), tname = g.str();
    GS_REQUIRE(g.ok, GS_ERR_IO, "truncated description in %s", gname.c_str());
    GS_REQUIRE(dumpmode == 1, GS_ERR_UNSUPPORTED, "light dumps (graph without data) cannot be searched");
    GS_REQUIRE(nbl == gs::fmt::NB_LAYER_MAX, GS_ERR_IO, "description says %u layers, hnsw_rs always dumps 16", nbl);
    GS_REQUIRE(distname.find("DistHamming") != std::string::npos, GS_ERR_UNSUPPORTED, "distance %s: gsearch databases use DistHamming", distname.c_str());
    const int kind = gs::kind_of_tname(tname);
    GS_REQUIRE(kind >= 0, GS_ERR_UNSUPPORTED, "element type %s is not a signature type of gsearch", tname.c_str());
    GS_REQUIRE(M8 >= 2 && n >= 1 && n < ((uint64_t)1 << 31) && dim >= 1 && dim < ((uint64_t)1 << 31), GS_ERR_IO, "implausible description (M %u, n %llu, dim %llu)", M8, (unsigned long long)n, (unsigned long long)dim);
    const uint32_t M = M8, ML = gs::fmt::NB_LAYER_MAX, m = (uint32_t)dim;
    const size_t esz = gs::kind_bytes(kind);
    // data file
    GS_REQUIRE(d.get<uint32_t>() == gs::fmt::MAGICDATAP && d.get<uint64_t>() == dim && d.ok, GS_ERR_IO, "%s: bad header", dname.c_str());
    // DataIds: gsearch's are the dictionary ranks 0..n-1 (dnasketch.rs:429-433) and then ARE the node numbers of this library; any other set of
    // distinct ids is kept as the caller's ids (gs_index_set_ids) over nodes numbered in the order of the data file
    std::vector<uint8_t> sigs((size_t)n * esz * m);
    std::vector<uint64_t> oid(n);
    const long data_pos = ftell(fd);
    bool dense = true;
    {
        std::vector<uint8_t> have(n, 0);
        for (uint64_t i = 0; i < n; i++) {
            const uint32_t mg = d.get<uint32_t>(); const uint64_t id = d.get<uint64_t>(), nb = d.get<uint64_t>();
            GS_REQUIRE(d.ok && mg == gs::fmt::MAGICDATAP && nb == esz * m, GS_ERR_IO, "%s: bad vector record %llu", dname.c_str(), (unsigned long long)i);
            oid[i] = id;
            if (id >= n || have[id]) dense = false; else have[id] = 1;
            GS_REQUIRE(fseek(fd, (long)(esz * m), SEEK_CUR) == 0, GS_ERR_IO, "%s is truncated", dname.c_str());
        }
    }
    std::unordered_map<uint64_t, uint64_t> node_of;
    if (!dense) {
        node_of.reserve(n * 2);
        for (uint64_t i = 0; i < n; i++) GS_REQUIRE(node_of.emplace(oid[i], i).second, GS_ERR_IO, "%s: id %llu appears twice", dname.c_str(), (unsigned long long)oid[i]);
    }
    auto node = [&](uint64_t id, bool *ok) -> uint64_t {          // DataId -> node number
        if (dense) { if (id >= n) *ok = false; return id; }
        auto it = node_of.find(id);
        if (it == node_of.end()) { *ok = false; return 0; }
        return it->second;
    };
    GS_REQUIRE(fseek(fd, data_pos, SEEK_SET) == 0, GS_ERR_IO, "%s: seek", dname.c_str());
    for (uint64_t i = 0; i < n; i++) {
        (void)d.get<uint32_t>(); const uint64_t id = d.get<uint64_t>(); (void)d.get<uint64_t>();
        bool ok = true;
        d.bytes(sigs.data() + node(id, &ok) * esz * m, esz * m);
        GS_REQUIRE(d.ok && ok, GS_ERR_IO, "%s is truncated", dname.c_str());
    }
    // graph
    GS_REQUIRE(g.get<uint8_t>() == gs::fmt::NB_LAYER_MAX && g.ok, GS_ERR_IO, "%s: bad layer count", gname.c_str());
    std::vector<uint8_t> lv(n, 0xFF); std::vector<uint32_t> d0(n, 0), n0((size_t)n * 2 * M, 0), c0((size_t)n * 2 * M, 0); std::vector<int32_t> up(n, -1);
    struct UpperList { std::vector<std::pair<uint64_t, uint32_t>> e; };          // (key = count<<32 | id) lists of one node, layers 1..
    std::vector<std::vector<UpperList>> upper;                                    // by upper index
    uint64_t seen = 0;
    const float fm = (float)m;
    for (uint32_t L = 0; L < ML; L++) {
        GS_REQUIRE(g.get<uint32_t>() == gs::fmt::MAGICLAYER && g.ok, GS_ERR_IO, "%s: layer %u header missing", gname.c_str(), L);
        const uint64_t np = g.get<uint64_t>();
        GS_REQUIRE(g.ok && seen + np <= n, GS_ERR_IO, "%s: layer %u claims %llu points", gname.c_str(), L, (unsigned long long)np);
        for (uint64_t j = 0; j < np; j++) {
            bool idok = true;
            const uint32_t mg = g.get<uint32_t>(); const uint64_t id = node(g.get<uint64_t>(), &idok); const uint8_t pl = g.get<uint8_t>(); (void)g.get<int32_t>();
            GS_REQUIRE(g.ok && idok && mg == gs::fmt::MAGICPOINT && id < n && pl == L && lv[id] == 0xFF, GS_ERR_IO, "%s: bad point record (layer %u, rank %llu)", gname.c_str(), L, (unsigned long long)j);
            lv[id] = (uint8_t)L;
            if (L > 0) { up[id] = (int32_t)upper.size(); upper.emplace_back(ML); }
            for (uint32_t l = 0; l < ML; l++) {
                const uint8_t cnt = g.get<uint8_t>();
                std::vector<uint64_t> keys(cnt);
                for (uint32_t t = 0; t < cnt; t++) {
                    bool nok = true;
                    const uint64_t nid = node(g.get<uint64_t>(), &nok); (void)g.get<uint8_t>(); (void)g.get<int32_t>(); const float dist = g.get<float>();
                    GS_REQUIRE(g.ok && nok && nid < n && dist >= 0.0f && dist <= 1.0f, GS_ERR_IO, "%s: bad neighbour of point %llu", gname.c_str(), (unsigned long long)id);
                    keys[t] = ((uint64_t)(uint32_t)llrintf(dist * fm) << 32) | nid;
                }
                GS_REQUIRE(g.ok, GS_ERR_IO, "%s is truncated", gname.c_str());
                GS_REQUIRE(cnt == 0 || l <= L, GS_ERR_IO, "%s: point %llu of layer %u has neighbours at layer %u", gname.c_str(), (unsigned long long)id, L, l);
                std::sort(keys.begin(), keys.end());                  // this library keeps lists in (count, id) order
                if (l == 0) {
                    GS_REQUIRE(cnt <= 2 * M, GS_ERR_IO, "%s: point %llu has %u layer-0 neighbours, more than 2M", gname.c_str(), (unsigned long long)id, cnt);
                    d0[id] = cnt;
                    for (uint32_t t = 0; t < cnt; t++) { n0[(uint64_t)id * 2 * M + t] = (uint32_t)keys[t]; c0[(uint64_t)id * 2 * M + t] = (uint32_t)(keys[t] >> 32); }
                } else if (cnt) {
                    GS_REQUIRE(cnt <= M, GS_ERR_IO, "%s: point %llu has %u neighbours at layer %u, more than M", gname.c_str(), (unsigned long long)id, cnt, l);
                    for (uint32_t t = 0; t < cnt; t++) upper.back()[l - 1].e.emplace_back(keys[t], 0);
                }
            }
        }
        seen += np;
    }
    GS_REQUIRE(seen == n, GS_ERR_IO, "%s holds %llu points, its description says %llu", gname.c_str(), (unsigned long long)seen, (unsigned long long)n);
    bool eok = true;
    const uint64_t entry = node(g.get<uint64_t>(), &eok);
    GS_REQUIRE(g.ok && eok && entry < n, GS_ERR_IO, "%s: entry point missing", gname.c_str());
    const uint64_t U = upper.size();
    std::vector<uint32_t> dU(std::max<uint64_t>(U, 1) * ML, 0), nU(std::max<uint64_t>(U, 1) * ML * M, 0), cU(std::max<uint64_t>(U, 1) * ML * M, 0);
    for (uint64_t u = 0; u < U; u++)
        for (uint32_t l = 0; l + 1 < ML; l++) {
            const auto &e = upper[u][l].e;
            dU[u * ML + l] = (uint32_t)e.size();
            for (size_t t = 0; t < e.size(); t++) { nU[(u * ML + l) * M + t] = (uint32_t)e[t].first; cU[(u * ML + l) * M + t] = (uint32_t)(e[t].first >> 32); }
        }
    gs_index_params prm;
    if (hint) prm = *hint; else { memset(&prm, 0, sizeof prm); prm.capacity = 1500000; prm.scale_modify = 1.0; prm.extend_candidates = 1; }
    prm.kind = kind; prm.m = m; prm.max_nb_conn = M; prm.max_layer = ML; prm.ef_construction = (uint32_t)std::max<uint64_t>(1, ef);
    gs_index *ix = nullptr;
    int rc = gs_index_create(c, &prm, &ix); if (rc) return rc;
    rc = gs_index_import(ix, sigs.data(), n, lv.data(), (int64_t)entry, d0.data(), n0.data(), c0.data(), up.data(), U, U ? dU.data() : nullptr, U ? nU.data() : nullptr, U ? cU.data() : nullptr);
    if (rc == GS_OK && !dense) rc = gs_index_set_ids(ix, oid.data(), n);
    if (rc) { gs_index_destroy(ix); return rc; }
    *out = ix;
    return GS_OK;
}

}  // extern "C"
