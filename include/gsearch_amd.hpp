// gsearch_amd.hpp — C++17 host-side mirror of the reference's operator interface for the sketch-and-query hot path,
// header-only over the C ABI (gsearch_amd.h). The reference is Rust; its extension points for this path are the traits
//   kmerutils::sketching::setsketchert::SeqSketcherT   (src/dna/dnasketch.rs:64-73, calls :336,357; dnarequest.rs:272,287)
//   anndists::dist::Distance / DistHamming             (src/dna/dnasketch.rs:72,139; src/bin/bindash.rs:93-99)
//   hnsw_rs::Hnsw                                      (src/dna/dnasketch.rs:139-141,159-160,435; src/dna/dnarequest.rs:353)
// The classes below keep their names, argument meaning and error behaviour (the reference panics / exits on internal
// failure, dnasketch.rs:228,285,380: here a gsearch::Error is thrown with the library's message).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <ostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#include "gsearch_amd.h"

namespace gsearch {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("gsearch_amd error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) { if (rc != GS_OK) throw Error(rc, gs_last_error()); }

class Context {   // one per (process, GPU)
public:
    explicit Context(int device = 0, void *stream = nullptr) { check(gs_ctx_create(&h_, device, stream)); }
    ~Context() { gs_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    gs_ctx *get() const { return h_; }
    void sync() { check(gs_ctx_sync(h_)); }
    void release_scratch() { check(gs_ctx_release_scratch(h_)); }   // device scratch kept between calls
private:
    gs_ctx *h_ = nullptr;
};

enum class SketchAlgo : uint32_t { PROB3A = GS_ALGO_PROB3A, SUPER = GS_ALGO_SUPER, SUPER2 = GS_ALGO_SUPER2, HLL = GS_ALGO_HLL, OPTDENS = GS_ALGO_OPTDENS, REVOPTDENS = GS_ALGO_REVOPTDENS };
enum class DataType : uint32_t { DNA = GS_DATA_DNA, AA = GS_DATA_AA, DNA_FWD = GS_DATA_DNA_FWD /* bindash.rs:346-354: k <= 14, no reverse-complement minimum */ };

// kmerutils::sketcharg::SeqSketcherParams::new(kmer_size, sketch_size, algo, data_t)  (src/bin/gsearch.rs:258-263)
class SeqSketcherParams {
public:
    SeqSketcherParams(uint32_t kmer_size, uint32_t sketch_size, SketchAlgo algo, DataType data_t = DataType::DNA)
        : p_{kmer_size, sketch_size, (uint32_t)algo, (uint32_t)data_t} { check(gs_check_params(&p_)); }
    uint32_t get_kmer_size() const { return p_.k; }
    uint32_t get_sketch_size() const { return p_.sketch_size; }
    SketchAlgo get_algo() const { return (SketchAlgo)p_.algo; }
    int sig_kind() const { return gs_sig_kind(&p_); }
    const gs_sketch_params *raw() const { return &p_; }
private:
    gs_sketch_params p_;
};

template <class T> constexpr int kind_of()
{
    if (std::is_same<T, float>::value) return GS_KIND_F32;
    if (std::is_same<T, uint32_t>::value) return GS_KIND_U32;
    if (std::is_same<T, uint64_t>::value) return GS_KIND_U64;
    return GS_KIND_U16;
}

// A `Sequence` of the reference is a 2-bit packed record; here a record is ASCII (the glue packs like encode_and_add, dnafiles.rs:70-71)
using Record = std::string;

// SeqSketcherT<Kmer>: `Sig` is the signature element type of the (algo, k) dispatch (dnasketch.rs:499-642)
template <class Sig>
class SeqSketcher {
public:
    SeqSketcher(Context &ctx, const SeqSketcherParams &params) : ctx_(ctx), params_(params)
    {
        if (params_.sig_kind() != kind_of<Sig>()) throw Error(GS_ERR_INVALID, "Sig type does not match the (algo, k) dispatch of the reference");
    }
    // all sequences are ONE genome -> exactly one signature (assert at dnasketch.rs:359)
    std::vector<std::vector<Sig>> sketch_compressedkmer_seqs(const std::vector<const Record *> &vseq) const { return run(vseq, true); }
    // one signature per input sequence, in input order (assert at dnasketch.rs:338)
    std::vector<std::vector<Sig>> sketch_compressedkmer(const std::vector<const Record *> &vseq) const { return run(vseq, false); }
    // batch form: genomes = lists of records -> row-major (n_genomes x sketch_size)
    std::vector<Sig> sketch_genomes(const std::vector<std::vector<Record>> &genomes) const
    {
        std::vector<const Record *> recs; std::vector<uint64_t> goff{0};
        for (auto &g : genomes) { for (auto &r : g) recs.push_back(&r); goff.push_back(recs.size()); }
        return sketch_flat(recs, goff);
    }
private:
    std::vector<Sig> sketch_flat(const std::vector<const Record *> &recs, const std::vector<uint64_t> &goff) const
    {
        const bool aa = params_.raw()->data_t == GS_DATA_AA;
        std::vector<uint64_t> rs(recs.size()), rl(recs.size());
        uint64_t total = 0;
        for (auto *r : recs) total += r->size() + 4;
        std::vector<uint8_t> seq(aa ? total + 64 : total / 4 + 64, 0);
        uint64_t off = 0;
        for (size_t i = 0; i < recs.size(); i++) {
            const uint8_t *a = (const uint8_t *)recs[i]->data();
            rs[i] = off;
            if (aa) { rl[i] = gs_filter_aa(a, recs[i]->size(), seq.data() + off); off += rl[i]; }
            else { rl[i] = gs_pack_dna(a, recs[i]->size(), seq.data(), off); off += (rl[i] + 3) / 4 * 4; }
        }
        const uint64_t ng = goff.size() - 1, m = params_.get_sketch_size();
        std::vector<Sig> out(ng * m);
        check(gs_sketch_batch(ctx_.get(), params_.raw(), seq.data(), seq.size(), rs.data(), rl.data(), recs.size(), goff.data(), ng, out.data()));
        return out;
    }
    std::vector<std::vector<Sig>> run(const std::vector<const Record *> &vseq, bool one_genome) const
    {
        std::vector<uint64_t> goff;
        if (one_genome) goff = {0, vseq.size()};
        else for (uint64_t i = 0; i <= vseq.size(); i++) goff.push_back(i);
        std::vector<Sig> flat = sketch_flat(vseq, goff);
        const uint64_t m = params_.get_sketch_size();
        std::vector<std::vector<Sig>> out(goff.size() - 1);
        for (size_t g = 0; g < out.size(); g++) out[g].assign(flat.begin() + g * m, flat.begin() + (g + 1) * m);
        return out;
    }
    Context &ctx_;
    SeqSketcherParams params_;
};
// the names gsearch instantiates (dnasketch.rs:499-642)
template <class Sig = float> using OptDensHashSketch = SeqSketcher<Sig>;
template <class Sig = float> using RevOptDensHashSketch = SeqSketcher<Sig>;
template <class Sig = float> using SuperHashSketch = SeqSketcher<Sig>;
template <class Sig> using SuperHash2Sketch = SeqSketcher<Sig>;
template <class Sig> using ProbHash3aSketch = SeqSketcher<Sig>;

// anndists::dist::DistHamming: eval(a, b) = count(a[i] != b[i]) / len as f32
class DistHamming {
public:
    explicit DistHamming(Context &ctx) : ctx_(&ctx) {}
    template <class T> float eval(const std::vector<T> &va, const std::vector<T> &vb) const
    {
        if (va.size() != vb.size()) throw Error(GS_ERR_INVALID, "signature lengths differ");
        float d = 0;
        check(gs_hamming_qxc(ctx_->get(), kind_of<T>(), (uint32_t)va.size(), va.data(), 1, vb.data(), 1, &d));
        return d;
    }
    // Q (nq x m) against C (nc x m), row-major: the all-pairs loop of bindash.rs:120-157 in one call
    template <class T> std::vector<float> eval_qxc(const std::vector<T> &Q, uint64_t nq, const std::vector<T> &C, uint64_t nc, uint32_t m) const
    {
        std::vector<float> out(nq * nc);
        check(gs_hamming_qxc(ctx_->get(), kind_of<T>(), m, Q.data(), nq, C.data(), nc, out.data()));
        return out;
    }
    Context &context() const { return *ctx_; }
private:
    Context *ctx_;
};
// reformat.rs:80-86
inline double calculate_ani(double distance, int kmer, int model) { return gs_ani(distance, kmer, model); }

// hnsw_rs::Neighbour{d_id, distance, p_id}: gsearch reads d_id (the DataId the point was inserted under: an index into its seqdict) and distance
// (answer.rs:42,55-57); p_id = PointId(layer, rank in layer)
struct PointId { uint8_t layer; int32_t rank; };
struct Neighbour { size_t d_id; float distance; PointId p_id{0xFF, -1}; float get_distance() const { return distance; } };

// (path, fasta id, sequence length) of one database / request item: what ReqAnswer::dump reads of
// utils::idsketch::ItemDict via get_id().get_path(), get_id().get_fasta_id(), get_len() (answer.rs:48-50,56,68-69)
struct ItemDict { std::string path, fasta_id; size_t len; };
using SeqDict = std::vector<ItemDict>;

// Rust's {:.5E} on an f32 (answer.rs:60): exact decimal expansion of the value rounded half-even to 5 decimals,
// exponent without padding and without '+': 6.07500E-1, 0.00000E0
inline std::string rust_5E(float x)
{
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%.5E", (double)x);
    std::string s(buf);
    const size_t e = s.find('E');
    return s.substr(0, e) + "E" + std::to_string(std::stoi(s.substr(e + 1)));
}

// answer.rs:14-76 ReqAnswer: text record of one request. Only neighbours with distance < threshold are written; the
// header line is written when any neighbour has distance <= threshold (answer.rs:42 vs :55 - the two tests differ).
class ReqAnswer {
public:
    ReqAnswer(size_t rank, ItemDict req_item, const std::vector<Neighbour> &neighbours) : rank_(rank), req_item_(std::move(req_item)), neighbours_(neighbours) {}
    size_t dump(const SeqDict &seqdict, float threshold, std::ostream &out) const
    {
        bool has_match = false;
        for (auto &n : neighbours_) has_match |= n.distance <= threshold;
        size_t nb_match = 0;
        if (!has_match) return 0;
        out << "\n" << rank_ << "\t" << req_item_.path << "\tfasta_id:\t" << req_item_.fasta_id << "\tlength:\t" << req_item_.len;
        for (auto &n : neighbours_) {
            if (!(n.distance < threshold)) continue;
            nb_match++;
            const ItemDict &d = seqdict.at(n.d_id);
            out << "\nquery_id:\t" << req_item_.path << "\tdistance:\t" << rust_5E(n.distance) << "\tanswer_fasta_path\t" << d.path << "\t"
                << d.fasta_id << " \t answer_seq_len:\t " << d.len;
        }
        return nb_match;
    }
    const ItemDict &get_request_id() const { return req_item_; }
private:
    size_t rank_;
    ItemDict req_item_;
    const std::vector<Neighbour> &neighbours_;
};

// bindash.rs:93-99 compute_distance: j = 1 - d (f32); frac = 2j/(1+j) (f32); 1.0f64 - frac.powf(1/k as f32) as f64
inline double bindash_compute_distance(float hamming_distance, size_t kmer_size)
{
    const float j = 1.0f - hamming_distance;
    const float frac = 2.0f * j / (1.0f + j);
    return 1.0 - (double)std::pow(frac, 1.0f / (float)kmer_size);
}

// hnsw_rs::Hnsw<T, DistHamming>
template <class T>
class Hnsw {
public:
    // Hnsw::new(max_nb_connection, max_elements, max_layer, ef_construction, dist_f)  (dnasketch.rs:139)
    Hnsw(uint32_t max_nb_connection, uint64_t max_elements, uint32_t max_layer, uint32_t ef_construction, const DistHamming &dist_f, uint64_t seed = 0)
        : ctx_(&dist_f.context())
    {
        prm_ = gs_index_params{kind_of<T>(), 0, max_nb_connection, max_elements, max_layer, ef_construction, 1.0, 0, 0, seed, 0};
    }
    ~Hnsw() { gs_index_destroy(h_); }
    Hnsw(const Hnsw &) = delete;
    Hnsw &operator=(const Hnsw &) = delete;
    void modify_level_scale(double f) { frozen(); prm_.scale_modify = f; }             // dnasketch.rs:141
    void set_extend_candidates(bool b) { frozen(); prm_.extend_candidates = b; }       // dnasketch.rs:159
    void set_keeping_pruned(bool b) { frozen(); prm_.keep_pruned = b; }                // dnasketch.rs:160
    size_t get_nb_point() const { return h_ ? gs_index_nb_point(h_) : 0; }
    // parallel_insert(&[(&Vec<T>, usize)]) (dnasketch.rs:429-435): any DataIds; searches return them as d_id
    void parallel_insert(const std::vector<std::pair<const std::vector<T> *, size_t>> &datas)
    {
        if (datas.empty()) return;
        const size_t m = datas[0].first->size();
        ensure(m);
        std::vector<T> flat(datas.size() * m);
        std::vector<uint64_t> ids(datas.size());
        for (size_t i = 0; i < datas.size(); i++) {
            if (datas[i].first->size() != m) throw Error(GS_ERR_INVALID, "signature length mismatch");
            ids[i] = datas[i].second;
            std::copy(datas[i].first->begin(), datas[i].first->end(), flat.begin() + i * m);
        }
        check(gs_index_parallel_insert_ids(h_, flat.data(), ids.data(), datas.size()));
    }
    // parallel_search(&[Vec<T>], knbn, ef) -> Vec<Vec<Neighbour>>, ascending distance (dnarequest.rs:353)
    std::vector<std::vector<Neighbour>> parallel_search(const std::vector<std::vector<T>> &datas, size_t knbn, size_t ef) const
    {
        if (!h_) throw Error(GS_ERR_STATE, "search on an empty index");
        const size_t nq = datas.size(), m = prm_.m;
        std::vector<T> flat(nq * m);
        for (size_t i = 0; i < nq; i++) { if (datas[i].size() != m) throw Error(GS_ERR_INVALID, "signature length mismatch"); std::copy(datas[i].begin(), datas[i].end(), flat.begin() + i * m); }
        std::vector<uint64_t> ids(nq * knbn); std::vector<float> dist(nq * knbn); std::vector<uint32_t> cnt(nq);
        std::vector<uint8_t> pl(nq * knbn); std::vector<int32_t> pr(nq * knbn);
        check(gs_index_parallel_search_pid(h_, flat.data(), nq, (uint32_t)knbn, (uint32_t)ef, ids.data(), dist.data(), cnt.data(), nullptr, pl.data(), pr.data()));
        std::vector<std::vector<Neighbour>> out(nq);
        for (size_t i = 0; i < nq; i++)
            for (uint32_t j = 0; j < cnt[i]; j++) out[i].push_back(Neighbour{(size_t)ids[i * knbn + j], dist[i * knbn + j], PointId{pl[i * knbn + j], pr[i * knbn + j]}});
        return out;
    }
    void file_dump(const std::string &path) const { check(gs_index_save(h_, path.c_str())); }      // dumpload.rs:31 (own format)
private:
    void frozen() const { if (h_) throw Error(GS_ERR_STATE, "index parameters are frozen once the index holds points"); }
    void ensure(size_t m) { if (!h_) { prm_.m = (uint32_t)m; check(gs_index_create(ctx_->get(), &prm_, &h_)); } }
    Context *ctx_;
    gs_index_params prm_;
    gs_index *h_ = nullptr;
};

}  // namespace gsearch
