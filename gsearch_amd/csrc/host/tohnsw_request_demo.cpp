// tohnsw_request_demo.cpp — the reference's call sequence for this path, restated in C++ over include/gsearch_amd.hpp:
//   dna_process_tohnsw (src/dna/dnasketch.rs:493-644): sketcher = OptDensHashSketch::new(params); Hnsw::new(...);
//       modify_level_scale; set_extend_candidates(true); set_keeping_pruned(false); sketch every file; ONE parallel_insert (:435)
//   get_sequence_matcher (src/dna/dnarequest.rs:395): sketch the queries; ONE parallel_search(knbn, ef) (:353);
//       ReqAnswer::dump (src/answer.rs:35-76) with out_threshold 0.99 (dnarequest.rs:83)
// Input: two FASTA files, every record = one genome ("--block" semantics). Not a CLI replacement: it exists so the C++ mirror
// is compiled and exercised by tests/test_gpu_parity.py::test_cpp_host_mirror.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include "../../../include/gsearch_amd.hpp"

using namespace gsearch;

// GPU-free mode `--answer-fixture FILE`: formats a literal request through gsearch::ReqAnswer (answer.rs:35-76) so the text format
// can be checked against a hand-derived expected file (tests/golden/reqanswer_*.txt). Lines of FILE, tab separated:
//   S path fasta_id len            next SeqDict entry
//   T threshold_f32_bits_hex       out_threshold (dnarequest.rs:83)
//   Q rank path fasta_id len       starts a request      N d_id distance_f32_bits_hex    a neighbour of it
//   B hamming_f32_bits_hex k       prints bindash.rs:93-99 compute_distance with 9 decimals on its own line
static std::vector<std::string> split_tabs(const std::string &l)
{
    std::vector<std::string> f; size_t b = 0;
    for (;;) { size_t e = l.find('\t', b); f.push_back(l.substr(b, e == std::string::npos ? e : e - b)); if (e == std::string::npos) break; b = e + 1; }
    return f;
}
static float f32_from_hex(const std::string &h) { uint32_t u = (uint32_t)std::stoul(h, nullptr, 16); float f; std::memcpy(&f, &u, 4); return f; }
static int answer_fixture(const char *path)
{
    std::ifstream f(path);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); return 1; }
    SeqDict seqdict; float threshold = 0.99f;
    struct Req { size_t rank; ItemDict item; std::vector<Neighbour> nb; };
    std::vector<Req> reqs; std::vector<std::pair<float, size_t>> bd;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        auto t = split_tabs(line);
        if (t[0] == "S") seqdict.push_back(ItemDict{t[1], t[2], std::stoull(t[3])});
        else if (t[0] == "T") threshold = f32_from_hex(t[1]);
        else if (t[0] == "Q") reqs.push_back(Req{std::stoull(t[1]), ItemDict{t[2], t[3], std::stoull(t[4])}, {}});
        else if (t[0] == "N") reqs.back().nb.push_back(Neighbour{std::stoull(t[1]), f32_from_hex(t[2])});
        else if (t[0] == "B") bd.emplace_back(f32_from_hex(t[1]), std::stoull(t[2]));
    }
    for (auto &r : reqs) ReqAnswer(r.rank, r.item, r.nb).dump(seqdict, threshold, std::cout);
    std::cout << "\n";
    for (auto &b : bd) { char buf[64]; std::snprintf(buf, sizeof buf, "%.9f\n", bindash_compute_distance(b.first, b.second)); std::cout << buf; }
    return 0;
}

static std::vector<std::pair<std::string, Record>> read_fasta(const char *path)
{
    std::vector<std::pair<std::string, Record>> out;
    std::ifstream f(path);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == '>') out.emplace_back(line.substr(1), Record());
        else if (!out.empty()) out.back().second += line;
    }
    return out;
}

int main(int argc, char **argv)
{
    if (argc == 3 && std::string(argv[1]) == "--answer-fixture") return answer_fixture(argv[2]);
    if (argc < 9) { std::fprintf(stderr, "usage: %s db.fa queries.fa k sketch_size max_nb_conn ef_construction ef_search knbn\n", argv[0]); return 2; }
    try {
        const uint32_t k = std::stoul(argv[3]), s = std::stoul(argv[4]), M = std::stoul(argv[5]), efc = std::stoul(argv[6]), ef = std::stoul(argv[7]), knbn = std::stoul(argv[8]);
        Context ctx(0);
        SeqSketcherParams params(k, s, SketchAlgo::OPTDENS, DataType::DNA);
        OptDensHashSketch<float> sketcher(ctx, params);
        DistHamming dist(ctx);
        Hnsw<float> hnsw(M, 1500000, 16, efc, dist, /*seed*/ 1);
        hnsw.modify_level_scale(0.25);
        hnsw.set_extend_candidates(true);
        hnsw.set_keeping_pruned(false);
        // ---- tohnsw
        auto db = read_fasta(argv[1]);
        std::vector<std::vector<Record>> genomes;
        for (auto &g : db) genomes.push_back({g.second});
        std::vector<float> flat = sketcher.sketch_genomes(genomes);
        std::vector<std::vector<float>> sigs(db.size());
        std::vector<std::pair<const std::vector<float> *, size_t>> data_for_hnsw;
        for (size_t i = 0; i < db.size(); i++) { sigs[i].assign(flat.begin() + i * s, flat.begin() + (i + 1) * s); data_for_hnsw.emplace_back(&sigs[i], i); }
        hnsw.parallel_insert(data_for_hnsw);
        // ---- request
        auto qs = read_fasta(argv[2]);
        std::vector<std::vector<float>> qsigs;
        for (auto &q : qs) { std::vector<const Record *> v{&q.second}; qsigs.push_back(sketcher.sketch_compressedkmer_seqs(v)[0]); }
        auto knn = hnsw.parallel_search(qsigs, knbn, ef);
        const float threshold = 0.99f;                       // out_threshold, dnarequest.rs:83
        SeqDict seqdict;
        for (auto &g : db) seqdict.push_back(ItemDict{argv[1], g.first, g.second.size()});
        for (size_t i = 0; i < knn.size(); i++)
            ReqAnswer(i, ItemDict{argv[2], qs[i].first, qs[i].second.size()}, knn[i]).dump(seqdict, threshold, std::cout);
        std::cout.flush();
        std::printf("\n");
    } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
