// tohnsw_request_demo.cpp — the reference's call sequence for this path, restated in C++ over include/gsearch_amd.hpp:
//   dna_process_tohnsw (src/dna/dnasketch.rs:493-644): sketcher = OptDensHashSketch::new(params); Hnsw::new(...);
//       modify_level_scale; set_extend_candidates(true); set_keeping_pruned(false); sketch every file; ONE parallel_insert (:435)
//   get_sequence_matcher (src/dna/dnarequest.rs:395): sketch the queries; ONE parallel_search(knbn, ef) (:353);
//       ReqAnswer::dump (src/answer.rs:35-76) with out_threshold 0.99 (dnarequest.rs:83)
// Input: two FASTA files, every record = one genome ("--block" semantics). Not a CLI replacement: it exists so the C++ mirror
// is compiled and exercised by tests/test_gpu_parity.py::test_cpp_host_mirror.
#include <cstdio>
#include <fstream>
#include <iostream>
#include "../../../include/gsearch_amd.hpp"

using namespace gsearch;

// Rust's {:.5E} (answer.rs:58-63): mantissa with 5 decimals, exponent without padding or '+' sign, e.g. 6.07500E-1
static std::string rust_5e(float x)
{
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%.5E", (double)x);
    std::string s(buf);
    const size_t e = s.find('E');
    return s.substr(0, e) + "E" + std::to_string(std::stoi(s.substr(e + 1)));
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
        const float threshold = 0.99f;
        for (size_t i = 0; i < knn.size(); i++) {
            bool has_match = false;
            for (auto &n : knn[i]) has_match |= n.distance <= threshold;
            if (!has_match) continue;
            std::printf("\n%zu\t%s\tfasta_id:\t%s\tlength:\t%zu", i, argv[2], qs[i].first.c_str(), qs[i].second.size());
            for (auto &n : knn[i])
                if (n.distance < threshold)
                    std::printf("\nquery_id:\t%s\tdistance:\t%s\tanswer_fasta_path\t%s\t%s \t answer_seq_len:\t %zu", argv[2], rust_5e(n.distance).c_str(), argv[1], db[n.d_id].first.c_str(), db[n.d_id].second.size());
        }
        std::printf("\n");
    } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
