#!/usr/bin/env python3
"""BASELINE configs[0] (the reference's own CPU-runnable case) on both sides:
   tohnsw on 1k synthetic 1 Mbp DNA genomes (10 roots x 100 mutants), k=21 s=12000 --algo optdens, M=128 efc=1600 scale 0.25,
   then request 100 queries n=50 ef=5000.  CPU = oracle (OpenMP, all cores), GPU = the C-ABI library. Full parity check:
   sketches, graph, neighbour ids/distances, recall@50 vs exhaustive search."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsearch_amd as G
import helpers as H
import oracle_lib as O

N, NQ, L, k, m, M, efc, ef, knbn, B = 1000, 100, 1_000_000, 21, 12000, 128, 1600, 5000, 50, 64
rng = np.random.default_rng(1)
roots = [H.rand_dna(rng, L) for _ in range(10)]
mus = [0.001, 0.005, 0.01, 0.02, 0.05, 0.10]
genomes = [[H.dna_ascii(H.mutate(rng, roots[i % 10], mus[(i // 10) % 6]))] for i in range(N)]
queries = [[H.dna_ascii(H.mutate(rng, roots[i % 10], 0.01))] for i in range(NQ)]
cores = os.cpu_count()
res = {"config": "C1: %d x %.0f Mbp, k=%d s=%d optdens, M=%d efc=%d, %d queries n=%d ef=%d" % (N, L / 1e6, k, m, M, efc, NQ, knbn, ef), "cores": cores}
# ---- CPU oracle
recs = [g[0] for g in genomes]
t = time.perf_counter(); seq, rs, rl = O.pack_dna(recs); res["cpu_pack_s"] = time.perf_counter() - t
t = time.perf_counter(); osig = O.sketch_batch(O.params(k, m, "optdens"), seq, rs, rl, np.arange(N + 1, dtype=np.uint64), nthreads=cores); res["cpu_sketch_s"] = time.perf_counter() - t
oix = O.Index(np.float32, m, M, efc, scale_modify=0.25, seed=7)
t = time.perf_counter(); oix.parallel_insert(osig, batch=B); res["cpu_insert_s"] = time.perf_counter() - t
qseq, qrs, qrl = O.pack_dna([q[0] for q in queries])
t = time.perf_counter(); oq = O.sketch_batch(O.params(k, m, "optdens"), qseq, qrs, qrl, np.arange(NQ + 1, dtype=np.uint64), nthreads=cores)
oids, odist, ocnt, oev = oix.parallel_search(oq, knbn, ef, nthreads=cores); res["cpu_request_s"] = time.perf_counter() - t
# ---- GPU
sk = G.OptDensHashSketch.new(G.SeqSketcherParams(k, m, "optdens"))
sk.sketch_genomes(genomes[:2])      # warm-up (module load)
t = time.perf_counter(); gsig = sk.sketch_genomes(genomes); res["gpu_sketch_s_incl_pack_and_pcie"] = time.perf_counter() - t
hn = G.Hnsw.new(M, 1_500_000, 16, efc, G.DistHamming(), seed=7, insert_batch=B)
hn.modify_level_scale(0.25); hn.set_extend_candidates(True); hn.set_keeping_pruned(False)
t = time.perf_counter(); hn.parallel_insert(gsig); res["gpu_insert_s"] = time.perf_counter() - t
t = time.perf_counter(); gq = sk.sketch_genomes(queries); ids, dist, cnt, ev = hn.search_arrays(gq, knbn, ef); res["gpu_request_s"] = time.perf_counter() - t
g, og = hn.export_graph(), oix.export()
graph_ok = bool(np.array_equal(g["deg0"], og["deg0"]) and all(np.array_equal(g["nbr0"][i, :og["deg0"][i]], og["nbr0"][i, :og["deg0"][i]]) for i in range(N)))
bi, bd = O.bruteforce_topk(osig, oq, knbn, nthreads=cores)
res.update({"sketch_bit_exact": bool(np.array_equal(gsig.view(np.uint32), osig.view(np.uint32))), "graph_identical": graph_ok,
            "ids_identical": bool(np.array_equal(ids, oids)), "dist_identical": bool(np.array_equal(dist, odist)), "evals_identical": bool(np.array_equal(ev, oev)),
            "recall50_gpu": float(np.mean([(dist[i] <= bd[i, -1]).mean() for i in range(NQ)])), "recall50_cpu": float(np.mean([(odist[i] <= bd[i, -1]).mean() for i in range(NQ)])),
            "mean_evals_per_query": float(ev.mean()), "ani_of_best_hit_q0": G.ani(float(dist[0][0]), k, 1)})
res["cpu_total_s"] = res["cpu_sketch_s"] + res["cpu_insert_s"] + res["cpu_request_s"]
res["gpu_total_s"] = res["gpu_sketch_s_incl_pack_and_pcie"] + res["gpu_insert_s"] + res["gpu_request_s"]
print(json.dumps(res))
