#!/usr/bin/env python3
"""bench.py — headline benchmark of the sketch-and-query hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sketch|request]

One rank per GPU. `python bench.py --gpus N` starts the N ranks itself (gsearch_amd.sharding.ensure_launched re-executes this
script under torch.distributed.run on 127.0.0.1); when a launcher already set RANK/WORLD_SIZE the script is one of its ranks.
A *step* is one pass of the hot path over one batch of synthetic input that is already resident in HBM:

  request (default; BASELINE.json configs[2]/[3]): sketch 10 000 query genomes (5 Mbp) + HNSW parallel_search (n=50, ef=5000)
          against a 300k-genome OptDens HNSW built on the GPU from synthetic genomes during (untimed) setup.
  sketch  (BASELINE.json configs[1]): 10k synthetic 5 Mbp genomes, k=21, s=18000, --algo optdens, per rank.

Rank 0 prints ONE JSON line (metric/value/... plus `roofline`, `kernels` and `cpu_baseline`; DESIGN.md "measurement").
Every kernel is priced against the ceiling of its real class: HBM bytes for the row-gather distance kernel,
memory-side atomics/s for the match-join, VALU issue for the sketch kernel, and for the latency-bound dense traversal its
algorithmic bytes (adjacency rows + 2-byte count lookups) against HBM peak next to pops/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ATOMICS_PEAK = 2.71e10         # scattered no-return 32-bit atomics/s, window resident in the Infinity Cache (<= 256 MB): the rate of the atomic units themselves, measured on MI355X
ATOMICS_SLAB = 1.92e10         # ... uniformly random over a window of the size of a 2500-query count-matrix slab (1.43 GB): tools/ubench_atomic, profiles/r04_ubench_atomic.txt
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md:52-54: a wave64 VALU op issues in 2 cycles (SIMD-32)
MASK64 = (1 << 64) - 1


def splitmix_mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    return z ^ (z >> np.uint64(31))


def synth_genome_packed(seed, g, length):
    """host twin of gs_synth_dna_dev: packed bytes of synthetic genome g (include/gsearch_amd.h)."""
    nw = (length + 31) // 32
    with np.errstate(over="ignore"):
        base = np.uint64((seed * 0x9e3779b97f4a7c15 + g * 0xbf58476d1ce4e5b9) & MASK64)
        x = splitmix_mix(base + np.arange(nw, dtype=np.uint64))
    nb = length - (nw - 1) * 32
    if nb < 32:
        be = int(x[-1].byteswap()) & (MASK64 << (64 - 2 * nb)) & MASK64
        x[-1] = np.uint64(be).byteswap()
    return x.view(np.uint8)


class Dist:
    """rank bookkeeping + the few collectives the bench itself needs (barrier, max/sum of a scalar)"""

    def __init__(self, backend):
        import torch
        from gsearch_amd import sharding as S
        self.torch = torch
        self.rank, self.world, self.local = S.rank_env()
        self.backend = backend
        self.on = "RANK" in os.environ            # under a launcher the collectives run even with one rank
        if backend == "nccl":
            torch.cuda.set_device(self.local)
        if self.on:
            import torch.distributed as td
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {"device_id": torch.device("cuda", self.local)} if backend == "nccl" else {}
            td.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
        self.device = torch.device("cuda", self.local) if backend == "nccl" else torch.device("cpu")

    def barrier_sync(self):
        if self.on:
            import torch.distributed as td
            td.barrier()
        if self.backend == "nccl":
            self.torch.cuda.synchronize()

    def _reduce(self, value, op):
        if not self.on:
            return value
        import torch.distributed as td
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device)
        td.all_reduce(t, op=op)
        return float(t.item())

    def max(self, v):
        import torch.distributed as td
        return self._reduce(v, td.ReduceOp.MAX)

    def sum(self, v):
        import torch.distributed as td
        return self._reduce(v, td.ReduceOp.SUM)

    def sum_i64(self, v):
        if not self.on:
            return int(v)
        import torch.distributed as td
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.device)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return int(t.item())

    def finish(self):
        if self.on:
            import torch.distributed as td
            td.barrier()                      # rank 0 may still be timing the CPU baseline: leave together
            td.destroy_process_group()


# ------------------------------------------------------------------------------------------------------
def run_selftest_launch(args, D):
    """launch-path self-test without GPUs (tests/test_bench_launch.py): the same launcher, rank bookkeeping, barrier/max timing and
    the packed single all-gather as the real run, with a stub searcher whose answers are a known function of the global query id."""
    from gsearch_amd import sharding as S
    torch = D.torch
    nq, knbn = 37, 5
    ex = S.TopkExchange(nq, knbn, D.world, D.device)
    gq = torch.arange(D.rank * nq, (D.rank + 1) * nq, dtype=torch.int64).view(nq, 1)
    j = torch.arange(knbn, dtype=torch.int64).view(1, knbn)
    ex.ids.copy_(gq * 1000 + j)
    ex.dist.copy_((gq * 8 + j).to(torch.float32) / 64.0)
    D.barrier_sync()
    t0 = time.perf_counter()
    ex.exchange()
    D.barrier_sync()
    dt = D.max(time.perf_counter() - t0)
    ids, dist = ex.gathered()
    allq = torch.arange(D.world * nq, dtype=torch.int64).view(-1, 1)
    ok = bool(torch.equal(ids, allq * 1000 + j) and torch.equal(dist, (allq * 8 + j).to(torch.float32) / 64.0))
    ranks = D.sum(float(1 << D.rank))
    # --shard db: every rank answers ALL nq queries on its shard of a 1000-node database (stub: node g of query q lies at distance ((g * 37 + q * 11) % 1000) / 1000),
    # ONE all-gather of the shard-major blocks, k-way merge under (distance, id) with the shards' id offsets: every rank must end with the global top-k
    db_ok = None
    if args.shard == "db":
        import numpy as np
        NDB = 1000
        lo, hi = S.shard_bounds(NDB, D.rank, D.world)
        q = np.arange(nq).reshape(nq, 1)
        g = np.arange(lo, hi).reshape(1, -1)
        dl = ((g * 37 + q * 11) % 1000).astype(np.float32) / np.float32(1000.0)
        order = np.lexsort((np.broadcast_to(g, dl.shape), dl), axis=1)[:, :knbn]
        ex.ids.copy_(torch.from_numpy(order.astype(np.int64)))                                   # LOCAL ids, as a shard's index returns them
        ex.dist.copy_(torch.from_numpy(np.take_along_axis(dl, order, axis=1)))
        ex.exchange()
        ids_g, dist_g = ex.gathered()
        offs = [S.shard_bounds(NDB, r, D.world)[0] for r in range(D.world)]
        ids_s = [ids_g[r * nq:(r + 1) * nq].cpu().numpy().astype(np.uint64) + np.uint64(offs[r]) for r in range(D.world)]
        dist_s = [dist_g[r * nq:(r + 1) * nq].cpu().numpy() for r in range(D.world)]
        mi, md = S.merge_topk_shards(ids_s, dist_s, knbn)
        ga = np.arange(NDB).reshape(1, -1)
        da = ((ga * 37 + q * 11) % 1000).astype(np.float32) / np.float32(1000.0)
        oa = np.lexsort((np.broadcast_to(ga, da.shape), da), axis=1)[:, :knbn]
        db_ok = D.sum_i64(1 if (np.array_equal(mi, oa.astype(np.uint64)) and np.array_equal(md, np.take_along_axis(da, oa, axis=1))) else 0) == D.world
    if D.rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": D.world, "backend": D.backend, "rank_mask": int(ranks), "collectives_per_step": 1,
                          "gathered_equals_expected": ok, "db_sharded_merge_equals_global_topk_on_every_rank": db_ok, "seconds": dt}))


# ------------------------------------------------------------------------------------------------------
def host_cpu_budget():
    """(logical CPUs the host reports, CPUs this process may really use): the second is capped by the affinity mask and by the
    cgroup CPU quota (cpu.max) - a 256-thread box with a quota of 16 CPUs runs 256 OpenMP threads no faster than 16"""
    logical = os.cpu_count() or 1
    eff = float(logical)
    try:
        eff = min(eff, float(len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt and txt[0] != "max":
                    eff = min(eff, float(txt[0]) / float(txt[1]))
            else:
                q = float(txt[0]); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    eff = min(eff, q / per)
            break
        except (OSError, ValueError, IndexError):
            continue
    return logical, max(1, int(eff + 0.999))


def cpu_sketch_baseline(O, op, args, kmers_per_genome, words, cores):
    """oracle sketch (OpenMP, one task per genome - the reference's decomposition, dnasketch.rs:322-357) on a bounded sample"""
    L = args.genome_len
    ns = max(args.cpu_sample, cores)
    pk = [synth_genome_packed(args.seed, i, L) for i in range(8)]
    wpad = words * 8                                   # the sample re-uses 8 distinct genomes round-robin: same work, bounded memory
    buf = np.zeros(len(pk) * wpad + 16, np.uint8)
    for i, b in enumerate(pk):
        buf[i * wpad:(i + 1) * wpad] = b
    srs = (np.arange(ns, dtype=np.uint64) % np.uint64(len(pk))) * np.uint64(words * 32)
    t0 = time.perf_counter()
    O.sketch_batch(op, buf, srs, np.full(ns, L, np.uint64), np.arange(ns + 1, dtype=np.uint64), nthreads=cores)
    cdt = time.perf_counter() - t0
    return {"value": kmers_per_genome * ns / cdt, "unit": "k-mers/s", "cores": cores, "threads": cores, "host_logical_cpus": host_cpu_budget()[0], "kind": "port",
            "sample": "%d genomes x %.1f Mbp, k=%d s=%d optdens, oracle/gs_oracle.c with OpenMP on the %d CPUs this process may use (affinity / cgroup quota; the host lists %d), one task per genome, %.1f s wall"
                      % (ns, L / 1e6, args.kmer, args.sketch_size, cores, host_cpu_budget()[0], cdt),
            "note": "CPU restatement (oracle), not upstream gsearch: the Rust reference cannot be built here"}


def sketch_valu_model(kmers_per_sec):
    """VALU issue model of k_sketch_min. Per-opcode issue costs come from the committed ISA mix of the full hash (tools/isa_mix.py +
    tools/ubench_valu -> profiles/r02_sketch_isa_mix.json); the instruction COUNT per k-mer is the one rocprofv3 measured for this very
    workload (SQ_INSTS_VALU, tools/pmc_sketch.sh -> profiles/r02_sketch_pmc.json), because the filtered emitter drops most k-mers after
    two of the three SplitMix64 mixes and its work per k-mer depends on the data."""
    import glob
    def newest(pattern):                                   # the latest round's file (r05_... sorts after r02_...)
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return f[-1] if f else None
    path = newest("r0*_sketch_isa_mix.json")
    if not path:
        return None
    mix = json.load(open(path))
    cyc_per_instr = mix["issue_cycles_per_kmer"] / mix["valu_per_kmer"]       # mean issue cycles of a VALU instruction of this mix (wave64)
    valu, src = mix["valu_per_kmer"], "static ISA count of the unfiltered loop"
    pmc = newest("r0*_sketch_pmc.json")
    if pmc:
        pj = json.load(open(pmc))
        valu, src = pj["valu_wave_instr_per_64_kmers"], "SQ_INSTS_VALU of profiles/%s (taken at %s)" % (os.path.basename(pmc), pj.get("head") or "an earlier commit")
    cyc = valu * cyc_per_instr
    peak = SIMDS * CLOCK_HZ / cyc * 64                     # k-mers/s if every SIMD issued nothing but this stream
    return {"bound": "valu", "valu_wave_instr_per_64_kmers": valu, "valu_wave_instr_per_64_kmers_full_hash": mix["valu_per_kmer"],
            "mean_issue_cycles_per_valu_instr": cyc_per_instr, "kmers_per_sec_at_issue_ceiling": peak, "frac": kmers_per_sec / peak,
            "source": "instruction count: %s; issue costs: profiles/%s + profiles/r02_ubench_valu.txt" % (src, os.path.basename(path))}


def join_valu_model(elements_per_launch, batch_ms):
    """the instruction budget of k_match_join per (slot, node) element, from the SQ counters of the request-time launches (tools/pmc_any.sh over
    `bench.py --steps 1`, condensed in profiles/r0*_step_sq_counters.txt): measured lane-instructions per element against a hand count of what one element needs,
    and how busy VALU issue is - the kernel is priced against the memory-side atomics (frac_of_atomics_ceiling), not against VALU."""
    import glob
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_step_sq_counters.txt")))
    if not f:
        return None
    vals = {}
    for line in open(f[-1], errors="replace"):
        if "k_match_join<3, unsigned int, 1, false>" in line and "grid=4194304" in line:
            p = line.split()
            name = [x for x in p if x.startswith(("SQ_", "GRBM", "FETCH", "WRITE"))]
            if name:
                vals[name[0]] = float(line.rsplit("mean=", 1)[1])
    if "SQ_INSTS_VALU" not in vals:
        return None
    el = 18000.0 * 300000.0 * (18000 - 48) / 18000                 # the launch the counters were taken on: slots 48.. of 300 000 nodes
    valu, salu, lds = vals["SQ_INSTS_VALU"] * 64 / el, vals.get("SQ_INSTS_SALU", 0) * 64 / el, vals.get("SQ_INSTS_LDS", 0) * 64 / el
    issue_ms = vals["SQ_INSTS_VALU"] * 4.2 / (SIMDS * CLOCK_HZ) * 1e3                    # ~4.2 cycles per wave64 VALU instruction of this mix (profiles/r02_ubench_valu.txt)
    return {"source": os.path.basename(f[-1]), "valu_lane_instr_per_element": valu, "salu_wave_instr_x64_per_element": salu, "lds_lane_instr_per_element": lds,
            "necessary_valu_per_element_hand_count": 32,
            "hand_count": "bitmap filter 10 (hash multiply, word index, LDS read, two bit tests, mask) + next value's load / validity 2 + table build amortised over the 8 values a lane "
                          "owns per slot: clear 3, 2.4 inserts x ~40 / 8 = 12 + ~7 % survivors x ~60 (queue entry, probe chain, accumulator rule) = 4 + loop 1 = 32; the other half "
                          "is the survivor queue's push / pop (eight ballot + mbcnt sequences each, executed by every lane of a wavefront that has ONE survivor)",
            "valu_issue_ms_per_launch_at_4.2_cycles": issue_ms, "valu_issue_frac_of_batch_time": issue_ms / batch_ms if batch_ms else None,
            "note": "VALU issue is about half busy: the binding resource is the memory-side atomic unit (frac_of_atomics_ceiling); halving the queue's instructions would not shorten the launch"}


def join_class_string():
    """what binds k_match_join, with the NEWEST committed evidence named (the string used to cite round-4 files whatever the round)"""
    import glob
    def newest(pattern):
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return os.path.basename(f[-1]) if f else None
    sq, ub, wo = newest("r0*_step_sq_counters.txt"), newest("r0*_ubench_atomic.txt"), newest("r0*_join_without_atomics.log")
    return ("atomics + valu: memory-side atomics at `frac_of_atomics_ceiling` of the scattered no-return rate of profiles/%s; VALU issue about half busy (`valu_model`, SQ counters of "
            "profiles/%s); with the atomics compiled out the launch keeps ~0.8 of its time (profiles/%s) - two walls, the atomic unit first" % (ub, sq, wo))


def run_sketch(args, D):
    import ctypes as C
    import gsearch_amd as G

    k, m, L = args.kmer, args.sketch_size, args.genome_len
    ng = args.genomes                      # per rank (weak scaling)
    ctx = G.Context(D.local)
    lib = ctx.L
    prm = G.SeqSketcherParams(k, m, "optdens")
    words = (L + 31) // 32
    seq_bytes = ng * words * 8
    d_seq = ctx.alloc(seq_bytes + 64)
    d_sig = ctx.alloc(ng * m * 4)
    rs = (np.arange(ng, dtype=np.uint64) * np.uint64(words * 32))
    rl = np.full(ng, L, dtype=np.uint64)
    goff = np.arange(ng + 1, dtype=np.uint64)
    d_rs, d_rl, d_goff = ctx.alloc(rs.nbytes), ctx.alloc(rl.nbytes), ctx.alloc(goff.nbytes)
    ctx.upload(d_rs, rs); ctx.upload(d_rl, rl); ctx.upload(d_goff, goff)
    first = D.rank * ng
    G._lib.check(lib.gs_synth_dna_dev(ctx.h, args.seed, first, ng, L, d_seq))
    ctx.sync()

    def step():
        G._lib.check(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, seq_bytes + 64, d_rs, d_rl, ng, d_goff, ng, d_sig))

    for _ in range(args.warmup):
        step()
    ctx.profile(True)
    ctx.profile_read(0, reset=True)
    D.barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    D.barrier_sync()
    dt = D.max(time.perf_counter() - t0)
    kern_ms, kern_n = ctx.profile_read(0, reset=True)
    ctx.profile(False)

    kmers_per_genome = L - k + 1
    value = float(kmers_per_genome) * ng * D.world * args.steps / dt
    out = {
        "metric": "sketch k-mers/sec", "value": value, "unit": "k-mers/s", "n_gpus": D.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "sketch-only: %d synthetic %.1f Mbp genomes per GPU, k=%d s=%d optdens (BASELINE configs[1])" % (ng, L / 1e6, k, m),
                   "genomes_per_gpu": ng, "genome_len": L, "kmer_size": k, "sketch_size": m, "algo": "optdens"},
        "genomes_per_sec": ng * D.world * args.steps / dt,
    }
    if D.rank == 0:
        # roofline of the dominant kernel (k_sketch_min): algorithmic bytes = ceil(L/4) + m*4 per genome
        alg_bytes_launch = (float((L + 3) // 4) + m * 4.0) * ng
        avg_ms = kern_ms / max(kern_n, 1)
        achieved = alg_bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kps = kmers_per_genome * ng / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": pmc_traffic("k_sketch_min"), "kernel": "k_sketch_min", "avg_launch_ms": avg_ms, "launches": kern_n,
                           "algorithmic_bytes_per_launch": alg_bytes_launch, "kmers_per_sec_kernel": kps, "class": "valu",
                           "note": "hash-bound kernel: 0.264 B of HBM traffic per k-mer; the binding resource is VALU issue (valu_model)"}
        out["roofline"]["valu_model"] = sketch_valu_model(kps)
        # parity spot check + CPU baseline (oracle = checker / baseline only, never the measured path)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        op = O.params(k, m, "optdens")
        sig_dev = ctx.download(d_sig, (ng, m), np.float32)
        chk = [0, ng - 1] if ng > 1 else [0]
        ok = True
        for g in chk:
            pk = np.concatenate([synth_genome_packed(args.seed, first + g, L), np.zeros(16, np.uint8)])
            ref = O.sketch_batch(op, pk, np.zeros(1, np.uint64), np.array([L], np.uint64), np.array([0, 1], np.uint64))
            ok &= bool(np.array_equal(ref.view(np.uint32)[0], sig_dev.view(np.uint32)[g]))
        out["parity_checked"] = {"genomes": chk, "bit_exact_vs_oracle": ok}
        out["cpu_baseline"] = cpu_sketch_baseline(O, op, args, kmers_per_genome, words, host_cpu_budget()[1]) if (D.world == 1 and not args.no_cpu_baseline) else None
        print(json.dumps(out))
    for p in (d_seq, d_sig, d_rs, d_rl, d_goff):
        ctx.free(p)


_PMC = None
def _pmc_defaults():
    """the committed sidecars, newest round first (profiles/r06_... sorts after r05_...)"""
    import glob
    return [os.path.relpath(f, ROOT) for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc_sidecar.json")), reverse=True)]


PMC_DEFAULTS = tuple(_pmc_defaults())


def _pmc_load():
    """rocprofv3 FETCH_SIZE / WRITE_SIZE per launch of the hot kernels, condensed by tools/pmc_bench.sh + tools/pmc_condense.py. The
    sidecar named by GS_PMC_SIDECAR (a PMC session of THIS run) wins; otherwise the committed sidecar of the builder's last PMC session
    over the same command is used and labelled as such (`traffic_source`): counters cannot be collected inside the timed run itself."""
    global _PMC
    if _PMC is None:
        _PMC = {"kernels": {}, "_source": None}
        cands = [os.environ["GS_PMC_SIDECAR"]] if os.environ.get("GS_PMC_SIDECAR") else [os.path.join(ROOT, c) for c in PMC_DEFAULTS]
        for p in cands:
            if p and os.path.exists(p):
                try:
                    _PMC = json.load(open(p))
                    _PMC["_source"] = ("this session (GS_PMC_SIDECAR): " if os.environ.get("GS_PMC_SIDECAR") else "builder PMC session, committed sidecar: ") + os.path.relpath(p, ROOT) + \
                                      ((" taken at " + _PMC["_head"]) if _PMC.get("_head") else "")
                    break
                except Exception:
                    _PMC = {"kernels": {}, "_source": None}
    return _PMC


def pmc_traffic(kernel, coalesced_bytes=None):
    """HBM bytes per launch of `kernel` from the memory-side counters, corrected as profiles/r03_fetchcal.txt calibrates them on gfx950
    (tools/ubench_fetchcal over an 8 GiB buffer): FETCH_SIZE reports HALF the bytes of coalesced reads (128-B requests tallied at 64 B;
    measured 0.500 for 16 B/lane AND for 4 B/lane streams), exactly 64 B per scattered 2-byte look-up (one request each), WRITE_SIZE is
    exact for stores and 32 B per no-return atomic. Streaming kernels: 2 x FETCH + WRITE. The dense traversal mixes both kinds of reads:
    its coalesced part (adjacency rows, `coalesced_bytes`, known from the pop counter) is under-counted by half, its look-ups are not:
    FETCH + coalesced_bytes / 2 + WRITE. None when no sidecar is available."""
    k = _pmc_load().get("kernels", {}).get(kernel)
    if not k:
        return None
    fetch, write = k.get("FETCH_SIZE_bytes_raw"), k.get("WRITE_SIZE_bytes", 0.0)
    if fetch is None:
        return k.get("hbm_bytes_per_launch")
    if coalesced_bytes is None:
        return 2.0 * fetch + write
    return fetch + min(coalesced_bytes, 2.0 * fetch) / 2.0 + write


def pmc_source():
    return _pmc_load().get("_source")


# ------------------------------------------------------------------------------------------------------
def run_request(args, D):
    """BASELINE configs[2] (and [3] for N>1): `request` = sketch the query genomes + HNSW parallel_search (ef=5000, n=50)
    against a prebuilt OptDens HNSW (M=128, efc=1600, scale 0.25) over `--db-genomes` synthetic genomes, DB replicated
    per GPU, queries sharded, top-k all-gathered over RCCL (one packed collective per step)."""
    import ctypes as C
    import gsearch_amd as G
    from gsearch_amd import sharding as S
    torch = D.torch
    chk = G._lib.check

    k, m, L = args.kmer, args.sketch_size, args.genome_len
    N, nq_rank, qps = args.db_genomes, args.queries, args.queries_per_step
    knbn, ef = args.knbn, args.ef_search
    n_roots = max(N // args.per_root, 1)
    mu_lo, mu_hi = 0.001, 0.08
    ctx = G.Context(D.local)
    lib = ctx.L
    prm = G.SeqSketcherParams(k, m, "optdens")
    words = (L + 31) // 32
    gbytes = words * 8
    hn = G.Hnsw.new(args.max_nb_conn, 1_500_000, 16, args.ef_construction, G.DistHamming(ctx), dtype=np.float32, seed=args.seed,
                    insert_batch=256, ctx=ctx)
    hn.modify_level_scale(args.scale_modify); hn.set_extend_candidates(True); hn.set_keeping_pruned(False)
    hn._ensure(m)

    def sketch_dev(d_seq, n, d_sig, d_rs, d_rl, d_goff):
        chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, n * gbytes + 64, d_rs, d_rl, n, d_goff, n, d_sig))

    # --shard db (the other decomposition of SURVEY 8e, the per-shard loop + merge of /root/reference/scripts/multiple_search.sh:71-107): the DATABASE is split over the
    # ranks (contiguous genome ranges), every rank sketches and answers ALL queries of a step on its shard, the gathered answers (one all-gather of world x qps blocks) are
    # merged on the device under (distance, id) with the shards' id offsets (gs_topk_merge_dev). Total work per step is fixed: "strong" scaling.
    db_shard = args.shard == "db"
    g_lo, g_hi = S.shard_bounds(N, D.rank, D.world) if db_shard else (0, N)
    N_loc = g_hi - g_lo
    # ---- tohnsw (untimed setup): generate -> sketch -> parallel_insert, in chunks
    chunk = min(args.build_chunk, max(N_loc, 1))
    d_seq = ctx.alloc(chunk * gbytes + 64)
    d_sig = ctx.alloc(chunk * m * 4)
    nrec = max(chunk, qps)                                  # record tables serve the build chunks and the query batches
    rs = np.arange(nrec, dtype=np.uint64) * np.uint64(words * 32)
    d_rs, d_rl, d_goff = ctx.alloc(rs.nbytes), ctx.alloc(rs.nbytes), ctx.alloc(8 * (nrec + 1))
    ctx.upload(d_rs, rs); ctx.upload(d_rl, np.full(nrec, L, np.uint64)); ctx.upload(d_goff, np.arange(nrec + 1, dtype=np.uint64))
    t_b = time.perf_counter()
    for g0 in range(g_lo, g_hi, chunk):
        n = min(chunk, g_hi - g0)
        chk(lib.gs_synth_dna_family_dev(ctx.h, args.seed, g0, n, L, n_roots, mu_lo, mu_hi, d_seq))
        sketch_dev(d_seq, n, d_sig, d_rs, d_rl, d_goff)
        chk(lib.gs_index_parallel_insert_dev(hn.h, d_sig, n))
        if D.rank == 0 and args.verbose:
            print("# built %d / %d in %.1fs" % (g0 - g_lo + n, N_loc, time.perf_counter() - t_b), file=sys.stderr, flush=True)
    ctx.sync()
    build_s = time.perf_counter() - t_b
    ctx.free(d_seq); ctx.free(d_sig)
    chk(lib.gs_index_release_build_scratch(hn.h))              # the build is over: the insert-time pair cache (up to 55 % of the device) goes back before the request

    # ---- query genomes resident in HBM before the timed region: fresh mutants of the DB's roots
    d_qseq = ctx.alloc(nq_rank * gbytes + 64)
    q_first = 1_000_000_000 + (0 if db_shard else D.rank * nq_rank)     # (db sharding: every rank answers the SAME queries)
    chk(lib.gs_synth_dna_family_dev(ctx.h, args.seed, q_first, nq_rank, L, n_roots, mu_lo, mu_hi, d_qseq))
    d_qsig = ctx.alloc(qps * m * 4)
    # the exchange: ONE all-gather of the per-rank top-k blocks per step, through the LIBRARY's communicator (gs_comm_*: its own RCCL communicator, pack kernel ->
    # ncclAllGather -> unpack kernel on the context's stream; torch only ships the 128-byte unique id at set-up) - the path a Rust host would call. Should that
    # communicator not come up on this node, the torch collective takes over and the line says so (`multi_gpu_check.collective`).
    ex, ex_kind = None, "none (single rank)"
    if D.on and D.backend == "nccl" and not os.environ.get("GS_BENCH_TORCH_EXCHANGE"):
        try:
            ex = S.LibExchange(ctx, qps, qps, knbn, D.world, D.rank, D.device)
            ex_kind = "gs_comm_allgatherv_topk_dev: one ncclAllGather of the packed top-k blocks per step (the library's RCCL communicator)"
        except Exception as e:                                 # noqa: BLE001 - any failure here must not cost the scaling run
            print("# rank %d: gs_comm set-up failed (%s): torch.distributed all_gather instead" % (D.rank, e), file=sys.stderr, flush=True)
            ex = None
    lib_ok = D.sum_i64(1 if isinstance(ex, S.LibExchange) else 0) if D.on else 0
    if D.on and lib_ok != D.world:                             # all ranks or none
        ex = None
    if ex is None:
        ex = S.TopkExchange(qps, knbn, D.world if D.on else 1, D.device)       # ids + distances in one send buffer -> ONE all-gather per step
        if D.on and D.world > 1:
            ex_kind = "one all_gather_into_tensor of the packed top-k blocks per step (torch.distributed / RCCL)"
    ids_t, dist_t = ex.ids, ex.dist
    cnt_t = torch.empty((qps,), dtype=torch.int32, device=D.device)
    ev_t = torch.zeros((qps,), dtype=torch.int64, device=D.device)
    nsteps_q = max(nq_rank // qps, 1)
    evals_steps = []

    merged_ids = torch.empty((qps, knbn), dtype=torch.int64, device=D.device) if db_shard else None
    merged_dist = torch.empty((qps, knbn), dtype=torch.float32, device=D.device) if db_shard else None
    shard_off = np.array([S.shard_bounds(N, r, D.world)[0] for r in range(D.world)], dtype=np.uint64)

    def step(i):
        b = i % nsteps_q
        sketch_dev(d_qseq + b * qps * gbytes, qps, d_qsig, d_rs, d_rl, d_goff)
        chk(lib.gs_index_parallel_search_dev(hn.h, d_qsig, qps, knbn, ef, ids_t.data_ptr(), dist_t.data_ptr(), cnt_t.data_ptr(), ev_t.data_ptr()))
        if D.on:           # RCCL all-gather of the per-rank top-k blocks
            ex.exchange()
        if db_shard:       # every rank merges the shards' answers for the step's queries (shard-major blocks, as gathered)
            if D.on and D.world > 1:
                a_ids, a_dist = (ex.all_ids, ex.all_dist) if isinstance(ex, S.LibExchange) else ex.gathered()
                if not isinstance(ex, S.LibExchange):
                    torch.cuda.current_stream().synchronize()      # (the torch collective ran on torch's stream, the merge runs on the library's)
            else:
                a_ids, a_dist = ids_t, dist_t
            G.topk_merge_dev(ctx, a_ids.data_ptr(), a_dist.data_ptr(), D.world, qps, knbn, knbn, merged_ids.data_ptr(), merged_dist.data_ptr(), id_offset=shard_off)

    for i in range(args.warmup):
        step(args.steps + i)
    ctx.profile(True)
    for fam in range(4):
        ctx.profile_read(fam, reset=True)
    stats0 = hn.search_stats(reset=True)
    D.barrier_sync()
    t0 = time.perf_counter()
    step_ms = []
    for i in range(args.steps):
        t_s = time.perf_counter()
        step(i)
        step_ms.append((time.perf_counter() - t_s) * 1e3)      # the library calls return when their results are ready
        evals_steps.append(ev_t.clone())
    D.barrier_sync()
    dt = D.max(time.perf_counter() - t0)
    srch_ms, srch_n = ctx.profile_read(2, reset=True)
    tile_ms, tile_n = ctx.profile_read(1, reset=True)
    sk_ms, sk_n = ctx.profile_read(0, reset=True)
    ctx.profile(False)
    st = hn.search_stats(reset=True)
    del stats0
    evals_total = float(sum(int(e.sum().item()) for e in evals_steps))
    # the exchange, checked on EVERY rank after the timed region: my block sits at my offset of the gathered buffer, and the gathered buffer
    # carries every rank's answers (checksum of all blocks == sum over ranks of the checksum of each rank's own block)
    M63 = (1 << 62) - 1
    own_sum = int((ids_t.to(torch.int64) & 0xFFFFFFFF).sum().item()) & M63
    exchange_ok, all_sum = True, own_sum
    if D.on and D.world > 1:
        ids_all, dist_all = ex.gathered()
        mine = slice(D.rank * qps, (D.rank + 1) * qps)
        exchange_ok = bool(torch.equal(ids_all[mine], ids_t) and torch.equal(dist_all[mine], dist_t))
        all_sum = int((ids_all.to(torch.int64) & 0xFFFFFFFF).sum().item()) & M63
    sum_of_own = D.sum_i64(own_sum) & M63
    n_ok = D.sum_i64(1 if (exchange_ok and all_sum == sum_of_own) else 0)
    rank_mask = D.sum_i64(1 << D.rank)
    ranks_seen = ex.ranks_seen() if isinstance(ex, S.LibExchange) else bin(rank_mask).count("1")     # gs_comm_size of the library's communicator
    value = (1 if db_shard else D.world) * qps * args.steps / dt
    out = {
        "metric": "query genomes/sec", "value": value, "unit": "genomes/s", "n_gpus": D.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if db_shard else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "request: %d query genomes x %.1f Mbp per GPU per step (k=%d s=%d optdens sketch + HNSW search n=%d ef=%d) against a %d-genome "
                               "OptDens HNSW (M=%d efc=%d scale %.2f) built on the GPU, DB replicated per GPU, queries sharded (BASELINE configs[2]/[3]); query genomes "
                               "HBM-resident before the timed region, %d distinct sets rotated over the steps; queries are fresh mutants of the DB's %d families (~%.1f per "
                               "family per step); dense strategy (match-join count matrix + look-up traversal); the many-isolates-per-species regime is `request_redundant`"
                               % (qps, L / 1e6, k, m, knbn, ef, N, args.max_nb_conn, args.ef_construction, args.scale_modify, max(nq_rank // qps, 1), n_roots, qps / n_roots),
                   "db_genomes": N, "queries_per_gpu_per_step": qps, "genome_len": L, "kmer_size": k, "sketch_size": m, "knbn": knbn, "ef_search": ef,
                   "max_nb_conn": args.max_nb_conn, "ef_construction": args.ef_construction, "collectives_per_step": 1 if D.on else 0,
                   "distinct_query_sets": max(nq_rank // qps, 1), "queries_per_db_family_per_step": qps / n_roots},
        "step_ms": [round(x, 2) for x in step_ms], "build_seconds": build_s, "build_genomes_per_sec": N / build_s, "dist_evals_per_query": evals_total / (qps * args.steps),
        "sketch_kmers_per_sec": (L - k + 1) * qps * sk_n / (sk_ms * 1e-3) if sk_ms > 0 else None,
        "multi_gpu_check": {"rccl_ranks_seen": ranks_seen, "ranks_whose_block_and_checksum_verified": n_ok, "world": D.world, "collective": ex_kind,
                            "sharding": "database split over the ranks, every rank answers all queries, gathered answers merged by gs_topk_merge_dev" if db_shard
                                        else "database replicated, queries split over the ranks"},
        "build_scratch_released": True,
    }
    if db_shard:
        out["db_shard"] = {"genomes_of_this_rank": int(N_loc), "first_genome": int(g_lo), "merged_best_distance_sum": float(merged_dist[:, 0].sum().item())}
    if D.rank == 0:
        out.update(request_accounting(args, ctx, hn, lib, chk, torch, D, dict(
            srch=(srch_ms, srch_n), tile=(tile_ms, tile_n), sk=(sk_ms, sk_n), stats=st, evals_total=evals_total, d_qsig=d_qsig, ids_t=ids_t,
            dist_t=dist_t, cnt_t=cnt_t)))
        if D.world > 1 or args.no_cpu_baseline:                # parity sample and CPU baseline: rank 0 at N=1 only
            out["cpu_baseline"] = None
        else:
            par, oix = request_parity_and_cpu(args, ctx, hn, d_qsig, d_qseq, ids_t, dist_t, ev_t, nsteps_q, gbytes, words)
            out.update(par)
            if not args.no_extra_legs and args.redundant_roots > 0:
                try:
                    out["request_redundant"] = redundant_leg(args, ctx, hn, lib, chk, torch, sketch_dev, d_rs, d_rl, d_goff, gbytes, oix, out["ms_per_step"])
                except Exception as e:                          # the headline line must survive a failing side leg
                    out["request_redundant"] = {"error": repr(e)}
            del oix
        if D.world == 1 and not args.no_extra_legs:
            try:
                out["extra_legs"] = extra_legs(args, ctx, lib, chk, d_qseq, gbytes, words, d_rs, d_rl, d_goff)
            except Exception as e:                              # the headline line must survive a failing side leg
                out["extra_legs"] = {"error": repr(e)}
            if args.skew_alpha > 0 and not args.no_cpu_baseline:
                try:                                            # last: it needs the memory of the headline index
                    hn.close(); ctx.free(d_qseq); ctx.free(d_qsig)
                    chk(lib.gs_ctx_release_scratch(ctx.h))
                    out["request_skewed"] = skewed_leg(args, ctx, lib, chk, d_rs, d_rl, d_goff, gbytes, out["ms_per_step"])
                except Exception as e:                          # noqa: BLE001
                    out["request_skewed"] = {"error": repr(e)}
        print(json.dumps(out))


def request_accounting(args, ctx, hn, lib, chk, torch, D, r):
    """per-kernel accounting over the timed region (HIP events around every launch on the library's stream + device-side work counters)"""
    m, N, qps, knbn, ef = args.sketch_size, args.db_genomes, args.queries_per_step, args.knbn, args.ef_search
    L, k = args.genome_len, args.kmer
    row_bytes = m * 4.0                                           # 72 000 B per (query,candidate) evaluation, SURVEY 8d
    (srch_ms, srch_n), (tile_ms, tile_n), (sk_ms, sk_n), st, evals_total = r["srch"], r["tile"], r["sk"], r["stats"], r["evals_total"]
    kernels = []
    join = os.environ.get("GS_DENSE_IMPL", "join") != "tile"
    dense = tile_n > 0
    if dense:   # dense mode: the counts of every (query, node) pair of the step are produced up front
        pairs_total = float(qps) * N * args.steps
        if join:        # a join batch is two launches (the first 48 slots, then the rest: gs_join.hip heavy blocks) - price the batch, not the launch
            tile_n = args.steps * (-(-qps // 3276))
        avg_ms = tile_ms / tile_n
        kd = {"kernel": "k_match_join" if join else "k_hamming_qxc", "total_ms": tile_ms, "launches": tile_n, "avg_launch_ms": avg_ms,
              "role": ("equi-join of a query batch (<= 3276 queries) with the column-major DB copy (all query x node pairs); `launches` counts batches, each = the join over "
                       "the first 48 slots + the join over the rest" if join
                       else "dense DistHamming compare tile (all query x node pairs)"),
              "nominal_bytes_avoided_per_launch": pairs_total / tile_n * row_bytes}
        if join:
            col_bytes = float(N) * row_bytes                                   # the column store is streamed once per launch
            atom = st.get("join_atomics", 0)
            kd.update({"class": join_class_string(),
                       "algorithmic_bytes_per_launch": col_bytes + 2.0 * pairs_total / tile_n,
                       "column_stream_GBps": col_bytes * tile_n / (tile_ms * 1e-3) / 1e9, "atomics_per_launch": atom / tile_n,
                       "atomics_per_sec": atom / (tile_ms * 1e-3), "atomics_peak_per_sec": ATOMICS_PEAK, "frac_of_atomics_ceiling": atom / (tile_ms * 1e-3) / ATOMICS_PEAK,
                       "atomics_uniform_random_slab_per_sec": ATOMICS_SLAB,
                       "traffic": pmc_traffic("k_match_join")})
            kd["achieved_GBps"] = kd["algorithmic_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
            kd["valu_model"] = join_valu_model(float(N) * (m - 48) , avg_ms)
        else:
            valu_peak = SIMDS * 64 / (2 * 2.0) * CLOCK_HZ                      # 2 VALU instr per pair-element, 2 cycles per wave64 instr
            kd.update({"class": "valu", "pair_elements_per_sec": pairs_total * m / (tile_ms * 1e-3), "valu_peak_pair_elements_per_sec": valu_peak,
                       "valu_frac": pairs_total * m / (tile_ms * 1e-3) / valu_peak, "algorithmic_bytes_per_launch": pairs_total / tile_n * row_bytes / 128.0,
                       "traffic": pmc_traffic("k_hamming_qxc")})
            kd["achieved_GBps"] = kd["algorithmic_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
        kernels.append(kd)
    avg_s = srch_ms / max(srch_n, 1)
    if dense:
        # algorithmic bytes of the dense traversal: the adjacency row of every popped candidate (deg x 4 B) + one 2-byte count per evaluation
        alg = (st.get("adj_bytes", 0) + 2.0 * evals_total) / max(srch_n, 1)
        pops = st.get("pops", 0)
        kt = {"kernel": "k_hnsw_search_dense", "class": "latency", "role": "HNSW traversal (dense mode: looks the counts up in the query x node matrix)",
              "total_ms": srch_ms, "launches": srch_n, "avg_launch_ms": avg_s, "algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / (avg_s * 1e-3) / 1e9 if avg_s else 0.0,
              "pops_per_query": pops / max(qps * args.steps, 1), "accepting_pops_per_query": st.get("accepting_pops", 0) / max(qps * args.steps, 1),
              "pops_per_sec": pops / (srch_ms * 1e-3) if srch_ms else 0.0, "workgroups_in_flight": st.get("wg_in_flight", 0),
              "ns_per_pop_per_workgroup": (srch_ms * 1e6 * st.get("wg_in_flight", 0) / pops) if pops else None,
              "nominal_bytes_avoided_per_launch": evals_total / max(srch_n, 1) * row_bytes,
              "traffic": pmc_traffic("k_hnsw_search_dense", coalesced_bytes=st.get("adj_bytes", 0) / max(srch_n, 1)),
              "traffic_rule": "FETCH_SIZE raw (64 B per 2-byte look-up) + half the adjacency-row bytes again (coalesced rows are tallied at half) + WRITE_SIZE; profiles/r03_fetchcal.txt"}
    else:
        alg = evals_total / max(srch_n, 1) * row_bytes
        kt = {"kernel": "k_hnsw_search", "class": "hbm", "role": "HNSW traversal (gather mode: streams one row per evaluation)", "total_ms": srch_ms, "launches": srch_n,
              "avg_launch_ms": avg_s, "algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / (avg_s * 1e-3) / 1e9 if avg_s else 0.0, "traffic": pmc_traffic("k_hnsw_search")}
    kernels.append(kt)
    kps = (L - k + 1) * qps * sk_n / (sk_ms * 1e-3) if sk_ms > 0 else 0.0
    sk_alg = (float((L + 3) // 4) + m * 4.0) * qps
    kernels.append({"kernel": "k_sketch_min", "class": "valu", "role": "query sketching", "total_ms": sk_ms, "launches": sk_n, "avg_launch_ms": sk_ms / max(sk_n, 1),
                    "algorithmic_bytes_per_launch": sk_alg, "achieved_GBps": sk_alg * sk_n / (sk_ms * 1e-3) / 1e9 if sk_ms else 0.0, "kmers_per_sec": kps,
                    "valu_model": sketch_valu_model(kps), "traffic": pmc_traffic("k_sketch_min")})
    dom = max(kernels, key=lambda kk: kk["total_ms"])
    out = {"kernels": kernels}
    out["roofline"] = {"bound": "hbm", "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["achieved_GBps"] / HBM_PEAK_GBS,
                       "traffic": dom.get("traffic"), "kernel": dom["kernel"], "class": dom.get("class"), "avg_launch_ms": dom["avg_launch_ms"], "launches": dom["launches"],
                       "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                       "note": "dominant kernel of the step by summed launch time; `achieved` = its algorithmic bytes / measured launch time (HIP events). "
                               "`class` names the resource that really binds it (see `kernels`); the HBM-bound distance kernel of the north star is `roofline_gather_mode`"}
    if dom.get("traffic"):
        out["roofline"]["traffic_GBps"] = dom["traffic"] / (dom["avg_launch_ms"] * 1e-3) / 1e9
        out["roofline"]["traffic_over_algorithmic"] = dom["traffic"] / dom["algorithmic_bytes_per_launch"] if dom["algorithmic_bytes_per_launch"] else None
    out["roofline"]["traffic_source"] = pmc_source()
    out["roofline"]["traffic_rule"] = dom.get("traffic_rule", "2 x FETCH_SIZE + WRITE_SIZE (coalesced streams are tallied at half: profiles/r03_fetchcal.txt)")
    # the row-streaming (gather) form of the same search on a sub-batch: the HBM-bound DistHamming kernel the north star prices
    # against the HBM roofline (>= 40 % target); results are identical, only the evaluation strategy differs
    ng_q = min(256, qps)
    prev_mode = os.environ.get("GS_DIST_MODE")
    os.environ["GS_DIST_MODE"] = "gather"
    try:
        ev_g = torch.zeros((ng_q,), dtype=torch.int64, device="cuda")
        ids_g2 = torch.empty((ng_q, knbn), dtype=torch.int64, device="cuda")
        dist_g2 = torch.empty((ng_q, knbn), dtype=torch.float32, device="cuda")
        ctx.profile(True); ctx.profile_read(2, reset=True)
        chk(lib.gs_index_parallel_search_dev(hn.h, r["d_qsig"], ng_q, knbn, ef, ids_g2.data_ptr(), dist_g2.data_ptr(), r["cnt_t"].data_ptr(), ev_g.data_ptr()))
        g_ms, g_n = ctx.profile_read(2, reset=True); ctx.profile(False)
        g_bytes = float(ev_g.sum().item()) * row_bytes
        out["roofline_gather_mode"] = {"bound": "hbm", "kernel": "k_hnsw_search (GS_DIST_MODE=gather)", "queries": ng_q, "launch_ms": g_ms / max(g_n, 1),
                                       "algorithmic_bytes": g_bytes, "achieved": g_bytes / (g_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("k_hnsw_search"),
                                       "same_answers_as_dense": bool(torch.equal(ids_g2, r["ids_t"][:ng_q]) and torch.equal(dist_g2, r["dist_t"][:ng_q]))}
    finally:
        if prev_mode is None:
            os.environ.pop("GS_DIST_MODE", None)
        else:
            os.environ["GS_DIST_MODE"] = prev_mode
    return out


def unpack_dna_ascii(packed, length):
    """2-bit packed genome (SPEC 1.1 layout) -> ASCII bases"""
    b = np.frombuffer(packed, np.uint8)
    codes = np.stack([(b >> 6) & 3, (b >> 4) & 3, (b >> 2) & 3, b & 3], axis=1).reshape(-1)[:length]
    return np.frombuffer(b"ACGT", np.uint8)[codes]


def _gz_write(job):
    import zlib
    path, header, body = job
    co = zlib.compressobj(6, zlib.DEFLATED, 31)                  # gzip container, level 6 (the default of gzip; what genome archives serve)
    with open(path, "wb") as f:
        f.write(co.compress(header) + co.compress(body) + co.flush())
    return path


def synth_proteome(seed, g, length):
    """host twin of gs_synth_aa_dev: the residues of synthetic proteome g (include/gsearch_amd.h), padded to whole 8-byte words"""
    nw = (length + 7) // 8
    with np.errstate(over="ignore"):
        base = np.uint64((((seed ^ 0xAA5EED) * 0x9e3779b97f4a7c15) + g * 0xbf58476d1ce4e5b9) & MASK64)
        x = splitmix_mix(base + np.arange(nw, dtype=np.uint64))
    return np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)[x.view(np.uint8) % 20]


def sig_row_roots(seed, first, n, n_roots):
    """host twin of the family assignment of gs_synth_sigs_dev: root of rows first .. first + n"""
    with np.errstate(over="ignore"):
        r = np.arange(first, first + n, dtype=np.uint64)
        return splitmix_mix(np.uint64((seed * 31) & MASK64) + r * np.uint64(0xA24BAED4963EE407) + np.uint64(3)) % np.uint64(n_roots)


def c5_distance_leg(args, ctx, lib, chk, O):
    """DistHamming on u64 signatures of m = 24000 (configs[4]) through the row-gather traversal: 192 kB per evaluation. 50 000 rows (9.6 GB, far beyond
    the 256 MB Infinity Cache) of 2000 families; every query from a DIFFERENT family. ef = 5000 makes each traversal evaluate >= 5000 of the 50 000
    rows, so concurrent queries re-read one another's rows from L2 / Infinity Cache: the ALGORITHMIC rate (192 kB x evaluations / time) is not an HBM
    rate. The roofline fraction printed is the one of the memory-side counters (tools/pmc_bench.sh collects FETCH_SIZE of this very launch through
    `bench.py --workload c5dist`); without a sidecar it is null."""
    import gsearch_amd as G
    maa, nd, n_roots, nqx, seed = 24000, args.c5_rows, 2000, 256, 77
    prev_mode = os.environ.get("GS_DIST_MODE")
    os.environ["GS_DIST_MODE"] = "gather"
    d_db = ctx.alloc(nd * maa * 8)
    try:
        chk(lib.gs_synth_sigs_dev(ctx.h, 2, maa, seed, 0, nd, n_roots, 0.3, 0.95, d_db))
        hx = G.Hnsw.new(24, max(nd, 1024), 16, 64, G.DistHamming(ctx), dtype=np.uint64, seed=5, insert_batch=256, ctx=ctx)
        hx.modify_level_scale(0.25); hx.set_extend_candidates(True); hx.set_keeping_pruned(False)
        hx._ensure(maa)
        t0 = time.perf_counter()
        chk(lib.gs_index_parallel_insert_dev(hx.h, d_db, nd)); ctx.sync()
        build_s = time.perf_counter() - t0
        # 256 query rows of pairwise different families: the first row of each new family among rows 5 000 000 ..
        roots = sig_row_roots(seed, 5_000_000, 4 * nqx, n_roots)
        _, firsts = np.unique(roots, return_index=True)
        rows = np.sort(firsts)[:nqx]
        assert len(rows) == nqx
        d_qx = ctx.alloc(nqx * maa * 8)
        for i, r in enumerate(rows):
            chk(lib.gs_synth_sigs_dev(ctx.h, 2, maa, seed, 5_000_000 + int(r), 1, n_roots, 0.3, 0.95, d_qx + i * maa * 8))
        d_ids, d_dist, d_cnt, d_ev = ctx.alloc(8 * nqx * 50), ctx.alloc(4 * nqx * 50), ctx.alloc(4 * nqx), ctx.alloc(8 * nqx)
        ms = None
        for rep in range(2):
            ctx.profile(True); ctx.profile_read(2, reset=True)
            chk(lib.gs_index_parallel_search_dev(hx.h, d_qx, nqx, 50, 5000, d_ids, d_dist, d_cnt, d_ev))
            g_ms, g_n = ctx.profile_read(2, reset=True); ctx.profile(False)
            ms = g_ms if ms is None or g_ms < ms else ms
        ev = ctx.download(d_ev, (nqx,), np.uint64)
        gb = float(ev.sum()) * maa * 8
        ids = ctx.download(d_ids, (nqx, 50), np.uint64); dist = ctx.download(d_dist, (nqx, 50), np.float32)
        oix = O.Index(np.uint64, maa, 24, 64, scale_modify=0.25, seed=5)
        oix.import_graph(ctx.download(d_db, (nd, maa), np.uint64), hx.export_graph(), view=True)
        qh = ctx.download(d_qx, (nqx, maa), np.uint64)[:16]
        oids, odist, _, oev = oix.parallel_search(qh, 50, 5000, nthreads=host_cpu_budget()[1])
        traffic = pmc_traffic("k_hnsw_search_u64")
        out = {"kernel": "k_hnsw_search<u64> (row gather, ballot/popcount)", "index_nodes": nd, "index_families": n_roots, "queries": nqx, "query_families": int(len(np.unique(roots[rows]))),
               "evals_per_query": float(ev.mean()), "launch_ms": ms, "algorithmic_bytes": gb, "algorithmic_GBps": gb / (ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBS,
               "traffic": traffic, "traffic_source": pmc_source() if traffic else None, "traffic_GBps": traffic / (ms * 1e-3) / 1e9 if traffic else None,
               "frac": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None, "rows_re_served_by_caches": (gb / traffic) if traffic else None, "index_build_s": build_s,
               "note": "frac = HBM bytes of the memory-side counters / launch time / 8 TB/s (null without a PMC sidecar); the algorithmic rate counts 192 kB per evaluation and, with every "
                       "traversal touching >= 10 % of a 50 000-row index, includes rows re-served by L2 / Infinity Cache - it is NOT an HBM fraction",
               "ids_distances_evals_equal_oracle_16_queries": bool(np.array_equal(oids, ids[:16]) and np.array_equal(odist, dist[:16]) and np.array_equal(oev, ev[:16]))}
        # ---- configs[4] AT SIZE: 10 000 u64 queries (fresh mutants of the index's families) through the DEFAULT strategy - for a batch
        # this size the cost model takes the dense one: match-join with 8-byte keys, then the look-up traversal - next to the same request forced through
        # the row-gather kernel on its first 512 queries; answers of a 16-query sample == oracle, and the two strategies agree on the 512
        nqs = int(os.environ.get("GS_C5_QUERIES", "10000"))
        if nqs > 0:
            for p_ in (d_qx, d_ids, d_dist, d_cnt, d_ev):
                ctx.free(p_)
            d_qx = ctx.alloc(nqs * maa * 8)
            chk(lib.gs_synth_sigs_dev(ctx.h, 2, maa, seed, 6_000_000, nqs, n_roots, 0.3, 0.95, d_qx))
            d_ids, d_dist, d_cnt, d_ev = ctx.alloc(8 * nqs * 50), ctx.alloc(4 * nqs * 50), ctx.alloc(4 * nqs), ctx.alloc(8 * nqs)
            if prev_mode is None:
                os.environ.pop("GS_DIST_MODE", None)
            else:
                os.environ["GS_DIST_MODE"] = prev_mode
            best = None
            for rep in range(3):
                ctx.profile(True)
                for fam in (1, 2):
                    ctx.profile_read(fam, reset=True)
                hx.search_stats(reset=True)
                ctx.sync(); t0 = time.perf_counter()
                chk(lib.gs_index_parallel_search_dev(hx.h, d_qx, nqs, 50, 5000, d_ids, d_dist, d_cnt, d_ev)); ctx.sync()
                dt = time.perf_counter() - t0
                t_ms, t_n = ctx.profile_read(2, reset=True); j_ms, j_n = ctx.profile_read(1, reset=True); ctx.profile(False)
                st = hx.search_stats(reset=True)
                if rep and (best is None or dt < best[0]):
                    best = (dt, t_ms, t_n, j_ms, j_n, st)
            dt, t_ms, t_n, j_ms, j_n, st = best
            ids_d = ctx.download(d_ids, (nqs, 50), np.uint64); dist_d = ctx.download(d_dist, (nqs, 50), np.float32); ev_d = ctx.download(d_ev, (nqs,), np.uint64)
            qh = ctx.download(d_qx, (16, maa), np.uint64)
            oids, odist, _, oev = oix.parallel_search(qh, 50, 5000, nthreads=host_cpu_budget()[1])
            os.environ["GS_DIST_MODE"] = "gather"
            ng = min(512, nqs)
            d_i2, d_d2, d_c2, d_e2 = ctx.alloc(8 * ng * 50), ctx.alloc(4 * ng * 50), ctx.alloc(4 * ng), ctx.alloc(8 * ng)
            ctx.profile(True); ctx.profile_read(2, reset=True)
            ctx.sync(); t0 = time.perf_counter()
            chk(lib.gs_index_parallel_search_dev(hx.h, d_qx, ng, 50, 5000, d_i2, d_d2, d_c2, d_e2)); ctx.sync()
            g_dt = time.perf_counter() - t0
            gg_ms, _ = ctx.profile_read(2, reset=True); ctx.profile(False)
            same = bool(np.array_equal(ctx.download(d_i2, (ng, 50), np.uint64), ids_d[:ng]) and np.array_equal(ctx.download(d_d2, (ng, 50), np.float32), dist_d[:ng]) and
                        np.array_equal(ctx.download(d_e2, (ng,), np.uint64), ev_d[:ng]))
            for p_ in (d_i2, d_d2, d_c2, d_e2):
                ctx.free(p_)
            out["request_at_size"] = {
                "queries": nqs, "strategy_chosen_by_the_cost_model": "dense: match-join (8-byte keys) + look-up traversal" if st["pops"] > 0 else "row gather",
                "call_ms": dt * 1e3, "queries_per_sec": nqs / dt, "traversal_kernel_ms": t_ms, "traversal_launches": t_n, "count_matrix_kernels_ms": j_ms, "count_matrix_launches": j_n,
                "join_atomics": int(st["join_atomics"]), "pops_per_query": st["pops"] / nqs, "evals_per_query": float(ev_d.mean()),
                "row_gather_same_request_first_%d_queries" % ng: {"call_ms": g_dt * 1e3, "kernel_ms": gg_ms, "queries_per_sec": ng / g_dt,
                                                                   "algorithmic_GBps": float(ev_d[:ng].sum()) * maa * 8 / (gg_ms * 1e-3) / 1e9, "same_ids_distances_evals_as_default": same},
                "speedup_over_row_gather_per_query": (g_dt / ng) / (dt / nqs),
                "ids_distances_evals_equal_oracle_16_queries": bool(np.array_equal(oids, ids_d[:16]) and np.array_equal(odist, dist_d[:16]) and np.array_equal(oev, ev_d[:16]))}
        del oix
        hx.close()
        for p_ in (d_qx, d_ids, d_dist, d_cnt, d_ev):
            ctx.free(p_)
        return out
    finally:
        ctx.free(d_db)
        if prev_mode is None:
            os.environ.pop("GS_DIST_MODE", None)
        else:
            os.environ["GS_DIST_MODE"] = prev_mode


def prob_request_leg(args, ctx, lib, chk, O, d_qseq, gbytes, d_rs, d_rl, d_goff):
    """`tohnsw` + `request` with --algo prob (ProbMinHash3a, the first sketcher north_star names): k = 21, s = 18000 -> u64 signatures (the type dispatch of
    /root/reference/src/dna/dnarequest.rs:417-455; sketcher /root/reference/src/dna/dnasketch.rs:499-518), 144 kB rows, 8-byte-key match-join. `--prob-db-genomes`
    synthetic genomes of the headline's families are generated, sketched and inserted in the collector's chunks; the resident query genomes of the first set are
    sketched and searched like a headline step (wall clock around sketch + search, inputs resident). A query sample is checked against the oracle: prob sketch of
    two query genomes, ids / distances / evaluation counts of 16 queries on the exported graph."""
    import ctypes as C
    import gsearch_amd as G
    k, m, L, qps, knbn, ef = args.kmer, args.sketch_size, args.genome_len, args.queries_per_step, args.knbn, args.ef_search
    N = args.prob_db_genomes
    n_roots = max(args.db_genomes // args.per_root, 1)            # the headline's families: the resident queries are their mutants
    prm = G.SeqSketcherParams(k, m, "prob")
    assert prm.sig_dtype() == np.uint64
    chunk = min(args.build_chunk, N)
    d_seq = ctx.alloc(chunk * gbytes + 64)
    d_sig = ctx.alloc(max(chunk, qps) * m * 8)
    hp = G.Hnsw.new(args.max_nb_conn, max(N, 1024), 16, args.ef_construction, G.DistHamming(ctx), dtype=np.uint64, seed=args.seed, insert_batch=256, ctx=ctx)
    hp.modify_level_scale(args.scale_modify); hp.set_extend_candidates(True); hp.set_keeping_pruned(False)
    hp._ensure(m)
    try:
        t_b = time.perf_counter(); sk_s = 0.0
        for g0 in range(0, N, chunk):
            n = min(chunk, N - g0)
            chk(lib.gs_synth_dna_family_dev(ctx.h, args.seed, g0, n, L, n_roots, 0.001, 0.08, d_seq)); ctx.sync()
            t0 = time.perf_counter()
            chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, n * gbytes + 64, d_rs, d_rl, n, d_goff, n, d_sig)); ctx.sync()
            sk_s += time.perf_counter() - t0
            chk(lib.gs_index_parallel_insert_dev(hp.h, d_sig, n))
        ctx.sync()
        build_s = time.perf_counter() - t_b
        ctx.free(d_seq); d_seq = None
        d_ids, d_dist, d_cnt, d_ev = ctx.alloc(8 * qps * knbn), ctx.alloc(4 * qps * knbn), ctx.alloc(4 * qps), ctx.alloc(8 * qps)
        steps = []
        for rep in range(3):                                        # rep 0 warms up (scratch, count matrix)
            ctx.profile(True)
            for fam in range(4):
                ctx.profile_read(fam, reset=True)
            hp.search_stats(reset=True)
            ctx.sync(); t0 = time.perf_counter()
            chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_qseq, qps * gbytes + 64, d_rs, d_rl, qps, d_goff, qps, d_sig)); ctx.sync()
            t1 = time.perf_counter()
            chk(lib.gs_index_parallel_search_dev(hp.h, d_sig, qps, knbn, ef, d_ids, d_dist, d_cnt, d_ev)); ctx.sync()
            t2 = time.perf_counter()
            srch, prod, sk = ctx.profile_read(2, reset=True), ctx.profile_read(1, reset=True), ctx.profile_read(0, reset=True)
            ctx.profile(False)
            st = hp.search_stats(reset=True)
            if rep:
                steps.append({"ms": (t2 - t0) * 1e3, "sketch_ms": (t1 - t0) * 1e3, "search_ms": (t2 - t1) * 1e3, "count_matrix_kernels_ms": prod[0], "count_matrix_launches": prod[1],
                              "traversal_kernel_ms": srch[0], "join_atomics": int(st.get("join_atomics", 0)), "pops_per_query": st.get("pops", 0) / qps})
        best = min(steps, key=lambda x: x["ms"])
        ns = 16
        ids = ctx.download(d_ids, (qps, knbn), np.uint64); dist = ctx.download(d_dist, (qps, knbn), np.float32); ev = ctx.download(d_ev, (qps,), np.uint64)
        qsig = ctx.download(d_sig, (qps, m), np.uint64)[:ns].copy()
        for p_ in (d_ids, d_dist, d_cnt, d_ev):
            ctx.free(p_)
        # oracle: prob sketches of two query genomes, the search of 16 queries on the exported graph
        qb = ctx.download(d_qseq, (2, gbytes), np.uint8)
        op = O.params(k, m, "prob")
        sk_ok = all(np.array_equal(O.sketch_batch(op, np.concatenate([qb[i].reshape(-1), np.zeros(16, np.uint8)]), np.zeros(1, np.uint64), np.array([L], np.uint64),
                                                  np.array([0, 1], np.uint64))[0], qsig[i]) for i in range(2))
        oix = O.Index(np.uint64, m, args.max_nb_conn, args.ef_construction, scale_modify=args.scale_modify, seed=args.seed)
        oix.import_graph(hp.get_data(), hp.export_graph(), view=True)
        oids, odist, _, oev = oix.parallel_search(qsig, knbn, ef, nthreads=min(host_cpu_budget()[1], ns))
        same = bool(np.array_equal(oids, ids[:ns]) and np.array_equal(odist.view(np.uint32), dist[:ns].view(np.uint32)) and np.array_equal(oev, ev[:ns]))
        del oix
        return {"workload": "tohnsw + request with --algo prob: %d genomes x %.1f Mbp (k=%d s=%d ProbMinHash3a, u64 signatures, 144 kB rows) in an HNSW (M=%d efc=%d), then %d resident "
                            "query genomes sketched (prob) and searched (n=%d ef=%d) per step" % (N, L / 1e6, k, m, args.max_nb_conn, args.ef_construction, qps, knbn, ef),
                "db_genomes": N, "build_seconds": build_s, "db_sketch_seconds": sk_s, "db_sketch_kmers_per_sec": float(L - k + 1) * N / sk_s,
                "ms_per_%d_queries" % qps: best["ms"], "genomes_per_sec": qps / (best["ms"] * 1e-3), "steps": steps,
                "query_sketch_kmers_per_sec": float(L - k + 1) * qps / (best["sketch_ms"] * 1e-3),
                "strategy": "dense: match-join (8-byte keys) + look-up traversal" if best["pops_per_query"] > 0 else "row gather",
                "evals_per_query": float(ev.mean()), "median_nearest_distance": float(np.median(dist[:, 0])),
                "prob_sketch_bit_exact_vs_oracle_2_query_genomes": bool(sk_ok), "ids_distances_evals_equal_oracle_%d_queries" % ns: same}
    finally:
        hp.close()
        if d_seq:
            ctx.free(d_seq)
        ctx.free(d_sig)
        chk(lib.gs_ctx_release_scratch(ctx.h))


def extra_legs(args, ctx, lib, chk, d_qseq, gbytes, words, d_rs, d_rl, d_goff):
    """Driver-visible numbers for the rest of BASELINE's configs and of DESIGN 4's rate table, measured AFTER the timed region in the same
    run (rank 0, N = 1; each leg a few seconds): other sketchers at (k=21, s=18000) on the resident 5 Mbp query genomes, configs[4]
    (AA k=7 s=24000 super2 u64: k-mers/s, bit-exact sample, u64 row-gather distance GB/s), and the file-inclusive gz ingest rate."""
    import ctypes as C
    import tempfile
    import shutil
    import gsearch_amd as G
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    out = {}
    k, L = args.kmer, args.genome_len
    # ---- (1) ProbMinHash3a / SetSketch / SuperMinHash2 at (21, 18000) on resident genomes
    rates = {}
    for algo, ng in (("prob", 256), ("hll", 512), ("super2", 1024), ("revoptdens", 1024)):
        ng = min(ng, args.queries)
        prm = G.SeqSketcherParams(k, args.sketch_size, algo)
        d_sig = ctx.alloc(ng * args.sketch_size * prm.sig_dtype().itemsize)
        best = None
        for rep in range(2):
            ctx.sync(); t0 = time.perf_counter()
            chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_qseq, ng * gbytes + 64, d_rs, d_rl, ng, d_goff, ng, d_sig))
            ctx.sync(); dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        sig = ctx.download(d_sig, (ng, args.sketch_size), prm.sig_dtype())
        qb = ctx.download(d_qseq, (1, gbytes), np.uint8)
        ref = O.sketch_batch(O.params(k, args.sketch_size, algo), np.concatenate([qb.reshape(-1), np.zeros(16, np.uint8)]), np.zeros(1, np.uint64), np.array([L], np.uint64), np.array([0, 1], np.uint64))
        rates[algo] = {"genomes": ng, "kmers_per_sec": (L - k + 1) * ng / best, "genomes_per_sec": ng / best, "wall_ms": best * 1e3,
                       "bit_exact_vs_oracle_genome0": bool(np.array_equal(ref[0].view(np.uint8), sig[0].view(np.uint8)))}
        ctx.free(d_sig)
        chk(lib.gs_ctx_release_scratch(ctx.h))                  # the sketchers' scratch (ProbMinHash: two copies of a chunk's k-mers) is not needed by the next leg
    out["other_sketchers_k21_s18000"] = rates
    # ---- (1b) `request` on ProbMinHash3a signatures at size
    if args.prob_db_genomes > 0:
        try:
            out["request_prob"] = prob_request_leg(args, ctx, lib, chk, O, d_qseq, gbytes, d_rs, d_rl, d_goff)
        except Exception as e:                                      # noqa: BLE001 - the other legs must survive
            out["request_prob"] = {"error": repr(e)}
    # ---- (2) BASELINE configs[4]: AA k=7 s=24000 super2 -> u64 signatures, all 50 000 proteomes (75 GB of residues generated in HBM, gs_synth_aa_dev)
    kaa, maa, Laa, NP = 7, 24000, 1_500_000, args.c5_proteomes
    pb = (Laa + 7) // 8 * 8
    d_aa = ctx.alloc(NP * pb + 64)
    chk(lib.gs_synth_aa_dev(ctx.h, args.seed, 0, NP, Laa, d_aa))
    rs = np.arange(NP, dtype=np.uint64) * np.uint64(pb)
    rl = np.full(NP, Laa, np.uint64)
    d_ars, d_arl, d_ago = ctx.alloc(8 * NP), ctx.alloc(8 * NP), ctx.alloc(8 * (NP + 1))
    ctx.upload(d_ars, rs); ctx.upload(d_arl, rl); ctx.upload(d_ago, np.arange(NP + 1, dtype=np.uint64))
    prm = G.SeqSketcherParams(kaa, maa, "super2", "aa")
    d_asig = ctx.alloc(NP * maa * 8)
    best = None
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_aa, NP * pb + 64, d_ars, d_arl, NP, d_ago, NP, d_asig))
        ctx.sync(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    pick = [0, NP // 2, NP - 1]
    asig = np.stack([ctx.download(d_asig + g * maa * 8, (maa,), np.uint64) for g in pick])
    text = np.concatenate([synth_proteome(args.seed, g, Laa) for g in pick] + [np.zeros(16, np.uint8)])
    ref = O.sketch_batch(O.params(kaa, maa, "super2", "aa"), text, np.arange(3, dtype=np.uint64) * np.uint64(pb), np.full(3, Laa, np.uint64), np.arange(4, dtype=np.uint64), nthreads=3)
    c5 = {"workload": "AA k=7 s=24000 super2 (u64 signatures): %d proteomes x %.2f M residues resident in HBM (BASELINE configs[4]; iid residues, gs_synth_aa_dev)" % (NP, Laa / 1e6),
          "proteomes": NP, "kmers_per_sec": float(Laa - kaa + 1) * NP / best, "proteomes_per_sec": NP / best, "wall_ms": best * 1e3,
          "bit_exact_vs_oracle_sample": bool(np.array_equal(ref, asig)), "sample": pick}
    for p_ in (d_aa, d_ars, d_arl, d_ago, d_asig):
        ctx.free(p_)
    c5["distance_gather_u64_m24000"] = c5_distance_leg(args, ctx, lib, chk, O)
    out["config4_aa_super2"] = c5
    # ---- (3) file-inclusive `request` input: the same 5 Mbp query genomes as gzip FASTA files on local disk -> gs_sketch_files
    nf = args.ingest_files
    if nf > 0:
        from multiprocessing import Pool
        d = tempfile.mkdtemp(prefix="gs_bench_gz_", dir="/tmp")
        try:
            cores = host_cpu_budget()[1]
            nd = min(128, nf, args.queries)                         # distinct genomes compressed; the other files are hard links to them (the
            qb = ctx.download(d_qseq, (nd, gbytes), np.uint8)       # readers and decoders do the same work per file, the disk holds 128)
            jobs = []
            for i in range(nd):
                seq = unpack_dna_ascii(qb[i], L)
                nl = (L + 79) // 80
                flat = np.full(nl * 80, 10, np.uint8)               # (the last line is padded with newlines: dropped by the reader like any non-ACGT byte)
                flat[:L] = seq
                body = np.full((nl, 81), 10, np.uint8)
                body[:, :80] = flat.reshape(nl, 80)
                jobs.append((os.path.join(d, "q%05d.fna.gz" % i), b">query%d synthetic\n" % i, body.tobytes()))
            with Pool(min(cores, 16)) as pool:
                paths = pool.map(_gz_write, jobs, chunksize=2)
            del jobs
            for i in range(nd, nf):
                q = os.path.join(d, "q%05d.fna.gz" % i)
                os.link(paths[i % nd], q)
                paths.append(q)
            sk = G.OptDensHashSketch.new(G.SeqSketcherParams(k, args.sketch_size, "optdens"), ctx=ctx)
            runs = {}
            for mode in ("dealt", "host_only"):                     # .gz files dealt between the host decoders and the device inflate kernel / host only
                prev = os.environ.get("GS_GZIP_DEVICE")
                os.environ["GS_GZIP_DEVICE"] = "1" if mode == "dealt" else "0"
                best, st_best = None, None
                for rep in range(3 if mode == "dealt" else 2):
                    t0 = time.perf_counter()
                    fsig_, nrec, nsym_, st = sk.sketch_files(paths)
                    dt = time.perf_counter() - t0
                    if best is None or dt < best:
                        best, st_best = dt, st
                    if mode == "dealt":
                        fsig, nsym = fsig_, nsym_
                if prev is None:
                    del os.environ["GS_GZIP_DEVICE"]
                else:
                    os.environ["GS_GZIP_DEVICE"] = prev
                runs[mode] = (best, st_best)
            best, st_best = runs["dealt"]
            # the resident path's signatures of the same genomes
            prm = G.SeqSketcherParams(k, args.sketch_size, "optdens")
            d_sig = ctx.alloc(nd * args.sketch_size * 4)
            chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_qseq, nd * gbytes + 64, d_rs, d_rl, nd, d_goff, nd, d_sig))
            rsig = ctx.download(d_sig, (nd, args.sketch_size), np.float32); ctx.free(d_sig)
            same = all(np.array_equal(fsig[i].view(np.uint32), rsig[i % nd].view(np.uint32)) for i in range(nf))
            out["ingest_gz_files"] = {"files": nf, "distinct_genomes": nd, "genome_len": L, "container": "gzip -6, one member, 80-column FASTA",
                                      "compressed_MB_per_genome": sum(os.path.getsize(p_) for p_ in paths[:nd]) / nd / 1e6,
                                      "genomes_per_sec_file_inclusive": nf / best, "wall_s": best, "host_cpus": cores,
                                      "genomes_per_sec_host_decoders_only": nf / runs["host_only"][0],
                                      "host_read_inflate_scan_cpu_s_per_genome": st_best["host_read_decode_scan_s"] / nf, "pcie_wait_s": st_best["pcie_wait_s"], "device_s": st_best["device_s"],
                                      "same_signatures_as_hbm_resident_path": bool(same), "bases_per_file_ok": bool((nsym == L).all()),
                                      "note": "page-cache resident files; single-member .gz files are dealt between the host decoders (libdeflate, host_cpus threads) and the device "
                                              "inflate kernel (k_inflate: one wave per member); `value` of this line stays the HBM-resident rate"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def request_parity_and_cpu(args, ctx, hn, d_qsig, d_qseq, ids_t, dist_t, ev_t, nsteps_q, gbytes, words):
    """parity / recall / CPU baseline on a bounded sample of the last step's queries (rank 0, N = 1 only)"""
    import gsearch_amd as G
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    k, m, L, qps, knbn, ef = args.kmer, args.sketch_size, args.genome_len, args.queries_per_step, args.knbn, args.ef_search
    logical, cores = host_cpu_budget()
    ns = min(max(args.cpu_sample_queries, cores), qps)             # at least one query per usable CPU: the CPU leg uses every core it may
    last = (args.steps - 1) % nsteps_q
    qsig = ctx.download(d_qsig, (qps, m), np.float32)[:ns]
    ids_g = ids_t.cpu().numpy().view(np.uint64)[:ns]
    dist_g = dist_t.cpu().numpy()[:ns]
    # (a) sketch parity + CPU sketch time for the sample's genomes
    op = O.params(k, m, "optdens")
    qbytes = ctx.download(d_qseq + last * qps * gbytes, (ns, gbytes), np.uint8)
    buf = np.concatenate([qbytes.reshape(-1), np.zeros(16, np.uint8)])
    t0 = time.perf_counter()
    osig = O.sketch_batch(op, buf, np.arange(ns, dtype=np.uint64) * np.uint64(words * 32), np.full(ns, L, np.uint64),
                          np.arange(ns + 1, dtype=np.uint64), nthreads=cores)
    cpu_sketch_s = time.perf_counter() - t0
    sketch_ok = bool(np.array_equal(osig.view(np.uint32), qsig.view(np.uint32)))
    # (b) the same graph searched by the oracle (CPU, all cores)
    db = hn.get_data()
    oix = O.Index(np.float32, m, args.max_nb_conn, args.ef_construction, scale_modify=args.scale_modify, seed=args.seed)
    oix.import_graph(db, hn.export_graph(), view=True)           # a pointer to the 21.6 GB of rows, not a second copy
    t0 = time.perf_counter()
    oids, odist, ocnt, oev = oix.parallel_search(qsig, knbn, ef, nthreads=min(cores, ns))
    cpu_search_s = time.perf_counter() - t0
    ids_ok = bool(np.array_equal(oids, ids_g)) and bool(np.array_equal(odist.view(np.uint32), dist_g.view(np.uint32)))
    evals_ok = bool(np.array_equal(oev, ev_t.cpu().numpy().view(np.uint64)[:ns]))
    # (c) recall@knbn against exhaustive search (tie-aware: a neighbour counts if it is within the k-th exact distance)
    nb = min(ns, 32)
    bi, bd = hn.bruteforce_search(qsig[:nb], knbn)
    rec_gpu = float(np.mean([(dist_g[i] <= bd[i, -1]).mean() for i in range(nb)]))
    rec_cpu = float(np.mean([(odist[i] <= bd[i, -1]).mean() for i in range(nb)]))
    ani_err = float(max(abs(G.ani(float(d), k) - O.ani(float(d), k)) for d in dist_g[0][: min(knbn, 8)]))
    out = {}
    out["parity_checked"] = {"queries": ns, "sketch_bit_exact_vs_oracle": sketch_ok, "neighbour_ids_and_distances_equal_oracle": ids_ok,
                             "dist_evaluation_counts_equal_oracle": evals_ok, "max_ani_abs_err": ani_err}
    out["recall_at_%d" % knbn] = {"gpu": rec_gpu, "cpu_oracle": rec_cpu, "queries": nb, "reference": "exhaustive DistHamming top-k, tie-aware"}
    out["cpu_baseline"] = {"value": ns / (cpu_sketch_s + cpu_search_s), "unit": "genomes/s", "cores": cores, "threads": min(cores, ns), "host_logical_cpus": logical, "kind": "port",
                           "sample": "%d of the step's query genomes on %d OpenMP threads = the CPUs this process may use (affinity / cgroup quota; the host lists %d): oracle sketch %.2fs + oracle parallel_search (same graph, ef=%d) %.2fs"
                                     % (ns, min(cores, ns), logical, cpu_sketch_s, ef, cpu_search_s),
                           "note": "CPU restatement (oracle), not upstream gsearch: the Rust reference cannot be built here"}
    return out, oix


def redundant_leg(args, ctx, hn, lib, chk, torch, sketch_dev, d_rs, d_rl, d_goff, gbytes, oix, headline_ms):
    """`request` on the data gsearch is run on (GTDB / NCBI prokaryotes, /root/reference/README.md:134: thousands of near-identical genomes per
    species): the same database, but the 10 000 queries of a step are isolates of only `--redundant-roots` of its families (~330 per family
    instead of ~3). Same code path, same parameters; sketch + search timed like a headline step (wall clock around the calls, inputs resident)."""
    import ctypes as C
    k, m, L, qps, knbn, ef = args.kmer, args.sketch_size, args.genome_len, args.queries_per_step, args.knbn, args.ef_search
    R = args.redundant_roots
    d_seq = ctx.alloc(qps * gbytes + 64)
    d_sig = ctx.alloc(qps * m * 4)
    ids = torch.empty((qps, knbn), dtype=torch.int64, device="cuda"); dist = torch.empty((qps, knbn), dtype=torch.float32, device="cuda")
    cnt = torch.empty((qps,), dtype=torch.int32, device="cuda"); ev = torch.zeros((qps,), dtype=torch.int64, device="cuda")
    try:
        steps = []
        for i in range(3):                                             # step 0 warms up; every step has its own query genomes
            chk(lib.gs_synth_dna_family_dev(ctx.h, args.seed, 2_000_000_000 + i * qps, qps, L, R, 0.001, 0.08, d_seq))
            ctx.sync()
            ctx.profile(True)
            for fam in range(4):
                ctx.profile_read(fam, reset=True)
            hn.search_stats(reset=True)
            t0 = time.perf_counter()
            sketch_dev(d_seq, qps, d_sig, d_rs, d_rl, d_goff)
            chk(lib.gs_index_parallel_search_dev(hn.h, d_sig, qps, knbn, ef, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(), ev.data_ptr()))
            ctx.sync()
            dt = time.perf_counter() - t0
            srch, prod, sk = ctx.profile_read(2, reset=True), ctx.profile_read(1, reset=True), ctx.profile_read(0, reset=True)
            ctx.profile(False)
            st = hn.search_stats(reset=True)
            if i:
                steps.append({"ms": dt * 1e3, "sketch_ms": sk[0], "producer_ms": prod[0], "producer_launches": prod[1], "traversal_ms": srch[0], "join_atomics": st.get("join_atomics", 0),
                              "join_shared_entry_expansions": st.get("join_shared_expansions", 0)})
        best = min(steps, key=lambda x: x["ms"])
        ns = 32
        qsig = ctx.download(d_sig, (qps, m), np.float32)[:ns]
        oids, odist, ocnt, oev = oix.parallel_search(qsig, knbn, ef, nthreads=min(host_cpu_budget()[1], ns))
        same = bool(np.array_equal(oids, ids.cpu().numpy().view(np.uint64)[:ns]) and np.array_equal(odist.view(np.uint32), dist.cpu().numpy()[:ns].view(np.uint32))
                    and np.array_equal(oev, ev.cpu().numpy().view(np.uint64)[:ns]))
        return {"workload": "request, redundant regime: %d query genomes x %.1f Mbp per step, all isolates of %d of the database's %d families (~%d queries per family; the headline has ~%.1f)"
                            % (qps, L / 1e6, R, max(args.db_genomes // args.per_root, 1), qps // R, qps / max(args.db_genomes // args.per_root, 1)),
                "genomes_per_sec": qps / (best["ms"] * 1e-3), "ms_per_step": best["ms"], "steps": steps, "ms_per_step_over_headline": best["ms"] / headline_ms,
                "ids_distances_evals_equal_oracle_%d_queries" % ns: same,
                "note": "count matrix by the match-join with heavy blocks (gs_join.hip): the (query, node) blocks of a species are written by the compare tile kernel, "
                        "equal keys of a block's queries enter the hash table once; `producer_ms` sums every kernel of the count-matrix producer (HIP events)"}
    finally:
        ctx.free(d_seq); ctx.free(d_sig)


def skewed_leg(args, ctx, lib, chk, d_rs, d_rl, d_goff, gbytes, headline_ms):
    """`request` against a database with SKEWED family sizes (the regime of NCBI / GTDB prokaryotes, /root/reference/README.md:134): the headline's
    parameters, but genome g belongs to family floor(n_roots u^3.5) - the largest family holds ~10 % of the database (>= 30 000 of 300 000), the sizes fall off as a power
    law - and the 10 000 queries of a step are fresh mutants drawn from the same law (~1000 of them isolates of the largest family). Runs LAST, after the headline index has
    been closed (two 300 k indexes with their pair caches do not fit). Timed like a headline step (wall clock around sketch + search, inputs resident); a 16-query sample
    against the oracle on the exported graph; the forced dense strategy and a row-streaming sub-batch beside `auto`."""
    import ctypes as C
    import gsearch_amd as G
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    k, m, L, qps, knbn, ef, N = args.kmer, args.sketch_size, args.genome_len, args.queries_per_step, args.knbn, args.ef_search, args.db_genomes
    n_roots, alpha = max(N // args.per_root, 1), args.skew_alpha
    prm = G.SeqSketcherParams(k, m, "optdens")
    hs = G.Hnsw.new(args.max_nb_conn, 1_500_000, 16, args.ef_construction, G.DistHamming(ctx), dtype=np.float32, seed=args.seed, insert_batch=256, ctx=ctx)
    hs.modify_level_scale(args.scale_modify); hs.set_extend_candidates(True); hs.set_keeping_pruned(False)
    hs._ensure(m)
    chunk = min(args.build_chunk, N)
    d_seq = ctx.alloc(max(chunk, qps) * gbytes + 64)
    d_sig = ctx.alloc(max(chunk, qps) * m * 4)
    bufs = []
    try:
        t_b = time.perf_counter()
        for g0 in range(0, N, chunk):
            n = min(chunk, N - g0)
            chk(lib.gs_synth_dna_family_skew_dev(ctx.h, args.seed + 1, g0, n, L, n_roots, 0.001, 0.08, alpha, d_seq))
            chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, n * gbytes + 64, d_rs, d_rl, n, d_goff, n, d_sig))
            chk(lib.gs_index_parallel_insert_dev(hs.h, d_sig, n))
        ctx.sync()
        build_s = time.perf_counter() - t_b
        d_ids, d_dist, d_cnt, d_ev = ctx.alloc(8 * qps * knbn), ctx.alloc(4 * qps * knbn), ctx.alloc(4 * qps), ctx.alloc(8 * qps)
        bufs = [d_ids, d_dist, d_cnt, d_ev]
        res = {}
        for mode in ("auto", "dense"):
            prev = os.environ.get("GS_DIST_MODE")
            if mode == "dense":
                os.environ["GS_DIST_MODE"] = "dense"
            steps = []
            for i in range(3):                                          # step 0 warms up; every step has its own query genomes
                chk(lib.gs_synth_dna_family_skew_dev(ctx.h, args.seed + 1, 3_000_000_000 + i * qps, qps, L, n_roots, 0.001, 0.08, alpha, d_seq)); ctx.sync()
                ctx.profile(True)
                for fam in range(4):
                    ctx.profile_read(fam, reset=True)
                hs.search_stats(reset=True)
                t0 = time.perf_counter()
                chk(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, qps * gbytes + 64, d_rs, d_rl, qps, d_goff, qps, d_sig))
                chk(lib.gs_index_parallel_search_dev(hs.h, d_sig, qps, knbn, ef, d_ids, d_dist, d_cnt, d_ev)); ctx.sync()
                dt = time.perf_counter() - t0
                srch, prod, sk = ctx.profile_read(2, reset=True), ctx.profile_read(1, reset=True), ctx.profile_read(0, reset=True)
                ctx.profile(False)
                st = hs.search_stats(reset=True)
                if i:
                    steps.append({"ms": dt * 1e3, "sketch_ms": sk[0], "producer_ms": prod[0], "producer_launches": prod[1], "traversal_ms": srch[0],
                                  "join_atomics": int(st.get("join_atomics", 0)), "join_shared_entry_expansions": int(st.get("join_shared_expansions", 0)),
                                  "pops_per_query": st.get("pops", 0) / qps})
            if prev is None:
                os.environ.pop("GS_DIST_MODE", None)
            else:
                os.environ["GS_DIST_MODE"] = prev
            res[mode] = min(steps, key=lambda x: x["ms"])
            if mode == "auto":
                ids = ctx.download(d_ids, (qps, knbn), np.uint64); dist = ctx.download(d_dist, (qps, knbn), np.float32); ev = ctx.download(d_ev, (qps,), np.uint64)
                qsig = ctx.download(d_sig, (qps, m), np.float32)[:16].copy()
        # row streaming on the first 128 queries of the last step's batch (the whole batch would take minutes)
        ng = min(128, qps)
        os.environ["GS_DIST_MODE"] = "gather"
        ctx.sync(); t0 = time.perf_counter()
        chk(lib.gs_index_parallel_search_dev(hs.h, d_sig, ng, knbn, ef, d_ids, d_dist, d_cnt, d_ev)); ctx.sync()
        g_dt = time.perf_counter() - t0
        os.environ.pop("GS_DIST_MODE", None)
        same_g = bool(np.array_equal(ctx.download(d_ids, (ng, knbn), np.uint64), ids[:ng]) and np.array_equal(ctx.download(d_ev, (ng,), np.uint64), ev[:ng]))
        # family sizes as the generator deals them (host twin of synth_root)
        with np.errstate(over="ignore"):
            r = np.arange(N, dtype=np.uint64)
            u = (splitmix_mix(np.uint64(((args.seed + 1) * 31) & MASK64) + r * np.uint64(0xA24BAED4963EE407) + np.uint64(3)) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
        sizes = np.bincount(np.minimum((n_roots * u ** alpha).astype(np.int64), n_roots - 1), minlength=n_roots)
        oix = O.Index(np.float32, m, args.max_nb_conn, args.ef_construction, scale_modify=args.scale_modify, seed=args.seed)
        oix.import_graph(hs.get_data(), hs.export_graph(), view=True)
        oids, odist, _, oev = oix.parallel_search(qsig, knbn, ef, nthreads=min(host_cpu_budget()[1], 16))
        same = bool(np.array_equal(oids, ids[:16]) and np.array_equal(odist.view(np.uint32), dist[:16].view(np.uint32)) and np.array_equal(oev, ev[:16]))
        del oix
        best_forced = min(res["dense"]["ms"], g_dt * 1e3 / ng * qps)
        return {"workload": "request, skewed database: %d genomes x %.1f Mbp in %d families of power-law sizes (family = floor(n u^%.1f): largest %d genomes, median %d), %d query genomes "
                            "per step drawn from the same law" % (N, L / 1e6, n_roots, alpha, int(sizes.max()), int(np.median(sizes)), qps),
                "largest_family": int(sizes.max()), "families_with_one_member_or_none": int((sizes <= 1).sum()), "build_seconds": build_s,
                "ms_per_step": res["auto"]["ms"], "genomes_per_sec": qps / (res["auto"]["ms"] * 1e-3), "ms_per_step_over_headline": res["auto"]["ms"] / headline_ms,
                "auto": res["auto"], "forced_dense": res["dense"], "row_streaming_first_%d_queries_ms" % ng: g_dt * 1e3, "row_streaming_ms_per_step_extrapolated": g_dt * 1e3 / ng * qps,
                "auto_over_better_forced_strategy": res["auto"]["ms"] / best_forced, "row_streaming_same_ids_and_evals": same_g,
                "evals_per_query": float(ev.mean()), "ids_distances_evals_equal_oracle_16_queries": same}
    finally:
        hs.close()
        for p_ in [d_seq, d_sig] + bufs:
            ctx.free(p_)
        chk(lib.gs_ctx_release_scratch(ctx.h))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shard", default="query", choices=["query", "db"], help="multi-GPU decomposition: `query` (north star: database replicated, queries split, one all-gather) or `db` "
                                                                                 "(database split, every rank answers all queries, all-gather + device merge)")
    ap.add_argument("--workload", default="request", choices=["sketch", "request", "c5dist"], help="c5dist: only the configs[4] u64 distance leg (the PMC passes of tools/pmc_bench.sh run it)")
    ap.add_argument("--genomes", type=int, default=10000)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--kmer", type=int, default=21)
    ap.add_argument("--sketch-size", type=int, default=18000)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--cpu-sample", type=int, default=512)
    # request workload (BASELINE configs[2])
    ap.add_argument("--db-genomes", type=int, default=300000)
    ap.add_argument("--queries", type=int, default=20000, help="query genomes per GPU resident in HBM: the steps rotate through queries / queries-per-step distinct sets")
    ap.add_argument("--queries-per-step", type=int, default=10000, help="query genomes per GPU per step (configs[2]: one request of 10k queries)")
    ap.add_argument("--knbn", type=int, default=50)
    ap.add_argument("--ef-search", type=int, default=5000, help="gsearch hard-codes 5000 (src/bin/gsearch.rs:893)")
    ap.add_argument("--max-nb-conn", type=int, default=128)
    ap.add_argument("--ef-construction", type=int, default=1600)
    ap.add_argument("--scale-modify", type=float, default=0.25)
    ap.add_argument("--per-root", type=int, default=100)
    ap.add_argument("--build-chunk", type=int, default=8192)
    ap.add_argument("--c5-proteomes", type=int, default=50000, help="proteomes of the configs[4] sketch leg (BASELINE: 50k)")
    ap.add_argument("--c5-rows", type=int, default=50000, help="index rows of the configs[4] distance leg")
    ap.add_argument("--prob-db-genomes", type=int, default=100000, help="database genomes of the `request_prob` leg: tohnsw + request on ProbMinHash3a (u64) signatures (0 = skip)")
    ap.add_argument("--skew-alpha", type=float, default=3.5, help="`request_skewed` leg: family = floor(n_roots u^alpha); 3.5 puts ~10 %% of the database in the largest family (0 = skip)")
    ap.add_argument("--redundant-roots", type=int, default=30, help="families the queries of the `request_redundant` side leg are drawn from (0 = skip the leg)")
    ap.add_argument("--cpu-sample-queries", type=int, default=128)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the post-run legs (other sketchers, configs[4], gz ingest)")
    ap.add_argument("--ingest-files", type=int, default=4096, help="gz FASTA files of the file-inclusive ingest leg (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle parity sample + CPU baseline leg (profiling passes: ~2 min of host work per run)")
    ap.add_argument("--selftest-launch", action="store_true", help="GPU-free check of the N-rank launch + single all-gather (gloo, stub searcher)")
    args = ap.parse_args()
    from gsearch_amd import sharding as S
    rc = S.ensure_launched(args.gpus, os.path.abspath(__file__), sys.argv[1:])     # --gpus N without a launcher: start the N ranks ourselves
    if rc is not None:
        sys.exit(rc)
    D = Dist("gloo" if args.selftest_launch else "nccl")
    if args.selftest_launch:
        run_selftest_launch(args, D)
    elif args.workload == "sketch":
        run_sketch(args, D)
    elif args.workload == "c5dist":
        import gsearch_amd as G
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        ctx = G.Context(D.local)
        print(json.dumps({"config4_distance_leg": c5_distance_leg(args, ctx, ctx.L, G._lib.check, O)}))
    else:
        run_request(args, D)
    D.finish()


if __name__ == "__main__":
    main()
