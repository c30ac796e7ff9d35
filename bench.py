#!/usr/bin/env python3
"""bench.py — headline benchmark of the sketch-and-query hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sketch|request]

One rank per GPU (launched by torch.distributed.run for N > 1). A *step* is one pass of the hot path over one
batch of synthetic input that is already resident in HBM:

  sketch  (BASELINE.json configs[1]): 10k synthetic 5 Mbp genomes, k=21, s=18000, --algo optdens, per rank.
  request (BASELINE.json configs[2]): queries against a sketch-level synthetic HNSW database.

Rank 0 prints ONE JSON line (metric/value/... plus `roofline` and `cpu_baseline`, see DESIGN.md "measurement").
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


def dist_init(n_gpus):
    import torch
    rank, world, local = 0, 1, 0
    if n_gpus > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return torch, rank, world, local


def barrier_sync(torch, world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(torch, world, value):
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(torch, world, value):
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ------------------------------------------------------------------------------------------------------
def run_sketch(args, torch, rank, world, local):
    import ctypes as C
    import gsearch_amd as G
    from gsearch_amd.api import _p

    k, m, L = args.kmer, args.sketch_size, args.genome_len
    ng = args.genomes                      # per rank (weak scaling)
    ctx = G.Context(local)
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
    first = rank * ng
    G._lib.check(lib.gs_synth_dna_dev(ctx.h, args.seed, first, ng, L, d_seq))
    ctx.sync()

    def step():
        G._lib.check(lib.gs_sketch_batch_dev(ctx.h, C.byref(prm.c), d_seq, seq_bytes + 64, d_rs, d_rl, ng, d_goff, ng, d_sig))

    for _ in range(args.warmup):
        step()
    ctx.profile(True)
    ctx.profile_read(0, reset=True)
    barrier_sync(torch, world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier_sync(torch, world)
    dt = max_over_ranks(torch, world, time.perf_counter() - t0)
    kern_ms, kern_n = ctx.profile_read(0, reset=True)
    ctx.profile(False)

    kmers_per_genome = L - k + 1
    total_kmers = float(kmers_per_genome) * ng * world * args.steps
    value = total_kmers / dt
    out = {
        "metric": "sketch k-mers/sec", "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "sketch-only: %d synthetic %.1f Mbp genomes per GPU, k=%d s=%d optdens (BASELINE configs[1])" % (ng, L / 1e6, k, m),
                   "genomes_per_gpu": ng, "genome_len": L, "kmer_size": k, "sketch_size": m, "algo": "optdens"},
        "genomes_per_sec": ng * world * args.steps / dt,
    }
    if rank == 0:
        # roofline of the dominant kernel (k_sketch_oph): algorithmic bytes = ceil(L/4) + m*4 per genome
        alg_bytes_launch = (float((L + 3) // 4) + m * 4.0) * ng
        avg_ms = kern_ms / max(kern_n, 1)
        achieved = alg_bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": None, "kernel": "k_sketch_oph", "avg_launch_ms": avg_ms, "launches": kern_n,
                           "algorithmic_bytes_per_launch": alg_bytes_launch,
                           "note": "hash-bound kernel: 0.264 B/k-mer of HBM traffic against ~250 VALU ops/k-mer (DESIGN.md)",
                           "kmers_per_sec_kernel": kmers_per_genome * ng / (avg_ms * 1e-3) if avg_ms > 0 else 0.0}
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
        cores = os.cpu_count() or 1
        ns = args.cpu_sample
        pk = [synth_genome_packed(args.seed, i, L) for i in range(min(ns, 8))]
        # the sample re-uses 8 distinct genomes round-robin: identical work per genome, bounded host memory
        wpad = words * 8
        buf = np.zeros(len(pk) * wpad + 16, np.uint8)
        for i, b in enumerate(pk):
            buf[i * wpad:(i + 1) * wpad] = b
        srs = (np.arange(ns, dtype=np.uint64) % np.uint64(len(pk))) * np.uint64(words * 32)
        srl = np.full(ns, L, np.uint64)
        t0 = time.perf_counter()
        O.sketch_batch(op, buf, srs, srl, np.arange(ns + 1, dtype=np.uint64), nthreads=cores)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": kmers_per_genome * ns / cdt, "unit": "k-mers/s", "cores": cores, "kind": "port",
                               "sample": "%d genomes x %.1f Mbp, k=%d s=%d optdens, oracle/gs_oracle.c with OpenMP (one task per genome), %.1f s wall"
                                         % (ns, L / 1e6, k, m, cdt),
                               "note": "CPU restatement (oracle), not upstream gsearch: the Rust reference cannot be built here"}
        print(json.dumps(out))
    for p in (d_seq, d_sig, d_rs, d_rl, d_goff):
        ctx.free(p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="sketch", choices=["sketch", "request"])
    ap.add_argument("--genomes", type=int, default=10000)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--kmer", type=int, default=21)
    ap.add_argument("--sketch-size", type=int, default=18000)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--cpu-sample", type=int, default=192)
    args = ap.parse_args()
    torch, rank, world, local = dist_init(args.gpus)
    if args.workload == "sketch":
        run_sketch(args, torch, rank, world, local)
    else:
        raise SystemExit("request workload: see bench_request (not wired yet)")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
