#!/usr/bin/env python3
"""PCIe-inclusive ingest rate: synthetic FASTA files on local disk -> gs_sketch_files (host threads read / decode /
scan, pinned double-buffered H2D, device filter + 2-bit pack, sketch) next to the HBM-resident sketch rate of the same genomes.
usage: ingest_rate.py [n_files] [genome_len] [plain|gz] [pio] [threads] [gzip_level] [n_distinct]
(n_distinct > 0: that many genomes are generated and compressed, the other files are copies - the decoders do the same work per file)"""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsearch_amd as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
pio = int(sys.argv[4]) if len(sys.argv) > 4 else 64
threads = int(sys.argv[5]) if len(sys.argv) > 5 else 0
level = int(sys.argv[6]) if len(sys.argv) > 6 else 1
ndist = int(sys.argv[7]) if len(sys.argv) > 7 else 0
d = tempfile.mkdtemp(prefix="gs_ingest_", dir="/tmp")
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", np.uint8)
t0 = time.perf_counter()
paths = []
for i in range(ndist if ndist else n):
    seq = acgt[rng.integers(0, 4, L)]
    lines = np.concatenate([seq.reshape(-1, 80) if L % 80 == 0 else np.resize(seq, (L // 80 + 1, 80)), np.full((L // 80 + (L % 80 != 0), 1), 10, np.uint8)], axis=1)
    p = os.path.join(d, "g%05d.fna" % i)
    with open(p, "wb") as f:
        f.write(b">genome%d synthetic\n" % i)
        f.write(lines.tobytes())
    paths.append(p)
if mode == "bgzf":                                              # bgzip-style members of 64 KB (the image has no bgzip: written here)
    from multiprocessing import Pool
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import helpers as H
    def _bg(p):
        open(p + ".gz", "wb").write(H.bgzf_bytes(open(p, "rb").read(), level=level)); os.unlink(p); return p + ".gz"
    with Pool(min(os.cpu_count() or 8, 16)) as pool:
        paths = pool.map(_bg, paths)
if mode == "gz":
    subprocess.check_call("ls %s/*.fna | xargs -P %d -n 4 gzip -%d" % (d, os.cpu_count() or 8, level), shell=True)
    paths = [p + ".gz" for p in paths]
if ndist:
    import shutil
    base = list(paths)
    for i in range(ndist, n):
        q = os.path.join(d, "c%05d" % i + base[0][base[0].index(".fna"):])
        os.link(base[i % ndist], q)          # same inode: no extra disk or page cache, the readers still copy and decode every file
        paths.append(q)
raw_bytes = sum(os.path.getsize(p) for p in paths)
print("wrote %d files (%s, %.2f GB on disk) in %.1fs" % (n, mode, raw_bytes / 1e9, time.perf_counter() - t0), flush=True)
sk = G.OptDensHashSketch.new(G.SeqSketcherParams(21, 18000, "optdens"))
for rep in range(2):
    t0 = time.perf_counter()
    sig, nrec, nsym, st = sk.sketch_files(paths, pio=pio, threads=threads)
    dt = time.perf_counter() - t0
    print("rep %d: %d files x %.1f Mbp (%s) in %.2fs -> %.0f genomes/s, %.2f GB/s of FASTA text | host read+decode+scan %.2fs (thread-seconds), waited for PCIe %.3fs, device pack+sketch %.2fs"
          % (rep, n, L / 1e6, mode, dt, n / dt, int(nsym.sum()) * 81 / 80 / dt / 1e9, st["host_read_decode_scan_s"], st["pcie_wait_s"], st["device_s"]), flush=True)
subprocess.call(["rm", "-rf", d])
