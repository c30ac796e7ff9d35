#!/usr/bin/env python3
"""Differential run of the file path (gs_sketch_files: reader threads, gzip on host or device, record scan, device filter + 2-bit pack, sketch) against an INDEPENDENT Python reading of the
same FASTA files + the oracle's encode + sketch: random records (headers with descriptions, `capsid` anywhere in the header, empty and shorter-than-k records), line widths, CRLF,
lower case, N / IUPAC runs, blank lines, no trailing newline, plain / gzip (levels, multi-member, bgzip-like blocks) files, by-sequence and --block modes, DNA and amino acids.
usage: ingest_fuzz.py [rounds] [seed]"""
import gzip, os, sys, tempfile, shutil, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsearch_amd as G
import helpers as H
import oracle_lib as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def py_records(text):
    """needletail's reading as the reference uses it (dnafiles.rs:52-70): header = the line behind '>', sequence = the following lines without their ends; records whose header
    contains `capsid` and empty sequences are dropped"""
    recs, cur, hdr = [], None, None
    for line in text.split(b"\n"):
        if line.endswith(b"\r"):
            line = line[:-1]
        if line.startswith(b">"):
            if cur is not None:
                recs.append((hdr, b"".join(cur)))
            hdr, cur = line[1:], []
        elif cur is not None:
            cur.append(line)
    if cur is not None:
        recs.append((hdr, b"".join(cur)))
    return [s for h, s in recs if b"capsid" not in h and len(s) > 0]


bad = 0
for rnd in range(rounds):
    data = "aa" if rng.random() < 0.3 else "dna"
    k = int(rng.choice([12, 16, 21])) if data == "dna" else int(rng.choice([5, 7]))
    m = int(rng.choice([64, 500, 2000]))
    block = bool(rng.random() < 0.3)
    d = tempfile.mkdtemp(prefix="gs_ingest_fuzz_", dir="/tmp")
    try:
        paths, texts = [], []
        for f in range(int(rng.integers(3, 40))):
            out = []
            for r in range(int(rng.integers(1, 30))):
                n = int(rng.choice([0, 3, k - 1, k, 50, 2000, 40000, 300000]))
                s = H.dna_ascii(H.rand_dna(rng, n)) if data == "dna" else H.aa_ascii(rng.integers(0, 20, n))
                if n > 100 and rng.random() < 0.3:
                    at = int(rng.integers(0, n - 20)); s = s[:at] + (b"NNNNNRYKMnnn" if data == "dna" else b"XXBZ*-") + s[at:]
                if n > 100 and rng.random() < 0.3:
                    at = int(rng.integers(0, n - 50)); s = s[:at] + s[at:at + 40].lower() + s[at + 40:]
                name = [b"contig%d" % r, b"NZ_CP0%05d.1 Escherichia coli strain K-12 chromosome" % r, b"phage capsid protein %d" % r, b"gene|capsid_%d" % r, b"x"][int(rng.integers(0, 5))]
                width = int(rng.choice([60, 70, 80, 100, 10 ** 9]))
                out.append(b">" + name + b"\n")
                out += [s[o:o + width] + b"\n" for o in range(0, len(s), width)]
                if rng.random() < 0.1:
                    out.append(b"\n")                                # a blank line between records
            t = b"".join(out)
            if rng.random() < 0.2:
                t = t.replace(b"\n", b"\r\n")
            if rng.random() < 0.2 and t.endswith(b"\n"):
                t = t[:-1]                                        # no newline at the end of the file
            kind = int(rng.integers(0, 4))
            ext = ".fna" if data == "dna" else ".faa"
            p = os.path.join(d, "f%03d%s" % (f, ext + (".gz" if kind else "")))
            if kind == 0:
                open(p, "wb").write(t)
            elif kind == 1:
                open(p, "wb").write(gzip.compress(t, int(rng.integers(1, 10))))
            elif kind == 2:                                          # several members back to back
                cut = sorted(int(x) for x in rng.integers(0, len(t) + 1, 3))
                open(p, "wb").write(b"".join(gzip.compress(t[a:b], 6) for a, b in zip([0] + cut, cut + [len(t)])))
            else:                                                    # bgzip-like: many small members
                open(p, "wb").write(b"".join(gzip.compress(t[a:a + 65280], 6) for a in range(0, max(len(t), 1), 65280)))
            paths.append(p); texts.append(t)
        sk = G.sketcher_for(G.SeqSketcherParams(k, m, "optdens", data))
        sig, nrec, nsym, st = sk.sketch_files(paths, block=block, pio=int(rng.choice([0, 3, 8])), threads=int(rng.choice([0, 2, 5])))
        genomes = []
        for t in texts:
            recs = py_records(t)
            genomes.append([b"".join(recs)] if block and recs else recs)
        if block and data == "dna":       # --block: the bases of a file form ONE sequence - but non-ACGT bytes are dropped per record first, which joining the text keeps
            pass
        flat = [r for g in genomes for r in g]
        goff = np.cumsum([0] + [len(g) for g in genomes]).astype(np.uint64)
        seq, rs, rl = (O.pack_dna(flat) if data == "dna" else O.filter_aa(flat)) if flat else (np.zeros(16, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        ref = O.sketch_batch(O.params(k, m, "optdens", data), seq, rs, rl, goff, nthreads=os.cpu_count())
        wrong = np.nonzero((sig.view(np.uint32) != ref.view(np.uint32)).any(axis=1))[0].tolist()
        bad += len(wrong)
        print("round %d %s k=%d m=%d block=%d: %d files, wrong files %s" % (rnd, data, k, m, block, len(paths), [(i, os.path.basename(paths[i])) for i in wrong[:6]]), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
print("%d rounds, %d wrong files" % (rounds, bad))
sys.exit(1 if bad else 0)
