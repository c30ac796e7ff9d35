"""Device-side gzip (gs_inflate.hip, the .gz path of gs_sketch_files) against zlib: the text a member inflates to must be byte-identical
for every block type and table shape the encoders in this image can produce, and damaged members must be reported, not decoded."""
import gzip
import os
import io
import zlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, memlevel, strategy)
    return c.compress(data) + c.flush()


def _fasta(rng, n, width=80):
    s = H.dna_ascii(H.rand_dna(rng, n))
    return b">seq%d some description\n" % n + b"\n".join(s[o:o + width] for o in range(0, len(s), width)) + b"\n"


def _cases():
    rng = np.random.default_rng(77)
    dna = _fasta(rng, 300_000)
    text = (b"the quick brown fox jumps over the lazy dog; " * 3000) + bytes(rng.integers(32, 127, 40_000, dtype=np.uint8))
    skew = bytes(np.minimum(rng.geometric(0.08, 200_000), 255).astype(np.uint8))          # long Huffman codes (> 10 bits) for the rare bytes
    cases = {
        "empty": _gz(b""),
        "one_byte": _gz(b"A"),
        "hello_fixed": _gz(b"hello hello hello", 6, zlib.Z_FIXED),
        "dna_l1": _gz(dna, 1), "dna_l6": _gz(dna, 6), "dna_l9": _gz(dna, 9),
        "dna_fixed": _gz(dna[:50_000], 6, zlib.Z_FIXED),
        "dna_huffman_only": _gz(dna[:100_000], 6, zlib.Z_HUFFMAN_ONLY),
        "dna_rle": _gz(dna[:100_000], 6, zlib.Z_RLE),
        "dna_memlevel1": _gz(dna[:120_000], 6, zlib.Z_DEFAULT_STRATEGY, 1),               # tiny blocks: hundreds of dynamic tables
        "text": _gz(text, 9),
        "skewed_bytes": _gz(skew, 9),
        "run_of_A": _gz(b"A" * 100_000 + b"CG" * 50_000 + b"ACGTT" * 30_000),            # overlapping copies: dist 1, 2, 5 < len 258
        "random_stored": _gz(bytes(rng.integers(0, 256, 150_000, dtype=np.uint8)), 6),     # incompressible -> stored blocks
        "level0_stored": _gz(dna[:70_000], 0),
        "window_edge": _gz(bytes(rng.integers(0, 256, 32_768, dtype=np.uint8)) * 3, 9),    # matches at distance 32768
        # copies whose source lies in text that the pipelined form has not stored yet (destinations of the batch in flight / being decoded):
        # two-symbol noise (short distances, every length), short periods, and mutated repeats a few dozen bytes back
        "two_symbols": _gz(bytes(rng.integers(0, 2, 200_000, dtype=np.uint8) + 65), 6),
        "period_7_and_13": _gz(bytes(rng.integers(65, 91, 7, dtype=np.uint8)) * 9000 + bytes(rng.integers(65, 91, 13, dtype=np.uint8)) * 5000, 9),
        "near_repeats": _gz(b"".join(bytes(rng.integers(65, 69, 24, dtype=np.uint8)) * int(rng.integers(2, 5)) for _ in range(4000)), 6),
        "protein": _gz(b">p1\n" + bytes(rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8), 200_000)) + b"\n", 6),
    }
    bio = io.BytesIO()
    with gzip.GzipFile(filename="genome_with_a_name.fna", mode="wb", fileobj=bio, mtime=12345) as f:       # FNAME header field
        f.write(dna[:80_000])
    cases["fname_header"] = bio.getvalue()
    return cases


@pytest.mark.parametrize("window", ["lds", "global", "pipe"])
def test_inflate_matches_zlib(gpu_ctx, monkeypatch, window):
    """the forms of k_inflate: history window in LDS (up to four members per CU) / no window, history read back from the text in HBM, one match
    at a time (global) or several in flight (pipe)"""
    import gsearch_amd as G
    monkeypatch.setenv("GS_INFLATE_WINDOW", window)
    cases = _cases()
    names = list(cases)
    res = G.gunzip_batch(gpu_ctx, [cases[k] for k in names])
    for k, (st, text) in zip(names, res):
        want = zlib.decompress(cases[k], 31)
        assert st == 0, (k, st)
        assert text == want, k


@pytest.mark.parametrize("window", ["lds", "global", "pipe", ""])
def test_inflate_large_members_and_many_streams(gpu_ctx, monkeypatch, window):
    """more streams than the device holds at once (4 per CU), each several window wraps long; "" = the launcher's own choice"""
    import gsearch_amd as G
    if window:
        monkeypatch.setenv("GS_INFLATE_WINDOW", window)
    rng = np.random.default_rng(5)
    members, wants = [], []
    for i in range(40):
        t = _fasta(rng, 150_000 + 7919 * i, 60 + i)
        wants.append(t)
        members.append(_gz(t, 1 + i % 9))
    big = _fasta(rng, 5_000_000)
    wants.append(big); members.append(_gz(big, 1))
    res = G.gunzip_batch(gpu_ctx, members * 30)
    for j, (st, text) in enumerate(res):
        assert st == 0, (j, st)
        assert text == wants[j % len(wants)], j


@pytest.mark.parametrize("window", ["lds", "global", "pipe"])
def test_inflate_reports_damage(gpu_ctx, monkeypatch, window):
    import gsearch_amd as G
    monkeypatch.setenv("GS_INFLATE_WINDOW", window)
    rng = np.random.default_rng(9)
    t = _fasta(rng, 200_000)
    good = _gz(t, 6)
    flipped = bytearray(good); flipped[len(good) // 2] ^= 0x10
    bad_crc = bytearray(good); bad_crc[-6] ^= 0xFF
    bad_isize = bytearray(good); bad_isize[-1] ^= 0x01
    cut = good[:len(good) // 2] + good[-8:]                   # half the deflate data, then a trailer that promises the whole text
    cases = [good, bytes(flipped), good[:len(good) // 2], bytes(bad_crc), good + good, b"not gzip at all, just text" * 4, bytes(bad_isize), cut]
    caps = [len(t)] * len(cases)
    res = G.gunzip_batch(gpu_ctx, cases, out_caps=caps)
    st = [r[0] for r in res]
    assert st[0] == 0 and res[0][1] == t
    assert st[1] != 0                     # a flipped bit: malformed data, or a text whose CRC does not check
    assert st[2] != 0                     # truncated
    assert st[3] == 103
    assert st[4] == 101 and res[4][1] == t      # a second member follows: the caller takes the host path for such files
    assert st[5] == 100
    assert st[6] in (102, 104)
    assert st[7] != 0 and res[7][1] != t
    # out_cap below ISIZE is refused before anything is decoded
    res = G.gunzip_batch(gpu_ctx, [good], out_caps=[len(t) - 1])
    assert res[0][0] == 104


@pytest.mark.parametrize("data,block", [("dna", False), ("dna", True), ("aa", False)])
def test_sketch_files_device_gzip_equals_host_gzip(gpu_ctx, tmp_path, monkeypatch, data, block):
    """.gz files through gs_sketch_files: members inflated on the device (records found by the device scan) must give the signatures,
    record counts and symbol counts of the host decoders (GS_GZIP_DEVICE=0) - multi-record files, CRLF, a `capsid` header, leading junk
    lines, no final newline, a named member, and a two-member file that the device path hands back to the host"""
    import gsearch_amd as G
    rng = np.random.default_rng(31)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)

    def seq(n):
        return H.dna_ascii(H.rand_dna(rng, n)) if data == "dna" else bytes(rng.choice(aa, n))

    def fasta(records, width, nl=b"\n", final=True):
        out = []
        for name, s in records:
            out.append(b">" + name + nl)
            out += [s[o:o + width] + nl for o in range(0, len(s), width)]
        t = b"".join(out)
        return t if final else t[:-len(nl)]

    ext = ".fna.gz" if data == "dna" else ".faa.gz"
    files = []
    for i in range(83):              # >= 64 .gz files: dealt between the host pipeline and the device pipeline
        recs = [(b"r%d_%d some text" % (i, j), seq(int(rng.integers(200, 40000)))) for j in range(1 + i % 5)]
        if i % 7 == 3:
            recs.insert(1, (b"x phage capsid protein", seq(900)))
        text = fasta(recs, 60 + i, b"\r\n" if i % 6 == 1 else b"\n", final=(i % 4 != 2))
        if i % 9 == 4:
            text = b"; a comment line before the first record\n\n" + text
        files.append(text)
    blobs = [_gz(t, 1 + i % 9) for i, t in enumerate(files)]
    blobs[5] = _gz(files[5][:3000], 6) + _gz(files[5][3000:], 6)                       # two members
    bio = io.BytesIO()
    with gzip.GzipFile(filename="named.fa", mode="wb", fileobj=bio, mtime=1) as f:
        f.write(files[6])
    blobs[6] = bio.getvalue()
    blobs[7] = _gz(b"")                                                                 # empty text
    paths = []
    for i, b in enumerate(blobs):
        (tmp_path / ("f%02d%s" % (i, ext))).write_bytes(b)
        paths.append(tmp_path / ("f%02d%s" % (i, ext)))
    prm = G.SeqSketcherParams(21, 1200, "optdens") if data == "dna" else G.SeqSketcherParams(7, 800, "optdens", data_t="aa")
    sk = G.sketcher_for(prm)
    monkeypatch.setenv("GS_GZIP_DEVICE", "0")
    sig0, nrec0, nsym0, _ = sk.sketch_files(paths, block=block, pio=16, threads=4)
    monkeypatch.setenv("GS_GZIP_DEVICE", "1")
    sig1, nrec1, nsym1, _ = sk.sketch_files(paths, block=block, pio=16, threads=4)
    sig2, nrec2, nsym2, _ = sk.sketch_files(paths, block=block, pio=0, threads=4)      # one big device group
    assert list(nrec0) == list(nrec1) == list(nrec2) and nrec0[7] == 0 and sum(nrec0) > 60
    assert list(nsym0) == list(nsym1) == list(nsym2)
    assert np.array_equal(_bits(sig0), _bits(sig1)) and np.array_equal(_bits(sig0), _bits(sig2))


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_sketch_files_bgzf_members_on_the_device(gpu_ctx, tmp_path):
    """bgzip-compressed FASTA (needletail reads it like any multi-member .gz, /root/reference/src/utils/files.rs:258-341): every <= 64 KB member is
    inflated by its own wavefront, a file's text is its members end to end; ordinary multi-member and single-member files beside them.
    Signatures == the same texts read as plain files."""
    import gzip
    import gsearch_amd as G
    import helpers as H
    rng = np.random.default_rng(11)
    texts = []
    for i in range(24):
        L = int(rng.integers(150_000, 700_000))
        seq = H.dna_ascii(H.rand_dna(rng, L))
        if i % 5 == 0:
            seq = seq[:70_000] + b"NNNNNNNNNN" + seq[70_000:200_000].lower() + seq[200_000:]
        lines = b"\n".join(seq[j:j + 70] for j in range(0, len(seq), 70))
        if i % 4 == 1:                                              # several records, one of them a capsid (skipped)
            cut = len(lines) // 3
            lines = lines[:cut] + b"\n>rec2 capsid protein\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n>rec3\n" + lines[cut:]
        texts.append(b">g%d synthetic\n" % i + lines + b"\n")
    texts.append(b">empty\n")
    texts.append(b"")
    plain, comp = [], []
    for i, t in enumerate(texts):
        pp = tmp_path / ("p%03d.fna" % i); pp.write_bytes(t); plain.append(str(pp))
        cp = tmp_path / ("c%03d.fna.gz" % i)
        if i % 6 == 2: cp.write_bytes(gzip.compress(t[:len(t) // 2], 6) + gzip.compress(t[len(t) // 2:], 6))      # plain multi-member: host decoders
        elif i % 6 == 4: cp.write_bytes(gzip.compress(t, 6))                                                        # single member
        else: cp.write_bytes(H.bgzf_bytes(t, block=int(rng.integers(20_000, 65281))))
        comp.append(str(cp))
    sk = G.OptDensHashSketch.new(G.SeqSketcherParams(21, 2000, "optdens"))
    want, nrec_w, nsym_w, _ = sk.sketch_files(plain)
    got, nrec, nsym, st = sk.sketch_files(comp)
    assert np.array_equal(nrec, nrec_w) and np.array_equal(nsym, nsym_w)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert st["gz_members_handed_back_to_host"] == 4             # the four ordinary multi-member files; every bgzip file stayed on the device
    bl = sk.sketch_files(comp, block=True)[0]
    assert np.array_equal(bl.view(np.uint32), sk.sketch_files(plain, block=True)[0].view(np.uint32))


def test_inflate_and_ingest_differential_runs():
    """round 6: random members against zlib through every form of the device decoder, with bit flips that must never come back as "status 0, other bytes"
    (tools/inflate_fuzz.py), and random FASTA files - headers with `capsid`, empty / short records, CRLF, blank lines, no final newline, plain / gzip / multi-member /
    bgzip-like, by-sequence and --block, DNA and amino acids - through gs_sketch_files against an independent Python reading of the files + the oracle (tools/ingest_fuzz.py)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = subprocess.run([sys.executable, "-u", os.path.join(root, "tools", "inflate_fuzz.py"), "2", "3"], capture_output=True, text=True, timeout=900)
    assert a.returncode == 0 and "2 rounds, 0 failures" in a.stdout, (a.stdout[-1500:], a.stderr[-800:])
    b = subprocess.run([sys.executable, "-u", os.path.join(root, "tools", "ingest_fuzz.py"), "5", "3"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0 and "5 rounds, 0 wrong files" in b.stdout, (b.stdout[-1500:], b.stderr[-800:])
