"""Host-side file helpers of the ingest path (no GPU): accepted names (files.rs:117-146), transparent gz / bz2 / xz decoding like
needletail (files.rs:220-250), recursive directory walk (files.rs:148-215)."""
import bz2
import gzip
import lzma
import os

import numpy as np
import pytest


def test_accepted_file_names_follow_files_rs():
    import gsearch_amd as G
    yes_dna = ["a.fna", "a.fa", "a.fasta", "x/y.fna.gz", "a.fa.gz", "a.fasta.gz", "a.fa.xz", "a.fna.xz", "a.fasta.xz", "a.fa.bz2", "a.fna.bz2", "a.fasta.bz2"]
    no_dna = ["a.faa", "a.txt", "a.fna.zst", "a.fastq", "a.gz", "fna.tar"]
    for f in yes_dna:
        assert G.is_fasta_file(f, "dna"), f
    for f in no_dna:
        assert not G.is_fasta_file(f, "dna"), f
    for f in ["p.faa", "p.faa.gz", "p.faa.xz", "p.faa.bz2"]:
        assert G.is_fasta_file(f, "aa") and not G.is_fasta_file(f, "dna"), f
    assert not G.is_fasta_file("a.fna", "aa")


def test_read_fasta_file_decodes_gz_bz2_xz(tmp_path):
    import gsearch_amd as G
    rng = np.random.default_rng(1)
    text = b">r1 desc\n" + bytes(np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, 300000)]) + b"\n>r2\nACGT\n"
    (tmp_path / "p.fna").write_bytes(text)
    (tmp_path / "g.fna.gz").write_bytes(gzip.compress(text))
    (tmp_path / "mm.fna.gz").write_bytes(gzip.compress(text[:1000]) + gzip.compress(text[1000:]))       # multi-member (bgzip style)
    (tmp_path / "b.fna.bz2").write_bytes(bz2.compress(text))
    (tmp_path / "x.fna.xz").write_bytes(lzma.compress(text))
    (tmp_path / "empty.fna").write_bytes(b"")
    for name in ("p.fna", "g.fna.gz", "mm.fna.gz", "b.fna.bz2", "x.fna.xz"):
        assert G.read_fasta_file(tmp_path / name) == text, name
    assert G.read_fasta_file(tmp_path / "empty.fna") == b""
    (tmp_path / "bad.fna.gz").write_bytes(gzip.compress(text)[:-200])
    with pytest.raises(G.GsError) as e:
        G.read_fasta_file(tmp_path / "bad.fna.gz")
    assert e.value.code == -5
    with pytest.raises(G.GsError):
        G.read_fasta_file(tmp_path / "missing.fna")


def test_directory_walk_is_recursive_and_filtered(tmp_path):
    import gsearch_amd as G
    for rel in ("b/z.fna", "a.fa", "b/c/deep.fasta.gz", "b/notes.txt", "p.faa", "b/q.faa.gz"):
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_bytes(b">x\nACGT\n")
    got = [os.path.relpath(p, tmp_path) for p in G.list_fasta_files(tmp_path, "dna")]
    assert got == ["a.fa", "b/z.fna", "b/c/deep.fasta.gz"]
    assert [os.path.relpath(p, tmp_path) for p in G.list_fasta_files(tmp_path, "aa")] == ["p.faa", "b/q.faa.gz"]
    with pytest.raises(G.GsError):
        G.list_fasta_files(tmp_path / "nope")
