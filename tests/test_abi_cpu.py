"""CPU-side tests of the C-ABI library: it loads, exports every symbol include/gsearch_amd.h declares, validates
parameters like the reference does, and fails LOUDLY (no fallback) when no GPU is present. No compute is launched."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import gsearch_amd as G
    hdr = open(os.path.join(ROOT, "include", "gsearch_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 40
    lib = C.CDLL(G.SO_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(G.SYMBOLS), (declared ^ set(G.SYMBOLS))


def test_parameter_validation_mirrors_reference():
    import gsearch_amd as G
    P = G.SeqSketcherParams
    assert P(21, 18000, "optdens").sig_dtype() == np.float32
    assert P(16, 1000, "prob").sig_dtype() == np.uint32
    assert P(17, 1000, "prob").sig_dtype() == np.uint64
    assert P(7, 24000, "super2", "aa").sig_dtype() == np.uint64           # BASELINE configs[4]
    for args in ((15, 100, "optdens"), (33, 100, "optdens"), (0, 100, "optdens"), (13, 100, "super2", "aa"), (21, 1, "optdens")):
        with pytest.raises(G.GsError) as e:
            P(*args)
        assert e.value.code == -1
    assert P(21, 18000, "hll").sig_dtype() == np.uint16                   # HyperLogLogSketch<Kmer, u16> (dnasketch.rs:541-574)
    assert P(21, 90000, "hll").sig_dtype() == np.uint16                   # sketch_size is taken as is (register file beyond LDS: global table)


def test_host_helpers_match_oracle():
    import gsearch_amd as G
    import oracle_lib as O
    recs = [b"ACGTNNacgtRYKM", b"", b"TTTTGGGGCCCCAAAA" * 5, b"N"]
    seq, rs, rl = G.pack_dna_records(recs)
    oseq, ors, orl = O.pack_dna(recs)
    assert np.array_equal(rs, ors) and np.array_equal(rl, orl)
    n = int((rs[-1] + rl[-1] + 3) // 4)
    assert np.array_equal(seq[:n], oseq[:n])
    aa, s, l = G.filter_aa_records([b"MKV*LLxz", b"acdef"])
    oaa, os_, ol = O.filter_aa([b"MKV*LLxz", b"acdef"])
    assert np.array_equal(l, ol) and bytes(aa[:int(l.sum())]) == bytes(oaa[:int(ol.sum())]) == b"MKVLLACDEF"
    for d in (0.0, 0.54, 0.9):
        assert abs(G.ani(d, 16, 1) - O.ani(d, 16, 1)) < 1e-12 and abs(G.ani(d, 16, 2) - O.ani(d, 16, 2)) < 1e-12


def test_no_silent_cpu_fallback():
    """without a GPU every compute entry point must refuse; with one this test is skipped"""
    import gsearch_amd as G
    try:
        ctx = G.Context(0)
    except G.GsError as e:
        assert e.code == -2 and "device" in str(e).lower()
        with pytest.raises(G.GsError):
            G.DistHamming().eval([1.0, 2.0], [1.0, 3.0])
        return
    ctx.close()
    pytest.skip("a GPU is present")


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gsearch_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                for pat in (r"#\s*include[^\n]*oracle", r"import\s+oracle", r"from\s+oracle", r"libgs_oracle", r"oracle_lib", r"dlopen[^\n]*oracle"):
                    assert not re.search(pat, txt), (os.path.join(dirpath, f), pat)


def test_fasta_scan_capsid_filter_uses_the_whole_header_line():
    """needletail's id() is the whole header line: dnafiles.rs:62-67 drops a record whose DESCRIPTION says capsid (host-only code)"""
    import gsearch_amd as G
    txt = (b">NC_0123.1 Foo virus capsid protein\nACGT\nAC\n>NC_2 major Capsid (upper case is kept)\nGG\n>x_capsid_y\nTT\n"
           b">NC_3 tail fibre\r\nACGTA\r\n>NC_4 ends with capsid\r\nCC\r\n>last\nA")
    recs = G.fasta_scan(txt)
    assert [r[0] for r in recs] == ["NC_2", "NC_3", "last"]
    assert [txt[b:e] for _, b, e in recs] == [b"GG\n", b"ACGTA\r\n", b"A"]
    assert [r[0] for r in G.fasta_scan(txt, skip_capsid=False)] == ["NC_0123.1", "NC_2", "x_capsid_y", "NC_3", "NC_4", "last"]


def test_integration_md_binds_every_exported_symbol():
    """INTEGRATION.md's Rust `extern "C"` block (the binding a maintainer of the reference would add) names every function the header
    declares, with the generated signature (tools/gen_rust_extern.py), and the ctypes table of the Python host does the same"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rust_extern", os.path.join(root, "tools", "gen_rust_extern.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    md = " ".join(open(os.path.join(root, "INTEGRATION.md")).read().split())
    import gsearch_amd as G
    names = []
    for name, params, ret in gen.functions():
        names.append(name)
        decl = "pub fn %s(%s)%s;" % (name, ", ".join("%s: %s" % p for p in params), (" -> " + ret) if ret else "")
        assert " ".join(decl.split()) in md, decl
    assert sorted(names) == sorted(G.SYMBOLS), set(names) ^ set(G.SYMBOLS)


def test_every_environment_knob_is_documented():
    """every GS_* environment variable the library or its Python host reads appears in INTEGRATION.md section 5 with a class
    (D diagnostic / S strategy override / A A-B switch / T tuning parameter)"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    knobs = set()
    for fn in glob.glob(os.path.join(root, "gsearch_amd", "csrc", "*.h*")):
        knobs |= set(re.findall(r'getenv\("(GS_[A-Z0-9_]+)"\)', open(fn).read()))
    for fn in glob.glob(os.path.join(root, "gsearch_amd", "*.py")):
        knobs |= set(re.findall(r'environ[^\n]*?"(GS_[A-Z0-9_]+)"', open(fn).read()))
    assert len(knobs) > 50
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    table = md[md.index("## 5. Runtime knobs"):]
    rows = {}
    for line in table.split("\n"):
        if line.startswith("| `GS_"):
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            for name in re.findall(r"`(GS_[A-Z0-9_]+)`", cells[0]):
                rows[name] = cells[1]
    missing = sorted(k for k in knobs if k not in rows)
    assert not missing, missing
    assert all(set(c.replace(" ", "").split("/")) <= {"D", "S", "A", "T"} for c in rows.values()), rows
