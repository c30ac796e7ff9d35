"""a13 / a15: the text output of the path and the bindash distance, against LITERAL expectations (not mirror-vs-mirror).

tests/golden/reqanswer_expected.txt was written by hand from the format strings of /root/reference/src/answer.rs:45-71
("\\n{rank}\\t{path}\\tfasta_id:\\t{id}\\tlength:\\t{len}" when any neighbour has d <= threshold, then for d < threshold
"\\nquery_id:\\t{path}\\tdistance:\\t{d:.5E}\\tanswer_fasta_path\\t{dbpath}\\t" + "{fasta_id} \\t answer_seq_len:\\t {len}") and the exact
decimal expansions of the f32 distances in tests/golden/reqanswer_case.txt. Both host mirrors (Python api.ReqAnswer and the C++
gsearch::ReqAnswer of include/gsearch_amd.hpp, run GPU-free through `tohnsw_request_demo --answer-fixture`) must reproduce it."""
import io
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, "tests", "golden", "reqanswer_case.txt")
EXPECTED = os.path.join(ROOT, "tests", "golden", "reqanswer_expected.txt")
# bindash.rs:93-99 evaluated by hand in f64 for (d, k) = (0, 21), (1, 21), (0.5, 21), (f32(0.6075), 16): 1 - (2j/(1+j))^(1/k), j = 1-d
BINDASH = [0.0, 1.0, 0.019122659390482077, 0.03518920659368241]


def _f32(hexbits):
    return float(np.frombuffer(struct.pack("<I", int(hexbits, 16)), dtype=np.float32)[0])


def _parse():
    seqdict, reqs, bd, thr = [], [], [], None
    for line in open(CASE, encoding="utf-8").read().split("\n"):
        if not line or line[0] == "#":
            continue
        t = line.split("\t")
        if t[0] == "S":
            seqdict.append((t[1], t[2], int(t[3])))
        elif t[0] == "T":
            thr = _f32(t[1])
        elif t[0] == "Q":
            reqs.append((int(t[1]), (t[2], t[3], int(t[4])), []))
        elif t[0] == "N":
            reqs[-1][2].append((int(t[1]), _f32(t[2])))
        elif t[0] == "B":
            bd.append((_f32(t[1]), int(t[2])))
    return seqdict, reqs, bd, thr


def test_python_reqanswer_equals_literal_fixture():
    import gsearch_amd as G
    seqdict, reqs, _, thr = _parse()
    assert np.float32(thr) == np.float32(0.99)                       # out_threshold of dnarequest.rs:83
    buf = io.StringIO()
    nb = [G.ReqAnswer(rank, item, [G.Neighbour(i, d) for i, d in nbs]).dump(seqdict, thr, buf) for rank, item, nbs in reqs]
    assert buf.getvalue() + "\n" == open(EXPECTED, encoding="utf-8").read()
    assert nb == [4, 0, 0, 5, 0]                                      # Ok(nb_match) of answer.rs:74


def test_cpp_reqanswer_equals_literal_fixture():
    import gsearch_amd as G
    exe = os.path.join(os.path.dirname(G.SO_PATH), "tohnsw_request_demo")
    out = subprocess.run([exe, "--answer-fixture", CASE], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    want = open(EXPECTED, encoding="utf-8").read()
    assert out.stdout[:len(want)] == want
    got = [float(x) for x in out.stdout[len(want):].split()]
    assert len(got) == len(BINDASH) and all(abs(a - b) < 1e-6 for a, b in zip(got, BINDASH))


def test_bindash_distance_python():
    import gsearch_amd as G
    _, _, bd, _ = _parse()
    for (d, k), want in zip(bd, BINDASH):
        assert abs(G.bindash_distance(d, k) - want) < 1e-6
    # f32 evaluation like the reference: j, frac and powf are single precision (bindash.rs:95-98)
    j = np.float32(1.0) - np.float32(0.6075)
    frac = np.float32(2.0) * j / (np.float32(1.0) + j)
    assert G.bindash_distance(np.float32(0.6075), 16) == 1.0 - float(np.power(frac, np.float32(1.0) / np.float32(16), dtype=np.float32))
    assert G.bindash_distance(0.0, 21) == 0.0 and G.bindash_distance(1.0, 21) == 1.0
