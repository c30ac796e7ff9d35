"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
AA20 = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def rand_dna(rng, n):
    return rng.integers(0, 4, n)


def mutate(rng, g, mu, alphabet=4):
    h = g.copy()
    mask = rng.random(len(g)) < mu
    h[mask] = (h[mask] + rng.integers(1, alphabet, int(mask.sum()))) % alphabet
    return h


def dna_ascii(codes):
    return bytes(ACGT[codes])


def aa_ascii(codes):
    return bytes(AA20[codes])


def revcomp_ascii(s):
    tr = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    return s.translate(tr)[::-1]


def family(rng, length, mus, alphabet=4):
    root = rng.integers(0, alphabet, length)
    return [root] + [mutate(rng, root, mu, alphabet) for mu in mus]


def synth_sig_db(n_roots, per, m, seed, dtype=np.float32, jlo=0.3, jhi=0.99):
    """sketch-level database of SURVEY 8d: members keep each root slot with probability J ~ U[jlo,jhi]."""
    rng = np.random.default_rng(seed)

    def rnd(shape):
        if np.dtype(dtype) == np.float32:
            return rng.integers(0, 1 << 23, shape).astype(np.float32) * np.float32(2.0 ** -23)
        if np.dtype(dtype) == np.uint32:
            return rng.integers(0, 1 << 32, shape, dtype=np.uint64).astype(np.uint32)
        if np.dtype(dtype) == np.uint16:                          # SetSketch registers: a narrow band of values, so chance agreement happens
            return rng.integers(18000, 18400, shape).astype(np.uint16)
        return rng.integers(0, 1 << 63, shape, dtype=np.uint64)

    roots = rnd((n_roots, m))
    db = np.repeat(roots, per, axis=0)
    J = rng.uniform(jlo, jhi, (n_roots * per, 1))
    mask = rng.random(db.shape) > J
    db[mask] = rnd(db.shape)[mask]
    perm = rng.permutation(len(db))
    return np.ascontiguousarray(db[perm])


def queries_from(db, nq, seed, frac=0.1):
    rng = np.random.default_rng(seed)
    qi = rng.integers(0, len(db), nq)
    q = db[qi].copy()
    mask = rng.random(q.shape) < frac
    other = db[rng.integers(0, len(db), nq)]
    q[mask] = other[::-1][mask] if False else np.roll(other, 1, axis=1)[mask]
    return np.ascontiguousarray(q)


def bgzf_bytes(data, block=65280, level=6):
    """`data` as a BGZF file (bgzip: independent gzip members of <= 64 KB of text, each with its size in a 'BC' extra field, + the EOF member)"""
    import struct, zlib
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        total = 18 + len(cdata) + 8
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1) + cdata +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)
