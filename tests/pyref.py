"""A second, INDEPENDENT restatement of SPEC.md sections 1.1, 2 and 3.1 in pure Python (the forward-only golden vectors come from the repo's own C oracle,
so a closure the oracle and the kernels share but the reference does not would pass them). Written from SPEC.md and /root/reference/src/bin/bindash.rs:346-354 /
src/dna/dnasketch.rs:164-169 only - it shares no code with oracle/gs_oracle.c or gs_spec.hpp. Small inputs only (Python loops).
Test infrastructure: nothing under gsearch_amd/ imports it."""
import struct

M64 = (1 << 64) - 1
CODE = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}          # A C G T a c g t; every other byte is dropped before windowing (dnafiles.rs:41,150-151)


def splitmix_next(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return x, z ^ (z >> 31)


class Rng:
    """xoshiro256++ seeded from four SplitMix64 outputs (rand_xoshiro seed_from_u64)"""

    def __init__(self, seed):
        x, s = seed & M64, []
        for _ in range(4):
            x, z = splitmix_next(x)
            s.append(z)
        self.s = s

    @staticmethod
    def _rotl(v, r):
        return ((v << r) | (v >> (64 - r))) & M64

    def next64(self):
        s = self.s
        r = (self._rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return r

    def u32f_bits(self):
        """U32f = (next32 >> 9) * 2^-23: the 23 bits themselves"""
        return (self.next64() >> 32) >> 9

    def uint(self, n):
        zone = M64 - ((1 << 64) % n)
        while True:
            p = self.next64() * n
            if (p & M64) <= zone:
                return p >> 64


def kmers(records, k, forward_only):
    """SPEC 1.1: every window of k consecutive kept bases of one record; canonical = min(fwd, reverse complement), forward-only = the window as read"""
    for rec in records:
        codes = [CODE[b] for b in rec if b in CODE]
        for i in range(len(codes) - k + 1):
            w = codes[i:i + k]
            fwd = 0
            for c in w:
                fwd = fwd * 4 + c
            if forward_only:
                yield fwd
            else:
                rc = 0
                for c in reversed(w):
                    rc = rc * 4 + (3 - c)
                yield min(fwd, rc)


def optdens(records, k, m, forward_only=False):
    """SPEC 3.1 optdens: slot[b] = min r over the k-mers, H = fx64; densification from the pre-densification snapshot. Returns the m f32 bit patterns."""
    slot = [None] * m
    for v in kmers(records, k, forward_only):
        g = Rng((v * 0x517CC1B727220A95) & M64)
        r = g.u32f_bits()
        b = g.uint(m)
        if slot[b] is None or r < slot[b]:
            slot[b] = r
    if all(s is None for s in slot):
        return [struct.unpack("<I", struct.pack("<f", 1.0))[0]] * m
    snap = list(slot)
    for b in range(m):
        if snap[b] is None:
            g = Rng(b)
            while True:
                j = g.uint(m)
                if snap[j] is not None:
                    slot[b] = snap[j]
                    break
    return [struct.unpack("<I", struct.pack("<f", s * 2.0 ** -23))[0] for s in slot]
