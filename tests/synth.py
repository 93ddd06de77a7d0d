"""tests/synth.py -- the synthetic FASTQ workload of SURVEY.md §8(d), stated in pure Python.

This file is the *specification* of the deterministic generator.  The product
side re-implements it once in C/HIP (2fast2q_amd/csrc/f2q_synth.h, used by the
host entry point ``f2q_synth_fastq`` and by the device generator behind
``f2q_synth_create``); tests/test_synth.py checks the two byte-for-byte.

Everything is a pure function of (seed, read index, field) through a
splitmix64-style mixer, so any slice of the stream can be generated anywhere.
Probabilities are 32-bit thresholds (p * 2**32) so there is no float in the path.
"""
M64 = (1 << 64) - 1
BASES = b"ACGT"


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def rnd(seed, i, f):
    """64 random bits for (stream seed, item index i, field f)."""
    return mix64(mix64((seed ^ (i * 0xD1342543DE82EF95)) & M64) + f)


def p32(p):
    """probability -> 32-bit threshold"""
    return min(int(round(p * 4294967296.0)), 0xFFFFFFFF)


# field numbers (shared with f2q_synth.h)
F_GUIDE, F_CLASS, F_SUBPOS, F_NFLAG, F_QUAL, F_OFFSET, F_RANDWIN, F_FLANK0 = 0, 1, 2, 3, 4, 5, 6, 8


def make_library(n_guides, length, seed):
    """n unique uniform ACGT strings of `length` (<=32) bases; candidate k = bases of
    rnd(seed, k, 0) (2 bits each, LSB first); duplicates are skipped."""
    out, seen, k = [], set(), 0
    while len(out) < n_guides:
        v = rnd(seed, k, 0)
        k += 1
        s = bytes(BASES[(v >> (2 * j)) & 3] for j in range(length))
        if s in seen:
            continue
        seen.add(s)
        out.append(s.decode())
    return out


class Spec:
    """Workload description.  cassette=False: window at fixed `start`.  cassette=True:
    up+guide+down placed at a uniform offset in [0, max_offset]."""

    def __init__(self, seed=0xF2A5, n_reads=1000, read_len=150, start=0, cassette=False, up="", down="",
                 max_offset=100, p_sub=0.10, p_rand=0.05, p_n=0.005, p_lowq=0.06, p_q29=0.01, p_q28=0.01):
        self.seed, self.n_reads, self.read_len, self.start = seed, n_reads, read_len, start
        self.cassette, self.up, self.down, self.max_offset = cassette, up, down, max_offset
        self.t_sub = p32(p_sub)
        self.t_rand = p32(p_sub + p_rand)
        self.t_n = p32(p_n)
        self.t_lowq = p32(p_lowq)
        self.t_q29 = p32(p_lowq + p_q29)
        self.t_q28 = p32(p_lowq + p_q29 + p_q28)


def make_read(spec, guides, i):
    """(seq bytes, qual bytes) of read i."""
    R, L, G = spec.read_len, len(guides[0]), len(guides)
    seq = bytearray(R)
    # flanks: uniform ACGT, 32 bases per 64-bit draw
    for w in range((R + 31) // 32):
        v = rnd(spec.seed, i, F_FLANK0 + w)
        for j in range(32):
            p = w * 32 + j
            if p < R:
                seq[p] = BASES[(v >> (2 * j)) & 3]
    g = rnd(spec.seed, i, F_GUIDE) % G
    win = bytearray(guides[g].encode())
    c = rnd(spec.seed, i, F_CLASS)
    cls = c & 0xFFFFFFFF
    if cls < spec.t_sub:
        v = rnd(spec.seed, i, F_SUBPOS)
        pos = (v & 0xFFFFFFFF) % L
        old = BASES.index(win[pos])
        win[pos] = BASES[(old + 1 + ((v >> 32) % 3)) & 3]
    elif cls < spec.t_rand:
        v = rnd(spec.seed, i, F_RANDWIN)
        for j in range(L):
            win[j] = BASES[(v >> (2 * j)) & 3]
    v = rnd(spec.seed, i, F_NFLAG)
    if (v & 0xFFFFFFFF) < spec.t_n:
        win[(v >> 32) % L] = ord("N")
    if spec.cassette:
        off = rnd(spec.seed, i, F_OFFSET) % (spec.max_offset + 1)
        cas = spec.up.encode() + bytes(win) + spec.down.encode()
        wstart = off + len(spec.up)
        seq[off:off + len(cas)] = cas
        del seq[R:]
    else:
        wstart = spec.start
        seq[wstart:wstart + L] = win
        del seq[R:]
    qual = bytearray(b"I" * R)
    v = rnd(spec.seed, i, F_QUAL)
    q = v & 0xFFFFFFFF
    qpos = wstart + (v >> 32) % L
    if qpos < R:
        if q < spec.t_lowq:
            qual[qpos] = ord("#")
        elif q < spec.t_q29:
            qual[qpos] = ord(">")
        elif q < spec.t_q28:
            qual[qpos] = ord("=")
    return bytes(seq), bytes(qual)


def make_fastq(spec, guides, lo=0, hi=None):
    hi = spec.n_reads if hi is None else hi
    parts = []
    for i in range(lo, hi):
        s, q = make_read(spec, guides, i)
        parts.append(b"@r%d\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(parts)
