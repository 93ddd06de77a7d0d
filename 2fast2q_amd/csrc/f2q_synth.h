// f2q_synth.h -- deterministic synthetic FASTQ workload; spec: tests/synth.py.
// One implementation compiled for host (f2q_synth_fastq) and device (f2q_synth_create).
#pragma once
#include <stdint.h>

#ifndef F2Q_HD
#ifdef __HIPCC__
#define F2Q_HD __host__ __device__ __forceinline__
#else
#define F2Q_HD inline
#endif
#endif

namespace f2q {

F2Q_HD uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
F2Q_HD uint64_t rnd(uint64_t seed, uint64_t i, uint64_t f)
{
    return mix64(mix64(seed ^ (i * 0xD1342543DE82EF95ull)) + f);
}

enum { F_GUIDE = 0, F_CLASS = 1, F_SUBPOS = 2, F_NFLAG = 3, F_QUAL = 4, F_OFFSET = 5, F_RANDWIN = 6, F_FLANK0 = 8 };

// plain-data view of a spec usable on the device (anchors as 2-bit codes)
struct SynthDev {
    uint64_t seed, n_reads, first_read;
    int32_t read_len, start, cassette, max_offset;
    int32_t up_len, down_len, glen, n_guides;
    uint32_t t_sub, t_rand, t_n, t_lowq, t_q29, t_q28;
    uint8_t up[64], down[64];          // ASCII
};

// One synthetic read.  emit(pos, base_char) is called for every position 0..R-1 in order,
// qual(pos) gives the quality char.  `guide2bit` returns the 2-bit codes (LSB first) of guide g.
struct SynthRead {
    uint64_t win;        // window bases as 2-bit codes (A0 C1 G2 T3), LSB first
    int32_t n_pos;       // position of 'N' inside the window or -1
    int32_t wstart;      // window start in the read
    int32_t off;         // cassette offset (cassette mode) else 0
    int32_t qpos;        // position of the special quality char or -1
    uint8_t qchar;
};

template <class GuideFn>
F2Q_HD SynthRead synth_plan(const SynthDev &s, uint64_t i, GuideFn guide2bit)
{
    SynthRead r;
    const int L = s.glen;
    uint64_t g = rnd(s.seed, i, F_GUIDE) % (uint64_t)s.n_guides;
    uint64_t win = guide2bit((uint32_t)g);
    uint32_t cls = (uint32_t)rnd(s.seed, i, F_CLASS);
    if (cls < s.t_sub) {
        uint64_t v = rnd(s.seed, i, F_SUBPOS);
        int pos = (int)((uint32_t)v % (uint32_t)L);
        uint32_t old = (uint32_t)(win >> (2 * pos)) & 3u;
        uint32_t nw = (old + 1u + (uint32_t)((v >> 32) % 3u)) & 3u;
        win = (win & ~(3ull << (2 * pos))) | ((uint64_t)nw << (2 * pos));
    } else if (cls < s.t_rand) {
        uint64_t v = rnd(s.seed, i, F_RANDWIN);
        win = (L >= 32) ? v : (v & ((1ull << (2 * L)) - 1ull));
    }
    uint64_t v = rnd(s.seed, i, F_NFLAG);
    r.n_pos = ((uint32_t)v < s.t_n) ? (int)((v >> 32) % (uint64_t)L) : -1;
    r.win = win;
    if (s.cassette) {
        r.off = (int)(rnd(s.seed, i, F_OFFSET) % (uint64_t)(s.max_offset + 1));
        r.wstart = r.off + s.up_len;
    } else {
        r.off = 0;
        r.wstart = s.start;
    }
    v = rnd(s.seed, i, F_QUAL);
    uint32_t q = (uint32_t)v;
    int qpos = r.wstart + (int)((v >> 32) % (uint64_t)L);
    r.qpos = -1; r.qchar = 'I';
    if (qpos < s.read_len) {
        if (q < s.t_lowq) { r.qpos = qpos; r.qchar = '#'; }
        else if (q < s.t_q29) { r.qpos = qpos; r.qchar = '>'; }
        else if (q < s.t_q28) { r.qpos = qpos; r.qchar = '='; }
    }
    return r;
}

// base character at position p of read i (flank word must be rnd(seed,i,F_FLANK0 + p/32))
F2Q_HD uint8_t synth_base(const SynthDev &s, const SynthRead &r, int p, uint64_t flank_word)
{
    const char *B = "ACGT";
    const int L = s.glen;
    if (s.cassette) {
        int c = p - r.off;
        if (c >= 0 && c < s.up_len) return s.up[c];
        int w = c - s.up_len;
        if (w >= 0 && w < L) return (w == r.n_pos) ? (uint8_t)'N' : (uint8_t)B[(r.win >> (2 * w)) & 3];
        int d = w - L;
        if (d >= 0 && d < s.down_len) return s.down[d];
    } else {
        int w = p - r.wstart;
        if (w >= 0 && w < L) return (w == r.n_pos) ? (uint8_t)'N' : (uint8_t)B[(r.win >> (2 * w)) & 3];
    }
    return (uint8_t)B[(flank_word >> (2 * (p & 31))) & 3];
}

} // namespace f2q
