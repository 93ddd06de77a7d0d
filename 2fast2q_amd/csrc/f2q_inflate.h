// Raw DEFLATE (RFC 1951) decoder for the file reader (f2q_reader.h).  Host only.
//
// zlib's inflate is what bounds `2fast2q -c` on .gz input once counting runs on the GPU (0.59 GB/s of text per
// thread on the test box).  This decoder does the same job with the usual word-at-a-time techniques: a 64-bit bit
// buffer refilled with one unaligned 8-byte load, an 11-bit first-level table for literal/length codes (8-bit for
// distances) with second-level tables behind it, 8-byte match copies.  Output is streamed: `run()` fills the
// caller's buffer and can stop anywhere (in the middle of a match too); references that reach back before the
// current buffer are served from a private copy of the last 32 KiB.  The compressed input must be addressable as
// one range (a memory-mapped file, or one BGZF member).
//
// Every table index and every input/output position is bounds-checked; a damaged stream ends in ERR, never in an
// out-of-range access.  tests/test_reader_cpu.py checks it against zlib on random, degenerate and damaged streams.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace f2qz {

#if defined(__x86_64__)
// CRC-32 of gzip by carry-less multiplication (Gopal et al., "Fast CRC computation for generic polynomials using
// PCLMULQDQ", Intel 2009): four 128-bit lanes folded 64 bytes at a time, then down to 32 bits by Barrett reduction.
// Works on the raw register value (callers complement before and after); n >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_fold(const uint8_t *buf, size_t n, uint32_t crc)
{
    static const uint64_t __attribute__((aligned(16))) k1k2[] = {0x0154442bd4ull, 0x01c6e41596ull};     // x^(4*128+32), x^(4*128-32) mod P
    static const uint64_t __attribute__((aligned(16))) k3k4[] = {0x01751997d0ull, 0x00ccaa009eull};     // x^(128+32), x^(128-32) mod P
    static const uint64_t __attribute__((aligned(16))) k5k0[] = {0x0163cd6124ull, 0x0000000000ull};     // x^64 mod P
    static const uint64_t __attribute__((aligned(16))) poly[] = {0x01db710641ull, 0x01f7011641ull};     // P, floor(x^64 / P)
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; n -= 64;
    while (n >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; n -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (n >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; n -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// CRC-32 of gzip (reflected 0xEDB88320): carry-less multiplication where the CPU has it, else slicing-by-16 (16 table
// look-ups per 16 input bytes).  zlib 1.2.11's crc32 runs at ~0.7 GB/s per thread on the test box, which made the
// checksum a third of a BGZF worker's time.
struct Crc32 {
    uint32_t t[16][256];
    bool clmul = false;
    Crc32()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (int s = 1; s < 16; s++)
            for (uint32_t i = 0; i < 256; i++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
#if defined(__x86_64__)
        const char *e = getenv("F2Q_NO_CLMUL");
        clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !(e && e[0] == '1');
        if (clmul) {                                    // (checked once against the tables: a wrong constant must not pass silently)
            uint8_t probe[211];
            for (size_t i = 0; i < sizeof probe; i++) probe[i] = (uint8_t)(i * 37u + 11u);
            clmul = false;
            const uint32_t want = update(0x1234u, probe, sizeof probe);
            clmul = true;
            if (update(0x1234u, probe, sizeof probe) != want) clmul = false;
        }
#endif
    }
    // crc = 0 for a new message; the value of the bytes so far to continue one (same convention as zlib)
    uint32_t update(uint32_t crc, const uint8_t *p, size_t n) const
    {
        crc = ~crc;
#if defined(__x86_64__)
        if (clmul && n >= 64) { const size_t k = n & ~(size_t)15; crc = crc32_fold(p, k, crc); p += k; n -= k; }
#endif
        while (n >= 16) {
            uint64_t a, b;
            memcpy(&a, p, 8); memcpy(&b, p + 8, 8);
            a ^= crc;
            crc = t[15][a & 0xFF] ^ t[14][(a >> 8) & 0xFF] ^ t[13][(a >> 16) & 0xFF] ^ t[12][(a >> 24) & 0xFF] ^
                  t[11][(a >> 32) & 0xFF] ^ t[10][(a >> 40) & 0xFF] ^ t[9][(a >> 48) & 0xFF] ^ t[8][a >> 56] ^
                  t[7][b & 0xFF] ^ t[6][(b >> 8) & 0xFF] ^ t[5][(b >> 16) & 0xFF] ^ t[4][(b >> 24) & 0xFF] ^
                  t[3][(b >> 32) & 0xFF] ^ t[2][(b >> 40) & 0xFF] ^ t[1][(b >> 48) & 0xFF] ^ t[0][b >> 56];
            p += 16; n -= 16;
        }
        while (n--) crc = t[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8);
        return ~crc;
    }
    static const Crc32 &get() { static const Crc32 c; return c; }
};

struct Inflater {
    enum Status { OUT_FULL = 0, DONE = 1, ERR = 2, BOUNDARY = 3 };     // BOUNDARY: stopped in front of a block header at or past stop_bit

    static constexpr int LT_BITS = 11, DT_BITS = 8;
    static constexpr uint32_t LT_CAP = 2048 + 2400, DT_CAP = 256 + 800;
    static constexpr uint32_t F_LIT = 1u << 15, F_EOB = 1u << 14, F_SUB = 1u << 13, F_BAD = 1u << 12;
    static constexpr uint32_t WSIZE = 32768;

    // entry: bits 0-7 bits to drop at this stage (code length + extra bits), 8-11 extra bits / sub-table index bits,
    // 12-15 flags, 16-31 value (literal, length / distance base, sub-table start)
    uint32_t lt[LT_CAP], dt[DT_CAP];
    uint64_t bb = 0; int bc = 0;
    const uint8_t *in = nullptr, *in_end = nullptr, *in_begin = nullptr;
    uint64_t stop_bit = ~0ull;                   // run() returns BOUNDARY in front of the first block header at a bit offset >= this (from in_begin)
    enum { S_HEADER, S_STORED, S_CODES, S_MATCH, S_DONE, S_ERR } st = S_HEADER;
    bool final_block = false;
    uint32_t stored_left = 0, m_len = 0, m_dist = 0;
    uint8_t win[WSIZE]; uint32_t win_len = 0;
    uint64_t total_out = 0;

    void reset(const uint8_t *p, size_t n)
    {
        in = p; in_end = p + n; in_begin = p; bb = 0; bc = 0; st = S_HEADER; final_block = false; stored_left = 0; m_len = m_dist = 0;
        win_len = 0; total_out = 0; stop_bit = ~0ull;
    }
    // start in front of the block header at bit offset `bit` of [p, p + n), with the `wn` <= 32768 bytes of text that
    // precede it as the window (the parallel decoder of f2q_pargz.h continues a stream this way)
    bool reset_at(const uint8_t *p, size_t n, uint64_t bit, const uint8_t *w, uint32_t wn)
    {
        reset(p, n);
        if ((bit >> 3) > n) return false;
        in = p + (bit >> 3);
        if (bit & 7u) { if (!need((int)(bit & 7u))) return false; take((int)(bit & 7u)); }
        if (wn > WSIZE) { w += wn - WSIZE; wn = WSIZE; }
        if (wn) memcpy(win, w, wn);
        win_len = wn;
        return true;
    }
    // bit offset (from in_begin) of the next unread bit
    uint64_t bit_pos() const { return (uint64_t)(in - in_begin) * 8u - (uint64_t)bc; }
    // first unread input byte once DONE (whole bytes still in the bit buffer are given back)
    const uint8_t *input_pos() const { return in - (bc >> 3); }

    // ---- bit input ---------------------------------------------------------------------------------
    static uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
    static void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
    void refill_fast() { bb |= load64(in) << bc; in += (63 - bc) >> 3; bc |= 56; }       // needs in_end - in >= 8
    void refill_safe() { while (bc < 56 && in < in_end) { bb |= (uint64_t)*in++ << bc; bc += 8; } }
    bool need(int n) { if (bc < n) refill_safe(); return bc >= n; }
    uint32_t take(int n) { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1ull)); bb >>= n; bc -= n; return v; }

    // ---- Huffman tables -----------------------------------------------------------------------------
    static uint32_t rev(uint32_t code, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; } return r; }

    static uint32_t litlen_entry(int sym, int len)
    {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t xb[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (sym < 256) return ((uint32_t)sym << 16) | F_LIT | (uint32_t)len;
        if (sym == 256) return F_EOB | (uint32_t)len;
        if (sym > 285) return F_BAD | (uint32_t)len;
        return ((uint32_t)base[sym - 257] << 16) | ((uint32_t)xb[sym - 257] << 8) | (uint32_t)(len + xb[sym - 257]);
    }
    static uint32_t dist_entry(int sym, int len)
    {
        static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t xb[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (sym > 29) return F_BAD | (uint32_t)len;
        return ((uint32_t)base[sym] << 16) | ((uint32_t)xb[sym] << 8) | (uint32_t)(len + xb[sym]);
    }

    // canonical code -> two-level table.  Rules as zlib's inflate_table: over-subscribed sets are rejected, incomplete
    // ones too unless they consist of a single 1-bit code (distances / literal-length of degenerate streams).
    template <bool DIST>
    static bool build(uint32_t *tab, uint32_t cap, int root, const uint8_t *lens, int n)
    {
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[lens[i]]++;
        int maxl = 15; while (maxl > 0 && count[maxl] == 0) maxl--;
        const uint32_t rsize = 1u << root;
        if (maxl == 0) { for (uint32_t i = 0; i < rsize; i++) tab[i] = F_BAD | 1u; return true; }   // no codes: any use is an error
        int left = 1;
        for (int l = 1; l <= 15; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        if (left > 0 && maxl != 1) return false;
        // symbols in canonical order
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        uint16_t sorted[320];
        for (int i = 0; i < n; i++) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
        const int n_codes = offs[15];
        for (uint32_t i = 0; i < rsize; i++) tab[i] = F_BAD | 1u;
        uint32_t next_free = rsize;
        uint32_t code = 0; int k = 0;                     // code: canonical code of sorted[k], len bits, MSB first
        for (int len = 1; len <= maxl; len++) {
            for (int c = 0; c < count[len]; c++, k++, code++) {
                const int sym = sorted[k];
                if (len <= root) {
                    const uint32_t e = DIST ? dist_entry(sym, len) : litlen_entry(sym, len);
                    for (uint32_t i = rev(code, len); i < rsize; i += 1u << len) tab[i] = e;
                } else {
                    const uint32_t prefix = code >> (len - root);             // its first `root` bits
                    const uint32_t slot = rev(prefix, root);
                    if (!(tab[slot] & F_SUB)) {
                        // sub-table wide enough for the longest code that shares the prefix (they follow in canonical order)
                        int sub_max = len; uint32_t cc = code; int kk = k, ll = len, left_in_len = count[len] - c;
                        for (;;) {
                            // advance to the next code
                            kk++; cc++; left_in_len--;
                            while (left_in_len == 0 && ll < maxl) { ll++; cc <<= 1; left_in_len = count[ll]; }
                            if (kk >= n_codes || left_in_len == 0) break;
                            if ((cc >> (ll - root)) != prefix) break;
                            sub_max = ll;
                        }
                        const uint32_t sbits = (uint32_t)(sub_max - root);
                        if (next_free + (1u << sbits) > cap) return false;
                        tab[slot] = (next_free << 16) | F_SUB | (sbits << 8) | (uint32_t)root;
                        for (uint32_t i = 0; i < (1u << sbits); i++) tab[next_free + i] = F_BAD | 1u;
                        next_free += 1u << sbits;
                    }
                    const uint32_t sub = tab[slot] >> 16, sbits = (tab[slot] >> 8) & 15u;
                    const int rest = len - root;                                   // bits decoded in the second stage
                    const uint32_t e = DIST ? dist_entry(sym, rest) : litlen_entry(sym, rest);
                    const uint32_t low = code & ((1u << rest) - 1u);
                    for (uint32_t i = rev(low, rest); i < (1u << sbits); i += 1u << rest) tab[sub + i] = e;
                }
            }
            code <<= 1;
        }
        return true;
    }

    bool fixed_tables()
    {
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        uint8_t d[32]; for (int i = 0; i < 32; i++) d[i] = 5;
        return build<false>(lt, LT_CAP, LT_BITS, l, 288) && build<true>(dt, DT_CAP, DT_BITS, d, 32);
    }

    bool dynamic_tables()
    {
        if (!need(14)) return false;
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; i++) { if (!need(3)) return false; cl[order[i]] = (uint8_t)take(3); }
        uint32_t ct[128 + 8];
        {   // code-length code: 7-bit root, lengths <= 7 -> no sub-tables; built with the DIST entry layout (value = symbol)
            int count[8] = {0};
            for (int i = 0; i < 19; i++) count[cl[i]]++;
            int left = 1, maxl = 7; while (maxl > 0 && count[maxl] == 0) maxl--;
            if (maxl == 0) return false;
            for (int l = 1; l <= 7; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
            if (left > 0) return false;                                     // zlib: the code-length code must be complete
            for (int i = 0; i < 128; i++) ct[i] = F_BAD | 1u;
            uint32_t code = 0;
            for (int len = 1; len <= 7; len++) {
                for (int s = 0; s < 19; s++) if (cl[s] == len) {
                    for (uint32_t i = rev(code, len); i < 128; i += 1u << len) ct[i] = ((uint32_t)s << 16) | (uint32_t)len;
                    code++;
                }
                code <<= 1;
            }
        }
        uint8_t lens[320]; memset(lens, 0, sizeof lens);
        int i = 0; const int total = hlit + hdist;
        while (i < total) {
            if (!need(7 + 7)) { if (bc < 1) return false; }                 // the last code may be shorter than 7 bits
            const uint32_t e = ct[bb & 127u];
            if (e & F_BAD) return false;
            const int len = (int)(e & 0xFF);
            if (len > bc) return false;
            take(len);
            const int sym = (int)(e >> 16);
            if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
            int rep, val = 0;
            if (sym == 16) { if (i == 0 || !need(2)) return false; val = lens[i - 1]; rep = 3 + (int)take(2); }
            else if (sym == 17) { if (!need(3)) return false; rep = 3 + (int)take(3); }
            else { if (!need(7)) return false; rep = 11 + (int)take(7); }
            if (i + rep > total) return false;
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) return false;                                   // no end-of-block code
        return build<false>(lt, LT_CAP, LT_BITS, lens, hlit) && build<true>(dt, DT_CAP, DT_BITS, lens + hlit, hdist);
    }

    // ---- output helpers -----------------------------------------------------------------------------
    // byte `back` positions before out (back >= 1); positions before the buffer come from the saved window
    uint8_t back_byte(const uint8_t *out_begin, const uint8_t *out, uint32_t back) const
    {
        const size_t have = (size_t)(out - out_begin);
        if (back <= have) return out[-(ptrdiff_t)back];
        return win[win_len - (back - (uint32_t)have)];
    }
    void save_window(const uint8_t *out_begin, const uint8_t *out)
    {
        const size_t have = (size_t)(out - out_begin);
        if (have >= WSIZE) { memcpy(win, out - WSIZE, WSIZE); win_len = WSIZE; }
        else {
            const uint32_t keep = (uint32_t)((win_len + have > WSIZE) ? WSIZE - have : win_len);
            memmove(win, win + (win_len - keep), keep);
            memcpy(win + keep, out_begin, have);
            win_len = keep + (uint32_t)have;
        }
        total_out += have;
    }

    // ---- the decoder --------------------------------------------------------------------------------
    // fills [out_begin, out_end); returns OUT_FULL (call again with the next buffer), DONE or ERR; *produced = bytes written
    Status run(uint8_t *out_begin, uint8_t *out_end, size_t *produced)
    {
        uint8_t *out = out_begin;
        Status rc = ERR;
        for (;;) {
            if (st == S_MATCH) {                                       // a match cut by the end of the previous buffer
                while (m_len && out < out_end) { *out = back_byte(out_begin, out, m_dist); out++; m_len--; }
                if (m_len) { rc = OUT_FULL; break; }
                st = S_CODES;
            }
            if (st == S_HEADER) {
                if (final_block) { st = S_DONE; rc = DONE; break; }
                if (bit_pos() >= stop_bit) { rc = BOUNDARY; break; }
                if (!need(3)) { st = S_ERR; break; }
                final_block = take(1) != 0;
                const uint32_t type = take(2);
                if (type == 0) {
                    take(bc & 7);                                      // to the byte boundary
                    if (!need(32)) { st = S_ERR; break; }
                    const uint32_t len = take(16), nlen = take(16);
                    if ((len ^ nlen) != 0xFFFFu) { st = S_ERR; break; }
                    stored_left = len; st = S_STORED;
                } else if (type == 1) { if (!fixed_tables()) { st = S_ERR; break; } st = S_CODES; }
                else if (type == 2) { if (!dynamic_tables()) { st = S_ERR; break; } st = S_CODES; }
                else { st = S_ERR; break; }
            }
            if (st == S_STORED) {
                // whole bytes still in the bit buffer first, then straight from the input
                while (stored_left && bc >= 8 && out < out_end) { *out++ = (uint8_t)take(8); stored_left--; }
                if (stored_left && bc < 8) {
                    bb = 0; bc = 0;
                    size_t n = stored_left;
                    if (n > (size_t)(out_end - out)) n = (size_t)(out_end - out);
                    if (n > (size_t)(in_end - in)) { st = S_ERR; break; }
                    memcpy(out, in, n); in += n; out += n; stored_left -= (uint32_t)n;
                }
                if (stored_left) { rc = OUT_FULL; break; }
                st = S_HEADER;
                continue;
            }
            if (st == S_CODES) {
                bool err = false, eob = false;
                // fast loop: room for the longest match plus a copy overshoot, 8 readable input bytes
                bool have_e = false; uint32_t e = 0;   // the next entry, looked up while the previous match was being copied
                while (out_end - out >= 258 + 16 && in_end - in >= 8) {
                    if (!have_e) { refill_fast(); e = lt[bb & ((1u << LT_BITS) - 1u)]; }
                    have_e = false;
                    if (__builtin_expect(e & F_SUB, 0)) { bb >>= LT_BITS; bc -= LT_BITS; e = lt[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))]; }
                    const uint64_t saved = bb;
                    bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                    if (e & F_LIT) {
                        *out++ = (uint8_t)(e >> 16);
                        // up to four more literals from the same refill: first-level literal codes are at most 11 bits,
                        // so 56 - 15 bits cover them (FASTQ sequence lines are runs of 2-3 bit literals)
                        e = lt[bb & ((1u << LT_BITS) - 1u)];
                        if (!(e & F_LIT)) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *out++ = (uint8_t)(e >> 16);
                        e = lt[bb & ((1u << LT_BITS) - 1u)];
                        if (!(e & F_LIT)) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *out++ = (uint8_t)(e >> 16);
                        e = lt[bb & ((1u << LT_BITS) - 1u)];
                        if (!(e & F_LIT)) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *out++ = (uint8_t)(e >> 16);
                        continue;
                    }
                    if (e & (F_EOB | F_BAD)) { if (e & F_BAD) err = true; else eob = true; break; }
                    // code and extra bits leave the buffer in one shift; the extra bits are read from the saved copy
                    const uint32_t xb = (e >> 8) & 15u;
                    uint32_t len = (e >> 16) + (uint32_t)((saved >> ((e & 0xFF) - xb)) & ((1u << xb) - 1u));
                    uint32_t d = dt[bb & ((1u << DT_BITS) - 1u)];
                    if (__builtin_expect(d & F_SUB, 0)) { bb >>= DT_BITS; bc -= DT_BITS; d = dt[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))]; }
                    if (d & F_BAD) { err = true; break; }
                    const uint64_t dsaved = bb;
                    bb >>= (d & 0xFF); bc -= (int)(d & 0xFF);
                    const uint32_t dxb = (d >> 8) & 15u;
                    const uint32_t dist = (d >> 16) + (uint32_t)((dsaved >> ((d & 0xFF) - dxb)) & ((1u << dxb) - 1u));
                    const size_t have = (size_t)(out - out_begin);
                    if (dist > have) {
                        if (dist - have > win_len) { err = true; break; }
                        while (len && dist > (size_t)(out - out_begin)) { *out = back_byte(out_begin, out, dist); out++; len--; }
                        if (!len) continue;
                    }
                    const uint8_t *src = out - dist;
                    if (in_end - in >= 8) { refill_fast(); e = lt[bb & ((1u << LT_BITS) - 1u)]; have_e = true; }
                    if (dist >= 8) {
                        uint8_t *dst = out; const uint8_t *s = src; uint8_t *const end = out + len;
                        store64(dst, load64(s)); store64(dst + 8, load64(s + 8));           // most matches are short
                        if (len > 16) { dst += 16; s += 16; do { store64(dst, load64(s)); dst += 8; s += 8; } while (dst < end); }
                    } else if (dist == 1) {
                        const uint64_t v = 0x0101010101010101ull * src[0];
                        uint8_t *dst = out; uint8_t *const end = out + len;
                        do { store64(dst, v); dst += 8; } while (dst < end);
                    } else {
                        for (uint32_t i = 0; i < len; i++) out[i] = src[i];
                    }
                    out += len;
                }
                if (err) { st = S_ERR; break; }
                if (eob) { st = S_HEADER; continue; }
                // careful loop: exact room and input checks, one symbol at a time
                for (;;) {
                    refill_safe();
                    uint32_t e = lt[bb & ((1u << LT_BITS) - 1u)];
                    int used = 0;
                    if (e & F_SUB) {
                        if (bc < LT_BITS) { err = true; break; }
                        used = LT_BITS;
                        e = lt[(e >> 16) + (uint32_t)((bb >> LT_BITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
                    }
                    if (e & F_BAD) { err = true; break; }
                    used += (int)(e & 0xFF);                               // code and extra bits
                    if (used > bc) { err = true; break; }                  // the stream ends inside a code
                    if (e & F_LIT) {
                        if (out == out_end) { rc = OUT_FULL; break; }      // nothing consumed: the symbol is decoded again next time
                        bb >>= used; bc -= used;
                        *out++ = (uint8_t)(e >> 16);
                        if (out_end - out >= 258 + 16 && in_end - in >= 8) break;   // back to the fast loop
                        continue;
                    }
                    if (e & F_EOB) { bb >>= used; bc -= used; eob = true; break; }
                    const uint32_t xb = (e >> 8) & 15u;
                    uint32_t len = (e >> 16) + (uint32_t)((bb >> (used - (int)xb)) & ((1u << xb) - 1u));
                    bb >>= used; bc -= used;                              // length taken; 15 + 13 more bits at most for the distance
                    refill_safe();
                    uint32_t d = dt[bb & ((1u << DT_BITS) - 1u)];
                    used = 0;
                    if (d & F_SUB) {
                        if (bc < DT_BITS) { err = true; break; }
                        used = DT_BITS;
                        d = dt[(d >> 16) + (uint32_t)((bb >> DT_BITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
                    }
                    if (d & F_BAD) { err = true; break; }
                    used += (int)(d & 0xFF);
                    const uint32_t dxb = (d >> 8) & 15u;
                    if (used > bc) { err = true; break; }
                    const uint32_t dist = (d >> 16) + (uint32_t)((bb >> (used - (int)dxb)) & ((1u << dxb) - 1u));
                    bb >>= used; bc -= used;
                    if (dist > (size_t)(out - out_begin) + win_len) { err = true; break; }
                    while (len && out < out_end) { *out = back_byte(out_begin, out, dist); out++; len--; }
                    if (len) { m_len = len; m_dist = dist; st = S_MATCH; rc = OUT_FULL; break; }
                    if (out_end - out >= 258 + 16 && in_end - in >= 8) break;
                }
                if (err) { st = S_ERR; break; }
                if (rc == OUT_FULL && (st == S_MATCH || out == out_end)) break;
                if (eob) { st = S_HEADER; continue; }
                continue;                                                  // fast loop again
            }
            if (st == S_DONE) { rc = DONE; break; }
            if (st == S_ERR) break;
        }
        if (st == S_ERR) rc = ERR;
        save_window(out_begin, out);
        *produced = (size_t)(out - out_begin);
        return rc;
    }
};

} // namespace f2qz
