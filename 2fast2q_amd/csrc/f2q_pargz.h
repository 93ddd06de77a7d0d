// Parallel decoding of ONE ordinary gzip member (host only; used by f2q_reader.h for regular .gz files).
//
// A DEFLATE stream is sequential in two ways: a block can only be found by decoding everything in front of it, and a
// match may copy from the 32 KiB of text in front of it.  Both are worked around the way pugz / rapidgzip do it:
//
//   1. the compressed bytes of a round are cut into chunks; for every chunk but the first a worker SEARCHES the first
//      position at which a dynamic-Huffman block header parses and two blocks decode cleanly (find_block_start);
//   2. every worker decodes its chunk from that position up to the start the next worker found, without knowing the
//      32 KiB in front of it: output symbols are 16 bits wide, 0..255 = a byte, 0x8000 | i = "byte i of the unknown
//      window" (SpecInflater); a match that copies such a marker copies the marker;
//   3. the chunks are chained in order -- a chunk is valid iff the chunk before it is valid and ended exactly where
//      this one started (the first chunk starts at a known position with a known window, and is decoded by the plain
//      Inflater); the window is handed from chunk to chunk by resolving just the last 32 KiB of each; then all chunks
//      are resolved to bytes in parallel (a table look-up per marker).
// A start that was guessed wrong (a false positive of the search) breaks the chain: the chunks behind it are thrown
// away and the next round starts at the true boundary the last valid chunk reached.  A damaged stream delivers the text
// decoded before the damage, exactly as the sequential decoder does.  CRC-32 / ISIZE are checked by the caller on the
// text it is handed, as for a sequential read.
#pragma once
#include <stdint.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <zlib.h>

#include "f2q_inflate.h"

namespace f2qz {

// a growable array that is never value-initialised (std::vector::resize would write every element once more)
template <class T>
struct RawBuf {
    T *p = nullptr; size_t cap = 0, n = 0;
    RawBuf() {}
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    void reserve(size_t want) { if (want > cap) { T *q = (T *)realloc(p, want * sizeof(T)); if (!q) abort(); p = q; cap = want; } }
    void clear() { n = 0; }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
};

// ---- decoding with an unknown window ------------------------------------------------------------------------------
struct SpecInflater : Inflater {
    static constexpr uint16_t MARK = 0x8000u;

    // Decodes blocks from the current position (set with reset_at) into `out` (appended) until a block header at a bit
    // offset >= stop_bit (BOUNDARY), the end of the final block (DONE), an invalid stream or more than `limit` symbols
    // (ERR).  *blocks = blocks completed.  `max_blocks`: stop (BOUNDARY) after that many blocks (the search's test run).
    Status run_spec(RawBuf<uint16_t> &out, size_t limit, uint32_t max_blocks, uint32_t *blocks)
    {
        size_t o = out.size();
        uint32_t nb = 0;
        auto room = [&](size_t need_syms) { if (out.cap < o + need_syms) out.reserve(std::max(out.cap * 3 / 2, o + need_syms + (1u << 16))); };
        Status rc = ERR;
        for (;;) {
            if (final_block) { rc = DONE; break; }
            if (bit_pos() >= stop_bit || nb >= max_blocks) { rc = BOUNDARY; break; }
            if (!need(3)) break;
            final_block = take(1) != 0;
            const uint32_t type = take(2);
            if (type == 0) {
                take(bc & 7);
                if (!need(32)) break;
                const uint32_t len = take(16), nlen = take(16);
                if ((len ^ nlen) != 0xFFFFu) break;
                if (o + len > limit) break;
                room(len);
                uint32_t left = len;
                while (left && bc >= 8) { out.p[o++] = (uint16_t)take(8); left--; }
                if (left) {
                    bb = 0; bc = 0;
                    if ((size_t)(in_end - in) < left) break;
                    for (uint32_t i = 0; i < left; i++) out.p[o + i] = in[i];
                    in += left; o += left;
                }
                nb++;
                continue;
            }
            if (type == 1) { if (!fixed_tables()) break; }
            else if (type == 2) { if (!dynamic_tables()) break; }
            else break;
            bool err = false, eob = false;
            while (!eob) {
                if (o + 64 * 264 > limit) { err = true; break; }
                room(64 * 264);                                        // a stretch: at least 64 symbols of at most 258 + 7 (copy overshoot) each
                uint16_t *const ob = out.data();
                uint16_t *op = ob + o, *const op_stop = ob + out.cap - 272;
                // fast loop (the shape of Inflater::run's): 8 readable input bytes, room for the longest match
                bool have_e = false; uint32_t e = 0;
                while (op < op_stop && in_end - in >= 8) {
                    if (!have_e) { refill_fast(); e = lt[bb & ((1u << LT_BITS) - 1u)]; }
                    have_e = false;
                    if (__builtin_expect(e & F_SUB, 0)) { bb >>= LT_BITS; bc -= LT_BITS; e = lt[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))]; }
                    const uint64_t saved = bb;
                    bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                    if (e & F_LIT) {
                        *op++ = (uint16_t)(e >> 16);
                        e = lt[bb & ((1u << LT_BITS) - 1u)];            // up to three more first-level literals from the same refill
                        if (!(e & F_LIT)) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *op++ = (uint16_t)(e >> 16);
                        e = lt[bb & ((1u << LT_BITS) - 1u)];
                        if (!(e & F_LIT)) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *op++ = (uint16_t)(e >> 16);
                        e = lt[bb & ((1u << LT_BITS) - 1u)];
                        if (!(e & F_LIT)) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *op++ = (uint16_t)(e >> 16);
                        continue;
                    }
                    if (e & (F_EOB | F_BAD)) { if (e & F_BAD) err = true; else eob = true; break; }
                    const uint32_t xb = (e >> 8) & 15u;
                    uint32_t n = (e >> 16) + (uint32_t)((saved >> ((e & 0xFF) - xb)) & ((1u << xb) - 1u));
                    uint32_t d = dt[bb & ((1u << DT_BITS) - 1u)];
                    if (__builtin_expect(d & F_SUB, 0)) { bb >>= DT_BITS; bc -= DT_BITS; d = dt[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))]; }
                    if (d & F_BAD) { err = true; break; }
                    const uint64_t dsaved = bb;
                    bb >>= (d & 0xFF); bc -= (int)(d & 0xFF);
                    const uint32_t dxb = (d >> 8) & 15u;
                    const uint32_t dist = (d >> 16) + (uint32_t)((dsaved >> ((d & 0xFF) - dxb)) & ((1u << dxb) - 1u));
                    const size_t have = (size_t)(op - ob);
                    if (__builtin_expect(dist > have, 0)) {               // (part of) the source lies in the unknown window
                        if (dist - have > WSIZE) { err = true; break; }
                        uint32_t w = WSIZE - (uint32_t)(dist - have);     // window index of the first byte
                        while (n && w < WSIZE) { *op++ = (uint16_t)(MARK | w); w++; n--; }
                        if (!n) continue;
                    }
                    const uint16_t *src = op - dist;
                    if (in_end - in >= 8) { refill_fast(); e = lt[bb & ((1u << LT_BITS) - 1u)]; have_e = true; }
                    uint16_t *const end = op + n;
                    if (dist >= 8) {
                        do { memcpy(op, src, 16); op += 8; src += 8; } while (op < end);      // 8 symbols at a time
                    } else if (dist == 1) {
                        const uint64_t v = 0x0001000100010001ull * src[0];
                        do { memcpy(op, &v, 8); memcpy(op + 4, &v, 8); op += 8; } while (op < end);
                    } else if (dist >= 4) {
                        do { memcpy(op, src, 8); op += 4; src += 4; } while (op < end);
                    } else {
                        for (uint32_t i = 0; i < n; i++) op[i] = src[i];
                    }
                    op = end;
                }
                // careful steps: few input bytes left (the end of the data), or the stretch is used up: one symbol
                while (!err && !eob && op < op_stop && in_end - in < 8) {
                    refill_safe();
                    uint32_t e2 = lt[bb & ((1u << LT_BITS) - 1u)];
                    int used = 0;
                    if (e2 & F_SUB) { if (bc < LT_BITS) { err = true; break; } used = LT_BITS; e2 = lt[(e2 >> 16) + (uint32_t)((bb >> LT_BITS) & ((1u << ((e2 >> 8) & 15u)) - 1u))]; }
                    if (e2 & F_BAD) { err = true; break; }
                    used += (int)(e2 & 0xFF);
                    if (used > bc) { err = true; break; }
                    if (e2 & F_LIT) { bb >>= used; bc -= used; *op++ = (uint16_t)(e2 >> 16); continue; }
                    if (e2 & F_EOB) { bb >>= used; bc -= used; eob = true; break; }
                    const uint32_t xb = (e2 >> 8) & 15u;
                    uint32_t n = (e2 >> 16) + (uint32_t)((bb >> (used - (int)xb)) & ((1u << xb) - 1u));
                    bb >>= used; bc -= used;
                    refill_safe();
                    uint32_t d = dt[bb & ((1u << DT_BITS) - 1u)];
                    used = 0;
                    if (d & F_SUB) { if (bc < DT_BITS) { err = true; break; } used = DT_BITS; d = dt[(d >> 16) + (uint32_t)((bb >> DT_BITS) & ((1u << ((d >> 8) & 15u)) - 1u))]; }
                    if (d & F_BAD) { err = true; break; }
                    used += (int)(d & 0xFF);
                    if (used > bc) { err = true; break; }
                    const uint32_t dxb = (d >> 8) & 15u;
                    const uint32_t dist = (d >> 16) + (uint32_t)((bb >> (used - (int)dxb)) & ((1u << dxb) - 1u));
                    bb >>= used; bc -= used;
                    const size_t have = (size_t)(op - ob);
                    if (dist > have + WSIZE) { err = true; break; }
                    for (uint32_t i = 0; i < n; i++, op++) {
                        const size_t at = (size_t)(op - ob);
                        *op = dist > at ? (uint16_t)(MARK | (WSIZE - (uint32_t)(dist - at))) : op[-(ptrdiff_t)dist];
                    }
                }
                o = (size_t)(op - ob);
                if (err) break;
            }
            if (err) break;
            nb++;
        }
        out.n = o;
        if (blocks) *blocks = nb;
        return rc;
    }
};

// symbols -> bytes with the window the chunk was decoded without: lut[s] = s for a byte, the window's byte for a marker
// (64 KiB, built once per chunk by resolve_table from the WSIZE bytes in front of the chunk, the last one at w[WSIZE - 1])
inline void resolve_table(const uint8_t *w, uint8_t *lut)
{
    for (uint32_t i = 0; i < 256; i++) lut[i] = (uint8_t)i;
    memset(lut + 256, 0, 0x8000 - 256);
    memcpy(lut + 0x8000, w, Inflater::WSIZE);
}
inline void resolve_symbols(const uint16_t *sym, size_t n, const uint8_t *lut, uint8_t *dst)
{
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i hi = _mm_set1_epi16((short)0x8000);
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(sym + i)), b = _mm_loadu_si128((const __m128i *)(sym + i + 8));
        if (_mm_movemask_epi8(_mm_and_si128(_mm_or_si128(a, b), hi)) == 0) {                 // no marker among the 16
            _mm_storeu_si128((__m128i *)(dst + i), _mm_packus_epi16(a, b));
        } else {
            for (size_t k = i; k < i + 16; k++) dst[k] = lut[sym[k]];
        }
    }
#endif
    for (; i < n; i++) dst[i] = lut[sym[i]];
}

// ---- finding a block ------------------------------------------------------------------------------------------------
// First bit offset in [from_bit, to_bit) of [p, p + n) at which a non-final dynamic-Huffman block header parses, the
// block and the one behind it decode without an error (with an unknown window) and a third header follows; ~0 if none.
// `probe` is scratch (55 KB of tables).  False positives are possible (the caller's chain check catches them).
inline uint64_t find_block_start(const uint8_t *p, size_t n, uint64_t from_bit, uint64_t to_bit, SpecInflater &probe, RawBuf<uint16_t> &scratch)
{
    const uint64_t last = n >= 16 ? ((uint64_t)n - 16) * 8 : 0;       // a header needs a few bytes behind it anyway
    if (to_bit > last) to_bit = last;
    uint64_t cand = 0, cand_base = from_bit;                            // positions cand_base + i with bit i set: BFINAL = 0, BTYPE = 2 there
    uint64_t next_scan = from_bit;
    for (;;) {
        if (!cand) {
            // 56 positions at once: the three header bits are 0, 0, 1 (LSB first) where ~x & ~(x >> 1) & (x >> 2) is set
            if (next_scan >= to_bit) break;
            const uint8_t *q0 = p + (next_scan >> 3);
            uint64_t x; memcpy(&x, q0, 8);
            x >>= (next_scan & 7u);                                      // 57 valid bits at least -> 55 positions with all three bits known
            cand = ~x & ~(x >> 1) & (x >> 2) & ((1ull << 55) - 1ull);
            cand_base = next_scan; next_scan += 55;
            continue;
        }
        const uint64_t bit = cand_base + (uint64_t)__builtin_ctzll(cand);
        cand &= cand - 1ull;
        if (bit >= to_bit) break;
        // 74 bits from `bit`: BFINAL, BTYPE, HLIT, HDIST, HCLEN, then up to 19 code-length-code lengths of 3 bits
        const uint8_t *q = p + (bit >> 3);
        const uint32_t sh = (uint32_t)(bit & 7u);
        uint64_t lo; memcpy(&lo, q, 8);
        uint64_t hi; memcpy(&hi, q + 8, 8);
        const uint64_t v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
        if (((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) continue;
        const uint32_t hclen = (uint32_t)((v >> 13) & 15u) + 4u;
        // the code-length code must be complete (zlib rejects anything else): Kraft sum of 2^(7 - len) == 128
        const uint64_t w2 = sh ? (hi >> sh) : hi;                      // bits 64.. of the field (the 17 + 57 bits reach bit 74)
        uint32_t kraft = 0, nz = 0;
        for (uint32_t i = 0; i < hclen; i++) {
            const uint32_t at = 17u + 3u * i;
            const uint32_t l = at + 3 <= 64 ? (uint32_t)((v >> at) & 7u) : at >= 64 ? (uint32_t)((w2 >> (at - 64)) & 7u)
                                                                                    : (uint32_t)(((v >> at) | (w2 << (64 - at))) & 7u);
            if (l) { kraft += 128u >> l; nz++; }
            if (kraft > 128u) break;
        }
        if (kraft != 128u || nz < 2) continue;
        // the tables, then two blocks of symbols
        if (!probe.reset_at(p, n, bit, nullptr, 0)) continue;
        scratch.clear();
        uint32_t nb = 0;
        const Inflater::Status r = probe.run_spec(scratch, (size_t)64 << 20, 2, &nb);
        if (r == Inflater::ERR || nb < 2) continue;
        if (r == Inflater::BOUNDARY) {                                 // a third header must be there (type 3 is reserved)
            const uint64_t nx = probe.bit_pos();
            if ((nx >> 3) + 1 >= n) continue;
            const uint32_t hb = (uint32_t)((p[nx >> 3] | ((uint32_t)p[(nx >> 3) + 1] << 8)) >> (nx & 7u));
            if (((hb >> 1) & 3u) == 3u) continue;
        }
        return bit;
    }
    return ~0ull;
}

// ---- one member, decoded in rounds of n_threads chunks ----------------------------------------------------------
struct ParGunzip {
    const uint8_t *base = nullptr; size_t len = 0;         // the member's deflate data and everything behind it in the mapping
    uint64_t pos_bit = 0;                                  // true block boundary the next round starts at
    std::vector<uint8_t> window;                           // the (up to) 32 KiB of text in front of pos_bit
    RawBuf<uint8_t> stage; size_t stage_pos = 0;           // text of the last round not handed out yet
    bool finished = false, failed = false;                 // final block decoded / stream damaged (after the staged text)
    uint32_t crc = 0; uint64_t total = 0;                  // CRC-32 and length of all the text decoded so far (handed out or staged)
    uint64_t end_bit = 0;                                  // finished: the bit behind the final block
    int n_threads = 1;
    size_t chunk_bytes = (size_t)2 << 20;                  // compressed bytes per chunk (F2Q_GZ_CHUNK_KB)
    uint64_t rounds = 0, chunks_ok = 0, chunks_dropped = 0;   // diagnostics
    double ratio = 6.0;                                    // text bytes per compressed byte in the last round (first guess: FASTQ)
    double lead_ratio = 0.0;                               // extra input (in chunks) the first chunk of a round takes, see round()
    Inflater first;                                        // the first chunk of a round: known window, plain decoder
    std::vector<SpecInflater *> spec;
    std::vector<RawBuf<uint16_t> *> sym, scratch;
    RawBuf<uint8_t> text0;                                 // the first chunk's bytes

    ~ParGunzip() { for (auto *s : spec) delete s; for (auto *b : sym) delete b; for (auto *b : scratch) delete b; }

    void start(const uint8_t *p, size_t n, int threads)
    {
        base = p; len = n; pos_bit = 0; window.clear(); stage.clear(); stage_pos = 0; finished = failed = false; end_bit = 0;
        crc = 0; total = 0;
        n_threads = std::max(1, threads);
        { const char *e = getenv("F2Q_GZ_CHUNK_KB"); if (e && atol(e) >= 64) chunk_bytes = (size_t)atol(e) << 10; }
    }
    // first compressed byte behind the member's final block (valid once finished)
    const uint8_t *input_end() const { return base + ((end_bit + 7) >> 3); }

    // up to `room` bytes of text; DONE = the member is complete and everything has been handed out, ERR = damaged (the
    // text before the damage has been handed out), OUT_FULL = room used up
    Inflater::Status read(uint8_t *dst, size_t room, size_t *produced)
    {
        size_t n = 0;
        for (;;) {
            if (stage_pos < stage.size()) {
                const size_t k = std::min(room - n, stage.size() - stage_pos);
                copy_out(dst + n, stage.data() + stage_pos, k);
                stage_pos += k; n += k;
                if (stage_pos == stage.size()) { stage.clear(); stage_pos = 0; }
            }
            if (n == room && room) { *produced = n; return Inflater::OUT_FULL; }
            if (failed) { *produced = n; return Inflater::ERR; }
            if (finished) { *produced = n; return Inflater::DONE; }
            if (room == 0) { *produced = 0; return Inflater::OUT_FULL; }
            // a round's text goes straight into dst when it fits; when the caller's piece is partly filled and the next
            // round is not expected to fit, the piece ends here (a short read) instead of being topped up through `stage`
            if (n > 0 && (double)(room - n) < ratio * 1.15 * (double)chunk_bytes * ((double)n_threads + lead_ratio)) { *produced = n; return Inflater::OUT_FULL; }
            n += round(dst + n, room - n);
        }
    }

private:
    void copy_out(uint8_t *dst, const uint8_t *src, size_t k)
    {
        const size_t slice = (size_t)8 << 20;
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, k / slice));
        if (T == 1) { memcpy(dst, src, k); return; }
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back([=] { const size_t a = k * (size_t)t / (size_t)T, b = k * (size_t)(t + 1) / (size_t)T; memcpy(dst + a, src + a, b - a); });
        memcpy(dst, src, k / (size_t)T);
        for (auto &x : th) x.join();
    }

    struct Chunk {
        uint64_t start = ~0ull, end = 0;           // bit offsets: where it started, the boundary it reached
        Inflater::Status rc = Inflater::ERR;
        size_t n_out = 0;                          // symbols (bytes for chunk 0)
    };

    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // decodes the next round; returns the bytes written to dst (0: the text is in `stage`)
    size_t round(uint8_t *dst, size_t dst_room)
    {
        rounds++;
        const bool trace = getenv("F2Q_GZ_TRACE") != nullptr;
        const double t_0 = now_s();
        const int T = n_threads;
        const uint64_t total_bits = (uint64_t)len * 8;
        while ((int)spec.size() < T) { spec.push_back(new SpecInflater()); sym.push_back(new RawBuf<uint16_t>()); scratch.push_back(new RawBuf<uint16_t>()); }
        std::vector<Chunk> ch((size_t)T);
        ch[0].start = pos_bit;
        const uint64_t first_byte = pos_bit >> 3;
        // chunk 0 knows its window and runs the plain decoder, 2.2x as fast as the speculative one and without a search:
        // it takes 2.2 chunks' worth of input so that the workers finish together
        // (lead_ratio: how much faster it was than the speculative decoders in the rounds so far, measured -- 1.2 on the
        //  GPU box's host, 0 .. 0.5 on slower machines)
        const uint64_t lead = T > 1 ? (uint64_t)((double)chunk_bytes * lead_ratio) : 0;
        // ---- 1 + 2. every worker finds where its chunk starts, then decodes it up to the start behind it ------------
        std::vector<uint64_t> stop((size_t)T, ~0ull);
        std::atomic<int> found{1};                              // starts known so far (chunk 0's is)
        const size_t sym_limit = std::max<size_t>((size_t)64 << 20, chunk_bytes * 64);
        double t_1 = t_0;
        std::vector<double> spent((size_t)T, 0.0);
        auto work = [&](int i) {
            Chunk &c = ch[(size_t)i];
            const double w_0 = now_s();
            struct Stamp { double &d, t0; ~Stamp() { d = now_s() - t0; } } stamp{spent[(size_t)i], w_0};
            if (i > 0) {
                const uint64_t from = (first_byte + lead + (uint64_t)i * chunk_bytes) * 8, to = from + (uint64_t)chunk_bytes * 8;
                if (from < total_bits) c.start = find_block_start(base, len, from, std::min(to, total_bits), *spec[(size_t)i], *scratch[(size_t)i]);
                found.fetch_add(1, std::memory_order_release);
            }
            while (found.load(std::memory_order_acquire) < T) std::this_thread::yield();     // (a few milliseconds at most)
            if (i == 0) { t_1 = now_s(); stamp.t0 = t_1; }            // (chunk 0 has only waited so far)
            // this chunk runs to the next start that was found; the last one to the first boundary behind the round's stretch
            uint64_t stop_at = (first_byte + lead + (uint64_t)T * chunk_bytes) * 8;
            for (int j = T - 1; j > i; j--) if (ch[(size_t)j].start != ~0ull) stop_at = ch[(size_t)j].start;
            stop[(size_t)i] = stop_at;
            if (c.start == ~0ull) return;
            if (i == 0) {
                if (!first.reset_at(base, len, c.start, window.data(), (uint32_t)window.size())) { c.rc = Inflater::ERR; return; }
                first.stop_bit = stop_at;
                text0.reserve(std::max<size_t>((size_t)((double)chunk_bytes * ratio * 1.25 * (1.0 + lead_ratio)), (size_t)1 << 20));
                size_t o = 0;
                for (;;) {
                    size_t got = 0;
                    const Inflater::Status r = first.run(text0.data() + o, text0.data() + text0.cap, &got);
                    o += got;
                    if (r == Inflater::OUT_FULL) { text0.reserve(text0.cap * 2); continue; }
                    c.rc = r; break;
                }
                text0.n = o; c.n_out = o; c.end = first.bit_pos();
                return;
            }
            SpecInflater &d = *spec[(size_t)i];
            if (!d.reset_at(base, len, c.start, nullptr, 0)) { c.rc = Inflater::ERR; return; }
            d.stop_bit = stop_at;
            sym[(size_t)i]->clear();
            sym[(size_t)i]->reserve(chunk_bytes * 5);
            c.rc = d.run_spec(*sym[(size_t)i], sym_limit, ~0u, nullptr);
            c.n_out = sym[(size_t)i]->size(); c.end = d.bit_pos();
        };
        {
            std::vector<std::thread> th;
            for (int i = 1; i < T; i++) th.emplace_back(work, i);
            work(0);
            for (auto &x : th) x.join();
        }
        const double t_2 = now_s();
        if (T > 1 && ch[0].end > ch[0].start && spent[0] > 0) {
            // bytes of input per second: the plain decoder of chunk 0 against the mean of the speculative ones (search included)
            double rs = 0; int ns = 0;
            for (int i = 1; i < T; i++)
                if (ch[(size_t)i].start != ~0ull && ch[(size_t)i].end > ch[(size_t)i].start && spent[(size_t)i] > 0 && ch[(size_t)i].rc != Inflater::ERR) { rs += (double)(ch[(size_t)i].end - ch[(size_t)i].start) / spent[(size_t)i]; ns++; }
            if (ns) {
                const double r0 = (double)(ch[0].end - ch[0].start) / spent[0];
                const double want = std::min(3.0, std::max(0.0, r0 / (rs / ns) - 1.0));
                lead_ratio = 0.5 * lead_ratio + 0.5 * want;
            }
        }
        // ---- 3. the chain: which chunks stand ----------------------------------------------------------------
        int last = 0;                                          // index of the last chunk whose text is kept
        bool ended = ch[0].rc != Inflater::BOUNDARY;
        for (int i = 1; i < T && !ended; i++) {
            if (ch[(size_t)i].start == ~0ull) continue;                              // nothing found there: the chunk before ran through
            if (ch[(size_t)i].start != ch[(size_t)last].end) break;                  // guessed wrong (or not reached): everything from here is dropped
            last = i;
            ended = ch[(size_t)i].rc != Inflater::BOUNDARY;
        }
        for (int i = 0; i < T; i++) if (ch[(size_t)i].start != ~0ull) { if (i <= last) chunks_ok++; else chunks_dropped++; }
        // ---- 4. windows, then bytes --------------------------------------------------------------------------
        std::vector<int> keep;
        for (int i = 0; i <= last; i++) if (i == 0 || ch[(size_t)i].start != ~0ull) keep.push_back(i);
        std::vector<size_t> off(keep.size() + 1, 0);
        for (size_t k = 0; k < keep.size(); k++) off[k + 1] = off[k] + ch[(size_t)keep[k]].n_out;
        const bool direct = off.back() <= dst_room;
        uint8_t *const outp = direct ? dst : (stage.reserve(off.back()), stage.data());
        stage.n = direct ? 0 : off.back(); stage_pos = 0;
        std::vector<uint32_t> ccrc(keep.size(), 0);
        // window in front of every kept chunk: WSIZE bytes, right-aligned (bytes that do not exist are never referenced
        // by a valid stream; they read as 0)
        std::vector<std::vector<uint8_t>> win(keep.size());
        std::vector<uint8_t> cur(Inflater::WSIZE, 0);
        if (!window.empty()) memcpy(cur.data() + (Inflater::WSIZE - window.size()), window.data(), window.size());
        auto push_tail = [&](const uint8_t *bytes, size_t n) {      // cur := last WSIZE bytes of (cur ++ bytes)
            if (n >= Inflater::WSIZE) memcpy(cur.data(), bytes + (n - Inflater::WSIZE), Inflater::WSIZE);
            else { memmove(cur.data(), cur.data() + n, Inflater::WSIZE - n); memcpy(cur.data() + (Inflater::WSIZE - n), bytes, n); }
        };
        std::vector<uint8_t> tail(Inflater::WSIZE), lut(65536);
        for (size_t k = 0; k < keep.size(); k++) {
            const int i = keep[k];
            win[k] = cur;
            const size_t n = ch[(size_t)i].n_out, t = std::min<size_t>(n, Inflater::WSIZE);
            if (i == 0) push_tail(text0.data() + (n - t), t);
            else {
                resolve_table(win[k].data(), lut.data());
                resolve_symbols(sym[(size_t)i]->data() + (n - t), t, lut.data(), tail.data());
                // (a marker in the tail that points into the part of the window the tail itself pushes out is fine: it was resolved against win[k])
                push_tail(tail.data(), t);
            }
        }
        auto resolve = [&](size_t k) {
            // in blocks that stay in the cache: symbols -> bytes, then the CRC of those bytes
            const int i = keep[k];
            const size_t n = ch[(size_t)i].n_out, B = (size_t)32 << 10;
            uint32_t c = 0;
            std::vector<uint8_t> tab;
            if (i != 0) { tab.resize(65536); resolve_table(win[k].data(), tab.data()); }
            for (size_t a = 0; a < n; a += B) {
                const size_t m = std::min(B, n - a);
                if (i == 0) memcpy(outp + off[k] + a, text0.data() + a, m);
                else resolve_symbols(sym[(size_t)i]->data() + a, m, tab.data(), outp + off[k] + a);
                c = Crc32::get().update(c, outp + off[k] + a, m);
            }
            ccrc[k] = c;
        };
        {
            std::vector<std::thread> th;
            for (size_t k = 1; k < keep.size(); k++) th.emplace_back(resolve, k);
            resolve(0);
            for (auto &x : th) x.join();
        }
        if (trace) {
            fprintf(stderr, "[pargz] round %llu: search %.1f ms, decode %.1f ms, resolve %.1f ms; kept %zu of %d chunks, %zu bytes;", (unsigned long long)rounds,
                    (t_1 - t_0) * 1e3, (t_2 - t_1) * 1e3, (now_s() - t_2) * 1e3, keep.size(), T, off.back());
            for (int i = 0; i < T; i++) fprintf(stderr, " [%d: %s start %lld end %lld out %zu]", i, ch[(size_t)i].rc == Inflater::BOUNDARY ? "B" : ch[(size_t)i].rc == Inflater::DONE ? "D" : "E",
                                                ch[(size_t)i].start == ~0ull ? -1ll : (long long)(ch[(size_t)i].start - pos_bit), (long long)(ch[(size_t)i].end - pos_bit), ch[(size_t)i].n_out);
            fprintf(stderr, "\n");
        }
        for (size_t k = 0; k < keep.size(); k++) {
            const size_t n = ch[(size_t)keep[k]].n_out;
            if (!n) continue;
            crc = total ? (uint32_t)crc32_combine(crc, ccrc[k], (z_off_t)n) : ccrc[k];
            total += n;
        }
        // ---- 5. where the stream stands now ---------------------------------------------------------------------
        const Chunk &lc = ch[(size_t)last];
        if (lc.end > ch[0].start + 8 && off.back()) ratio = (double)off.back() * 8.0 / (double)(lc.end - ch[0].start);
        window.assign(cur.begin() + (ptrdiff_t)(Inflater::WSIZE - std::min<size_t>(Inflater::WSIZE, window.size() + off.back())), cur.end());
        pos_bit = lc.end;
        if (lc.rc == Inflater::DONE) { finished = true; end_bit = lc.end; }
        else if (lc.rc == Inflater::ERR) failed = true;
        else if (off.back() == 0 && lc.end == ch[0].start) failed = true;       // no progress: cannot happen on a BOUNDARY stop; stay safe
        return direct ? off.back() : 0;
    }
};

} // namespace f2qz
