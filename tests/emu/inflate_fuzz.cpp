// Checker for 2fast2q_amd/csrc/f2q_inflate.h (the DEFLATE decoder of the file reader) against zlib: random data of
// several kinds deflated with every level / strategy / window size / flush pattern and decoded in pieces of random
// size; then damaged copies (bit flips, cut-offs): the decoder must agree with zlib on accept/reject, deliver the
// same bytes, consume the same input, and never touch memory outside its buffers (build with -fsanitize=address,undefined;
// every input and output buffer is an exact-size heap block).  usage: inflate_fuzz <iterations> [seed]
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <zlib.h>
#include "../../2fast2q_amd/csrc/f2q_inflate.h"
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
static std::vector<uint8_t> make(int kind, size_t n)
{
    std::vector<uint8_t> d(n);
    switch (kind) {
    case 0: for (auto &c : d) c = (uint8_t)rnd(); break;                               // incompressible
    case 1: for (auto &c : d) c = "ACGT"[rnd() & 3]; break;
    case 2: for (size_t i = 0; i < n; i++) d[i] = (uint8_t)(i % 7 == 0 ? rnd() : 'I'); break;   // long runs
    case 3: { size_t i = 0; while (i < n) { size_t L = 1 + rnd() % 300, back = i ? 1 + rnd() % (i < 40000 ? i : 40000) : 0; for (size_t j = 0; j < L && i < n; j++, i++) d[i] = (back && (rnd() & 7)) ? d[i - back] : (uint8_t)rnd(); } } break;
    case 4: for (auto &c : d) c = 0; break;
    default: for (size_t i = 0; i < n; i++) d[i] = (uint8_t)((i * 2654435761u) >> (rnd() % 3 ? 24 : 28)); break;
    }
    return d;
}
int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 300;
    if (argc > 2) rs ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
    f2qz::Inflater *inf = new f2qz::Inflater();
    long ok = 0, damaged_ok = 0, damaged_err = 0;
    for (int it = 0; it < iters; it++) {
        const int kind = rnd() % 6;
        size_t n = (rnd() % 8 == 0) ? rnd() % 5 : (rnd() % 4 == 0 ? rnd() % 600000 : rnd() % 20000);
        std::vector<uint8_t> data = make(kind, n);
        const int level = rnd() % 10, strat = (int[]){Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}[rnd() % 5];
        const int wbits = 9 + rnd() % 7, memlevel = 1 + rnd() % 9;
        std::vector<uint8_t> comp(n + n / 4 + 70000);
        z_stream zs = {}; if (deflateInit2(&zs, level, Z_DEFLATED, -wbits, memlevel, strat) != Z_OK) { printf("deflateInit2 failed\n"); return 1; }
        zs.next_in = data.data(); zs.avail_in = n; zs.next_out = comp.data(); zs.avail_out = comp.size();
        // sprinkle flush points (empty stored blocks, block splits)
        if (n > 10 && rnd() % 3 == 0) { zs.avail_in = n / 3; deflate(&zs, (rnd() & 1) ? Z_FULL_FLUSH : Z_SYNC_FLUSH); zs.avail_in = n - n / 3; }
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { printf("deflate failed\n"); return 1; }
        size_t clen = zs.total_out; deflateEnd(&zs);
        // exact-size heap copy of the input so that ASan sees any over-read
        uint8_t *cin = (uint8_t *)malloc(clen ? clen : 1); memcpy(cin, comp.data(), clen);
        inf->reset(cin, clen);
        std::vector<uint8_t> out;
        bool fail = false;
        for (;;) {
            size_t room = (rnd() % 5 == 0) ? 1 + rnd() % 16 : (rnd() % 2 ? 1 + rnd() % 5000 : 1 + rnd() % 200000);
            uint8_t *buf = (uint8_t *)malloc(room);
            size_t got = 0; auto r = inf->run(buf, buf + room, &got);
            out.insert(out.end(), buf, buf + got); free(buf);
            if (r == f2qz::Inflater::ERR) { fail = true; break; }
            if (r == f2qz::Inflater::DONE) break;
            if (got != room) { printf("iter %d: short piece\n", it); return 1; }
            if (out.size() > n + 10) { fail = true; break; }
        }
        if (fail || out != data || inf->input_pos() != cin + clen) { printf("iter %d MISMATCH kind %d n %zu level %d strat %d wbits %d fail %d outsz %zu\n", it, kind, n, level, strat, wbits, fail, out.size()); return 1; }
        ok++;
        // damaged copy: flip bits / truncate; must not crash, and must agree with zlib when zlib accepts
        for (int rep = 0; rep < 3 && clen > 2; rep++) {
            size_t dl = (rnd() & 1) ? clen : 1 + rnd() % clen;
            uint8_t *bad = (uint8_t *)malloc(dl); memcpy(bad, cin, dl);
            for (int f = 0, nf = rnd() % 4; f < nf; f++) bad[rnd() % dl] ^= (uint8_t)(1u << (rnd() & 7));
            std::vector<uint8_t> zo(n + 70000);
            z_stream zi = {}; inflateInit2(&zi, -15); zi.next_in = bad; zi.avail_in = dl; zi.next_out = zo.data(); zi.avail_out = zo.size();
            int zr = inflate(&zi, Z_FINISH); size_t zn = zi.total_out; size_t zused = zi.total_in; inflateEnd(&zi);
            inf->reset(bad, dl);
            std::vector<uint8_t> mo; bool merr = false, mdone = false;
            for (;;) {
                size_t room = 1 + rnd() % 50000; uint8_t *buf = (uint8_t *)malloc(room);
                size_t got = 0; auto r = inf->run(buf, buf + room, &got);
                mo.insert(mo.end(), buf, buf + got); free(buf);
                if (r == f2qz::Inflater::ERR) { merr = true; break; }
                if (r == f2qz::Inflater::DONE) { mdone = true; break; }
                if (mo.size() > n + 70000) break;
            }
            if (zr == Z_STREAM_END) {
                if (!mdone || mo.size() != zn || (zn && memcmp(mo.data(), zo.data(), zn) != 0) || (size_t)(inf->input_pos() - bad) != zused) { printf("iter %d damaged: zlib accepts (%zu bytes), ours %s %zu\n", it, zn, mdone ? "done" : (merr ? "err" : "open"), mo.size()); return 1; }
                damaged_ok++;
            } else {
                if (mdone) { printf("iter %d damaged: zlib rejects (%d) but ours DONE\n", it, zr); return 1; }
                // what we delivered before the error must be a prefix of what zlib delivered, or vice versa
                size_t m = mo.size() < zn ? mo.size() : zn;
                if (m && memcmp(mo.data(), zo.data(), m) != 0) { printf("iter %d damaged: prefix differs\n", it); return 1; }
                damaged_err++;
            }
            free(bad);
        }
        free(cin);
    }
    printf("ok %ld, damaged accepted-by-both %ld, rejected-by-both %ld\n", ok, damaged_ok, damaged_err);
}
