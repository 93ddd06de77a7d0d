// Checker for 2fast2q_amd/csrc/f2q_pargz.h (one gzip member decoded by several threads: block search, decoding with an
// unknown window, chained resolution) against zlib: data of several kinds deflated with every level / strategy / flush
// pattern, decoded with 1..5 threads in chunks of a few KiB (many chunks, rounds, guessed starts) and handed out in
// pieces of random size; then damaged copies (bit flips, cut-offs): never a crash or an out-of-range access (build with
// -fsanitize=address,undefined; inputs are exact-size heap blocks), the same accept / reject verdict as zlib, and
// the bytes delivered before an error are the bytes zlib delivers.  usage: pargz_fuzz <iterations> [seed]
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <zlib.h>
#include "../../2fast2q_amd/csrc/f2q_pargz.h"
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
static std::vector<uint8_t> make(int kind, size_t n)
{
    std::vector<uint8_t> d(n);
    switch (kind) {
    case 0: for (auto &c : d) c = (uint8_t)rnd(); break;
    case 1: for (auto &c : d) c = "ACGT"[rnd() & 3]; break;
    case 2: for (size_t i = 0; i < n; i++) d[i] = (uint8_t)(i % 7 == 0 ? rnd() : 'I'); break;
    case 3: { size_t i = 0; while (i < n) { size_t L = 1 + rnd() % 300, back = i ? 1 + rnd() % (i < 40000 ? i : 40000) : 0; for (size_t j = 0; j < L && i < n; j++, i++) d[i] = (back && (rnd() & 7)) ? d[i - back] : (uint8_t)rnd(); } } break;
    case 4: {   // FASTQ-like records
        size_t i = 0; unsigned r = 0;
        while (i < n) {
            char h[64]; int hl = snprintf(h, sizeof h, "@m:%u:%u\n", r / 100, r); r++;
            const size_t L = 20 + rnd() % 130;
            for (int k = 0; k < hl && i < n; k++) d[i++] = (uint8_t)h[k];
            for (size_t k = 0; k < L && i < n; k++) d[i++] = "ACGTN"[rnd() % 5];
            if (i < n) d[i++] = '\n'; if (i < n) d[i++] = '+'; if (i < n) d[i++] = '\n';
            for (size_t k = 0; k < L && i < n; k++) d[i++] = "IIIIII?5#F"[rnd() % 10];
            if (i < n) d[i++] = '\n';
        }
    } break;
    default: for (size_t i = 0; i < n; i++) d[i] = (uint8_t)((i * 2654435761u) >> (rnd() % 3 ? 24 : 28)); break;
    }
    return d;
}
// read everything in random pieces; returns DONE / ERR
static f2qz::Inflater::Status drain(f2qz::ParGunzip &pg, std::vector<uint8_t> &out, size_t cap)
{
    for (;;) {
        const size_t room = (rnd() % 5 == 0) ? 1 + rnd() % 64 : (rnd() % 2 ? 1 + rnd() % 20000 : 1 + rnd() % 3000000);
        uint8_t *buf = (uint8_t *)malloc(room);
        size_t got = 0;
        const f2qz::Inflater::Status r = pg.read(buf, room, &got);
        out.insert(out.end(), buf, buf + got); free(buf);
        if (r != f2qz::Inflater::OUT_FULL) return r;
        if (got == 0) { printf("no progress\n"); exit(1); }          // (a piece may end early; it must not be empty)
        if (out.size() > cap) return f2qz::Inflater::ERR;
    }
}
int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 100;
    if (argc > 2) rs ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
    f2qz::ParGunzip *pg = new f2qz::ParGunzip();
    long ok = 0, damaged_ok = 0, damaged_err = 0; unsigned long long dropped = 0, kept = 0;
    for (int it = 0; it < iters; it++) {
        const int kind = rnd() % 6;
        const size_t n = (rnd() % 10 == 0) ? rnd() % 50 : (rnd() % 3 == 0 ? 200000 + rnd() % 1500000 : rnd() % 120000);
        std::vector<uint8_t> data = make(kind, n);
        const int level = rnd() % 10, strat = (int[]){Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}[rnd() % 5];
        const int memlevel = 1 + rnd() % 9;
        std::vector<uint8_t> comp(n + n / 4 + 70000);
        z_stream zs = {}; if (deflateInit2(&zs, level, Z_DEFLATED, -15, memlevel, strat) != Z_OK) return 1;
        zs.next_in = data.data(); zs.avail_in = n; zs.next_out = comp.data(); zs.avail_out = comp.size();
        if (n > 10 && rnd() % 3 == 0) { zs.avail_in = n / 3; deflate(&zs, (rnd() & 1) ? Z_FULL_FLUSH : Z_SYNC_FLUSH); zs.avail_in = n - n / 3; }
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return 1;
        const size_t clen = zs.total_out; deflateEnd(&zs);
        const size_t tail = rnd() % 3 ? 8 + rnd() % 40 : 0;                       // what follows the member in a file: trailer, more members
        uint8_t *cin = (uint8_t *)malloc(clen + tail ? clen + tail : 1); memcpy(cin, comp.data(), clen);
        for (size_t i = 0; i < tail; i++) cin[clen + i] = (uint8_t)rnd();
        const int threads = 1 + rnd() % 5;
        pg->start(cin, clen + tail, threads);
        pg->chunk_bytes = (size_t[]){700, 3000, 16000, 70000, 1u << 20}[rnd() % 5];
        std::vector<uint8_t> out;
        const f2qz::Inflater::Status r = drain(*pg, out, n + 10);
        const uint32_t want_crc = (uint32_t)crc32(0, data.data(), (uInt)n);
        if (r != f2qz::Inflater::DONE || out != data || pg->input_end() != cin + clen || pg->total != n || (n && pg->crc != want_crc)) {
            printf("iter %d MISMATCH kind %d n %zu level %d strat %d threads %d chunk %zu: status %d out %zu end %ld crc %08x/%08x\n", it, kind, n, level, strat, threads,
                   pg->chunk_bytes, (int)r, out.size(), (long)(pg->input_end() - cin) - (long)clen, pg->crc, want_crc);
            return 1;
        }
        ok++;
        for (int rep = 0; rep < 2 && clen > 2; rep++) {
            const size_t dl = (rnd() & 1) ? clen : 1 + rnd() % clen;
            uint8_t *bad = (uint8_t *)malloc(dl); memcpy(bad, cin, dl);
            for (int f = 0, nf = rnd() % 4; f < nf; f++) bad[rnd() % dl] ^= (uint8_t)(1u << (rnd() & 7));
            std::vector<uint8_t> zo(n + 70000);
            z_stream zi = {}; inflateInit2(&zi, -15); zi.next_in = bad; zi.avail_in = (uInt)dl; zi.next_out = zo.data(); zi.avail_out = (uInt)zo.size();
            const int zr = inflate(&zi, Z_FINISH); const size_t zn = zi.total_out, zused = zi.total_in; inflateEnd(&zi);
            pg->start(bad, dl, 1 + rnd() % 4);
            pg->chunk_bytes = (size_t[]){700, 3000, 16000, 70000}[rnd() % 4];
            std::vector<uint8_t> mo;
            const f2qz::Inflater::Status mr = drain(*pg, mo, n + 70000);
            if (zr == Z_STREAM_END) {
                if (mr != f2qz::Inflater::DONE || mo.size() != zn || (zn && memcmp(mo.data(), zo.data(), zn) != 0) || (size_t)(pg->input_end() - bad) != zused) {
                    printf("iter %d damaged: zlib accepts (%zu bytes), ours %d %zu\n", it, zn, (int)mr, mo.size()); return 1; }
                damaged_ok++;
            } else {
                if (mr == f2qz::Inflater::DONE) { printf("iter %d damaged: zlib rejects (%d) but ours DONE\n", it, zr); return 1; }
                const size_t m = mo.size() < zn ? mo.size() : zn;
                if (m && memcmp(mo.data(), zo.data(), m) != 0) { printf("iter %d damaged: prefix differs\n", it); return 1; }
                damaged_err++;
            }
            free(bad);
        }
        dropped += pg->chunks_dropped; kept += pg->chunks_ok;
        free(cin);
    }
    printf("ok %ld, damaged accepted-by-both %ld, rejected-by-both %ld (chunks kept %llu, dropped %llu in the last runs)\n", ok, damaged_ok, damaged_err, kept, dropped);
}
