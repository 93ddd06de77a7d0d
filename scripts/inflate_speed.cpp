#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#include <zlib.h>
#include "../2fast2q_amd/csrc/f2q_inflate.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    int level = argc > 1 ? atoi(argv[1]) : 6;
    std::vector<uint8_t> data; srand(1);
    const char *guides[64]; char gb[64][21]; for (int g = 0; g < 64; g++) { for (int j = 0; j < 20; j++) gb[g][j] = "ACGT"[rand() & 3]; gb[g][20] = 0; guides[g] = gb[g]; }
    for (int i = 0; i < 400000; i++) { char h[64]; int n = sprintf(h, "@SRR1234567.%d %d/1\n", i, i); data.insert(data.end(), h, h + n); for (int j = 0; j < 150; j++) data.push_back(j >= 30 && j < 50 ? guides[i & 63][j - 30] : "ACGT"[rand() & 3]); data.push_back('\n'); data.push_back('+'); data.push_back('\n'); for (int j = 0; j < 150; j++) data.push_back((rand() % 20) ? 'I' : (char)('#' + rand() % 40)); data.push_back('\n'); }
    std::vector<uint8_t> comp(data.size() + 100000);
    z_stream zs = {}; deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = data.data(); zs.avail_in = data.size(); zs.next_out = comp.data(); zs.avail_out = comp.size(); deflate(&zs, Z_FINISH); size_t clen = zs.total_out; deflateEnd(&zs);
    printf("level %d: %zu -> %zu (%.2fx)\n", level, data.size(), clen, (double)data.size() / clen);
    std::vector<uint8_t> out(data.size() + 64);
    double bz = 0, bo = 0; for (int rep = 0; rep < 7; rep++) {
        double t0 = now(); z_stream zi = {}; inflateInit2(&zi, -15); zi.next_in = comp.data(); zi.avail_in = clen; zi.next_out = out.data(); zi.avail_out = out.size(); inflate(&zi, Z_FINISH); inflateEnd(&zi); double t1 = now();
        f2qz::Inflater *inf = new f2qz::Inflater(); inf->reset(comp.data(), clen); size_t got = 0; auto r = inf->run(out.data(), out.data() + out.size(), &got); double t2 = now();
        printf("  zlib %.0f MB/s   ours %.0f MB/s (%s, %zu)\n", data.size() / (t1 - t0) / 1e6, data.size() / (t2 - t1) / 1e6, r == f2qz::Inflater::DONE && got == data.size() && !memcmp(out.data(), data.data(), got) ? "ok" : "BAD", got);
        delete inf; if (data.size() / (t1 - t0) > bz) bz = data.size() / (t1 - t0); if (data.size() / (t2 - t1) > bo) bo = data.size() / (t2 - t1);
    }
    printf("BEST zlib %.0f ours %.0f ratio %.2f\n", bz/1e6, bo/1e6, bo/bz);
}
