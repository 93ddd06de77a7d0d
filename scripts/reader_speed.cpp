// host-only timing of the file reader (f2q_reader.h): decoded bytes per second into ordinary memory
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include "../2fast2q_amd/csrc/f2q_reader.h"
int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    const size_t cap = (size_t)256 << 20;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (int rep = 0; rep < 3; rep++) {
        TextSource src; std::string err;
        if (src.open(argv[1], err)) { printf("%s\n", err.c_str()); return 1; }
        auto t0 = std::chrono::steady_clock::now();
        size_t total = 0, n;
        while ((n = src.read(buf, cap)) > 0) total += n;
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s (%s, %s): %zu bytes in %.3f s = %.0f MB/s%s\n", argv[1], src.kind_name(), src.zmap ? "own inflate" : "zlib", total, dt, total / dt / 1e6, src.truncated() ? " TRUNCATED" : "");
    }
}
