// What the tile layout's access pattern can reach on its own: a kernel that only FETCHES what k_count_fixed4_lds fetches
// (per 256-read tile: base rows 0-1 of a 10-row tile, quality rows 0-4 of a 38-row tile, the 512-byte length row; one
// 16-byte load per lane and row, a wave per tile, 16 waves per workgroup, one workgroup per CU, the next tile's rows
// requested before this tile's are consumed) and XORs it together.  Variants: the same bytes as ONE contiguous run per
// tile (a hypothetical layout), and a plain streaming read of the same volume.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/row_pattern scripts/row_pattern.hip && /tmp/row_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int DEPTH, int WORK, int WARM>      // WORK: integer instructions per tile beside the fetch; WARM: touch every line of the tile WARM tiles ahead (one dword per line)
// MODE 0: tile layout; 1: one 7.5-KiB run per tile; 2: streaming (tile t = 7.5 KiB at t * 7.5 KiB)
__global__ __launch_bounds__(1024) void k_rows(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ qual,
                                               const uint32_t *__restrict__ len, uint32_t n_tiles, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t stride = gridDim.x * 16u;
    v4 acc = {0, 0, 0, 0};
    struct Rows { v4 r[7]; uint32_t l0, l1; };
    auto request = [&](Rows &r, uint32_t t) {
        t = t < n_tiles ? t : n_tiles - 1u;
        if (MODE == 0) {
            const uint32_t *bp = bases + (uint64_t)t * 10u * 256u + 4u * lane, *qp = qual + (uint64_t)t * 38u * 256u + 4u * lane;
#pragma unroll
            for (int i = 0; i < 2; i++) r.r[i] = __builtin_nontemporal_load((const v4 *)(bp + i * 256));
#pragma unroll
            for (int i = 0; i < 5; i++) r.r[2 + i] = __builtin_nontemporal_load((const v4 *)(qp + i * 256));
            const uint2 lv = *(const uint2 *)(len + (uint64_t)t * 128u + 2u * lane);
            r.l0 = lv.x; r.l1 = lv.y;
        } else {
            const uint32_t *p = bases + (uint64_t)t * (MODE == 1 ? 12288u : 1920u) + 4u * lane;   // 1: runs 48 KiB apart; 2: back to back
#pragma unroll
            for (int i = 0; i < 7; i++) r.r[i] = __builtin_nontemporal_load((const v4 *)(p + i * 256));
            const uint2 lv = *(const uint2 *)(p + 7 * 256 - 2u * lane);
            r.l0 = lv.x; r.l1 = lv.y;
        }
    };
    uint32_t warm_acc = 0;
    auto warm = [&](uint32_t t) {                     // 60 lines of 128 B per tile: lanes 0-15 the base rows, 16-55 the quality rows, 56-59 the lengths
        t = t < n_tiles ? t : n_tiles - 1u;
        const uint32_t *p = lane < 16u ? bases + (uint64_t)t * 10u * 256u + lane * 32u
                          : lane < 56u ? qual + (uint64_t)t * 38u * 256u + (lane - 16u) * 32u
                                       : len + (uint64_t)t * 128u + ((lane - 56u) & 3u) * 32u;
        warm_acc ^= *p;
    };
    auto consume = [&](const Rows &r) {
#pragma unroll
        for (int i = 0; i < 7; i++) acc ^= r.r[i];
        acc.x ^= r.l0; acc.y ^= r.l1;
        uint32_t a = acc.x, b = acc.y;
#pragma unroll 8
        for (int i = 0; i < WORK / 4; i++) { a += b; b ^= a; a = (a << 5) | (a >> 27); b += 0x9E3779B9u; }   // WORK full-rate integer instructions
        acc.z ^= a; acc.w ^= b;
    };
    Rows q[DEPTH + 1];
    uint32_t t = blockIdx.x * 16u + wave;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) request(q[d], t + d * stride);
    for (; t < n_tiles; t += (DEPTH + 1) * stride) {
#pragma unroll
        for (int k = 0; k <= DEPTH; k++) {
            if (t + k * stride < n_tiles) {
                if (WARM) warm(t + (k + DEPTH + WARM) * stride);
                request(q[(k + DEPTH) % (DEPTH + 1)], t + (k + DEPTH) * stride);
                consume(q[k]);
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ warm_acc) == 0x12345u) out[0] = 1;       // (keeps the loads)
}

template <int MODE, int DEPTH, int WORK = 0, int WARM = 0>
static int run(const char *what, const uint32_t *b, const uint32_t *q, const uint32_t *l, uint32_t n_tiles, uint32_t *out, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_rows<MODE, DEPTH, WORK, WARM>), dim3(grid), dim3(1024), 0, 0, b, q, l, n_tiles, out);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_rows<MODE, DEPTH, WORK, WARM>), dim3(grid), dim3(1024), 0, 0, b, q, l, n_tiles, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)n_tiles * 7680.0;
    printf("%-64s grid %4d  %7.1f us  %6.0f GB/s  (%.2f of 8 TB/s)\n", what, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12);
    return 0;
}

int main()
{
    const uint32_t n_tiles = 195313;                                  // 50 M reads
    uint32_t *b, *q, *l, *out;
    CK(hipMalloc(&b, (size_t)n_tiles * 49152)); CK(hipMalloc(&q, (size_t)n_tiles * 38 * 1024)); CK(hipMalloc(&l, (size_t)n_tiles * 512)); CK(hipMalloc(&out, 4));
    CK(hipMemset(b, 1, (size_t)n_tiles * 49152)); CK(hipMemset(q, 2, (size_t)n_tiles * 38 * 1024)); CK(hipMemset(l, 3, (size_t)n_tiles * 512));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cu = pr.multiProcessorCount;
    printf("%s, %d CUs; 50 M reads = %u tiles, 7680 B fetched per tile (30 B/read)\n", pr.name, cu, n_tiles);
    // the same with work between the fetches (the counting kernels issue 530-700 vector instructions per tile and wave)
    if (run<0, 1, 300>("tile layout, 1 tile ahead, 300 instructions per tile", b, q, l, n_tiles, out, cu)) return 1;
    if (run<0, 1, 500>("tile layout, 1 tile ahead, 500 instructions per tile", b, q, l, n_tiles, out, cu)) return 1;
    if (run<0, 1, 700>("tile layout, 1 tile ahead, 700 instructions per tile", b, q, l, n_tiles, out, cu)) return 1;
    if (run<0, 2, 500>("tile layout, 2 tiles ahead, 500 instructions per tile", b, q, l, n_tiles, out, cu)) return 1;
    if (run<0, 2, 700>("tile layout, 2 tiles ahead, 700 instructions per tile", b, q, l, n_tiles, out, cu)) return 1;
    if (run<0, 1, 500, 1>("tile layout, 1 ahead + lines of the tile after touched, 500 instr", b, q, l, n_tiles, out, cu)) return 1;
    if (run<0, 1, 700, 1>("tile layout, 1 ahead + lines of the tile after touched, 700 instr", b, q, l, n_tiles, out, cu)) return 1;
    if (run<0, 1, 500, 2>("tile layout, 1 ahead + lines two tiles after touched, 500 instr", b, q, l, n_tiles, out, cu)) return 1;
    for (int g : {cu}) {
        if (run<0, 1>("tile layout (2 KiB + 5 KiB + 512 B per tile), 1 tile ahead", b, q, l, n_tiles, out, g)) return 1;
        if (run<0, 2>("tile layout, 2 tiles ahead", b, q, l, n_tiles, out, g)) return 1;
        if (run<0, 3>("tile layout, 3 tiles ahead", b, q, l, n_tiles, out, g)) return 1;
        if (run<1, 1>("one 7.5-KiB run per tile, runs 48 KiB apart, 1 tile ahead", b, q, l, n_tiles, out, g)) return 1;
        if (run<1, 2>("one 7.5-KiB run per tile, runs 48 KiB apart, 2 tiles ahead", b, q, l, n_tiles, out, g)) return 1;
        if (run<2, 1>("the same volume back to back (streaming), 1 tile ahead", b, q, l, n_tiles, out, g)) return 1;
        if (run<2, 2>("the same volume back to back (streaming), 2 tiles ahead", b, q, l, n_tiles, out, g)) return 1;
    }
    return 0;
}
