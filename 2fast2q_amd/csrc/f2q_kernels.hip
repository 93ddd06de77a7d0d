// f2q_kernels.hip -- libf2q_hip.so: HIP kernels for gfx950 + the C ABI of include/f2q.h.
//
// Kernels
//   k_count_fixed   fast path, fixed-offset window: one lane per read of a 256-read packed tile,
//                   tile planes laid out [word][lane] so every load is a 1 KiB coalesced row;
//                   Phred mask by SWAR on the quality words, 2-bit key, exact probe of the
//                   library hash, pigeonhole + popcount search for <= m mismatches, counts by
//                   LDS-privatised histogram (global atomics for large libraries).
//   k_count_general any mode, any symbols: one lane per raw-byte record (the reads the packed
//                   layout cannot carry, multi-window keys, anchored search, Extract+Count).
//   k_synth_*       device-side generator of the SURVEY §8(d) workload (bench input).
//   k_ec_rehash     grows the Extract+Count table.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "f2q_device.h"
#include "f2q_host.h"
#include "f2q_synth.h"

using namespace f2q;

// ===============================================================================================
// kernels
// ===============================================================================================
#define F2Q_HIST_MAX 24576u     // features whose u32 histogram fits the workgroup's LDS budget (96 KiB)

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Workgroup-level sum of the 5 reference counters: same-address global atomics serialise in L2
// (20k of them cost ~0.2 ms per launch), so a workgroup issues at most one per counter -- or none
// when it can leave its sums in a slab row for k_reduce_slabs.
__device__ __forceinline__ void flush_stats(const Accum &acc, unsigned long long st[5], unsigned long long *lds8,
                                            unsigned long long *slab_row)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < 8) lds8[threadIdx.x] = 0;
    __syncthreads();
    for (int k = 0; k < 5; k++) {
        unsigned long long v = wave_sum(st[k]);
        if (lane == 0 && v) atomicAdd(&lds8[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        unsigned long long v = lds8[threadIdx.x];
        if (slab_row) gpw(slab_row)[threadIdx.x] = v;
        else if (v) acc_add(&acc.stats[threadIdx.x], v);
    }
}

// Fast path, fixed offset.  Persistent workgroups stride over the tiles.  USE_LDS: per-workgroup
// u32 histogram in LDS, flushed once with 64-bit global atomics.
template <bool USE_LDS>
__global__ __launch_bounds__(F2Q_TILE) void k_count_fixed(const RunDev *__restrict__ runp,
                                                            const LibDev *__restrict__ libp, PackedBlock pb,
                                                            Accum acc)
{
    extern __shared__ uint32_t hist[];
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    const uint32_t nf = lib.n_features;
    if (USE_LDS) {
        for (uint32_t i = threadIdx.x; i < nf; i += F2Q_TILE) hist[i] = 0;
        __syncthreads();
    }
    unsigned long long st1 = 0, st2 = 0, st3 = 0, st4 = 0, st0 = 0;
    for (uint32_t tile = blockIdx.x; tile < pb.n_tiles; tile += gridDim.x) {
        uint32_t idx = 0;
        int res = fixed_lane(run, lib, pb, tile, threadIdx.x, idx);
        if (res == 1 || res == 2) {
            if (USE_LDS) atomicAdd(&hist[idx], 1u);
            else acc_add(&acc.counts[idx], 1ull);
        }
        st0 += (res != 0); st1 += (res == 1); st2 += (res == 2); st3 += (res == 3); st4 += (res == 4);
    }
    __shared__ unsigned long long st_lds[8];
    unsigned long long stv[5] = {st0, st1, st2, st3, st4};
    flush_stats(acc, stv, st_lds, nullptr);
    if (USE_LDS) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nf; i += F2Q_TILE) {
            uint32_t c = hist[i];
            if (c) acc_add(&acc.counts[i], (unsigned long long)c);
        }
    }
}

// ---- fast path v2 -----------------------------------------------------------------------------------
// One wave per 256-read tile, lane l owns reads 4l..4l+3: every tile row is one 16-byte load per lane
// (1 KiB per wave instruction).  Exact probes hit the packed table (key|index in one u64, two slots in
// flight per round, four reads in flight per lane).  Reads whose exact probe misses are not searched in
// place -- that would run the pigeonhole chains with ~15 % of the lanes active -- but pushed into an LDS
// ring; once a workgroup's ring holds a full workgroup of keys they are searched one per lane.
#define F2Q_V2_THREADS 512
#define F2Q_V2_WAVES (F2Q_V2_THREADS / 64)
#define F2Q_V2_QCAP 320u      // entries per wave ring: < 64 left over + <= 256 pushed per tile

template <bool NT>
__device__ __forceinline__ U4 ld_u4(const uint32_t F2Q_GLOBAL *p)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4 F2Q_GLOBAL *q = (const v4 F2Q_GLOBAL *)p;
    v4 v = NT ? __builtin_nontemporal_load(q) : *q;      // NT: stream once, keep L2 for the tables
    return U4{v.x, v.y, v.z, v.w};
}

__device__ __noinline__ int slow_read(const RunDev *run, const LibDev *lib, const PackedBlock *pb, uint32_t tile,
                                      uint32_t slot, uint32_t *idx)
{
    return fixed_lane(*run, *lib, *pb, tile, slot, *idx);
}

// NQ / NB: number of quality / base rows under the window when known at compile time (the common
// geometries get their own instantiation so that all row loads sit in one basic block and issue
// back to back); 0 = run-time geometry, rows beyond the window are clamped re-loads of the last one.
template <bool USE_LDS, int NQ, int NB>
// launch bound 4 waves/SIMD = two 512-thread workgroups per CU
__global__ __launch_bounds__(F2Q_V2_THREADS, 4) void k_count_fixed4(const RunDev *__restrict__ runp,
                                                                  const LibDev *__restrict__ libp, PackedBlock pb,
                                                                  Accum acc)
{
    extern __shared__ unsigned long long smem64[];
    // one ring of keys per wave: pushes and drains are wave-synchronous, so the tile loop has no barrier
    unsigned long long *queue = smem64 + (threadIdx.x >> 6) * F2Q_V2_QCAP;                 // keys
    uint32_t *qforced = reinterpret_cast<uint32_t *>(smem64 + F2Q_V2_WAVES * F2Q_V2_QCAP) + (threadIdx.x >> 6) * F2Q_V2_QCAP;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem64 + F2Q_V2_WAVES * F2Q_V2_QCAP) + F2Q_V2_WAVES * F2Q_V2_QCAP;  // USE_LDS
    // !USE_LDS (library too large for an LDS histogram): the same region holds the read slot of every ring entry
    uint32_t *qslot = hist + (threadIdx.x >> 6) * F2Q_V2_QCAP;

    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    const uint32_t nf = lib.n_features;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (USE_LDS) for (uint32_t i = tid; i < nf; i += F2Q_V2_THREADS) hist[i] = 0;
    __syncthreads();
    uint32_t q_head = 0, q_tail = 0;          // the ring belongs to this wave alone: head and tail live in (uniform) registers
    const FixedGeom g = fixed_geom(run);
    const int need = g.st + g.L;
    const bool do_near = run.miss > 0;
    const PackedPiece ex = lib.pk.exact;
    const uint32_t ib = lib.pk.ib, exm = (1u << ex.bits) - 1u;
    const uint64_t imask = (1ull << ib) - 1ull;
    const auto ptab = gp(lib.ptab);
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;
#ifdef F2Q_STAMP
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, t0_ = 0, t1_;
#define STAMP4(i) do { __builtin_amdgcn_sched_barrier(0); t1_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); tp[i] += t1_ - t0_; t0_ = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP4(i) do {} while (0)
#endif

    // a hit: LDS histogram, or (large library) the feature index is stored at the read's slot for k_hist_ranges
    auto count_hit = [&](uint32_t idx, uint64_t slot) {
        if (USE_LDS) atomicAdd(&hist[idx], 1u);
        else gpw(acc.hit_buf)[slot] = idx;
    };

    for (uint32_t base = blockIdx.x * F2Q_V2_WAVES; base < pb.n_tiles; base += gridDim.x * F2Q_V2_WAVES) {
        const uint32_t tile = base + wave;
#ifdef F2Q_STAMP
        __builtin_amdgcn_sched_barrier(0); t0_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
        int res[4] = {R_SKIP, R_SKIP, R_SKIP, R_SKIP};
        uint64_t key[4] = {0, 0, 0, 0};
        uint32_t forced[4] = {0, 0, 0, 0};
        if (tile < pb.n_tiles) {
            constexpr int QR = NQ ? NQ : F2Q_MAXQROWS, BR = NB ? NB : F2Q_MAXBROWS;
            constexpr bool NT = true;                    // tile rows are streamed once: keep L2 for the tables
            U4 brow[BR], qrow[QR];
            const auto qp = gp(pb.qual) + (uint64_t)tile * pb.wq * F2Q_TILE + 4u * lane;
            const auto bp = gp(pb.bases) + (uint64_t)tile * pb.wb * F2Q_TILE + 4u * lane;
            // every load of the tile is issued before anything is consumed
#pragma unroll
            for (int r = 0; r < BR; r++) {
                uint32_t row = (uint32_t)g.bw0 + (uint32_t)(NB ? r : (r < g.nb ? r : (g.nb > 0 ? g.nb - 1 : 0)));
                row = row < pb.wb ? row : pb.wb - 1u;       // reads shorter than the window: stay inside the tile
                brow[r] = ld_u4<NT>(bp + (uint64_t)row * F2Q_TILE);
            }
            if (NQ || g.add_hi) {                        // --ph <= 1: no quality row is needed at all
#pragma unroll
                for (int r = 0; r < QR; r++) {
                    uint32_t row = (uint32_t)g.qw0 + (uint32_t)(NQ ? r : (r < g.nq ? r : (g.nq > 0 ? g.nq - 1 : 0)));
                    row = row < pb.wq ? row : pb.wq - 1u;
                    qrow[r] = ld_u4<NT>(qp + (uint64_t)row * F2Q_TILE);
                }
            } else {
#pragma unroll
                for (int r = 0; r < QR; r++) qrow[r] = U4{0, 0, 0, 0};
            }
            uint32_t len01 = 0, len23 = 0;
            if (pb.len) {
                typedef uint32_t v2 __attribute__((ext_vector_type(2)));
                v2 lv = *(const v2 F2Q_GLOBAL *)(gp(pb.len) + (uint64_t)tile * F2Q_TILE + 4u * lane);
                len01 = lv.x; len23 = lv.y;
            }
            uint32_t bad[4] = {0, 0, 0, 0};
            if (g.add_hi) {
#pragma unroll
                for (int r = 0; r < QR; r++)
                    if (NQ || r < g.nq) fixed4_qrow(g, r, qrow[r], bad);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t l = pb.len ? (((j < 2 ? len01 : len23) >> (16 * (j & 1))) & 0xFFFFu) : pb.rmax;
                const bool have_q = NQ || g.add_hi;            // the flag bits travel in the quality rows
                if (l == F2Q_LEN_SKIP) res[j] = R_SKIP;
                else if ((int)(l & 0x7FFFu) < need || g.L < 1 || ((l & F2Q_LEN_FLAG) && !have_q)) res[j] = R_SLOW;
                else if (bad[j]) res[j] = R_QFAIL;
                else {
                    res[j] = R_NEAR; key[j] = fixed4_key(g, brow, j);
                    if (l & F2Q_LEN_FLAG) {                    // non-ACGT symbols in the window (rare)
                        forced[j] = fixed4_flags(g, qrow, j);
                        if (forced[j]) res[j] = (!do_near || __popc(forced[j]) > run.miss) ? R_NONALIGNED : R_FORCED;
                    }
                }
            }
            STAMP4(0);                              // row loads, Phred, keys
            // exact probes: up to 4 reads x 2 slots in flight per lane
            uint32_t s[4]; bool pend[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { pend[j] = (res[j] == R_NEAR); s[j] = hash32(key[j], ex.bits); }
            while (pend[0] | pend[1] | pend[2] | pend[3]) {
                uint64_t v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (pend[j]) { v0[j] = ptab[ex.off + s[j]]; v1[j] = ptab[ex.off + ((s[j] + 1u) & exm)]; }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (!pend[j]) continue;
                    if (v0[j] == KEY_EMPTY) pend[j] = false;
                    else if ((v0[j] >> ib) == key[j]) { pend[j] = false; res[j] = R_PERFECT; count_hit((uint32_t)(v0[j] & imask), (uint64_t)tile * F2Q_TILE + 4u * lane + j); }
                    else if (v1[j] == KEY_EMPTY) pend[j] = false;
                    else if ((v1[j] >> ib) == key[j]) { pend[j] = false; res[j] = R_PERFECT; count_hit((uint32_t)(v1[j] & imask), (uint64_t)tile * F2Q_TILE + 4u * lane + j); }
                    else s[j] = (s[j] + 2u) & exm;
                }
            }
            STAMP4(1);                              // exact probes
            uint32_t npush = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (res[j] == R_SLOW) {                    // clipped window / odd geometry: the one-read routine
                    uint32_t idx = 0;
                    int r1 = slow_read(runp, libp, &pb, tile, 4u * lane + (uint32_t)j, &idx);
                    if (r1 == 1 || r1 == 2) count_hit(idx, (uint64_t)tile * F2Q_TILE + 4u * lane + j);
                    res[j] = r1;
                } else if (res[j] == R_NEAR) {
                    if (do_near) npush++; else res[j] = R_NONALIGNED;
                } else if (res[j] == R_FORCED) npush++;
                st0 += (res[j] != R_SKIP); st1 += (res[j] == R_PERFECT); st2 += (res[j] == R_IMPERFECT);
                st3 += (res[j] == R_NONALIGNED); st4 += (res[j] == R_QFAIL);
            }
            {
                // ring slots by a wave prefix sum of npush (0..4) over three ballots -- no LDS atomics
                const unsigned long long b0 = __ballot(npush & 1u), b1 = __ballot(npush & 2u), b2 = __ballot(npush & 4u);
                uint32_t at = q_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u))
                              + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u))
                              + 4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
                q_tail += (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (res[j] == R_NEAR || res[j] == R_FORCED) {
                        queue[at % F2Q_V2_QCAP] = key[j]; qforced[at % F2Q_V2_QCAP] = forced[j];
                        if (!USE_LDS) qslot[at % F2Q_V2_QCAP] = tile * F2Q_TILE + 4u * lane + (uint32_t)j;   // slot inside the block (< 2^32)
                        at++;
                    }
            }
        }
        STAMP4(2);                                  // histogram, slow reads, ring push
        if (do_near) {
            // LDS operations of one wave complete in order; the fence keeps the compiler from moving the reads up
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t tail = q_tail;
            while (tail - q_head >= 64u) {
                uint32_t idx = 0;
                int r = packed_near_decide(run, lib, queue[(q_head + lane) % F2Q_V2_QCAP], qforced[(q_head + lane) % F2Q_V2_QCAP], idx);
                if (r == R_IMPERFECT || r == R_PERFECT) { count_hit(idx, USE_LDS ? 0u : qslot[(q_head + lane) % F2Q_V2_QCAP]); st2++; } else st3++;
                q_head += 64u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        STAMP4(3);                                  // ring drain
    }
    if (do_near) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t tail = q_tail;
        if (lane < tail - q_head) {
            uint32_t idx = 0;
            int r = packed_near_decide(run, lib, queue[(q_head + lane) % F2Q_V2_QCAP], qforced[(q_head + lane) % F2Q_V2_QCAP], idx);
            if (r == R_IMPERFECT || r == R_PERFECT) { count_hit(idx, USE_LDS ? 0u : qslot[(q_head + lane) % F2Q_V2_QCAP]); st2++; } else st3++;
        }
    }
#ifdef F2Q_STAMP
    if (lane == 0 && acc.stamp) for (int i = 0; i < 4; i++) atomicAdd(&acc.stamp[i], tp[i]);
#endif
    __shared__ unsigned long long st_lds[8];
    unsigned long long stv[5] = {st0, st1, st2, st3, st4};
    flush_stats(acc, stv, st_lds, acc.stat_slab ? acc.stat_slab + (uint64_t)blockIdx.x * 8u : nullptr);
    if (USE_LDS) {
        // the workgroup's histogram leaves as one coalesced slab row; k_reduce_slabs sums the rows
        __syncthreads();
        auto row = gpw(acc.slab) + (uint64_t)blockIdx.x * nf;
        for (uint32_t i = tid; i < nf; i += F2Q_V2_THREADS) row[i] = hist[i];
    }
}

// Extract+Count with a fixed window (--mo EC --st/--l): same tile walk and Phred test as k_count_fixed4, but every
// passing window (clipped to the read, possibly empty) is a key of the single-word device table -- no library.
__global__ __launch_bounds__(F2Q_V2_THREADS) void k_extract_fixed4(const RunDev *__restrict__ runp, EcDev ec, PackedBlock pb,
                                                                    Accum acc, uint64_t read_base)
{
    const RunDev &run = *runp;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const FixedGeom g = fixed_geom(run);
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    for (uint32_t base = blockIdx.x * F2Q_V2_WAVES; base < pb.n_tiles; base += gridDim.x * F2Q_V2_WAVES) {
        const uint32_t tile = base + wave;
        if (tile >= pb.n_tiles) continue;
        U4 brow[F2Q_MAXBROWS], qrow[F2Q_MAXQROWS];
        const auto qp = gp(pb.qual) + (uint64_t)tile * pb.wq * F2Q_TILE + 4u * lane;
        const auto bp = gp(pb.bases) + (uint64_t)tile * pb.wb * F2Q_TILE + 4u * lane;
#pragma unroll
        for (int r = 0; r < F2Q_MAXBROWS; r++) {
            uint32_t row = (uint32_t)g.bw0 + (uint32_t)(r < g.nb ? r : (g.nb > 0 ? g.nb - 1 : 0));
            row = row < pb.wb ? row : pb.wb - 1u;
            brow[r] = ld_u4<true>(bp + (uint64_t)row * F2Q_TILE);
        }
#pragma unroll
        for (int r = 0; r < F2Q_MAXQROWS; r++) {
            const uint32_t want = (uint32_t)g.qw0 + (uint32_t)(r < g.nq ? r : (g.nq > 0 ? g.nq - 1 : 0));
            const uint32_t row = want < pb.wq ? want : pb.wq - 1u;
            // a row past the tile's last one holds no byte of any read of the block: it must test as "nothing fails"
            qrow[r] = (g.add_hi && want < pb.wq) ? ld_u4<true>(qp + (uint64_t)row * F2Q_TILE) : U4{0, 0, 0, 0};
        }
        uint32_t len01 = 0, len23 = 0;
        if (pb.len) {
            typedef uint32_t v2 __attribute__((ext_vector_type(2)));
            v2 lv = *(const v2 F2Q_GLOBAL *)(gp(pb.len) + (uint64_t)tile * F2Q_TILE + 4u * lane);
            len01 = lv.x; len23 = lv.y;
        }
        uint32_t bad[4] = {0, 0, 0, 0};
        if (g.add_hi) {
#pragma unroll
            for (int r = 0; r < F2Q_MAXQROWS; r++)
                if (r < g.nq) fixed4_qrow(g, r, qrow[r], bad);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t l = pb.len ? (((j < 2 ? len01 : len23) >> (16 * (j & 1))) & 0xFFFFu) : pb.rmax;
            if (l == F2Q_LEN_SKIP) continue;
            st[0]++;
            // bytes past the end of a short read are stored as 0 and never fail, so bad[] already is the clipped test
            if (bad[j]) { st[4]++; continue; }
            const int rl = (int)(l & 0x7FFFu);
            int L = (rl < g.st + g.L ? rl : g.st + g.L) - g.st;             // Python slice clipping (:354)
            if (L < 0) L = 0;
            const uint64_t key = fixed4_key(g, brow, j) & (L >= 32 ? ~0ull : ((1ull << (2 * L)) - 1ull));
            const uint64_t slot = (uint64_t)tile * F2Q_TILE + 4u * lane + (uint32_t)j;
            ec64_insert(ec, key, L, read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot));
            st[1]++;
        }
    }
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, nullptr);
}

// ---- fast path, anchored ------------------------------------------------------------------------------
// One lane = one read of a planar tile (bit-planes, see f2q_device.h).  Every byte of every read is
// needed here (the anchors can sit anywhere, the Phred tests follow them), so this is the kernel that
// streams the full 188 B/read: 2*NW base words + 8*NW quality words + the length per lane, all as
// coalesced 256-byte rows.  Counter mode: exact probe, misses queued per wave for the pigeonhole
// search, LDS histogram.  Extract+Count mode: single-word insert into the device table.
#define F2Q_AN_THREADS 256
#define F2Q_AN_WAVES (F2Q_AN_THREADS / 64)
#define F2Q_AN_QCAP 512u          // entries per wave ring (power of two: the ring index is a mask, not a multiply)

// odd geometry (negative-index slices, windows the 2-bit tables cannot hold): the byte-exact general routine
// on a private copy of the read, rebuilt from the tile in memory so that the caller keeps nothing live for it
__device__ __noinline__ void anchor_slow(const RunDev *run, const LibDev *lib, const EcDev *ec, const Accum *acc,
                                         const PackedBlock *pb, uint32_t tile, uint32_t slot, int r,
                                         unsigned long long read_index, unsigned long long *st)
{
    uint8_t seq[F2Q_ANCHOR_MAXLEN], qual[F2Q_ANCHOR_MAXLEN];
    const uint32_t nw = pb->planar_nw;
    const auto bp = gp(pb->bases) + (uint64_t)tile * pb->wb * F2Q_TILE + slot;
    const auto qp = gp(pb->qual) + (uint64_t)tile * pb->wq * F2Q_TILE + slot;
    if (r > F2Q_ANCHOR_MAXLEN) r = F2Q_ANCHOR_MAXLEN;
    for (int i = 0; i < r; i++) {
        const uint32_t lo = bp[(uint64_t)(i >> 5) * F2Q_TILE], hi = bp[(uint64_t)(nw + (i >> 5)) * F2Q_TILE];
        seq[i] = (uint8_t)"ACGT"[((lo >> (i & 31)) & 1u) | (((hi >> (i & 31)) & 1u) << 1)];
        qual[i] = (uint8_t)((qp[(uint64_t)(i >> 2) * F2Q_TILE] >> (8 * (i & 3))) & 0xFFu);
        if (qual[i] & 0x80u) { seq[i] = (uint8_t)'N'; qual[i] &= 0x7Fu; }      // flagged: a symbol that equals nothing
    }
    general_read<const uint8_t *>(*run, *lib, *ec, *acc, seq, r, qual, r, read_index, st);
}

// SAMEQ: --qsu == --qsd == --ph (the default), one fail vector serves all three Phred tests
template <int NW, int KB, bool EC, bool USE_LDS, bool SAMEQ>
__global__ __launch_bounds__(F2Q_AN_THREADS) void k_count_anchor(const RunDev *__restrict__ runp,
                                                                  const LibDev *__restrict__ libp, EcDev ec,
                                                                  PackedBlock pb, Accum acc, uint64_t read_base)
{
    constexpr int NQW = 8 * NW;
    extern __shared__ unsigned long long smem64[];
    unsigned long long *queue = smem64 + (threadIdx.x >> 6) * F2Q_AN_QCAP;
    uint32_t *qforced = reinterpret_cast<uint32_t *>(smem64 + F2Q_AN_WAVES * F2Q_AN_QCAP) + (threadIdx.x >> 6) * F2Q_AN_QCAP;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem64 + F2Q_AN_WAVES * F2Q_AN_QCAP) + F2Q_AN_WAVES * F2Q_AN_QCAP;
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    const uint32_t nf = lib.n_features;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (USE_LDS && !EC) for (uint32_t i = tid; i < nf; i += F2Q_AN_THREADS) hist[i] = 0;
    __syncthreads();
    uint32_t q_head = 0, q_tail = 0;          // this wave's ring: head and tail in (uniform) registers
    const bool do_near = run.miss > 0;
    const int pk_len = (int)lib.pk.len;
    const uint32_t ah_w = phred_add_hi(run.thr), ah_u = phred_add_hi(run.thr_up), ah_d = phred_add_hi(run.thr_down);
    unsigned long long st[5] = {0, 0, 0, 0, 0};
#ifdef F2Q_STAMP
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, t0_, t1_;
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); t1_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); tp[i] += t1_ - t0_; t0_ = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

    auto count_hit = [&](uint32_t idx) {
        if (USE_LDS) atomicAdd(&hist[idx], 1u);
        else acc_add(&acc.counts[idx], 1ull);
    };

    // Software pipeline: the rows of the workgroup's NEXT tile are requested before the current tile is processed, so
    // a wave's HBM latency hides under its own anchor search.  The kernel sits at 2 waves/SIMD anyway (LDS: histogram +
    // rings, two workgroups per CU), so the 10*NW + 1 extra registers cost no occupancy.
    uint32_t nLO[NW], nHI[NW], nQ[NQW], nl = F2Q_LEN_SKIP;
    auto request_tile = [&](uint32_t t) {
        const auto bpn = gp(pb.bases) + (uint64_t)t * pb.wb * F2Q_TILE + tid;
        const auto qpn = gp(pb.qual) + (uint64_t)t * pb.wq * F2Q_TILE + tid;
        nl = pb.len ? gp(pb.len)[(uint64_t)t * F2Q_TILE + tid] : pb.rmax;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            nLO[w] = __builtin_nontemporal_load(bpn + (uint64_t)w * F2Q_TILE);
            nHI[w] = __builtin_nontemporal_load(bpn + (uint64_t)(NW + w) * F2Q_TILE);
        }
#pragma unroll
        for (int i = 0; i < NQW; i++) nQ[i] = __builtin_nontemporal_load(qpn + (uint64_t)i * F2Q_TILE);   // planar tiles always hold 8*NW rows
    };
    if (blockIdx.x < pb.n_tiles) request_tile(blockIdx.x);
    for (uint32_t tile = blockIdx.x; tile < pb.n_tiles; tile += gridDim.x) {
#ifdef F2Q_STAMP
        __builtin_amdgcn_sched_barrier(0); t0_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
        const uint32_t l = nl;
        uint32_t LO[NW], HI[NW], Q[NQW];
#pragma unroll
        for (int w = 0; w < NW; w++) { LO[w] = nLO[w]; HI[w] = nHI[w]; }
#pragma unroll
        for (int i = 0; i < NQW; i++) Q[i] = nQ[i];
        __builtin_amdgcn_sched_barrier(0);
        if (tile + gridDim.x < pb.n_tiles) request_tile(tile + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        // quality words -> per-base fail vectors.  All 8*NW loads are issued together (one memory round trip);
        // the scheduling barrier keeps the compiler from stretching their live ranges into the anchor search.
        uint32_t FW[NW], FU[SAMEQ ? 1 : NW], FD[SAMEQ ? 1 : NW], FLG[NW];
        const bool flagged = (l != F2Q_LEN_SKIP) && (l & F2Q_LEN_FLAG);
        {
#pragma unroll
            for (int cw = 0; cw < NW; cw++) {
                uint32_t q8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q8[i] = Q[8 * cw + i];
                FW[cw] = fail_word8(q8, ah_w);
                if (!SAMEQ) { FU[cw] = fail_word8(q8, ah_u); FD[cw] = fail_word8(q8, ah_d); }
                FLG[cw] = flagged ? flag_word8(q8) : 0u;          // non-ACGT symbols (rare reads)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(0);                                   // loads + fail vectors
        bool push = false; uint64_t push_key = 0; uint32_t push_forced = 0;
        if (l != F2Q_LEN_SKIP) {
            const int r = (int)(l & 0x7FFFu);
            const uint64_t slot = (uint64_t)tile * F2Q_TILE + tid;
            const unsigned long long gi = read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot);
            AnchorWin aw;
            if constexpr (SAMEQ) aw = anchor_window<NW, KB, KB>(run, LO, HI, FLG, r, FW, FW, FW);
            else aw = anchor_window<NW, KB, KB>(run, LO, HI, FLG, r, FU, FD, FW);
            STAMP(1);                               // anchor search + region tests
            const int L = aw.end - aw.start;
            if (aw.ok == 0) { st[4]++; st[0]++; }
            else if (aw.ok == 1 && !EC && (L < 1 || L > F2Q_REG_MAXLEN)) {
                // Counter mode, all-ACGT library of <= 31-base features (the packed path's precondition): an empty
                // or longer window passed its Phred test but can equal or approach no feature (:683) -> not aligned
                st[3]++; st[0]++;
            } else if (aw.ok == 1 && EC && L > F2Q_EC64_MAXLEN) {
                // Extract+Count key too long for the single-word table: decode the window and use the byte-string table
                uint8_t kb[32 * NW];
#pragma unroll
                for (int cw = 0; cw < NW; cw++) {
                    const int off = 32 * cw, n = L - off < 32 ? L - off : 32;
                    if (n > 0) {
                        const uint32_t lo = plane_extract<NW>(LO, aw.start + off, n), hi = plane_extract<NW>(HI, aw.start + off, n);
                        for (int j = 0; j < n; j++) kb[off + j] = (uint8_t)"ACGT"[((lo >> j) & 1u) | (((hi >> j) & 1u) << 1)];
                    }
                }
                KeyView kv; kv.seq = kb; kv.nseg = 1; kv.a[0] = 0; kv.b[0] = L; kv.len = L;
                ec_insert(ec, kv, gi);
                st[1]++; st[0]++;
            } else if (aw.ok == 2) {
                // negative-index slices (down-only anchor near the read start, negative --l): byte-exact routine
                unsigned long long st2[5] = {0, 0, 0, 0, 0};
                const EcDev ec2 = ec; const Accum acc2 = acc; const PackedBlock pb2 = pb;
                anchor_slow(runp, libp, &ec2, &acc2, &pb2, tile, tid, r, gi, st2);
#pragma unroll
                for (int k = 0; k < 5; k++) st[k] += st2[k];
            } else if (flagged && plane_extract<NW>(FLG, aw.start, L) != 0u) {
                // the window itself holds non-ACGT symbols
                st[0]++;
                const uint32_t forced = plane_extract<NW>(FLG, aw.start, L);
                if (EC) {
                    // unreachable: Extract+Count keys hold the symbol itself, so the packer never flags reads in EC runs
                    st[3]++;
                } else if (!do_near || __popc(forced) > run.miss) st[3]++;
                else {
                    const uint64_t key = plane_key<NW>(LO, HI, aw.start, L);
                    if (L == pk_len) { push = true; push_key = key; push_forced = forced; }   // queued with its forced mask
                    else {
                        MinTrack t; t.init(run.miss);
                        lib_near(lib, key, L, spread32(forced), t);       // wide tables, in place (rare)
                        if (t.cnt == 1) { count_hit(t.idx); st[2]++; } else st[3]++;
                    }
                }
            } else {
                const uint64_t key = plane_key<NW>(LO, HI, aw.start, L);
                st[0]++;
                if (EC) { ec64_insert(ec, key, L, gi); st[1]++; }
                else if (L == pk_len) {
                    const int e = packed_exact(lib, key);
                    if (e >= 0) { count_hit((uint32_t)e); st[1]++; }
                    else if (!do_near) st[3]++;
                    else { push = true; push_key = key; push_forced = 0u; }
                } else {
                    // a window of another length than the packed tables index: wide tables, in place (rare)
                    const int e = lib_exact(lib, key, L);
                    if (e >= 0) { count_hit((uint32_t)e); st[1]++; }
                    else {
                        MinTrack t; t.init(run.miss);
                        if (do_near) lib_near(lib, key, L, 0ull, t);
                        if (t.cnt == 1) { count_hit(t.idx); st[2]++; } else st[3]++;
                    }
                }
            }
        }
        if (!EC && do_near) {
            // ring slot by ballot prefix (no LDS atomic)
            const unsigned long long pm = __ballot(push);
            if (push) {
                const uint32_t at = q_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                queue[at % F2Q_AN_QCAP] = push_key; qforced[at % F2Q_AN_QCAP] = push_forced;
            }
            q_tail += (uint32_t)__popcll(pm);
        }
        STAMP(2);                                   // key, probe, insert
        if (!EC && do_near) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t tail = q_tail;
            while (tail - q_head >= 64u) {
                uint32_t idx = 0;
                int rr = packed_near_decide(run, lib, queue[(q_head + lane) % F2Q_AN_QCAP], qforced[(q_head + lane) % F2Q_AN_QCAP], idx);
                if (rr == R_IMPERFECT || rr == R_PERFECT) { count_hit(idx); st[2]++; } else st[3]++;
                q_head += 64u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        STAMP(3);                                   // ring drain
    }
    if (!EC && do_near) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t tail = q_tail;
        if (lane < tail - q_head) {
            uint32_t idx = 0;
            int rr = packed_near_decide(run, lib, queue[(q_head + lane) % F2Q_AN_QCAP], qforced[(q_head + lane) % F2Q_AN_QCAP], idx);
            if (rr == R_IMPERFECT || rr == R_PERFECT) { count_hit(idx); st[2]++; } else st[3]++;
        }
    }
#ifdef F2Q_STAMP
    if (lane == 0 && acc.stamp) for (int i = 0; i < 4; i++) atomicAdd(&acc.stamp[i], tp[i]);
#endif
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, acc.stat_slab ? acc.stat_slab + (uint64_t)blockIdx.x * 8u : nullptr);
    if (USE_LDS && !EC) {
        __syncthreads();
        auto row = gpw(acc.slab) + (uint64_t)blockIdx.x * nf;
        for (uint32_t i = tid; i < nf; i += F2Q_AN_THREADS) row[i] = hist[i];
    }
}

// Large libraries (no per-workgroup LDS histogram of the whole library): the counting kernel leaves the feature index
// of every read in hit_buf; here workgroup (range, part) histograms the indices of its part that fall into its range
// of F2Q_HIST_MAX features in LDS and writes that stretch of slab row `part`.  hit_buf is read n_ranges times, from
// the Infinity Cache when it fits (4 B per read).
__global__ __launch_bounds__(1024) void k_hist_ranges(const uint32_t *__restrict__ hit_buf, uint64_t n_slots, uint32_t nf,
                                                       uint32_t n_parts, uint32_t *__restrict__ slab)
{
    extern __shared__ uint32_t rh[];
    const uint32_t range = blockIdx.x / n_parts, part = blockIdx.x % n_parts;
    const uint32_t f0 = range * F2Q_HIST_MAX, fn = (nf - f0) < F2Q_HIST_MAX ? (nf - f0) : F2Q_HIST_MAX;
    for (uint32_t i = threadIdx.x; i < fn; i += 1024u) rh[i] = 0;
    __syncthreads();
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const uint64_t n4 = n_slots / 4;                       // n_slots is a multiple of the tile size
    const auto hb = (const v4 F2Q_GLOBAL *)hit_buf;
    for (uint64_t i = (uint64_t)part * 1024u + threadIdx.x; i < n4; i += (uint64_t)n_parts * 1024u) {
        const v4 v = hb[i];
        const uint32_t e[4] = {v.x - f0, v.y - f0, v.z - f0, v.w - f0};
#pragma unroll
        for (int k = 0; k < 4; k++) if (e[k] < fn) atomicAdd(&rh[e[k]], 1u);   // 0xFFFFFFFF - f0 is never < fn
    }
    __syncthreads();
    auto row = (uint32_t F2Q_GLOBAL *)slab + (uint64_t)part * nf + f0;
    for (uint32_t i = threadIdx.x; i < fn; i += 1024u) row[i] = rh[i];
}

// counts[f] += sum over workgroups of slab[w][f].  Block = 64 features x 4 row lanes; grid.y splits
// the rows further so that every thread has ~16 independent loads in flight.
#define F2Q_RED_SPLIT 8u
__global__ __launch_bounds__(256) void k_reduce_slabs(const uint32_t *__restrict__ slab, uint32_t n_rows, uint32_t nf,
                                                       unsigned long long *__restrict__ counts,
                                                       const unsigned long long *__restrict__ stat_slab, uint32_t n_stat_rows,
                                                       unsigned long long *__restrict__ stats)
{
    __shared__ unsigned long long part[256];
    if (blockIdx.x == 0 && blockIdx.y == 0 && stat_slab) {          // the 5 reference counters: rows of 8
        const uint32_t k = threadIdx.x & 7u, sub = threadIdx.x >> 3;   // 32 row lanes x 8 columns
        unsigned long long sv = 0;
        if (k < 5) for (uint32_t w = sub; w < n_stat_rows; w += 32u) sv += stat_slab[(uint64_t)w * 8u + k];
        part[threadIdx.x] = sv;
        __syncthreads();
        if (threadIdx.x < 5) {
            unsigned long long tot = 0;
            for (uint32_t q = 0; q < 32u; q++) tot += part[q * 8u + threadIdx.x];
            if (tot) atomicAdd(&stats[threadIdx.x], tot);
        }
        __syncthreads();
    }
    const uint32_t fx = threadIdx.x & 63u, ry = threadIdx.x >> 6;
    const uint32_t f = blockIdx.x * 64u + fx;
    unsigned long long sum = 0;
    if (f < nf) {
        const uint32_t step = F2Q_RED_SPLIT * 4u;
#pragma unroll 8
        for (uint32_t w = blockIdx.y * 4u + ry; w < n_rows; w += step) sum += slab[(uint64_t)w * nf + f];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (ry == 0 && f < nf) {
        sum = part[fx] + part[64 + fx] + part[128 + fx] + part[192 + fx];
        if (sum) atomicAdd(&counts[f], sum);
    }
}

struct RawBlock {
    uint64_t n;
    uint64_t first_index;                  // global index of the block's read 0
    const uint8_t *raw;                    // record bytes (host packer: seq then quality; device packer: the FASTQ text itself)
    const unsigned long long *off;         // offset of the sequence line
    const unsigned long long *qoff;        // offset of the quality line, or nullptr: it follows the sequence
    const uint32_t *len, *qlen, *index;    // index: position inside the block (nullptr: == record id)
};

__global__ __launch_bounds__(256) void k_count_general(const RunDev *__restrict__ runp,
                                                        const LibDev *__restrict__ libp, EcDev ec, RawBlock rb,
                                                        Accum acc)
{
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rb.n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        gbytes seq = gp(rb.raw) + gp(rb.off)[i];
        const int r = (int)gp(rb.len)[i], qn = (int)gp(rb.qlen)[i];
        gbytes qual = rb.qoff ? gp(rb.raw) + gp(rb.qoff)[i] : seq + r;
        const unsigned long long gi = rb.first_index + (rb.index ? gp(rb.index)[i] : i);
        general_read(run, lib, ec, acc, seq, r, qual, qn, gi, st);
    }
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, nullptr);
}

// re-insert every entry of `old` into `nw` (table growth)
__global__ void k_ec_rehash(EcDev old, unsigned long long n_old, EcDev nw)
{
    unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_old) return;
    const uint32_t len = old.ent_len[e];
    const uint32_t *src = old.arena + old.ent_off[e];
    const int nwords = (int)((len + 3) >> 2);
    // same hash as key_hash() over the stored bytes
    uint64_t h = 1469598103934665603ull ^ (uint64_t)len;
    for (uint32_t k = 0; k < len; k++) { h ^= (src[k >> 2] >> (8 * (k & 3))) & 0xFFu; h *= 1099511628211ull; }
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const unsigned long long fp = (h >> 32) & 0xFFFFFFFFull;
    // keys are distinct, so plain claim-by-CAS of an empty slot is enough
    unsigned long long ne = atomicAdd(&nw.ctr[0], 1ull);
    unsigned long long off = atomicAdd(&nw.ctr[1], (unsigned long long)nwords);
    for (int w = 0; w < nwords; w++) nw.arena[off + w] = src[w];
    nw.ent_off[ne] = off; nw.ent_len[ne] = len;
    nw.ent_count[ne] = old.ent_count[e]; nw.ent_first[ne] = old.ent_first[e];
    uint32_t s = (uint32_t)h & nw.mask;
    for (;;) {
        unsigned long long prev = atomicCAS(&nw.slots[s], 0ull, (fp << 32) | (ne + 1ull));
        if (prev == 0ull) break;
        s = (s + 1) & nw.mask;
    }
}

// move every key of the old single-word table into the new one (keys are distinct)
__global__ void k_ec64_rehash(EcDev old, EcDev nw)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > old.k64_mask) return;
    const unsigned long long k = old.k64_slots[i];
    if (k == KEY_EMPTY) return;
    uint32_t s = hash32(k ^ (k >> 29), 32) & nw.k64_mask;
    for (;;) {
        unsigned long long prev = atomicCAS(&nw.k64_slots[s], KEY_EMPTY, k);
        if (prev == KEY_EMPTY) break;
        s = (s + 1) & nw.k64_mask;
    }
    nw.k64_count[s] = old.k64_count[i]; nw.k64_first[s] = old.k64_first[i];
    atomicAdd(&nw.ctr[3], 1ull);
}

// ---- synthetic workload, device side ----------------------------------------------------------
struct SynthOut {
    // packed planes (may be null when everything goes to the general path)
    uint32_t *bases, *qual; uint16_t *len; uint32_t wb, wq, planar_nw;
    // general records: fixed stride R for seq and R for quality
    uint8_t *raw; unsigned long long *off; uint32_t *glen, *gqlen, *gindex;
    unsigned long long *g_count; unsigned long long g_cap;
    int all_general;           // 1: every read is written as a raw record
    int inband_n;              // 1: an 'N' inside the window is flagged in place instead of taking the general path
};

__global__ __launch_bounds__(F2Q_TILE) void k_synth(SynthDev s, const uint64_t *__restrict__ guide_keys, SynthOut o,
                                                     uint64_t n_slots)
{
    const uint64_t slot = (uint64_t)blockIdx.x * F2Q_TILE + threadIdx.x;
    if (slot >= n_slots) return;
    const uint64_t tile = slot / F2Q_TILE, lane = slot % F2Q_TILE;
    if (slot >= s.n_reads) { if (o.len) o.len[slot] = (uint16_t)F2Q_LEN_SKIP; return; }
    const uint64_t i = s.first_read + slot;
    SynthRead r = synth_plan(s, i, [&](uint32_t g) { return guide_keys[g]; });
    const int R = s.read_len;
    // does the read hold a symbol the packed planes cannot carry?  (only 'N' is ever generated)
    bool dirty = o.all_general != 0;
    const int npos = (r.n_pos >= 0 && r.wstart + r.n_pos < R) ? r.wstart + r.n_pos : -1;
    if (!dirty && npos >= 0 && !o.inband_n) dirty = true;
    if (dirty) {
        if (o.len) o.len[slot] = (uint16_t)F2Q_LEN_SKIP;
        unsigned long long g = atomicAdd(o.g_count, 1ull);
        if (g >= o.g_cap) return;                      // host checks g_count against g_cap afterwards
        uint8_t *dst = o.raw + g * (unsigned long long)(2 * R);
        uint64_t fw = 0;
        for (int p = 0; p < R; p++) {
            if ((p & 31) == 0) fw = rnd(s.seed, i, F_FLANK0 + (p >> 5));
            dst[p] = synth_base(s, r, p, fw);
            dst[R + p] = (p == r.qpos) ? r.qchar : (uint8_t)'I';
        }
        o.off[g] = g * (unsigned long long)(2 * R);
        o.glen[g] = (uint32_t)R; o.gqlen[g] = (uint32_t)R; o.gindex[g] = (uint32_t)slot;
        // the packed slot stays zero-filled and is skipped through the len plane
        return;
    }
    if (o.len) o.len[slot] = (uint16_t)((uint32_t)R | (npos >= 0 ? F2Q_LEN_FLAG : 0u));
    uint32_t *bp = o.bases + (tile * o.wb) * F2Q_TILE + lane;
    uint32_t *qp = o.qual + (tile * o.wq) * F2Q_TILE + lane;
    uint64_t fw = 0;
    uint32_t bw = 0, qw = 0, lw = 0, hw = 0;
    for (int p = 0; p < R; p++) {
        if ((p & 31) == 0) fw = rnd(s.seed, i, F_FLANK0 + (p >> 5));
        uint8_t c = synth_base(s, r, p, fw);
        uint32_t code = base_code(c); if (code > 3u) code = 0;
        qw |= ((uint32_t)((p == r.qpos) ? r.qchar : (uint8_t)'I') | (p == npos ? 0x80u : 0u)) << (8 * (p & 3));
        if ((p & 3) == 3 || p == R - 1) { qp[(uint64_t)(p >> 2) * F2Q_TILE] = qw; qw = 0; }
        if (o.planar_nw) {                       // anchored runs: bit-planes, 32 bases per word
            lw |= (code & 1u) << (p & 31); hw |= (code >> 1) << (p & 31);
            if ((p & 31) == 31 || p == R - 1) {
                bp[(uint64_t)(p >> 5) * F2Q_TILE] = lw; bp[(uint64_t)(o.planar_nw + (p >> 5)) * F2Q_TILE] = hw;
                lw = 0; hw = 0;
            }
        } else {
            bw |= code << (2 * (p & 15));
            if ((p & 15) == 15 || p == R - 1) { bp[(uint64_t)(p >> 4) * F2Q_TILE] = bw; bw = 0; }
        }
    }
}

// ---- device-side ingest: FASTQ text -> record table -> tiles ------------------------------------------
// The host only moves the text to the device.  k_nl_count / k_line_starts find every line start (two passes
// around a device-wide prefix sum), k_classify applies fastq_parser's framing (4 rstrip()-ed lines per record,
// fast2q.py:324-328) and decides per read whether the tile planes can carry it, k_pack lays clean reads
// into tiles and lists the others as raw records that point into the text itself.
#define F2Q_NL_CHUNK 4096u          // bytes per workgroup: 256 threads x 16 bytes

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t F2Q_GLOBAL *text, uint64_t pos, uint64_t nbytes)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 v = *(const v4 F2Q_GLOBAL *)(text + pos);                  // the buffer is padded to a chunk multiple
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (((w[k] >> (8 * b)) & 0xFFu) == (uint32_t)'\n' && pos + (uint64_t)(4 * k + b) < nbytes) m |= 1u << (4 * k + b);
    return m;
}

__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *text, uint64_t nbytes, uint32_t *chunk_counts)
{
    typedef hipcub::BlockReduce<uint32_t, 256> BR;
    __shared__ typename BR::TempStorage tmp;
    const uint64_t pos = (uint64_t)blockIdx.x * F2Q_NL_CHUNK + threadIdx.x * 16u;
    const uint32_t c = __popc(nl_mask16(gp(text), pos, nbytes));
    const uint32_t tot = BR(tmp).Sum(c);
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = tot;
}

// line_start[k] = offset of line k; line_start[n_newlines + 1] = nbytes + 1 (end sentinel for an unterminated last line)
__global__ __launch_bounds__(256) void k_line_starts(const uint8_t *text, uint64_t nbytes, const uint32_t *chunk_prefix,
                                                      uint32_t *line_start)
{
    typedef hipcub::BlockScan<uint32_t, 256> BS;
    __shared__ typename BS::TempStorage tmp;
    const uint64_t pos = (uint64_t)blockIdx.x * F2Q_NL_CHUNK + threadIdx.x * 16u;
    uint32_t m = nl_mask16(gp(text), pos, nbytes);
    uint32_t before = 0;
    BS(tmp).ExclusiveSum((uint32_t)__popc(m), before);
    uint32_t k = chunk_prefix[blockIdx.x] + before + 1u;
    while (m) { const uint32_t b = (uint32_t)__ffs((int)m) - 1u; m &= m - 1u; gpw(line_start)[k++] = (uint32_t)(pos + b + 1u); }
    if (blockIdx.x == 0 && threadIdx.x == 0) gpw(line_start)[0] = 0u;
}

struct IngestDev {
    const uint8_t *text; const uint32_t *line_start; uint32_t n_records;
    uint32_t *r_off, *r_len, *r_qoff, *r_qlen;   // per record: sequence / quality line (offset, rstrip()-ed length)
    uint32_t *clean;                             // per record: 1 = goes into the tiles
    uint32_t *meta;                              // [0] longest packed length among clean reads
};

__device__ __forceinline__ uint32_t rstrip_dev(gbytes p, uint32_t n)
{
    while (n > 0) {
        const uint8_t c = p[n - 1];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c) n--; else break;
    }
    return n;
}

__global__ __launch_bounds__(256) void k_classify(IngestDev d, PackPlan pl)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= d.n_records) return;
    const auto ls = gp(d.line_start);
    const uint32_t s0 = ls[4u * r + 1u], e0 = ls[4u * r + 2u] - 1u, s1 = ls[4u * r + 3u], e1 = ls[4u * r + 4u] - 1u;
    RecT<gbytes> rec;
    rec.seq = gp(d.text) + s0; rec.qual = gp(d.text) + s1;
    rec.len = rstrip_dev(rec.seq, e0 - s0); rec.qlen = rstrip_dev(rec.qual, e1 - s1);
    gpw(d.r_off)[r] = s0; gpw(d.r_len)[r] = rec.len; gpw(d.r_qoff)[r] = s1; gpw(d.r_qlen)[r] = rec.qlen;
    const bool clean = read_is_clean(pl, rec);
    gpw(d.clean)[r] = clean ? 1u : 0u;
    if (clean) atomicMax(&d.meta[0], packed_len(pl, rec));
}

struct DevSink {
    uint32_t F2Q_GLOBAL *bp; uint32_t F2Q_GLOBAL *qp; uint16_t F2Q_GLOBAL *lp;
    __device__ void base(uint32_t w, uint32_t v) { bp[(uint64_t)w * F2Q_TILE] = v; }
    __device__ void qual(uint32_t w, uint32_t v) { qp[(uint64_t)w * F2Q_TILE] = v; }
    __device__ void len(uint32_t v) { *lp = (uint16_t)v; }
};

struct PackOut {
    uint32_t *bases, *qual; uint16_t *len; uint32_t *c_index; uint32_t wb, wq, planar_nw;
    unsigned long long *g_off, *g_qoff; uint32_t *g_len, *g_qlen, *g_index;
};

// clean_before = exclusive prefix sum of IngestDev::clean
__global__ __launch_bounds__(256) void k_pack(IngestDev d, PackPlan pl, const uint32_t *clean_before, PackOut o)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= d.n_records) return;
    const uint32_t slot = gp(clean_before)[r];
    RecT<gbytes> rec;
    rec.seq = gp(d.text) + gp(d.r_off)[r]; rec.qual = gp(d.text) + gp(d.r_qoff)[r];
    rec.len = gp(d.r_len)[r]; rec.qlen = gp(d.r_qlen)[r];
    if (gp(d.clean)[r]) {
        const uint64_t tile = slot / F2Q_TILE, lane = slot % F2Q_TILE;
        DevSink sink{gpw(o.bases) + tile * o.wb * F2Q_TILE + lane, gpw(o.qual) + tile * o.wq * F2Q_TILE + lane,
                     gpw(o.len) + tile * F2Q_TILE + lane};
        pack_read(pl, rec, o.planar_nw, sink);
        if (o.c_index) gpw(o.c_index)[slot] = r;
    } else {
        const uint32_t g = r - slot;
        gpw(o.g_off)[g] = gp(d.r_off)[r]; gpw(o.g_qoff)[g] = gp(d.r_qoff)[r];
        gpw(o.g_len)[g] = rec.len; gpw(o.g_qlen)[g] = rec.qlen; gpw(o.g_index)[g] = r;
    }
}

// ===============================================================================================
// host side
// ===============================================================================================
struct DevBuf {
    void *p = nullptr; size_t n = 0;
};

struct f2q_block {
    PackedBlock pb{};
    RawBlock rb{};
    std::vector<void *> allocs;
    uint64_t n_reads = 0, n_general = 0, dev_bytes = 0;
};

struct f2q_ctx {
    f2q_params prm{};
    std::vector<std::string> up_s, down_s;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
    RunDev run_h{};
    RunDev *run_d = nullptr;
    PackPlan plan{};
    // library
    bool have_lib = false;
    HostIndex ix;
    LibDev lib_h{};
    LibDev *lib_d = nullptr;
    std::vector<void *> lib_allocs;
    uint64_t *guide_keys_d = nullptr;
    std::vector<uint64_t> synth_keys;    // generator guides set by f2q_synth_guides (else the library's)
    uint64_t *synth_keys_d = nullptr;
    uint32_t synth_glen = 0;
    // accumulators: counts[n_features] then stats[5]
    unsigned long long *acc_d = nullptr;
    uint64_t acc_n = 0;
    uint32_t *slab_d = nullptr;          // per-workgroup histogram rows of the v2 kernel
    size_t slab_n = 0;
    unsigned long long *stat_slab_d = nullptr;
    size_t stat_slab_n = 0;
    uint32_t *hit_buf_d = nullptr;       // large libraries: feature index per read slot of the block being counted
    size_t hit_buf_n = 0;
    // Extract+Count table
    EcDev ec{};
    std::vector<void *> ec_allocs;
    uint64_t ec_slots = 0;
    uint64_t reads_seen = 0;             // global read index of the next block's read 0
    int n_cu = 256;
    bool force_generic = false;           // F2Q_GENERIC=1: run-time window geometry even where a specialisation exists
    bool host_pack = false;               // F2Q_HOST_PACK=1: frame/classify/pack on the host (the round-1 first path; A/B runs)
    bool force_general = false;           // F2Q_FORCE_GENERAL=1: every read through the byte-exact general kernel (cross-checks)
    bool force_v1 = false;                // F2Q_FORCE_V1=1: keep the one-read-per-lane kernel (A/B runs)
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(f2q_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}

#define HIPC(ctx, call)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, F2Q_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

template <class T>
static int dev_upload(f2q_ctx *c, const T *src, size_t n, T **dst, std::vector<void *> &owner)
{
    void *p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIPC(c, hipMalloc(&p, bytes));
    owner.push_back(p);
    if (n) HIPC(c, hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dst = (T *)p;
    return F2Q_OK;
}
template <class T>
static int dev_alloc(f2q_ctx *c, size_t n, T **dst, std::vector<void *> &owner, int fill = -1)
{
    void *p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIPC(c, hipMalloc(&p, bytes));
    owner.push_back(p);
    if (fill >= 0) HIPC(c, hipMemsetAsync(p, fill, bytes, c->stream));
    *dst = (T *)p;
    return F2Q_OK;
}
static void free_all(std::vector<void *> &v)
{
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

extern "C" int f2q_version(void) { return F2Q_ABI_VERSION; }

extern "C" const char *f2q_last_error(const f2q_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" void *f2q_stream(f2q_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

static int setup_run(f2q_ctx *c)
{
    f2q_params p = c->prm;
    for (int i = 0; i < p.n_upstream && i < F2Q_MAX_ITER; i++) p.upstream[i] = c->up_s[i].c_str();
    for (int i = 0; i < p.n_downstream && i < F2Q_MAX_ITER; i++) p.downstream[i] = c->down_s[i].c_str();
    std::string err;
    int rc = fill_run(p, c->run_h, err);
    if (rc) return fail(c, rc, err);
    c->plan = make_plan(c->run_h);
    if (c->force_general) { c->plan.fast_fixed = false; c->plan.fast_anchor = false; }
    return F2Q_OK;
}

static int upload_lib(f2q_ctx *c)
{
    free_all(c->lib_allocs);
    LibDev &L = c->lib_h;
    memset(&L, 0, sizeof L);
    L.n_features = c->ix.n_features;
    L.n_irregular = c->ix.n_irregular;
    memcpy(L.grp, c->ix.grp, sizeof L.grp);
    L.pk = c->ix.pk;
    uint64_t *tk; uint32_t *ti; uint8_t *fb; uint32_t *fo; uint32_t *ir; uint64_t *gk; uint64_t *pt;
    int rc;
    if ((rc = dev_upload(c, c->ix.tab_keys.data(), c->ix.tab_keys.size(), &tk, c->lib_allocs))) return rc;
    if ((rc = dev_upload(c, c->ix.tab_idx.data(), c->ix.tab_idx.size(), &ti, c->lib_allocs))) return rc;
    if ((rc = dev_upload(c, c->ix.feat_bytes.data(), c->ix.feat_bytes.size(), &fb, c->lib_allocs))) return rc;
    if ((rc = dev_upload(c, c->ix.feat_off.data(), c->ix.feat_off.size(), &fo, c->lib_allocs))) return rc;
    if ((rc = dev_upload(c, c->ix.irr_ids.data(), c->ix.irr_ids.size(), &ir, c->lib_allocs))) return rc;
    if ((rc = dev_upload(c, c->ix.key2.data(), c->ix.key2.size(), &gk, c->lib_allocs))) return rc;
    if ((rc = dev_upload(c, c->ix.ptab.data(), c->ix.ptab.size(), &pt, c->lib_allocs))) return rc;
    L.ptab = pt;
    L.tab_keys = tk; L.tab_idx = ti; L.feat_bytes = fb; L.feat_off = fo; L.irr_ids = ir;
    c->guide_keys_d = gk;
    LibDev *ld;
    if ((rc = dev_upload(c, &L, 1, &ld, c->lib_allocs))) return rc;
    c->lib_d = ld;
    HIPC(c, hipStreamSynchronize(c->stream));
    return F2Q_OK;
}

static int alloc_acc(f2q_ctx *c, uint64_t n_features)
{
    if (c->acc_d) { (void)hipFree(c->acc_d); c->acc_d = nullptr; }
    c->acc_n = n_features + 5;
    HIPC(c, hipMalloc((void **)&c->acc_d, c->acc_n * sizeof(unsigned long long)));
    HIPC(c, hipMemsetAsync(c->acc_d, 0, c->acc_n * sizeof(unsigned long long), c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return F2Q_OK;
}

extern "C" int f2q_create(const f2q_params *p, f2q_ctx **out)
{
    if (!p || !out) return fail(nullptr, F2Q_EINVAL, "null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, F2Q_ENODEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (p->device < 0 || p->device >= ndev) return fail(nullptr, F2Q_EINVAL, "device ordinal out of range");
    if (p->mode != 0 && p->mode != 1) return fail(nullptr, F2Q_EINVAL, "mode must be 0 (C) or 1 (EC)");
    f2q_ctx *c = new f2q_ctx();
    c->prm = *p;
    for (int i = 0; i < p->n_upstream && i < F2Q_MAX_ITER; i++) c->up_s.push_back(p->upstream[i] ? p->upstream[i] : "");
    for (int i = 0; i < p->n_downstream && i < F2Q_MAX_ITER; i++) c->down_s.push_back(p->downstream[i] ? p->downstream[i] : "");
    c->device = p->device;
    { const char *fv = getenv("F2Q_FORCE_V1"); c->force_v1 = fv && fv[0] == '1'; }
    { const char *fv = getenv("F2Q_GENERIC"); c->force_generic = fv && fv[0] == '1'; }
    { const char *fv = getenv("F2Q_HOST_PACK"); c->host_pack = fv && fv[0] == '1'; }
    { const char *fv = getenv("F2Q_FORCE_GENERAL"); c->force_general = fv && fv[0] == '1'; }
    int rc = setup_run(c);
    if (rc) { g_create_err = c->err; delete c; return rc; }
#define CREATE_HIP(call)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            g_create_err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            f2q_destroy(c);                                                                         \
            return F2Q_EHIP;                                                                        \
        }                                                                                           \
    } while (0)
    CREATE_HIP(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_HIP(hipEventCreate(&c->ev_a)); CREATE_HIP(hipEventCreate(&c->ev_b));
    CREATE_HIP(hipEventCreate(&c->ev_k0)); CREATE_HIP(hipEventCreate(&c->ev_k1));
    CREATE_HIP(hipMalloc((void **)&c->run_d, sizeof(RunDev)));
    CREATE_HIP(hipMemcpyAsync(c->run_d, &c->run_h, sizeof(RunDev), hipMemcpyHostToDevice, c->stream));
    // an empty library so that EC mode (and a Counter run before set_features fails cleanly) has valid pointers
    build_index(c->ix, "", (const uint32_t[]){0}, 0, c->run_h.miss, 0);
    rc = upload_lib(c);
    if (!rc) rc = alloc_acc(c, 0);
    if (rc) { g_create_err = c->err; f2q_destroy(c); return rc; }
    *out = c;
    return F2Q_OK;
}

extern "C" void f2q_destroy(f2q_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_all(c->lib_allocs); free_all(c->ec_allocs);
    if (c->acc_d) (void)hipFree(c->acc_d);
    if (c->synth_keys_d) (void)hipFree(c->synth_keys_d);
    if (c->slab_d) (void)hipFree(c->slab_d);
    if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
    if (c->hit_buf_d) (void)hipFree(c->hit_buf_d);
    if (c->run_d) (void)hipFree(c->run_d);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->ev_k0) (void)hipEventDestroy(c->ev_k0);
    if (c->ev_k1) (void)hipEventDestroy(c->ev_k1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int f2q_set_features(f2q_ctx *c, const char *seqs, const uint32_t *offs, uint32_t n)
{
    if (!c || !offs || (!seqs && n)) return fail(c, F2Q_EINVAL, "null argument");
    if (c->prm.mode != 0) return fail(c, F2Q_ESTATE, "Extract+Count mode takes no feature library (fast2q.py:1701)");
    HIPC(c, hipSetDevice(c->device));
    for (uint32_t i = 0; i < n; i++) if (offs[i + 1] < offs[i]) return fail(c, F2Q_EINVAL, "offsets must be non-decreasing");
    int packed_len = c->plan.fast_fixed ? c->run_h.length : 0;
    if (c->plan.fast_anchor) {
        if (c->run_h.has_up && c->run_h.has_down) {          // variable windows: index the most common feature length
            std::vector<uint32_t> hist(F2Q_REG_MAXLEN + 1, 0);
            for (uint32_t i = 0; i < n; i++) { uint32_t l = offs[i + 1] - offs[i]; if (l >= 1 && l <= F2Q_REG_MAXLEN) hist[l]++; }
            packed_len = (int)(std::max_element(hist.begin(), hist.end()) - hist.begin());
        } else packed_len = c->run_h.length;
    }
    build_index(c->ix, seqs ? seqs : "", offs, n, c->run_h.miss, packed_len);
    int rc = upload_lib(c);
    if (rc) return rc;
    rc = alloc_acc(c, n);
    if (rc) return rc;
    c->plan.inband_n = (c->plan.fast_fixed || c->plan.fast_anchor) && c->ix.n_irregular == 0;
    if (c->ix.n_irregular) c->plan.fast_anchor = false;      // irregular features need the byte-exact routine
    c->have_lib = true;
    return F2Q_OK;
}

extern "C" int f2q_reset_counts(f2q_ctx *c)
{
    if (!c) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemsetAsync(c->acc_d, 0, c->acc_n * sizeof(unsigned long long), c->stream));
    if (c->prm.mode == 1) { free_all(c->ec_allocs); memset(&c->ec, 0, sizeof c->ec); c->ec_slots = 0; }
    c->reads_seen = 0;
    HIPC(c, hipStreamSynchronize(c->stream));
    return F2Q_OK;
}

extern "C" int f2q_set_read_base(f2q_ctx *c, uint64_t first_read_index)
{
    if (!c) return F2Q_EINVAL;
    c->reads_seen = first_read_index;
    return F2Q_OK;
}

extern "C" int f2q_read_counts(f2q_ctx *c, int64_t *counts, int64_t stats[5])
{
    if (!c) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    std::vector<unsigned long long> h(c->acc_n);
    HIPC(c, hipMemcpyAsync(h.data(), c->acc_d, c->acc_n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (counts) for (uint64_t i = 0; i + 5 < c->acc_n; i++) counts[i] = (int64_t)h[i];
    if (stats) for (int k = 0; k < 5; k++) stats[k] = (int64_t)h[c->acc_n - 5 + k];
    return F2Q_OK;
}

extern "C" int f2q_counts_device_ptr(f2q_ctx *c, void **dptr, uint64_t *n_int64)
{
    if (!c || !dptr || !n_int64) return F2Q_EINVAL;
    *dptr = c->acc_d; *n_int64 = c->acc_n;
    return F2Q_OK;
}

// ---- Extract+Count table management -----------------------------------------------------------
static int ec_alloc(f2q_ctx *c, EcDev &e, std::vector<void *> &owner, uint64_t max_entries, uint64_t arena_words)
{
    memset(&e, 0, sizeof e);
    uint64_t slots = 1024;
    while (slots < 2 * max_entries) slots <<= 1;
    if (slots > (1ull << 32)) return fail(c, F2Q_ENOMEM, "Extract+Count table would exceed 2^32 slots");
    int rc;
    if ((rc = dev_alloc(c, slots, &e.slots, owner, 0))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_off, owner))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_len, owner))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_count, owner, 0))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_first, owner, 0xFF))) return rc;
    if ((rc = dev_alloc(c, arena_words, &e.arena, owner))) return rc;
    if ((rc = dev_alloc(c, (size_t)4, &e.ctr, owner, 0))) return rc;
    if ((rc = dev_alloc(c, slots, &e.k64_slots, owner, 0xFF))) return rc;
    if ((rc = dev_alloc(c, slots, &e.k64_count, owner, 0))) return rc;
    if ((rc = dev_alloc(c, slots, &e.k64_first, owner, 0xFF))) return rc;
    e.k64_mask = (uint32_t)(slots - 1);
    e.mask = (uint32_t)(slots - 1); e.max_entries = (uint32_t)std::min<uint64_t>(max_entries, 0xFFFFFFFEull);
    e.arena_words = arena_words;
    return F2Q_OK;
}

// make room for `reads` more reads whose keys total at most `key_bytes` bytes
static int ec_reserve(f2q_ctx *c, uint64_t reads, uint64_t key_bytes)
{
    unsigned long long ctr[4] = {0, 0, 0, 0};
    if (c->ec.slots) {
        HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
    }
    // both tables are sized for "every read brings a new key": slots = 2 * max_entries
    const uint64_t need_e = std::max(ctr[0], ctr[3]) + reads + 16, need_w = ctr[1] + (key_bytes + 3) / 4 + reads + 16;
    if (c->ec.slots && need_e <= c->ec.max_entries && need_w <= c->ec.arena_words) return F2Q_OK;
    uint64_t ne = std::max<uint64_t>(need_e * 2, 1u << 16), nw = std::max<uint64_t>(need_w * 2, 1u << 18);
    EcDev fresh; std::vector<void *> owner;
    int rc = ec_alloc(c, fresh, owner, ne, nw);
    if (rc) { free_all(owner); return rc; }
    if (c->ec.slots && ctr[0]) {
        hipLaunchKernelGGL(k_ec_rehash, dim3((unsigned)((ctr[0] + 255) / 256)), dim3(256), 0, c->stream, c->ec, ctr[0], fresh);
        HIPC(c, hipGetLastError());
    }
    if (c->ec.slots && ctr[3]) {
        hipLaunchKernelGGL(k_ec64_rehash, dim3((unsigned)(((uint64_t)c->ec.k64_mask + 256) / 256)), dim3(256), 0, c->stream, c->ec, fresh);
        HIPC(c, hipGetLastError());
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    free_all(c->ec_allocs);
    c->ec_allocs = owner; c->ec = fresh;
    return F2Q_OK;
}

// ---- launching ----------------------------------------------------------------------------------
static int launch_block(f2q_ctx *c, const f2q_block *b, f2q_timing *t)
{
    if (c->prm.mode == 0 && !c->have_lib) return fail(c, F2Q_ESTATE, "f2q_set_features must be called before counting in Counter mode");
    Accum acc{c->acc_d, c->acc_d + (c->acc_n - 5), nullptr, nullptr, nullptr, nullptr};
#ifdef F2Q_STAMP
    static unsigned long long *stamp_d = nullptr;
    if (!stamp_d) { (void)hipMalloc((void **)&stamp_d, 64); (void)hipMemset(stamp_d, 0, 64); }
    acc.stamp = stamp_d;
#endif
    if (c->prm.mode == 1 && b->n_reads) {
        // worst case every read inserts a new key made of all its windows
        uint64_t key_bytes = b->dev_bytes;   // upper bound: no key is longer than the read's bytes + separators
        int rc = ec_reserve(c, b->n_reads, key_bytes + (uint64_t)b->n_reads * F2Q_MAX_ITER);
        if (rc) return rc;
    }
    uint32_t launches = 0;
    HIPC(c, hipEventRecord(c->ev_k0, c->stream));
    if (b->pb.n_tiles && b->pb.planar_nw) {
        // packed anchored path
        const bool ecm = c->prm.mode == 1;
        const bool lds = !ecm && c->lib_h.n_features <= F2Q_HIST_MAX;
        uint32_t an_mult = 4u; { const char *e = getenv("F2Q_AN_GRID"); if (e && atoi(e) > 0) an_mult = (uint32_t)atoi(e); }
        const uint32_t grid = std::min<uint32_t>(b->pb.n_tiles, (uint32_t)c->n_cu * an_mult);
        const size_t shmem = (size_t)F2Q_AN_WAVES * F2Q_AN_QCAP * 12 + (lds ? (size_t)c->lib_h.n_features * 4 : 0);
        if (!ecm) {
            const size_t need = lds ? (size_t)grid * c->lib_h.n_features : 0;
            if (need > c->slab_n || (size_t)grid > c->stat_slab_n) {
                if (c->slab_d) (void)hipFree(c->slab_d);
                if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
                c->slab_d = nullptr; c->slab_n = 0; c->stat_slab_d = nullptr; c->stat_slab_n = 0;
                HIPC(c, hipMalloc((void **)&c->slab_d, std::max<size_t>(need, 1) * sizeof(uint32_t)));
                HIPC(c, hipMalloc((void **)&c->stat_slab_d, (size_t)grid * 8 * sizeof(unsigned long long)));
                c->slab_n = need; c->stat_slab_n = grid;
            }
            if (lds) { acc.slab = c->slab_d; acc.stat_slab = c->stat_slab_d; }
        }
        const int nw = (int)b->pb.planar_nw, kb = c->plan.kb;
        const bool sameq = c->run_h.thr_up == c->run_h.thr && c->run_h.thr_down == c->run_h.thr;
#define F2Q_LAUNCH_AN2(NW_, KB_, SQ_)                                                                                \
        do {                                                                                                         \
            if (ecm) hipLaunchKernelGGL((k_count_anchor<NW_, KB_, true, false, SQ_>), dim3(grid), dim3(F2Q_AN_THREADS),   \
                                        shmem, c->stream, c->run_d, c->lib_d, c->ec, b->pb, acc, c->reads_seen);     \
            else if (lds) hipLaunchKernelGGL((k_count_anchor<NW_, KB_, false, true, SQ_>), dim3(grid),               \
                                             dim3(F2Q_AN_THREADS), shmem, c->stream, c->run_d, c->lib_d, c->ec,      \
                                             b->pb, acc, c->reads_seen);                                             \
            else hipLaunchKernelGGL((k_count_anchor<NW_, KB_, false, false, SQ_>), dim3(grid), dim3(F2Q_AN_THREADS),  \
                                    shmem, c->stream, c->run_d, c->lib_d, c->ec, b->pb, acc, c->reads_seen);         \
        } while (0)
#define F2Q_LAUNCH_AN(NW_, KB_) do { if (sameq) F2Q_LAUNCH_AN2(NW_, KB_, true); else F2Q_LAUNCH_AN2(NW_, KB_, false); } while (0)
        if (nw == 3 && kb == 0) F2Q_LAUNCH_AN(3, 0);
        else if (nw == 3 && kb == 1) F2Q_LAUNCH_AN(3, 1);
        else if (nw == 3) F2Q_LAUNCH_AN(3, 3);
        else if (kb == 0) F2Q_LAUNCH_AN(5, 0);
        else if (kb == 1) F2Q_LAUNCH_AN(5, 1);
        else F2Q_LAUNCH_AN(5, 3);
#undef F2Q_LAUNCH_AN
#undef F2Q_LAUNCH_AN2
        HIPC(c, hipGetLastError());
        launches++;
        if (lds && c->lib_h.n_features) {
            hipLaunchKernelGGL(k_reduce_slabs, dim3((c->lib_h.n_features + 63) / 64, F2Q_RED_SPLIT), dim3(256), 0, c->stream,
                               c->slab_d, grid, c->lib_h.n_features, acc.counts, c->stat_slab_d, grid, acc.stats);
            HIPC(c, hipGetLastError());
            launches++;
        }
    } else if (b->pb.n_tiles && c->prm.mode == 1) {
        const uint32_t wgs = (b->pb.n_tiles + F2Q_V2_WAVES - 1) / F2Q_V2_WAVES;
        const uint32_t grid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu * 4u);
        hipLaunchKernelGGL(k_extract_fixed4, dim3(grid), dim3(F2Q_V2_THREADS), 0, c->stream, c->run_d, c->ec, b->pb, acc, c->reads_seen);
        HIPC(c, hipGetLastError());
        launches++;
    } else if (b->pb.n_tiles) {
        const bool lds = c->lib_h.n_features <= F2Q_HIST_MAX;
        const bool v2 = !c->force_v1 && c->lib_h.pk.len == (uint32_t)c->run_h.length && c->lib_h.pk.len > 0 &&
                        c->lib_h.n_irregular == 0;
        if (v2) {
            const uint32_t wgs = (b->pb.n_tiles + F2Q_V2_WAVES - 1) / F2Q_V2_WAVES;
            const uint32_t grid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu * 2u);
            const size_t shmem = (size_t)F2Q_V2_WAVES * F2Q_V2_QCAP * 12 + (lds ? (size_t)c->lib_h.n_features * 4 : (size_t)F2Q_V2_WAVES * F2Q_V2_QCAP * 4);
            const FixedGeom fg = fixed_geom(c->run_h);
            const bool spec52 = !c->force_generic && fg.nq == 5 && fg.nb == 2 && c->run_h.thr >= 33;
            auto kern = lds ? (spec52 ? k_count_fixed4<true, 5, 2> : k_count_fixed4<true, 0, 0>)
                            : (spec52 ? k_count_fixed4<false, 5, 2> : k_count_fixed4<false, 0, 0>);
            const uint32_t nf_ = c->lib_h.n_features;
            const uint32_t n_ranges = lds ? 1u : (nf_ + F2Q_HIST_MAX - 1) / F2Q_HIST_MAX;
            const uint32_t n_parts = lds ? grid : std::max<uint32_t>(1u, (uint32_t)c->n_cu / n_ranges);
            {
                // slab rows: one per counting workgroup (LDS histogram) or one per part of k_hist_ranges; stats rows per workgroup
                const size_t need = (size_t)n_parts * nf_;
                if (need > c->slab_n || (size_t)grid > c->stat_slab_n) {
                    if (c->slab_d) (void)hipFree(c->slab_d);
                    if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
                    c->slab_d = nullptr; c->slab_n = 0; c->stat_slab_d = nullptr; c->stat_slab_n = 0;
                    HIPC(c, hipMalloc((void **)&c->slab_d, std::max<size_t>(need, 1) * sizeof(uint32_t)));
                    HIPC(c, hipMalloc((void **)&c->stat_slab_d, (size_t)grid * 8 * sizeof(unsigned long long)));
                    c->slab_n = need; c->stat_slab_n = grid;
                }
                acc.slab = c->slab_d;
                acc.stat_slab = c->stat_slab_d;
            }
            if (!lds) {
                if (b->pb.n_slots > c->hit_buf_n) {
                    if (c->hit_buf_d) (void)hipFree(c->hit_buf_d);
                    c->hit_buf_d = nullptr; c->hit_buf_n = 0;
                    HIPC(c, hipMalloc((void **)&c->hit_buf_d, b->pb.n_slots * sizeof(uint32_t)));
                    c->hit_buf_n = b->pb.n_slots;
                }
                HIPC(c, hipMemsetAsync(c->hit_buf_d, 0xFF, b->pb.n_slots * sizeof(uint32_t), c->stream));
                acc.hit_buf = c->hit_buf_d;
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(F2Q_V2_THREADS), shmem, c->stream, c->run_d, c->lib_d, b->pb, acc);
            HIPC(c, hipGetLastError());
            if (!lds && nf_) {
                hipLaunchKernelGGL(k_hist_ranges, dim3(n_ranges * n_parts), dim3(1024), (size_t)F2Q_HIST_MAX * 4, c->stream,
                                   c->hit_buf_d, (uint64_t)b->pb.n_slots, nf_, n_parts, c->slab_d);
                HIPC(c, hipGetLastError());
                launches++;
            }
            if (nf_) {
                hipLaunchKernelGGL(k_reduce_slabs, dim3((nf_ + 63) / 64, F2Q_RED_SPLIT), dim3(256), 0, c->stream,
                                   c->slab_d, n_parts, nf_, acc.counts, c->stat_slab_d, grid, acc.stats);
                launches++;
            }
        } else {
            const uint32_t grid = std::min<uint32_t>(b->pb.n_tiles, (uint32_t)c->n_cu * 8u);
            if (lds) {
                size_t shmem = std::max<size_t>(4, (size_t)c->lib_h.n_features * 4);
                hipLaunchKernelGGL(k_count_fixed<true>, dim3(grid), dim3(F2Q_TILE), shmem, c->stream, c->run_d, c->lib_d, b->pb, acc);
            } else {
                hipLaunchKernelGGL(k_count_fixed<false>, dim3(grid), dim3(F2Q_TILE), 0, c->stream, c->run_d, c->lib_d, b->pb, acc);
            }
        }
        HIPC(c, hipGetLastError());
        launches++;
    }
    if (b->rb.n) {
        RawBlock rb = b->rb;
        rb.first_index += c->reads_seen;
        const uint64_t wg = (rb.n + 255) / 256;
        const uint32_t grid = (uint32_t)std::min<uint64_t>(wg, (uint64_t)c->n_cu * 16u);
        hipLaunchKernelGGL(k_count_general, dim3(grid), dim3(256), 0, c->stream, c->run_d, c->lib_d, c->ec, rb, acc);
        HIPC(c, hipGetLastError());
        launches++;
    }
    HIPC(c, hipEventRecord(c->ev_k1, c->stream));
    c->reads_seen += b->n_reads;
#ifdef F2Q_STAMP
    { unsigned long long h[4]; (void)hipStreamSynchronize(c->stream); (void)hipMemcpy(h, acc.stamp, 32, hipMemcpyDeviceToHost);
      (void)hipMemset(acc.stamp, 0, 64);
      unsigned long long tot = h[0] + h[1] + h[2] + h[3];
      if (tot) fprintf(stderr, "[stamp] loads+fail %.1f%%  anchors %.1f%%  key/probe %.1f%%  drain %.1f%%  (%.0f cycles/wave-tile)\n",
                       100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot,
                       (double)tot / ((double)b->pb.n_tiles * 4)); }
#endif
    if (t) {
        HIPC(c, hipEventSynchronize(c->ev_k1));
        float ms = 0;
        HIPC(c, hipEventElapsedTime(&ms, c->ev_k0, c->ev_k1));
        t->kernel_ms = ms; t->reads = b->n_reads; t->general_reads = b->n_general;
        t->fast_reads = b->n_reads - b->n_general; t->launches = launches;
    }
    if (c->prm.mode == 1 && b->n_reads) {
        unsigned long long ctr[4];
        HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error)");
    }
    return F2Q_OK;
}

extern "C" int f2q_count_resident(f2q_ctx *c, const f2q_block *b, f2q_timing *t)
{
    if (!c || !b) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    if (t) { memset(t, 0, sizeof *t); HIPC(c, hipEventRecord(c->ev_a, c->stream)); }
    int rc = launch_block(c, b, t);
    if (rc) return rc;
    if (t) {
        HIPC(c, hipEventRecord(c->ev_b, c->stream));
        HIPC(c, hipEventSynchronize(c->ev_b));
        float ms = 0; HIPC(c, hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
        t->total_ms = ms;
    }
    return F2Q_OK;
}

extern "C" void f2q_block_free(f2q_ctx *c, f2q_block *b)
{
    if (!b) return;
    if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    free_all(b->allocs);
    delete b;
}

extern "C" int f2q_block_info(const f2q_block *b, uint64_t *n_reads, uint64_t *n_general, uint64_t *device_bytes)
{
    if (!b) return F2Q_EINVAL;
    if (n_reads) *n_reads = b->n_reads;
    if (n_general) *n_general = b->n_general;
    if (device_bytes) *device_bytes = b->dev_bytes;
    return F2Q_OK;
}

static int block_from_records(f2q_ctx *c, const std::vector<Rec> &recs, f2q_block **out)
{
    HostPacked hp;
    pack_records(c->plan, recs, hp);
    f2q_block *b = new f2q_block();
    b->n_reads = recs.size(); b->n_general = hp.g_len.size();
    int rc = F2Q_OK;
    do {
        if (hp.n_tiles) {
            uint32_t *db, *dq; uint16_t *dl;
            if ((rc = dev_upload(c, hp.bases.data(), hp.bases.size(), &db, b->allocs))) break;
            if ((rc = dev_upload(c, hp.qual.data(), hp.qual.size(), &dq, b->allocs))) break;
            if ((rc = dev_upload(c, hp.len.data(), hp.len.size(), &dl, b->allocs))) break;
            uint32_t *dci = nullptr;
            if (c->prm.mode == 1 && (rc = dev_upload(c, hp.c_index.data(), hp.c_index.size(), &dci, b->allocs))) break;
            b->pb.index = dci;
            b->pb.n_slots = (uint64_t)hp.n_tiles * F2Q_TILE; b->pb.n_tiles = hp.n_tiles;
            b->pb.wb = hp.wb; b->pb.wq = hp.wq; b->pb.rmax = hp.rmax; b->pb.planar_nw = hp.planar_nw;
            b->pb.bases = db; b->pb.qual = dq; b->pb.len = dl;
            b->dev_bytes += hp.bases.size() * 4 + hp.qual.size() * 4 + hp.len.size() * 2;
        }
        if (!hp.g_len.empty()) {
            uint8_t *dr; unsigned long long *doff; uint32_t *dlen, *dqlen, *dix;
            if ((rc = dev_upload(c, hp.raw.data(), hp.raw.size(), &dr, b->allocs))) break;
            if ((rc = dev_upload(c, hp.g_off.data(), hp.g_off.size(), &doff, b->allocs))) break;
            if ((rc = dev_upload(c, hp.g_len.data(), hp.g_len.size(), &dlen, b->allocs))) break;
            if ((rc = dev_upload(c, hp.g_qlen.data(), hp.g_qlen.size(), &dqlen, b->allocs))) break;
            if ((rc = dev_upload(c, hp.g_index.data(), hp.g_index.size(), &dix, b->allocs))) break;
            b->rb.n = hp.g_len.size(); b->rb.raw = dr; b->rb.off = doff; b->rb.len = dlen; b->rb.qlen = dqlen; b->rb.index = dix;
            b->dev_bytes += hp.raw.size() + hp.g_len.size() * 20;
        }
        hipError_t e = hipStreamSynchronize(c->stream);      // host staging vectors die with this frame
        if (e != hipSuccess) { rc = fail(c, F2Q_EHIP, hipGetErrorString(e)); break; }
    } while (0);
    if (rc) { free_all(b->allocs); delete b; return rc; }
    *out = b;
    return F2Q_OK;
}


// FASTQ text -> resident block, framing and packing done by the device (k_nl_count .. k_pack).  Handles up to
// 2 GiB of text per call; *consumed = bytes up to the end of the last complete record.
static int block_from_text_device(f2q_ctx *c, const uint8_t *fastq, size_t nbytes, size_t *consumed, f2q_block **out)
{
    *out = nullptr; *consumed = 0;
    f2q_block *b = new f2q_block();
    if (nbytes == 0) { *out = b; return F2Q_OK; }    // an empty buffer is an empty block
    std::vector<void *> tmp;                         // scratch freed before returning
    int rc = F2Q_OK;
    auto bail = [&](int code) { free_all(tmp); free_all(b->allocs); delete b; return code; };
    const uint32_t n_chunks = (uint32_t)((nbytes + F2Q_NL_CHUNK - 1) / F2Q_NL_CHUNK);
    const size_t padded = (size_t)n_chunks * F2Q_NL_CHUNK + 16;
    uint8_t *d_text; uint32_t *d_cc, *d_cp;
    if ((rc = dev_alloc(c, padded, &d_text, b->allocs))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_chunks + 1, &d_cc, tmp, 0))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_chunks + 1, &d_cp, tmp))) return bail(rc);
#define ING(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(c, F2Q_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); return bail(F2Q_EHIP); } } while (0)
    ING(hipMemsetAsync(d_text + nbytes, 0, padded - nbytes, c->stream));
    ING(hipMemcpyAsync(d_text, fastq, nbytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_nl_count, dim3(n_chunks), dim3(256), 0, c->stream, d_text, (uint64_t)nbytes, d_cc);
    ING(hipGetLastError());
    size_t cub_bytes = 0;
    ING(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_bytes, d_cc, d_cp, (int)n_chunks + 1, c->stream));
    uint8_t *d_cub;
    if ((rc = dev_alloc(c, cub_bytes + 16, &d_cub, tmp))) return bail(rc);
    ING(hipcub::DeviceScan::ExclusiveSum(d_cub, cub_bytes, d_cc, d_cp, (int)n_chunks + 1, c->stream));
    uint32_t n_newlines = 0;
    ING(hipMemcpyAsync(&n_newlines, d_cp + n_chunks, 4, hipMemcpyDeviceToHost, c->stream));
    ING(hipStreamSynchronize(c->stream));
    const bool open_tail = nbytes > 0 && fastq[nbytes - 1] != '\n';
    const uint64_t n_lines = (uint64_t)n_newlines + (open_tail ? 1 : 0);
    const uint32_t n_rec = (uint32_t)(n_lines / 4);
    uint32_t *d_ls;
    if ((rc = dev_alloc(c, (size_t)n_newlines + 2, &d_ls, tmp))) return bail(rc);
    hipLaunchKernelGGL(k_line_starts, dim3(n_chunks), dim3(256), 0, c->stream, d_text, (uint64_t)nbytes, d_cp, d_ls);
    ING(hipGetLastError());
    const uint32_t sentinel = (uint32_t)nbytes + 1u;
    ING(hipMemcpyAsync(d_ls + n_newlines + 1, &sentinel, 4, hipMemcpyHostToDevice, c->stream));
    b->n_reads = n_rec;
    if (n_rec == 0) { ING(hipStreamSynchronize(c->stream)); free_all(tmp); *out = b; return F2Q_OK; }
    // bytes consumed: the start of line 4*n_rec, or everything when the last record's last line is unterminated
    uint32_t cons32 = (uint32_t)nbytes;
    if ((uint64_t)4 * n_rec <= n_newlines) ING(hipMemcpyAsync(&cons32, d_ls + (size_t)4 * n_rec, 4, hipMemcpyDeviceToHost, c->stream));
    IngestDev ing{};
    ing.text = d_text; ing.line_start = d_ls; ing.n_records = n_rec;
    uint32_t *d_before;
    if ((rc = dev_alloc(c, (size_t)n_rec, &ing.r_off, tmp))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_rec, &ing.r_len, tmp))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_rec, &ing.r_qoff, tmp))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_rec, &ing.r_qlen, tmp))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_rec + 1, &ing.clean, tmp, 0))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_rec + 1, &d_before, tmp))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)4, &ing.meta, tmp, 0))) return bail(rc);
    const unsigned rgrid = (n_rec + 255u) / 256u;
    hipLaunchKernelGGL(k_classify, dim3(rgrid), dim3(256), 0, c->stream, ing, c->plan);
    ING(hipGetLastError());
    size_t cub2 = 0;
    ING(hipcub::DeviceScan::ExclusiveSum(nullptr, cub2, ing.clean, d_before, (int)n_rec + 1, c->stream));
    uint8_t *d_cub2;
    if ((rc = dev_alloc(c, cub2 + 16, &d_cub2, tmp))) return bail(rc);
    ING(hipcub::DeviceScan::ExclusiveSum(d_cub2, cub2, ing.clean, d_before, (int)n_rec + 1, c->stream));
    uint32_t n_clean = 0, rmax_in = 0;
    ING(hipMemcpyAsync(&n_clean, d_before + n_rec, 4, hipMemcpyDeviceToHost, c->stream));
    ING(hipMemcpyAsync(&rmax_in, ing.meta, 4, hipMemcpyDeviceToHost, c->stream));
    ING(hipStreamSynchronize(c->stream));
    *consumed = cons32;
    const uint32_t n_dirty = n_rec - n_clean;
    PackOut o{};
    if (n_clean) {
        uint32_t rmax, nw, wb, wq;
        tile_geometry(c->plan, rmax_in, rmax, nw, wb, wq);
        const uint32_t n_tiles = (n_clean + F2Q_TILE - 1) / F2Q_TILE;
        if ((rc = dev_alloc(c, (size_t)n_tiles * wb * F2Q_TILE, &o.bases, b->allocs, 0))) return bail(rc);
        if ((rc = dev_alloc(c, (size_t)n_tiles * wq * F2Q_TILE, &o.qual, b->allocs, 0))) return bail(rc);
        if ((rc = dev_alloc(c, (size_t)n_tiles * F2Q_TILE, &o.len, b->allocs, 0xFF))) return bail(rc);
        if (c->prm.mode == 1 && (rc = dev_alloc(c, (size_t)n_tiles * F2Q_TILE, &o.c_index, b->allocs, 0))) return bail(rc);
        o.wb = wb; o.wq = wq; o.planar_nw = nw;
        b->pb.n_slots = (uint64_t)n_tiles * F2Q_TILE; b->pb.n_tiles = n_tiles; b->pb.wb = wb; b->pb.wq = wq; b->pb.rmax = rmax;
        b->pb.planar_nw = nw; b->pb.bases = o.bases; b->pb.qual = o.qual; b->pb.len = o.len; b->pb.index = o.c_index;
        b->dev_bytes += (uint64_t)n_tiles * F2Q_TILE * ((wb + wq) * 4 + 2);
    }
    if (n_dirty) {
        if ((rc = dev_alloc(c, (size_t)n_dirty, &o.g_off, b->allocs))) return bail(rc);
        if ((rc = dev_alloc(c, (size_t)n_dirty, &o.g_qoff, b->allocs))) return bail(rc);
        if ((rc = dev_alloc(c, (size_t)n_dirty, &o.g_len, b->allocs))) return bail(rc);
        if ((rc = dev_alloc(c, (size_t)n_dirty, &o.g_qlen, b->allocs))) return bail(rc);
        if ((rc = dev_alloc(c, (size_t)n_dirty, &o.g_index, b->allocs))) return bail(rc);
        b->rb.n = n_dirty; b->rb.raw = d_text; b->rb.off = o.g_off; b->rb.qoff = o.g_qoff; b->rb.len = o.g_len;
        b->rb.qlen = o.g_qlen; b->rb.index = o.g_index;
        b->dev_bytes += nbytes;                       // upper bound of the key bytes an Extract+Count run can add
    }
    b->n_general = n_dirty;
    hipLaunchKernelGGL(k_pack, dim3(rgrid), dim3(256), 0, c->stream, ing, c->plan, d_before, o);
    ING(hipGetLastError());
    ING(hipStreamSynchronize(c->stream));             // scratch dies with this frame
#undef ING
    free_all(tmp);
    *out = b;
    return F2Q_OK;
}

extern "C" int f2q_block_from_fastq(f2q_ctx *c, const uint8_t *fastq, size_t nbytes, f2q_block **out)
{
    if (!c || !out || (!fastq && nbytes)) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    if (!c->host_pack && nbytes < ((size_t)1 << 31)) { size_t used; return block_from_text_device(c, fastq, nbytes, &used, out); }
    std::vector<Rec> recs;
    frame_fastq(fastq, nbytes, recs);
    return block_from_records(c, recs, out);
}

extern "C" int f2q_count_block(f2q_ctx *c, const uint8_t *fastq, size_t nbytes, size_t *consumed, f2q_timing *t)
{
    if (!c || (!fastq && nbytes)) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    if (t) { memset(t, 0, sizeof *t); HIPC(c, hipEventRecord(c->ev_a, c->stream)); }
    if (consumed) *consumed = 0;
    int rc = F2Q_OK;
    size_t pos = 0;
    f2q_timing sum; memset(&sum, 0, sizeof sum);
    while (pos < nbytes) {
        // the device packer indexes the text with 32 bits: feed it at most 1 GiB at a time (record aligned by itself)
        const size_t take = std::min<size_t>(nbytes - pos, (size_t)1 << 30);
        f2q_block *b = nullptr; size_t used = 0;
        if (!c->host_pack) rc = block_from_text_device(c, fastq + pos, take, &used, &b);
        else {
            std::vector<Rec> recs;
            used = frame_fastq(fastq + pos, take, recs);
            rc = recs.empty() ? F2Q_OK : block_from_records(c, recs, &b);
        }
        if (rc) return rc;
        f2q_timing one; memset(&one, 0, sizeof one);
        if (b && b->n_reads) rc = launch_block(c, b, t ? &one : nullptr);
        if (b) f2q_block_free(c, b);
        if (rc) return rc;
        sum.kernel_ms += one.kernel_ms; sum.reads += one.reads; sum.fast_reads += one.fast_reads;
        sum.general_reads += one.general_reads; sum.launches += one.launches;
        if (used == 0) break;                          // no complete record left in this window
        pos += used;
        if (take < ((size_t)1 << 30)) break;           // that was the tail: what is left is a partial record
    }
    if (consumed) *consumed = pos;
    if (t) {
        hipError_t e = hipEventRecord(c->ev_b, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev_b);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_a, c->ev_b);
        if (e != hipSuccess) return fail(c, F2Q_EHIP, hipGetErrorString(e));
        *t = sum; t->total_ms = ms;
    }
    return F2Q_OK;
}

// reads_counter's file half (fast2q.py:560-578): gzip or plain by extension; streamed in blocks
extern "C" int f2q_count_file(f2q_ctx *c, const char *path, f2q_timing *t)
{
    if (!c || !path) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    size_t CH = (size_t)256 << 20;                 // bytes of text per block; F2Q_FILE_CHUNK overrides (tests)
    { const char *e = getenv("F2Q_FILE_CHUNK"); if (e && atol(e) >= 4096) CH = (size_t)atol(e); }
    // gzip by content (1f 8b), like gzip.open() by extension upstream (:567); plain files are read() straight
    // into the pinned buffer, without zlib's pass-through copy
    FILE *pf = fopen(path, "rb");
    if (!pf) return fail(c, F2Q_EIO, std::string("cannot open ") + path);
    unsigned char magic[2] = {0, 0};
    const size_t got_magic = fread(magic, 1, 2, pf);
    const bool is_gz = got_magic == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    rewind(pf);
    gzFile f = nullptr;
    if (is_gz) {
        fclose(pf); pf = nullptr;
        f = gzopen(path, "rb");
        if (!f) return fail(c, F2Q_EIO, std::string("cannot open ") + path);
        gzbuffer(f, 1 << 20);
    }
    // page-locked staging buffer: the H2D copy of the text then runs at DMA speed
    struct Pinned {
        uint8_t *p = nullptr; size_t n = 0;
        ~Pinned() { if (p) (void)hipHostFree(p); }
        uint8_t *data() { return p; }
        size_t size() const { return n; }
        bool resize(size_t m) {
            uint8_t *q = nullptr;
            if (hipHostMalloc((void **)&q, m, hipHostMallocDefault) != hipSuccess) return false;
            if (p) { memcpy(q, p, n < m ? n : m); (void)hipHostFree(p); }
            p = q; n = m; return true;
        }
    } buf;
    if (!buf.resize(CH)) { if (f) gzclose(f); if (pf) fclose(pf); return fail(c, F2Q_ENOMEM, "cannot allocate the pinned read buffer"); }
    size_t have = 0;
    f2q_timing sum; memset(&sum, 0, sizeof sum);
    int rc = F2Q_OK; bool truncated = false;
    for (;;) {
        long got;
        if (f) {
            got = gzread(f, buf.data() + have, (unsigned)std::min<size_t>(buf.size() - have, 1u << 30));
            if (got < 0) { truncated = true; got = 0; }
        } else got = (long)fread(buf.data() + have, 1, buf.size() - have, pf);
        have += (size_t)got;
        const bool eof = (got == 0);
        if (have == 0) break;
        size_t used = 0; f2q_timing one;
        if (eof) {
            rc = f2q_count_block(c, buf.data(), have, &used, &one);     // trailing partial record is dropped (:392)
            used = have;
        } else {
            // only hand over whole lines: cut at the last newline so a line is never split between blocks
            size_t cut = have;
            while (cut > 0 && buf.data()[cut - 1] != 0x0a) cut--;
            if (cut == 0) { if (have == buf.size() && !buf.resize(buf.size() * 2)) { rc = fail(c, F2Q_ENOMEM, "line longer than memory"); break; } continue; }
            rc = f2q_count_block(c, buf.data(), cut, &used, &one);
        }
        if (rc) break;
        sum.kernel_ms += one.kernel_ms; sum.total_ms += one.total_ms; sum.reads += one.reads;
        sum.fast_reads += one.fast_reads; sum.general_reads += one.general_reads; sum.launches += one.launches;
        memmove(buf.data(), buf.data() + used, have - used);
        have -= used;
        if (eof) break;
        if (have == buf.size() && !buf.resize(buf.size() * 2)) { rc = fail(c, F2Q_ENOMEM, "record longer than memory"); break; }
    }
    if (f) {
        int zerr = 0; (void)gzerror(f, &zerr);
        if (zerr == Z_BUF_ERROR || zerr == Z_DATA_ERROR) truncated = true;
        gzclose(f);
    }
    if (pf) fclose(pf);
    if (t) *t = sum;
    if (rc) return rc;
    if (truncated) return fail(c, F2Q_ETRUNCATED, std::string(path) + " is an incomplete or corrupted gzip file");
    return F2Q_OK;
}

// ---- synthetic workload ---------------------------------------------------------------------------
extern "C" int f2q_synth_guides(f2q_ctx *c, const char *seqs, uint32_t n, uint32_t length)
{
    if (!c || !seqs || n == 0 || length < 1 || length > F2Q_REG_MAXLEN) return fail(c, F2Q_EINVAL, "f2q_synth_guides: bad argument");
    HIPC(c, hipSetDevice(c->device));
    std::vector<uint64_t> keys(n);
    for (uint32_t g = 0; g < n; g++)
        if (!feature_key((const uint8_t *)seqs + (size_t)g * length, length, keys[g])) return fail(c, F2Q_EINVAL, "synthetic guides must be ACGT");
    if (c->synth_keys_d) { (void)hipFree(c->synth_keys_d); c->synth_keys_d = nullptr; }
    HIPC(c, hipMalloc((void **)&c->synth_keys_d, (size_t)n * 8));
    HIPC(c, hipMemcpy(c->synth_keys_d, keys.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    c->synth_keys.swap(keys); c->synth_glen = length;
    return F2Q_OK;
}

static int synth_to_dev(f2q_ctx *c, const f2q_synth *s, SynthDev &d)
{
    memset(&d, 0, sizeof d);
    uint32_t glen;
    if (!c->synth_keys.empty()) { glen = c->synth_glen; d.n_guides = (int)c->synth_keys.size(); }
    else {
        if (!c->have_lib || c->ix.n_features == 0) return fail(c, F2Q_ESTATE, "synthetic reads need guides (f2q_set_features or f2q_synth_guides)");
        glen = c->ix.feat_off[1] - c->ix.feat_off[0];
        for (uint32_t f = 0; f < c->ix.n_features; f++) {
            uint64_t k;
            if (c->ix.feat_off[f + 1] - c->ix.feat_off[f] != glen || !feature_key(c->ix.feat_bytes.data() + c->ix.feat_off[f], glen, k))
                return fail(c, F2Q_EUNSUPPORTED, "synthetic reads need a uniform-length ACGT library (<= 31 bp)");
        }
        d.n_guides = (int)c->ix.n_features;
    }
    d.seed = s->seed; d.n_reads = s->n_reads; d.first_read = s->first_read;
    d.read_len = s->read_len; d.start = s->start; d.cassette = s->cassette; d.max_offset = s->max_offset;
    d.glen = (int)glen;
    d.t_sub = s->t_sub; d.t_rand = s->t_rand; d.t_n = s->t_n; d.t_lowq = s->t_lowq; d.t_q29 = s->t_q29; d.t_q28 = s->t_q28;
    if (s->read_len < 1 || s->read_len > F2Q_PACK_MAXLEN) return fail(c, F2Q_EINVAL, "read_len must be 1..512");
    if (s->cassette) {
        size_t ul = s->up ? strlen(s->up) : 0, dl = s->down ? strlen(s->down) : 0;
        if (ul > 63 || dl > 63) return fail(c, F2Q_EINVAL, "cassette flanks longer than 63");
        d.up_len = (int)ul; d.down_len = (int)dl;
        if (ul) memcpy(d.up, s->up, ul);
        if (dl) memcpy(d.down, s->down, dl);
    }
    return F2Q_OK;
}

extern "C" int f2q_synth_library(uint64_t seed, uint32_t n, uint32_t length, char *out)
{
    if (!out || length < 1 || length > 32) return F2Q_EINVAL;
    // tests/synth.py make_library: candidate k = bases of rnd(seed, k, 0); duplicates skipped
    std::vector<uint64_t> seen; seen.reserve(n);
    std::vector<uint64_t> sorted;
    uint64_t k = 0; uint32_t made = 0;
    const uint64_t mask = length >= 32 ? ~0ull : ((1ull << (2 * length)) - 1ull);
    // open-addressing set
    uint64_t cap = 16; while (cap < 4ull * n) cap <<= 1;
    std::vector<uint64_t> set(cap, ~0ull);
    std::vector<uint8_t> used(cap, 0);
    while (made < n) {
        uint64_t v = rnd(seed, k++, 0) & mask;
        uint64_t h = mix64(v) & (cap - 1);
        bool dup = false;
        while (used[h]) { if (set[h] == v) { dup = true; break; } h = (h + 1) & (cap - 1); }
        if (dup) continue;
        used[h] = 1; set[h] = v;
        for (uint32_t j = 0; j < length; j++) out[(size_t)made * length + j] = "ACGT"[(v >> (2 * j)) & 3];
        made++;
        if (k > 64ull * n + 1024) return F2Q_EINVAL;    // sequence space exhausted
    }
    return F2Q_OK;
}

extern "C" int f2q_synth_fastq(f2q_ctx *c, const f2q_synth *s, uint64_t lo, uint64_t hi, uint8_t *buf, size_t *nbytes)
{
    if (!c || !s || !nbytes || hi < lo) return F2Q_EINVAL;
    SynthDev d; int rc = synth_to_dev(c, s, d);
    if (rc) return rc;
    const int R = d.read_len;
    size_t need = 0;
    for (uint64_t i = lo; i < hi; i++) {
        char name[32]; int nl = snprintf(name, sizeof name, "@r%llu\n", (unsigned long long)i);
        need += (size_t)nl + (size_t)R + 3 + (size_t)R + 1;
    }
    if (!buf) { *nbytes = need; return F2Q_OK; }
    if (*nbytes < need) { *nbytes = need; return fail(c, F2Q_EINVAL, "buffer too small"); }
    size_t o = 0;
    const uint64_t *keys = c->synth_keys.empty() ? c->ix.key2.data() : c->synth_keys.data();
    for (uint64_t i = lo; i < hi; i++) {
        SynthRead r = synth_plan(d, i, [&](uint32_t g) { return keys[g]; });
        o += (size_t)snprintf((char *)buf + o, 32, "@r%llu\n", (unsigned long long)i);
        uint64_t fw = 0;
        for (int p = 0; p < R; p++) {
            if ((p & 31) == 0) fw = rnd(d.seed, i, F_FLANK0 + (p >> 5));
            buf[o + p] = synth_base(d, r, p, fw);
        }
        o += (size_t)R;
        buf[o++] = '\n'; buf[o++] = '+'; buf[o++] = '\n';
        for (int p = 0; p < R; p++) buf[o + p] = (p == r.qpos) ? r.qchar : (uint8_t)'I';
        o += (size_t)R;
        buf[o++] = '\n';
    }
    *nbytes = o;
    return F2Q_OK;
}

extern "C" int f2q_synth_create(f2q_ctx *c, const f2q_synth *s, f2q_block **out)
{
    if (!c || !s || !out) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    SynthDev d; int rc = synth_to_dev(c, s, d);
    if (rc) return rc;
    const int R = d.read_len;
    f2q_block *b = new f2q_block();
    b->n_reads = s->n_reads;
    const bool planar = c->plan.fast_anchor && R <= F2Q_ANCHOR_MAXLEN;
    const bool fast = c->plan.fast_fixed || planar;
    SynthOut o; memset(&o, 0, sizeof o);
    o.all_general = fast ? 0 : 1;
    o.inband_n = c->plan.inband_n ? 1 : 0;
    o.planar_nw = planar ? (R <= 96 ? 3u : 5u) : 0u;
    const uint64_t n_tiles = (s->n_reads + F2Q_TILE - 1) / F2Q_TILE;
    const uint64_t n_slots = n_tiles * F2Q_TILE;
    // general-path capacity: everything, or the expected 'N' share with a wide margin
    double pn = (double)d.t_n / 4294967296.0;
    uint64_t gcap = fast ? (uint64_t)((double)s->n_reads * pn * 1.5 + 6.0 * sqrt((double)s->n_reads * pn + 1.0) + 1024.0) : s->n_reads;
    if (gcap > s->n_reads) gcap = s->n_reads;
    if (gcap == 0) gcap = 1;
    do {
        if (fast) {
            o.wb = planar ? 2 * o.planar_nw : (uint32_t)((R + 15) / 16);
            o.wq = planar ? 8 * o.planar_nw : (uint32_t)((R + 3) / 4);
            if ((rc = dev_alloc(c, (size_t)n_tiles * o.wb * F2Q_TILE, &o.bases, b->allocs, 0))) break;
            if ((rc = dev_alloc(c, (size_t)n_tiles * o.wq * F2Q_TILE, &o.qual, b->allocs, 0))) break;
            if ((rc = dev_alloc(c, (size_t)n_slots, &o.len, b->allocs))) break;
            b->dev_bytes += (uint64_t)n_tiles * (o.wb + o.wq) * F2Q_TILE * 4 + n_slots * 2;
        }
        if ((rc = dev_alloc(c, (size_t)gcap * 2 * R + 8, &o.raw, b->allocs))) break;
        if ((rc = dev_alloc(c, (size_t)gcap, &o.off, b->allocs))) break;
        if ((rc = dev_alloc(c, (size_t)gcap, &o.glen, b->allocs))) break;
        if ((rc = dev_alloc(c, (size_t)gcap, &o.gqlen, b->allocs))) break;
        if ((rc = dev_alloc(c, (size_t)gcap, &o.gindex, b->allocs))) break;
        if ((rc = dev_alloc(c, (size_t)1, &o.g_count, b->allocs, 0))) break;
        o.g_cap = gcap;
        hipLaunchKernelGGL(k_synth, dim3((unsigned)n_tiles), dim3(F2Q_TILE), 0, c->stream, d, c->synth_keys.empty() ? c->guide_keys_d : c->synth_keys_d, o, n_slots);
        hipError_t e = hipGetLastError();
        unsigned long long g = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&g, o.g_count, sizeof g, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(c, F2Q_EHIP, hipGetErrorString(e)); break; }
        if (g > gcap) { rc = fail(c, F2Q_ENOMEM, "synthetic general-path capacity exceeded"); break; }
        b->n_general = g;
        if (fast) {
            b->pb.n_slots = n_slots; b->pb.n_tiles = (uint32_t)n_tiles; b->pb.wb = o.wb; b->pb.wq = o.wq; b->pb.rmax = (uint32_t)R;
            b->pb.planar_nw = o.planar_nw;
            b->pb.bases = o.bases; b->pb.qual = o.qual; b->pb.len = o.len;
        }
        b->rb.n = g; b->rb.raw = o.raw; b->rb.off = o.off; b->rb.len = o.glen; b->rb.qlen = o.gqlen; b->rb.index = o.gindex;
        b->rb.first_index = 0;
        b->dev_bytes += g * (uint64_t)(2 * R + 20);
    } while (0);
    if (rc) { free_all(b->allocs); delete b; return rc; }
    *out = b;
    return F2Q_OK;
}

// ---- Extract+Count results ----------------------------------------------------------------------
// both Extract+Count tables, pulled to the host: the byte-string entries then the occupied single-word slots
struct EcHost {
    std::vector<std::string> keys;
    std::vector<unsigned long long> cnt, first;
};
static int ec_pull(f2q_ctx *c, EcHost &h)
{
    if (!c->ec.slots) return F2Q_OK;
    unsigned long long ctr[4];
    HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error)");
    const uint64_t n = ctr[0];
    if (n) {
        std::vector<uint32_t> len(n), arena(ctr[1] ? ctr[1] : 1);
        std::vector<unsigned long long> off(n), cnt(n), first(n);
        HIPC(c, hipMemcpy(len.data(), c->ec.ent_len, n * 4, hipMemcpyDeviceToHost));
        HIPC(c, hipMemcpy(off.data(), c->ec.ent_off, n * 8, hipMemcpyDeviceToHost));
        HIPC(c, hipMemcpy(cnt.data(), c->ec.ent_count, n * 8, hipMemcpyDeviceToHost));
        HIPC(c, hipMemcpy(first.data(), c->ec.ent_first, n * 8, hipMemcpyDeviceToHost));
        if (ctr[1]) HIPC(c, hipMemcpy(arena.data(), c->ec.arena, ctr[1] * 4, hipMemcpyDeviceToHost));
        for (uint64_t e = 0; e < n; e++) {
            h.keys.emplace_back((const char *)(arena.data() + off[e]), len[e]);
            h.cnt.push_back(cnt[e]); h.first.push_back(first[e]);
        }
    }
    if (ctr[3]) {
        const size_t ns = (size_t)c->ec.k64_mask + 1;
        std::vector<unsigned long long> ks(ns), kc(ns), kf(ns);
        HIPC(c, hipMemcpy(ks.data(), c->ec.k64_slots, ns * 8, hipMemcpyDeviceToHost));
        HIPC(c, hipMemcpy(kc.data(), c->ec.k64_count, ns * 8, hipMemcpyDeviceToHost));
        HIPC(c, hipMemcpy(kf.data(), c->ec.k64_first, ns * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ns; i++) {
            if (ks[i] == KEY_EMPTY) continue;
            const uint32_t len = (uint32_t)(ks[i] >> 58);
            std::string k(len, 'A');
            for (uint32_t j = 0; j < len; j++) k[j] = "ACGT"[(ks[i] >> (2 * j)) & 3];
            h.keys.push_back(k); h.cnt.push_back(kc[i]); h.first.push_back(kf[i]);
        }
    }
    return F2Q_OK;
}

extern "C" int f2q_ec_size(f2q_ctx *c, uint64_t *n_keys, uint64_t *n_bytes)
{
    if (!c || !n_keys || !n_bytes) return F2Q_EINVAL;
    *n_keys = 0; *n_bytes = 0;
    if (c->prm.mode != 1) return fail(c, F2Q_ESTATE, "not in Extract+Count mode");
    HIPC(c, hipSetDevice(c->device));
    EcHost h; int rc = ec_pull(c, h);
    if (rc) return rc;
    uint64_t nb = 0; for (auto &k : h.keys) nb += k.size();
    *n_keys = h.keys.size(); *n_bytes = nb;
    return F2Q_OK;
}

extern "C" int f2q_ec_fetch(f2q_ctx *c, char *keys, uint64_t *offs, int64_t *counts, uint64_t *first_read)
{
    if (!c || !offs) return F2Q_EINVAL;
    if (c->prm.mode != 1) return fail(c, F2Q_ESTATE, "not in Extract+Count mode");
    offs[0] = 0;
    HIPC(c, hipSetDevice(c->device));
    EcHost h; int rc = ec_pull(c, h);
    if (rc) return rc;
    uint64_t o = 0;
    for (size_t e = 0; e < h.keys.size(); e++) {
        if (keys) memcpy(keys + o, h.keys[e].data(), h.keys[e].size());
        o += h.keys[e].size(); offs[e + 1] = o;
        if (counts) counts[e] = (int64_t)h.cnt[e];
        if (first_read) first_read[e] = h.first[e];
    }
    return F2Q_OK;
}
