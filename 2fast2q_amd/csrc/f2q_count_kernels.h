// f2q_count_kernels.h -- the counting kernels of libf2q_hip.so (included by f2q_lib.hip only).
//   k_count_fixed4    fixed-offset Counter mode, 4 reads per lane on interleaved tiles (the bench kernel)
//   k_extract_fixed4  fixed-window Extract+Count
//   k_count_anchor    --us/--ds runs on bit-plane tiles (Counter and Extract+Count)
//   k_count_fixed     one read per lane (wide tables / A-B runs)
//   k_count_general   byte-exact routine on raw records
//   k_hist_ranges, k_reduce_slabs   histogram / counter reduction
#pragma once

// ===============================================================================================
// kernels
// ===============================================================================================
#define F2Q_HIST_MAX 24576u     // features whose u32 histogram fits the workgroup's LDS budget (96 KiB)
#define F2Q_HIST_RANGE 38912u   // features per pass of k_hist_ranges, which has a CU's LDS to itself (152 KiB)

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

__device__ __forceinline__ int slow_read(const RunDev *run, const LibDev *lib, const PackedBlock *pb, uint32_t tile,
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
    // the tiles hold every row the window touches (false only when all reads of the block end before it does: the
    // row clamp below would then re-read a row, so such blocks keep the byte-exact routine for their short reads)
    const bool rows_ok = (uint32_t)(g.qw0 + g.nq) <= pb.wq && (uint32_t)(g.bw0 + g.nb) <= pb.wb &&
                         g.L >= 1 && g.L <= F2Q_REG_MAXLEN && lib.grp[g.L].n == nf;      // and every feature is L long
    const bool do_near = run.miss > 0;
    const PackedPiece ex = lib.pk.exact;
    const uint32_t ib = lib.pk.ib, exm = (1u << ex.bits) - 1u;
    const uint64_t imask = (1ull << ib) - 1ull;
    const auto ptab = gp(lib.ptab);
    uint32_t st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;      // per lane and launch: far below 2^32
#ifdef F2Q_STAMP
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, t0_ = 0, t1_;
#define STAMP4(i) do { __builtin_amdgcn_sched_barrier(0); t1_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); tp[i] += t1_ - t0_; t0_ = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP4(i) do {} while (0)
#endif

    // a hit: LDS histogram, or (large library) the feature index is stored at the read's slot for k_hist_ranges.  The
    // exact hits of a lane's four reads leave as ONE 16-byte store that also clears the slots of the other reads (no
    // memset of hit_buf); a hit found later by the near search overwrites its word.
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
        uint32_t hid[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};       // !USE_LDS: feature of each read, so far
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
            // the quality rows are fetched even when the Phred rule is off (--ph <= 1): bit 7 of their bytes flags the
            // non-ACGT symbols, and sending every flagged read down the byte-exact path instead cost 2x with 0.5 % of them
#pragma unroll
            for (int r = 0; r < QR; r++) {
                uint32_t row = (uint32_t)g.qw0 + (uint32_t)(NQ ? r : (r < g.nq ? r : (g.nq > 0 ? g.nq - 1 : 0)));
                row = row < pb.wq ? row : pb.wq - 1u;
                qrow[r] = ld_u4<NT>(qp + (uint64_t)row * F2Q_TILE);
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
                if (l == F2Q_LEN_SKIP) res[j] = R_SKIP;
                else if (g.L < 1) res[j] = R_SLOW;
                // a read that ends inside the window gives a clipped, shorter key (:354).  With rows_ok every feature
                // is L long, so the key can equal or approach none (:683), and its bytes past the end are stored as 0
                // and never fail, so bad[] already is the Phred test of the clipped window (:355-357); otherwise the
                // byte-exact routine decides.  (Written as one test per line on purpose: hipcc 7.2 turned the
                // equivalent `L < 1 || (short && !rows_ok)` form into code that dropped reads — fuzz case 101.)
                else if ((int)(l & F2Q_LEN_MASK) < need) res[j] = rows_ok ? (bad[j] ? R_QFAIL : R_NONALIGNED) : R_SLOW;
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
            for (int j = 0; j < 4; j++) { pend[j] = (res[j] == R_NEAR); s[j] = packed_start(key[j], ex.bits); }
            while (pend[0] | pend[1] | pend[2] | pend[3]) {
                uint64_t v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (pend[j]) { const Slot2 pr = packed_pair(ptab, ex.off + s[j]); v0[j] = pr.a; v1[j] = pr.b; }   // one 16-byte load
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (!pend[j]) continue;
                    if (v0[j] == KEY_EMPTY) pend[j] = false;
                    else if ((v0[j] >> ib) == key[j]) { pend[j] = false; res[j] = R_PERFECT; if (USE_LDS) count_hit((uint32_t)(v0[j] & imask), 0u); else hid[j] = (uint32_t)(v0[j] & imask); }
                    else if (v1[j] == KEY_EMPTY) pend[j] = false;
                    else if ((v1[j] >> ib) == key[j]) { pend[j] = false; res[j] = R_PERFECT; if (USE_LDS) count_hit((uint32_t)(v1[j] & imask), 0u); else hid[j] = (uint32_t)(v1[j] & imask); }
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
                    if (r1 == 1 || r1 == 2) { if (USE_LDS) count_hit(idx, 0u); else hid[j] = idx; }
                    res[j] = r1;
                } else if (res[j] == R_NEAR) {
                    if (do_near) npush++; else res[j] = R_NONALIGNED;
                } else if (res[j] == R_FORCED) npush++;
                st0 += (res[j] != R_SKIP); st1 += (res[j] == R_PERFECT); st2 += (res[j] == R_IMPERFECT);
                st3 += (res[j] == R_NONALIGNED); st4 += (res[j] == R_QFAIL);
            }
            if (!USE_LDS) {
                // Ordering: this 16-byte store also writes 0xFFFFFFFF into the words of the reads whose exact probe missed;
                // the near search further down (drain) patches such a word with a 4-byte store issued by ANOTHER lane of
                // this same wave.  Both stores belong to one wave's vector-memory queue, which the memory pipeline
                // serves in issue order for accesses to the same address (a wave's global stores are not reordered among
                // themselves), and the patch is always issued later in program order (the ring is drained after the
                // tile's push), so the patch lands on top.  Pinned by test_hit_buf_patch_after_wide_store (most hits at
                // distance 1, a library beyond the LDS histogram, against the oracle).
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                v4 hv; hv.x = hid[0]; hv.y = hid[1]; hv.z = hid[2]; hv.w = hid[3];
                *reinterpret_cast<v4 F2Q_GLOBAL *>(gpw(acc.hit_buf) + (uint64_t)tile * F2Q_TILE + 4u * lane) = hv;
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

// ---- multi-window runs (--st a,b[,c,d]) on the packed path -----------------------------------------------------
// The tile walk of k_count_fixed4 with the windows taken one after another: for each window the rows under it, the
// Phred verdict of the four reads of the lane and the 2-bit part; a read's key is the concatenation of the parts that
// passed, in window order, and is looked up among the features with that many parts (LibDev::mpk[k - 1]; the ':'
// between the parts is implied by k, fast2q.py:349-363).  No part passed = quality failed (:389-390).  Exact probes per
// part count k, misses queued per wave (the key's two top bits carry k - 1) and searched 64 at a time; reads that end
// inside a window take the byte-exact routine on a copy rebuilt from the tile.
__device__ __noinline__ void multi_slow(const RunDev *run, const LibDev *lib, const Accum *acc, const PackedBlock *pb,
                                        uint32_t tile, uint32_t slot, int r, unsigned long long *st)
{
    uint8_t seq[F2Q_ANCHOR_MAXLEN], qual[F2Q_ANCHOR_MAXLEN];
    const auto bp = gp(pb->bases) + (uint64_t)tile * pb->wb * F2Q_TILE + slot;
    const auto qp = gp(pb->qual) + (uint64_t)tile * pb->wq * F2Q_TILE + slot;
    if (r > F2Q_ANCHOR_MAXLEN) r = F2Q_ANCHOR_MAXLEN;
    for (int i = 0; i < r; i++) {
        seq[i] = (uint8_t)"ACGT"[(bp[(uint64_t)(i >> 4) * F2Q_TILE] >> (2 * (i & 15))) & 3u];
        qual[i] = (uint8_t)((qp[(uint64_t)(i >> 2) * F2Q_TILE] >> (8 * (i & 3))) & 0xFFu);
        if (qual[i] & 0x80u) { seq[i] = (uint8_t)'N'; qual[i] &= 0x7Fu; }      // flagged: a symbol that equals nothing
    }
    const EcDev none{};
    general_read<const uint8_t *>(*run, *lib, none, *acc, seq, r, qual, r, 0ull, st);
}

template <bool USE_LDS>
__global__ __launch_bounds__(F2Q_V2_THREADS, 4) void k_count_multi4(const RunDev *__restrict__ runp,
                                                                  const LibDev *__restrict__ libp, PackedBlock pb,
                                                                  Accum acc, int need)
{
    extern __shared__ unsigned long long smem64[];
    unsigned long long *queue = smem64 + (threadIdx.x >> 6) * F2Q_V2_QCAP;                 // keys | (k - 1) << 62
    uint32_t *qforced = reinterpret_cast<uint32_t *>(smem64 + F2Q_V2_WAVES * F2Q_V2_QCAP) + (threadIdx.x >> 6) * F2Q_V2_QCAP;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem64 + F2Q_V2_WAVES * F2Q_V2_QCAP) + F2Q_V2_WAVES * F2Q_V2_QCAP;  // USE_LDS
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    const uint32_t nf = lib.n_features;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (USE_LDS) for (uint32_t i = tid; i < nf; i += F2Q_V2_THREADS) hist[i] = 0;
    __syncthreads();
    uint32_t q_head = 0, q_tail = 0;
    const int W = run.n_iter, L = run.length;
    const bool do_near = run.miss > 0;
    const auto ptab = gp(lib.ptab);
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    auto count_hit = [&](uint32_t idx) {
        if (USE_LDS) atomicAdd(&hist[idx], 1u);
        else acc_add(&acc.counts[idx], 1ull);
    };
    auto drain = [&](bool all) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t tail = q_tail;
        while (all ? (tail != q_head) : (tail - q_head >= 64u)) {
            if (lane < tail - q_head) {
                const unsigned long long e = queue[(q_head + lane) % F2Q_V2_QCAP];
                uint32_t idx = 0;
                const int r = packed_near_decide(run, lib, lib.mpk[(uint32_t)(e >> 62)], e & ((1ull << 62) - 1ull),
                                                 qforced[(q_head + lane) % F2Q_V2_QCAP], idx);
                if (r == R_IMPERFECT || r == R_PERFECT) { count_hit(idx); st[2]++; } else st[3]++;
            }
            q_head += (tail - q_head < 64u) ? (tail - q_head) : 64u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    for (uint32_t base = blockIdx.x * F2Q_V2_WAVES; base < pb.n_tiles; base += gridDim.x * F2Q_V2_WAVES) {
        const uint32_t tile = base + wave;
        if (tile < pb.n_tiles) {
            int res[4]; uint64_t key[4] = {0, 0, 0, 0}; uint32_t forced[4] = {0, 0, 0, 0}, npart[4] = {0, 0, 0, 0}, lv[4];
            {
                typedef uint32_t v2 __attribute__((ext_vector_type(2)));
                v2 len2 = {pb.rmax | (pb.rmax << 16), pb.rmax | (pb.rmax << 16)};
                if (pb.len) len2 = *(const v2 F2Q_GLOBAL *)(gp(pb.len) + (uint64_t)tile * F2Q_TILE + 4u * lane);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    lv[j] = ((j < 2 ? len2.x : len2.y) >> (16 * (j & 1))) & 0xFFFFu;
                    res[j] = lv[j] == F2Q_LEN_SKIP ? R_SKIP : ((int)(lv[j] & F2Q_LEN_MASK) < need) ? R_SLOW : R_NEAR;
                }
            }
            const auto qp = gp(pb.qual) + (uint64_t)tile * pb.wq * F2Q_TILE + 4u * lane;
            const auto bp = gp(pb.bases) + (uint64_t)tile * pb.wb * F2Q_TILE + 4u * lane;
            for (int w = 0; w < W; w++) {
                const FixedGeom g = fixed_geom_of(run, w);
                U4 brow[F2Q_MAXBROWS], qrow[F2Q_MAXQROWS];
#pragma unroll
                for (int r = 0; r < F2Q_MAXBROWS; r++) {
                    uint32_t row = (uint32_t)g.bw0 + (uint32_t)(r < g.nb ? r : g.nb - 1);
                    row = row < pb.wb ? row : pb.wb - 1u;       // blocks whose reads all end before this window
                    brow[r] = ld_u4<true>(bp + (uint64_t)row * F2Q_TILE);
                }
#pragma unroll
                for (int r = 0; r < F2Q_MAXQROWS; r++) {
                    uint32_t row = (uint32_t)g.qw0 + (uint32_t)(r < g.nq ? r : g.nq - 1);
                    row = row < pb.wq ? row : pb.wq - 1u;
                    qrow[r] = ld_u4<true>(qp + (uint64_t)row * F2Q_TILE);
                }
                uint32_t bad[4] = {0, 0, 0, 0};
                if (g.add_hi) {
#pragma unroll
                    for (int r = 0; r < F2Q_MAXQROWS; r++)
                        if (r < g.nq) fixed4_qrow(g, r, qrow[r], bad);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (res[j] != R_NEAR || bad[j]) continue;          // this part fails its Phred test: omitted (:357-360)
                    key[j] |= fixed4_key(g, brow, j) << (2u * (uint32_t)L * npart[j]);
                    if (lv[j] & F2Q_LEN_FLAG) forced[j] |= fixed4_flags(g, qrow, j) << ((uint32_t)L * npart[j]);
                    npart[j]++;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (res[j] == R_SLOW) {
                    unsigned long long st2[5] = {0, 0, 0, 0, 0};
                    const Accum acc2 = acc; const PackedBlock pb2 = pb;
                    multi_slow(runp, libp, &acc2, &pb2, tile, 4u * lane + (uint32_t)j, (int)(lv[j] & F2Q_LEN_MASK), st2);
#pragma unroll
                    for (int k = 0; k < 5; k++) st[k] += st2[k];
                    res[j] = R_SKIP;                                   // counted by the routine itself
                } else if (res[j] == R_NEAR) {
                    st[0]++;
                    if (npart[j] == 0u) { res[j] = R_QFAIL; st[4]++; }
                    else if (forced[j]) res[j] = (!do_near || __popc(forced[j]) > run.miss) ? R_NONALIGNED : R_FORCED;
                }
            }
            // exact probes, one part count at a time (almost every read has all W parts)
            for (int k = W; k >= 1; k--) {
                bool mine[4]; bool any = false;
#pragma unroll
                for (int j = 0; j < 4; j++) { mine[j] = res[j] == R_NEAR && npart[j] == (uint32_t)k; any |= mine[j]; }
                if (__ballot(any) == 0ull) continue;
                const PackedGroup &pk = lib.mpk[k - 1];
                const PackedPiece ex = pk.exact;
                const uint32_t ib = pk.ib, exm = (1u << ex.bits) - 1u;
                const uint64_t imask = (1ull << ib) - 1ull;
                if (pk.len == 0u) continue;                           // no feature has k parts: nothing to hit (misses stay R_NEAR)
                uint32_t s[4];
#pragma unroll
                for (int j = 0; j < 4; j++) s[j] = packed_start(key[j], ex.bits);
                while (mine[0] | mine[1] | mine[2] | mine[3]) {
                    uint64_t v0[4], v1[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (mine[j]) { const Slot2 pr = packed_pair(ptab, ex.off + s[j]); v0[j] = pr.a; v1[j] = pr.b; }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (!mine[j]) continue;
                        if (v0[j] == KEY_EMPTY) mine[j] = false;
                        else if ((v0[j] >> ib) == key[j]) { mine[j] = false; res[j] = R_PERFECT; count_hit((uint32_t)(v0[j] & imask)); }
                        else if (v1[j] == KEY_EMPTY) mine[j] = false;
                        else if ((v1[j] >> ib) == key[j]) { mine[j] = false; res[j] = R_PERFECT; count_hit((uint32_t)(v1[j] & imask)); }
                        else s[j] = (s[j] + 2u) & exm;
                    }
                }
            }
            uint32_t npush = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (res[j] == R_NEAR) { if (do_near) npush++; else res[j] = R_NONALIGNED; }
                else if (res[j] == R_FORCED) npush++;
                st[1] += (res[j] == R_PERFECT); st[3] += (res[j] == R_NONALIGNED);
            }
            {
                const unsigned long long b0 = __ballot(npush & 1u), b1 = __ballot(npush & 2u), b2 = __ballot(npush & 4u);
                uint32_t at = q_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u))
                              + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u))
                              + 4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
                q_tail += (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (res[j] == R_NEAR || res[j] == R_FORCED) {
                        queue[at % F2Q_V2_QCAP] = key[j] | ((unsigned long long)(npart[j] - 1u) << 62);
                        qforced[at % F2Q_V2_QCAP] = forced[j];
                        at++;
                    }
            }
        }
        if (do_near) drain(false);
    }
    if (do_near) drain(true);
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, acc.stat_slab ? acc.stat_slab + (uint64_t)blockIdx.x * 8u : nullptr);
    if (USE_LDS) {
        __syncthreads();
        auto row = gpw(acc.slab) + (uint64_t)blockIdx.x * nf;
        for (uint32_t i = tid; i < nf; i += F2Q_V2_THREADS) row[i] = hist[i];
    }
}

// ---- fast path v3: the library in LDS ------------------------------------------------------------------
// Fixed-offset Counter mode for uniform libraries of 14..21-base features searched with --m <= 1 (the reference's
// default, and BASELINE configs 2 and 3): the tile walk of k_count_fixed4, but every lookup -- exact hit and the
// unique-feature-at-distance-1 search -- is answered from two cuckoo tables of 32-bit tags held in the workgroup's LDS
// (f2q_device.h, "LDS tables"): four 8-byte LDS reads per read, no table traffic to L2 at all, no ring, no second pass.
// One 1024-thread workgroup per CU owns the whole 160 KiB: 2 x 64 KiB of tags + a 32 KiB histogram of u16 counters
// (two per word, indexed by table-0 slot).  A counter that reaches 0x8000 moves 0x8000 counts to the global vector
// (lt_count), so no count is lost however skewed the library's popularity is.  The rows of the wave's next tile are
// requested before the current tile is decided (one tile in flight per wave hides the HBM latency).
#define F2Q_LT_THREADS 1024
#define F2Q_LT_WAVES (F2Q_LT_THREADS / 64)

__device__ __forceinline__ U2 lds_u2(const uint32_t *p)
{
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    const v2 v = *reinterpret_cast<const v2 *>(p);              // ds_read_b64
    return U2{v.x, v.y};
}

__device__ __forceinline__ void lt_count(uint32_t *cnt, uint32_t slot, const Accum &acc, const LtDesc &lt)
{
    const uint32_t sh = (slot & 1u) << 4;
    const uint32_t old = atomicAdd(&cnt[slot >> 1], 1u << sh);
    // exactly one adder sees the counter pass 0x7FFF -> 0x8000; it takes 0x8000 out again (no borrow: the counter only
    // grows until then, and by far less than another 0x8000) and credits the feature's global counter
    if (((old >> sh) & 0xFFFFu) == 0x7FFFu) {
        atomicSub(&cnt[slot >> 1], 0x8000u << sh);
        acc_add(&acc.counts[gp(lt.feat_of)[slot]], 0x8000ull);
    }
}

// A20: the window is 20 bases starting at a multiple of 16 (--l 20 with --st 0, 16, ...: the usual guide-counting run):
// the key needs no shift, every quality row is tested whole, and the table geometry (two 10-base halves) is constant.
// MW: a multi-window run (--st a,b[,c,d]) on tiles that hold the windows back to back (PackPlan::n_win) against a library
// whose features all have one part per window: the compact window is one window of n_iter * length bases, except that
// the Phred rule is applied part by part -- a read with a failed part has a key of fewer parts, which equals or
// approaches no feature; only a read whose parts all fail counts as failed (fast2q.py:357-363).
template <int NQ, int NB, bool NEAR, bool A20, bool MW = false>
__global__ __launch_bounds__(F2Q_LT_THREADS) void k_count_fixed4_lds(const RunDev *__restrict__ runp,
                                                                     const LibDev *__restrict__ libp, PackedBlock pb,
                                                                     Accum acc)
{
    extern __shared__ uint32_t lt_smem[];
    constexpr uint32_t NT = NEAR ? 2u : 1u;
    uint32_t *tg = lt_smem;                                     // [NT][F2Q_LT_SLOTS] tags
    uint32_t *cnt = lt_smem + NT * F2Q_LT_SLOTS;                // [F2Q_LT_BUCKETS] two u16 counters per word
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    LtDesc lt = lib.lt;
    if (A20) { lt.hb0 = 20u; lt.hb1 = 20u; lt.len = 20u; }      // constants for the compiler
    const uint32_t nf = lib.n_features;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 F2Q_GLOBAL *src = (const v4 F2Q_GLOBAL *)gp(lt.tags);
        v4 *dst = reinterpret_cast<v4 *>(tg);
        for (uint32_t i = tid; i < NT * F2Q_LT_SLOTS / 4u; i += F2Q_LT_THREADS) dst[i] = src[i];
        for (uint32_t i = tid; i < F2Q_LT_BUCKETS; i += F2Q_LT_THREADS) cnt[i] = 0;
    }
    __syncthreads();
    FixedGeom g = MW ? fixed_geom_at(0, run.n_iter * run.length, run.thr) : fixed_geom(run);
    if (A20) { g.L = 20; g.nq = 5; g.nb = 2; g.sh = 0; g.qm_first = 0x80808080u; g.qm_last = 0x80808080u; g.kmask = (1ull << 40) - 1ull; }
    const int need = g.st + g.L;
    // the five reference counters of this wave, kept in scalar registers: the per-read verdicts are lane masks already
    uint32_t w_reads = 0, w_perfect = 0, w_imperfect = 0, w_qfail = 0;
    constexpr int QR = NQ ? NQ : F2Q_MAXQROWS, BR = NB ? NB : F2Q_MAXBROWS;
    constexpr bool PIPE = NQ != 0;                              // run-time geometry: 12 rows, too many to keep two tiles of

    struct Rows { U4 b[BR], q[QR]; uint32_t len01, len23; };
    // All loads of a tile are unconditional and their number is fixed (the host guarantees the rows and the length plane
    // exist): the compiler's s_waitcnt bookkeeping is then exact, and waiting for this tile's rows does not also wait
    // for the next tile's, which were requested after them.  One 64-bit address per plane, the rows at immediate offsets.
    const uint64_t q_stride = (uint64_t)pb.wq * F2Q_TILE, b_stride = (uint64_t)pb.wb * F2Q_TILE;
    const auto q_base = gp(pb.qual) + (uint64_t)g.qw0 * F2Q_TILE + 4u * lane;
    const auto b_base = gp(pb.bases) + (uint64_t)g.bw0 * F2Q_TILE + 4u * lane;
    const auto l_base = gp(pb.len) + 4u * lane;
    auto request_tile = [&](Rows &r, uint32_t t) {
        const auto qp = q_base + (uint64_t)t * q_stride;
        const auto bp = b_base + (uint64_t)t * b_stride;
#pragma unroll
        for (int i = 0; i < BR; i++) r.b[i] = ld_u4<true>(bp + (NB ? i : (i < g.nb ? i : g.nb - 1)) * F2Q_TILE);
#pragma unroll
        for (int i = 0; i < QR; i++) r.q[i] = ld_u4<true>(qp + (NQ ? i : (i < g.nq ? i : g.nq - 1)) * F2Q_TILE);
        typedef uint32_t v2 __attribute__((ext_vector_type(2)));
        const v2 lv = __builtin_nontemporal_load((const v2 F2Q_GLOBAL *)(l_base + (uint64_t)t * F2Q_TILE));
        r.len01 = lv.x; r.len23 = lv.y;
    };
    auto decide_tile = [&](const Rows &r) {
        uint32_t bad[4] = {0, 0, 0, 0};
        bool part_fail[4] = {false, false, false, false};             // MW: some part, but not every part, fails its Phred test
        if (MW && g.add_hi) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int pv = mw_part_verdict(fixed4_failbits(g, r.q, j), run.n_iter, run.length);
                bad[j] = pv == 2 ? 1u : 0u; part_fail[j] = pv == 1;
            }
        } else if (g.add_hi) {
            if (A20) {
                // every byte of the five rows is under the window: OR the rows' verdicts, mask once
#pragma unroll
                for (int i = 0; i < QR; i++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t w = u4get(r.q[i], j) & 0x7F7F7F7Fu;       // bit 7 = non-ACGT flag, not quality
                        bad[j] |= (w + g.add_lo) & ~(w + g.add_hi);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) bad[j] &= 0x80808080u;
            } else {
#pragma unroll
                for (int i = 0; i < QR; i++)
                    if (NQ || i < g.nq) fixed4_qrow(g, i, r.q[i], bad);
            }
        }
        // two reads at a time: both reads' eight bucket reads are in flight together, and nothing below branches except the
        // rare cases (flagged symbols, features sharing a half, a hit found through table 1, a counter passing 0x8000)
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
            LtProbe q[2]; U2 e[2][4]; uint32_t forced[2]; bool cand[2];
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int j = jp + a;
                const uint32_t l = ((j < 2 ? r.len01 : r.len23) >> (16 * (j & 1))) & 0xFFFFu;
                const bool live = l != F2Q_LEN_SKIP, qf = live && bad[j] != 0u;
                // a read that ends inside the window gives a shorter key (:354); every feature is L long, so it can equal
                // or approach none (:683); its bytes past the end are stored as 0 and never fail the Phred test
                cand[a] = live && !qf && (int)(l & F2Q_LEN_MASK) >= need && !(MW && part_fail[j]);
                w_reads += (uint32_t)__popcll(__ballot(live));
                w_qfail += (uint32_t)__popcll(__ballot(qf));
                forced[a] = 0;
                if ((l & F2Q_LEN_FLAG) && cand[a]) forced[a] = fixed4_flags(g, r.q, j);   // non-ACGT symbols in the window (rare)
                uint64_t key = fixed4_key(g, r.b, j);
                if (MW && lt.mix) { key = mw_mix(key, lt.mix); if (forced[a]) forced[a] = mw_mix_mask(forced[a], lt.mix); }
                q[a] = lt_probe(lt, key);
#pragma unroll
                for (int k = 0; k < (NEAR ? 4 : 2); k++) e[a][k] = lds_u2(tg + (uint32_t)(k >> 1) * F2Q_LT_SLOTS + 2u * q[a].b[k]);
                if (!NEAR) { e[a][2] = U2{F2Q_LT_EMPTY, F2Q_LT_EMPTY}; e[a][3] = e[a][2]; }
            }
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const LtVerdict v = lt_decide<NEAR>(lt, q[a], e[a], forced[a], [&](uint32_t bk) { return lds_u2(tg + 2u * bk); });
                const LtPred cm = LT_P(cand[a]), perfect = v.perfect & cm, imperfect = v.imperfect & cm;
                if (LT_TRUE(perfect | imperfect)) lt_count(cnt, v.slot, acc, lt);
                w_perfect += (uint32_t)__popcll(perfect);
                w_imperfect += (uint32_t)__popcll(imperfect);
            }
        }
    };

    const uint32_t stride = gridDim.x * F2Q_LT_WAVES;
    uint32_t tile = blockIdx.x * F2Q_LT_WAVES + wave;
    const uint32_t last = pb.n_tiles - 1u;
    if (PIPE) {
        // two register sets of rows, used alternately: the next tile's rows travel while this tile is decided (past the
        // end the last tile is requested again and dropped: an unconditional request keeps the wait counts exact)
        Rows ra, rb;
        if (tile < pb.n_tiles) {
            request_tile(ra, tile);
            for (;;) {
                __builtin_amdgcn_sched_barrier(0);
                request_tile(rb, min(tile + stride, last));
                __builtin_amdgcn_sched_barrier(0);
                decide_tile(ra);
                tile += stride;
                if (tile >= pb.n_tiles) break;
                __builtin_amdgcn_sched_barrier(0);
                request_tile(ra, min(tile + stride, last));
                __builtin_amdgcn_sched_barrier(0);
                decide_tile(rb);
                tile += stride;
                if (tile >= pb.n_tiles) break;
            }
        }
    } else {
        for (; tile < pb.n_tiles; tile += stride) { Rows r; request_tile(r, tile); decide_tile(r); }
    }
    __syncthreads();
    // the histogram leaves as one slab row in feature order; k_reduce_slabs sums the rows
    {
        auto row = gpw(acc.slab) + (uint64_t)blockIdx.x * nf;
        for (uint32_t i = tid; i < nf; i += F2Q_LT_THREADS) {
            const uint32_t s = gp(lt.slot_of)[i];
            row[i] = (cnt[s >> 1] >> ((s & 1u) << 4)) & 0xFFFFu;
        }
    }
    __syncthreads();
    // wave totals -> one stats row per workgroup (lane 0 of every wave carries its wave's counters)
    const bool l0 = lane == 0;
    unsigned long long stv[5] = {l0 ? w_reads : 0u, l0 ? w_perfect : 0u, l0 ? w_imperfect : 0u,
                                 l0 ? w_reads - w_perfect - w_imperfect - w_qfail : 0u, l0 ? w_qfail : 0u};
    flush_stats(acc, stv, reinterpret_cast<unsigned long long *>(lt_smem), acc.stat_slab + (uint64_t)blockIdx.x * 8u);   // the tables are done with
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
    uint32_t n_new = 0;                       // keys this lane put into the table
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
            // (the rows are fetched with the Phred rule off too: bit 7 of their bytes flags the window's 'N's)
            qrow[r] = want < pb.wq ? ld_u4<true>(qp + (uint64_t)row * F2Q_TILE) : U4{0, 0, 0, 0};
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
            const uint64_t slot = (uint64_t)tile * F2Q_TILE + 4u * lane + (uint32_t)j;
            n_new += ec64_insert_word(ec, fixed4_ec_word(g, brow, qrow, j, l), read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot));
            st[1]++;
        }
    }
    ec64_report_new(ec, n_new);
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
                                         unsigned long long read_index, unsigned long long *st, bool lower = false)
{   // lower: the read's marks are lower-case bases (F2Q_LEN_CASE)
    uint8_t seq[F2Q_ANCHOR_MAXLEN], qual[F2Q_ANCHOR_MAXLEN];
    const uint32_t nw = pb->planar_nw;
    const auto bp = gp(pb->bases) + (uint64_t)tile * pb->wb * F2Q_TILE + slot;
    const auto qp = gp(pb->qual) + (uint64_t)tile * pb->wq * F2Q_TILE + slot;
    if (r > F2Q_ANCHOR_MAXLEN) r = F2Q_ANCHOR_MAXLEN;
    for (int i = 0; i < r; i++) {
        const uint32_t lo = bp[(uint64_t)(i >> 5) * F2Q_TILE], hi = bp[(uint64_t)(nw + (i >> 5)) * F2Q_TILE];
        seq[i] = (uint8_t)"ACGT"[((lo >> (i & 31)) & 1u) | (((hi >> (i & 31)) & 1u) << 1)];
        qual[i] = (uint8_t)((qp[(uint64_t)planar_qword((uint32_t)i) * F2Q_TILE] >> (8 * planar_qbyte((uint32_t)i))) & 0xFFu);
        if (qual[i] & 0x80u) { seq[i] = lower ? (uint8_t)(seq[i] | 0x20u) : (uint8_t)'N'; qual[i] &= 0x7Fu; }      // marked: a lower-case base, or a symbol that equals nothing
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
    uint32_t n_new = 0;                       // Extract+Count: keys this lane put into the single-word table
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
        if (__ballot(flagged) == 0ull) {              // the usual tile: no flag bits to strip, no flag planes to build
#pragma unroll
            for (int cw = 0; cw < NW; cw++) {
                uint32_t q8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q8[i] = Q[8 * cw + i];
                FW[cw] = fail_word8<false>(q8, ah_w);
                if (!SAMEQ) { FU[cw] = fail_word8<false>(q8, ah_u); FD[cw] = fail_word8<false>(q8, ah_d); }
                FLG[cw] = 0u;
            }
        } else {
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
        const bool keyflag = flagged && !(l & F2Q_LEN_CASE);      // the marks count for the key too (lower-case bases: only for the anchors)
        __builtin_amdgcn_sched_barrier(0);
        STAMP(0);                                   // loads + fail vectors
        bool push = false; uint64_t push_key = 0; uint32_t push_forced = 0;
        if (l != F2Q_LEN_SKIP) {
            const int r = (int)(l & F2Q_LEN_MASK);
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
            } else if (aw.ok == 1 && EC && (L > F2Q_EC64_MAXLEN || (keyflag && L > 0 && !ec64_fits(plane_extract<NW>(FLG, aw.start, L), L)))) {
                // Extract+Count key the single-word table cannot hold (too long, or it spells an 'N'): decode the window
                // and use the byte-string table
                uint8_t kb[32 * NW];
#pragma unroll
                for (int cw = 0; cw < NW; cw++) {
                    const int off = 32 * cw, n = L - off < 32 ? L - off : 32;
                    if (n > 0) {
                        const uint32_t lo = plane_extract<NW>(LO, aw.start + off, n), hi = plane_extract<NW>(HI, aw.start + off, n);
                        const uint32_t fl = keyflag ? plane_extract<NW>(FLG, aw.start + off, n) : 0u;
                        for (int j = 0; j < n; j++) kb[off + j] = ((fl >> j) & 1u) ? (uint8_t)'N' : (uint8_t)"ACGT"[((lo >> j) & 1u) | (((hi >> j) & 1u) << 1)];
                    }
                }
                KeyView kv; kv.seq = kb; kv.nseg = 1; kv.a[0] = 0; kv.b[0] = L; kv.len = L;
                ec_insert(ec, kv, gi);
                st[1]++; st[0]++;
            } else if (aw.ok == 2) {
                // negative-index slices (down-only anchor near the read start, negative --l): byte-exact routine
                unsigned long long st2[5] = {0, 0, 0, 0, 0};
                const EcDev ec2 = ec; const Accum acc2 = acc; const PackedBlock pb2 = pb;
                anchor_slow(runp, libp, &ec2, &acc2, &pb2, tile, tid, r, gi, st2, (l & F2Q_LEN_CASE) != 0u);
#pragma unroll
                for (int k = 0; k < 5; k++) st[k] += st2[k];
            } else if (keyflag && plane_extract<NW>(FLG, aw.start, L) != 0u) {
                // the window itself holds non-ACGT symbols
                st[0]++;
                const uint32_t forced = plane_extract<NW>(FLG, aw.start, L);
                if (EC) {                          // an 'N' in the window, and the key has a single-word form
                    unsigned long long w = 0;
                    ec64_word(plane_key<NW>(LO, HI, aw.start, L), forced, L, w);
                    n_new += ec64_insert_word(ec, w, gi); st[1]++;
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
                if (EC) { n_new += ec64_insert_n(ec, key, L, gi); st[1]++; }   // (no flagged base in the window)
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
    if (EC) ec64_report_new(ec, n_new);
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, acc.stat_slab ? acc.stat_slab + (uint64_t)blockIdx.x * 8u : nullptr);
    if (USE_LDS && !EC) {
        __syncthreads();
        auto row = gpw(acc.slab) + (uint64_t)blockIdx.x * nf;
        for (uint32_t i = tid; i < nf; i += F2Q_AN_THREADS) row[i] = hist[i];
    }
}

// ---- several --us/--ds pairs (pairs_lane, f2q_device.h) ---------------------------------------------------------------
// k_count_anchor's tile walk; every pair is searched on the planes the lane already holds, the joined key is matched as
// a string (byte-string index, or the 2-bit tables when neither key nor library holds a ':').  Counter mode counts in an
// LDS histogram when the library fits, Extract+Count inserts in place (the host reserves for every read of the view).
template <int NW, int KB, bool SAMEQ, bool USE_LDS>
__global__ __launch_bounds__(F2Q_AN_THREADS, 4) void k_count_anchor_pairs(const RunDev *__restrict__ runp, const LibDev *__restrict__ libp,
                                                                       EcDev ec, PackedBlock pb, Accum acc, uint64_t read_base)
{
    constexpr int NQW = 8 * NW;
    extern __shared__ uint32_t pairs_hist[];
    __shared__ __attribute__((aligned(4))) uint8_t pairs_keys[F2Q_AN_THREADS * F2Q_PAIRS_KEYMAX];
    uint8_t *kb = pairs_keys + threadIdx.x * F2Q_PAIRS_KEYMAX;
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    const uint32_t nf = lib.n_features, tid = threadIdx.x;
    // the histogram: two u16 counters per word, a counter that reaches 0x8000 hands 0x8000 counts to the global vector
    if (USE_LDS) for (uint32_t i = tid; i < (nf + 1u) / 2u; i += F2Q_AN_THREADS) pairs_hist[i] = 0;
    __syncthreads();
    const uint32_t ah_w = phred_add_hi(run.thr), ah_u = phred_add_hi(run.thr_up), ah_d = phred_add_hi(run.thr_down);
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    uint32_t n_new = 0;
    for (uint32_t tile = blockIdx.x; tile < pb.n_tiles; tile += gridDim.x) {
        const uint64_t slot = (uint64_t)tile * F2Q_TILE + tid;
        const uint32_t l = pb.len ? gp(pb.len)[slot] : pb.rmax;
        const auto bp = gp(pb.bases) + (uint64_t)tile * pb.wb * F2Q_TILE + tid;
        const auto qp = gp(pb.qual) + (uint64_t)tile * pb.wq * F2Q_TILE + tid;
        uint32_t LO[NW], HI[NW], Q[NQW];
#pragma unroll
        for (int w = 0; w < NW; w++) { LO[w] = __builtin_nontemporal_load(bp + (uint64_t)w * F2Q_TILE); HI[w] = __builtin_nontemporal_load(bp + (uint64_t)(NW + w) * F2Q_TILE); }
#pragma unroll
        for (int i = 0; i < NQW; i++) Q[i] = __builtin_nontemporal_load(qp + (uint64_t)i * F2Q_TILE);
        if (l == F2Q_LEN_SKIP) continue;
        const bool flagged = (l & F2Q_LEN_FLAG) != 0;
        uint32_t FW[NW], FU[SAMEQ ? 1 : NW], FD[SAMEQ ? 1 : NW], FLG[NW];
#pragma unroll
        for (int cw = 0; cw < NW; cw++) {
            uint32_t q8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q8[i] = Q[8 * cw + i];
            FW[cw] = fail_word8(q8, ah_w);
            if (!SAMEQ) { FU[cw] = fail_word8(q8, ah_u); FD[cw] = fail_word8(q8, ah_d); }
            FLG[cw] = flagged ? flag_word8(q8) : 0u;
        }
        const int r = (int)(l & F2Q_LEN_MASK);
        const unsigned long long gi = read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot);
        uint32_t idx = 0;
        int res;
        const bool keyflag = !(l & F2Q_LEN_CASE);                // (lower-case bases: marks for the anchors only)
        if constexpr (SAMEQ) res = pairs_lane<NW, KB>(run, lib, ec, kb, LO, HI, FLG, r, FW, FW, FW, gi, idx, &n_new, keyflag);
        else res = pairs_lane<NW, KB>(run, lib, ec, kb, LO, HI, FLG, r, FU, FD, FW, gi, idx, &n_new, keyflag);
        if (res < 0) {
            const EcDev ec2 = ec; const Accum acc2 = acc; const PackedBlock pb2 = pb;
            anchor_slow(runp, libp, &ec2, &acc2, &pb2, tile, tid, r, gi, st, !keyflag);
            continue;
        }
        st[0]++;
        if (res == 0) st[1]++;                                   // Extract+Count: the key went into a table (:387)
        else {
            st[res]++;
            if (res == 1 || res == 2) {
                if (USE_LDS) {
                    const uint32_t sh = (idx & 1u) << 4;
                    const uint32_t old = atomicAdd(&pairs_hist[idx >> 1], 1u << sh);
                    if (((old >> sh) & 0xFFFFu) == 0x7FFFu) { atomicSub(&pairs_hist[idx >> 1], 0x8000u << sh); acc_add(&acc.counts[idx], 0x8000ull); }
                } else acc_add(&acc.counts[idx], 1ull);
            }
        }
    }
    if (run.mode == 1) ec64_report_new(ec, n_new);
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, nullptr);
    if (USE_LDS) {
        __syncthreads();
        for (uint32_t i = tid; i < nf; i += F2Q_AN_THREADS) { const uint32_t n = (pairs_hist[i >> 1] >> ((i & 1u) << 4)) & 0xFFFFu; if (n) acc_add(&acc.counts[i], (unsigned long long)n); }
    }
}

// ---- anchored runs with the library in LDS ---------------------------------------------------------------------
// k_count_anchor's extraction stage (bit-plane tiles, bit-parallel anchor search, fail vectors) in front of
// k_count_fixed4_lds's matching stage: Counter mode, uniform library of 14..21-base features, --m <= 1.  Every window
// of another length than the features can equal or approach none of them (fast2q.py:683), so the only lookups are
// the four LDS bucket reads of lt_decide: no table traffic to L2, no ring, no drain, and -- nothing else in the loop
// touching memory -- the rows of the group's next tile really do travel while the current tile is searched.
// One workgroup per CU (the LDS tables take all of it); each group of 256 threads walks tiles like a k_count_anchor
// workgroup does.
#ifndef F2Q_ALT_THREADS
#define F2Q_ALT_THREADS 768
#endif
#define F2Q_ALT_GROUPS (F2Q_ALT_THREADS / F2Q_TILE)

template <int NW, int KB, bool SAMEQ, bool NEAR>
__global__ __launch_bounds__(F2Q_ALT_THREADS) void k_count_anchor_lt(const RunDev *__restrict__ runp,
                                                                    const LibDev *__restrict__ libp, PackedBlock pb,
                                                                    Accum acc)
{
    constexpr int NQW = 8 * NW;
    extern __shared__ uint32_t lt_smem[];
    constexpr uint32_t NT = NEAR ? 2u : 1u;
    uint32_t *tg = lt_smem;                                     // [NT][F2Q_LT_SLOTS] tags
    uint32_t *cnt = lt_smem + NT * F2Q_LT_SLOTS;                // [F2Q_LT_BUCKETS] two u16 counters per word
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    const LtDesc lt = lib.lt;
    const uint32_t nf = lib.n_features;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, slot_in_tile = tid & (F2Q_TILE - 1u), group = tid / F2Q_TILE;
    {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 F2Q_GLOBAL *src = (const v4 F2Q_GLOBAL *)gp(lt.tags);
        v4 *dst = reinterpret_cast<v4 *>(tg);
        for (uint32_t i = tid; i < NT * F2Q_LT_SLOTS / 4u; i += F2Q_ALT_THREADS) dst[i] = src[i];
        for (uint32_t i = tid; i < F2Q_LT_BUCKETS; i += F2Q_ALT_THREADS) cnt[i] = 0;
    }
    __syncthreads();
    const uint32_t ah_w = phred_add_hi(run.thr), ah_u = phred_add_hi(run.thr_up), ah_d = phred_add_hi(run.thr_down);
    const int flen = (int)lt.len;
    uint32_t w_reads = 0, w_perfect = 0, w_imperfect = 0, w_qfail = 0;      // this wave's counters (scalar registers)
    unsigned long long st_slow[5] = {0, 0, 0, 0, 0};                         // reads that took the byte-exact routine

    struct Planes { uint32_t lo[NW], hi[NW], q[NQW], len; };
    const auto b_base = gp(pb.bases) + slot_in_tile, q_base = gp(pb.qual) + slot_in_tile;
    const auto l_base = gp(pb.len) + slot_in_tile;
    const uint64_t b_stride = (uint64_t)pb.wb * F2Q_TILE, q_stride = (uint64_t)pb.wq * F2Q_TILE;
    // unconditional, fixed number of loads per tile (the host guarantees the length plane): exact wait counts
    auto request_tile = [&](Planes &p, uint32_t t) {
        const auto bp = b_base + (uint64_t)t * b_stride, qp = q_base + (uint64_t)t * q_stride;
        p.len = l_base[(uint64_t)t * F2Q_TILE];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            p.lo[w] = __builtin_nontemporal_load(bp + (uint64_t)w * F2Q_TILE);
            p.hi[w] = __builtin_nontemporal_load(bp + (uint64_t)(NW + w) * F2Q_TILE);
        }
#pragma unroll
        for (int i = 0; i < NQW; i++) p.q[i] = __builtin_nontemporal_load(qp + (uint64_t)i * F2Q_TILE);
    };
    auto decide_tile = [&](const Planes &p, uint32_t tile) {
        const uint32_t l = p.len;
        uint32_t FW[NW], FU[SAMEQ ? 1 : NW], FD[SAMEQ ? 1 : NW], FLG[NW];
        const bool live = l != F2Q_LEN_SKIP;
        const bool flagged = live && (l & F2Q_LEN_FLAG);
        if (__ballot(flagged) == 0ull) {              // the usual tile: no flag bits to strip, no flag planes to build
#pragma unroll
            for (int cw = 0; cw < NW; cw++) {
                uint32_t q8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q8[i] = p.q[8 * cw + i];
                FW[cw] = fail_word8<false>(q8, ah_w);
                if (!SAMEQ) { FU[cw] = fail_word8<false>(q8, ah_u); FD[cw] = fail_word8<false>(q8, ah_d); }
                FLG[cw] = 0u;
            }
        } else {
#pragma unroll
            for (int cw = 0; cw < NW; cw++) {
                uint32_t q8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q8[i] = p.q[8 * cw + i];
                FW[cw] = fail_word8(q8, ah_w);
                if (!SAMEQ) { FU[cw] = fail_word8(q8, ah_u); FD[cw] = fail_word8(q8, ah_d); }
                FLG[cw] = flagged ? flag_word8(q8) : 0u;
            }
        }
        const int r = (int)(l & F2Q_LEN_MASK);
        AnchorWin aw;
        if constexpr (SAMEQ) aw = anchor_window<NW, KB, KB>(run, p.lo, p.hi, FLG, r, FW, FW, FW);
        else aw = anchor_window<NW, KB, KB>(run, p.lo, p.hi, FLG, r, FU, FD, FW);
        const int L = aw.end - aw.start;
        const bool qf = live && aw.ok == 0;
        const bool slow = live && aw.ok == 2;
        // a window of another length than the features passed its Phred test but can equal or approach none of them
        const bool cand = live && aw.ok == 1 && L == flen;
        w_reads += (uint32_t)__popcll(__ballot(live && !slow));
        w_qfail += (uint32_t)__popcll(__ballot(qf));
        if (slow) {
            // negative-index slices (down-only anchor near the read start, negative --l): byte-exact routine
            const EcDev ec2{}; const Accum acc2 = acc; const PackedBlock pb2 = pb;
            anchor_slow(runp, libp, &ec2, &acc2, &pb2, tile, slot_in_tile, r, 0ull, st_slow, (l & F2Q_LEN_CASE) != 0u);
        }
        const int ws = cand ? aw.start : 0;
        const uint32_t forced = (cand && flagged && !(l & F2Q_LEN_CASE)) ? plane_extract<NW>(FLG, ws, flen) : 0u;   // (lower-case bases: marks for the anchors only)
        const LtProbe q = lt_probe(lt, plane_key<NW>(p.lo, p.hi, ws, flen));
        U2 e[4];
#pragma unroll
        for (int k = 0; k < (NEAR ? 4 : 2); k++) e[k] = lds_u2(tg + (uint32_t)(k >> 1) * F2Q_LT_SLOTS + 2u * q.b[k]);
        if (!NEAR) { e[2] = U2{F2Q_LT_EMPTY, F2Q_LT_EMPTY}; e[3] = e[2]; }
        const LtVerdict v = lt_decide<NEAR>(lt, q, e, forced, [&](uint32_t bk) { return lds_u2(tg + 2u * bk); });
        const LtPred cm = LT_P(cand), perfect = v.perfect & cm, imperfect = v.imperfect & cm;
        if (LT_TRUE(perfect | imperfect)) lt_count(cnt, v.slot, acc, lt);
        w_perfect += (uint32_t)__popcll(perfect);
        w_imperfect += (uint32_t)__popcll(imperfect);
    };

    const uint32_t stride = gridDim.x * F2Q_ALT_GROUPS, last = pb.n_tiles - 1u;
    uint32_t tile = blockIdx.x * F2Q_ALT_GROUPS + group;
    {
        Planes pa, pb2;
        if (tile < pb.n_tiles) {
            request_tile(pa, tile);
            for (;;) {
                __builtin_amdgcn_sched_barrier(0);
                request_tile(pb2, min(tile + stride, last));
                __builtin_amdgcn_sched_barrier(0);
                decide_tile(pa, tile);
                tile += stride;
                if (tile >= pb.n_tiles) break;
                __builtin_amdgcn_sched_barrier(0);
                request_tile(pa, min(tile + stride, last));
                __builtin_amdgcn_sched_barrier(0);
                decide_tile(pb2, tile);
                tile += stride;
                if (tile >= pb.n_tiles) break;
            }
        }
    }
    __syncthreads();
    {
        auto row = gpw(acc.slab) + (uint64_t)blockIdx.x * nf;
        for (uint32_t i = tid; i < nf; i += F2Q_ALT_THREADS) {
            const uint32_t s = gp(lt.slot_of)[i];
            row[i] = (cnt[s >> 1] >> ((s & 1u) << 4)) & 0xFFFFu;
        }
    }
    __syncthreads();
    const bool l0 = lane == 0;
    unsigned long long stv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) stv[k] = st_slow[k];
    if (l0) { stv[0] += w_reads; stv[1] += w_perfect; stv[2] += w_imperfect; stv[3] += w_reads - w_perfect - w_imperfect - w_qfail; stv[4] += w_qfail; }
    flush_stats(acc, stv, reinterpret_cast<unsigned long long *>(lt_smem), acc.stat_slab + (uint64_t)blockIdx.x * 8u);
}

// ---- anchored Extract+Count with the hot keys in LDS (EcHot, f2q_device.h) ------------------------------------------
// k_count_anchor_lt's tile walk and extraction stage; the single-word form of a passing window's key is looked up in
// the workgroup's copy of the hot keys (two buckets of two key words, 2 x ds_read_b128) and a hit is counted in an LDS
// counter; everything else takes the single-word table's insert with a bounded probe sequence.  Reads that have no
// single-word form (window over 29 bases, more than three 'N's), negative-index slices and reads that meet a full table
// are only noted in `defer`: the host sizes the tables for exactly those, and k_ec_deferred_keys / k_ec_deferred_slow
// decide them after the launch.  Reads whose only odd symbol is 'N' arrive flagged (a flag reads 'N').
struct HotPair { unsigned long long a, b; };
__device__ __forceinline__ HotPair lds_k2(const unsigned long long *p)
{
    typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
    const v2 v = *reinterpret_cast<const v2 *>(p);              // ds_read_b128
    return HotPair{v.x, v.y};
}
__device__ __forceinline__ uint32_t hot_slot_of(const HotPair &p1, const HotPair &p2, const HotProbe &q, unsigned long long k)
{
    return p1.a == k ? 2u * q.b1 : p1.b == k ? 2u * q.b1 + 1u : p2.a == k ? 2u * q.b2 : p2.b == k ? 2u * q.b2 + 1u : F2Q_HOT_NONE;
}

// an entry of the list of reads set aside: slot of the view, kind (1: byte-exact routine), window start and length
#define F2Q_DEFER_SLOW 0x80000000ull
#define F2Q_DEFER_FLAGS 0x40000000ull   // the read holds flagged bases ('N')
__device__ __forceinline__ unsigned long long defer_entry(uint64_t slot, bool slow, int start, int L)
{
    return ((unsigned long long)slot << 32) | (slow ? F2Q_DEFER_SLOW : 0ull) | ((unsigned long long)(start & 0x3FFF) << 16) | (unsigned long long)(L & 0xFFFF);
}

#ifndef F2Q_HOT_THREADS
#define F2Q_HOT_THREADS 768             // 12 waves per CU (the tables take the CU's LDS: one workgroup each); 1024 threads
                                        // leave 128 registers per lane and the two tile buffers spill (2.78 vs 2.69 ms)
#endif
#define F2Q_HOT_GROUPS (F2Q_HOT_THREADS / F2Q_TILE)
template <int NW, int KB, bool SAMEQ, bool LEARN>
__global__ __launch_bounds__(F2Q_HOT_THREADS) void k_extract_anchor_hot(const RunDev *__restrict__ runp, EcDev ec, EcHot hot,
                                                                       PackedBlock pb, Accum acc, uint64_t read_base,
                                                                       unsigned long long *__restrict__ defer, uint64_t slot_base,
                                                                       uint64_t defer_cap)
{
    // pb: a view of the block's tiles that starts slot_base slots into it; the reads set aside are listed by their slot
    // in the block.  (defer has room for every slot of the block; an index past it would be a logic error and is
    // reported, not written)
    unsigned long long *const defer_n = ec.ctr + F2Q_CTR_ASIDE;
    constexpr int NQW = 8 * NW;
    extern __shared__ uint32_t hot_smem[];
    unsigned long long *hk = reinterpret_cast<unsigned long long *>(hot_smem);   // [F2Q_HOT_SLOTS] key words
    uint32_t *cnt = hot_smem + 2u * F2Q_HOT_SLOTS;              // [F2Q_HOT_SLOTS] hits of this workgroup
    const RunDev &run = *runp;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, slot_in_tile = tid & (F2Q_TILE - 1u), group = tid / F2Q_TILE;
    {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 F2Q_GLOBAL *src = (const v4 F2Q_GLOBAL *)gp(hot.keys);
        v4 *dst = reinterpret_cast<v4 *>(hk);
        for (uint32_t i = tid; i < F2Q_HOT_SLOTS / 2u; i += F2Q_HOT_THREADS) dst[i] = src[i];
        for (uint32_t i = tid; i < F2Q_HOT_SLOTS; i += F2Q_HOT_THREADS) cnt[i] = 0;
    }
    __syncthreads();
    const unsigned long long first_max = gp(hot.meta)[0];       // a read at or past it cannot lower a hot key's first read
    const uint32_t ah_w = phred_add_hi(run.thr), ah_u = phred_add_hi(run.thr_up), ah_d = phred_add_hi(run.thr_down);
    uint32_t w_reads = 0, w_pass = 0, w_qfail = 0;              // this wave's counters (scalar registers)
    uint32_t n_new = 0;

    struct Planes { uint32_t lo[NW], hi[NW], q[NQW], len; };
    const auto b_base = gp(pb.bases) + slot_in_tile, q_base = gp(pb.qual) + slot_in_tile;
    const auto l_base = gp(pb.len) + slot_in_tile;
    const uint64_t b_stride = (uint64_t)pb.wb * F2Q_TILE, q_stride = (uint64_t)pb.wq * F2Q_TILE;
    auto request_tile = [&](Planes &p, uint32_t t) {
        const auto bp = b_base + (uint64_t)t * b_stride, qp = q_base + (uint64_t)t * q_stride;
        p.len = l_base[(uint64_t)t * F2Q_TILE];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            p.lo[w] = __builtin_nontemporal_load(bp + (uint64_t)w * F2Q_TILE);
            p.hi[w] = __builtin_nontemporal_load(bp + (uint64_t)(NW + w) * F2Q_TILE);
        }
#pragma unroll
        for (int i = 0; i < NQW; i++) p.q[i] = __builtin_nontemporal_load(qp + (uint64_t)i * F2Q_TILE);
    };
    // append the lanes of `m` to the list (rare: one counter bump per wave)
    auto set_aside = [&](unsigned long long m, bool mine, unsigned long long entry) {
        unsigned long long at = 0;
        const int first = __builtin_ctzll(m);
        if (lane == (uint32_t)first) at = ec_fetch_add(defer_n, (unsigned long long)__popcll(m));
        at = __shfl(at, first, 64);
        const unsigned long long di = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (mine && di < defer_cap) gpw(defer)[di] = entry;
        else if (mine) F2Q_ST64(&ec.ctr[2], 7ull);
    };
    auto decide_tile = [&](const Planes &p, uint32_t tile) {
        const uint32_t l = p.len;
        uint32_t FW[NW], FU[SAMEQ ? 1 : NW], FD[SAMEQ ? 1 : NW], FLG[NW];
        const bool live = l != F2Q_LEN_SKIP;
        const bool flagged = live && (l & F2Q_LEN_FLAG);       // the read holds 'N's (flag bits in the quality bytes)
        if (__ballot(flagged) == 0ull) {              // the usual tile: no flag bits to strip, no flag planes to build
#pragma unroll
            for (int cw = 0; cw < NW; cw++) {
                uint32_t q8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q8[i] = p.q[8 * cw + i];
                FW[cw] = fail_word8<false>(q8, ah_w);
                if (!SAMEQ) { FU[cw] = fail_word8<false>(q8, ah_u); FD[cw] = fail_word8<false>(q8, ah_d); }
                FLG[cw] = 0u;
            }
        } else {
#pragma unroll
            for (int cw = 0; cw < NW; cw++) {
                uint32_t q8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q8[i] = p.q[8 * cw + i];
                FW[cw] = fail_word8(q8, ah_w);
                if (!SAMEQ) { FU[cw] = fail_word8(q8, ah_u); FD[cw] = fail_word8(q8, ah_d); }
                FLG[cw] = flagged ? flag_word8(q8) : 0u;
            }
        }
        const int r = (int)(l & F2Q_LEN_MASK);
        AnchorWin aw;
        if constexpr (SAMEQ) aw = anchor_window<NW, KB, KB>(run, p.lo, p.hi, FLG, r, FW, FW, FW);
        else aw = anchor_window<NW, KB, KB>(run, p.lo, p.hi, FLG, r, FU, FD, FW);
        const int L = aw.end - aw.start;
        const bool qf = live && aw.ok == 0;
        const bool pass = live && aw.ok == 1;
        const bool slow = live && aw.ok == 2;                   // negative-index slices: byte-exact routine
        // the key's single-word form; a window that has none (too long, too many 'N's) is set aside for the byte-string table
        const int wl0 = (pass && L >= 0 && L <= F2Q_EC64_MAXLEN) ? L : 0, ws0 = (pass && L >= 0 && L <= F2Q_EC64_MAXLEN) ? aw.start : 0;
        const bool keyflag = flagged && !(l & F2Q_LEN_CASE);    // (lower-case bases: marks for the anchors only, plain bases in the key)
        const uint32_t nmask = (keyflag && wl0 > 0) ? plane_extract<NW>(FLG, ws0, wl0) : 0u;
        unsigned long long k = 0;
        const bool has_word = ec64_word(plane_key<NW>(p.lo, p.hi, ws0, wl0), nmask, wl0, k) && pass && L <= F2Q_EC64_MAXLEN;
        const bool later = slow || (pass && !has_word);
        const bool ins = pass && !later;
        w_reads += (uint32_t)__popcll(__ballot(live && !later));
        w_qfail += (uint32_t)__popcll(__ballot(qf));
        w_pass += (uint32_t)__popcll(__ballot(ins));
        const uint64_t slot = (uint64_t)tile * F2Q_TILE + slot_in_tile;
        const unsigned long long lm = __ballot(later);
        if (lm) {
            set_aside(lm, later, defer_entry(slot_base + slot, slow, aw.start, L) | (keyflag ? F2Q_DEFER_FLAGS : 0ull));
            const unsigned long long sm = __ballot(slow);
            if (sm && lane == 0) ec_fetch_add(ec.ctr + F2Q_CTR_ASIDE_SLOW, (unsigned long long)__popcll(sm));
        }
        const HotProbe q = hot_probe(k);
        const HotPair p1 = lds_k2(hk + 2u * q.b1), p2 = lds_k2(hk + 2u * q.b2);
        const uint32_t s = hot_slot_of(p1, p2, q, k);
        unsigned long long gi = 0;
        if (ins) gi = read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot);
        const bool hit = ins && s != F2Q_HOT_NONE && gi >= first_max;
        bool full = false;
        if (hit) atomicAdd(&cnt[s], 1u);
        else if (ins) {
            uint32_t ts = 0; unsigned long long before = 0;
            const uint32_t rr = ec64_try_insert<LEARN>(ec, k, gi, F2Q_HOT_MAXPROBE, ts, before);
            full = rr == 2u;
            n_new += rr & 1u;
            if (LEARN && rr != 2u && before + 1ull == F2Q_HOT_MINCOUNT) {       // a few thousand times per sample
                const unsigned long long at = ec_fetch_add(ec.ctr + F2Q_CTR_CAND, 1ull);
                if (at < F2Q_HOT_CAND) gpw(hot.cand)[at] = ts;
            }
        }
        const unsigned long long fm = __ballot(full);
        if (fm) {                                               // the table is (nearly) full: decided after it has grown
            set_aside(fm, full, defer_entry(slot_base + slot, true, aw.start, L) | (keyflag ? F2Q_DEFER_FLAGS : 0ull));
            if (lane == 0) ec_fetch_add(ec.ctr + F2Q_CTR_ASIDE_SLOW, (unsigned long long)__popcll(fm));
            w_reads -= (uint32_t)__popcll(fm); w_pass -= (uint32_t)__popcll(fm);
        }
    };

    const uint32_t stride = gridDim.x * F2Q_HOT_GROUPS, last = pb.n_tiles - 1u;
    uint32_t tile = blockIdx.x * F2Q_HOT_GROUPS + group;
    {
        Planes pa, pb2;
        if (tile < pb.n_tiles) {
            request_tile(pa, tile);
            for (;;) {
                __builtin_amdgcn_sched_barrier(0);
                request_tile(pb2, min(tile + stride, last));
                __builtin_amdgcn_sched_barrier(0);
                decide_tile(pa, tile);
                tile += stride;
                if (tile >= pb.n_tiles) break;
                __builtin_amdgcn_sched_barrier(0);
                request_tile(pa, min(tile + stride, last));
                __builtin_amdgcn_sched_barrier(0);
                decide_tile(pb2, tile);
                tile += stride;
                if (tile >= pb.n_tiles) break;
            }
        }
    }
    ec64_report_new(ec, n_new);
    __syncthreads();
    for (uint32_t i = tid; i < F2Q_HOT_SLOTS; i += F2Q_HOT_THREADS) {
        const uint32_t n = cnt[i];
        if (n) {
            const uint32_t ts = gp(hot.slot)[i];
            if (ts <= ec.k64_mask) ec_fetch_add(&ec.k64_count[ts], (unsigned long long)n);
            else F2Q_ST64(&ec.ctr[2], 8ull);                     // a stale link: reported by the host, never written
        }
    }
    __syncthreads();
    unsigned long long stv[5] = {0, 0, 0, 0, 0};
    if (lane == 0) { stv[0] = w_reads; stv[1] = w_pass; stv[3] = w_reads - w_pass - w_qfail; stv[4] = w_qfail; }
    flush_stats(acc, stv, reinterpret_cast<unsigned long long *>(hot_smem), nullptr);
}

// key bytes read straight from the bit planes of a tile slot, 32 bases per pair of loads (ec_insert needs len,
// key_hash() and key_word())
struct PlaneKV {
    uint32_t lo[5], hi[5];                  // the slot's base planes (loaded together, once)
    const uint32_t F2Q_GLOBAL *qp;          // the slot's column of the quality planes (flag bits), or nullptr: no flagged base
    uint32_t nw;
    int start, len;
    __device__ void load(const uint32_t F2Q_GLOBAL *bp, uint32_t nw_)
    {
        nw = nw_;
#pragma unroll
        for (uint32_t w = 0; w < 5u; w++) {
            const uint32_t ww = w < nw_ ? w : nw_ - 1u;
            lo[w] = bp[(uint64_t)ww * F2Q_TILE]; hi[w] = bp[(uint64_t)(nw_ + ww) * F2Q_TILE];
        }
    }
    __device__ uint32_t codes16(int from) const      // 16 bases from position `from` as 2-bit codes (LSB first)
    {
        const uint32_t w = (uint32_t)from >> 5, sh = (uint32_t)from & 31u;
        uint32_t l0 = 0, h0 = 0, l1 = 0, h1 = 0;
#pragma unroll
        for (uint32_t i = 0; i < 5u; i++) {               // register arrays: select, never index
            if (i == w) { l0 = lo[i]; h0 = hi[i]; }
            if (i == w + 1u) { l1 = lo[i]; h1 = hi[i]; }
        }
        const uint64_t l = (uint64_t)l0 | ((uint64_t)l1 << 32), h = (uint64_t)h0 | ((uint64_t)h1 << 32);
        return spread16((uint32_t)(l >> sh) & 0xFFFFu) | (spread16((uint32_t)(h >> sh) & 0xFFFFu) << 1);
    }
    __device__ bool is_n(int pos) const
    {
        return qp && ((qp[(uint64_t)planar_qword((uint32_t)pos) * F2Q_TILE] >> (8u * planar_qbyte((uint32_t)pos) + 7u)) & 1u);
    }
    __device__ uint8_t byte_of(uint32_t code, int pos) const { return is_n(pos) ? (uint8_t)'N' : (uint8_t)(0x54474341u >> (8u * (code & 3u))); }   // "ACGT"
    __device__ uint8_t at(int k) const { return byte_of(codes16(start + k), start + k); }
};
__device__ __forceinline__ uint64_t key_hash(const PlaneKV &kv)
{
    uint64_t h = 1469598103934665603ull ^ (uint64_t)kv.len;
    for (int k0 = 0; k0 < kv.len; k0 += 16) {
        uint32_t c = kv.codes16(kv.start + k0);
        const int n = kv.len - k0 < 16 ? kv.len - k0 : 16;
        for (int j = 0; j < n; j++, c >>= 2) { h ^= kv.byte_of(c, kv.start + k0 + j); h *= 1099511628211ull; }
    }
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}
__device__ __forceinline__ uint32_t key_word(const PlaneKV &kv, int w)
{
    uint32_t c = kv.codes16(kv.start + 4 * w), v = 0;
    for (int j = 0; j < 4; j++, c >>= 2) if (4 * w + j < kv.len) v |= (uint32_t)kv.byte_of(c, kv.start + 4 * w + j) << (8 * j);
    return v;
}

// the reads k_extract_anchor_hot set aside.  Windows the single-word table cannot hold: the key goes to the byte-string
// table from the planes, at the place the search already found ...
__global__ __launch_bounds__(256) void k_ec_deferred_keys(EcDev ec, PackedBlock pb, Accum acc, uint64_t read_base,
                                                          const unsigned long long *__restrict__ defer)
{
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    unsigned long long n = ec.ctr[F2Q_CTR_ASIDE];
    if (n > pb.n_slots) n = pb.n_slots;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long e = gp(defer)[i];
        const uint64_t slot = e >> 32;
        if ((e & F2Q_DEFER_SLOW) || slot >= pb.n_slots) continue;
        PlaneKV kv;
        kv.load(gp(pb.bases) + (slot / F2Q_TILE) * (uint64_t)pb.wb * F2Q_TILE + (slot % F2Q_TILE), pb.planar_nw);
        kv.qp = (e & F2Q_DEFER_FLAGS) ? gp(pb.qual) + (slot / F2Q_TILE) * (uint64_t)pb.wq * F2Q_TILE + (slot % F2Q_TILE) : nullptr;
        kv.start = (int)((e >> 16) & 0x3FFFu); kv.len = (int)(e & 0xFFFFu);
        const unsigned long long gi = read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot);
        ec_insert(ec, kv, gi);
        st[0]++; st[1]++;
    }
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, nullptr);
}
// ... and the rest (negative-index slices, reads that met a full table): one thread per read, byte-exact routine on
// the decoded planes
__global__ __launch_bounds__(256) void k_ec_deferred_slow(const RunDev *__restrict__ runp, const LibDev *__restrict__ libp, EcDev ec,
                                                          PackedBlock pb, Accum acc, uint64_t read_base,
                                                          const unsigned long long *__restrict__ defer)
{
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    unsigned long long n = ec.ctr[F2Q_CTR_ASIDE];
    if (n > pb.n_slots) n = pb.n_slots;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long e = gp(defer)[i];
        const uint64_t slot = e >> 32;
        if (!(e & F2Q_DEFER_SLOW) || slot >= pb.n_slots) continue;
        const uint32_t l = gp(pb.len)[slot];
        const unsigned long long gi = read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot);
        const EcDev ec2 = ec; const Accum acc2 = acc; const PackedBlock pb2 = pb;
        anchor_slow(runp, libp, &ec2, &acc2, &pb2, (uint32_t)(slot / F2Q_TILE), (uint32_t)(slot % F2Q_TILE), (int)(l & F2Q_LEN_MASK), gi, st, (l & F2Q_LEN_CASE) != 0u);
    }
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, nullptr);
}

// ---- Extract+Count with a fixed window and the hot keys in LDS ---------------------------------------------------------
// k_extract_fixed4's tile walk (one wave per tile, 4 reads per lane) in front of k_extract_anchor_hot's counting stage:
// amplicon-like samples repeat a few thousand windows, and those are counted in LDS instead of one device-scope atomic
// per read.  The packer passes only reads whose window has a single-word form (<= 29 bases; with 'N's: what ec64_word
// holds), so the only reads set aside are those that meet a full table (k_ec_deferred_fixed inserts them after the
// table has grown).
#define F2Q_FH_THREADS 1024
#define F2Q_FH_WAVES (F2Q_FH_THREADS / 64)
template <bool LEARN>
__global__ __launch_bounds__(F2Q_FH_THREADS) void k_extract_fixed4_hot(const RunDev *__restrict__ runp, EcDev ec, EcHot hot, PackedBlock pb,
                                                                       Accum acc, uint64_t read_base,
                                                                       unsigned long long *__restrict__ defer, uint64_t slot_base, uint64_t defer_cap)
{
    extern __shared__ uint32_t hot_smem[];
    unsigned long long *hk = reinterpret_cast<unsigned long long *>(hot_smem);
    uint32_t *cnt = hot_smem + 2u * F2Q_HOT_SLOTS;
    const RunDev &run = *runp;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 F2Q_GLOBAL *src = (const v4 F2Q_GLOBAL *)gp(hot.keys);
        v4 *dst = reinterpret_cast<v4 *>(hk);
        for (uint32_t i = tid; i < F2Q_HOT_SLOTS / 2u; i += F2Q_FH_THREADS) dst[i] = src[i];
        for (uint32_t i = tid; i < F2Q_HOT_SLOTS; i += F2Q_FH_THREADS) cnt[i] = 0;
    }
    __syncthreads();
    const unsigned long long first_max = gp(hot.meta)[0];
    const FixedGeom g = fixed_geom(run);
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    uint32_t n_new = 0;
    for (uint32_t base = blockIdx.x * F2Q_FH_WAVES; base < pb.n_tiles; base += gridDim.x * F2Q_FH_WAVES) {
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
            qrow[r] = want < pb.wq ? ld_u4<true>(qp + (uint64_t)row * F2Q_TILE) : U4{0, 0, 0, 0};
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
        // the four reads' tag buckets are read together, then verified and counted one after the other
        unsigned long long k[4]; bool ins[4]; HotProbe q[4]; HotPair p1[4], p2[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t l = pb.len ? (((j < 2 ? len01 : len23) >> (16 * (j & 1))) & 0xFFFFu) : pb.rmax;
            const bool live = l != F2Q_LEN_SKIP;
            ins[j] = live && !bad[j];
            st[0] += live; st[4] += live && bad[j];
            k[j] = fixed4_ec_word(g, brow, qrow, j, l);                     // Python slice clipping (:354); 'N's spelt from the flag bits
            q[j] = hot_probe(k[j]);
            p1[j] = lds_k2(hk + 2u * q[j].b1); p2[j] = lds_k2(hk + 2u * q[j].b2);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t slot = (uint64_t)tile * F2Q_TILE + 4u * lane + (uint32_t)j;
            const uint32_t s = hot_slot_of(p1[j], p2[j], q[j], k[j]);
            bool full = false;
            unsigned long long gi = 0;
            if (ins[j]) gi = read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot);
            const bool hit = ins[j] && s != F2Q_HOT_NONE && gi >= first_max;
            if (hit) { atomicAdd(&cnt[s], 1u); st[1]++; }
            else if (ins[j]) {
                uint32_t ts = 0; unsigned long long before = 0;
                const uint32_t rr = ec64_try_insert<LEARN>(ec, k[j], gi, F2Q_HOT_MAXPROBE, ts, before);
                full = rr == 2u;
                n_new += rr & 1u;
                st[1] += !full;
                if (LEARN && rr != 2u && before + 1ull == F2Q_HOT_MINCOUNT) {
                    const unsigned long long at = ec_fetch_add(ec.ctr + F2Q_CTR_CAND, 1ull);
                    if (at < F2Q_HOT_CAND) gpw(hot.cand)[at] = ts;
                }
            }
            const unsigned long long fm = __ballot(full);
            if (fm) {                                           // the table is (nearly) full: inserted after it has grown
                unsigned long long at = 0;
                const int first = __builtin_ctzll(fm);
                if (lane == (uint32_t)first) at = ec_fetch_add(ec.ctr + F2Q_CTR_ASIDE, (unsigned long long)__popcll(fm));
                at = __shfl(at, first, 64);
                const unsigned long long di = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                if (full && di < defer_cap) gpw(defer)[di] = defer_entry(slot_base + slot, true, 0, 0);
                else if (full) F2Q_ST64(&ec.ctr[2], 7ull);
                if (lane == 0) ec_fetch_add(ec.ctr + F2Q_CTR_ASIDE_SLOW, (unsigned long long)__popcll(fm));
            }
        }
    }
    ec64_report_new(ec, n_new);
    __syncthreads();
    for (uint32_t i = tid; i < F2Q_HOT_SLOTS; i += F2Q_FH_THREADS) {
        const uint32_t n = cnt[i];
        if (n) {
            const uint32_t ts = gp(hot.slot)[i];
            if (ts <= ec.k64_mask) ec_add(&ec.k64_count[ts], (unsigned long long)n);
            else F2Q_ST64(&ec.ctr[2], 8ull);
        }
    }
    __syncthreads();
    flush_stats(acc, st, reinterpret_cast<unsigned long long *>(hot_smem), nullptr);
}

// the reads k_extract_fixed4_hot set aside (they met a full table): the window again from the tile, plain insert
__global__ __launch_bounds__(256) void k_ec_deferred_fixed(const RunDev *__restrict__ runp, EcDev ec, PackedBlock pb, Accum acc, uint64_t read_base,
                                                           const unsigned long long *__restrict__ defer)
{
    const RunDev &run = *runp;
    const FixedGeom g = fixed_geom(run);
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    uint32_t n_new = 0;
    unsigned long long n = ec.ctr[F2Q_CTR_ASIDE];
    if (n > pb.n_slots) n = pb.n_slots;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
        const uint64_t slot = gp(defer)[i] >> 32;
        if (slot >= pb.n_slots) continue;
        const uint32_t tile = (uint32_t)(slot / F2Q_TILE), in_tile = (uint32_t)(slot % F2Q_TILE), lane4 = in_tile >> 2, j = in_tile & 3u;
        U4 brow[F2Q_MAXBROWS], qrow[F2Q_MAXQROWS];
        const auto bp = gp(pb.bases) + (uint64_t)tile * pb.wb * F2Q_TILE + 4u * lane4;
        const auto qp = gp(pb.qual) + (uint64_t)tile * pb.wq * F2Q_TILE + 4u * lane4;
#pragma unroll
        for (int r = 0; r < F2Q_MAXBROWS; r++) {
            uint32_t row = (uint32_t)g.bw0 + (uint32_t)(r < g.nb ? r : (g.nb > 0 ? g.nb - 1 : 0));
            row = row < pb.wb ? row : pb.wb - 1u;
            brow[r] = ld_u4<false>(bp + (uint64_t)row * F2Q_TILE);
        }
        const uint32_t l = pb.len ? gp(pb.len)[slot] : pb.rmax;
#pragma unroll
        for (int r = 0; r < F2Q_MAXQROWS; r++) {                   // the flag bits of a window with 'N's
            const uint32_t want = (uint32_t)g.qw0 + (uint32_t)(r < g.nq ? r : (g.nq > 0 ? g.nq - 1 : 0));
            const uint32_t row = want < pb.wq ? want : pb.wq - 1u;
            qrow[r] = ((l & F2Q_LEN_FLAG) && want < pb.wq) ? ld_u4<false>(qp + (uint64_t)row * F2Q_TILE) : U4{0, 0, 0, 0};
        }
        n_new += ec64_insert_word(ec, fixed4_ec_word(g, brow, qrow, (int)j, l), read_base + pb.first_index + (pb.index ? (uint64_t)gp(pb.index)[slot] : slot));
        st[1]++;
    }
    ec64_report_new(ec, n_new);
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, nullptr);
}

// hot-key set of the single-word table: built from the candidates the learning launches noted (in the order they
// reached F2Q_HOT_MINCOUNT reads, the first F2Q_HOT_CAP of them), re-linked after the table has grown
__global__ __launch_bounds__(256) void k_ec_hot_build(EcDev ec, EcHot hot)
{
    unsigned long long n = ec.ctr[F2Q_CTR_CAND];
    if (n > F2Q_HOT_CAND) n = F2Q_HOT_CAND;
    if (n > F2Q_HOT_CAP) n = F2Q_HOT_CAP;
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = hot.cand[i];
    if (s > ec.k64_mask) return;
    const unsigned long long k = ec.k64_slots[s];
    if (k == KEY_EMPTY) return;
    const HotProbe q = hot_probe(k);
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t b = pass ? q.b2 : q.b1;
        for (uint32_t j = 0; j < 2u; j++) {
            const unsigned long long old = atomicCAS(&hot.keys[2u * b + j], KEY_EMPTY, k);
            if (old == KEY_EMPTY) {
                hot.slot[2u * b + j] = s;
                atomicMax(&hot.meta[0], ec.k64_first[s]);
                return;
            }
            if (old == k) return;                                // (a slot noted twice)
        }
    }
    // both buckets full: this key stays cold
}
__global__ __launch_bounds__(256) void k_ec_hot_relink(EcDev ec, EcHot hot)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= F2Q_HOT_SLOTS) return;
    const unsigned long long k = hot.keys[i];
    if (k == KEY_EMPTY) return;
    uint32_t s = hash32(k ^ (k >> 29), 32) & ec.k64_mask;
    for (uint32_t guard = 0; guard <= ec.k64_mask; guard++) {
        const unsigned long long v = ec.k64_slots[s];
        if (v == k) { hot.slot[i] = s; return; }
        if (v == KEY_EMPTY) break;
        s = (s + 1) & ec.k64_mask;
    }
    hot.keys[i] = KEY_EMPTY;                                     // cannot happen (growth keeps every key); stay exact anyway
}

// Large libraries (no per-workgroup LDS histogram of the whole library): the counting kernel leaves the feature index
// of every read in hit_buf; here workgroup (range, part) histograms the indices of its part that fall into its range
// of F2Q_HIST_RANGE features in LDS and writes that stretch of slab row `part`.  hit_buf is read n_ranges times, from
// the Infinity Cache when it fits (4 B per read).
__global__ __launch_bounds__(1024) void k_hist_ranges(const uint32_t *__restrict__ hit_buf, uint64_t n_slots, uint32_t nf,
                                                       uint32_t n_parts, uint32_t *__restrict__ slab)
{
    extern __shared__ uint32_t rh[];
    const uint32_t range = blockIdx.x / n_parts, part = blockIdx.x % n_parts;
    const uint32_t f0 = range * F2Q_HIST_RANGE, fn = (nf - f0) < F2Q_HIST_RANGE ? (nf - f0) : F2Q_HIST_RANGE;
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

// One read per lane, byte-exact routine.  The routine walks the record byte by byte with dependent loads, so each lane
// first copies its record into LDS (aligned 4-byte loads, all in flight together; odd word stride per lane: no bank
// conflicts) and then works on LDS pointers; a record longer than the staging area is walked in global memory as before.
// 64-thread workgroups of 24 KiB: they fit beside a workgroup of the hot-key kernel (128 KiB) on the same CU.
#define F2Q_GEN_THREADS 64
#define F2Q_GEN_WORDS 48u               // words of a sequence line / of a quality line the staging area holds (192 bytes)
#define F2Q_GEN_STRIDE 97u              // words per lane: 2 x 48 + 1
__device__ __forceinline__ void stage_line(uint32_t *dst, gbytes raw, unsigned long long off, int n)
{
    // bytes [off, off + n) -> dst, keeping the misalignment: byte i of the line sits at ((uint8_t *)dst)[(off & 3) + i]
    const uint32_t mis = (uint32_t)off & 3u, full = (mis + (uint32_t)n) >> 2;
    const uint32_t F2Q_GLOBAL *src = (const uint32_t F2Q_GLOBAL *)(raw + (off - mis));
#pragma unroll 8
    for (uint32_t k = 0; k < full; k++) dst[k] = src[k];
    uint8_t *db = reinterpret_cast<uint8_t *>(dst);
    for (uint32_t k = 4u * full; k < mis + (uint32_t)n; k++) db[k] = raw[off - mis + k];      // <= 3 bytes, never past the line
}
__global__ __launch_bounds__(F2Q_GEN_THREADS) void k_count_general(const RunDev *__restrict__ runp,
                                                                    const LibDev *__restrict__ libp, EcDev ec, RawBlock rb,
                                                                    Accum acc)
{
    __shared__ uint32_t stage[F2Q_GEN_THREADS * F2Q_GEN_STRIDE];
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    uint32_t n_new = 0;
    const auto raw = gp(rb.raw);
    uint32_t *mine = stage + threadIdx.x * F2Q_GEN_STRIDE;
    for (uint64_t i = (uint64_t)blockIdx.x * F2Q_GEN_THREADS + threadIdx.x; i < rb.n; i += (uint64_t)gridDim.x * F2Q_GEN_THREADS) {
        const unsigned long long so = gp(rb.off)[i];
        const int r = (int)gp(rb.len)[i], qn = (int)gp(rb.qlen)[i];
        const unsigned long long qo = rb.qoff ? gp(rb.qoff)[i] : so + (unsigned long long)r;
        const unsigned long long gi = rb.first_index + (rb.index ? gp(rb.index)[i] : i);
        const uint32_t ms = (uint32_t)so & 3u, mq = (uint32_t)qo & 3u;
        if (r >= 0 && qn >= 0 && ms + (uint32_t)r <= 4u * F2Q_GEN_WORDS && mq + (uint32_t)qn <= 4u * F2Q_GEN_WORDS) {
            stage_line(mine, raw, so, r);
            stage_line(mine + F2Q_GEN_WORDS, raw, qo, qn);
            const uint8_t *sl = reinterpret_cast<const uint8_t *>(mine) + ms;
            const uint8_t *ql = reinterpret_cast<const uint8_t *>(mine + F2Q_GEN_WORDS) + mq;
            general_read<const uint8_t *, true>(run, lib, ec, acc, sl, r, ql, qn, gi, st, &n_new);
        } else {
            general_read<gbytes, true>(run, lib, ec, acc, raw + so, r, raw + qo, qn, gi, st, &n_new);
        }
    }
    if (run.mode == 1) ec64_report_new(ec, n_new);
    __shared__ unsigned long long st_lds[8];
    flush_stats(acc, st, st_lds, nullptr);
}

