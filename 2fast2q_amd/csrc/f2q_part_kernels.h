// f2q_part_kernels.h -- fixed-offset Counter mode for uniform libraries too large for one workgroup's LDS (BASELINE
// config 4: 100 k guides x 20 bases, --m 1), included by f2q_lib.hip after f2q_count_kernels.h.
//
// The library is dealt into partitions by a hash of the features' half 0 (f2q_device.h: PtDesc); a partition's table 0
// is what one workgroup can hold in LDS.  A block is counted in two passes over chunks of its tiles:
//   k_part_scatter  streams the tile rows under the window once (the 30 B/read of the LDS-table kernel), applies the
//                   Phred rule, builds the 2-bit keys and deals every candidate read, as one 8-byte entry, into the
//                   stream of its partition: each wave keeps a 128-entry ring per partition in LDS (positions handed out
//                   by an LDS atomic) and writes a ring out 64 entries = 512 contiguous bytes at a time, to the end of the
//                   stream its WORKGROUP keeps for that partition (the place is reserved with one LDS atomic of the
//                   workgroup's cursor) -- no global atomics, no ordering between workgroups, and every stream grows
//                   and is later read as one long sequential run;
//   k_part_count    workgroup (partition p, member m) copies partition p's table 0 into LDS once and walks the streams
//                   of p: exact hits and the neighbours sharing the read's half 0 are decided from LDS exactly as in
//                   k_count_fixed4_lds; only reads without an exact hit probe the one global table 1 (two 8-byte loads,
//                   L2 resident) for the neighbours sharing half 1.  Hits are counted in the workgroup's u16 LDS
//                   histogram by table-0 slot; a hit found through table 1 belongs to another partition and is counted by
//                   table-1 slot in a global u32 array (5 % of the reads);
//   k_part_reduce   per feature: its partition's slab rows + its table-1 slot's count -> the int64 vector.
// Reference semantics: fast2q.py:349-357 (window, Phred), :365-367 (exact hit), :692-750 + :660-690 (unique feature at
// distance 1), :366-393 (counters).
#pragma once

#define F2Q_PS_RING 128u                     // ring entries per (wave, partition); a ring is written out 64 at a time
#define F2Q_PC_THREADS 1024                  // (the count kernel takes its wave count from the launch)
#define F2Q_PC_STEP 128u                     // entries a wave takes per step of k_part_count (two per lane, one 16-byte load)

struct PartScratch {
    unsigned long long *streams;             // [n_wg1][n_parts][cap] entries (pt_entry): one stream per scatter workgroup and partition
    uint32_t *cnt;                           // [n_wg1][n_parts] entries written
    uint32_t cap, n_wg1;
};

// the workgroup's sums of the five reference counters, ADDED to its own row of 8 (the rounds of a block accumulate there;
// k_part_reduce sums the rows and clears them)
__device__ __forceinline__ void flush_stats_add(unsigned long long st[5], unsigned long long *lds8, unsigned long long *row)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < 8) lds8[threadIdx.x] = 0;
    __syncthreads();
    for (int k = 0; k < 5; k++) {
        unsigned long long v = wave_sum(st[k]);
        if (lane == 0 && v) atomicAdd(&lds8[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 5) { const unsigned long long v = lds8[threadIdx.x]; if (v) gpw(row)[threadIdx.x] += v; }
}

// ---- pass 1 --------------------------------------------------------------------------------------------------
// PW waves per workgroup: a wave's rings take n_parts KiB of LDS, so PW = 16 up to 8 partitions, 8 up to 16, 4 up to 32.
template <int NQ, int NB, bool A20, int PW>
__global__ __launch_bounds__(64 * PW) void k_part_scatter(const RunDev *__restrict__ runp, const LibDev *__restrict__ libp,
                                                          PackedBlock pb, Accum acc, PartScratch ps, uint32_t tile0, uint32_t tile1)
{
    extern __shared__ unsigned long long ps_smem[];
    const RunDev &run = *runp;
    const LibDev &lib = *libp;
    const uint32_t P = lib.pt.n_parts;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));    // wave-uniform for the compiler: loop control below stays on the scalar unit
    unsigned long long *ring = ps_smem + (size_t)wave * P * F2Q_PS_RING;                         // [P][F2Q_PS_RING]
    uint32_t *cur = reinterpret_cast<uint32_t *>(ps_smem + (size_t)PW * P * F2Q_PS_RING) + wave * P;   // [P] entries pushed so far
    uint32_t *wg_cur = reinterpret_cast<uint32_t *>(ps_smem + (size_t)PW * P * F2Q_PS_RING) + PW * P;  // [P] entries of the workgroup's streams
    if (lane < P) cur[lane] = 0;
    if (tid < P) wg_cur[tid] = 0;
    __syncthreads();
    uint32_t headv = 0;                                             // lane p < P: entries of partition p written out so far
    uint32_t hb0 = lib.pt.hb0, L = lib.pt.len;
    if (A20) { hb0 = 20u; L = 20u; }
    FixedGeom g = fixed_geom(run);
    if (A20) { g.L = 20; g.nq = 5; g.nb = 2; g.sh = 0; g.qm_first = 0x80808080u; g.qm_last = 0x80808080u; g.kmask = (1ull << 40) - 1ull; }
    const int need = g.st + g.L;
    const bool do_near = run.miss > 0;
    uint32_t w_reads = 0, w_qfail = 0, w_nonal = 0;                 // wave totals in scalar registers
    constexpr int QR = NQ ? NQ : F2Q_MAXQROWS, BR = NB ? NB : F2Q_MAXBROWS;
    constexpr bool PIPE = NQ != 0;
    const uint32_t gw = blockIdx.x * PW + wave, n_waves = gridDim.x * PW;
    unsigned long long F2Q_GLOBAL *mine = gpw(ps.streams) + (uint64_t)blockIdx.x * P * ps.cap;    // this workgroup's P streams

    struct Rows { U4 b[BR], q[QR]; uint32_t len01, len23; };
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
    // write n <= 64 entries of partition p's ring (from ring position hp) to the end of the workgroup's stream of p
    auto write_out = [&](uint32_t p, uint32_t hp, uint32_t n) {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&wg_cur[p], n);                   // (one lane: the place of these n entries)
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (lane < n) mine[(uint64_t)p * ps.cap + at + lane] = ring[p * F2Q_PS_RING + ((hp + lane) & (F2Q_PS_RING - 1u))];
    };
    auto scatter_tile = [&](const Rows &r) {
        uint32_t bad[4] = {0, 0, 0, 0};
        if (g.add_hi) {
            if (A20) {
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
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t l = ((j < 2 ? r.len01 : r.len23) >> (16 * (j & 1))) & 0xFFFFu;
            const bool live = l != F2Q_LEN_SKIP, qf = live && bad[j] != 0u;
            // a read that ends inside the window can equal or approach no feature (every feature is L long, :683)
            bool cand = live && !qf && (int)(l & F2Q_LEN_MASK) >= need;
            uint32_t forced = 0;
            if ((l & F2Q_LEN_FLAG) && cand) {                       // non-ACGT symbols in the window (rare): forced mismatches
                forced = fixed4_flags(g, r.q, j);
                if (forced && (!do_near || (forced & (forced - 1u)) != 0u)) cand = false;    // more of them than --m allows
            }
            w_reads += (uint32_t)__popcll(__ballot(live));
            w_qfail += (uint32_t)__popcll(__ballot(qf));
            w_nonal += (uint32_t)__popcll(__ballot(live && !qf && !cand));
            const uint64_t key = fixed4_key(g, r.b, j);
            const uint32_t part = pt_part((uint32_t)key & ((1u << hb0) - 1u), P);
            if (cand) {
                const uint32_t pos = atomicAdd(&cur[part], 1u);                              // ds_add_rtn_u32: a slot of this wave's ring
                ring[part * F2Q_PS_RING + (pos & (F2Q_PS_RING - 1u))] = pt_entry(key, forced, L);
            }
            // a ring that holds 64 entries or more is written out (it held < 64 before this round's <= 64 pushes)
            const uint32_t fill = (lane < P ? cur[lane] : 0u) - headv;
            unsigned long long full = __ballot(lane < P && fill >= 64u);
            while (full) {
                const uint32_t p = (uint32_t)__builtin_ctzll(full);
                full &= full - 1ull;
                write_out(p, (uint32_t)__builtin_amdgcn_readlane((int)headv, (int)p), 64u);
                if (lane == p) headv += 64u;
            }
        }
    };

    const uint32_t n_t = tile1 - tile0;
    uint32_t t = gw;
    const uint32_t last = n_t - 1u;
    if (PIPE) {
        Rows ra, rb;
        if (t < n_t) {
            request_tile(ra, tile0 + t);
            for (;;) {
                __builtin_amdgcn_sched_barrier(0);
                request_tile(rb, tile0 + min(t + n_waves, last));
                __builtin_amdgcn_sched_barrier(0);
                scatter_tile(ra);
                t += n_waves;
                if (t >= n_t) break;
                __builtin_amdgcn_sched_barrier(0);
                request_tile(ra, tile0 + min(t + n_waves, last));
                __builtin_amdgcn_sched_barrier(0);
                scatter_tile(rb);
                t += n_waves;
                if (t >= n_t) break;
            }
        }
    } else {
        for (; t < n_t; t += n_waves) { Rows r; request_tile(r, tile0 + t); scatter_tile(r); }
    }
    // what is left in the rings (< 64 entries each), and the stream lengths
    for (uint32_t p = 0; p < P; p++) {
        const uint32_t hp = (uint32_t)__builtin_amdgcn_readlane((int)headv, (int)p);
        const uint32_t n = cur[p] - hp;
        if (n) write_out(p, hp, n);
    }
    __syncthreads();
    if (tid < P) gpw(ps.cnt)[(uint64_t)blockIdx.x * P + tid] = wg_cur[tid];
    __syncthreads();
    const bool l0 = lane == 0;
    unsigned long long stv[5] = {l0 ? w_reads : 0u, 0u, 0u, l0 ? w_nonal : 0u, l0 ? w_qfail : 0u};
    flush_stats_add(stv, ps_smem, acc.stat_slab + (uint64_t)blockIdx.x * 8u);                // the rings are done with
}

// ---- pass 2 --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pt_count(uint32_t *cnt, uint32_t slot, const Accum &acc, const uint32_t *feat0_of)
{
    const uint32_t sh = (slot & 1u) << 4;
    const uint32_t old = atomicAdd(&cnt[slot >> 1], 1u << sh);
    if (((old >> sh) & 0xFFFFu) == 0x7FFFu) {               // see lt_count: the adder that sees 0x7FFF -> 0x8000 hands 0x8000 on
        atomicSub(&cnt[slot >> 1], 0x8000u << sh);
        acc_add(&acc.counts[gp(feat0_of)[slot]], 0x8000ull);
    }
}

#define F2Q_PC_RING 256u                     // entries of a wave's ring of reads without an exact hit (< 64 left over + <= 128 pushed per step)
#define F2Q_PC_VIA 128u                      // table-1 slots a wave collects before it adds them to hist1 (64 at a time)

// grid = K * n_parts workgroups: workgroup b serves partition b % n_parts as member b / n_parts of K.  Every entry is
// first asked for an exact hit in the partition's table 0 (two hashes, two 8-byte LDS reads: 85 % of a screen's reads
// end here); the others wait in a ring of the wave and are decided 64 at a time with every lane busy: table-0 and
// table-1 neighbours (lt_decide), the two table-1 buckets fetched from global memory.  A hit found through table 1
// belongs to a feature of another partition: it is counted by table-1 slot in hist1 (one 4-byte atomic without return
// per hit, 5 % of the reads).  slab0 row b receives the workgroup's histogram in gid order (rows are ADDED to: the
// launches of a block's rounds accumulate; k_part_reduce clears them).
template <bool NEAR>
__global__ __launch_bounds__(F2Q_PC_THREADS) void k_part_count(const RunDev *__restrict__ runp, const LibDev *__restrict__ libp,
                                                               Accum acc, PartScratch ps, uint32_t *__restrict__ slab0,
                                                               uint32_t *__restrict__ hist1)
{
    extern __shared__ uint32_t pc_smem[];
    uint32_t *tg = pc_smem;                                         // [F2Q_LT_SLOTS] tags of the partition's table 0
    uint32_t *cnt = pc_smem + F2Q_LT_SLOTS;                         // [F2Q_LT_BUCKETS] two u16 counters per word
    const LibDev &lib = *libp;
    const PtDesc pt = lib.pt;
    const uint32_t P = pt.n_parts, K = gridDim.x / P;
    const uint32_t p = blockIdx.x % P, m = blockIdx.x / P;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));    // wave-uniform for the compiler: loop control below stays on the scalar unit
    const uint32_t n_thr = blockDim.x, n_wv = n_thr >> 6;
    unsigned long long *ring = reinterpret_cast<unsigned long long *>(pc_smem + F2Q_LT_SLOTS + F2Q_LT_BUCKETS) + wave * F2Q_PC_RING;
    uint32_t *via = pc_smem + F2Q_LT_SLOTS + F2Q_LT_BUCKETS + 2u * n_wv * F2Q_PC_RING + wave * F2Q_PC_VIA;   // hits found through table 1
    uint32_t v_head = 0, v_tail = 0;
    LtDesc lt{};
    lt.hb0 = pt.hb0; lt.hb1 = pt.hb1; lt.len = pt.len;
    {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 F2Q_GLOBAL *src = (const v4 F2Q_GLOBAL *)(gp(pt.tags0) + (size_t)p * F2Q_LT_SLOTS);
        v4 *dst = reinterpret_cast<v4 *>(tg);
        for (uint32_t i = tid; i < F2Q_LT_SLOTS / 4u; i += n_thr) dst[i] = src[i];
        for (uint32_t i = tid; i < F2Q_LT_BUCKETS; i += n_thr) cnt[i] = 0;
    }
    __syncthreads();
    const auto tags1 = gp(pt.tags1);
    const uint32_t *feat0_of = pt.feat0_of + (size_t)p * F2Q_LT_SLOTS;
    const uint64_t kmask = (1ull << (2u * pt.len)) - 1ull;
    const uint32_t fmask = (1u << pt.len) - 1u, h0mask = (1u << pt.hb0) - 1u;
    uint32_t w_perfect = 0, w_imperfect = 0, w_seen = 0;
    uint32_t r_head = 0, r_tail = 0;                                // the ring belongs to this wave alone
    // this workgroup's streams: those of the scatter workgroups m, m + K, ... (their lengths sit in lanes 0, 1, ...);
    // its waves walk a stream together, wave v of W taking the steps v, v + W, ...: one sequential run per workgroup
    const uint32_t ns = m < ps.n_wg1 ? (ps.n_wg1 - m + K - 1u) / K : 0u;                        // <= 64 (the host sees to it)
    uint32_t nvec = 0;
    if (lane < ns) nvec = gp(ps.cnt)[(uint64_t)(m + lane * K) * P + p];

    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    struct Pos { uint32_t s, o, n; };
    auto n_of = [&](uint32_t s) { return s < ns ? (uint32_t)__builtin_amdgcn_readlane((int)nvec, (int)s) : 0u; };
    auto settle = [&](Pos &c) { while (c.s < ns && c.o >= c.n) { c.s++; c.o = wave * F2Q_PC_STEP; c.n = n_of(c.s); } };
    auto fetch = [&](const Pos &c) {
        const bool in = c.s < ns;                                   // past the end: any valid address (the step is not taken)
        const auto base = gp(ps.streams) + ((uint64_t)(m + (in ? c.s : 0u) * K) * P + p) * ps.cap + (in ? c.o : 0u) + 2u * lane;
        return __builtin_nontemporal_load((const v4 F2Q_GLOBAL *)base);
    };
    // The reads without an exact hit, 64 (or the last n) at a time, one per lane, in two halves a round apart: near_ask
    // takes the batch out of the ring and requests its two table-1 buckets; near_decide, after the next round's exact-hit
    // work, decides.  A wave's memory operations complete in issue order and the compiler must place waits that hold on
    // every path: so EVERY round issues the same operations in the same order -- two bucket loads (of bucket 0 when
    // there is no batch: one cache line for the wave), then the load of the entries three rounds ahead -- and every wait
    // count is exact: waiting for this round's entries or for a batch's buckets never also waits for younger loads.
    // Table 1 is asked for ONE bucket, the first-choice one: its first tag says whether any feature of that bucket had to
    // move to its second choice (PtDesc::spill; table 1 is kept at load 0.2, so fewer than 1 % of the buckets say so),
    // and only then is the second bucket read -- by the few lanes concerned, in the deciding half.
#ifdef F2Q_STAMP
    unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime(), t1_;
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); t1_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); tp[i] += t1_ - t0_; t0_ = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(i) do {} while (0)
#endif
    struct NearBatch { LtProbe q; U2 e[4]; uint32_t forced; uint32_t n; };
    auto near_ask = [&](NearBatch &nb, uint32_t n) {
        nb.n = n;
        uint32_t b2 = 0;
        if (n) {
            const unsigned long long ent = ring[(r_head + lane) & (F2Q_PC_RING - 1u)];
            nb.forced = lane < n ? (uint32_t)(ent >> (2u * pt.len)) & fmask : 0u;
            nb.q = lt_probe(lt, ent & kmask, pt.bb1);
            nb.e[0] = lds_u2(tg + 2u * nb.q.b[0]); nb.e[1] = lds_u2(tg + 2u * nb.q.b[1]);
            b2 = nb.q.b[2];
            r_head += n;
        }
        const v2 t2 = *(const v2 F2Q_GLOBAL *)(tags1 + 2u * b2);
        nb.e[2] = U2{t2.x, t2.y};
    };
    auto near_decide = [&](NearBatch &nb) {
        // No entry of the ring has an exact hit (step() looked, and a flagged read can have none), so the verdict is
        // "exactly one candidate at distance 1 among the eight tags" (lt_near1: every tag of the query's four buckets
        // that shares the bucketing half and differs in one base of the other; a flagged base is that one base).  With
        // 100 k features one read in six meets several features sharing one of its halves, and three in a hundred carry a
        // flagged base: the general routine IS the common case here, so it runs branch-free for the whole batch.
        nb.e[3] = U2{F2Q_LT_EMPTY, F2Q_LT_EMPTY};
        if (pt.spill) {
            const bool more = lane < nb.n && nb.e[2].x != F2Q_LT_EMPTY && (nb.e[2].x & pt.spill) != 0u;
            if (nb.e[2].x != F2Q_LT_EMPTY) nb.e[2].x &= ~pt.spill;  // (an empty first slot with the mark stays what it is: it equals no tag)
            PSTAMP(4);
            if (__ballot(more)) {
                if (more) { const v2 t3 = *(const v2 F2Q_GLOBAL *)(tags1 + 2u * nb.q.b[3]); nb.e[3] = U2{t3.x, t3.y}; }
                if (nb.e[3].x != F2Q_LT_EMPTY) nb.e[3].x &= ~pt.spill;   // (that bucket is some other half's first choice: its mark is not ours)
            }
        } else if (lane < nb.n) {
            const v2 t3 = *(const v2 F2Q_GLOBAL *)(tags1 + 2u * nb.q.b[3]); nb.e[3] = U2{t3.x, t3.y};
        }
        PSTAMP(5);
        uint32_t hit = 0, hitw = 0;
        const uint32_t ncand = lt_near1(lt, nb.q, nb.e, nb.forced, hit, hitw);
        const bool imp = lane < nb.n && ncand == 1u, far = imp && (hit >> 31) != 0u;
        if (imp && !far) pt_count(cnt, hit, acc, feat0_of);
        // a hit through table 1 is another partition's feature: its table-1 slot is noted in LDS and hist1 receives 64 of
        // them with one instruction (a global atomic in every batch would sit in the wave's queue of memory operations
        // behind the entries requested for the coming steps -- and make the next wait for entries wait for those too)
        PSTAMP(6);
        const unsigned long long fm = __ballot(far);
        if (far) via[(v_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))) & (F2Q_PC_VIA - 1u)] = hit & 0x7FFFFFFFu;
        v_tail += (uint32_t)__popcll(fm);
        if (v_tail - v_head >= 64u) {
            __hip_atomic_fetch_add(gpw(hist1) + via[(v_head + lane) & (F2Q_PC_VIA - 1u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v_head += 64u;
        }
        w_imperfect += (uint32_t)__popcll(__ballot(imp));
        PSTAMP(7);
    };
    auto step = [&](const v4 &ev, const Pos &c) {
        const unsigned long long ent[2] = {((unsigned long long)ev.y << 32) | ev.x, ((unsigned long long)ev.w << 32) | ev.z};
        uint32_t b0[2], b1[2], w0[2], w1[2]; U2 e0[2], e1[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const uint32_t h0 = (uint32_t)ent[a] & h0mask, h1 = (uint32_t)((ent[a] & kmask) >> pt.hb0);
            uint32_t c0, c1;
            lt_hash(h0, pt.hb0, pt.hb1, 0, b0[a], c0); lt_hash(h0, pt.hb0, pt.hb1, 1, b1[a], c1);
            w0[a] = lt_tag(c0, h1, pt.hb1); w1[a] = lt_tag(c1, h1, pt.hb1);
            e0[a] = lds_u2(tg + 2u * b0[a]); e1[a] = lds_u2(tg + 2u * b1[a]);
        }
        uint32_t old[2], osl[2]; bool hitv[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const bool valid = c.o + 2u * lane + (uint32_t)a < c.n;
            const bool forced = (uint32_t)(ent[a] >> (2u * pt.len)) != 0u;
            // a key sits in at most one slot (bucket + tag determine it): equality with the query's own tag is the exact hit
            const bool a0 = e0[a].x == w0[a], a1 = e0[a].y == w0[a], c0 = e1[a].x == w1[a], c1 = e1[a].y == w1[a];
            const bool hit = valid && !forced && (a0 | a1 | c0 | c1);
            const uint32_t slot = (a0 | a1) ? 2u * b0[a] + (uint32_t)a1 : 2u * b1[a] + (uint32_t)c1;
            // the u16 counter of the slot (pt_count), its overflow looked at further down: the adds of both entries and the
            // ring pushes are on their way before anything waits for what an add returned
            hitv[a] = hit; osl[a] = slot; old[a] = 0;
            if (hit) old[a] = atomicAdd(&cnt[slot >> 1], 1u << ((slot & 1u) << 4));
            const unsigned long long hm = __ballot(hit), vm = __ballot(valid);
            w_perfect += (uint32_t)__popcll(hm);
            w_seen += (uint32_t)__popcll(vm);
            if (NEAR) {
                const unsigned long long pm = vm & ~hm;                 // no exact hit: into the ring
                const uint32_t at = r_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                if (valid && !hit) ring[at & (F2Q_PC_RING - 1u)] = ent[a];
                r_tail += (uint32_t)__popcll(pm);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const uint32_t sh = (osl[a] & 1u) << 4;
            if (hitv[a] && ((old[a] >> sh) & 0xFFFFu) == 0x7FFFu) {     // this add took the counter to 0x8000: hand 0x8000 on (pt_count)
                atomicSub(&cnt[osl[a] >> 1], 0x8000u << sh);
                acc_add(&acc.counts[gp(feat0_of)[osl[a]]], 0x8000ull);
            }
        }
    };
    // one round: ask for a batch of the ring (or for nothing), request the entries three rounds ahead, take this round's
    // exact hits, decide the batch asked for in the last round; a ring that still holds 128 entries or more (a stretch of
    // reads without exact hits) is brought below that before the next round pushes up to 128 more
    auto round = [&](const v4 &cur, const Pos &c, v4 &nxt, const Pos &nx, NearBatch &ask, NearBatch &due) {
        PSTAMP(3);
        if (NEAR) near_ask(ask, r_tail - r_head >= 64u ? 64u : 0u);
        __builtin_amdgcn_sched_barrier(0);
        nxt = fetch(nx);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(2);
#ifdef F2Q_STAMP
        asm volatile("" :: "v"(cur.x), "v"(cur.w));                 // (the wait for this round's entries lands here)
        PSTAMP(0);
#endif
        step(cur, c);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(1);
        if (NEAR && due.n) near_decide(due);
        if (NEAR) while (r_tail - r_head >= 128u) { NearBatch nb2; near_ask(nb2, 64u); near_decide(nb2); }
        PSTAMP(7);
    };

    // four register sets of entries in rotation: the set requested in one round is consumed three rounds later
    auto next_pos = [&](const Pos &c) { Pos nx = c; nx.o += n_wv * F2Q_PC_STEP; settle(nx); return nx; };
    Pos pq[4]; v4 eq[4]; NearBatch nbq[2];
    nbq[0].n = 0; nbq[1].n = 0;
    pq[0] = Pos{0u, wave * F2Q_PC_STEP, n_of(0u)};
    settle(pq[0]);
    pq[1] = next_pos(pq[0]); pq[2] = next_pos(pq[1]); pq[3] = pq[2];
    if (pq[0].s < ns) {
        eq[0] = fetch(pq[0]); eq[1] = fetch(pq[1]); eq[2] = fetch(pq[2]); eq[3] = eq[2];
        bool more = true;
        while (more) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (more) {
                    pq[(k + 3) & 3] = next_pos(pq[(k + 2) & 3]);
                    round(eq[k], pq[k], eq[(k + 3) & 3], pq[(k + 3) & 3], nbq[k & 1], nbq[(k + 1) & 1]);
                    more = pq[(k + 1) & 3].s < ns;
                    if (!more && NEAR && nbq[k & 1].n) near_decide(nbq[k & 1]);       // the batch asked for in the last round
                }
            }
        }
    }
#ifdef F2Q_STAMP
    if (lane == 0 && acc.stamp) for (int i = 0; i < 8; i++) atomicAdd(&acc.stamp[i], tp[i]);
#endif
    if (NEAR) while (r_tail != r_head) { NearBatch nb; const uint32_t left = r_tail - r_head; near_ask(nb, left < 64u ? left : 64u); near_decide(nb); }
    if (NEAR && lane < v_tail - v_head)
        __hip_atomic_fetch_add(gpw(hist1) + via[(v_head + lane) & (F2Q_PC_VIA - 1u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    {
        // the histogram in gid order, added to this workgroup's slab row
        const uint32_t g0 = gp(pt.pstart)[p], gn = gp(pt.pstart)[p + 1u] - g0;
        auto row = gpw(slab0) + (uint64_t)blockIdx.x * pt.max_part;
        for (uint32_t i = tid; i < gn; i += n_thr) {
            const uint32_t s = gp(pt.slot0_of)[g0 + i];
            const uint32_t v = (cnt[s >> 1] >> ((s & 1u) << 4)) & 0xFFFFu;
            if (v) row[i] += v;
        }
    }
    __syncthreads();
    const bool l0 = lane == 0;
    unsigned long long stv[5] = {0u, l0 ? w_perfect : 0u, l0 ? w_imperfect : 0u, l0 ? w_seen - w_perfect - w_imperfect : 0u, 0u};
    flush_stats_add(stv, reinterpret_cast<unsigned long long *>(pc_smem), acc.stat_slab + (uint64_t)blockIdx.x * 8u);
}

// counts[feature of gid] += its partition's slab-0 rows + the count of its table-1 slot; both are cleared on the way.
// Block = 64 gids x 4 row lanes (the K rows of a partition are read by four threads per gid, eight loads in flight each).
__global__ __launch_bounds__(256) void k_part_reduce(const LibDev *__restrict__ libp, uint32_t *__restrict__ slab0, uint32_t K,
                                                     uint32_t *__restrict__ hist1, unsigned long long *__restrict__ counts,
                                                     unsigned long long *__restrict__ stat_rows, uint32_t n_stat_rows,
                                                     unsigned long long *__restrict__ stats)
{
    __shared__ unsigned long long part[256];
    if (blockIdx.x == 0) {                                          // the five reference counters: rows of 8, cleared on the way
        const uint32_t k = threadIdx.x & 7u, sub = threadIdx.x >> 3;
        unsigned long long sv = 0;
        if (k < 5) for (uint32_t w = sub; w < n_stat_rows; w += 32u) { const unsigned long long v = stat_rows[(uint64_t)w * 8u + k]; if (v) { sv += v; stat_rows[(uint64_t)w * 8u + k] = 0; } }
        part[threadIdx.x] = sv;
        __syncthreads();
        if (threadIdx.x < 5) {
            unsigned long long tot = 0;
            for (uint32_t q = 0; q < 32u; q++) tot += part[q * 8u + threadIdx.x];
            if (tot) atomicAdd(&stats[threadIdx.x], tot);
        }
        __syncthreads();
    }
    const PtDesc pt = libp->pt;
    const uint32_t fx = threadIdx.x & 63u, ry = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * 64u + fx, nf = libp->n_features;
    unsigned long long sum = 0;
    if (gid < nf) {
        uint32_t p = 0;
        while (p + 1u < pt.n_parts && gid >= gp(pt.pstart)[p + 1u]) p++;
        const uint32_t i = gid - gp(pt.pstart)[p];
#pragma unroll 8
        for (uint32_t m = ry; m < K; m += 4u) {
            uint32_t F2Q_GLOBAL *at = gpw(slab0) + (uint64_t)(m * pt.n_parts + p) * pt.max_part + i;
            const uint32_t v = *at;
            if (v) { sum += v; *at = 0u; }
        }
        if (ry == 0 && hist1) {
            uint32_t F2Q_GLOBAL *at = gpw(hist1) + gp(pt.slot1_of)[gid];
            const uint32_t v = *at;
            if (v) { sum += v; *at = 0u; }
        }
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (ry == 0 && gid < nf) {
        sum = part[fx] + part[64 + fx] + part[128 + fx] + part[192 + fx];
        if (sum) atomicAdd(&counts[gp(pt.feat_of)[gid]], sum);
    }
}
