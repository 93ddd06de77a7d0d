// f2q_aux_kernels.h -- kernels around the counting path (included by f2q_lib.hip only): Extract+Count table
// growth, the synthetic workload generator, and the device-side FASTQ ingest (framing + classification + packing).
#pragma once

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
    // entries keep their number and their place in the arena (the host copies both wholesale and sets the counters:
    // a device-wide counter bumped per key is one address for everybody); only the slots are re-derived.  Keys are
    // distinct, so plain claim-by-CAS of an empty slot is enough
    const unsigned long long ne = e;
    (void)nwords;
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
    int n_win, win_len, win_end, win_start[F2Q_MW_MAX];   // multi-window runs: only the windows are stored (PackPlan::n_win)
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
    if (!dirty && o.n_win && R < o.win_end) dirty = true;        // a window the read ends in
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
    uint32_t *bp = o.bases + (tile * o.wb) * F2Q_TILE + lane;
    uint32_t *qp = o.qual + (tile * o.wq) * F2Q_TILE + lane;
    if (o.n_win) {                               // several windows: their bases and quality bytes back to back, nothing else
        const int S = o.n_win * o.win_len;
        uint64_t fw = 0; int fwi = -1; bool fl = false;
        uint32_t bw = 0, qw = 0;
        for (int k = 0; k < S; k++) {
            const int p = o.win_start[k / o.win_len] + k % o.win_len;
            if ((p >> 5) != fwi) { fwi = p >> 5; fw = rnd(s.seed, i, F_FLANK0 + fwi); }
            uint32_t code = base_code(synth_base(s, r, p, fw)); if (code > 3u) code = 0;
            fl |= p == npos;
            qw |= ((uint32_t)((p == r.qpos) ? r.qchar : (uint8_t)'I') | (p == npos ? 0x80u : 0u)) << (8 * (k & 3));
            if ((k & 3) == 3 || k == S - 1) { qp[(uint64_t)(k >> 2) * F2Q_TILE] = qw; qw = 0; }
            bw |= code << (2 * (k & 15));
            if ((k & 15) == 15 || k == S - 1) { bp[(uint64_t)(k >> 4) * F2Q_TILE] = bw; bw = 0; }
        }
        if (o.len) o.len[slot] = (uint16_t)((uint32_t)S | (fl ? F2Q_LEN_FLAG : 0u));
        return;
    }
    if (o.len) o.len[slot] = (uint16_t)((uint32_t)R | (npos >= 0 ? F2Q_LEN_FLAG : 0u));
    uint64_t fw = 0;
    uint32_t bw = 0, qw = 0, lw = 0, hw = 0;
    for (int p = 0; p < R; p++) {
        if ((p & 31) == 0) fw = rnd(s.seed, i, F_FLANK0 + (p >> 5));
        uint8_t c = synth_base(s, r, p, fw);
        uint32_t code = base_code(c); if (code > 3u) code = 0;
        if (!o.planar_nw) {
            qw |= ((uint32_t)((p == r.qpos) ? r.qchar : (uint8_t)'I') | (p == npos ? 0x80u : 0u)) << (8 * (p & 3));
            if ((p & 3) == 3 || p == R - 1) { qp[(uint64_t)(p >> 2) * F2Q_TILE] = qw; qw = 0; }
        }
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
    if (o.planar_nw) {                           // quality bytes of planar tiles: transposed 32-base groups (planar_qpos)
        const uint32_t nq = 8u * (((uint32_t)R + 31u) / 32u);
        for (uint32_t w = 0; w < nq; w++) {
            uint32_t v = 0;
            for (uint32_t j = 0; j < 4; j++) {
                const int p = (int)planar_qpos(w, j);
                if (p < R) v |= ((uint32_t)((p == r.qpos) ? r.qchar : (uint8_t)'I') | (p == npos ? 0x80u : 0u)) << (8 * j);
            }
            qp[(uint64_t)w * F2Q_TILE] = v;
        }
    }
}

// ---- exclusive prefix sum of u32 (the two sums of the ingest: newlines per chunk, clean reads per record) -------------
// Three small launches: every workgroup scans its block of F2Q_SCAN_BLOCK values in LDS and leaves the block total,
// one workgroup scans the totals, the third launch adds them back.  n is a few million at most (one value per 4 KiB of
// text or per record), so this is microseconds; out[i] = in[0] + ... + in[i-1].
#define F2Q_SCAN_THREADS 256
#define F2Q_SCAN_ITEMS 8
#define F2Q_SCAN_BLOCK (F2Q_SCAN_THREADS * F2Q_SCAN_ITEMS)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds /* [F2Q_SCAN_THREADS / 64 + 1] */, uint32_t &total)
{
    // inclusive scan inside the wave by shuffles, wave totals through LDS
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= (uint32_t)off) x += y; }
    if (lane == 63u) lds[wave] = x;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < F2Q_SCAN_THREADS / 64; w++) { const uint32_t t = lds[w]; if (w < wave) before += t; tot += t; }
    __syncthreads();
    total = tot;
    return before + x - v;
}
__global__ __launch_bounds__(F2Q_SCAN_THREADS) void k_scan_blocks(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n,
                                                                  uint32_t *__restrict__ block_sums)
{
    __shared__ uint32_t lds[F2Q_SCAN_THREADS / 64 + 1];
    const uint32_t base = blockIdx.x * F2Q_SCAN_BLOCK + threadIdx.x * F2Q_SCAN_ITEMS;
    uint32_t v[F2Q_SCAN_ITEMS], mine = 0;
#pragma unroll
    for (int i = 0; i < F2Q_SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : 0u; mine += v[i]; }
    uint32_t total;
    uint32_t run = block_exclusive_scan(mine, lds, total);
#pragma unroll
    for (int i = 0; i < F2Q_SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(F2Q_SCAN_THREADS) void k_scan_sums(uint32_t *__restrict__ block_sums, uint32_t n_blocks)
{
    __shared__ uint32_t lds[F2Q_SCAN_THREADS / 64 + 1];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += F2Q_SCAN_THREADS) {
        const uint32_t i = b0 + threadIdx.x, v = i < n_blocks ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, lds, total);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += total;
    }
}
__global__ __launch_bounds__(F2Q_SCAN_THREADS) void k_scan_add(uint32_t *__restrict__ out, uint32_t n, const uint32_t *__restrict__ block_sums)
{
    const uint32_t add = block_sums[blockIdx.x];
    const uint32_t base = blockIdx.x * F2Q_SCAN_BLOCK + threadIdx.x * F2Q_SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < F2Q_SCAN_ITEMS; i++) if (base + i < n) out[base + i] += add;
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
    __shared__ uint32_t lds[F2Q_SCAN_THREADS / 64 + 1];
    const uint64_t pos = (uint64_t)blockIdx.x * F2Q_NL_CHUNK + threadIdx.x * 16u;
    const uint32_t c = __popc(nl_mask16(gp(text), pos, nbytes));
    uint32_t tot;
    (void)block_exclusive_scan(c, lds, tot);
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = tot;
}

// line_start[k] = offset of line k; line_start[n_newlines + 1] = nbytes + 1 (end sentinel for an unterminated last line)
__global__ __launch_bounds__(256) void k_line_starts(const uint8_t *text, uint64_t nbytes, const uint32_t *chunk_prefix,
                                                      uint32_t *line_start)
{
    __shared__ uint32_t lds[F2Q_SCAN_THREADS / 64 + 1];
    const uint64_t pos = (uint64_t)blockIdx.x * F2Q_NL_CHUNK + threadIdx.x * 16u;
    uint32_t m = nl_mask16(gp(text), pos, nbytes);
    uint32_t unused_total;
    const uint32_t before = block_exclusive_scan((uint32_t)__popc(m), lds, unused_total);
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

template <class P>
__device__ __forceinline__ uint32_t rstrip_dev(P p, uint32_t n)
{
    while (n > 0) {
        const uint8_t c = p[n - 1];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c) n--; else break;
    }
    return n;
}

// The records of a wave are neighbours in the text, so the wave first copies the stretch that holds all 64 of them
// into LDS with 16-byte loads (every byte of the text is fetched once, coalesced) and the lanes then walk their
// records there.  One thread per record walking global memory byte by byte re-fetched every 128-byte line dozens of
// times (r02 PMC: 47x and 68x the text for k_classify / k_pack).  A stretch longer than the staging area (reads of
// several hundred bases) is walked in global memory as before.
#define F2Q_ING_THREADS 64
#define F2Q_ING_LDS (40u * 1024u)
// text[lo, hi) -> lds; byte x of the text is then at lds[x - (lo & ~15)]
__device__ __forceinline__ void stage_span(uint8_t *lds, gbytes text, uint32_t lo, uint32_t hi)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63u, a0 = lo & ~15u, a1 = hi & ~15u;
    const v4 F2Q_GLOBAL *src = (const v4 F2Q_GLOBAL *)(text + a0);
    v4 *dst = reinterpret_cast<v4 *>(lds);
    const uint32_t ng = a1 > a0 ? (a1 - a0) >> 4 : 0u;
    for (uint32_t g = lane; g < ng; g += 64u) dst[g] = src[g];
    for (uint32_t x = (a1 > a0 ? a1 : a0) + lane; x < hi; x += 64u) lds[x - a0] = text[x];   // the last < 16 bytes, never past hi
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__global__ __launch_bounds__(F2Q_ING_THREADS) void k_classify(IngestDev d, PackPlan pl)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[F2Q_ING_LDS];
    const uint32_t r0 = blockIdx.x * F2Q_ING_THREADS, r = r0 + threadIdx.x;
    const uint32_t n_act = d.n_records - r0 < F2Q_ING_THREADS ? d.n_records - r0 : F2Q_ING_THREADS;
    const bool live = r < d.n_records;
    const auto ls = gp(d.line_start);
    const uint32_t rr = live ? r : d.n_records - 1u;
    const uint32_t s0 = ls[4u * rr + 1u], e0 = ls[4u * rr + 2u] - 1u, s1 = ls[4u * rr + 3u], e1 = ls[4u * rr + 4u] - 1u;
    const uint32_t lo = __shfl(s0, 0, 64), hi = __shfl(e1, (int)n_act - 1, 64);
    const bool fits = hi >= lo && hi - (lo & ~15u) <= F2Q_ING_LDS;
    bool clean = false; uint32_t len = 0, qlen = 0, plen = 0;
    if (fits) {
        stage_span(stage, gp(d.text), lo, hi);
        if (live) {
            const uint32_t a0 = lo & ~15u;
            RecT<const uint8_t *> rec;
            rec.seq = stage + (s0 - a0); rec.qual = stage + (s1 - a0);
            rec.len = rstrip_dev(rec.seq, e0 - s0); rec.qlen = rstrip_dev(rec.qual, e1 - s1);
            clean = read_is_clean(pl, rec); len = rec.len; qlen = rec.qlen; plen = packed_len(pl, rec);
        }
    } else if (live) {
        RecT<gbytes> rec;
        rec.seq = gp(d.text) + s0; rec.qual = gp(d.text) + s1;
        rec.len = rstrip_dev(rec.seq, e0 - s0); rec.qlen = rstrip_dev(rec.qual, e1 - s1);
        clean = read_is_clean(pl, rec); len = rec.len; qlen = rec.qlen; plen = packed_len(pl, rec);
    }
    if (live) {
        gpw(d.r_off)[r] = s0; gpw(d.r_len)[r] = len; gpw(d.r_qoff)[r] = s1; gpw(d.r_qlen)[r] = qlen;
        gpw(d.clean)[r] = clean ? 1u : 0u;
    }
    // the longest packed read: one atomic per wave
    uint32_t m = (live && clean) ? plen : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(m, off, 64); m = o > m ? o : m; }
    if (threadIdx.x == 0 && m) atomicMax(&d.meta[0], m);
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
__global__ __launch_bounds__(F2Q_ING_THREADS) void k_pack(IngestDev d, PackPlan pl, const uint32_t *clean_before, PackOut o)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[F2Q_ING_LDS];
    const uint32_t r0 = blockIdx.x * F2Q_ING_THREADS, r = r0 + threadIdx.x;
    const uint32_t n_act = d.n_records - r0 < F2Q_ING_THREADS ? d.n_records - r0 : F2Q_ING_THREADS;
    const bool live = r < d.n_records;
    const uint32_t rr = live ? r : d.n_records - 1u;
    const uint32_t off = gp(d.r_off)[rr], qoff = gp(d.r_qoff)[rr], len = gp(d.r_len)[rr], qlen = gp(d.r_qlen)[rr];
    const bool clean = live && gp(d.clean)[rr] != 0u;
    const uint32_t slot = gp(clean_before)[rr];
    // the stretch of text the wave's clean records lie in (none clean: nothing to read)
    const unsigned long long cm = __ballot(clean);
    bool fits = false; uint32_t lo = 0;
    if (cm) {
        lo = __shfl(off, __builtin_ctzll(cm), 64);
        const uint32_t hi = __shfl(qoff + qlen, 63 - __builtin_clzll(cm), 64);
        (void)n_act;
        fits = hi >= lo && hi - (lo & ~15u) <= F2Q_ING_LDS;
        if (fits) stage_span(stage, gp(d.text), lo, hi);
    }
    if (!live) return;
    if (clean) {
        const uint64_t tile = slot / F2Q_TILE, lane = slot % F2Q_TILE;
        DevSink sink{gpw(o.bases) + tile * o.wb * F2Q_TILE + lane, gpw(o.qual) + tile * o.wq * F2Q_TILE + lane,
                     gpw(o.len) + tile * F2Q_TILE + lane};
        if (fits) {
            const uint32_t a0 = lo & ~15u;
            RecT<const uint8_t *> rec; rec.seq = stage + (off - a0); rec.qual = stage + (qoff - a0); rec.len = len; rec.qlen = qlen;
            pack_read(pl, rec, o.planar_nw, sink);
        } else {
            RecT<gbytes> rec; rec.seq = gp(d.text) + off; rec.qual = gp(d.text) + qoff; rec.len = len; rec.qlen = qlen;
            pack_read(pl, rec, o.planar_nw, sink);
        }
        if (o.c_index) gpw(o.c_index)[slot] = r;
    } else {
        const uint32_t g = r - slot;
        gpw(o.g_off)[g] = off; gpw(o.g_qoff)[g] = qoff;
        gpw(o.g_len)[g] = len; gpw(o.g_qlen)[g] = qlen; gpw(o.g_index)[g] = r;
    }
}

