// f2q_device.h -- per-read logic of the counting path, written once as host/device inline
// functions.  The HIP kernels (f2q_count_kernels.h) call these per lane; tests/emu compiles the same
// functions with g++ to unit-test the lane logic on a machine without a GPU (test infrastructure
// only -- the product never executes them on the host).
//
// Reference semantics restated here (fast2q/fast2q.py): window extraction :349-355, Phred rule
// :1112-1129/:357, anchored search :215-285 + :628-658, exact hit :365-367, unique-nearest
// mismatch search :692-750 + :660-690, counters :310-316/:366-393, EC dict :382-387.
#pragma once
#include <stdint.h>

#ifndef F2Q_HD
#ifdef __HIPCC__
#define F2Q_HD __host__ __device__ __forceinline__
#else
#define F2Q_HD inline
#endif
#endif

#define F2Q_DEV_MAX_ITER 16
#define F2Q_ANCHOR_MAX 128
#define F2Q_MAX_PIECES 8
#define F2Q_REG_MAXLEN 31          // longest feature/window handled as a 2-bit u64 key
#define F2Q_TILE 256               // reads per packed tile (= threads per workgroup)
#define F2Q_LEN_SKIP 0xFFFFu       // len-plane marker: slot handled by the general path
#define F2Q_LEN_FLAG 0x8000u       // len-plane bit: the window holds non-ACGT symbols, marked by bit 7 of their quality bytes
#define F2Q_LEN_CASE 0x4000u       // ... (anchored runs, with F2Q_LEN_FLAG) the marked bases are LOWER-CASE ACGT, nothing else: they can match no
                                   // anchor symbol (the anchor search is case-sensitive, fast2q.py:337) but the window is upper-cased
                                   // (:354), so for the KEY they are ordinary bases (their codes are stored) and the marks are ignored
#define F2Q_LEN_MASK 0x3FFFu       // the read length in a len-plane entry

namespace f2q {

// Pointers that arrive inside by-value structs or from memory are "generic" to the compiler, which
// then emits flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) per access (flat ops complete out of order),
// i.e. one memory round trip at a time.  gp()/gpw() re-type them as global (address space 1) so the
// loads become global_load_* and stay in flight together.  On the host they are the identity.
#if defined(__HIP_DEVICE_COMPILE__)
#define F2Q_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ const T F2Q_GLOBAL *gp(const T *p) { return (const T F2Q_GLOBAL *)p; }
template <class T> __device__ __forceinline__ T F2Q_GLOBAL *gpw(T *p) { return (T F2Q_GLOBAL *)p; }
#else
#define F2Q_GLOBAL
template <class T> F2Q_HD const T *gp(const T *p) { return p; }
template <class T> F2Q_HD T *gpw(T *p) { return p; }
#endif
typedef const uint8_t F2Q_GLOBAL *gbytes;      // read-only bytes in device global memory

static const uint64_t KEY_EMPTY = ~0ull;

// ---------------------------------------------------------------------------------------------
// run parameters (device copy of f2q_params after set-up)
// ---------------------------------------------------------------------------------------------
struct RunDev {
    int32_t mode, miss, length, fixed, n_iter;
    int32_t thr, thr_up, thr_down;     // a quality byte c fails iff 33 <= c <= thr  (thr < 33: never)
    int32_t msu, msd, has_up, has_down;
    int32_t start[F2Q_DEV_MAX_ITER];
    int32_t up_len[F2Q_DEV_MAX_ITER], down_len[F2Q_DEV_MAX_ITER];
    uint8_t up[F2Q_DEV_MAX_ITER][F2Q_ANCHOR_MAX];
    uint8_t down[F2Q_DEV_MAX_ITER][F2Q_ANCHOR_MAX];
    // packed anchored path (one --us/--ds pair, ACGT-only anchors of 1..32 bases): 2-bit codes of pair 0
    int32_t anchors_packed;            // 1: up_codes/down_codes are valid
    int32_t compact;                   // 1: several --st windows whose bases the tiles hold back to back, window w at stored position w * length (PackPlan::n_win)
    uint64_t up_codes, down_codes;     // symbol j = (codes >> 2j) & 3  -- one scalar load, no per-symbol memory access
    uint32_t up_pos[4], down_pos[4];   // per symbol c: bit j set iff anchor symbol j == c
    // several --us/--ds pairs, every anchor ACGT-only and 1..32 long: the same masks per pair (k_count_anchor_pairs)
    int32_t pairs_packed, pad2_;
    uint32_t mp_up_pos[F2Q_DEV_MAX_ITER][4], mp_down_pos[F2Q_DEV_MAX_ITER][4];
};

// ---------------------------------------------------------------------------------------------
// library index
// ---------------------------------------------------------------------------------------------
struct PieceDesc { uint32_t off, bits, shift, pad; uint64_t mask; };   // table = 1<<bits slots at tab[off]
struct LenGroup { uint32_t n, n_pieces; PieceDesc exact; PieceDesc piece[F2Q_MAX_PIECES]; };

// "packed" tables for the one feature length a fixed-offset run looks up (v2 fast kernel): a slot is
// (2-bit key << ib) | feature index, so one 8-byte load resolves a probe; hashing is 32-bit.
struct PackedPiece { uint32_t off, bits, shift, pad; uint64_t mask; };
struct PackedGroup {
    uint32_t len;                      // feature length these tables index (0: not built)
    uint32_t ib;                       // index bits
    uint32_t n_pieces, pad;
    PackedPiece exact;
    PackedPiece piece[F2Q_MAX_PIECES];
};

// "LDS tables" of a uniform library of 14..21-base features searched with --m <= 1 (see the section further down):
// two cuckoo tables of 32-bit tags small enough for a workgroup to keep in LDS next to its histogram.
#define F2Q_LT_BBITS 13u
#define F2Q_LT_BUCKETS (1u << F2Q_LT_BBITS)       // per table; a bucket is two slots (one 8-byte LDS read)
#define F2Q_LT_SLOTS (2u * F2Q_LT_BUCKETS)
#define F2Q_LT_EMPTY 0xFFFFFFFFu
struct LtDesc {
    uint32_t ok;                       // 1: tables built for this library
    uint32_t len;                      // feature length L
    uint32_t hb0, hb1;                 // bits of half 0 (bases [0, L/2)) and half 1 (the rest)
    uint32_t mix, pad_;                // > 0: two-window runs, keys are looked up as mw_mix(key, mix) (mix = bases per window)
    const uint32_t *tags;              // [2][F2Q_LT_SLOTS]: table t is bucketed by half t and stores the other half in its tags
    const uint16_t *slot_of;           // [n_features] table-0 slot of a feature = its counter in the LDS histogram
    const uint32_t *feat_of;           // [F2Q_LT_SLOTS] feature of a table-0 slot (unused slots: 0)
};

// "Partitioned tables": the same tag tables for a uniform library too large for one workgroup's LDS (BASELINE config 4:
// 100 k guides).  The features are dealt into n_parts partitions by a hash of their half 0; every partition has its own
// table 0 (F2Q_LT_SLOTS tags: what one workgroup holds in LDS next to its u16 histogram), so exact hits and the
// neighbours that share the query's half 0 are decided in LDS by the workgroups of the query's partition.  Neighbours
// that share half 1 can sit in any partition: ONE table 1 over all features stays in global memory (2 << bb1 tags, L2
// resident) and is probed only by the reads without an exact hit.  Features are renumbered partition by partition
// ("gid"); a hit is counted by table-0 slot inside its partition or, when found through table 1, by table-1 slot.
#define F2Q_PT_MAXP 32u
#define F2Q_PT_FILL 12800u         // features per partition the builder aims at (of F2Q_LT_SLOTS = 16384 tag slots)
#define F2Q_PT_SPILL (1u << 30)    // PtDesc::spill when the tags leave bit 30 free
#define F2Q_PT_EMPTY_SPILL 0xFFFFFFFEu   // an empty first slot of a bucket that carries the mark (equals no tag: the free bits are set)
struct PtDesc {
    uint32_t ok, n_parts;              // ok = 1: tables built
    uint32_t len, hb0, hb1, bb1;       // feature length, bits of the halves, bucket bits of table 1
    uint32_t max_part;                 // features of the largest partition
    uint32_t spill;                    // 0, or the bit of a table-1 bucket's FIRST tag that says "a feature whose first-choice bucket this is
                                       // sits in its second-choice bucket": clear = the second bucket need not be read
    const uint32_t *tags0;             // [n_parts][F2Q_LT_SLOTS]
    const uint32_t *tags1;             // [2 << bb1]
    const uint32_t *pstart;            // [n_parts + 1]: partition p holds gids pstart[p] .. pstart[p + 1] - 1
    const uint16_t *slot0_of;          // [n_features] by gid: table-0 slot inside the feature's partition
    const uint32_t *slot1_of;          // [n_features] by gid: table-1 slot
    const uint32_t *feat_of;           // [n_features] by gid: index of the feature in the library
    const uint32_t *feat0_of;          // [n_parts][F2Q_LT_SLOTS]: library index of the feature in a table-0 slot (u16 counter hand-off)
};

// "General keys": an index of ALL features as byte strings, by length -- what the byte-exact routine looks keys up in
// that are no plain ACGT string of <= 31 bases (':'-joined multi-part keys, long features, odd symbols), and every key of
// a run whose library holds such features.  Per length group an exact table (hash of all bytes) and m+1 pigeonhole
// piece tables (hash of a byte range); entries are feature ids + 1, candidates are verified on the feature bytes.
#define F2Q_GK_MAXP 8
struct GkGroup {
    uint32_t len, n, ids_off;          // feature length, features of that length, their ids at ids[ids_off ..]
    uint32_t exact_off, bits;          // tables of 1 << bits slots at tab[exact_off], tab[piece_off[p]]
    uint32_t n_pieces;                 // 0: no pigeonhole tables (--m 0, or fewer bytes than pieces: the group is scanned)
    uint32_t piece_off[F2Q_GK_MAXP];
    uint32_t cut[F2Q_GK_MAXP + 1];     // piece p = bytes [cut[p], cut[p+1])
};
// fw / fwoff: every feature's bytes again as little-endian 8-byte words, zero padded (feature f at fw[fwoff[f]]): a
// candidate is checked with (len + 7) / 8 independent loads instead of a byte loop with an early exit
#define F2Q_GK_MAXW 13             // keys of up to 104 bytes take the word path
struct GkDesc { uint32_t n_groups, pad; const GkGroup *grp; const uint32_t *tab; const uint32_t *ids;
                const unsigned long long *fw; const uint32_t *fwoff; };

// multi-window runs (--st a,b,...): a key is the ':'-joined windows that passed their Phred test (fast2q.py:349-363);
// features made of k ACGT runs of --l bases joined by ':' ("k-part features") are indexed by their k*l bases
#define F2Q_MW_MAX 4               // windows of a run the packed path handles (k * l <= 31 bases)
// "Pair tables": runs with two --us/--ds pairs (k_count_anchor_pairs) against a library whose features ALL read A:B with
// A of la and B of lb ACGT bases (la, lb <= 20) -- dual-guide libraries, also combinatorial ones.  The joined key is held
// as two 2-bit words instead of a string; three open-addressing tables of 16-byte slots {A | feature << 40, B}: one hashed
// by (A, B) for the exact hit, one by A and one by B for --m 1 (a feature at distance 1 from the key agrees with it on
// exactly one of the two parts; the features sharing that part sit along its probe sequence).  Empty slot: B = ~0.
struct PwDesc {
    uint32_t ok, la, lb, bits;         // every table has 1 << bits slots
    const uint64_t *tab;               // [3][1 << bits][2]
};

struct LibDev {
    uint32_t n_features, n_irregular;
    const uint64_t *ptab;              // packed slots (KEY_EMPTY = free)
    PackedGroup pk;
    PackedGroup mpk[F2Q_MW_MAX];       // multi-window runs: packed tables of the k-part features, k = index + 1 (len = k * l)
    uint32_t mw_ok, mw_pad;            // 1: every feature a multi-window key can equal or approach is a k-part feature
    LtDesc lt;
    PtDesc pt;
    GkDesc gk;
    PwDesc pw;
    const uint64_t *tab_keys;          // open-addressing slots: 2-bit feature key or KEY_EMPTY
    const uint32_t *tab_idx;           // feature index of the slot
    const uint8_t *feat_bytes;         // all features, raw (upper-case) bytes
    const uint32_t *feat_off;          // n_features + 1
    const uint32_t *irr_ids;           // features that are not ACGT-only / longer than 31
    LenGroup grp[F2Q_REG_MAXLEN + 1];  // by feature length
};

// ---------------------------------------------------------------------------------------------
// accumulators
// ---------------------------------------------------------------------------------------------
struct Accum {
    unsigned long long *counts;        // [n_features]
    unsigned long long *stats;         // [5]
    uint32_t *slab;                    // [gridDim.x][n_features] per-workgroup histograms (or nullptr)
    unsigned long long *stat_slab;     // [gridDim.x][8] per-workgroup stats (or nullptr)
    unsigned long long *stamp;         // diagnostic builds (-DF2Q_STAMP): per-phase cycle sums, else nullptr
    uint32_t *hit_buf;                 // large libraries: feature index per read slot (0xFFFFFFFF = none), histogrammed by k_hist_ranges
};

struct EcDev {
    unsigned long long *slots;         // 0 empty | (fp<<32)|0xFFFFFFFF locked | (fp<<32)|(entry+1)
    uint32_t mask;                     // slots - 1
    uint32_t max_entries;
    unsigned long long *ent_off;       // arena offset of the key (in 4-byte words)
    uint32_t *ent_len;                 // key length in bytes
    unsigned long long *ent_count;
    unsigned long long *ent_first;     // min global read index that produced the key
    uint32_t *arena;                   // key bytes, 4 per word, little endian
    unsigned long long arena_words;
    unsigned long long *ctr;           // [0] n_entries  [1] arena words used  [2] overflow flag  [3] keys in the u64 table
    // keys that are plain ACGT strings of <= 29 bases live in a second, single-word table:
    // slot = (length << 58) | 2-bit key, claimed with one CAS; ~0 = empty
    unsigned long long *k64_slots;
    unsigned long long *k64_count;     // reads that carried the slot's key, MINUS ONE (a new key costs one atomic less: two out of three reads that reach this table bring a new key)
    unsigned long long *k64_first;
    uint32_t k64_mask, k64_room;       // slots - 1; keys the table may hold before it grows (3/4 of the slots)
};
#define F2Q_EC64_MAXLEN 29
// Single-word form of an Extract+Count key (the text of one window, upper case):
//   plain ACGT, len <= 29:            (len << 58) | 2-bit codes, base j in bits 2j..2j+1
//   ACGT with 1..3 'N's, 2*len + 2 + 5*nN <= 58:  ((32 | len) << 58) | codes (an 'N' stored as 0) | nN << 2*len |
//                                     the positions of the 'N's, ascending, 5 bits each, from bit 2*len + 2
// Every other key (longer, more 'N's, any other symbol, several parts) lives in the byte-string table; the rule is the
// same wherever a key is made, so a key never sits in both tables.  ~0 is the empty slot (length field 63: never made).
#define F2Q_EC64_NFLAG 32u
#define F2Q_EC64_MAXN 3
F2Q_HD bool ec64_word(uint64_t codes, uint32_t nmask, int len, unsigned long long &word)
{
    if (len < 0 || len > F2Q_EC64_MAXLEN) return false;
    if (nmask == 0u) { word = ((unsigned long long)len << 58) | codes; return true; }
    int nn = 0; for (uint32_t m = nmask; m; m &= m - 1u) nn++;
    if (nn > F2Q_EC64_MAXN || 2 * len + 2 + 5 * nn > 58) return false;
    unsigned long long w = codes | ((unsigned long long)nn << (2 * len));
    int sh = 2 * len + 2;
    for (uint32_t m = nmask; m; m &= m - 1u) {
        int pos = 0; while (!((m >> pos) & 1u)) pos++;
        w |= (unsigned long long)pos << sh; sh += 5;
    }
    word = ((unsigned long long)(F2Q_EC64_NFLAG | (uint32_t)len) << 58) | w;
    return true;
}
F2Q_HD bool ec64_fits(uint32_t nmask, int len)
{
    unsigned long long w;
    return ec64_word(0ull, nmask, len, w);
}
// the key text of a single-word slot -> out (at most 29 bytes); returns its length
F2Q_HD uint32_t ec64_text(unsigned long long word, char *out)
{
    const uint32_t lf = (uint32_t)(word >> 58), len = lf & 31u;
    for (uint32_t j = 0; j < len; j++) out[j] = "ACGT"[(word >> (2 * j)) & 3];
    if (lf & F2Q_EC64_NFLAG) {
        const uint32_t nn = (uint32_t)(word >> (2 * len)) & 3u;
        for (uint32_t i = 0; i < nn; i++) out[(word >> (2 * len + 2 + 5 * i)) & 31u] = 'N';
    }
    return len;
}

// Extract+Count, hot keys.  Counting a read costs one device-scope atomic on a scattered address (~10 G/s on MI355X), and
// in a screen most reads carry one of a few thousand keys.  Once the single-word table has seen F2Q_HOT_LEARN reads, the
// keys seen most often are copied into a small 2-choice bucket table (2 full key words per bucket, one ds_read_b128);
// a workgroup holds the keys and a u32 counter per slot in LDS, counts hits there and adds each counter to the table
// once, when it ends.  The LDS copy holds the whole key word, so a hit needs no memory access at all.  A hot key's
// first read is already in the table; only a read with an index BELOW every hot key's recorded first read (meta[0]:
// blocks counted out of order) must take the ordinary insert, which lowers the minimum.
#define F2Q_HOT_BUCKETS 6656u
#define F2Q_HOT_SLOTS (2u * F2Q_HOT_BUCKETS)     // 104 KiB of keys + 52 KiB of counters
#define F2Q_HOT_CAP 11264u           // keys admitted: load <= 0.85
#define F2Q_HOT_MINCOUNT 6u          // a key becomes a candidate when the learning reads bring its count to this
#define F2Q_HOT_CAND 32768u          // candidates noted (in the order they got there: the most frequent first)
#define F2Q_HOT_NONE 0xFFFFFFFFu
#define F2Q_HOT_MAXPROBE 96u         // slots an insert looks at before the read is set aside for a grown table
// words of EcDev.ctr past the four table counters
#define F2Q_CTR_ASIDE 4              // reads the hot-key kernel set aside in its last launch ...
#define F2Q_CTR_ASIDE_SLOW 5         // ... of those for the byte-exact routine
#define F2Q_CTR_CAND 6               // candidates noted so far in this sample
#define F2Q_CTR_WORDS 8
struct EcHot {
    unsigned long long *keys;        // [F2Q_HOT_SLOTS] key words, ~0 = empty (the image the workgroups copy into LDS)
    uint32_t *slot;                  // [F2Q_HOT_SLOTS]: the key's slot in the single-word table
    uint32_t *cand;                  // [F2Q_HOT_CAND]: table slots of the candidates
    unsigned long long *meta;        // [0] the highest first-read index recorded for a hot key when the set was built
};
struct HotProbe { uint32_t b1, b2; };
F2Q_HD HotProbe hot_probe(unsigned long long k)
{
    unsigned long long h = (k ^ (k >> 31)) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    HotProbe q;
    q.b1 = (uint32_t)(((h & 0xFFFFFFFFull) * F2Q_HOT_BUCKETS) >> 32);
    q.b2 = (uint32_t)(((h >> 32) * F2Q_HOT_BUCKETS) >> 32);
    if (q.b2 == q.b1) q.b2 = q.b1 + 1u == F2Q_HOT_BUCKETS ? 0u : q.b1 + 1u;
    return q;
}

F2Q_HD void acc_add(unsigned long long *p, unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(gpw(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p += v;
#endif
}

F2Q_HD uint32_t hash_slot(uint64_t k, uint32_t bits)
{
    return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64u - bits));
}

F2Q_HD uint32_t hash32(uint64_t k, uint32_t bits)
{
    uint32_t x = (uint32_t)k ^ ((uint32_t)(k >> 32) * 0x9E3779B1u);
    x *= 0x85EBCA6Bu;
    x ^= x >> 15;
    x *= 0xC2B2AE35u;
    return x >> (32u - bits);
}
// packed tables: a probe sequence starts at an even slot, so that the two slots of a round are one aligned 16-byte load
F2Q_HD uint32_t packed_start(uint64_t k, uint32_t bits) { return hash32(k, bits) & ~1u; }
// the pair of slots at even index s of a packed table
struct Slot2 { uint64_t a, b; };
template <class PT>
F2Q_HD Slot2 packed_pair(PT ptab, uint32_t at)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
    const v2 v = *reinterpret_cast<const v2 F2Q_GLOBAL *>(ptab + at);
    return Slot2{v.x, v.y};
#else
    return Slot2{ptab[at], ptab[at + 1]};
#endif
}

F2Q_HD uint8_t up8(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
F2Q_HD bool q_fails(uint8_t c, int thr) { return c >= 33 && (int)c <= thr; }

// Python slice bounds x[a:b] for len n
F2Q_HD void py_slice(int n, int a, int b, int &oa, int &ob)
{
    if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
    if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
    if (b < a) b = a;
    oa = a; ob = b;
}

// 2-bit code of an upper-case base, 4 = not ACGT
F2Q_HD uint32_t base_code(uint8_t c)
{
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

F2Q_HD int popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
// number of differing bases between two 2-bit keys (unused high bits equal in both)
F2Q_HD int ham2(uint64_t x)
{
    return popc64((x | (x >> 1)) & 0x5555555555555555ull);
}

// running unique-nearest state: best distance seen so far (starts at miss), how many features sit
// at it, and one of them.  Equivalent to the reference's iterative deepening (:734-750): assign
// iff the minimum distance d* <= miss is attained by exactly one feature.
struct MinTrack {
    int best, cnt; uint32_t idx;
    F2Q_HD void init(int miss) { best = miss; cnt = 0; idx = 0; }
    F2Q_HD void offer(int d, uint32_t i)
    {
        if (d < best) { best = d; cnt = 1; idx = i; }
        else if (d == best) { cnt++; idx = i; }
    }
};

// exact probe of the 2-bit table: feature index or -1
F2Q_HD int lib_exact(const LibDev &lib, uint64_t key, int L)
{
    const LenGroup &g = lib.grp[L];
    if (g.n == 0) return -1;
    const uint32_t m = (1u << g.exact.bits) - 1u;
    uint32_t s = hash_slot(key, g.exact.bits);
    const auto tab_keys = gp(lib.tab_keys);
    for (;;) {
        uint64_t k = tab_keys[g.exact.off + s];
        if (k == key) return (int)gp(lib.tab_idx)[g.exact.off + s];
        if (k == KEY_EMPTY) return -1;
        s = (s + 1) & m;
    }
}

// pigeonhole search over the regular features of length L: every feature within `miss` of the
// query agrees with it exactly on at least one of the miss+1 pieces, so only the features
// sharing a piece value are verified (XOR + popcount).  A feature seen through several pieces
// is counted at the first one.  `forced` = positions of the query that mismatch everything
// (bit 2j set for base j), used for windows holding a non-ACGT symbol.
F2Q_HD void lib_near(const LibDev &lib, uint64_t key, int L, uint64_t forced, MinTrack &t)
{
    const LenGroup &g = lib.grp[L];
    if (g.n == 0) return;
    const int nforced = popc64(forced);
    const auto tab_keys = gp(lib.tab_keys);
    const auto tab_idx = gp(lib.tab_idx);
    for (uint32_t p = 0; p < g.n_pieces; p++) {
        const PieceDesc pd = g.piece[p];
        if ((forced >> pd.shift) & pd.mask & 0x5555555555555555ull) continue;   // piece can never agree
        const uint64_t pv = (key >> pd.shift) & pd.mask;
        const uint32_t m = (1u << pd.bits) - 1u;
        uint32_t s = hash_slot(pv, pd.bits);
        for (;;) {
            uint64_t k = tab_keys[pd.off + s];
            if (k == KEY_EMPTY) break;
            uint64_t x = k ^ key;
            if (((x >> pd.shift) & pd.mask) == 0) {
                bool dup = false;
                for (uint32_t q = 0; q < p; q++) {
                    const PieceDesc qd = g.piece[q];
                    if ((forced >> qd.shift) & qd.mask & 0x5555555555555555ull) continue;
                    if (((x >> qd.shift) & qd.mask) == 0) { dup = true; break; }
                }
                if (!dup) {
                    uint64_t xm = x & ~(forced | (forced << 1));
                    t.offer(ham2(xm) + nforced, tab_idx[pd.off + s]);
                }
            }
            s = (s + 1) & m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// general path: keys as byte strings (any symbols, any length, ':'-joined multi-window keys)
// ---------------------------------------------------------------------------------------------
template <class P>
struct KeyViewT {
    P seq;                              // the read's sequence line (raw case)
    int nseg; int a[F2Q_DEV_MAX_ITER], b[F2Q_DEV_MAX_ITER];
    int len;                            // total key length incl. ':' separators
    F2Q_HD uint8_t at(int k) const
    {
        for (int s = 0; s < nseg; s++) {
            int n = b[s] - a[s];
            if (k < n) return up8(seq[a[s] + k]);
            k -= n;
            if (s + 1 < nseg) { if (k == 0) return (uint8_t)':'; k--; }
        }
        return 0;
    }
};
typedef KeyViewT<const uint8_t *> KeyView;      // key bytes in any address space (e.g. a register-built window)
typedef KeyViewT<gbytes> KeyViewG;              // key bytes inside a raw record in global memory

// distance between the key and feature f (same length), giving up once it exceeds `limit`
template <class KV>
F2Q_HD int key_dist(const KV &kv, gbytes fb, int limit)
{
    int d = 0, k = 0;
    for (int s = 0; s < kv.nseg; s++) {
        for (int j = kv.a[s]; j < kv.b[s]; j++, k++) {
            if (up8(kv.seq[j]) != fb[k]) { if (++d > limit) return d; }
        }
        if (s + 1 < kv.nseg) { if (fb[k] != (uint8_t)':') { if (++d > limit) return d; } k++; }
    }
    return d;
}

// brute force over a list of features (ids == nullptr: all features)
template <class KV>
F2Q_HD void lib_scan(const LibDev &lib, const KV &kv, const uint32_t *ids_, uint32_t n, MinTrack &t)
{
    const auto ids = gp(ids_);
    const auto feat_off = gp(lib.feat_off);
    for (uint32_t e = 0; e < n; e++) {
        uint32_t f = ids_ ? ids[e] : e;
        uint32_t o = feat_off[f];
        if ((int)(feat_off[f + 1] - o) != kv.len) continue;              // only same-length features (:683)
        int d = key_dist(kv, gp(lib.feat_bytes) + o, t.best);
        if (d <= t.best) t.offer(d, f);
    }
}

// ---- general keys (GkDesc) ------------------------------------------------------------------------------------
F2Q_HD uint64_t gk_mix(uint64_t h) { h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; return h; }
// FNV-1a over bytes [a, b) of a feature / of a key (upper-cased, ':' between the segments), salted by the piece number
template <class P>
F2Q_HD uint64_t gk_hash_bytes(P fb, int a, int b, uint32_t salt)
{
    uint64_t h = 1469598103934665603ull ^ (uint64_t)salt;
    for (int k = a; k < b; k++) { h ^= fb[k]; h *= 1099511628211ull; }
    return gk_mix(h);
}
template <class KV>
F2Q_HD uint64_t gk_hash_key(const KV &kv, int a, int b, uint32_t salt)
{
    uint64_t h = 1469598103934665603ull ^ (uint64_t)salt;
    for (int k = a; k < b; k++) { h ^= kv.at(k); h *= 1099511628211ull; }
    return gk_mix(h);
}
template <class KV>
F2Q_HD bool gk_range_equal(const KV &kv, gbytes fb, int a, int b)
{
    for (int k = a; k < b; k++) if (kv.at(k) != fb[k]) return false;
    return true;
}
F2Q_HD const GkGroup *gk_find(const LibDev &lib, int len)
{
    int lo = 0, hi = (int)lib.gk.n_groups - 1;                 // groups are sorted by length
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t l = lib.gk.grp[mid].len;
        if ((int)l == len) return &lib.gk.grp[mid];
        if ((int)l < len) lo = mid + 1; else hi = mid - 1;
    }
    return nullptr;
}
// the key as words (see GkDesc::fw), for keys of at most 8 * F2Q_GK_MAXW bytes
struct GkKey { unsigned long long w[F2Q_GK_MAXW]; };
template <class KV>
F2Q_HD void gk_key_words(const KV &kv, GkKey &k)
{
#pragma unroll
    for (int wi = 0; wi < F2Q_GK_MAXW; wi++) {
        unsigned long long v = 0;
        if (8 * wi < kv.len) {
#pragma unroll
            for (int b = 0; b < 8; b++) if (8 * wi + b < kv.len) v |= (unsigned long long)kv.at(8 * wi + b) << (8 * b);
        }
        k.w[wi] = v;
    }
}
// bit i of (lo, hi) set <=> byte i of the key differs from byte i of feature f
F2Q_HD void gk_diff_bytes(const LibDev &lib, const GkKey &k, uint32_t f, int len, unsigned long long &lo, unsigned long long &hi)
{
    const auto fw = gp(lib.gk.fw) + gp(lib.gk.fwoff)[f];
    lo = 0; hi = 0;
#pragma unroll
    for (int wi = 0; wi < F2Q_GK_MAXW; wi++) {
        if (8 * wi < len) {
            const unsigned long long x = k.w[wi] ^ fw[wi];
            const unsigned long long nz = ((((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull) >> 7;   // 1 per differing byte
            const unsigned long long bits = (nz * 0x0102040810204080ull) >> 56;                                                        // the 8 flags side by side
            if (wi < 8) lo |= bits << (8 * wi); else hi |= bits << (8 * (wi - 8));
        }
    }
}
F2Q_HD bool gk_range_clear(unsigned long long lo, unsigned long long hi, int a, int b)      // no differing byte in [a, b)
{
    for (int i = a; i < b; i++) if (((i < 64 ? lo >> i : hi >> (i - 64)) & 1ull)) return false;
    return true;
}
// exact hit among the features of the group: feature index or -1
// WORDS: check candidates on 8-byte words (13 more live registers: only where the byte-string index is the main road --
// the general kernel, the pair kernel; the packed kernels reach it for a stray read and keep the byte loop)
template <bool WORDS, class KV>
F2Q_HD int gk_exact(const LibDev &lib, const GkGroup &g, const KV &kv)
{
    const uint32_t m = (1u << g.bits) - 1u;
    uint32_t s = (uint32_t)(gk_hash_key(kv, 0, kv.len, 0xE0u) >> (64u - g.bits));
    const auto tab = gp(lib.gk.tab);
    const bool words = WORDS && kv.len <= 8 * F2Q_GK_MAXW && lib.gk.fw != nullptr;
    GkKey k;
    if (WORDS && words) gk_key_words(kv, k);
    for (;;) {
        const uint32_t e = tab[g.exact_off + s];
        if (e == 0u) return -1;
        if (WORDS && words) {
            unsigned long long lo, hi;
            gk_diff_bytes(lib, k, e - 1u, kv.len, lo, hi);
            if ((lo | hi) == 0ull) return (int)(e - 1u);
        } else if (gk_range_equal(kv, gp(lib.feat_bytes) + gp(lib.feat_off)[e - 1u], 0, kv.len)) return (int)(e - 1u);
        s = (s + 1u) & m;
    }
}
// pigeonhole search over the group (see lib_near): every feature within t.best of the key agrees with it on a whole piece
template <bool WORDS, class KV>
F2Q_HD void gk_near(const LibDev &lib, const GkGroup &g, const KV &kv, MinTrack &t)
{
    const uint32_t m = (1u << g.bits) - 1u;
    const auto tab = gp(lib.gk.tab);
    const bool words = WORDS && kv.len <= 8 * F2Q_GK_MAXW && lib.gk.fw != nullptr;
    GkKey k;
    if (WORDS && words) gk_key_words(kv, k);
    for (uint32_t p = 0; p < g.n_pieces; p++) {
        const int a = (int)g.cut[p], b = (int)g.cut[p + 1];
        uint32_t s = (uint32_t)(gk_hash_key(kv, a, b, p) >> (64u - g.bits));
        for (;;) {
            const uint32_t e = tab[g.piece_off[p] + s];
            if (e == 0u) break;
            if (WORDS && words) {
                unsigned long long lo, hi;
                gk_diff_bytes(lib, k, e - 1u, kv.len, lo, hi);
                if (gk_range_clear(lo, hi, a, b)) {
                    bool dup = false;                          // counted at the first piece it agrees on
                    for (uint32_t q = 0; q < p && !dup; q++) dup = gk_range_clear(lo, hi, (int)g.cut[q], (int)g.cut[q + 1]);
                    const int d = popc64(lo) + popc64(hi);
                    if (!dup && d <= t.best) t.offer(d, e - 1u);
                }
            } else {
                gbytes fb = gp(lib.feat_bytes) + gp(lib.feat_off)[e - 1u];
                if (gk_range_equal(kv, fb, a, b)) {
                    bool dup = false;
                    for (uint32_t q = 0; q < p && !dup; q++) dup = gk_range_equal(kv, fb, (int)g.cut[q], (int)g.cut[q + 1]);
                    if (!dup) { const int d = key_dist(kv, fb, t.best); if (d <= t.best) t.offer(d, e - 1u); }
                }
            }
            s = (s + 1u) & m;
        }
    }
}

// Counter-mode decision for one extracted key: returns 1 perfect, 2 imperfect, 3 non-aligned,
// and the feature index in `idx`.
template <bool WORDS = false, class KV>
F2Q_HD int match_key(const RunDev &run, const LibDev &lib, const KV &kv, uint32_t &idx)
{
    // is the key a plain ACGT string short enough for the 2-bit index?
    bool regular = (kv.nseg == 1 && kv.len >= 1 && kv.len <= F2Q_REG_MAXLEN);
    uint64_t key = 0, forced = 0;
    int nforced = 0;
    if (regular) {
        for (int j = 0; j < kv.len; j++) {
            uint32_t c = base_code(up8(kv.seq[kv.a[0] + j]));
            if (c > 3u) { forced |= 1ull << (2 * j); nforced++; c = 0; }
            key |= (uint64_t)c << (2 * j);
        }
    }
    MinTrack t; t.init(run.miss);
    if (regular && lib.n_irregular == 0) {
        if (nforced == 0) {
            int e = lib_exact(lib, key, kv.len);
            if (e >= 0) { idx = (uint32_t)e; return 1; }
        }
        // regular features: a non-ACGT query symbol mismatches every one of them
        if (run.miss > 0 && nforced <= run.miss) lib_near(lib, key, kv.len, forced, t);
    } else {
        // ':'-joined, long or odd-symbol keys, and every key once the library itself holds such features: the byte-string
        // index over ALL features of the key's length (the reference compares with every same-length feature, :683)
        const GkGroup *g = gk_find(lib, kv.len);
        if (g) {
            const int e = gk_exact<WORDS>(lib, *g, kv);
            if (e >= 0) { idx = (uint32_t)e; return 1; }
            if (run.miss > 0) {
                if (g->n_pieces) gk_near<WORDS>(lib, *g, kv, t);
                else lib_scan(lib, kv, lib.gk.ids + g->ids_off, g->n, t);   // fewer bytes than pieces: the whole group is within reach
            }
        }
    }
    if (t.cnt == 1) { idx = t.idx; return t.best == 0 ? 1 : 2; }
    // best == 0 with cnt > 1 cannot happen (library sequences are unique)
    return 3;
}

// Phred test of quality bytes [a,b) against threshold thr
template <class P>
F2Q_HD bool qual_range_fails(P q, int a, int b, int thr)
{
    if (thr < 33) return false;
    for (int i = a; i < b; i++) if (q_fails(q[i], thr)) return true;
    return false;
}

// border_finder (:628-658) on raw bytes: first p in [from, r-s] within k mismatches, else -1
template <class P>
F2Q_HD int border_find(gbytes anchor, int s, P read, int r, int k, int from)
{
    if (from < 0) from = 0;
    for (int p = from; p + s <= r && p < r; p++) {
        int d = 0;
        for (int j = 0; j < s; j++) { if (anchor[j] != read[p + j]) { if (++d > k) break; } }
        if (d <= k) return p;
    }
    return -1;
}

// sequence_tinder (:215-285): true + (start,end) or false
template <class P>
F2Q_HD bool tinder(const RunDev &run, P seq, int r, P qual, int qn, int i, int &start, int &end)
{
    int a, b;
    if (run.has_up && run.has_down) {
        int st = border_find(gp(&run.up[i][0]), run.up_len[i], seq, r, run.msu, 0);
        if (st < 0) return false;
        int en = border_find(gp(&run.down[i][0]), run.down_len[i], seq, r, run.msd, st + run.up_len[i]);
        if (en < 0) return false;
        py_slice(qn, st, st + run.up_len[i], a, b);
        if (qual_range_fails(qual, a, b, run.thr_up)) return false;
        py_slice(qn, en, en + run.down_len[i], a, b);
        if (qual_range_fails(qual, a, b, run.thr_down)) return false;
        start = st + run.up_len[i]; end = en;
        return true;
    } else if (run.has_up) {
        int st = border_find(gp(&run.up[i][0]), run.up_len[i], seq, r, run.msu, 0);
        if (st < 0) return false;
        py_slice(qn, st, st + run.up_len[i], a, b);
        if (qual_range_fails(qual, a, b, run.thr_up)) return false;
        start = st + run.up_len[i]; end = start + run.length;
        return true;
    } else if (run.has_down) {
        int en = border_find(gp(&run.down[i][0]), run.down_len[i], seq, r, run.msd, 0);
        if (en < 0) return false;
        py_slice(qn, en, en + run.down_len[i], a, b);
        if (qual_range_fails(qual, a, b, run.thr_down)) return false;
        start = en - run.length; end = en;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// EC byte-string table (device side of the de-novo dict, :382-387)
// ---------------------------------------------------------------------------------------------
template <class KV>
F2Q_HD uint64_t key_hash(const KV &kv)
{
    uint64_t h = 1469598103934665603ull ^ (uint64_t)kv.len;
    for (int k = 0; k < kv.len; k++) { h ^= kv.at(k); h *= 1099511628211ull; }
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}

#if defined(__HIP_DEVICE_COMPILE__)
#ifndef F2Q_EC_SCOPE
#define F2Q_EC_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#define F2Q_LD64(p) __hip_atomic_load(gpw(p), __ATOMIC_RELAXED, F2Q_EC_SCOPE)
#define F2Q_LD32(p) __hip_atomic_load(gpw(p), __ATOMIC_RELAXED, F2Q_EC_SCOPE)
#define F2Q_ST64(p, v) __hip_atomic_store(gpw(p), (v), __ATOMIC_RELAXED, F2Q_EC_SCOPE)
#define F2Q_ST32(p, v) __hip_atomic_store(gpw(p), (v), __ATOMIC_RELAXED, F2Q_EC_SCOPE)
#else
#define F2Q_LD64(p) (*(p))
#define F2Q_LD32(p) (*(p))
#define F2Q_ST64(p, v) (*(p) = (v))
#define F2Q_ST32(p, v) (*(p) = (v))
#endif

F2Q_HD unsigned long long ec_cas(unsigned long long *p, unsigned long long cmp, unsigned long long val)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_compare_exchange_strong(gpw(p), &cmp, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, F2Q_EC_SCOPE);
    return cmp;
#else
    unsigned long long old = *p; if (old == cmp) *p = val; return old;
#endif
}
F2Q_HD void ec_add(unsigned long long *p, unsigned long long v)           // result unused: the compiler emits the no-return form
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)__hip_atomic_fetch_add(gpw(p), v, __ATOMIC_RELAXED, F2Q_EC_SCOPE);
#else
    *p += v;
#endif
}
F2Q_HD unsigned long long ec_fetch_add(unsigned long long *p, unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_fetch_add(gpw(p), v, __ATOMIC_RELAXED, F2Q_EC_SCOPE);
#else
    unsigned long long old = *p; *p += v; return old;
#endif
}
F2Q_HD void ec_min(unsigned long long *p, unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_min(gpw(p), v, __ATOMIC_RELAXED, F2Q_EC_SCOPE);
#else
    if (v < *p) *p = v;
#endif
}

template <class KV>
F2Q_HD uint32_t key_word(const KV &kv, int w)
{
    uint32_t v = 0;
    for (int j = 0; j < 4; j++) { int k = 4 * w + j; if (k < kv.len) v |= (uint32_t)kv.at(k) << (8 * j); }
    return v;
}

// Insert-or-increment.  All table words are read and written with agent-scope atomics (sc1), so
// the table is coherent across XCDs inside one launch.  A slot is claimed with one CAS (locked),
// filled, then published; lanes that meet a locked slot of the same fingerprint re-poll it.  The
// claimer never waits on anybody, so the loop cannot deadlock inside a wave; polls are bounded.
template <class KV>
F2Q_HD void ec_insert(const EcDev &ec, const KV &kv, unsigned long long read_index)
{
    const uint64_t h = key_hash(kv);
    const unsigned long long fp = (h >> 32) & 0xFFFFFFFFull;
    const unsigned long long locked = (fp << 32) | 0xFFFFFFFFull;
    const int nw = (kv.len + 3) >> 2;
    uint32_t s = (uint32_t)h & ec.mask;
    for (uint32_t guard = 0; guard < (1u << 22); guard++) {
        unsigned long long v = F2Q_LD64(&ec.slots[s]);
        if (v == 0ull) {
            unsigned long long old = ec_cas(&ec.slots[s], 0ull, locked);
            if (old == 0ull) {
                unsigned long long e = ec_fetch_add(&ec.ctr[0], 1ull);
                unsigned long long off = ec_fetch_add(&ec.ctr[1], (unsigned long long)nw);
                if (e >= ec.max_entries || off + (unsigned long long)nw > ec.arena_words) {
                    F2Q_ST64(&ec.ctr[2], 1ull);          // overflow: the host grows the table and re-runs
                    return;                              // (slot stays locked; the table is discarded)
                }
                for (int w = 0; w < nw; w++) F2Q_ST32(&ec.arena[off + w], key_word(kv, w));
                F2Q_ST64(&ec.ent_off[e], off);
                F2Q_ST32(&ec.ent_len[e], (uint32_t)kv.len);
                ec_fetch_add(&ec.ent_count[e], 1ull);
                ec_min(&ec.ent_first[e], read_index);
#if defined(__HIP_DEVICE_COMPILE__)
                __threadfence();                         // entry + key bytes before the publish
#endif
                F2Q_ST64(&ec.slots[s], (fp << 32) | (e + 1ull));
                return;
            }
            v = old;
        }
        if ((v >> 32) == fp) {
            if ((v & 0xFFFFFFFFull) == 0xFFFFFFFFull) {                       // being filled: poll again
                if ((guard & 1023u) == 1023u && F2Q_LD64(&ec.ctr[2]) != 0ull) return;   // table overflowed
                continue;
            }
            unsigned long long e = (v & 0xFFFFFFFFull) - 1ull;
            if (F2Q_LD32(&ec.ent_len[e]) == (uint32_t)kv.len) {
                unsigned long long off = F2Q_LD64(&ec.ent_off[e]);
                bool same = true;
                for (int w = 0; w < nw && same; w++) same = (F2Q_LD32(&ec.arena[off + w]) == key_word(kv, w));
                if (same) {
                    ec_fetch_add(&ec.ent_count[e], 1ull);
                    if (read_index < F2Q_LD64(&ec.ent_first[e])) ec_min(&ec.ent_first[e], read_index);
                    return;
                }
            }
        }
        s = (s + 1) & ec.mask;
    }
    F2Q_ST64(&ec.ctr[2], 2ull);                         // probe bound hit: reported as an error
}

// single-word insert-or-increment for regular keys (see EcDev); returns 1 when the key is new.  The number of keys
// (ctr[3], read by the host between launches only) is NOT bumped here: a counter every new key increments is one
// address for the whole device, and 0.6 M same-address atomics cost ~9 ms — callers add up their new keys and
// report them once per wave (ec64_report_new).
F2Q_HD uint32_t ec64_insert_word(const EcDev &ec, unsigned long long k, unsigned long long read_index)
{
    uint32_t s = hash32(k ^ (k >> 29), 32) & ec.k64_mask;
    for (uint32_t guard = 0; guard <= ec.k64_mask; guard++) {
        unsigned long long v = F2Q_LD64(&ec.k64_slots[s]);
        uint32_t fresh = 0;
        if (v == KEY_EMPTY) {
            v = ec_cas(&ec.k64_slots[s], KEY_EMPTY, k);
            if (v == KEY_EMPTY) { fresh = 1; v = k; }
        }
        if (v == k) {
            if (!fresh) ec_fetch_add(&ec.k64_count[s], 1ull);    // the count word holds n - 1: the read that claims a slot need not touch it
            // the minimum only ever decreases: a plain look first saves the read-modify-write for almost every read
            if (read_index < F2Q_LD64(&ec.k64_first[s])) ec_min(&ec.k64_first[s], read_index);
            return fresh;
        }
        s = (s + 1) & ec.k64_mask;
    }
    F2Q_ST64(&ec.ctr[2], 3ull);                          // table full: reported as an error by the host
    return 0;
}
F2Q_HD uint32_t ec64_insert_n(const EcDev &ec, uint64_t key, int len, unsigned long long read_index)
{
    return ec64_insert_word(ec, ((unsigned long long)len << 58) | key, read_index);
}
// the same with a bounded probe sequence: 2 = gave up after max_probe slots, nothing changed (the caller sets the read
// aside; the host grows the table before such reads are decided).  The key word and the first-read word of a slot are
// fetched together (one wait instead of two; a stale first-read word is only ever too high, which costs a spare
// atomic-min at worst).  WANT_COUNT: the count before this read comes back (learning launches note candidates by it);
// without it the increment is a fire-and-forget atomic.
template <bool WANT_COUNT>
F2Q_HD uint32_t ec64_try_insert(const EcDev &ec, unsigned long long k, unsigned long long read_index, uint32_t max_probe,
                                uint32_t &slot, unsigned long long &count_before)
{
    uint32_t s = hash32(k ^ (k >> 29), 32) & ec.k64_mask;
    for (uint32_t guard = 0; guard < max_probe; guard++) {
        unsigned long long v = F2Q_LD64(&ec.k64_slots[s]);
        unsigned long long f = F2Q_LD64(&ec.k64_first[s]);
        uint32_t fresh = 0;
        if (v == KEY_EMPTY) {
            v = ec_cas(&ec.k64_slots[s], KEY_EMPTY, k);
            if (v == KEY_EMPTY) { fresh = 1; v = k; }
        }
        if (v == k) {
            // (the count word holds n - 1, see ec64_insert_word; count_before = reads that carried the key before this one)
            if (fresh) count_before = 0;
            else if (WANT_COUNT) count_before = ec_fetch_add(&ec.k64_count[s], 1ull) + 1ull;
            else ec_add(&ec.k64_count[s], 1ull);
            slot = s;
            if (read_index < f) ec_min(&ec.k64_first[s], read_index);
            return fresh;
        }
        s = (s + 1) & ec.k64_mask;
    }
    return 2u;
}
// one lane at a time (general path, host twin): report immediately
F2Q_HD void ec64_insert(const EcDev &ec, uint64_t key, int len, unsigned long long read_index)
{
    if (ec64_insert_n(ec, key, len, read_index)) ec_fetch_add(&ec.ctr[3], 1ull);
}
// every lane of the wave calls this once, after its last insert
F2Q_HD void ec64_report_new(const EcDev &ec, uint32_t n_new)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_new += __shfl_down(n_new, off, 64);
    if ((threadIdx.x & 63u) == 0 && n_new) ec_fetch_add(&ec.ctr[3], (unsigned long long)n_new);
#else
    ec.ctr[3] += n_new;
#endif
}

// Extract+Count: one key of a read.  Keys that have a single-word form (ec64_word) go to that table, every other key to
// the byte-string table -- one rule for every path, so that a key never sits in both.  n_new: the caller sums new
// single-word keys and reports them once per wave (ec64_report_new); nullptr: reported here.
template <class KV>
F2Q_HD void ec_count_key(const EcDev &ec, const KV &kv, unsigned long long read_index, uint32_t *n_new)
{
    bool regular = (kv.nseg == 1 && kv.len <= F2Q_EC64_MAXLEN && ec.k64_slots != nullptr);
    uint64_t key = 0; uint32_t nmask = 0;
    for (int j = 0; regular && j < kv.len; j++) {
        const uint8_t ch = up8(kv.seq[kv.a[0] + j]);
        uint32_t c = base_code(ch);
        if (c > 3u) { if (ch == (uint8_t)'N') { nmask |= 1u << j; c = 0; } else regular = false; }
        key |= (uint64_t)(c & 3u) << (2 * j);
    }
    unsigned long long word = 0;
    regular = regular && ec64_word(key, nmask, kv.len, word);
    if (regular && n_new) *n_new += ec64_insert_word(ec, word, read_index);
    else if (regular) { if (ec64_insert_word(ec, word, read_index)) ec_fetch_add(&ec.ctr[3], 1ull); }
    else ec_insert(ec, kv, read_index);
}

// ---------------------------------------------------------------------------------------------
// general path: one read given as raw bytes.  st[] = the 5 reference counters (thread-local).
// ---------------------------------------------------------------------------------------------
// WORDS: see gk_exact (true where this routine is the main road: k_count_general, the host twin)
template <class P, bool WORDS = false>
F2Q_HD void general_read(const RunDev &run, const LibDev &lib, const EcDev &ec, const Accum &acc,
                         P seq, int r, P qual, int qn,
                         unsigned long long read_index, unsigned long long st[5], uint32_t *n_new = nullptr)
{
    KeyViewT<P> kv; kv.seq = seq; kv.nseg = 0; kv.len = 0;
    bool all_failed = true;
    for (int i = 0; i < run.n_iter; i++) {
        int start, end;
        if (run.fixed) { start = run.start[i]; end = run.start[i] + run.length; }
        else {
            if (!tinder(run, seq, r, qual, qn, i, start, end)) continue;
            if (end < start) continue;                                           // :343-345
        }
        int a, b, qa, qb;
        py_slice(r, start, end, a, b);                                           // :354
        py_slice(qn, start, end, qa, qb);                                        // :355
        if (qual_range_fails(qual, qa, qb, run.thr)) continue;                   // :357-360
        all_failed = false;
        kv.a[kv.nseg] = a; kv.b[kv.nseg] = b; kv.nseg++;
        kv.len += (b - a) + (kv.nseg > 1 ? 1 : 0);
    }
    if (kv.nseg > 0) {
        if (run.mode == 0) {
            uint32_t idx = 0;
            int res = match_key<WORDS>(run, lib, kv, idx);
            if (res == 1 || res == 2) acc_add(&acc.counts[idx], 1ull);
            st[res]++;
        } else {
            ec_count_key(ec, kv, read_index, n_new);
            st[1]++;                                                             // :387
        }
    }
    if (all_failed) st[4]++;                                                     // :389-390
    st[0]++;                                                                     // :393
}

// ---------------------------------------------------------------------------------------------
// fast path, fixed offset: one lane = one read of a packed tile.
// tile planes are [word][lane]: bases 16 per u32 (2 bits, LSB first), qualities 4 per u32.
// ---------------------------------------------------------------------------------------------
struct PackedBlock {
    uint64_t n_slots;          // read slots (multiple of F2Q_TILE; slots >= n_valid hold len = SKIP)
    uint64_t first_index;      // global index of slot 0 (for EC first-occurrence ordering)
    uint32_t n_tiles, wb, wq, rmax;
    uint32_t planar_nw, pad0;  // > 0: bases are bit-planes of planar_nw words (anchored runs), wb = 2 * planar_nw
    const uint32_t *bases;     // [n_tiles][wb][F2Q_TILE]
    const uint32_t *qual;      // [n_tiles][wq][F2Q_TILE]
    const uint16_t *len;       // [n_tiles][F2Q_TILE] or nullptr (all reads rmax long, none skipped)
    const uint32_t *index;     // [n_tiles][F2Q_TILE] position of the slot's read inside the block, or nullptr (== slot)
};

// any byte of the 4 in w (all < 128) inside [33, thr]?  mask selects the bytes to test (0x80 per byte)
F2Q_HD uint32_t qfail4(uint32_t w, uint32_t add_lo, uint32_t add_hi, uint32_t mask)
{
    // byte >= 33  <=>  bit7 of (byte + 95);   byte > thr  <=>  bit7 of (byte + 127 - thr)
    return (w + add_lo) & ~(w + add_hi) & mask;
}

// returns the reference counter to bump (1 perfect, 2 imperfect, 3 non-aligned, 4 quality failed,
// 0 = slot skipped) and the feature index.  Needs: fixed mode, one window, length <= 31, start >= 0.
F2Q_HD int fixed_lane(const RunDev &run, const LibDev &lib, const PackedBlock &pb, uint32_t tile,
                      uint32_t lane, uint32_t &idx)
{
    int rlen = (int)pb.rmax;
    bool flagged = false;
    if (pb.len) {
        uint32_t l = gp(pb.len)[(uint64_t)tile * F2Q_TILE + lane];
        if (l == F2Q_LEN_SKIP) return 0;
        flagged = (l & F2Q_LEN_FLAG) != 0;
        rlen = (int)(l & F2Q_LEN_MASK);
    }
    const int st = run.start[0];
    int a = st < rlen ? st : rlen;
    int b = (st + run.length) < rlen ? (st + run.length) : rlen;
    const int L = b - a;                                     // clipped window (Python slice, :354)
    // ---- Phred mask over quality bytes [a,b); bit 7 of a byte marks a non-ACGT base (flagged reads only) ----
    uint64_t forced = 0;
    if ((run.thr >= 33 || flagged) && L > 0) {
        const uint32_t add_lo = 0x5F5F5F5Fu;                                   // +95
        const uint32_t add_hi = run.thr >= 33 ? (uint32_t)(127 - run.thr) * 0x01010101u : 0u;
        const auto qp = gp(pb.qual) + ((uint64_t)tile * pb.wq) * F2Q_TILE + lane;
        uint32_t bad = 0;
        const int w0 = a >> 2, w1 = (b - 1) >> 2;
        for (int w = w0; w <= w1; w++) {
            uint32_t m = 0x80808080u;
            if (w == w0) m &= 0xFFFFFFFFu << (8 * (a & 3));
            if (w == w1) m &= 0xFFFFFFFFu >> (8 * (3 - ((b - 1) & 3)));
            const uint32_t word = qp[(uint64_t)w * F2Q_TILE];
            if (add_hi) bad |= qfail4(word & 0x7F7F7F7Fu, add_lo, add_hi, m);
            uint32_t fl = word & m;
            for (int k = 0; fl && k < 4; k++)
                if (fl & (0x80u << (8 * k))) forced |= 1ull << (2 * (4 * w + k - a));
        }
        if (bad) return 4;
    }
    // ---- 2-bit key of bases [a,b) ----
    uint64_t key = 0;
    if (L > 0) {
        const auto bp = gp(pb.bases) + ((uint64_t)tile * pb.wb) * F2Q_TILE + lane;
        const int w0 = a >> 4, w1 = (b - 1) >> 4;              // at most 3 words for L <= 31
        uint64_t lo = bp[(uint64_t)w0 * F2Q_TILE];
        uint64_t mid = (w1 > w0) ? bp[(uint64_t)(w0 + 1) * F2Q_TILE] : 0u;
        uint64_t hi = (w1 > w0 + 1) ? bp[(uint64_t)(w0 + 2) * F2Q_TILE] : 0u;
        const int sh = 2 * (a & 15);
        key = (lo | (mid << 32)) >> sh;
        if (sh) key |= hi << (64 - sh);
        key &= (L >= 32) ? ~0ull : ((1ull << (2 * L)) - 1ull);
    }
    if (forced) {
        // non-ACGT symbols mismatch every feature (the packer only flags reads when the library is all-ACGT)
        const int nforced = popc64(forced);
        if (run.miss == 0 || nforced > run.miss) return 3;
        MinTrack tf; tf.init(run.miss);
        lib_near(lib, key, L, forced, tf);
        if (tf.cnt == 1) { idx = tf.idx; return 2; }
        return 3;
    }
    if (L < 1) {
        // empty window: the key "" can only match an (irregular) empty feature
        KeyView kv; kv.seq = nullptr; kv.nseg = 1; kv.a[0] = 0; kv.b[0] = 0; kv.len = 0;
        return match_key(run, lib, kv, idx);
    }
    if (lib.n_irregular) {
        // a library that also holds odd symbols or other lengths: decode the window, the byte-string index decides
        uint8_t wbuf[F2Q_REG_MAXLEN + 1];
        for (int j = 0; j < L; j++) wbuf[j] = (uint8_t)"ACGT"[(key >> (2 * j)) & 3];
        KeyView kv; kv.seq = wbuf; kv.nseg = 1; kv.a[0] = 0; kv.b[0] = L; kv.len = L;
        return match_key(run, lib, kv, idx);
    }
    int e = lib_exact(lib, key, L);
    if (e >= 0) { idx = (uint32_t)e; return 1; }
    if (run.miss == 0) return 3;
    MinTrack t; t.init(run.miss);
    lib_near(lib, key, L, 0ull, t);
    if (t.cnt == 1) { idx = t.idx; return t.best == 0 ? 1 : 2; }
    return 3;
}

// ---------------------------------------------------------------------------------------------
// fast path v2: one wave per 256-read tile, lane l owns reads 4l..4l+3 so that every tile row is
// fetched with one 16-byte load per lane; probes go to the packed tables.
// ---------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
F2Q_HD uint32_t u4get(const U4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

#define F2Q_MAXQROWS 9     // a <= 31-base window touches at most 9 quality words
#define F2Q_MAXBROWS 3     // ... and at most 3 base words

// wave-uniform geometry of the window in tile rows
struct FixedGeom {
    int st, L;                 // window start / length (full, unclipped)
    int qw0, nq;               // first quality row, number of rows
    int bw0, nb;               // first base row, number of rows
    uint32_t qm_first, qm_last;   // 0x80-per-byte masks of the bytes tested in the first / last row
    uint32_t add_lo, add_hi;   // SWAR constants of the Phred rule (thr >= 33), add_hi == 0: rule off
    int sh;                    // bit offset of the window inside the first base row
    uint64_t kmask;            // (1 << 2L) - 1
};

F2Q_HD FixedGeom fixed_geom_at(int st, int L, int thr)
{
    FixedGeom g;
    g.st = st; g.L = L;
    const int a = g.st, b = g.st + g.L;
    g.qw0 = a >> 2; g.nq = g.L > 0 ? ((b - 1) >> 2) - g.qw0 + 1 : 0;
    g.bw0 = a >> 4; g.nb = g.L > 0 ? ((b - 1) >> 4) - g.bw0 + 1 : 0;
    g.qm_first = 0x80808080u & (0xFFFFFFFFu << (8 * (a & 3)));
    g.qm_last = g.L > 0 ? (0x80808080u & (0xFFFFFFFFu >> (8 * (3 - ((b - 1) & 3))))) : 0u;
    if (g.nq == 1) { g.qm_first &= g.qm_last; g.qm_last = g.qm_first; }
    g.add_lo = 0x5F5F5F5Fu;
    g.add_hi = thr >= 33 ? (uint32_t)(127 - thr) * 0x01010101u : 0u;
    g.sh = 2 * (a & 15);
    g.kmask = g.L >= 32 ? ~0ull : ((1ull << (2 * g.L)) - 1ull);
    return g;
}
F2Q_HD FixedGeom fixed_geom(const RunDev &run) { return fixed_geom_at(run.start[0], run.length, run.thr); }

// result codes of one read in the v2 kernel
enum { R_SKIP = 0, R_PERFECT = 1, R_IMPERFECT = 2, R_NONALIGNED = 3, R_QFAIL = 4, R_SLOW = 5, R_NEAR = 6, R_FORCED = 7 };

// Phred test of one quality row for the 4 reads of a lane (bad[j] != 0: read j fails)
F2Q_HD void fixed4_qrow(const FixedGeom &g, int r, const U4 &q, uint32_t bad[4])
{
    const uint32_t m = (r == 0) ? g.qm_first : (r == g.nq - 1) ? g.qm_last : 0x80808080u;
    // bit 7 of a byte is the non-ACGT flag, not quality: strip it before the 7-bit SWAR test
    bad[0] |= qfail4(q.x & 0x7F7F7F7Fu, g.add_lo, g.add_hi, m);
    bad[1] |= qfail4(q.y & 0x7F7F7F7Fu, g.add_lo, g.add_hi, m);
    bad[2] |= qfail4(q.z & 0x7F7F7F7Fu, g.add_lo, g.add_hi, m);
    bad[3] |= qfail4(q.w & 0x7F7F7F7Fu, g.add_lo, g.add_hi, m);
}

// 2-bit key of read j of a lane from the base rows the lane loaded
template <int BR>
F2Q_HD uint64_t fixed4_key(const FixedGeom &g, const U4 (&b)[BR], int j)
{
    uint64_t lo = u4get(b[0], j);
    uint64_t mid = (BR > 1 && g.nb > 1) ? u4get(b[BR > 1 ? 1 : 0], j) : 0u;
    uint64_t hi = (BR > 2 && g.nb > 2) ? u4get(b[BR > 2 ? 2 : 0], j) : 0u;
    uint64_t k = (lo | (mid << 32)) >> g.sh;
    if (g.sh) k |= hi << (64 - g.sh);
    return k & g.kmask;
}

// exact probe of the packed table: feature index or -1.  Two slots are fetched per round.
F2Q_HD int packed_exact(const LibDev &lib, const PackedGroup &pk, uint64_t key)
{
    const PackedPiece &e = pk.exact;
    const uint32_t m = (1u << e.bits) - 1u, ib = pk.ib;
    uint32_t s = packed_start(key, e.bits);
    const auto ptab = gp(lib.ptab);
    for (;;) {
        const Slot2 pr = packed_pair(ptab, e.off + s);
        const uint64_t v0 = pr.a, v1 = pr.b;
        if (v0 != KEY_EMPTY && (v0 >> ib) == key) return (int)(v0 & ((1ull << ib) - 1ull));
        if (v0 == KEY_EMPTY) return -1;
        if (v1 != KEY_EMPTY && (v1 >> ib) == key) return (int)(v1 & ((1ull << ib) - 1ull));
        if (v1 == KEY_EMPTY) return -1;
        s = (s + 2) & m;
    }
}
F2Q_HD int packed_exact(const LibDev &lib, uint64_t key) { return packed_exact(lib, lib.pk, key); }

// bit i -> bit 2i
F2Q_HD uint64_t spread32(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// flag bits (bit 7 of the quality bytes under the window) of read j of a lane -> one bit per window base
template <int QR>
F2Q_HD uint32_t fixed4_flags(const FixedGeom &g, const U4 (&q)[QR], int j)
{
    uint64_t bits = 0;                       // bit (4r + k) = byte k of row r
#pragma unroll
    for (int r = 0; r < QR; r++) {
        if (r < g.nq) {
            const uint32_t m = (r == 0) ? g.qm_first : (r == g.nq - 1) ? g.qm_last : 0x80808080u;
            const uint32_t f = (u4get(q[r], j) & m) >> 7;                       // 0/1 at bits 0, 8, 16, 24
            bits |= (uint64_t)((f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu) << (4 * r);   // gathered into a nibble (no multiply: v_mul_lo_u32 is quarter rate)
        }
    }
    return (uint32_t)(bits >> (g.st & 3));
}

// Two windows A, B of L bases each: the joined key with its middle quarters swapped, [A_lo B_lo A_hi B_hi] (A_lo = the
// first L/2 bases of A).  The LDS tables bucket by the halves of the key they are given; with the plain joined key a half
// IS a window, and in a combinatorial pair library (one guide with many partners) more features share a half than a
// bucket holds.  Mixed, two features share a half only if they agree on half of BOTH guides.  A permutation of the
// bases changes no Hamming distance; the forced-mismatch mask is permuted the same way (unit = bits per base: 2 / 1).
F2Q_HD uint64_t mw_mix_bits(uint64_t x, uint32_t L, uint32_t unit)
{
    const uint32_t a = (L / 2u) * unit, b = (L - L / 2u) * unit, w = L * unit;        // bits of A_lo (= B_lo), A_hi (= B_hi), a window
    const uint64_t alo = x & ((1ull << a) - 1ull), ahi = (x >> a) & ((1ull << b) - 1ull);
    const uint64_t blo = (x >> w) & ((1ull << a) - 1ull), bhi = (x >> (w + a)) & ((1ull << b) - 1ull);
    return alo | (blo << a) | (ahi << (2u * a)) | (bhi << (2u * a + b));
}
F2Q_HD uint64_t mw_mix(uint64_t key, uint32_t L) { return mw_mix_bits(key, L, 2u); }
F2Q_HD uint32_t mw_mix_mask(uint32_t forced, uint32_t L) { return (uint32_t)mw_mix_bits(forced, L, 1u); }

// Phred verdict per window base of read j of a lane (bit i set: base i of the window fails) -- the multi-window form of
// the kernels tests each part of the compact window by itself (a failed part is omitted, fast2q.py:357-360)
template <int QR>
F2Q_HD uint32_t fixed4_failbits(const FixedGeom &g, const U4 (&q)[QR], int j)
{
    uint64_t bits = 0;
#pragma unroll
    for (int r = 0; r < QR; r++) {
        if (r < g.nq) {
            const uint32_t m = (r == 0) ? g.qm_first : (r == g.nq - 1) ? g.qm_last : 0x80808080u;
            const uint32_t f = qfail4(u4get(q[r], j) & 0x7F7F7F7Fu, g.add_lo, g.add_hi, m) >> 7;
            bits |= (uint64_t)((f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu) << (4 * r);
        }
    }
    return (uint32_t)(bits >> (g.st & 3));
}
// ... and what they say about a read of n_parts windows of part_len bases: 0 = every part passes, 1 = some part fails
// (the key has fewer parts: in a library of n_parts-part features it can equal or approach none), 2 = every part fails
F2Q_HD int mw_part_verdict(uint32_t failbits, int n_parts, int part_len)
{
    int failed = 0;
    for (int w = 0; w < n_parts; w++) failed += (failbits >> (w * part_len)) & ((1u << part_len) - 1u) ? 1 : 0;
    return failed == 0 ? 0 : failed == n_parts ? 2 : 1;
}

// Extract+Count with a fixed window: the single-word key of read j of a lane (l = its length word).  The window is
// clipped to the read like a Python slice (fast2q.py:354); an 'N' inside it travels as a flag bit and is spelt into the
// word (ec64_word) -- the packer passes only reads whose window has a single-word form (read_is_clean).
template <int BR, int QR>
F2Q_HD unsigned long long fixed4_ec_word(const FixedGeom &g, const U4 (&b)[BR], const U4 (&q)[QR], int j, uint32_t l)
{
    const int rl = (int)(l & F2Q_LEN_MASK);
    int L = (rl < g.st + g.L ? rl : g.st + g.L) - g.st;
    if (L < 0) L = 0;
    const uint64_t key = fixed4_key(g, b, j) & (L >= 32 ? ~0ull : ((1ull << (2 * L)) - 1ull));
    unsigned long long w = ((unsigned long long)L << 58) | key;
    if (l & F2Q_LEN_FLAG) ec64_word(key, fixed4_flags(g, q, j) & (uint32_t)((1ull << L) - 1ull), L, w);
    return w;
}

// pigeonhole search on the packed piece tables; forced2 = 2-bit-spaced mask of query positions that
// mismatch every feature (non-ACGT symbols).  The first slot of every piece's chain (up to 4 pieces) is
// fetched before any chain is walked, so the usual m = 1 lookup costs one memory round trip, not two.
F2Q_HD void packed_near(const LibDev &lib, const PackedGroup &pk, uint64_t key, uint64_t forced2, MinTrack &t)
{
    const uint32_t ib = pk.ib;
    const uint64_t imask = (1ull << ib) - 1ull;
    const auto ptab = gp(lib.ptab);
    const int nforced = popc64(forced2);
    const uint64_t keep = ~(forced2 | (forced2 << 1));
    const uint32_t np = pk.n_pieces;
    uint32_t s0[4]; uint64_t v0[4], v1[4];
#pragma unroll
    for (uint32_t p = 0; p < 4; p++) {
        if (p < np) {
            const PackedPiece pd = pk.piece[p];
            s0[p] = packed_start((key >> pd.shift) & pd.mask, pd.bits);
            const Slot2 pr = packed_pair(ptab, pd.off + s0[p]);             // the chain's first two slots, one 16-byte load: at
            v0[p] = pr.a; v1[p] = pr.b;                                      // load factor <= 0.25 most chains end here
        } else { s0[p] = 0; v0[p] = KEY_EMPTY; v1[p] = KEY_EMPTY; }
    }
#pragma unroll
    for (uint32_t p = 0; p < F2Q_MAX_PIECES; p++) {
        if (p >= np) break;
        const PackedPiece pd = pk.piece[p];
        if ((forced2 >> pd.shift) & pd.mask) continue;                       // this piece can never agree
        const uint32_t m = (1u << pd.bits) - 1u;
        uint32_t s = p < 4 ? s0[p < 4 ? p : 0] : packed_start((key >> pd.shift) & pd.mask, pd.bits);
        uint64_t v = p < 4 ? v0[p < 4 ? p : 0] : ptab[pd.off + s];
        uint64_t vn = p < 4 ? v1[p < 4 ? p : 0] : KEY_EMPTY;
        bool have_next = p < 4;
        for (;;) {
            if (v == KEY_EMPTY) break;
            uint64_t x = (v >> ib) ^ key;
            if (((x >> pd.shift) & pd.mask) == 0) {
                bool dup = false;
                for (uint32_t q = 0; q < p; q++) {
                    const PackedPiece qd = pk.piece[q];
                    if ((forced2 >> qd.shift) & qd.mask) continue;
                    if (((x >> qd.shift) & qd.mask) == 0) { dup = true; break; }
                }
                if (!dup) t.offer(ham2(x & keep) + nforced, (uint32_t)(v & imask));
            }
            s = (s + 1) & m;
            if (have_next) { v = vn; have_next = false; }
            else v = ptab[pd.off + s];
        }
    }
}

// decision for a key whose exact probe missed, or that holds flagged symbols (one bit per base in
// `forced`): the work the v2 kernel queues and compacts
F2Q_HD int packed_near_decide(const RunDev &run, const LibDev &lib, const PackedGroup &pk, uint64_t key, uint32_t forced, uint32_t &idx)
{
    MinTrack t; t.init(run.miss);
    packed_near(lib, pk, key, forced ? spread32(forced) : 0ull, t);
    if (t.cnt == 1) { idx = t.idx; return t.best == 0 ? R_PERFECT : R_IMPERFECT; }
    return R_NONALIGNED;
}
F2Q_HD int packed_near_decide(const RunDev &run, const LibDev &lib, uint64_t key, uint32_t forced, uint32_t &idx)
{
    return packed_near_decide(run, lib, lib.pk, key, forced, idx);
}

// ---------------------------------------------------------------------------------------------
// multi-window runs on the packed path (--st a,b[,c,d] --l n, Counter mode, k * n <= 31): for every window the Phred
// test and the 2-bit part; the parts that passed are concatenated in window order (the ':' between them is implied by
// the part count k, fast2q.py:349-363) and looked up among the k-part features.  One lane = 4 reads, windows one after
// another.  geometry of window w = that of a single-window run starting at run.start[w].
// ---------------------------------------------------------------------------------------------
F2Q_HD FixedGeom fixed_geom_of(const RunDev &run, int w) { return fixed_geom_at(run.compact ? w * run.length : run.start[w], run.length, run.thr); }

// ---------------------------------------------------------------------------------------------
// LDS tables: the whole library inside the workgroup's LDS (north star: "the feature table tiled into LDS").
// Applies to Counter-mode runs whose features all have the window length L, 14 <= L <= 21, ACGT only, searched with
// --m <= 1, when the cuckoo build below succeeds (up to ~13 k features).  A key is cut into half 0 (bases [0, L/2))
// and half 1 (the rest).  Table t holds every feature once, in one of the two buckets that half t alone selects
// (two bijective scrambles of the half; a bucket = 2 slots = one ds_read_b64), as a 32-bit tag
//     [choice c : 1][0...][half t scrambled by choice c, bits above the bucket index][other half, unscrambled]
// so bucket + tag determine the key, and an empty slot (all ones) equals no tag.  One read costs four 8-byte LDS
// loads and no memory access outside the streamed tile rows:
//   exact hit    (fast2q.py:365-367)   an entry of table 0 with the same half 0 and the same half 1
//   1 mismatch   (:692-750, --m 1)     a feature at distance exactly 1 agrees with the query on exactly one half, so
//                                      it sits in the query's buckets of exactly one table: the entries with the same
//                                      half t whose other half differs in one base are ALL the candidates; assign iff
//                                      there is exactly one (unique nearest, distance 1)
// A flagged (non-ACGT) query base is a forced mismatch: with one of them the candidates are the entries of the table
// bucketed by the clean half whose other half agrees everywhere else.  Counts go to a u16 histogram indexed by the
// table-0 slot (lt_count); a table-1 hit knows its feature's whole key and looks its table-0 slot up (lt_slot_of_t1).
// ---------------------------------------------------------------------------------------------
F2Q_HD uint32_t lt_mul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // both operands < 2^24: one full-rate v_mul_u32_u24.  As an asm statement: the optimiser otherwise merges the
    // caller's mask into a 32-bit v_mul_lo_u32, which issues at a quarter of the rate
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
#else
    return a * b;
#endif
}
// bijective scramble c of a half value (bits wide, 13 < bits <= 22): multiply by an odd constant modulo 2^bits, then
// fold the high bits onto the low ones (both steps are invertible); bucket = low 13 bits, the rest goes into the tag
F2Q_HD uint32_t lt_perm(uint32_t h, uint32_t bits, int c)
{
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t v = lt_mul24(h, c ? 0x9E3779u : 0x85EBCBu) & mask;
    v ^= v >> (bits - 10u);                    // the product's high bits are the well-mixed ones: fold them onto the bucket bits
    return v;
}
// choice c of half value h of table t: bucket and what (entry >> ob) must equal for "same half" (ob = bits of the other half)
// bb: bucket-index bits of the table (F2Q_LT_BBITS for a table held in LDS; the global table 1 of a partitioned library is larger)
F2Q_HD void lt_hash(uint32_t h, uint32_t bits, uint32_t ob, int c, uint32_t &bucket, uint32_t &cmp, uint32_t bb = F2Q_LT_BBITS)
{
    const uint32_t v = lt_perm(h, bits, c);
    bucket = v & ((1u << bb) - 1u);
    cmp = ((uint32_t)c << (31u - ob)) | (v >> bb);
}
F2Q_HD uint32_t lt_tag(uint32_t cmp, uint32_t other, uint32_t ob) { return (cmp << ob) | other; }
// partition of a half-0 value (PtDesc): two full-rate 24-bit multiplies; every feature and every query with that half 0
F2Q_HD uint32_t pt_part(uint32_t h0, uint32_t n_parts)
{
    const uint32_t x = lt_mul24(h0 ^ (h0 >> 11), 0xB5297Au);
    return lt_mul24((x >> 8) & 0xFFFFu, n_parts) >> 16;
}
// a candidate read as it travels from the scatter pass to the count pass of a partitioned library: the 2L key bits, then
// one bit per window base that is no ACGT symbol (a forced mismatch)
F2Q_HD unsigned long long pt_entry(uint64_t key, uint32_t forced, uint32_t L) { return key | ((unsigned long long)forced << (2u * L)); }

F2Q_HD uint32_t ham2_32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc((x | (x >> 1)) & 0x55555555u);
#else
    return (uint32_t)__builtin_popcount((x | (x >> 1)) & 0x55555555u);
#endif
}
F2Q_HD uint32_t lt_ctz(uint32_t x)            // index of the lowest set bit (x != 0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffs((int)x) - 1);
#else
    return (uint32_t)__builtin_ctz(x);
#endif
}
F2Q_HD uint32_t spread16(uint32_t v)          // bit i -> bit 2i, i < 16
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

struct U2 { uint32_t x, y; };
// the four bucket addresses of a key (table-0 choices 0/1, table-1 choices 0/1) and their "same half" comparands
struct LtProbe { uint32_t b[4], cmp[4], h0, h1; };
// bb1: bucket bits of table 1 (a partitioned library keeps ONE table 1 for all partitions, in global memory)
F2Q_HD LtProbe lt_probe(const LtDesc &lt, uint64_t key, uint32_t bb1 = F2Q_LT_BBITS)
{
    LtProbe q;
    q.h0 = (uint32_t)key & ((1u << lt.hb0) - 1u);
    q.h1 = (uint32_t)(key >> lt.hb0);
    lt_hash(q.h0, lt.hb0, lt.hb1, 0, q.b[0], q.cmp[0]);
    lt_hash(q.h0, lt.hb0, lt.hb1, 1, q.b[1], q.cmp[1]);
    lt_hash(q.h1, lt.hb1, lt.hb0, 0, q.b[2], q.cmp[2], bb1);
    lt_hash(q.h1, lt.hb1, lt.hb0, 1, q.b[3], q.cmp[3], bb1);
    return q;
}
// exact hit among the two table-0 buckets: slot or -1 (branch-free: a key sits in at most one slot)
F2Q_HD int lt_exact(const LtDesc &lt, const LtProbe &q, const U2 &e0, const U2 &e1)
{
    const uint32_t w0 = lt_tag(q.cmp[0], q.h1, lt.hb1), w1 = lt_tag(q.cmp[1], q.h1, lt.hb1);
    const bool a0 = e0.x == w0, a1 = e0.y == w0, b0 = e1.x == w1, b1 = e1.y == w1;
    const uint32_t slot = (a0 | a1) ? 2u * q.b[0] + (uint32_t)a1 : 2u * q.b[1] + (uint32_t)b1;
    return (a0 | a1 | b0 | b1) ? (int)slot : -1;
}
// candidates at distance exactly 1 among the four buckets e[0..3] (see the section comment).  forced: one bit per base
// of the window that mismatches every feature.  Returns the number of candidates; for one of them: hit = table (0/1)
// << 31 | its slot in that table, hitw = its tag.
F2Q_HD uint32_t lt_near1(const LtDesc &lt, const LtProbe &q, const U2 (&e)[4], uint32_t forced, uint32_t &hit, uint32_t &hitw)
{
    const uint32_t l0 = lt.hb0 >> 1;
    const uint32_t f0 = forced & ((1u << l0) - 1u), f1 = forced >> l0;
    uint32_t nf = 0, keep0 = ~0u, keep1 = ~0u;
    if (forced) {
#if defined(__HIP_DEVICE_COMPILE__)
        nf = (uint32_t)__popc(forced);
#else
        nf = (uint32_t)__builtin_popcount(forced);
#endif
        const uint32_t s0 = spread16(f0), s1 = spread16(f1);
        keep0 = ~(s0 | (s0 << 1)); keep1 = ~(s1 | (s1 << 1));
    }
    uint32_t n = 0;
    hit = 0; hitw = 0;
    const uint32_t m0 = (1u << lt.hb0) - 1u, m1 = (1u << lt.hb1) - 1u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool t1 = k >= 2;                       // table 1: same half 1, other half = half 0
        // a flagged base in the bucketing half: that half can agree with nothing
        const bool open = t1 ? (f1 == 0u) : (f0 == 0u);
        const uint32_t ob = t1 ? lt.hb0 : lt.hb1, om = t1 ? m0 : m1, oq = t1 ? q.h0 : q.h1, keep = t1 ? keep0 : keep1;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t w = i ? e[k].y : e[k].x;
            const bool same = (w >> ob) == q.cmp[k];
            const uint32_t d = ham2_32(((w ^ oq) & om) & keep) + nf;
            const bool c = open & same & (d == 1u);
            n += (uint32_t)c;
            hit = c ? (((uint32_t)t1 << 31) | (2u * q.b[k] + (uint32_t)i)) : hit;
            hitw = c ? w : hitw;
        }
    }
    return n;
}
// the table-0 slot (= histogram counter) of the feature a table-1 entry stands for: the entry's tag holds the feature's
// half 0, the bucket it was found through its half 1, so the feature's key is known and table 0 is probed for it.
// rd(bucket) reads the two tags of a table-0 bucket.
template <class RD>
F2Q_HD uint32_t lt_slot_of_t1(const LtDesc &lt, const LtProbe &q, uint32_t hitw, RD rd)
{
    const uint32_t fh0 = hitw & ((1u << lt.hb0) - 1u);
    uint32_t b0, c0, b1, c1;
    lt_hash(fh0, lt.hb0, lt.hb1, 0, b0, c0); lt_hash(fh0, lt.hb0, lt.hb1, 1, b1, c1);
    const U2 e0 = rd(b0), e1 = rd(b1);
    const uint32_t w0 = lt_tag(c0, q.h1, lt.hb1), w1 = lt_tag(c1, q.h1, lt.hb1);
    const bool a1 = e0.y == w0, hit0 = (e0.x == w0) | a1, b1y = e1.y == w1;
    return hit0 ? 2u * b0 + (uint32_t)a1 : 2u * b1 + (uint32_t)b1y;
}

// Lane predicates.  On the device a predicate is a 64-bit lane mask in scalar registers (what v_cmp writes): boolean
// algebra on it runs on the scalar unit, beside the vector instructions, and selecting by it costs nothing extra.
// On the host (tests/emu) it is a bool.
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long LtPred;
#define LT_P(c) __ballot(c)
#define LT_TRUE(p) __builtin_amdgcn_inverse_ballot_w64(p)
#define LT_NOT(p) (~(p))
#else
typedef bool LtPred;
#define LT_P(c) (c)
#define LT_TRUE(p) (p)
#define LT_NOT(p) (!(p))
#endif
#define LT_SEL(p, a, b) (LT_TRUE(p) ? (a) : (b))

// One table's answer for a query, the cheap way: every tag of the query's two buckets is XORed with the tag the query
// itself would have there (same choice, same scrambled half, the query's own other half).  x < 2^ob <=> the entry has the
// query's half (an empty slot or a tag of another half or choice differs above bit ob), and then x is the difference
// of the other halves: 0 = the query itself, one base pair set = distance 1.  Almost always at most one entry of a
// table has the query's half (several features would have to share a half): that entry is picked by priority select,
// `multi` reports the other case, which takes the general routine (lt_near1).
struct LtSide { uint32_t x, slot; LtPred any, multi, first; };
F2Q_HD LtSide lt_side(uint32_t want0, uint32_t want1, uint32_t limit, uint32_t b0, uint32_t b1, const U2 &e0, const U2 &e1)
{
    const uint32_t x0 = e0.x ^ want0, x1 = e0.y ^ want0, x2 = e1.x ^ want1, x3 = e1.y ^ want1;
    const LtPred s0 = LT_P(x0 < limit), s1 = LT_P(x1 < limit), s2 = LT_P(x2 < limit), s3 = LT_P(x3 < limit);
    LtSide r;
    r.x = LT_SEL(s0, x0, LT_SEL(s1, x1, LT_SEL(s2, x2, x3)));
    r.first = s0 | s1;
    r.any = r.first | s2 | s3;
    r.multi = (s0 & (s1 | s2 | s3)) | (s1 & (s2 | s3)) | (s2 & s3);
    const LtPred odd = (LT_NOT(s0) & s1) | (LT_NOT(r.first) & LT_NOT(s2));      // the second slot of its bucket
    r.slot = 2u * LT_SEL(r.first, b0, b1) + (LT_TRUE(odd) ? 1u : 0u);
    return r;
}
// The Counter-mode decision of one read from its four buckets: is it a perfect hit, is it the unique feature at
// distance 1, and the histogram slot to bump.  NEAR = run with --m 1 (false: --m 0, table 0 only, e[2..3] unused).
// rd0(bucket) reads a table-0 bucket (needed when the unique neighbour was found through table 1).
// T1SLOT (partitioned libraries): a hit found through table 1 is reported as its TABLE-1 slot with `via1` set, instead of
// being translated into the feature's table-0 slot (that slot lives in another partition's table); rd0 is then unused.
struct LtVerdict { LtPred perfect, imperfect, via1; uint32_t slot; };
template <bool NEAR, bool T1SLOT = false, class RD>
F2Q_HD LtVerdict lt_decide(const LtDesc &lt, const LtProbe &q, const U2 (&e)[4], uint32_t forced, RD rd0)
{
    const uint32_t w0 = lt_tag(q.cmp[0], q.h1, lt.hb1), w1 = lt_tag(q.cmp[1], q.h1, lt.hb1);
    const LtSide a = lt_side(w0, w1, 1u << lt.hb1, q.b[0], q.b[1], e[0], e[1]);
    LtSide b = a;
    uint32_t w2 = 0, w3 = 0;
    const LtPred unforced = LT_P(forced == 0u);
    // fast: at most one entry per table has the query's half.  Without NEAR a flagged read can match nothing.
    LtPred fast = LT_NOT(a.multi);
    if (!NEAR) fast = fast & unforced;
    LtPred perfect = unforced & fast & a.any & LT_P(a.x == 0u), near = perfect & LT_NOT(perfect);      // near = false
    LtPred through1 = near;                                              // false
    uint32_t slot = a.slot;
    if (NEAR) {
        w2 = lt_tag(q.cmp[2], q.h0, lt.hb0); w3 = lt_tag(q.cmp[3], q.h0, lt.hb0);
        b = lt_side(w2, w3, 1u << lt.hb0, q.b[2], q.b[3], e[2], e[3]);
        fast = fast & LT_NOT(b.multi);
        perfect = perfect & fast;
        const LtPred c0 = a.any & LT_P(ham2_32(a.x) == 1u), c1 = b.any & LT_P(ham2_32(b.x) == 1u);
        near = unforced & fast & LT_NOT(perfect) & (c0 ^ c1);            // exactly one feature at distance 1
        LtPred via1 = near & c1;
#if defined(__HIP_DEVICE_COMPILE__)
        const bool some_forced = LT_NOT(unforced) != 0ull;               // wave-uniform: most tiles skip this
#else
        const bool some_forced = forced != 0u;
#endif
        if (some_forced) {
            // ONE flagged base (a forced mismatch): the candidates are the
            // entries of the table bucketed by the CLEAN half whose other half agrees everywhere but at the flagged
            // base -- with at most one entry of the query's half per table (fast) that is one mask and one compare
            const uint32_t l0 = lt.hb0 >> 1;
            const uint32_t p = forced ? lt_ctz(forced) : 0u;
            const bool in0 = p < l0;                                      // flagged base in half 0: table 1 decides
            const uint32_t keep = ~(3u << (2u * (in0 ? p : p - l0)));
            const uint32_t xs = in0 ? b.x : a.x;
            const LtPred one = LT_P((forced & (forced - 1u)) == 0u);      // two flagged bases are two mismatches: nothing within --m 1
            const LtPred g = LT_NOT(unforced) & one & fast & LT_P((xs & keep) == 0u) & LT_P(in0 ? LT_TRUE(b.any) : LT_TRUE(a.any));
            near = near | g;
            via1 = via1 | (g & LT_P(in0));
        }
        if (T1SLOT) { slot = LT_SEL(via1, b.slot, slot); through1 = via1; }
        else if (LT_TRUE(via1)) slot = lt_slot_of_t1(lt, q, b.x ^ LT_SEL(b.first, w2, w3), rd0);
    }
    // several features sharing one of the query's halves: the general routine
    int sres = 0;
    bool sv1 = false;
    if (!LT_TRUE(fast)) {
        sres = R_NONALIGNED;
        const int ex = forced == 0u ? lt_exact(lt, q, e[0], e[1]) : -1;
        if (ex >= 0) { sres = R_PERFECT; slot = (uint32_t)ex; }
#if defined(__HIP_DEVICE_COMPILE__)
        else if (NEAR && __popc(forced) <= 1) {
#else
        else if (NEAR && __builtin_popcount(forced) <= 1) {
#endif
            uint32_t hit = 0, hitw = 0;
            if (lt_near1(lt, q, e, forced, hit, hitw) == 1u) {
                sres = R_IMPERFECT;
                if (T1SLOT) { slot = hit & 0x7FFFFFFFu; sv1 = (hit >> 31) != 0u; }
                else slot = (hit >> 31) ? lt_slot_of_t1(lt, q, hitw, rd0) : (hit & 0x7FFFFFFFu);
            }
        }
    }
    LtVerdict v;
    v.perfect = perfect | LT_P(sres == R_PERFECT);
    v.imperfect = near | LT_P(sres == R_IMPERFECT);
    v.via1 = through1 | LT_P(sv1);
    v.slot = slot;
    return v;
}

// ---------------------------------------------------------------------------------------------
// fast path, anchored (--us / --ds): one lane = one read held as two bit-planes.
// Planar tile layout (anchored runs): base rows 0..NW-1 hold the LOW bit of 32 bases per word,
// rows NW..2NW-1 the HIGH bit (A=00 C=01 G=10 T=11; bit i of word w = base 32w+i).
// The anchor search is bit-parallel over all start positions at once: for anchor symbol j the
// "differs from this symbol" plane is shifted right by j and added into bit-sliced saturating
// counters; positions whose count is <= k are hits and the first one is a count-trailing-zeros.
// ---------------------------------------------------------------------------------------------
F2Q_HD uint32_t funnel_shr(uint32_t hi, uint32_t lo, int sh)      // (hi:lo) >> sh, 0 <= sh <= 31
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
#else
    return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
#endif
}
F2Q_HD int ctz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}
// bits [lo, hi) of a 32-bit word (arguments may lie outside 0..32)
F2Q_HD uint32_t range_mask32(int lo, int hi)
{
    lo = lo < 0 ? 0 : lo; hi = hi > 32 ? 32 : hi;
    if (hi <= lo) return 0u;
    uint32_t upper = hi >= 32 ? ~0u : ((1u << hi) - 1u);
    return upper & (lo >= 32 ? 0u : (~0u << lo));
}

// one anchor symbol: add (plane >> j) into the bit-sliced saturating counters
template <int NW, int KB>
F2Q_HD void anchor_step(const uint32_t (&P)[NW], int j, uint32_t (&cnt)[KB > 0 ? KB : 1][NW], uint32_t (&ovf)[NW])
{
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint32_t x = funnel_shr(w + 1 < NW ? P[w + 1] : ~0u, P[w], j);   // beyond the planes: "differs"
        if (KB == 0) ovf[w] |= x;
        else {
#pragma unroll
            for (int b = 0; b < KB; b++) { uint32_t c = cnt[b][w] & x; cnt[b][w] ^= x; x = c; }
            ovf[w] |= x;
        }
    }
}

// two positions of the same plane at once: the low counter bit takes both through one full adder (sum and carry are
// single three-input instructions on gfx950), the carry ripples as before -- 5 instructions per word for two
// positions with one counter bit instead of 6, 9 instead of 16 with three
template <int NW, int KB>
F2Q_HD void anchor_step2(const uint32_t (&P)[NW], int j1, int j2, uint32_t (&cnt)[KB > 0 ? KB : 1][NW], uint32_t (&ovf)[NW])
{
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t up = w + 1 < NW ? P[w + 1] : ~0u;
        const uint32_t x1 = funnel_shr(up, P[w], j1), x2 = funnel_shr(up, P[w], j2);
        if (KB == 0) ovf[w] |= x1 | x2;
        else {
            const uint32_t c0 = cnt[0][w];
#if defined(__HIP_DEVICE_COMPILE__)
            uint32_t x = __builtin_amdgcn_bitop3_b32(c0, x1, x2, 0xE8);          // majority = the carry
            cnt[0][w] = __builtin_amdgcn_bitop3_b32(c0, x1, x2, 0x96);           // parity = the sum
#else
            uint32_t x = (c0 & x1) | (c0 & x2) | (x1 & x2);
            cnt[0][w] = c0 ^ x1 ^ x2;
#endif
#pragma unroll
            for (int b = 1; b < KB; b++) { uint32_t c = cnt[b][w] & x; cnt[b][w] ^= x; x = c; }
            ovf[w] |= x;
        }
    }
}
template <int NW, int KB>
F2Q_HD void anchor_steps(const uint32_t (&P)[NW], uint32_t m, uint32_t (&cnt)[KB > 0 ? KB : 1][NW], uint32_t (&ovf)[NW])
{
    // pairs in a branch-free loop body, the odd position after it (a two-way body costs a register shuffle per pass)
    while (m & (m - 1u)) {
        const int j1 = ctz32(m); m &= m - 1u;
        const int j2 = ctz32(m); m &= m - 1u;
        anchor_step2<NW, KB>(P, j1, j2, cnt, ovf);
    }
    if (m) anchor_step<NW, KB>(P, ctz32(m), cnt, ovf);
}

// bit-sliced "count <= k" over the counters of one anchor
template <int NW, int KB>
F2Q_HD void anchor_compare(const uint32_t (&cnt)[KB > 0 ? KB : 1][NW], const uint32_t (&ovf)[NW], int k, uint32_t (&hit)[NW])
{
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint32_t gt = 0, eq = ~0u;
#pragma unroll
        for (int b = KB - 1; b >= 0; b--) {
            if ((k >> b) & 1) eq &= cnt[b][w];
            else { gt |= eq & cnt[b][w]; eq &= ~cnt[b][w]; }
        }
        hit[w] = ~gt & ~ovf[w];
    }
}

// Hit vectors of the upstream and/or downstream anchor in one pass: positions p at which the anchor sits
// within k mismatches (KB counter bits: 0 exact, 1: k <= 1, 3: k <= 7).  For each of the four symbols the
// "base differs from it" plane is formed once and then shifted and counted for every position of either
// anchor that holds the symbol (run.up_pos / run.down_pos are per-symbol position masks), so one plane is
// live at a time and the scalar loop runs once per anchor symbol.  All loops/branches are wave-uniform.
template <int NW, int KBU, int KBD>
F2Q_HD void anchor_hits2p(const RunDev &run, const uint32_t *up_pos, const uint32_t *down_pos,
                          const uint32_t (&LO)[NW], const uint32_t (&HI)[NW], const uint32_t (&FLG)[NW],
                          uint32_t (&hu)[NW], uint32_t (&hd)[NW])
{
    uint32_t cu[KBU > 0 ? KBU : 1][NW], ou[NW], cd[KBD > 0 ? KBD : 1][NW], od[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        ou[w] = 0; od[w] = 0;
#pragma unroll
        for (int b = 0; b < (KBU > 0 ? KBU : 1); b++) cu[b][w] = 0;
#pragma unroll
        for (int b = 0; b < (KBD > 0 ? KBD : 1); b++) cd[b][w] = 0;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t la = (c & 1) ? ~0u : 0u, ha = (c & 2) ? ~0u : 0u;
        uint32_t P[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) P[w] = (LO[w] ^ la) | (HI[w] ^ ha) | FLG[w];   // 1 = base is not symbol c (flagged: never)
        if (run.has_up) anchor_steps<NW, KBU>(P, up_pos[c], cu, ou);
        if (run.has_down) anchor_steps<NW, KBD>(P, down_pos[c], cd, od);
    }
    anchor_compare<NW, KBU>(cu, ou, run.msu, hu);
    anchor_compare<NW, KBD>(cd, od, run.msd, hd);
}
template <int NW, int KBU, int KBD>
F2Q_HD void anchor_hits2(const RunDev &run, const uint32_t (&LO)[NW], const uint32_t (&HI)[NW], const uint32_t (&FLG)[NW],
                         uint32_t (&hu)[NW], uint32_t (&hd)[NW])
{
    anchor_hits2p<NW, KBU, KBD>(run, run.up_pos, run.down_pos, LO, HI, FLG, hu, hd);
}

// first set bit at a position in [from, to] (inclusive) or -1
template <int NW>
F2Q_HD int first_hit(const uint32_t (&hit)[NW], int from, int to)
{
    int pos = -1;
#pragma unroll
    for (int w = NW - 1; w >= 0; w--) {
        const uint32_t m = hit[w] & range_mask32(from - 32 * w, to + 1 - 32 * w);
        if (m) pos = 32 * w + ctz32(m);
    }
    return pos;
}

// any set bit of F in [a, b)
template <int NW>
F2Q_HD bool any_in_range(const uint32_t (&F)[NW], int a, int b)
{
    uint32_t acc = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) acc |= F[w] & range_mask32(a - 32 * w, b - 32 * w);
    return acc != 0;
}

// L (<= 32) bits of a plane starting at bit `start` (start + L may pass the end: zeros)
template <int NW>
F2Q_HD uint32_t plane_extract(const uint32_t (&P)[NW], int start, int L)
{
    // every adjacent word pair is funnel-shifted by the in-word offset (the shift amount is per lane), then the word
    // index picks one of them through a 3-level select tree on its bits: NW + 4 instructions + 3 compares instead of
    // a compare-and-select pair per word for each of the two source words.  Needs 0 <= start < 32 * (NW <= 8 ? 8 : 16).
    static_assert(NW <= 16, "plane_extract: at most 16 words");
    const uint32_t sh = (uint32_t)start & 31u, wi = (uint32_t)start >> 5;
    constexpr int NV = NW <= 8 ? 8 : 16;
    uint32_t V[NV];
#pragma unroll
    for (int w = 0; w < NV; w++) V[w] = w < NW ? funnel_shr(w + 1 < NW ? P[w + 1] : 0u, P[w], sh) : 0u;
    const bool b0 = wi & 1u, b1 = wi & 2u, b2 = wi & 4u;
    const uint32_t t0 = b0 ? V[1] : V[0], t1 = b0 ? V[3] : V[2], t2 = b0 ? V[5] : V[4], t3 = b0 ? V[7] : V[6];
    const uint32_t u0 = b1 ? t1 : t0, u1 = b1 ? t3 : t2;
    uint32_t v = b2 ? u1 : u0;
    if (NV == 16) {                                        // reads of 161 .. 320 bases: one more level of the tree
        const bool b3 = wi & 8u;
        const uint32_t t4 = b0 ? V[NV - 7] : V[NV - 8], t5 = b0 ? V[NV - 5] : V[NV - 6], t6 = b0 ? V[NV - 3] : V[NV - 4], t7 = b0 ? V[NV - 1] : V[NV - 2];
        const uint32_t u2 = b1 ? t5 : t4, u3 = b1 ? t7 : t6;
        const uint32_t vh = b2 ? u3 : u2;
        v = b3 ? vh : v;
    }
    return L >= 32 ? v : (v & ((1u << L) - 1u));
}

// Phred fail bits of 32 bases from their 8 quality words (bytes < 128); add_hi == 0: rule off
// Planar tiles keep the quality bytes of each 32-base group transposed: byte b of the group's word w holds base
// 32g + 8b + w (planar_qpos).  The SWAR test leaves its verdict in bit 7 of every byte, so word w shifted right by
// 7 - w drops its four verdicts on bits w, 8 + w, 16 + w, 24 + w: eight shift-ORs build the 32-base fail word, no
// bit gather (the multiply it took before is a quarter-rate instruction).
F2Q_HD uint32_t planar_qpos(uint32_t w, uint32_t b) { return 32u * (w >> 3) + 8u * b + (w & 7u); }
F2Q_HD uint32_t planar_qword(uint32_t pos) { return 8u * (pos >> 5) + (pos & 7u); }
F2Q_HD uint32_t planar_qbyte(uint32_t pos) { return (pos & 31u) >> 3; }
// STRIP = false: the caller knows that no byte carries a flag (bit 7), e.g. no read of the wave is flagged
template <bool STRIP = true>
F2Q_HD uint32_t fail_word8(const uint32_t (&q)[8], uint32_t add_hi)
{
    if (!add_hi) return 0u;
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        f |= qfail4(STRIP ? (q[i] & 0x7F7F7F7Fu) : q[i], 0x5F5F5F5Fu, add_hi, 0x80808080u) >> (7 - i);   // bit 7 = flag, not quality
    return f;
}
F2Q_HD uint32_t flag_word8(const uint32_t (&q)[8])
{
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) f |= (q[i] & 0x80808080u) >> (7 - i);
    return f;
}
F2Q_HD uint32_t phred_add_hi(int thr) { return thr >= 33 ? (uint32_t)(127 - thr) * 0x01010101u : 0u; }

// result of the extraction stage of one packed read in an anchored run
struct AnchorWin { int ok; int start, end; };      // ok: 0 = no window (counts as quality-failed, :345-347),
                                                   //     1 = window [start,end) passed every Phred test, 2 = take the slow routine

// sequence_tinder (:215-285) + the window Phred test (:357) on bit-planes.  FU/FD/FW: fail vectors
// for --qsu / --qsd / --ph (they may alias when the thresholds coincide).
// any set bit of F in [a, a+len), len <= 32 (anchors are <= 32 symbols, fast-path windows <= 31)
template <int NW>
F2Q_HD bool any_fail_short(const uint32_t (&F)[NW], int a, int len)
{
    return len > 0 && plane_extract<NW>(F, a, len) != 0u;
}

template <int NW, int KBU, int KBD>
F2Q_HD AnchorWin anchor_window_pair(const RunDev &run, int su, int sd, const uint32_t *up_pos, const uint32_t *down_pos,
                                    const uint32_t (&LO)[NW], const uint32_t (&HI)[NW],
                                    const uint32_t (&FLG)[NW], int r,
                                    const uint32_t (&FU)[NW], const uint32_t (&FD)[NW], const uint32_t (&FW)[NW])
{
    AnchorWin out; out.ok = 0; out.start = 0; out.end = 0;
    int start, end;
    uint32_t hu[NW], hd[NW];
    anchor_hits2p<NW, KBU, KBD>(run, up_pos, down_pos, LO, HI, FLG, hu, hd);
    if (run.has_up && run.has_down) {
        const int pu = first_hit<NW>(hu, 0, r - su);
        if (pu < 0) return out;
        const int pd = first_hit<NW>(hd, pu + su, r - sd);
        if (pd < 0) return out;
        if (any_fail_short<NW>(FU, pu, su) || any_fail_short<NW>(FD, pd, sd)) return out;
        start = pu + su; end = pd;
    } else if (run.has_up) {
        const int pu = first_hit<NW>(hu, 0, r - su);
        if (pu < 0) return out;
        if (any_fail_short<NW>(FU, pu, su)) return out;
        start = pu + su; end = start + run.length;
        if (run.length < 0) { out.ok = 2; return out; }
        if (end > r) end = r;                                   // Python slice clipping (:354)
    } else {
        const int pd = first_hit<NW>(hd, 0, r - sd);
        if (pd < 0) return out;
        if (any_fail_short<NW>(FD, pd, sd)) return out;
        start = pd - run.length; end = pd;
        if (start < 0 || run.length < 0) { out.ok = 2; return out; }   // negative-index slice: rare, slow routine
    }
    if (end - start <= 32 ? any_fail_short<NW>(FW, start, end - start) : any_in_range<NW>(FW, start, end)) return out;   // :357
    out.ok = 1; out.start = start; out.end = end;
    return out;
}
template <int NW, int KBU, int KBD>
F2Q_HD AnchorWin anchor_window(const RunDev &run, const uint32_t (&LO)[NW], const uint32_t (&HI)[NW],
                               const uint32_t (&FLG)[NW], int r,
                               const uint32_t (&FU)[NW], const uint32_t (&FD)[NW], const uint32_t (&FW)[NW])
{
    return anchor_window_pair<NW, KBU, KBD>(run, run.up_len[0], run.down_len[0], run.up_pos, run.down_pos, LO, HI, FLG, r, FU, FD, FW);
}

// One read of a run with several --us/--ds pairs, on the planes (fast2q.py:333-363): every pair is searched on its own,
// the windows that pass are joined with ':' in pair order, pairs that fail are left out; no window at all counts as a
// quality failure.  The joined key is spelt out (flag bits read 'N') and matched as a string: Counter mode through
// match_key (the byte-string index when the library holds ':' features), Extract+Count through ec_count_key.
// Returns the reference counter to bump (1..4) with the feature in idx, 0 after an Extract+Count insert, or -1: a
// pair needs a negative-index slice or the key outgrows the buffer -- the byte-exact routine takes the read.
// kb: F2Q_PAIRS_KEYMAX bytes of the lane's own (the kernel hands out LDS, 68-byte stride: an odd number of words; two
// parts of 31 bases and their ':' fit, a longer key takes the byte-exact routine)
#define F2Q_PAIRS_KEYMAX 68
// 2-bit interleaved key of window [start, start+L) (L <= 31) from the planes
template <int NW>
F2Q_HD uint64_t plane_key(const uint32_t (&LO)[NW], const uint32_t (&HI)[NW], int start, int L)
{
    return spread32(plane_extract<NW>(LO, start, L)) | (spread32(plane_extract<NW>(HI, start, L)) << 1);
}

// ---- pair tables (PwDesc): the joined key of a two-pair run as two 2-bit words --------------------------------------
#define F2Q_PW_MAXLEN 20
#define F2Q_PW_AMASK ((1ull << 40) - 1ull)
F2Q_HD uint32_t pw_hash(uint64_t a, uint64_t b, uint32_t bits) { return hash32(a * 0x9E3779B97F4A7C15ull + (b ^ (b >> 17)) * 0xC2B2AE3D27D4EB4Full, bits); }
// The Counter-mode verdict for the key A:B (fa / fb: one bit per base of A / B that is no ACGT symbol -- a forced
// mismatch): exact hit (fast2q.py:365-367), else with --m 1 the unique feature at distance 1 (:692-750).  idx = feature.
F2Q_HD int pw_decide(const PwDesc &pw, uint64_t a, uint64_t b, uint32_t fa, uint32_t fb, int miss, uint32_t &idx)
{
    const uint32_t mask = (1u << pw.bits) - 1u;
    const uint64_t *t0 = pw.tab, *t1 = pw.tab + ((size_t)2 << pw.bits), *t2 = pw.tab + ((size_t)4 << pw.bits);
    if (!(fa | fb)) {
        for (uint32_t s = pw_hash(a, b, pw.bits);; s = (s + 1u) & mask) {
            const uint64_t w0 = gp(t0)[2u * s], w1 = gp(t0)[2u * s + 1u];
            if (w1 == ~0ull) break;
            if ((w0 & F2Q_PW_AMASK) == a && w1 == b) { idx = (uint32_t)(w0 >> 40); return 1; }
        }
    }
    if (miss < 1) return 3;
    const int nf = popc64(fa) + popc64(fb);
    if (nf > 1) return 3;
    uint32_t cnt = 0;
    if (!fa) {                                   // the features with this A: is their B one base (the flagged one) away?
        const uint64_t keep = ~(spread32(fb) * 3ull);
        for (uint32_t s = pw_hash(a, 0ull, pw.bits);; s = (s + 1u) & mask) {
            const uint64_t w0 = gp(t1)[2u * s], w1 = gp(t1)[2u * s + 1u];
            if (w1 == ~0ull) break;
            if ((w0 & F2Q_PW_AMASK) == a && ham2((w1 ^ b) & keep) + nf == 1) { cnt++; idx = (uint32_t)(w0 >> 40); }
        }
    }
    if (!fb) {                                   // ... and those with this B
        const uint64_t keep = ~(spread32(fa) * 3ull);
        for (uint32_t s = pw_hash(b, 1ull, pw.bits);; s = (s + 1u) & mask) {
            const uint64_t w0 = gp(t2)[2u * s], w1 = gp(t2)[2u * s + 1u];
            if (w1 == ~0ull) break;
            if (w1 == b && ham2(((w0 & F2Q_PW_AMASK) ^ a) & keep) + nf == 1) { cnt++; idx = (uint32_t)(w0 >> 40); }
        }
    }
    return cnt == 1u ? 2 : 3;
}

template <int NW, int KB>
F2Q_HD int pairs_lane(const RunDev &run, const LibDev &lib, const EcDev &ec, uint8_t *kb,
                      const uint32_t (&LO)[NW], const uint32_t (&HI)[NW], const uint32_t (&FLG)[NW], int r,
                      const uint32_t (&FU)[NW], const uint32_t (&FD)[NW], const uint32_t (&FW)[NW],
                      unsigned long long read_index, uint32_t &idx, uint32_t *n_new, bool keyflag = true)
{   // keyflag = false: the marks of FLG are lower-case bases (F2Q_LEN_CASE): mismatches for the anchors, plain bases in the key
    if (run.mode == 0 && lib.pw.ok && run.n_iter == 2) {
        // two pairs against a library of A:B features: the joined key as two 2-bit words (PwDesc), no string
        const PwDesc &pw = lib.pw;
        uint64_t key[2] = {0, 0}; uint32_t fl[2] = {0, 0}; int len[2] = {0, 0}, np = 0;
        for (int i = 0; i < 2; i++) {
            const AnchorWin aw = anchor_window_pair<NW, KB, KB>(run, run.up_len[i], run.down_len[i], run.mp_up_pos[i], run.mp_down_pos[i],
                                                                LO, HI, FLG, r, FU, FD, FW);
            if (aw.ok == 2) return -1;
            if (aw.ok != 1) continue;
            const int L = aw.end - aw.start;
            if (L > F2Q_REG_MAXLEN) return -1;                            // (no 2-bit form: the byte-exact routine)
            key[np] = plane_key<NW>(LO, HI, aw.start, L);
            fl[np] = keyflag ? plane_extract<NW>(FLG, aw.start, L) : 0u;
            len[np] = L; np++;
        }
        if (!np) return 4;                                                // :389-390
        if (np == 2) {
            // parts of other lengths: the ':' of the key and of every feature sit at different places, two mismatches at least
            if (len[0] != (int)pw.la || len[1] != (int)pw.lb) return 3;
            return pw_decide(pw, key[0], key[1], fl[0], fl[1], run.miss, idx);
        }
        // one part (the other pair was not found): a plain key.  As long as A:B it can be one substitution away from a
        // feature -- the base where the feature has its ':' -- when everything else agrees
        if (len[0] == (int)(pw.la + 1u + pw.lb) && run.miss >= 1 && (fl[0] & ~(1u << pw.la)) == 0u) {
            const uint64_t a = key[0] & ((1ull << (2u * pw.la)) - 1ull), b = key[0] >> (2u * (pw.la + 1u));
            return pw_decide(pw, a, b, 0u, 0u, 0, idx) == 1 ? 2 : 3;
        }
        return 3;
    }
    int klen = 0, nparts = 0;
    for (int i = 0; i < run.n_iter; i++) {
        const AnchorWin aw = anchor_window_pair<NW, KB, KB>(run, run.up_len[i], run.down_len[i], run.mp_up_pos[i], run.mp_down_pos[i],
                                                            LO, HI, FLG, r, FU, FD, FW);
        if (aw.ok == 2) return -1;
        if (aw.ok != 1) continue;
        const int L = aw.end - aw.start;
        if (klen + L + 1 > F2Q_PAIRS_KEYMAX) return -1;
        if (nparts) kb[klen++] = (uint8_t)':';
        for (int off = 0; off < L; off += 32) {
            const int n = L - off < 32 ? L - off : 32;
            const uint32_t lo = plane_extract<NW>(LO, aw.start + off, n), hi = plane_extract<NW>(HI, aw.start + off, n);
            const uint32_t fl = keyflag ? plane_extract<NW>(FLG, aw.start + off, n) : 0u;
            for (int j = 0; j < n; j++)
                kb[klen++] = ((fl >> j) & 1u) ? (uint8_t)'N' : (uint8_t)(0x54474341u >> (8u * (((lo >> j) & 1u) | (((hi >> j) & 1u) << 1))));
        }
        nparts++;
    }
    if (!nparts) return 4;                                               // :389-390
    KeyView kv; kv.seq = kb; kv.nseg = 1; kv.a[0] = 0; kv.b[0] = klen; kv.len = klen;
    if (run.mode == 0) return match_key<true>(run, lib, kv, idx);
    ec_count_key(ec, kv, read_index, n_new);
    return 0;
}


// ---------------------------------------------------------------------------------------------
// packing: which reads the tile planes can carry, and how one read is laid into them.  Shared by the
// host packer (f2q_host.h) and the device packer (k_pack in f2q_aux_kernels.h).
// ---------------------------------------------------------------------------------------------
#define F2Q_PACK_MAXLEN 512
#define F2Q_ANCHOR_MAXLEN 320      // longest read the packed anchored kernels hold in registers (10 x 32 bases: MiSeq 2 x 300 amplicons)

struct PackPlan {
    bool fast_fixed = false;       // fixed offset, one window of 0..31 bases (Counter mode; Extract+Count: 0..29), or `multi`
    bool multi = false;            // fixed offset, 2..4 windows of n bases each, k * n <= 31, Counter mode
    int need = 0;                  // fixed mode: bases [0, need) are all the fast kernel can touch
    int from = 0;                  // ... and only [from, need) is ever looked at
    // `multi`: the tiles hold ONLY the windows' bases and quality bytes, window w at stored positions [w * win_len,
    // (w + 1) * win_len) -- windows far apart in the read (dual-guide vectors) cost the kernels no more rows than one
    // window of n_win * win_len bases; need = n_win * win_len, from = 0.  A read that ends inside a window is not packed.
    int n_win = 0, win_len = 0, win_end = 0;       // win_end: every window lies inside [0, win_end) of the read
    int win_start[F2Q_MW_MAX] = {0, 0, 0, 0};
    bool inband_n = false;         // non-ACGT symbols travel as flag bits (all-ACGT library only)
    bool n_only = false;           // ... but only the symbol 'N' (Extract+Count: the key spells the symbol, a flag reads 'N'; fixed windows: 'n' too, upper-cased)
    bool fast_anchor = false;      // --us/--ds with ACGT anchors: packed bit-plane path
    bool multi_pair = false;       // ... several pairs: k_count_anchor_pairs (one search per pair on the same planes)
    int kb = 1;                    // counter bits of the anchor search (0: exact, 1: k <= 1, 3: k <= 7)
};

template <class P>
struct RecT { P seq; P qual; uint32_t len, qlen; };
// read position of stored position s (fixed-offset tiles)
F2Q_HD uint32_t pack_src(const PackPlan &pl, uint32_t s) { return pl.n_win ? (uint32_t)pl.win_start[s / (uint32_t)pl.win_len] + s % (uint32_t)pl.win_len : s; }

// Can this read go through a packed fast path?  The planes cannot carry a quality line of another length or
// quality bytes >= 128 (bit 7 is the flag bit and the Phred SWAR test relies on 7-bit bytes); non-ACGT symbols
// only as flag bits (all-ACGT library); anchored runs carry lower-case bases as marked bases (F2Q_LEN_CASE), but not
// next to non-ACGT symbols in the same read.
template <class P>
F2Q_HD bool read_is_clean(const PackPlan &pl, const RecT<P> &r)
{
    if (pl.fast_anchor) {
        if (r.qlen != r.len || r.len > F2Q_ANCHOR_MAXLEN) return false;
        bool lower = false, odd = false;
        for (uint32_t j = 0; j < r.len; j++) {
            if (r.qual[j] & 0x80) return false;
            const uint8_t c = r.seq[j];
            if (base_code(c) > 3u) {
                if (c == 'a' || c == 'c' || c == 'g' || c == 't') lower = true;          // marked; its code is stored (F2Q_LEN_CASE)
                else { if (!pl.inband_n || (pl.n_only && c != 'N')) return false; odd = true; }
            }
        }
        return !(lower && odd);                            // one kind of mark per read: both kinds take the byte-exact routine
    }
    if (!pl.fast_fixed) return false;
    if (r.qlen != r.len) return false;
    if (pl.n_win && r.len < (uint32_t)pl.win_end) return false;              // a window the read ends in: the byte-exact routine clips it
    const uint32_t b = pl.n_win ? (uint32_t)pl.need : (r.len < (uint32_t)pl.need ? r.len : (uint32_t)pl.need);
    uint32_t nmask = 0;
    for (uint32_t j = (uint32_t)pl.from; j < b; j++) {
        const uint32_t at = pack_src(pl, j);
        const uint8_t c = up8(r.seq[at]);                                     // the window is upper-cased (:354)
        if (base_code(c) > 3u) {
            if (!pl.inband_n || (pl.n_only && c != 'N')) return false;
            nmask |= 1u << ((j - (uint32_t)pl.from) & 31u);
        }
        if (r.qual[at] & 0x80) return false;
    }
    // Extract+Count: the key spells its 'N's, and the packed kernels make single-word keys only
    if (pl.n_only && nmask && !ec64_fits(nmask, (int)(b - (uint32_t)pl.from))) return false;
    return true;
}

// number of bases of the read that are stored
template <class P>
F2Q_HD uint32_t packed_len(const PackPlan &pl, const RecT<P> &r)
{
    return pl.fast_anchor ? r.len : pl.n_win ? (uint32_t)pl.need : (r.len < (uint32_t)pl.need ? r.len : (uint32_t)pl.need);
}

// tile geometry for a block whose longest stored read is rmax_in
F2Q_HD void tile_geometry(const PackPlan &pl, uint32_t rmax_in, uint32_t &rmax, uint32_t &planar_nw, uint32_t &wb, uint32_t &wq)
{
    rmax = rmax_in ? rmax_in : 1u;
    if (pl.fast_anchor) { planar_nw = rmax <= 96u ? 3u : rmax <= 160u ? 5u : 10u; wb = 2u * planar_nw; wq = 8u * planar_nw; }
    else { planar_nw = 0; wb = (rmax + 15u) / 16u; wq = (rmax + 3u) / 4u; }
}

// one clean read -> its words (sink.base(w, v), sink.qual(w, v), sink.len(v))
template <class P, class Sink>
F2Q_HD void pack_read(const PackPlan &pl, const RecT<P> &r, uint32_t planar_nw, Sink &sink)
{
    const uint32_t l = packed_len(pl, r);
    const uint32_t from = pl.fast_anchor ? 0u : (uint32_t)pl.from;
    bool flagged = false, lower = false;
    if (planar_nw) {
        for (uint32_t w = 0; w * 32 < l; w++) {
            uint32_t lo = 0, hi = 0;
            for (uint32_t j = 0; j < 32 && w * 32 + j < l; j++) {
                uint32_t c = base_code(r.seq[w * 32 + j]);
                if (c > 3u) {
                    flagged = true;
                    c = base_code(up8(r.seq[w * 32 + j]));          // a lower-case base keeps its code (the key is upper-cased) ...
                    if (c > 3u) c = 0; else lower = true;          // ... anything else is stored as 'A'
                }
                lo |= (c & 1u) << j; hi |= ((c >> 1) & 1u) << j;
            }
            sink.base(w, lo); sink.base(planar_nw + w, hi);
        }
    } else {
        for (uint32_t w = 0; w * 16 < l; w++) {
            uint32_t v = 0;
            for (uint32_t j = 0; j < 16 && w * 16 + j < l; j++) {
                uint32_t c = base_code(up8(r.seq[pack_src(pl, w * 16 + j)]));     // non-ACGT: stored as 'A'; inside the window it is
                if (c > 3u) { c = 0; flagged |= (w * 16 + j >= from); }   // flagged, outside it is never looked at
                v |= c << (2 * j);
            }
            sink.base(w, v);
        }
    }
    const uint32_t n_qwords = planar_nw ? 8u * ((l + 31u) / 32u) : (l + 3u) / 4u;
    for (uint32_t w = 0; w < n_qwords; w++) {
        uint32_t v = 0;
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t pos = planar_nw ? planar_qpos(w, j) : w * 4 + j;     // planar tiles: transposed groups (fail_word8)
            if (pos >= l) continue;
            const uint32_t at = planar_nw ? pos : pack_src(pl, pos);
            uint32_t q = r.qual[at];
            q = (q & 0x80u) ? 0u : q;                               // keep every stored byte 7-bit (SWAR)
            const uint8_t sc = planar_nw ? (uint8_t)r.seq[at] : up8(r.seq[at]);
            if (pos >= from && base_code(sc) > 3u) q |= 0x80u;      // flag bit
            v |= q << (8 * j);
        }
        sink.qual(w, v);
    }
    sink.len(l | (flagged ? F2Q_LEN_FLAG : 0u) | (lower ? F2Q_LEN_CASE : 0u));      // (read_is_clean: never both kinds of marks)
}

} // namespace f2q
