// f2q_host.h -- host-side helpers of libf2q_hip.so: library index construction and FASTQ
// framing / read classification / tile packing.  Plain C++ (no HIP calls) so the same code is
// unit-tested on a GPU-less machine through tests/emu.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/f2q.h"
#include "f2q_device.h"

namespace f2q {

// ---------------------------------------------------------------------------------------------
// library index (replaces binary_converter, fast2q.py:188-213)
// ---------------------------------------------------------------------------------------------
struct HostIndex {
    std::vector<uint64_t> tab_keys;
    std::vector<uint32_t> tab_idx;
    std::vector<uint8_t> feat_bytes;
    std::vector<uint32_t> feat_off;
    std::vector<uint32_t> irr_ids;
    std::vector<uint64_t> key2;          // 2-bit key per feature (0 for irregular ones)
    std::vector<uint64_t> ptab;          // packed slots of the run's feature length (v2 fast kernel)
    PackedGroup pk;
    PackedGroup mpk[F2Q_MW_MAX];         // multi-window runs: tables of the k-part features (index k - 1)
    uint32_t mw_ok = 0;
    // LDS tables (f2q_device.h: LtDesc); lt.ok == 0: not applicable / the cuckoo build failed
    LtDesc lt{};
    std::vector<uint32_t> lt_tags, lt_feat_of;
    std::vector<uint16_t> lt_slot_of;
    // partitioned tables (f2q_device.h: PtDesc); pt.ok == 0: not applicable / not needed / the build failed
    PtDesc pt{};
    int pt_force_parts = 0;              // > 0: build that many partitions whatever the library's size (set before build_index)
    std::vector<uint32_t> pt_tags0, pt_tags1, pt_pstart, pt_slot1_of, pt_feat_of, pt_feat0_of;
    std::vector<uint16_t> pt_slot0_of;
    // general keys (GkDesc): byte-string index of all features by length
    std::vector<GkGroup> gk_groups;
    std::vector<uint32_t> gk_tab, gk_ids, gk_fwoff;
    std::vector<unsigned long long> gk_fw;
    LenGroup grp[F2Q_REG_MAXLEN + 1];
    uint32_t n_features = 0, n_irregular = 0;
    // pair tables (f2q_device.h: PwDesc); pw.ok == 0: the library is not a pure A:B library
    PwDesc pw{};
    std::vector<uint64_t> pw_tab;
};

inline bool feature_key(const uint8_t *s, uint32_t n, uint64_t &key)
{
    if (n < 1 || n > F2Q_REG_MAXLEN) return false;
    key = 0;
    for (uint32_t j = 0; j < n; j++) {
        uint32_t c = base_code(s[j]);
        if (c > 3u) return false;
        key |= (uint64_t)c << (2 * j);
    }
    return true;
}

inline uint32_t table_bits(uint32_t n)
{
    uint32_t bits = 4;
    while ((1ull << bits) < 2ull * n) bits++;
    return bits;
}

inline void table_insert(std::vector<uint64_t> &keys, std::vector<uint32_t> &idx, const PieceDesc &pd,
                         uint64_t hashed, uint64_t full_key, uint32_t id)
{
    uint32_t m = (1u << pd.bits) - 1u, s = hash_slot(hashed, pd.bits);
    while (keys[pd.off + s] != KEY_EMPTY) s = (s + 1) & m;
    keys[pd.off + s] = full_key; idx[pd.off + s] = id;
}

inline void packed_insert(std::vector<uint64_t> &tab, const PackedPiece &pd, uint64_t hashed, uint64_t slot)
{
    uint32_t m = (1u << pd.bits) - 1u, s = packed_start(hashed, pd.bits);   // (lookups fetch aligned pairs of slots)
    while (tab[pd.off + s] != KEY_EMPTY) s = (s + 1) & m;
    tab[pd.off + s] = slot;
}

// packed tables for features of `len` bases (the window length of a fixed-offset run; k windows' worth for the k-part
// features of a multi-window run): built when key bits + index bits fit one u64 and the pigeonhole pieces are
// regular.  The tables are appended to ix.ptab.
inline void build_packed(HostIndex &ix, const std::vector<uint32_t> &ids, int len, int miss, PackedGroup &g)
{
    memset(&g, 0, sizeof g);
    if (ix.ptab.empty()) ix.ptab.assign(1, KEY_EMPTY);
    if (len < 1 || len > F2Q_REG_MAXLEN || ids.empty()) return;
    uint32_t ib = 1;
    while ((1ull << ib) < (uint64_t)ix.n_features + 1) ib++;
    if (2 * len + (int)ib > 64) return;
    const int P = miss > 0 ? miss + 1 : 0;
    if (P > len || P > F2Q_MAX_PIECES) return;
    const uint32_t bits = table_bits((uint32_t)ids.size()) + 1;          // load factor <= 0.25
    if (bits > 30) return;
    g.len = (uint32_t)len; g.ib = ib; g.n_pieces = (uint32_t)P;
    if (ix.ptab.size() & 1u) ix.ptab.push_back(KEY_EMPTY);              // tables start at even slots: 16-byte aligned pairs
    uint32_t off = (uint32_t)ix.ptab.size();
    g.exact.off = off; g.exact.bits = bits; g.exact.shift = 0; g.exact.mask = ~0ull; off += 1u << bits;
    for (int p = 0; p < P; p++) {
        int b0 = (int)((long)p * len / P), b1 = (int)((long)(p + 1) * len / P);
        g.piece[p].off = off; g.piece[p].bits = bits; off += 1u << bits;
        g.piece[p].shift = (uint32_t)(2 * b0);
        g.piece[p].mask = (b1 - b0 >= 32) ? ~0ull : ((1ull << (2 * (b1 - b0))) - 1ull);
    }
    ix.ptab.resize(off, KEY_EMPTY);
    for (uint32_t f : ids) {
        const uint64_t k = ix.key2[f], slot = (k << ib) | f;
        packed_insert(ix.ptab, g.exact, k, slot);
        for (int p = 0; p < P; p++) packed_insert(ix.ptab, g.piece[p], (k >> g.piece[p].shift) & g.piece[p].mask, slot);
    }
}

// a k-part feature of a multi-window run: k runs of exactly `l` ACGT symbols joined by ':' (what the ':'-joined
// windows of a read look like, fast2q.py:358-363); key = the k*l bases, 2 bits each
inline int parts_key(const uint8_t *s, uint32_t n, int l, int max_parts, uint64_t &key)
{
    if (l < 1 || (n + 1) % (uint32_t)(l + 1) != 0) return 0;
    const int k = (int)((n + 1) / (uint32_t)(l + 1));
    if (k < 2 || k > max_parts || k * l > F2Q_REG_MAXLEN) return 0;
    key = 0;
    int b = 0;
    for (uint32_t j = 0; j < n; j++) {
        const bool sep = (j % (uint32_t)(l + 1)) == (uint32_t)l;
        if (sep) { if (s[j] != (uint8_t)':') return 0; continue; }
        const uint32_t c = base_code(s[j]);
        if (c > 3u) return 0;
        key |= (uint64_t)c << (2 * b++);
    }
    return k;
}

// Cuckoo insertion of one tag table (f2q_device.h, "LDS tables"): every feature of `ids` goes into one of the two buckets
// (of two slots) its half t selects, bb bucket-index bits; an occupied pair of buckets evicts a random resident, which
// moves to its other bucket, up to a bounded number of moves.  tg / ow: 2 << bb tags and the feature held by each slot.
inline bool lt_cuckoo(const HostIndex &ix, const std::vector<uint32_t> &ids, int t, uint32_t hb0, uint32_t hb1, uint32_t bb,
                      uint32_t *tg, uint32_t *ow)
{
    const uint32_t hb = t ? hb1 : hb0, ob = t ? hb0 : hb1;
    auto halves = [&](uint32_t f, uint32_t &h, uint32_t &o) {
        const uint64_t k = ix.key2[f];
        const uint32_t h0 = (uint32_t)k & ((1u << hb0) - 1u), h1 = (uint32_t)(k >> hb0);
        h = t ? h1 : h0; o = t ? h0 : h1;
    };
    uint64_t rs = 0x9E3779B97F4A7C15ull + (uint64_t)t;       // fixed seed: the tables are a pure function of the library
    for (uint32_t f0 : ids) {
        uint32_t f = f0, from = ~0u;                 // slot the feature in hand was evicted from
        bool placed = false;
        for (int moves = 0; moves < 20000 && !placed; moves++) {
            uint32_t h, o; halves(f, h, o);
            uint32_t b[2], cmp[2];
            lt_hash(h, hb, ob, 0, b[0], cmp[0], bb); lt_hash(h, hb, ob, 1, b[1], cmp[1], bb);
            for (int c = 0; c < 2 && !placed; c++)
                for (int i = 0; i < 2 && !placed; i++) {
                    const uint32_t s = 2u * b[c] + (uint32_t)i;
                    if (tg[s] == F2Q_LT_EMPTY) { tg[s] = lt_tag(cmp[c], o, ob); ow[s] = f; placed = true; }
                }
            if (placed) break;
            // random walk: evict a random resident of the four slots, not the one just vacated for this feature
            uint32_t s; int c;
            for (int tries = 0;; tries++) {
                rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
                c = (int)((rs >> 33) & 1u);
                s = 2u * b[c] + (uint32_t)((rs >> 34) & 1u);
                if (s != from || tries > 8) break;
            }
            const uint32_t g = ow[s];
            tg[s] = lt_tag(cmp[c], o, ob); ow[s] = f;
            from = s; f = g;
        }
        if (!placed) return false;
    }
    return true;
}

// LDS tables for a library whose features all have the window length (f2q_device.h, "LDS tables").
inline void build_lt(HostIndex &ix, const std::vector<uint32_t> &ids, int len, int miss)
{
    memset(&ix.lt, 0, sizeof ix.lt);
    ix.lt_tags.assign(1, F2Q_LT_EMPTY); ix.lt_feat_of.assign(1, 0); ix.lt_slot_of.assign(1, 0);
    if (len < 14 || len > 21 || miss > 1 || ids.empty() || ids.size() != ix.n_features) return;
    if (ids.size() > (size_t)(F2Q_LT_SLOTS * 0.87)) return;
    LtDesc lt{};
    lt.len = (uint32_t)len; lt.hb0 = 2u * (uint32_t)(len / 2); lt.hb1 = 2u * (uint32_t)len - lt.hb0;
    std::vector<uint32_t> tags(2 * (size_t)F2Q_LT_SLOTS, F2Q_LT_EMPTY);
    std::vector<uint32_t> owner(2 * (size_t)F2Q_LT_SLOTS, ~0u);         // feature stored in a slot
    for (int t = 0; t < 2; t++)
        if (!lt_cuckoo(ix, ids, t, lt.hb0, lt.hb1, F2Q_LT_BBITS, tags.data() + (size_t)t * F2Q_LT_SLOTS, owner.data() + (size_t)t * F2Q_LT_SLOTS))
            return;                                    // no LDS tables for this library (lt.ok stays 0)
    ix.lt_slot_of.assign(ix.n_features, 0); ix.lt_feat_of.assign(F2Q_LT_SLOTS, 0);
    for (uint32_t s = 0; s < F2Q_LT_SLOTS; s++)
        if (owner[s] != ~0u) { ix.lt_slot_of[owner[s]] = (uint16_t)s; ix.lt_feat_of[s] = owner[s]; }
    ix.lt_tags.swap(tags);
    lt.ok = 1;
    ix.lt = lt;
}

// Partitioned tables (f2q_device.h: PtDesc) for a uniform library that is too large for one workgroup's LDS: the features
// are dealt into partitions by pt_part(half 0); a table 0 per partition, one table 1 over all of them.  force_parts > 0
// builds that many partitions whatever the library's size (tests, A/B runs).
inline void build_pt(HostIndex &ix, const std::vector<uint32_t> &ids, int len, int miss, int force_parts)
{
    memset(&ix.pt, 0, sizeof ix.pt);
    ix.pt_tags0.assign(1, F2Q_LT_EMPTY); ix.pt_tags1.assign(1, F2Q_LT_EMPTY); ix.pt_pstart.assign(2, 0);
    ix.pt_slot0_of.assign(1, 0); ix.pt_slot1_of.assign(1, 0); ix.pt_feat_of.assign(1, 0); ix.pt_feat0_of.assign(1, 0);
    if (len < 14 || len > 21 || miss > 1 || ids.empty() || ids.size() != ix.n_features) return;
    if (force_parts <= 0 && ix.lt.ok) return;                            // the whole library fits one workgroup's LDS
    const uint32_t n = (uint32_t)ids.size();
    PtDesc pt{};
    pt.len = (uint32_t)len; pt.hb0 = 2u * (uint32_t)(len / 2); pt.hb1 = 2u * (uint32_t)len - pt.hb0;
    // table 1: load <= 0.2, so that a bucket is seldom full and nearly every feature sits in its first-choice bucket (a
    // lookup then reads ONE bucket: see PtDesc::spill); tag = [choice][half 1 above the bucket bits][half 0]
    uint32_t bb1 = 4;
    while ((2ull << bb1) < 5ull * n) bb1++;
    if (bb1 > pt.hb1) bb1 = pt.hb1;
    if ((2ull << bb1) < (uint64_t)n + n / 4 || 1u + (pt.hb1 - bb1) + pt.hb0 > 32u) return;
    std::vector<uint32_t> tags1((size_t)2 << bb1, F2Q_LT_EMPTY), owner1((size_t)2 << bb1, ~0u);
    if (!lt_cuckoo(ix, ids, 1, pt.hb0, pt.hb1, bb1, tags1.data(), owner1.data())) return;   // > 4 features share a half 1
    uint32_t P = force_parts > 0 ? (uint32_t)force_parts : (n + F2Q_PT_FILL - 1) / F2Q_PT_FILL;
    for (; P <= F2Q_PT_MAXP; P++) {
        std::vector<std::vector<uint32_t>> by_part(P);
        for (uint32_t f : ids) by_part[pt_part((uint32_t)ix.key2[f] & ((1u << pt.hb0) - 1u), P)].push_back(f);
        std::vector<uint32_t> tags0((size_t)P * F2Q_LT_SLOTS, F2Q_LT_EMPTY), owner0((size_t)P * F2Q_LT_SLOTS, ~0u);
        bool ok = true;
        for (uint32_t p = 0; p < P && ok; p++)
            ok = by_part[p].size() <= (size_t)(F2Q_LT_SLOTS * 0.87) &&
                 lt_cuckoo(ix, by_part[p], 0, pt.hb0, pt.hb1, F2Q_LT_BBITS, tags0.data() + (size_t)p * F2Q_LT_SLOTS, owner0.data() + (size_t)p * F2Q_LT_SLOTS);
        if (!ok) { if (force_parts > 0) return; continue; }            // a crowded partition: deal into one more
        pt.n_parts = P; pt.bb1 = bb1;
        ix.pt_pstart.assign(P + 1, 0);
        ix.pt_feat_of.clear(); ix.pt_slot0_of.assign(n, 0); ix.pt_slot1_of.assign(n, 0);
        ix.pt_feat0_of.assign((size_t)P * F2Q_LT_SLOTS, 0);
        std::vector<uint32_t> gid_of(ix.n_features, 0);
        for (uint32_t p = 0; p < P; p++) {
            ix.pt_pstart[p] = (uint32_t)ix.pt_feat_of.size();
            for (uint32_t f : by_part[p]) { gid_of[f] = (uint32_t)ix.pt_feat_of.size(); ix.pt_feat_of.push_back(f); }
            pt.max_part = std::max<uint32_t>(pt.max_part, (uint32_t)by_part[p].size());
            for (uint32_t s = 0; s < F2Q_LT_SLOTS; s++) {
                const uint32_t f = owner0[(size_t)p * F2Q_LT_SLOTS + s];
                if (f != ~0u) { ix.pt_slot0_of[gid_of[f]] = (uint16_t)s; ix.pt_feat0_of[(size_t)p * F2Q_LT_SLOTS + s] = f; }
            }
        }
        ix.pt_pstart[P] = n;
        for (size_t s = 0; s < owner1.size(); s++) if (owner1[s] != ~0u) ix.pt_slot1_of[gid_of[owner1[s]]] = (uint32_t)s;
        // the mark on a first-choice bucket whose feature went to its second choice (bit 30 of the bucket's first tag,
        // free when choice bit + the half-1 bits above the bucket index + half 0 stay below it)
        if (pt.hb0 + (pt.hb1 - bb1) <= 30u) {
            pt.spill = F2Q_PT_SPILL;
            for (size_t s = 0; s < owner1.size(); s++) {
                if (owner1[s] == ~0u) continue;
                const uint32_t h1 = (uint32_t)(ix.key2[owner1[s]] >> pt.hb0);
                uint32_t b0, c0;
                lt_hash(h1, pt.hb1, pt.hb0, 0, b0, c0, bb1);
                if ((uint32_t)(s >> 1) != b0) {                      // sits in its second-choice bucket
                    uint32_t &w = tags1[2u * (size_t)b0];
                    w = w == F2Q_LT_EMPTY ? F2Q_PT_EMPTY_SPILL : (w == F2Q_PT_EMPTY_SPILL ? w : (w | F2Q_PT_SPILL));
                }
            }
        }
        ix.pt_tags0.swap(tags0); ix.pt_tags1.swap(tags1);
        pt.ok = 1;
        ix.pt = pt;
        return;
    }
}

// Pair tables (f2q_device.h: PwDesc): every feature reads A:B, all A of one length and all B of one length (<= 20 ACGT
// bases each), --m <= 1.  Built for every such library; only two-pair anchored runs in Counter mode ask them.
inline void build_pw(HostIndex &ix, int miss)
{
    memset(&ix.pw, 0, sizeof ix.pw);
    ix.pw_tab.assign(2, ~0ull);
    const uint32_t n = ix.n_features;
    if (n == 0 || miss > 1 || n >= (1u << 24)) return;
    std::vector<uint64_t> ka(n), kb(n);
    uint32_t la = 0, lb = 0;
    for (uint32_t f = 0; f < n; f++) {
        const uint8_t *s = ix.feat_bytes.data() + ix.feat_off[f];
        const uint32_t len = ix.feat_off[f + 1] - ix.feat_off[f];
        const uint8_t *c = (const uint8_t *)memchr(s, ':', len);
        if (!c) return;
        const uint32_t a = (uint32_t)(c - s), b = len - a - 1u;
        if (f == 0) { la = a; lb = b; }
        if (a != la || b != lb || a < 1 || b < 1 || a > F2Q_PW_MAXLEN || b > F2Q_PW_MAXLEN) return;
        if (!feature_key(s, a, ka[f]) || !feature_key(c + 1, b, kb[f])) return;       // (a second ':' or any other symbol)
    }
    uint32_t bits = 4;
    while ((1ull << bits) < 4ull * n) bits++;
    const uint32_t mask = (1u << bits) - 1u;
    ix.pw_tab.assign((size_t)6 << bits, ~0ull);
    auto put = [&](int t, uint32_t s0, uint32_t f) {
        uint64_t *tab = ix.pw_tab.data() + ((size_t)(2 * t) << bits);
        uint32_t s = s0;
        while (tab[2u * s + 1u] != ~0ull) s = (s + 1u) & mask;
        tab[2u * s] = ka[f] | ((uint64_t)f << 40); tab[2u * s + 1u] = kb[f];
    };
    for (uint32_t f = 0; f < n; f++) {
        put(0, pw_hash(ka[f], kb[f], bits), f);
        put(1, pw_hash(ka[f], 0ull, bits), f);
        put(2, pw_hash(kb[f], 1ull, bits), f);
    }
    ix.pw.ok = 1; ix.pw.la = la; ix.pw.lb = lb; ix.pw.bits = bits;
}

// byte-string index of ALL features, by length (f2q_device.h: GkDesc): exact table + m+1 pigeonhole piece tables per group
inline void build_gk(HostIndex &ix, int miss)
{
    ix.gk_groups.clear(); ix.gk_tab.assign(1, 0u); ix.gk_ids.clear();
    std::vector<uint32_t> order(ix.n_features);
    for (uint32_t f = 0; f < ix.n_features; f++) order[f] = f;
    auto flen = [&](uint32_t f) { return ix.feat_off[f + 1] - ix.feat_off[f]; };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return flen(a) < flen(b); });
    ix.gk_ids = order;
    if (ix.gk_ids.empty()) ix.gk_ids.push_back(0);
    for (size_t i = 0; i < order.size();) {
        size_t j = i;
        while (j < order.size() && flen(order[j]) == flen(order[i])) j++;
        GkGroup g; memset(&g, 0, sizeof g);
        g.len = flen(order[i]); g.n = (uint32_t)(j - i); g.ids_off = (uint32_t)i;
        g.bits = 4; while ((1ull << g.bits) < 2ull * g.n) g.bits++;
        const int P = miss > 0 ? miss + 1 : 0;
        g.n_pieces = (P >= 1 && P <= F2Q_GK_MAXP && (uint32_t)P <= g.len) ? (uint32_t)P : 0u;
        g.exact_off = (uint32_t)ix.gk_tab.size();
        ix.gk_tab.resize(ix.gk_tab.size() + ((size_t)1 << g.bits), 0u);
        for (uint32_t p = 0; p < g.n_pieces; p++) {
            g.cut[p] = (uint32_t)((uint64_t)p * g.len / g.n_pieces);
            g.piece_off[p] = (uint32_t)ix.gk_tab.size();
            ix.gk_tab.resize(ix.gk_tab.size() + ((size_t)1 << g.bits), 0u);
        }
        g.cut[g.n_pieces] = g.len;
        const uint32_t m = (1u << g.bits) - 1u;
        for (size_t q = i; q < j; q++) {
            const uint32_t f = order[q];
            const uint8_t *fb = ix.feat_bytes.data() + ix.feat_off[f];
            uint32_t s = (uint32_t)(gk_hash_bytes(fb, 0, (int)g.len, 0xE0u) >> (64u - g.bits));
            while (ix.gk_tab[g.exact_off + s]) s = (s + 1u) & m;
            ix.gk_tab[g.exact_off + s] = f + 1u;
            for (uint32_t p = 0; p < g.n_pieces; p++) {
                s = (uint32_t)(gk_hash_bytes(fb, (int)g.cut[p], (int)g.cut[p + 1], p) >> (64u - g.bits));
                while (ix.gk_tab[g.piece_off[p] + s]) s = (s + 1u) & m;
                ix.gk_tab[g.piece_off[p] + s] = f + 1u;
            }
        }
        ix.gk_groups.push_back(g);
        i = j;
    }
    if (ix.gk_groups.empty()) { GkGroup g; memset(&g, 0, sizeof g); g.len = 0xFFFFFFFFu; ix.gk_groups.push_back(g); }
    // the features once more as zero-padded 8-byte words (GkDesc::fw)
    ix.gk_fw.clear(); ix.gk_fwoff.assign(ix.n_features + 1, 0u);
    for (uint32_t f = 0; f < ix.n_features; f++) {
        const uint32_t len = ix.feat_off[f + 1] - ix.feat_off[f];
        ix.gk_fwoff[f] = (uint32_t)ix.gk_fw.size();
        const uint8_t *fb = ix.feat_bytes.data() + ix.feat_off[f];
        for (uint32_t w = 0; 8 * w < len; w++) {
            unsigned long long v = 0;
            for (uint32_t b = 0; b < 8 && 8 * w + b < len; b++) v |= (unsigned long long)fb[8 * w + b] << (8 * b);
            ix.gk_fw.push_back(v);
        }
    }
    ix.gk_fwoff[ix.n_features] = (uint32_t)ix.gk_fw.size();
    for (int pad = 0; pad < F2Q_GK_MAXW; pad++) ix.gk_fw.push_back(0ull);     // (the word loop of a short last feature may look past it)
}

// packed_len: the feature length the packed tables index (the window length of a fixed-offset run); mw_windows >= 2:
// a multi-window run of that many windows, whose k-part features get packed tables of their own
inline void build_index(HostIndex &ix, const char *seqs, const uint32_t *offs, uint32_t n, int miss, int packed_len = 0, int mw_windows = 0)
{
    { const int keep = ix.pt_force_parts; ix = HostIndex(); ix.pt_force_parts = keep; }
    ix.n_features = n;
    ix.feat_off.assign(offs, offs + n + 1);
    uint32_t base = offs[0];
    for (auto &o : ix.feat_off) o -= base;
    ix.feat_bytes.assign((const uint8_t *)seqs + base, (const uint8_t *)seqs + offs[n]);
    ix.feat_bytes.resize(ix.feat_bytes.size() + 8, 0);
    ix.key2.assign(n, 0);
    memset(ix.grp, 0, sizeof ix.grp);
    std::vector<std::vector<uint32_t>> by_len(F2Q_REG_MAXLEN + 1);
    std::vector<std::vector<uint32_t>> by_parts(F2Q_MW_MAX + 1);          // multi-window runs: k-part features, k >= 2
    const bool multi = mw_windows >= 2 && mw_windows <= F2Q_MW_MAX && packed_len >= 1;
    bool mw_ok = multi;
    for (uint32_t f = 0; f < n; f++) {
        uint32_t len = ix.feat_off[f + 1] - ix.feat_off[f];
        uint64_t k;
        int parts;
        if (feature_key(ix.feat_bytes.data() + ix.feat_off[f], len, k)) {
            ix.key2[f] = k; by_len[len].push_back(f);
            // a plain feature as long as a joined key of k >= 2 parts could be within --m of such a key (the ':' being
            // one of the mismatches): the packed multi-window tables would not see it
            if (multi) for (int q = 2; q <= mw_windows; q++) if ((int)len == q * packed_len + q - 1) mw_ok = false;
        } else if (multi && (parts = parts_key(ix.feat_bytes.data() + ix.feat_off[f], len, packed_len, mw_windows, k)) != 0) {
            ix.key2[f] = k; by_parts[parts].push_back(f);
        } else ix.irr_ids.push_back(f);
    }
    uint32_t off = 0;
    for (int L = 1; L <= F2Q_REG_MAXLEN; L++) {
        LenGroup &g = ix.grp[L];
        g.n = (uint32_t)by_len[L].size();
        if (!g.n) continue;
        const uint32_t bits = table_bits(g.n);
        g.exact.off = off; g.exact.bits = bits; g.exact.shift = 0; g.exact.mask = ~0ull;
        off += 1u << bits;
        // pigeonhole pieces: miss+1 contiguous base ranges; when that many do not fit (or exceed the
        // descriptor), one zero-width piece puts every feature in one chain (exhaustive scan)
        int P = miss > 0 ? miss + 1 : 0;
        if (P > L || P > F2Q_MAX_PIECES) P = 1, g.n_pieces = 1;
        else g.n_pieces = (uint32_t)P;
        for (uint32_t p = 0; p < g.n_pieces; p++) {
            PieceDesc &pd = g.piece[p];
            pd.off = off; pd.bits = bits; off += 1u << bits;
            if (miss + 1 > L || miss + 1 > F2Q_MAX_PIECES) { pd.shift = 0; pd.mask = 0; }
            else {
                int b0 = (int)((long)p * L / P), b1 = (int)((long)(p + 1) * L / P);
                pd.shift = (uint32_t)(2 * b0);
                pd.mask = (b1 - b0 >= 32) ? ~0ull : ((1ull << (2 * (b1 - b0))) - 1ull);
            }
        }
    }
    ix.tab_keys.assign(off ? off : 1, KEY_EMPTY);
    ix.tab_idx.assign(off ? off : 1, 0);
    for (int L = 1; L <= F2Q_REG_MAXLEN; L++) {
        LenGroup &g = ix.grp[L];
        for (uint32_t f : by_len[L]) {
            uint64_t k = ix.key2[f];
            table_insert(ix.tab_keys, ix.tab_idx, g.exact, k, k, f);
            for (uint32_t p = 0; p < g.n_pieces; p++)
                table_insert(ix.tab_keys, ix.tab_idx, g.piece[p], (k >> g.piece[p].shift) & g.piece[p].mask, k, f);
        }
    }
    ix.ptab.clear();
    build_packed(ix, (packed_len >= 1 && packed_len <= F2Q_REG_MAXLEN) ? by_len[packed_len] : std::vector<uint32_t>(),
                 packed_len, miss, ix.pk);
    for (auto &g : ix.mpk) memset(&g, 0, sizeof g);
    ix.mw_ok = 0;
    if (multi) {
        ix.mpk[0] = ix.pk;
        for (int q = 2; q <= mw_windows; q++) {
            build_packed(ix, by_parts[q], q * packed_len, miss, ix.mpk[q - 1]);
            if (!by_parts[q].empty() && ix.mpk[q - 1].len == 0) mw_ok = false;      // tables could not be built
        }
        if (!by_len[packed_len].empty() && ix.pk.len == 0) mw_ok = false;
        ix.mw_ok = mw_ok ? 1u : 0u;
    }
    {
        // a multi-window run whose features ALL have as many parts as the run has windows: the LDS tables index the joined
        // keys (the compact window of the tiles is mw_windows * packed_len bases long)
        bool mw_lt = multi && mw_ok && by_len[packed_len].empty() && by_parts[mw_windows].size() == (size_t)n;
        if (mw_lt) {
            build_lt(ix, by_parts[mw_windows], mw_windows * packed_len, miss);
            if (!ix.lt.ok && mw_windows == 2) {
                // two windows and features that share whole windows (a combinatorial pair library): the tables are built
                // on, and asked with, the mixed form of the joined keys (mw_mix: a dozen instructions per read more)
                std::vector<uint64_t> plain = ix.key2;
                for (uint32_t f : by_parts[2]) ix.key2[f] = mw_mix(plain[f], (uint32_t)packed_len);
                build_lt(ix, by_parts[2], 2 * packed_len, miss);
                ix.key2.swap(plain);
                ix.lt.mix = (uint32_t)packed_len;
            }
        }
        else build_lt(ix, (packed_len >= 1 && packed_len <= F2Q_REG_MAXLEN && !multi) ? by_len[packed_len] : std::vector<uint32_t>(), packed_len, miss);
    }
    build_pt(ix, (packed_len >= 1 && packed_len <= F2Q_REG_MAXLEN) ? by_len[packed_len] : std::vector<uint32_t>(), packed_len, miss, ix.pt_force_parts);
    build_gk(ix, miss);
    build_pw(ix, miss);
    ix.n_irregular = (uint32_t)ix.irr_ids.size();
    if (ix.irr_ids.empty()) ix.irr_ids.push_back(0);     // keep the device array non-empty
}

// ---------------------------------------------------------------------------------------------
// FASTQ framing (fastq_parser :324-328): '\n'-separated lines, rstrip(), 4 lines per record
// ---------------------------------------------------------------------------------------------
typedef RecT<const uint8_t *> Rec;

inline uint32_t rstrip_len(const uint8_t *p, size_t n)
{
    while (n > 0) {
        uint8_t c = p[n - 1];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c) n--; else break;
    }
    return (uint32_t)n;
}

// frames complete records of buf; returns bytes consumed up to the end of the last complete record
inline size_t frame_fastq(const uint8_t *buf, size_t nbytes, std::vector<Rec> &out)
{
    size_t pos = 0, consumed = 0;
    const uint8_t *ln[4]; uint32_t ll[4]; int k = 0;
    while (pos < nbytes) {
        const uint8_t *nl = (const uint8_t *)memchr(buf + pos, '\n', nbytes - pos);
        size_t eol = nl ? (size_t)(nl - buf) : nbytes;
        ln[k] = buf + pos; ll[k] = rstrip_len(buf + pos, eol - pos); k++;
        pos = nl ? eol + 1 : nbytes;
        if (k == 4) { out.push_back(Rec{ln[1], ln[3], ll[1], ll[3]}); k = 0; consumed = pos; }
    }
    return consumed;
}

// ---------------------------------------------------------------------------------------------
// packing plan (PackPlan, read_is_clean, pack_read live in f2q_device.h: the device packer shares them)
// ---------------------------------------------------------------------------------------------
inline PackPlan make_plan(const RunDev &run)
{
    PackPlan pl;
    // Counter mode: windows up to 31 bases (2-bit library tables); Extract+Count: up to 29 (single-word key table)
    pl.fast_fixed = run.fixed && run.n_iter == 1 && run.start[0] >= 0 && run.length >= 0 &&
                    run.length <= (run.mode == 0 ? F2Q_REG_MAXLEN : F2Q_EC64_MAXLEN) &&
                    run.start[0] + run.length <= F2Q_PACK_MAXLEN;
    pl.need = pl.fast_fixed ? run.start[0] + run.length : 0;
    pl.from = pl.fast_fixed ? run.start[0] : 0;
    // several windows (--st a,b,...): Counter mode, every window inside the packed range, all parts in one 2-bit key
    if (run.fixed && run.mode == 0 && run.n_iter >= 2 && run.n_iter <= F2Q_MW_MAX && run.length >= 1 &&
        run.n_iter * run.length <= F2Q_REG_MAXLEN) {
        int lo = run.start[0], hi = run.start[0];
        for (int i = 1; i < run.n_iter; i++) { lo = run.start[i] < lo ? run.start[i] : lo; hi = run.start[i] > hi ? run.start[i] : hi; }
        if (lo >= 0 && hi + run.length <= F2Q_PACK_MAXLEN) {
            pl.fast_fixed = true; pl.multi = true;
            pl.n_win = run.n_iter; pl.win_len = run.length; pl.win_end = hi + run.length;       // only the windows are stored, back to back
            for (int i = 0; i < run.n_iter; i++) pl.win_start[i] = run.start[i];
            pl.from = 0; pl.need = run.n_iter * run.length;
        }
    }
    pl.fast_anchor = !run.fixed && run.n_iter == 1 && run.anchors_packed && run.msu >= 0 && run.msd >= 0 &&
                     run.msu <= 7 && run.msd <= 7 && run.length >= 0 && run.length <= F2Q_ANCHOR_MAXLEN;
    // several pairs: every pair searched on the same planes, the parts joined with ':' (fast2q.py:333-363)
    if (!run.fixed && run.n_iter >= 2 && run.pairs_packed && run.msu >= 0 && run.msd >= 0 && run.msu <= 7 && run.msd <= 7 &&
        run.length >= 0 && run.length <= F2Q_ANCHOR_MAXLEN) { pl.fast_anchor = true; pl.multi_pair = true; }
    pl.kb = (run.msu == 0 && run.msd == 0) ? 0 : (run.msu <= 1 && run.msd <= 1) ? 1 : 3;
    // anchored Extract+Count: a read with 'N's keeps to the packed path (an 'N' equals no anchor base; a window that
    // holds one is spelt out from the planes and the flag bits), any other odd symbol sends the read to the byte-exact path
    if (run.mode == 1 && pl.fast_anchor) { pl.inband_n = true; pl.n_only = true; }
    // Extract+Count with a fixed window: the same for windows whose 'N's fit the single-word key (read_is_clean asks ec64_fits)
    if (run.mode == 1 && pl.fast_fixed && !pl.multi) { pl.inband_n = true; pl.n_only = true; }
    return pl;
}

struct HostPacked {
    uint32_t n_tiles = 0, wb = 0, wq = 0, rmax = 0;
    uint32_t planar_nw = 0;                    // anchored runs: bases stored as bit-planes of this many 32-base words
    uint64_t n_clean = 0;
    std::vector<uint32_t> bases, qual;
    std::vector<uint16_t> len;
    std::vector<uint32_t> c_index;             // per packed slot: position of the read inside the block
    // general-path records (raw bytes)
    std::vector<uint8_t> raw;                  // seq bytes then qual bytes per record
    std::vector<unsigned long long> g_off;     // per record: offset of seq in raw (qual follows at +len)
    std::vector<uint32_t> g_len, g_qlen, g_index;
};

struct HostSink {
    uint32_t *bp, *qp; uint16_t *lp;
    void base(uint32_t w, uint32_t v) { bp[(size_t)w * F2Q_TILE] = v; }
    void qual(uint32_t w, uint32_t v) { qp[(size_t)w * F2Q_TILE] = v; }
    void len(uint32_t v) { *lp = (uint16_t)v; }
};

inline void pack_records(const PackPlan &pl, const std::vector<Rec> &recs, HostPacked &hp)
{
    hp = HostPacked();
    std::vector<uint32_t> clean; clean.reserve(recs.size());
    uint32_t rmax = 0;
    for (uint32_t i = 0; i < recs.size(); i++) {
        const Rec &r = recs[i];
        if (read_is_clean(pl, r)) {
            clean.push_back(i);
            uint32_t l = packed_len(pl, r);
            if (l > rmax) rmax = l;
        } else {
            hp.g_off.push_back(hp.raw.size());
            hp.g_len.push_back(r.len); hp.g_qlen.push_back(r.qlen); hp.g_index.push_back(i);
            hp.raw.insert(hp.raw.end(), r.seq, r.seq + r.len);
            hp.raw.insert(hp.raw.end(), r.qual, r.qual + r.qlen);
        }
    }
    hp.raw.resize(hp.raw.size() + 8, 0);
    hp.n_clean = clean.size();
    if (clean.empty()) return;
    hp.c_index.assign(((clean.size() + F2Q_TILE - 1) / F2Q_TILE) * F2Q_TILE, 0);
    for (size_t s = 0; s < clean.size(); s++) hp.c_index[s] = clean[s];
    tile_geometry(pl, rmax, hp.rmax, hp.planar_nw, hp.wb, hp.wq);
    hp.n_tiles = (uint32_t)((clean.size() + F2Q_TILE - 1) / F2Q_TILE);
    hp.bases.assign((size_t)hp.n_tiles * hp.wb * F2Q_TILE, 0);
    hp.qual.assign((size_t)hp.n_tiles * hp.wq * F2Q_TILE, 0);
    hp.len.assign((size_t)hp.n_tiles * F2Q_TILE, (uint16_t)F2Q_LEN_SKIP);
    for (size_t s = 0; s < clean.size(); s++) {
        const size_t tile = s / F2Q_TILE, lane = s % F2Q_TILE;
        HostSink sink{hp.bases.data() + tile * hp.wb * F2Q_TILE + lane, hp.qual.data() + tile * hp.wq * F2Q_TILE + lane,
                      hp.len.data() + tile * F2Q_TILE + lane};
        pack_read(pl, recs[clean[s]], hp.planar_nw, sink);
    }
}

// ---------------------------------------------------------------------------------------------
// run set-up (reads_counter :536-558, initializer :1112-1129)
// ---------------------------------------------------------------------------------------------
inline int phred_threshold(int ph)
{
    if (ph <= 0) ph = 1;                       // :1118-1125
    int thr = ph + 31;                         // fail set = chr(33) .. chr(33 + ph - 2)
    return thr > 126 ? 126 : thr;              // quality_list stops at '~' (:1114)
}

// f2q_params -> RunDev; returns an F2Q_E* code and a message
inline int fill_run(const f2q_params &p, RunDev &r, std::string &err)
{
    memset(&r, 0, sizeof r);
    r.mode = p.mode; r.miss = p.miss < 0 ? 0 : p.miss; r.length = p.length;
    r.thr = phred_threshold(p.phred); r.thr_up = phred_threshold(p.qual_up); r.thr_down = phred_threshold(p.qual_down);
    r.msu = p.miss_search_up; r.msd = p.miss_search_down;
    if (p.n_upstream == 0 && p.n_downstream == 0) {               // fast2q.py:538-541
        if (p.n_start < 1 || p.n_start > F2Q_MAX_ITER) { err = "n_start must be 1..16"; return F2Q_EINVAL; }
        r.fixed = 1; r.n_iter = p.n_start;
        for (int i = 0; i < p.n_start; i++) r.start[i] = p.start[i];
        r.compact = make_plan(r).multi ? 1 : 0;                   // the packed tiles of a multi-window run hold the windows only
        return F2Q_OK;
    }
    // :543-558
    if (p.n_upstream < 0 || p.n_upstream > F2Q_MAX_ITER || p.n_downstream < 0 || p.n_downstream > F2Q_MAX_ITER) {
        err = "at most 16 upstream/downstream search sequences"; return F2Q_EINVAL;
    }
    if (p.n_upstream && p.n_downstream && p.n_upstream != p.n_downstream) {
        err = "Up and Downstream sequences must be submitted in concurrent pairs"; return F2Q_EINVAL;   // :553-556
    }
    r.fixed = 0; r.has_up = p.n_upstream > 0; r.has_down = p.n_downstream > 0;
    r.n_iter = p.n_upstream > p.n_downstream ? p.n_upstream : p.n_downstream;
    for (int side = 0; side < 2; side++) {
        const int n = side ? p.n_downstream : p.n_upstream;
        for (int i = 0; i < n; i++) {
            const char *s = side ? p.downstream[i] : p.upstream[i];
            if (!s) { err = "null search sequence"; return F2Q_EINVAL; }
            size_t l = strlen(s);
            if (l > F2Q_ANCHOR_MAX) { err = "search sequence longer than 128"; return F2Q_EUNSUPPORTED; }
            (side ? r.down_len : r.up_len)[i] = (int)l;
            for (size_t k = 0; k < l; k++) (side ? r.down : r.up)[i][k] = up8((uint8_t)s[k]);      // :547,:550
        }
    }
    // one pair of ACGT-only anchors of 1..32 symbols: eligible for the packed bit-plane search
    r.anchors_packed = (r.n_iter == 1);
    for (int side = 0; side < 2 && r.anchors_packed; side++) {
        const int has = side ? r.has_down : r.has_up;
        if (!has) continue;
        const int l = side ? r.down_len[0] : r.up_len[0];
        if (l < 1 || l > 32) { r.anchors_packed = 0; break; }
        for (int k = 0; k < l; k++) {
            uint32_t c = base_code(side ? r.down[0][k] : r.up[0][k]);
            if (c > 3u) { r.anchors_packed = 0; break; }
            (side ? r.down_codes : r.up_codes) |= (uint64_t)c << (2 * k);
            (side ? r.down_pos : r.up_pos)[c] |= 1u << k;
        }
    }
    // several pairs: the same per-symbol position masks for every pair (all anchors ACGT-only, 1..32 symbols)
    r.pairs_packed = (r.n_iter >= 2);
    for (int i = 0; i < r.n_iter && r.pairs_packed; i++)
        for (int side = 0; side < 2 && r.pairs_packed; side++) {
            if (!(side ? r.has_down : r.has_up)) continue;
            const int l = side ? r.down_len[i] : r.up_len[i];
            if (l < 1 || l > 32) { r.pairs_packed = 0; break; }
            for (int k = 0; k < l; k++) {
                const uint32_t c = base_code(side ? r.down[i][k] : r.up[i][k]);
                if (c > 3u) { r.pairs_packed = 0; break; }
                (side ? r.mp_down_pos : r.mp_up_pos)[i][c] |= 1u << k;
            }
        }
    return F2Q_OK;
}

} // namespace f2q
