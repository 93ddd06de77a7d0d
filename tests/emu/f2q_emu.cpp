// tests/emu/f2q_emu.cpp -- TEST INFRASTRUCTURE.  Compiles the product's per-lane logic
// (2fast2q_amd/csrc/f2q_device.h, f2q_host.h, f2q_synth.h) with g++ and runs it lane by lane on
// the host, so that the logic the HIP kernels execute can be checked against the oracle on a
// machine without a GPU.  The product never uses this file; libf2q_hip.so has no host path.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../2fast2q_amd/csrc/f2q_device.h"
#include "../../2fast2q_amd/csrc/f2q_host.h"
#include "../../2fast2q_amd/csrc/f2q_synth.h"
#include "../../2fast2q_amd/csrc/f2q_reader.h"

using namespace f2q;

struct Emu {
    RunDev run; PackPlan plan; HostIndex ix; LibDev lib;
    std::vector<unsigned long long> acc;      // counts + 5 stats
    EcDev ec; std::vector<unsigned long long> slots, ent_off, ent_count, ent_first, ctr, k64s, k64c, k64f; std::vector<uint32_t> ent_len, arena;
    uint64_t reads_seen = 0, fast = 0, general = 0, v2_reads = 0, anchor_reads = 0;
    int use_v2 = 1, use_lt = 1, use_pw = 1;
    uint64_t lt_reads = 0, pt_reads = 0;
    std::string err;
};

static void bind_lib(Emu *e)
{
    LibDev &L = e->lib;
    L.n_features = e->ix.n_features; L.n_irregular = e->ix.n_irregular;
    L.tab_keys = e->ix.tab_keys.data(); L.tab_idx = e->ix.tab_idx.data();
    L.ptab = e->ix.ptab.data(); L.pk = e->ix.pk;
    memcpy(L.mpk, e->ix.mpk, sizeof L.mpk); L.mw_ok = e->ix.mw_ok;
    L.feat_bytes = e->ix.feat_bytes.data(); L.feat_off = e->ix.feat_off.data(); L.irr_ids = e->ix.irr_ids.data();
    L.lt = e->ix.lt; L.lt.tags = e->ix.lt_tags.data(); L.lt.slot_of = e->ix.lt_slot_of.data();
    L.pw = e->ix.pw; L.pw.tab = e->ix.pw_tab.data(); if (!e->use_pw) L.pw.ok = 0;
    L.lt.feat_of = e->ix.lt_feat_of.data();
    L.pt = e->ix.pt; L.pt.tags0 = e->ix.pt_tags0.data(); L.pt.tags1 = e->ix.pt_tags1.data(); L.pt.pstart = e->ix.pt_pstart.data();
    L.pt.slot0_of = e->ix.pt_slot0_of.data(); L.pt.slot1_of = e->ix.pt_slot1_of.data(); L.pt.feat_of = e->ix.pt_feat_of.data();
    L.pt.feat0_of = e->ix.pt_feat0_of.data();
    L.gk.n_groups = e->ix.n_features ? (uint32_t)e->ix.gk_groups.size() : 0u; L.gk.grp = e->ix.gk_groups.data();
    L.gk.tab = e->ix.gk_tab.data(); L.gk.ids = e->ix.gk_ids.data();
    L.gk.fw = e->ix.gk_fw.data(); L.gk.fwoff = e->ix.gk_fwoff.data();
    memcpy(L.grp, e->ix.grp, sizeof L.grp);
    e->acc.assign(e->ix.n_features + 5, 0);
}

extern "C" {

void *emu_create(const f2q_params *p)
{
    Emu *e = new Emu();
    if (fill_run(*p, e->run, e->err)) { delete e; return nullptr; }
    e->plan = make_plan(e->run);
    uint32_t z = 0;
    build_index(e->ix, "", &z, 0, e->run.miss, 0);
    bind_lib(e);
    const size_t cap = 1 << 16, slots = 1 << 18;
    e->slots.assign(slots, 0); e->ent_off.assign(cap, 0); e->ent_len.assign(cap, 0); e->ent_count.assign(cap, 0);
    e->ent_first.assign(cap, ~0ull); e->arena.assign(1 << 20, 0); e->ctr.assign(4, 0);
    e->ec.slots = e->slots.data(); e->ec.mask = slots - 1; e->ec.max_entries = cap; e->ec.ent_off = e->ent_off.data();
    e->ec.ent_len = e->ent_len.data(); e->ec.ent_count = e->ent_count.data(); e->ec.ent_first = e->ent_first.data();
    e->ec.arena = e->arena.data(); e->ec.arena_words = e->arena.size(); e->ec.ctr = e->ctr.data();
    e->k64s.assign(slots, ~0ull); e->k64c.assign(slots, 0); e->k64f.assign(slots, ~0ull);
    e->ec.k64_slots = e->k64s.data(); e->ec.k64_count = e->k64c.data(); e->ec.k64_first = e->k64f.data(); e->ec.k64_mask = slots - 1;
    return e;
}
void emu_destroy(void *h) { delete (Emu *)h; }

void emu_set_features(void *h, const char *seqs, const uint32_t *offs, uint32_t n)
{
    Emu *e = (Emu *)h;
    e->plan = make_plan(e->run);
    int packed_len = e->plan.fast_fixed ? e->run.length : 0;
    if (e->plan.fast_anchor) {
        if (e->run.has_up && e->run.has_down) {
            std::vector<uint32_t> hist(F2Q_REG_MAXLEN + 1, 0);
            for (uint32_t i = 0; i < n; i++) { uint32_t l = offs[i + 1] - offs[i]; if (l >= 1 && l <= F2Q_REG_MAXLEN) hist[l]++; }
            packed_len = (int)(std::max_element(hist.begin(), hist.end()) - hist.begin());
        } else packed_len = e->run.length;
    }
    build_index(e->ix, seqs, offs, n, e->run.miss, packed_len, e->plan.multi ? e->run.n_iter : 0);
    if (e->plan.multi && (!e->ix.mw_ok || e->ix.n_irregular)) { e->plan.multi = false; e->plan.fast_fixed = false; }
    bind_lib(e);
    e->plan.inband_n = (e->plan.fast_fixed || e->plan.fast_anchor) && e->ix.n_irregular == 0;
    if (e->plan.fast_anchor && e->plan.multi_pair && e->run.mode == 0 && e->run.n_iter == 2 && e->lib.pw.ok) { e->plan.inband_n = true; e->plan.n_only = true; }
    if (e->ix.n_irregular && !e->plan.multi_pair) e->plan.fast_anchor = false;
}

// the same two-stream split the library does: packed tiles through fixed_lane, the rest through general_read
size_t emu_count_block(void *h, const uint8_t *buf, size_t n)
{
    Emu *e = (Emu *)h;
    std::vector<Rec> recs;
    size_t used = frame_fastq(buf, n, recs);
    HostPacked hp;
    pack_records(e->plan, recs, hp);
    Accum acc{e->acc.data(), e->acc.data() + e->ix.n_features, nullptr, nullptr, nullptr, nullptr};
    PackedBlock pb{};
    pb.n_tiles = hp.n_tiles; pb.wb = hp.wb; pb.wq = hp.wq; pb.rmax = hp.rmax; pb.n_slots = (uint64_t)hp.n_tiles * F2Q_TILE;
    pb.bases = hp.bases.data(); pb.qual = hp.qual.data(); pb.len = hp.len.data(); pb.planar_nw = hp.planar_nw;
    const std::vector<uint32_t> &hp_index = hp.c_index;
    if (hp.planar_nw) {
        // the anchored kernel's per-lane sequence (k_count_anchor), NW = 3, 5 or 10, KB = plan.kb
        auto lane_fn = [&](auto nwc, auto kbc, uint32_t t, uint32_t lane) {
            constexpr int NW = decltype(nwc)::value, KB = decltype(kbc)::value, NQW = 8 * NW;
            const uint32_t l = pb.len[(uint64_t)t * F2Q_TILE + lane];
            if (l == F2Q_LEN_SKIP) return;
            uint32_t LO[NW], HI[NW], Q[NQW];
            const uint32_t *bp = pb.bases + (uint64_t)t * pb.wb * F2Q_TILE + lane;
            const uint32_t *qp = pb.qual + (uint64_t)t * pb.wq * F2Q_TILE + lane;
            for (int w = 0; w < NW; w++) { LO[w] = bp[(uint64_t)w * F2Q_TILE]; HI[w] = bp[(uint64_t)(NW + w) * F2Q_TILE]; }
            for (int i = 0; i < NQW; i++) Q[i] = qp[(uint64_t)((uint32_t)i < pb.wq ? i : pb.wq - 1) * F2Q_TILE];
            const int r = (int)(l & F2Q_LEN_MASK);
            const bool flagged = (l & F2Q_LEN_FLAG) != 0;
            const bool lower = (l & F2Q_LEN_CASE) != 0, keyflag = flagged && !lower;   // lower-case bases: marks for the anchors only
            const unsigned long long gi = e->reads_seen + hp_index[(uint64_t)t * F2Q_TILE + lane];
            uint32_t FW[NW], FU[NW], FD[NW], FLG[NW];
            for (int cw = 0; cw < NW; cw++) {
                uint32_t q8[8];
                for (int i = 0; i < 8; i++) q8[i] = Q[8 * cw + i];
                FW[cw] = fail_word8(q8, phred_add_hi(e->run.thr));
                FU[cw] = fail_word8(q8, phred_add_hi(e->run.thr_up));
                FD[cw] = fail_word8(q8, phred_add_hi(e->run.thr_down));
                FLG[cw] = flagged ? flag_word8(q8) : 0u;
            }
            if (e->plan.multi_pair) {
                // k_count_anchor_pairs: every pair on the same planes, the joined key matched as a string
                uint32_t idx = 0;
                uint8_t kb[F2Q_PAIRS_KEYMAX];
                const int res = pairs_lane<NW, KB>(e->run, e->lib, e->ec, kb, LO, HI, FLG, r, FU, FD, FW, gi, idx, nullptr, !lower);
                if (res < 0) {
                    uint8_t sq[F2Q_ANCHOR_MAXLEN], ql[F2Q_ANCHOR_MAXLEN];
                    for (int i = 0; i < r; i++) {
                        sq[i] = (uint8_t)"ACGT"[((LO[i >> 5] >> (i & 31)) & 1u) | (((HI[i >> 5] >> (i & 31)) & 1u) << 1)];
                        ql[i] = (uint8_t)((Q[planar_qword((uint32_t)i)] >> (8 * planar_qbyte((uint32_t)i))) & 0xFFu);
                        if (ql[i] & 0x80u) { sq[i] = lower ? (uint8_t)(sq[i] | 0x20u) : (uint8_t)'N'; ql[i] &= 0x7Fu; }
                    }
                    general_read<const uint8_t *>(e->run, e->lib, e->ec, acc, sq, r, ql, r, gi, acc.stats);
                    return;
                }
                acc.stats[0]++;
                if (res == 0) acc.stats[1]++;
                else { acc.stats[res]++; if (res == 1 || res == 2) acc.counts[idx]++; }
                return;
            }
            const AnchorWin aw = anchor_window<NW, KB, KB>(e->run, LO, HI, FLG, r, FU, FD, FW);
            const int L = aw.end - aw.start;
            const bool ecm = e->run.mode == 1;
            if (aw.ok == 0) { acc.stats[4]++; acc.stats[0]++; }
            else if (aw.ok == 1 && !ecm && (L < 1 || L > F2Q_REG_MAXLEN)) { acc.stats[3]++; acc.stats[0]++; }
            else if (aw.ok == 1 && ecm && (L > F2Q_EC64_MAXLEN || (keyflag && L > 0 && !ec64_fits(plane_extract<NW>(FLG, aw.start, L), L)))) {
                uint8_t kb[32 * NW];
                for (int cw = 0; cw < NW; cw++) {
                    const int off = 32 * cw, n = L - off < 32 ? L - off : 32;
                    if (n > 0) {
                        const uint32_t lo = plane_extract<NW>(LO, aw.start + off, n), hi = plane_extract<NW>(HI, aw.start + off, n);
                        const uint32_t fl = keyflag ? plane_extract<NW>(FLG, aw.start + off, n) : 0u;
                        for (int j = 0; j < n; j++) kb[off + j] = ((fl >> j) & 1u) ? (uint8_t)'N' : (uint8_t)"ACGT"[((lo >> j) & 1u) | (((hi >> j) & 1u) << 1)];
                    }
                }
                KeyView kv; kv.seq = kb; kv.nseg = 1; kv.a[0] = 0; kv.b[0] = L; kv.len = L;
                ec_insert(e->ec, kv, gi);
                acc.stats[1]++; acc.stats[0]++;
            }
            else if (aw.ok == 2) {
                uint8_t sq[F2Q_ANCHOR_MAXLEN], ql[F2Q_ANCHOR_MAXLEN];
                for (int i = 0; i < r; i++) {
                    sq[i] = (uint8_t)"ACGT"[((LO[i >> 5] >> (i & 31)) & 1u) | (((HI[i >> 5] >> (i & 31)) & 1u) << 1)];
                    ql[i] = (uint8_t)((Q[planar_qword((uint32_t)i)] >> (8 * planar_qbyte((uint32_t)i))) & 0xFFu);
                    if (ql[i] & 0x80u) { sq[i] = lower ? (uint8_t)(sq[i] | 0x20u) : (uint8_t)'N'; ql[i] &= 0x7Fu; }
                }
                general_read<const uint8_t *>(e->run, e->lib, e->ec, acc, sq, r, ql, r, gi, acc.stats);
            } else if (keyflag && plane_extract<NW>(FLG, aw.start, L) != 0u) {
                acc.stats[0]++;
                const uint32_t forced = plane_extract<NW>(FLG, aw.start, L);
                if (ecm) {
                    unsigned long long w = 0;
                    ec64_word(plane_key<NW>(LO, HI, aw.start, L), forced, L, w);
                    if (ec64_insert_word(e->ec, w, gi)) e->ec.ctr[3]++;
                    acc.stats[1]++;
                } else if (e->run.miss == 0 || __builtin_popcount(forced) > e->run.miss) acc.stats[3]++;
                else {
                    const uint64_t key = plane_key<NW>(LO, HI, aw.start, L);
                    MinTrack tt; tt.init(e->run.miss);
                    lib_near(e->lib, key, L, spread32(forced), tt);
                    if (tt.cnt == 1) { acc.counts[tt.idx]++; acc.stats[2]++; } else acc.stats[3]++;
                }
            } else {
                const uint64_t key = plane_key<NW>(LO, HI, aw.start, L);
                acc.stats[0]++;
                if (ecm) { ec64_insert(e->ec, key, L, gi); acc.stats[1]++; }
                else {
                    uint32_t idx = 0; int res;
                    if (L == (int)e->lib.pk.len) {
                        int ex = packed_exact(e->lib, key);
                        if (ex >= 0) { res = R_PERFECT; idx = (uint32_t)ex; }
                        else if (e->run.miss > 0) res = packed_near_decide(e->run, e->lib, key, 0u, idx);
                        else res = R_NONALIGNED;
                    } else {
                        int ex = lib_exact(e->lib, key, L);
                        if (ex >= 0) { res = R_PERFECT; idx = (uint32_t)ex; }
                        else {
                            MinTrack tt; tt.init(e->run.miss);
                            if (e->run.miss > 0) lib_near(e->lib, key, L, 0ull, tt);
                            if (tt.cnt == 1) { res = R_IMPERFECT; idx = tt.idx; } else res = R_NONALIGNED;
                        }
                    }
                    if (res == 1 || res == 2) acc.counts[idx]++;
                    acc.stats[res]++;
                }
            }
        };
        for (uint32_t t = 0; t < hp.n_tiles; t++)
            for (uint32_t lane = 0; lane < F2Q_TILE; lane++) {
                if (hp.planar_nw == 10) lane_fn(std::integral_constant<int, 10>(), std::integral_constant<int, 3>(), t, lane);     // reads of 161 .. 320 bases
                else if (hp.planar_nw == 3 && e->plan.kb == 0) lane_fn(std::integral_constant<int, 3>(), std::integral_constant<int, 0>(), t, lane);
                else if (hp.planar_nw == 5 && e->plan.kb == 0) lane_fn(std::integral_constant<int, 5>(), std::integral_constant<int, 0>(), t, lane);
                else if (hp.planar_nw == 3 && e->plan.kb == 1) lane_fn(std::integral_constant<int, 3>(), std::integral_constant<int, 1>(), t, lane);
                else if (hp.planar_nw == 3) lane_fn(std::integral_constant<int, 3>(), std::integral_constant<int, 3>(), t, lane);
                else if (e->plan.kb == 1) lane_fn(std::integral_constant<int, 5>(), std::integral_constant<int, 1>(), t, lane);
                else lane_fn(std::integral_constant<int, 5>(), std::integral_constant<int, 3>(), t, lane);
            }
        e->anchor_reads += hp.n_clean;
    } else {
    if (e->run.mode == 1) {
        // k_extract_fixed4: Extract+Count with a fixed window on packed tiles
        const FixedGeom g = fixed_geom(e->run);
        for (uint32_t t = 0; t < hp.n_tiles; t++)
            for (uint32_t lane = 0; lane < 64; lane++) {
                U4 b[F2Q_MAXBROWS], qr[F2Q_MAXQROWS]; uint32_t bad[4] = {0, 0, 0, 0};
                const uint32_t *qp = pb.qual + ((uint64_t)t * pb.wq) * F2Q_TILE + 4 * lane;
                const uint32_t *bp = pb.bases + ((uint64_t)t * pb.wb) * F2Q_TILE + 4 * lane;
                for (int r = 0; r < F2Q_MAXBROWS; r++) {
                    uint32_t row = g.bw0 + (r < g.nb ? r : (g.nb > 0 ? g.nb - 1 : 0));
                    row = row < pb.wb ? row : pb.wb - 1;
                    const uint32_t *p = bp + (uint64_t)row * F2Q_TILE; b[r] = U4{p[0], p[1], p[2], p[3]};
                }
                for (int r = 0; r < F2Q_MAXQROWS; r++) {
                    const uint32_t want = g.qw0 + (r < g.nq ? r : (g.nq > 0 ? g.nq - 1 : 0));
                    const uint32_t row = want < pb.wq ? want : pb.wq - 1;
                    const uint32_t *p = qp + (uint64_t)row * F2Q_TILE;
                    qr[r] = want < pb.wq ? U4{p[0], p[1], p[2], p[3]} : U4{0, 0, 0, 0};
                }
                if (g.add_hi)
                    for (int r = 0; r < F2Q_MAXQROWS; r++)
                        if (r < g.nq) fixed4_qrow(g, r, qr[r], bad);
                for (int j = 0; j < 4; j++) {
                    const uint32_t l = pb.len[(uint64_t)t * F2Q_TILE + 4 * lane + j];
                    if (l == F2Q_LEN_SKIP) continue;
                    acc.stats[0]++; e->v2_reads++;
                    if (bad[j]) { acc.stats[4]++; continue; }
                    if (ec64_insert_word(e->ec, fixed4_ec_word(g, b, qr, j, l), e->reads_seen + hp_index[(uint64_t)t * F2Q_TILE + 4 * lane + j])) e->ec.ctr[3]++;
                    acc.stats[1]++;
                }
            }
    } else {
    const bool v2 = e->use_v2 && e->lib.pk.len == (uint32_t)e->run.length && e->lib.pk.len > 0 && e->lib.n_irregular == 0;
    const int mw_total = e->run.n_iter * e->run.length;
    if (e->plan.multi && e->use_lt && e->lib.lt.ok && e->lib.lt.len == (uint32_t)mw_total && e->run.miss <= 1) {
        // k_count_fixed4_lds<.., MW>'s per-lane sequence: the compact window as ONE window, the Phred rule per part, the
        // joined key against the LDS tables of the library's n_iter-part features
        const FixedGeom g = fixed_geom_at(0, mw_total, e->run.thr);
        const LtDesc &lt = e->lib.lt;
        for (uint32_t t = 0; t < hp.n_tiles; t++)
            for (uint32_t lane = 0; lane < 64; lane++) {
                U4 b[F2Q_MAXBROWS], qr[F2Q_MAXQROWS];
                const uint32_t *qp = pb.qual + ((uint64_t)t * pb.wq) * F2Q_TILE + 4 * lane;
                const uint32_t *bp = pb.bases + ((uint64_t)t * pb.wb) * F2Q_TILE + 4 * lane;
                for (int r = 0; r < F2Q_MAXBROWS; r++) {
                    uint32_t row = g.bw0 + (r < g.nb ? r : (g.nb > 0 ? g.nb - 1 : 0));
                    row = row < pb.wb ? row : pb.wb - 1;
                    const uint32_t *p = bp + (uint64_t)row * F2Q_TILE; b[r] = U4{p[0], p[1], p[2], p[3]};
                }
                for (int r = 0; r < F2Q_MAXQROWS; r++) {
                    uint32_t row = g.qw0 + (r < g.nq ? r : (g.nq > 0 ? g.nq - 1 : 0));
                    row = row < pb.wq ? row : pb.wq - 1;
                    const uint32_t *p = qp + (uint64_t)row * F2Q_TILE; qr[r] = U4{p[0], p[1], p[2], p[3]};
                }
                for (int j = 0; j < 4; j++) {
                    const uint32_t l = pb.len[(uint64_t)t * F2Q_TILE + 4 * lane + j];
                    if (l == F2Q_LEN_SKIP) continue;
                    acc.stats[0]++; e->v2_reads++; e->lt_reads++;
                    const int pv = g.add_hi ? mw_part_verdict(fixed4_failbits(g, qr, j), e->run.n_iter, e->run.length) : 0;
                    if (pv == 2) { acc.stats[R_QFAIL]++; continue; }
                    if (pv == 1 || (int)(l & F2Q_LEN_MASK) < mw_total) { acc.stats[R_NONALIGNED]++; continue; }
                    uint32_t forced = (l & F2Q_LEN_FLAG) ? fixed4_flags(g, qr, j) : 0u;
                    uint64_t key = fixed4_key(g, b, j);
                    if (lt.mix) { key = mw_mix(key, lt.mix); forced = mw_mix_mask(forced, lt.mix); }
                    const LtProbe q = lt_probe(lt, key);
                    U2 en[4];
                    for (int k = 0; k < 4; k++) {
                        const uint32_t *tb = lt.tags + (size_t)(k >> 1) * F2Q_LT_SLOTS + 2u * q.b[k];
                        en[k] = U2{tb[0], tb[1]};
                    }
                    auto rd0 = [&](uint32_t bk) { return U2{lt.tags[2u * bk], lt.tags[2u * bk + 1u]}; };
                    int res; uint32_t slot = 0;
                    if (forced && (e->run.miss == 0 || __builtin_popcount(forced) > e->run.miss)) res = R_NONALIGNED;
                    else {
                        const LtVerdict v = e->run.miss > 0 ? lt_decide<true>(lt, q, en, forced, rd0) : lt_decide<false>(lt, q, en, forced, rd0);
                        res = v.perfect ? R_PERFECT : v.imperfect ? R_IMPERFECT : R_NONALIGNED; slot = v.slot;
                    }
                    if (res == R_PERFECT || res == R_IMPERFECT) acc.counts[lt.feat_of[slot]]++;
                    acc.stats[res]++;
                }
            }
    } else
    if (e->plan.multi) {
        // k_count_multi4's per-lane sequence: windows one after another, parts that pass concatenated, k-part tables
        const int W = e->run.n_iter, L = e->run.length;
        for (uint32_t t = 0; t < hp.n_tiles; t++)
            for (uint32_t lane = 0; lane < 64; lane++) {
                const uint32_t *qp = pb.qual + ((uint64_t)t * pb.wq) * F2Q_TILE + 4 * lane;
                const uint32_t *bp = pb.bases + ((uint64_t)t * pb.wb) * F2Q_TILE + 4 * lane;
                uint64_t key[4] = {0, 0, 0, 0}; uint32_t forced[4] = {0, 0, 0, 0}, npart[4] = {0, 0, 0, 0}, lv[4]; int res[4];
                for (int j = 0; j < 4; j++) {
                    lv[j] = pb.len[(uint64_t)t * F2Q_TILE + 4 * lane + j];
                    res[j] = lv[j] == F2Q_LEN_SKIP ? R_SKIP : ((int)(lv[j] & F2Q_LEN_MASK) < e->plan.need) ? R_SLOW : R_NEAR;
                }
                for (int w = 0; w < W; w++) {
                    const FixedGeom g = fixed_geom_of(e->run, w);
                    U4 b[F2Q_MAXBROWS], qr[F2Q_MAXQROWS]; uint32_t bad[4] = {0, 0, 0, 0};
                    for (int r = 0; r < F2Q_MAXBROWS; r++) {
                        uint32_t row = g.bw0 + (r < g.nb ? r : g.nb - 1);
                        row = row < pb.wb ? row : pb.wb - 1;
                        const uint32_t *p = bp + (uint64_t)row * F2Q_TILE; b[r] = U4{p[0], p[1], p[2], p[3]};
                    }
                    for (int r = 0; r < F2Q_MAXQROWS; r++) {
                        uint32_t row = g.qw0 + (r < g.nq ? r : g.nq - 1);
                        row = row < pb.wq ? row : pb.wq - 1;
                        const uint32_t *p = qp + (uint64_t)row * F2Q_TILE; qr[r] = U4{p[0], p[1], p[2], p[3]};
                    }
                    if (g.add_hi)
                        for (int r = 0; r < F2Q_MAXQROWS; r++)
                            if (r < g.nq) fixed4_qrow(g, r, qr[r], bad);
                    for (int j = 0; j < 4; j++) {
                        if (res[j] != R_NEAR || bad[j]) continue;
                        key[j] |= fixed4_key(g, b, j) << (2u * (uint32_t)L * npart[j]);
                        if (lv[j] & F2Q_LEN_FLAG) forced[j] |= fixed4_flags(g, qr, j) << ((uint32_t)L * npart[j]);
                        npart[j]++;
                    }
                }
                for (int j = 0; j < 4; j++) {
                    if (res[j] == R_SKIP) continue;
                    e->v2_reads++;
                    if (res[j] == R_SLOW) {
                        const uint32_t slot = 4 * lane + j; const int r = (int)(lv[j] & F2Q_LEN_MASK);
                        uint8_t sq[F2Q_ANCHOR_MAXLEN], ql[F2Q_ANCHOR_MAXLEN];
                        const uint32_t *bps = pb.bases + ((uint64_t)t * pb.wb) * F2Q_TILE + slot, *qps = pb.qual + ((uint64_t)t * pb.wq) * F2Q_TILE + slot;
                        for (int i = 0; i < r; i++) {
                            sq[i] = (uint8_t)"ACGT"[(bps[(uint64_t)(i >> 4) * F2Q_TILE] >> (2 * (i & 15))) & 3u];
                            ql[i] = (uint8_t)((qps[(uint64_t)(i >> 2) * F2Q_TILE] >> (8 * (i & 3))) & 0xFFu);
                            if (ql[i] & 0x80u) { sq[i] = (uint8_t)'N'; ql[i] &= 0x7Fu; }
                        }
                        general_read<const uint8_t *>(e->run, e->lib, e->ec, acc, sq, r, ql, r, 0ull, acc.stats);
                        continue;
                    }
                    acc.stats[0]++;
                    uint32_t idx = 0; int rr;
                    if (npart[j] == 0) rr = R_QFAIL;
                    else {
                        const PackedGroup &pk = e->lib.mpk[npart[j] - 1];
                        if (forced[j]) rr = (e->run.miss == 0 || __builtin_popcount(forced[j]) > e->run.miss) ? R_NONALIGNED
                                                                                                             : packed_near_decide(e->run, e->lib, pk, key[j], forced[j], idx);
                        else {
                            const int ex = pk.len ? packed_exact(e->lib, pk, key[j]) : -1;
                            if (ex >= 0) { rr = R_PERFECT; idx = (uint32_t)ex; }
                            else rr = e->run.miss > 0 ? packed_near_decide(e->run, e->lib, pk, key[j], 0u, idx) : R_NONALIGNED;
                        }
                    }
                    if (rr == R_PERFECT || rr == R_IMPERFECT) acc.counts[idx]++;
                    acc.stats[rr]++;
                }
            }
    } else
    if (v2) {
        // the v2 kernel's per-lane sequence: 4 reads per lane from 16-byte row loads, packed tables
        const FixedGeom g = fixed_geom(e->run);
        const int need = g.st + g.L;
        const bool rows_ok = (uint32_t)(g.qw0 + g.nq) <= pb.wq && (uint32_t)(g.bw0 + g.nb) <= pb.wb &&
                             g.L >= 1 && g.L <= F2Q_REG_MAXLEN && e->lib.grp[g.L].n == e->lib.n_features;
        for (uint32_t t = 0; t < hp.n_tiles; t++)
            for (uint32_t lane = 0; lane < 64; lane++) {
                U4 b[F2Q_MAXBROWS], qr[F2Q_MAXQROWS]; uint32_t bad[4] = {0, 0, 0, 0};
                const uint32_t *qp = pb.qual + ((uint64_t)t * pb.wq) * F2Q_TILE + 4 * lane;
                const uint32_t *bp = pb.bases + ((uint64_t)t * pb.wb) * F2Q_TILE + 4 * lane;
                for (int r = 0; r < F2Q_MAXBROWS; r++) {
                    uint32_t row = g.bw0 + (r < g.nb ? r : (g.nb > 0 ? g.nb - 1 : 0));
                    row = row < pb.wb ? row : pb.wb - 1;
                    const uint32_t *p = bp + (uint64_t)row * F2Q_TILE; b[r] = U4{p[0], p[1], p[2], p[3]};
                }
                for (int r = 0; r < F2Q_MAXQROWS; r++) {
                    uint32_t row = g.qw0 + (r < g.nq ? r : (g.nq > 0 ? g.nq - 1 : 0));
                    row = row < pb.wq ? row : pb.wq - 1;
                    const uint32_t *p = qp + (uint64_t)row * F2Q_TILE;
                    qr[r] = U4{p[0], p[1], p[2], p[3]};          // also with the Phred rule off: the flag bits travel here
                }
                if (g.add_hi)
                    for (int r = 0; r < F2Q_MAXQROWS; r++)
                        if (r < g.nq) fixed4_qrow(g, r, qr[r], bad);
                for (int j = 0; j < 4; j++) {
                    uint32_t l = pb.len[(uint64_t)t * F2Q_TILE + 4 * lane + j];
                    int res; uint32_t idx = 0;
                    if (l == F2Q_LEN_SKIP) res = R_SKIP;
                    else if (g.L < 1 || ((int)(l & F2Q_LEN_MASK) < need && !rows_ok))
                        res = fixed_lane(e->run, e->lib, pb, t, 4 * lane + j, idx);
                    else if ((int)(l & F2Q_LEN_MASK) < need) res = bad[j] ? R_QFAIL : R_NONALIGNED;   // clipped window, uniform library
                    else if (bad[j]) res = R_QFAIL;
                    else if (e->lib.pt.ok && rows_ok && (e->ix.pt_force_parts > 0 || !e->lib.lt.ok)) {
                        // k_part_scatter + k_part_count: the entry as it travels between the passes, the partition's table 0,
                        // the global table 1, hits by table-0 slot or (through table 1) by table-1 slot
                        const PtDesc &pt = e->lib.pt;
                        const uint32_t forced = (l & F2Q_LEN_FLAG) ? fixed4_flags(g, qr, j) : 0u;
                        if (forced && (e->run.miss == 0 || (forced & (forced - 1u)) != 0u)) res = R_NONALIGNED;
                        else {
                            const unsigned long long ent = pt_entry(fixed4_key(g, b, j), forced, pt.len);
                            const uint64_t key = ent & ((1ull << (2u * pt.len)) - 1ull);
                            const uint32_t fo = (uint32_t)(ent >> (2u * pt.len)) & ((1u << pt.len) - 1u);
                            const uint32_t p = pt_part((uint32_t)key & ((1u << pt.hb0) - 1u), pt.n_parts);
                            LtDesc lt{}; lt.hb0 = pt.hb0; lt.hb1 = pt.hb1; lt.len = pt.len;
                            const LtProbe q = lt_probe(lt, key, pt.bb1);
                            const uint32_t *t0 = pt.tags0 + (size_t)p * F2Q_LT_SLOTS;
                            U2 en[4];
                            en[0] = U2{t0[2u * q.b[0]], t0[2u * q.b[0] + 1u]}; en[1] = U2{t0[2u * q.b[1]], t0[2u * q.b[1] + 1u]};
                            // table 1: the first-choice bucket; the second one only where the first carries the mark (PtDesc::spill)
                            en[2] = U2{pt.tags1[2u * q.b[2]], pt.tags1[2u * q.b[2] + 1u]}; en[3] = U2{F2Q_LT_EMPTY, F2Q_LT_EMPTY};
                            if (!pt.spill || (en[2].x != F2Q_LT_EMPTY && (en[2].x & pt.spill))) en[3] = U2{pt.tags1[2u * q.b[3]], pt.tags1[2u * q.b[3] + 1u]};
                            if (pt.spill && en[2].x != F2Q_LT_EMPTY) en[2].x &= ~pt.spill;
                            if (pt.spill && en[3].x != F2Q_LT_EMPTY) en[3].x &= ~pt.spill;
                            if (e->run.miss == 0) { en[2] = U2{F2Q_LT_EMPTY, F2Q_LT_EMPTY}; en[3] = en[2]; }
                            // k_part_count: the exact hit from the partition's table 0 (equality with the query's own tags);
                            // every other entry waits in the ring and is decided by lt_near1 over the eight tags
                            const int ex = fo == 0u ? lt_exact(lt, q, en[0], en[1]) : -1;
                            uint32_t hit = 0, hitw = 0;
                            if (ex >= 0) { res = R_PERFECT; idx = pt.feat0_of[(size_t)p * F2Q_LT_SLOTS + (uint32_t)ex]; }
                            else if (e->run.miss > 0 && lt_near1(lt, q, en, fo, hit, hitw) == 1u) {
                                res = R_IMPERFECT;
                                if (hit >> 31) {                 // k_part_reduce: the feature whose table-1 slot this is
                                    idx = ~0u;
                                    for (uint32_t gi = 0; gi < e->lib.n_features; gi++) if (pt.slot1_of[gi] == (hit & 0x7FFFFFFFu)) { idx = pt.feat_of[gi]; break; }
                                } else idx = pt.feat0_of[(size_t)p * F2Q_LT_SLOTS + hit];
                            } else res = R_NONALIGNED;
                        }
                        e->pt_reads++;
                    }
                    else if (e->use_lt && e->lib.lt.ok && rows_ok) {
                        // k_count_fixed4_lds: the LDS tables decide (exact hit, else the unique feature at distance 1)
                        const LtDesc &lt = e->lib.lt;
                        const uint64_t key = fixed4_key(g, b, j);
                        const uint32_t forced = (l & F2Q_LEN_FLAG) ? fixed4_flags(g, qr, j) : 0u;
                        const LtProbe q = lt_probe(lt, key);
                        U2 en[4];
                        for (int k = 0; k < 4; k++) {
                            const uint32_t *tb = lt.tags + (size_t)(k >> 1) * F2Q_LT_SLOTS + 2u * q.b[k];
                            en[k] = U2{tb[0], tb[1]};
                        }
                        auto rd0 = [&](uint32_t bk) { return U2{lt.tags[2u * bk], lt.tags[2u * bk + 1u]}; };
                        uint32_t slot = 0;
                        if (forced && (e->run.miss == 0 || __builtin_popcount(forced) > e->run.miss)) res = R_NONALIGNED;
                        else {
                            const LtVerdict v = e->run.miss > 0 ? lt_decide<true>(lt, q, en, forced, rd0) : lt_decide<false>(lt, q, en, forced, rd0);
                            res = v.perfect ? R_PERFECT : v.imperfect ? R_IMPERFECT : R_NONALIGNED; slot = v.slot;
                        }
                        idx = lt.feat_of[slot];
                        e->lt_reads++;
                    }
                    else {
                        uint64_t key = fixed4_key(g, b, j);
                        uint32_t forced = (l & F2Q_LEN_FLAG) ? fixed4_flags(g, qr, j) : 0u;
                        if (forced) {
                            if (e->run.miss == 0 || __builtin_popcount(forced) > e->run.miss) res = R_NONALIGNED;
                            else res = packed_near_decide(e->run, e->lib, key, forced, idx);
                        } else {
                            int ex = packed_exact(e->lib, key);
                            if (ex >= 0) { res = R_PERFECT; idx = (uint32_t)ex; }
                            else if (e->run.miss > 0) res = packed_near_decide(e->run, e->lib, key, 0u, idx);
                            else res = R_NONALIGNED;
                        }
                    }
                    if (res == 1 || res == 2) acc.counts[idx]++;
                    if (res) { acc.stats[0]++; acc.stats[res]++; e->v2_reads++; }
                }
            }
    } else
    for (uint32_t t = 0; t < hp.n_tiles; t++)
        for (uint32_t lane = 0; lane < F2Q_TILE; lane++) {
            uint32_t idx = 0;
            int res = fixed_lane(e->run, e->lib, pb, t, lane, idx);
            if (res == 1 || res == 2) acc.counts[idx]++;
            if (res) { acc.stats[0]++; acc.stats[res]++; }
        }
    }
    }
    for (size_t g = 0; g < hp.g_len.size(); g++) {
        const uint8_t *seq = hp.raw.data() + hp.g_off[g];
        general_read<const uint8_t *, true>(e->run, e->lib, e->ec, acc, seq, (int)hp.g_len[g], seq + hp.g_len[g], (int)hp.g_qlen[g],
                     e->reads_seen + hp.g_index[g], acc.stats);
    }
    e->reads_seen += recs.size(); e->fast += hp.n_clean; e->general += hp.g_len.size();
    return used;
}

void emu_read_counts(void *h, int64_t *counts, int64_t *stats, uint64_t *fast, uint64_t *general)
{
    Emu *e = (Emu *)h;
    for (uint32_t i = 0; i < e->ix.n_features; i++) counts[i] = (int64_t)e->acc[i];
    for (int k = 0; k < 5; k++) stats[k] = (int64_t)e->acc[e->ix.n_features + k];
    if (fast) *fast = e->fast;
    if (general) *general = e->general;
}

void emu_set_read_base(void *h, uint64_t b) { ((Emu *)h)->reads_seen = b; }
void emu_reset(void *h) { Emu *e = (Emu *)h; std::fill(e->acc.begin(), e->acc.end(), 0ull); e->reads_seen = 0; e->fast = 0; e->general = 0; }   // f2q_reset_counts (Counter mode)
void emu_use_v2(void *h, int on) { ((Emu *)h)->use_v2 = on; }
void emu_use_lt(void *h, int on) { ((Emu *)h)->use_lt = on; }
void emu_use_pw(void *h, int on) { ((Emu *)h)->use_pw = on; }      // before emu_set_features
int emu_pw_ok(void *h) { return (int)((Emu *)h)->lib.pw.ok; }
uint64_t emu_lt_reads(void *h) { return ((Emu *)h)->lt_reads; }
int emu_lt_ok(void *h) { return (int)((Emu *)h)->ix.lt.ok; }
void emu_pt_force(void *h, int parts) { ((Emu *)h)->ix.pt_force_parts = parts; }      // before emu_set_features
uint64_t emu_pt_reads(void *h) { return ((Emu *)h)->pt_reads; }
int emu_pt_parts(void *h) { return ((Emu *)h)->ix.pt.ok ? (int)((Emu *)h)->ix.pt.n_parts : 0; }
uint64_t emu_v2_reads(void *h) { return ((Emu *)h)->v2_reads; }
uint64_t emu_anchor_reads(void *h) { return ((Emu *)h)->anchor_reads; }
// entries: first the byte-string table, then the occupied slots of the single-word table
static std::vector<size_t> k64_live(Emu *e) { std::vector<size_t> v; for (size_t i = 0; i < e->k64s.size(); i++) if (e->k64s[i] != ~0ull) v.push_back(i); return v; }
uint64_t emu_ec_n(void *h) { Emu *e = (Emu *)h; return e->ctr[0] + k64_live(e).size(); }
uint64_t emu_ec_overflow(void *h) { return ((Emu *)h)->ctr[2]; }
void emu_ec_get(void *h, uint64_t e_, char *key, uint32_t *len, int64_t *count, uint64_t *first)
{
    Emu *e = (Emu *)h;
    if (e_ >= e->ctr[0]) {
        const size_t s = k64_live(e)[e_ - e->ctr[0]];
        const unsigned long long k = e->k64s[s];
        *len = ec64_text(k, key); *count = (int64_t)e->k64c[s] + 1; *first = e->k64f[s];
        return;
    }
    *len = e->ent_len[e_]; *count = (int64_t)e->ent_count[e_]; *first = e->ent_first[e_];
    memcpy(key, (const uint8_t *)(e->arena.data() + e->ent_off[e_]), *len);
}

// host twin of the synthetic generator (same code the library's f2q_synth_fastq runs)
size_t emu_synth_fastq(void *h, const f2q_synth *s, uint64_t lo, uint64_t hi, uint8_t *buf)
{
    Emu *e = (Emu *)h;
    SynthDev d; memset(&d, 0, sizeof d);
    d.seed = s->seed; d.n_reads = s->n_reads; d.first_read = s->first_read; d.read_len = s->read_len; d.start = s->start;
    d.cassette = s->cassette; d.max_offset = s->max_offset; d.glen = (int)(e->ix.feat_off[1] - e->ix.feat_off[0]);
    d.n_guides = (int)e->ix.n_features;
    d.t_sub = s->t_sub; d.t_rand = s->t_rand; d.t_n = s->t_n; d.t_lowq = s->t_lowq; d.t_q29 = s->t_q29; d.t_q28 = s->t_q28;
    if (s->cassette) { d.up_len = (int)strlen(s->up); d.down_len = (int)strlen(s->down); memcpy(d.up, s->up, d.up_len); memcpy(d.down, s->down, d.down_len); }
    const int R = d.read_len; size_t o = 0;
    const uint64_t *keys = e->ix.key2.data();
    for (uint64_t i = lo; i < hi; i++) {
        SynthRead r = synth_plan(d, i, [&](uint32_t g) { return keys[g]; });
        o += (size_t)sprintf((char *)buf + o, "@r%llu\n", (unsigned long long)i);
        uint64_t fw = 0;
        for (int p = 0; p < R; p++) { if ((p & 31) == 0) fw = rnd(d.seed, i, F_FLANK0 + (p >> 5)); buf[o + p] = synth_base(d, r, p, fw); }
        o += R; buf[o++] = '\n'; buf[o++] = '+'; buf[o++] = '\n';
        for (int p = 0; p < R; p++) buf[o + p] = (p == r.qpos) ? r.qchar : (uint8_t)'I';
        o += R; buf[o++] = '\n';
    }
    return o;
}

uint32_t emu_crc32(uint32_t crc, const uint8_t *p, size_t n) { return f2qz::Crc32::get().update(crc, p, n); }

// the file reader of f2q_count_file (f2q_reader.h), piece size `cap`: returns the bytes decoded, -1 if out is too small
long long emu_read_file(const char *path, size_t cap, int threads, uint8_t *out, size_t out_cap, int *truncated, int *kind)
{
    TextSource src; std::string err;
    if (src.open(path, err) != 0) return -2;
    if (threads > 0) src.n_threads = threads;
    if (kind) *kind = (int)src.kind;
    std::vector<uint8_t> piece(cap);
    size_t total = 0;
    for (;;) {
        const size_t n = src.read(piece.data(), cap);
        if (n == 0) break;
        if (total + n > out_cap) return -1;
        memcpy(out + total, piece.data(), n); total += n;
    }
    if (truncated) *truncated = src.truncated() ? 1 : 0;
    if (kind) *kind = (int)src.kind;
    return (long long)total;
}

} // extern "C"
