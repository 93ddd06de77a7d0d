// f2q_lib.hip -- libf2q_hip.so: the one translation unit.  Kernels live in f2q_count_kernels.h and
// f2q_aux_kernels.h, the per-lane logic in f2q_device.h; this file holds the host side and the C ABI of include/f2q.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <unordered_map>
#include <mutex>
#include <thread>
#include <vector>

#include "f2q_device.h"
#include "f2q_host.h"
#include "f2q_synth.h"
#include "f2q_reader.h"

using namespace f2q;

#include "f2q_count_kernels.h"
#include "f2q_part_kernels.h"
#include "f2q_aux_kernels.h"

// ===============================================================================================
// host side
// ===============================================================================================
struct f2q_block {
    PackedBlock pb{};
    RawBlock rb{};
    std::vector<void *> allocs;
    uint64_t n_reads = 0, n_general = 0, dev_bytes = 0;
    uint64_t raw_key_bytes = 0;            // upper bound of the key bytes the raw records can produce (Extract+Count arena sizing)
};

struct f2q_ctx {
    f2q_params prm{};
    std::vector<std::string> up_s, down_s;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
    std::vector<hipEvent_t> q_ev;          // f2q_count_resident_queued: a pair of events per queued step
    uint32_t q_n = 0;                      // steps queued since the last f2q_queued_times
    hipStream_t copy_stream = nullptr;   // f2q_count_file: the text of the next piece travels while this one is counted
    hipEvent_t ev_copy = nullptr;
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
    std::vector<void *> ec_allocs;       // byte-string side: slots, entry arrays, arena
    std::vector<void *> ec_allocs64;     // single-word side: k64_*
    std::vector<void *> ec_allocs_ctr;   // the four counters (outlive the growth of either side)
    // raw records of an Extract+Count block are decided on a second stream while the packed tiles are counted
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_aux0 = nullptr, ev_aux1 = nullptr;
    bool aux_busy = false;
    uint64_t ec_slots = 0;
    // hot keys of the single-word table (EcHot): learnt from the first hot_learn reads of a sample, kept in LDS by
    // k_extract_anchor_hot; `defer` lists the reads that kernel sets aside for k_ec_deferred
    EcHot hot{};
    std::vector<void *> hot_allocs;
    bool hot_valid = false, no_hot = false;
    uint64_t hot_learn = (uint64_t)1 << 18, ec_learned = 0;
    unsigned long long *defer_d = nullptr; size_t defer_cap = 0;
    uint64_t reads_seen = 0;             // global read index of the next block's read 0
    uint32_t last_path = 0;              // F2Q_PATH_* of the last packed-tile launch (f2q_timing.path)
    int n_cu = 256;
    bool force_generic = false;           // F2Q_GENERIC=1: run-time window geometry even where a specialisation exists
    bool host_pack = false;               // F2Q_HOST_PACK=1: frame/classify/pack on the host (the round-1 first path; A/B runs)
    bool force_general = false;           // F2Q_FORCE_GENERAL=1: every read through the byte-exact general kernel (cross-checks)
    bool force_v1 = false;                // F2Q_FORCE_V1=1: keep the one-read-per-lane kernel (A/B runs)
    bool no_lt = false;                   // F2Q_NO_LT=1: never the LDS-table kernel (A/B runs, cross-checks)
    bool no_pt = false;                   // F2Q_NO_PT=1: never the partitioned-table kernels (A/B runs, cross-checks)
    uint64_t pt_chunk_reads = (uint64_t)1 << 28;   // F2Q_PT_CHUNK: most reads per scatter/count round of the partitioned path (1.7 GB of streams per 200 M reads; 400 M reads in two rounds run 8 % faster than in six)
    uint64_t pt_min_reads = (uint64_t)1 << 21;     // F2Q_PT_MIN_READS: smaller blocks keep the packed-table kernel (three launches and the
                                                   // tables' way into LDS do not pay for a block that small)
    // scratch of the partitioned path (k_part_*): entry streams and their lengths, the two slabs, the stats rows
    PartScratch pt_s{};
    size_t pt_streams_n = 0, pt_cnt_n = 0, pt_slab0_n = 0, pt_slab1_n = 0;
    uint32_t *pt_slab0_d = nullptr, *pt_slab1_d = nullptr;
    unsigned long long *pt_stat_d = nullptr; size_t pt_stat_n = 0;
    // device memory freed by blocks / scratch is kept (idle, after a stream sync) for the next piece of the same
    // size class: a streamed file costs ~20 allocations per piece otherwise
    std::multimap<size_t, void *> dev_idle;
    std::unordered_map<void *, size_t> dev_size;
    size_t dev_idle_bytes = 0, dev_idle_cap = (size_t)8 << 30;
    std::string err;
    // F2Q_TRACE=1: wall-clock split of the host entry points, printed by f2q_count_file (diagnostics only)
    bool trace = false, trace_sync = false;
    double tr_frame = 0, tr_count = 0, tr_free = 0, tr_copy = 0, tr_malloc = 0, tr_hipfree = 0, tr_reserve = 0;
    uint64_t n_malloc = 0, n_hipfree = 0, n_reuse = 0, n_rehash = 0;
};

static thread_local std::string g_create_err;
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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

static int dev_get(f2q_ctx *c, size_t bytes, void **out)
{
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = c->dev_idle.lower_bound(bytes);
    if (it != c->dev_idle.end() && it->first <= bytes + bytes / 4 + (64 << 10)) {
        *out = it->second; c->dev_idle_bytes -= it->first; c->dev_idle.erase(it);
        c->n_reuse++;
        return F2Q_OK;
    }
    void *p = nullptr;
    const double m0 = now_ms();
    hipError_t e = hipMalloc(&p, bytes);
    c->tr_malloc += now_ms() - m0; c->n_malloc++;
    if (e != hipSuccess && !c->dev_idle.empty()) {           // give the idle memory back and retry once
        for (auto &kv : c->dev_idle) { c->dev_size.erase(kv.second); (void)hipFree(kv.second); }
        c->dev_idle.clear(); c->dev_idle_bytes = 0;
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) return fail(c, F2Q_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    c->dev_size[p] = bytes;
    *out = p;
    return F2Q_OK;
}
template <class T>
static int dev_upload(f2q_ctx *c, const T *src, size_t n, T **dst, std::vector<void *> &owner)
{
    void *p = nullptr;
    int rc = dev_get(c, (n ? n : 1) * sizeof(T), &p);
    if (rc) return rc;
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
    int rc = dev_get(c, bytes, &p);
    if (rc) return rc;
    owner.push_back(p);
    if (fill >= 0) HIPC(c, hipMemsetAsync(p, fill, bytes, c->stream));
    *dst = (T *)p;
    return F2Q_OK;
}
// memory goes back to the idle list only once nothing queued on the stream can still touch it
static void free_all(f2q_ctx *c, std::vector<void *> &v)
{
    if (v.empty()) return;
    if (!c) { for (void *p : v) (void)hipFree(p); v.clear(); return; }      // block outliving its context
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : v) {
        auto it = c->dev_size.find(p);
        if (it != c->dev_size.end() && c->dev_idle_bytes + it->second <= c->dev_idle_cap) {
            c->dev_idle.emplace(it->second, p); c->dev_idle_bytes += it->second;
        } else {
            if (it != c->dev_size.end()) c->dev_size.erase(it);
            const double f0 = now_ms();
            (void)hipFree(p);
            c->tr_hipfree += now_ms() - f0; c->n_hipfree++;
        }
    }
    v.clear();
}

extern "C" int f2q_version(void) { return F2Q_ABI_VERSION; }

#ifndef F2Q_BUILD_ID
#define F2Q_BUILD_ID "unknown"
#endif
// the "F2Q_BUILD_ID=" prefix lets build() find the id in the file without loading it
static const char g_build_id[] = "F2Q_BUILD_ID=" F2Q_BUILD_ID;
extern "C" const char *f2q_build_id(void) { return g_build_id + 13; }

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
    if (c->force_general) { c->plan.fast_fixed = false; c->plan.fast_anchor = false; c->plan.multi_pair = false; }
    return F2Q_OK;
}

static int upload_lib(f2q_ctx *c)
{
    free_all(c, c->lib_allocs);
    LibDev &L = c->lib_h;
    memset(&L, 0, sizeof L);
    L.n_features = c->ix.n_features;
    L.n_irregular = c->ix.n_irregular;
    memcpy(L.grp, c->ix.grp, sizeof L.grp);
    L.pk = c->ix.pk;
    memcpy(L.mpk, c->ix.mpk, sizeof L.mpk);
    L.mw_ok = c->ix.mw_ok;
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
    {
        uint32_t *lt_tags, *lt_feat; uint16_t *lt_slot;
        if ((rc = dev_upload(c, c->ix.lt_tags.data(), c->ix.lt_tags.size(), &lt_tags, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.lt_feat_of.data(), c->ix.lt_feat_of.size(), &lt_feat, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.lt_slot_of.data(), c->ix.lt_slot_of.size(), &lt_slot, c->lib_allocs))) return rc;
        L.lt = c->ix.lt; L.lt.tags = lt_tags; L.lt.feat_of = lt_feat; L.lt.slot_of = lt_slot;
        uint64_t *pw_tab;
        if ((rc = dev_upload(c, c->ix.pw_tab.data(), c->ix.pw_tab.size(), &pw_tab, c->lib_allocs))) return rc;
        L.pw = c->ix.pw; L.pw.tab = pw_tab;
        { const char *e = getenv("F2Q_NO_PW"); if (e && e[0] == '1') L.pw.ok = 0; }      // A/B runs: the joined key as a string (byte-string index)
        uint32_t *pt_t0, *pt_t1, *pt_ps, *pt_s1, *pt_fo, *pt_f0; uint16_t *pt_s0;
        if ((rc = dev_upload(c, c->ix.pt_tags0.data(), c->ix.pt_tags0.size(), &pt_t0, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.pt_tags1.data(), c->ix.pt_tags1.size(), &pt_t1, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.pt_pstart.data(), c->ix.pt_pstart.size(), &pt_ps, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.pt_slot0_of.data(), c->ix.pt_slot0_of.size(), &pt_s0, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.pt_slot1_of.data(), c->ix.pt_slot1_of.size(), &pt_s1, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.pt_feat_of.data(), c->ix.pt_feat_of.size(), &pt_fo, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.pt_feat0_of.data(), c->ix.pt_feat0_of.size(), &pt_f0, c->lib_allocs))) return rc;
        L.pt = c->ix.pt; L.pt.tags0 = pt_t0; L.pt.tags1 = pt_t1; L.pt.pstart = pt_ps; L.pt.slot0_of = pt_s0; L.pt.slot1_of = pt_s1;
        L.pt.feat_of = pt_fo; L.pt.feat0_of = pt_f0;
        GkGroup *gk_grp; uint32_t *gk_tab, *gk_ids;
        if ((rc = dev_upload(c, c->ix.gk_groups.data(), c->ix.gk_groups.size(), &gk_grp, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.gk_tab.data(), c->ix.gk_tab.size(), &gk_tab, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.gk_ids.data(), c->ix.gk_ids.size(), &gk_ids, c->lib_allocs))) return rc;
        L.gk.n_groups = c->ix.n_features ? (uint32_t)c->ix.gk_groups.size() : 0u; L.gk.grp = gk_grp; L.gk.tab = gk_tab; L.gk.ids = gk_ids;
        unsigned long long *gk_fw; uint32_t *gk_fwoff;
        if ((rc = dev_upload(c, c->ix.gk_fw.data(), c->ix.gk_fw.size(), &gk_fw, c->lib_allocs))) return rc;
        if ((rc = dev_upload(c, c->ix.gk_fwoff.data(), c->ix.gk_fwoff.size(), &gk_fwoff, c->lib_allocs))) return rc;
        L.gk.fw = gk_fw; L.gk.fwoff = gk_fwoff;
    }
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
    { const char *tr = getenv("F2Q_TRACE"); c->trace = tr && tr[0] == '1'; }
    { const char *tr = getenv("F2Q_TRACE_SYNC"); c->trace_sync = tr && tr[0] == '1'; }
    { const char *dc = getenv("F2Q_DEV_CACHE_MB"); if (dc && atol(dc) >= 0) c->dev_idle_cap = (size_t)atol(dc) << 20; }
    { const char *fv = getenv("F2Q_GENERIC"); c->force_generic = fv && fv[0] == '1'; }
    { const char *fv = getenv("F2Q_NO_LT"); c->no_lt = fv && fv[0] == '1'; }
    { const char *fv = getenv("F2Q_NO_HOT"); c->no_hot = fv && fv[0] == '1'; }
    { const char *fv = getenv("F2Q_NO_PT"); c->no_pt = fv && fv[0] == '1'; }
    { const char *fv = getenv("F2Q_PT_PARTS"); if (fv && atoi(fv) > 0) c->ix.pt_force_parts = std::min(atoi(fv), (int)F2Q_PT_MAXP); }
    { const char *fv = getenv("F2Q_PT_CHUNK"); if (fv && atol(fv) > 0) c->pt_chunk_reads = (uint64_t)atol(fv); }
    { const char *fv = getenv("F2Q_PT_MIN_READS"); if (fv && atol(fv) >= 0) c->pt_min_reads = (uint64_t)atol(fv); }
    { const char *fv = getenv("F2Q_HOT_LEARN"); if (fv && atol(fv) > 0) c->hot_learn = (uint64_t)atol(fv); }
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
    if (c->trace) fprintf(stderr, "[f2q trace] device memory: %llu hipMalloc %.1f ms, %llu hipFree %.1f ms, %llu reused; Extract+Count reserve %.1f ms (%llu rehashes)\n",
                          (unsigned long long)c->n_malloc, c->tr_malloc, (unsigned long long)c->n_hipfree, c->tr_hipfree, (unsigned long long)c->n_reuse, c->tr_reserve, (unsigned long long)c->n_rehash);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    free_all(c, c->lib_allocs); free_all(c, c->ec_allocs); free_all(c, c->ec_allocs64); free_all(c, c->ec_allocs_ctr); free_all(c, c->hot_allocs);
    if (c->defer_d) (void)hipFree(c->defer_d);
    for (auto &kv : c->dev_idle) (void)hipFree(kv.second);
    c->dev_idle.clear(); c->dev_size.clear();
    if (c->acc_d) (void)hipFree(c->acc_d);
    if (c->synth_keys_d) (void)hipFree(c->synth_keys_d);
    if (c->slab_d) (void)hipFree(c->slab_d);
    if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
    if (c->hit_buf_d) (void)hipFree(c->hit_buf_d);
    if (c->pt_s.streams) (void)hipFree(c->pt_s.streams);
    if (c->pt_s.cnt) (void)hipFree(c->pt_s.cnt);
    if (c->pt_stat_d) (void)hipFree(c->pt_stat_d);
    if (c->pt_slab0_d) (void)hipFree(c->pt_slab0_d);
    if (c->pt_slab1_d) (void)hipFree(c->pt_slab1_d);
    if (c->run_d) (void)hipFree(c->run_d);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->ev_k0) (void)hipEventDestroy(c->ev_k0);
    if (c->ev_k1) (void)hipEventDestroy(c->ev_k1);
    for (hipEvent_t e : c->q_ev) (void)hipEventDestroy(e);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->ev_aux0) (void)hipEventDestroy(c->ev_aux0);
    if (c->ev_aux1) (void)hipEventDestroy(c->ev_aux1);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int f2q_set_features(f2q_ctx *c, const char *seqs, const uint32_t *offs, uint32_t n)
{
    if (!c || !offs || (!seqs && n)) return fail(c, F2Q_EINVAL, "null argument");
    if (c->prm.mode != 0) return fail(c, F2Q_ESTATE, "Extract+Count mode takes no feature library (fast2q.py:1701)");
    HIPC(c, hipSetDevice(c->device));
    for (uint32_t i = 0; i < n; i++) if (offs[i + 1] < offs[i]) return fail(c, F2Q_EINVAL, "offsets must be non-decreasing");
    c->plan = make_plan(c->run_h);                   // the library decides below whether the packed paths apply
    if (c->force_general) { c->plan.fast_fixed = false; c->plan.fast_anchor = false; c->plan.multi = false; c->plan.multi_pair = false; }
    int packed_len = c->plan.fast_fixed ? c->run_h.length : 0;
    if (c->plan.fast_anchor) {
        if (c->run_h.has_up && c->run_h.has_down) {          // variable windows: index the most common feature length
            std::vector<uint32_t> hist(F2Q_REG_MAXLEN + 1, 0);
            for (uint32_t i = 0; i < n; i++) { uint32_t l = offs[i + 1] - offs[i]; if (l >= 1 && l <= F2Q_REG_MAXLEN) hist[l]++; }
            packed_len = (int)(std::max_element(hist.begin(), hist.end()) - hist.begin());
        } else packed_len = c->run_h.length;
    }
    build_index(c->ix, seqs ? seqs : "", offs, n, c->run_h.miss, packed_len, c->plan.multi ? c->run_h.n_iter : 0);
    // multi-window runs stay on the packed path only when every reachable feature is a k-part feature
    if (c->plan.multi && (!c->ix.mw_ok || c->ix.n_irregular)) { c->plan.multi = false; c->plan.fast_fixed = false; }
    int rc = upload_lib(c);
    if (rc) return rc;
    rc = alloc_acc(c, n);
    if (rc) return rc;
    c->plan.inband_n = (c->plan.fast_fixed || c->plan.fast_anchor) && c->ix.n_irregular == 0;
    // two pairs against a pure A:B library (pair tables): an 'N' travels as a flag bit (a forced mismatch; it equals no
    // symbol of any feature), every other odd symbol still sends the read to the byte-exact routine
    if (c->plan.fast_anchor && c->plan.multi_pair && c->prm.mode == 0 && c->run_h.n_iter == 2 && c->lib_h.pw.ok) { c->plan.inband_n = true; c->plan.n_only = true; }
    if (c->ix.n_irregular && !c->plan.multi_pair) c->plan.fast_anchor = false;      // irregular features need the byte-exact routine (the pair kernel matches strings)
    c->have_lib = true;
    return F2Q_OK;
}

extern "C" int f2q_reset_counts(f2q_ctx *c)
{
    if (!c) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemsetAsync(c->acc_d, 0, c->acc_n * sizeof(unsigned long long), c->stream));
    if (c->prm.mode == 1) {
        if (c->aux_stream) HIPC(c, hipStreamSynchronize(c->aux_stream));
        c->aux_busy = false;
        free_all(c, c->ec_allocs); free_all(c, c->ec_allocs64); free_all(c, c->ec_allocs_ctr);
        memset(&c->ec, 0, sizeof c->ec); c->ec_slots = 0; c->hot_valid = false; c->ec_learned = 0;
    }
    c->reads_seen = 0;
    // no host synchronisation: the clear is ordered on the context's stream like every launch and read-back after it
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

// F2Q_TRACE_SYNC=1: name every step of the Extract+Count path on stderr and wait for it (fault localisation only)
#define EC_POINT(c, what) do { if ((c)->trace_sync) { (void)hipStreamSynchronize((c)->stream); fprintf(stderr, "[f2q sync] %s done\n", what); fflush(stderr); } } while (0)

// ---- Extract+Count table management -----------------------------------------------------------
// the two sides of the Extract+Count tables grow independently: the single-word table (k64_*: one slot per plain ACGT
// key of <= 29 bases) and the byte-string table (slots, entry arrays, arena); the four counters outlive both
static int ec_alloc64(f2q_ctx *c, EcDev &e, std::vector<void *> &owner, uint64_t max_keys64)
{
    // load factor <= 0.75: smaller tables are cheaper to clear and stay in the Infinity Cache longer
    uint64_t slots64 = 1024;
    while (3 * slots64 < 4 * max_keys64) slots64 <<= 1;
    if (slots64 > (1ull << 32)) return fail(c, F2Q_ENOMEM, "Extract+Count table would exceed 2^32 slots");
    int rc;
    if ((rc = dev_alloc(c, slots64, &e.k64_slots, owner, 0xFF))) return rc;
    if ((rc = dev_alloc(c, slots64, &e.k64_count, owner, 0))) return rc;
    if ((rc = dev_alloc(c, slots64, &e.k64_first, owner, 0xFF))) return rc;
    e.k64_mask = (uint32_t)(slots64 - 1);
    e.k64_room = (uint32_t)std::min<uint64_t>(3 * slots64 / 4, 0xFFFFFFFEull);   // keys the table takes before it must grow
    return F2Q_OK;
}
static int ec_allocB(f2q_ctx *c, EcDev &e, std::vector<void *> &owner, uint64_t max_entries, uint64_t arena_words)
{
    uint64_t slots = 1024;
    while (3 * slots < 4 * max_entries) slots <<= 1;
    if (slots > (1ull << 32)) return fail(c, F2Q_ENOMEM, "Extract+Count table would exceed 2^32 slots");
    int rc;
    if ((rc = dev_alloc(c, slots, &e.slots, owner, 0))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_off, owner))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_len, owner))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_count, owner, 0))) return rc;
    if ((rc = dev_alloc(c, max_entries, &e.ent_first, owner, 0xFF))) return rc;
    if ((rc = dev_alloc(c, arena_words, &e.arena, owner))) return rc;
    e.mask = (uint32_t)(slots - 1); e.max_entries = (uint32_t)std::min<uint64_t>(max_entries, 0xFFFFFFFEull);
    e.arena_words = arena_words;
    return F2Q_OK;
}

// make room for `keys64` more keys in the single-word table and `reads` more entries of at most `key_bytes` bytes in
// the byte-string table
// known: the counters as the caller has just read them (nothing has run since), else they are fetched
static int ec_reserve(f2q_ctx *c, uint64_t keys64, uint64_t reads, uint64_t key_bytes, const unsigned long long *known = nullptr)
{
    unsigned long long ctr[4] = {0, 0, 0, 0};
    if (known) memcpy(ctr, known, sizeof ctr);
    else if (c->ec.ctr) {
        HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error, code " + std::to_string(ctr[2]) + ")");
    }
    // the byte-string table is limited by its entry arrays and its arena, the single-word table by its load factor (<= 3/4)
    const uint64_t need_b = ctr[0] + reads + 16, need_r = ctr[3] + keys64 + 16, need_w = ctr[1] + (key_bytes + 3) / 4 + reads + 16;
    const bool grow64 = !c->ec.k64_slots || need_r > c->ec.k64_room;
    const bool growB = !c->ec.slots || need_b > c->ec.max_entries || need_w > c->ec.arena_words;
    if (!grow64 && !growB) return F2Q_OK;
    const double rs0 = now_ms();
    // nothing may still be counting into the tables that are about to move
    if (c->aux_busy) { HIPC(c, hipStreamSynchronize(c->aux_stream)); c->aux_busy = false; }
    if (c->ec.ctr) {                                             // the exact counters: they say how much is carried over
        HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error, code " + std::to_string(ctr[2]) + ")");
    }
    int rc;
    if (!c->ec.ctr && (rc = dev_alloc(c, (size_t)F2Q_CTR_WORDS, &c->ec.ctr, c->ec_allocs_ctr, 0))) return rc;
    if ((c->ec.k64_slots && grow64 && ctr[3]) || (c->ec.slots && growB && ctr[0])) {
        c->n_rehash++;
        if (c->trace && c->n_rehash <= 6)
            fprintf(stderr, "[f2q trace] Extract+Count tables grow (%s%s): keys %llu/%llu, entries needed %llu of %u, single-word keys needed %llu of %u, arena words needed %llu of %llu\n",
                    grow64 ? "single-word " : "", growB ? "byte-string" : "", ctr[0], ctr[3], (unsigned long long)need_b, c->ec.max_entries,
                    (unsigned long long)need_r, c->ec.k64_room, (unsigned long long)need_w, c->ec.arena_words);
    }
    // growth doubles the room of the keys already there (amortised rehash), not the head room of one launch
    if (grow64) {
        const uint64_t nr = std::max<uint64_t>(need_r + std::max<uint64_t>(ctr[3], keys64 / 4), 1u << 16);
        EcDev fresh = c->ec; std::vector<void *> owner;
        if ((rc = ec_alloc64(c, fresh, owner, nr))) { free_all(c, owner); return rc; }
        if (c->ec.k64_slots && ctr[3]) {
            hipLaunchKernelGGL(k_ec64_rehash, dim3((unsigned)(((uint64_t)c->ec.k64_mask + 256) / 256)), dim3(256), 0, c->stream, c->ec, fresh);
            HIPC(c, hipGetLastError());
            EC_POINT(c, "reserve: k_ec64_rehash");
            if (c->hot_valid) {                                  // the hot keys' slots moved with the table
                hipLaunchKernelGGL(k_ec_hot_relink, dim3(F2Q_HOT_SLOTS / 256), dim3(256), 0, c->stream, fresh, c->hot);
                HIPC(c, hipGetLastError());
                EC_POINT(c, "reserve: k_ec_hot_relink");
            }
        }
        free_all(c, c->ec_allocs64);                             // (waits for the stream)
        c->ec_allocs64 = owner;
        c->ec.k64_slots = fresh.k64_slots; c->ec.k64_count = fresh.k64_count; c->ec.k64_first = fresh.k64_first;
        c->ec.k64_mask = fresh.k64_mask; c->ec.k64_room = fresh.k64_room;
    }
    if (growB) {
        const uint64_t nb = need_b > c->ec.max_entries || !c->ec.slots ? std::max<uint64_t>(need_b + std::max<uint64_t>(ctr[0], reads / 4), 1u << 12)
                                                                        : (uint64_t)c->ec.max_entries;
        const uint64_t nw = need_w > c->ec.arena_words || !c->ec.slots ? std::max<uint64_t>(need_w + std::max<uint64_t>(ctr[1], key_bytes / 16), 1u << 14)
                                                                       : c->ec.arena_words;
        EcDev fresh = c->ec; std::vector<void *> owner;
        if ((rc = ec_allocB(c, fresh, owner, nb, nw))) { free_all(c, owner); return rc; }
        if (c->ec.slots && ctr[0]) {
            HIPC(c, hipMemcpyAsync(fresh.arena, c->ec.arena, ctr[1] * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPC(c, hipMemcpyAsync(fresh.ent_off, c->ec.ent_off, ctr[0] * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPC(c, hipMemcpyAsync(fresh.ent_len, c->ec.ent_len, ctr[0] * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPC(c, hipMemcpyAsync(fresh.ent_count, c->ec.ent_count, ctr[0] * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPC(c, hipMemcpyAsync(fresh.ent_first, c->ec.ent_first, ctr[0] * 8, hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(k_ec_rehash, dim3((unsigned)((ctr[0] + 255) / 256)), dim3(256), 0, c->stream, c->ec, ctr[0], fresh);
            HIPC(c, hipGetLastError());
            EC_POINT(c, "reserve: k_ec_rehash");
        }
        free_all(c, c->ec_allocs);
        c->ec_allocs = owner;
        c->ec.slots = fresh.slots; c->ec.mask = fresh.mask; c->ec.max_entries = fresh.max_entries;
        c->ec.ent_off = fresh.ent_off; c->ec.ent_len = fresh.ent_len; c->ec.ent_count = fresh.ent_count; c->ec.ent_first = fresh.ent_first;
        c->ec.arena = fresh.arena; c->ec.arena_words = fresh.arena_words;
    }
    c->tr_reserve += now_ms() - rs0;
    return F2Q_OK;
}

// ---- the partitioned path (f2q_part_kernels.h) -------------------------------------------------------------------
template <class T>
static int pt_grow(f2q_ctx *c, T **buf, size_t &have, size_t want, bool zero)
{
    if (want <= have) return F2Q_OK;
    HIPC(c, hipStreamSynchronize(c->stream));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; have = 0;
    HIPC(c, hipMalloc((void **)buf, want * sizeof(T)));
    if (zero) HIPC(c, hipMemsetAsync(*buf, 0, want * sizeof(T), c->stream));
    have = want;
    return F2Q_OK;
}
// fixed-offset Counter mode on a partitioned library: the block's tiles in chunks, each scattered and then counted
static int launch_part(f2q_ctx *c, const PackedBlock &pb, Accum &acc, uint32_t &launches)
{
    const PtDesc &pt = c->lib_h.pt;
    const uint32_t P = pt.n_parts, nf = c->lib_h.n_features;
    const int pw = P <= 8 ? 16 : P <= 16 ? 8 : 4;                    // scatter waves per workgroup: n_parts KiB of rings each
    // rounds of equal size, none above pt_chunk_reads (the scratch memory of a round is 8 bytes x partitions x its reads)
    // ... and that memory is kept to 16 GiB (a stream must be able to take every entry of its workgroup)
    const uint64_t chunk_reads = std::min<uint64_t>(c->pt_chunk_reads, ((uint64_t)16 << 30) / (8ull * P));
    const uint32_t max_tiles = (uint32_t)std::max<uint64_t>(chunk_reads / F2Q_TILE, 1);
    const uint32_t n_rounds = std::max<uint32_t>(1u, (pb.n_tiles + max_tiles - 1) / max_tiles);
    const uint32_t chunk_tiles = std::max<uint32_t>(1u, (pb.n_tiles + n_rounds - 1) / n_rounds);
    const uint32_t grid1 = std::min<uint32_t>((uint32_t)c->n_cu, (chunk_tiles + pw - 1) / pw);
    const uint32_t tiles_per_wg = (chunk_tiles + grid1 - 1) / grid1 + (uint32_t)pw;          // (waves take tiles round robin)
    // entries per stream: what a workgroup can produce, in whole steps of the count pass, plus 11 x 256 bytes: the streams
    // grow at the same pace, and at a power-of-two distance their write positions would share memory channels
    const uint32_t cap = ((tiles_per_wg * F2Q_TILE + F2Q_PC_STEP - 1) / F2Q_PC_STEP) * F2Q_PC_STEP + F2Q_PC_STEP + 352u;
    const uint32_t K = std::max<uint32_t>(1u, (uint32_t)c->n_cu / P), grid2 = K * P, cw = F2Q_PC_THREADS / 64;
    if ((grid1 + K - 1) / K > 64) return fail(c, F2Q_EUNSUPPORTED, "partitioned path: more than 64 streams per counting workgroup");
    const uint32_t n_slots1 = 2u << pt.bb1, rows = grid1 + grid2;
    int rc;
    if ((rc = pt_grow(c, &c->pt_s.streams, c->pt_streams_n, (size_t)grid1 * P * cap, false))) return rc;
    if ((rc = pt_grow(c, &c->pt_s.cnt, c->pt_cnt_n, (size_t)grid1 * P, false))) return rc;
    if ((rc = pt_grow(c, &c->pt_slab0_d, c->pt_slab0_n, (size_t)grid2 * std::max<uint32_t>(pt.max_part, 1u), true))) return rc;
    if ((rc = pt_grow(c, &c->pt_slab1_d, c->pt_slab1_n, (size_t)n_slots1, true))) return rc;
    if ((rc = pt_grow(c, &c->pt_stat_d, c->pt_stat_n, (size_t)rows * 8, true))) return rc;     // rows accumulate; k_part_reduce clears them
    PartScratch ps = c->pt_s;
    ps.cap = cap; ps.n_wg1 = grid1;
    Accum a1 = acc, a2 = acc;
    a1.stat_slab = c->pt_stat_d; a2.stat_slab = a1.stat_slab + (size_t)grid1 * 8;
    const FixedGeom fg = fixed_geom(c->run_h);
    const bool spec52 = !c->force_generic && fg.nq == 5 && fg.nb == 2 && c->run_h.thr >= 33;
    const bool a20 = spec52 && fg.L == 20 && (fg.st & 15) == 0;
    const bool near = c->run_h.miss > 0;
    const size_t shmem1 = (size_t)pw * P * F2Q_PS_RING * 8 + (size_t)(pw + 1) * P * 4;
    const size_t shmem2 = ((size_t)F2Q_LT_SLOTS + F2Q_LT_BUCKETS) * 4 + (near ? (size_t)cw * (F2Q_PC_RING * 8 + F2Q_PC_VIA * 4) : 0);
    auto kern2 = near ? k_part_count<true> : k_part_count<false>;
    (void)hipFuncSetAttribute((const void *)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2);
    for (uint32_t t0 = 0; t0 < pb.n_tiles; t0 += chunk_tiles) {
        const uint32_t t1 = std::min<uint32_t>(pb.n_tiles, t0 + chunk_tiles);
#define F2Q_LAUNCH_PS(PW_)                                                                                             \
        do {                                                                                                           \
            auto k1 = a20 ? k_part_scatter<5, 2, true, PW_> : spec52 ? k_part_scatter<5, 2, false, PW_> : k_part_scatter<0, 0, false, PW_>; \
            (void)hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem1);      \
            hipLaunchKernelGGL(k1, dim3(grid1), dim3(64 * PW_), shmem1, c->stream, c->run_d, c->lib_d, pb, a1, ps, t0, t1); \
        } while (0)
        if (pw == 16) F2Q_LAUNCH_PS(16); else if (pw == 8) F2Q_LAUNCH_PS(8); else F2Q_LAUNCH_PS(4);
#undef F2Q_LAUNCH_PS
        HIPC(c, hipGetLastError());
        hipLaunchKernelGGL(kern2, dim3(grid2), dim3(64 * cw), shmem2, c->stream, c->run_d, c->lib_d, a2, ps, c->pt_slab0_d, c->pt_slab1_d);
        HIPC(c, hipGetLastError());
        launches += 2;
    }
    hipLaunchKernelGGL(k_part_reduce, dim3(std::max<uint32_t>(1u, (nf + 63) / 64)), dim3(256), 0, c->stream, c->lib_d, c->pt_slab0_d, K,
                       near ? c->pt_slab1_d : (uint32_t *)nullptr, acc.counts, c->pt_stat_d, rows, acc.stats);
    HIPC(c, hipGetLastError());
    launches++;
    return F2Q_OK;
}

// ---- launching ----------------------------------------------------------------------------------
// one set of launches over a view of a block (all of it in Counter mode, a step of it in Extract+Count mode)
static int launch_view(f2q_ctx *c, const PackedBlock &pb, const RawBlock &rbv, Accum &acc, uint32_t &launches)
{
    if (pb.n_tiles) c->last_path = F2Q_PATH_NONE;
    if (pb.n_tiles && pb.planar_nw && c->plan.multi_pair) {
        // several --us/--ds pairs on the planes
        c->last_path = F2Q_PATH_PAIRS;
        const bool lds = c->prm.mode == 0 && c->lib_h.n_features <= F2Q_HIST_MAX;
        const uint32_t grid = std::min<uint32_t>(pb.n_tiles, (uint32_t)c->n_cu * 8u);          // (four workgroups of 256 threads are resident per CU at 128 VGPRs: two rounds)
        const size_t shmem = lds ? std::max<size_t>(4, (((size_t)c->lib_h.n_features + 1) / 2) * 4) : 4;
        const int nw = (int)pb.planar_nw, kb = c->plan.kb;
        const bool sameq = c->run_h.thr_up == c->run_h.thr && c->run_h.thr_down == c->run_h.thr;
#define F2Q_LAUNCH_MP2(NW_, KB_, SQ_)                                                                                  \
        do {                                                                                                           \
            if (lds) hipLaunchKernelGGL((k_count_anchor_pairs<NW_, KB_, SQ_, true>), dim3(grid), dim3(F2Q_AN_THREADS), shmem, \
                                        c->stream, c->run_d, c->lib_d, c->ec, pb, acc, c->reads_seen);                 \
            else hipLaunchKernelGGL((k_count_anchor_pairs<NW_, KB_, SQ_, false>), dim3(grid), dim3(F2Q_AN_THREADS), shmem, \
                                    c->stream, c->run_d, c->lib_d, c->ec, pb, acc, c->reads_seen);                     \
        } while (0)
#define F2Q_LAUNCH_MP(NW_, KB_) do { if (sameq) F2Q_LAUNCH_MP2(NW_, KB_, true); else F2Q_LAUNCH_MP2(NW_, KB_, false); } while (0)
        if (nw == 10) F2Q_LAUNCH_MP2(10, 3, false);      // reads of 161 .. 320 bases: the general instantiation (any --msu/--msd <= 7, any --qsu/--qsd)
        else if (nw == 3 && kb == 0) F2Q_LAUNCH_MP(3, 0);
        else if (nw == 3 && kb == 1) F2Q_LAUNCH_MP(3, 1);
        else if (nw == 3) F2Q_LAUNCH_MP(3, 3);
        else if (kb == 0) F2Q_LAUNCH_MP(5, 0);
        else if (kb == 1) F2Q_LAUNCH_MP(5, 1);
        else F2Q_LAUNCH_MP(5, 3);
#undef F2Q_LAUNCH_MP
#undef F2Q_LAUNCH_MP2
        HIPC(c, hipGetLastError());
        launches++;
    } else if (pb.n_tiles && pb.planar_nw) {
        // packed anchored path
        const bool ecm = c->prm.mode == 1;
        const bool lds = !ecm && c->lib_h.n_features <= F2Q_HIST_MAX;
        c->last_path = ecm ? F2Q_PATH_EXTRACT : F2Q_PATH_ANCHOR;
        uint32_t an_mult = 4u; { const char *e = getenv("F2Q_AN_GRID"); if (e && atoi(e) > 0) an_mult = (uint32_t)atoi(e); }
        const uint32_t grid = std::min<uint32_t>(pb.n_tiles, (uint32_t)c->n_cu * an_mult);
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
        const int nw = (int)pb.planar_nw, kb = c->plan.kb;
        const bool sameq = c->run_h.thr_up == c->run_h.thr && c->run_h.thr_down == c->run_h.thr;
        // the library in LDS (Counter mode, uniform 14..21-base library, --m <= 1, default --qsu/--qsd)
        if (!ecm && lds && sameq && !c->no_lt && c->lib_h.lt.ok && c->run_h.miss <= 1 && pb.len != nullptr && nw != 10) {
            c->last_path = F2Q_PATH_ANCHOR_LDS;
            const uint32_t groups = (pb.n_tiles + F2Q_ALT_GROUPS - 1) / F2Q_ALT_GROUPS;
            const uint32_t lgrid = std::min<uint32_t>(groups, (uint32_t)c->n_cu);
            const bool near = c->run_h.miss > 0;
            const size_t lshmem = ((near ? 2u : 1u) * (size_t)F2Q_LT_SLOTS + F2Q_LT_BUCKETS) * 4;
            const uint32_t nf_ = c->lib_h.n_features;
            const size_t need = (size_t)lgrid * nf_;
            if (need > c->slab_n || (size_t)lgrid > c->stat_slab_n) {
                if (c->slab_d) (void)hipFree(c->slab_d);
                if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
                c->slab_d = nullptr; c->slab_n = 0; c->stat_slab_d = nullptr; c->stat_slab_n = 0;
                HIPC(c, hipMalloc((void **)&c->slab_d, std::max<size_t>(need, 1) * sizeof(uint32_t)));
                HIPC(c, hipMalloc((void **)&c->stat_slab_d, (size_t)lgrid * 8 * sizeof(unsigned long long)));
                c->slab_n = need; c->stat_slab_n = lgrid;
            }
            acc.slab = c->slab_d; acc.stat_slab = c->stat_slab_d;
#define F2Q_LAUNCH_ALT(NW_, KB_)                                                                                       \
            do {                                                                                                       \
                auto kern = near ? k_count_anchor_lt<NW_, KB_, true, true> : k_count_anchor_lt<NW_, KB_, true, false>;  \
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lshmem);\
                hipLaunchKernelGGL(kern, dim3(lgrid), dim3(F2Q_ALT_THREADS), lshmem, c->stream, c->run_d, c->lib_d, pb, acc); \
            } while (0)
            if (nw == 3 && kb == 0) F2Q_LAUNCH_ALT(3, 0);
            else if (nw == 3 && kb == 1) F2Q_LAUNCH_ALT(3, 1);
            else if (nw == 3) F2Q_LAUNCH_ALT(3, 3);
            else if (kb == 0) F2Q_LAUNCH_ALT(5, 0);
            else if (kb == 1) F2Q_LAUNCH_ALT(5, 1);
            else F2Q_LAUNCH_ALT(5, 3);
#undef F2Q_LAUNCH_ALT
            HIPC(c, hipGetLastError());
            hipLaunchKernelGGL(k_reduce_slabs, dim3((nf_ + 63) / 64, F2Q_RED_SPLIT), dim3(256), 0, c->stream,
                               c->slab_d, lgrid, nf_, acc.counts, c->stat_slab_d, lgrid, acc.stats);
            HIPC(c, hipGetLastError());
            launches += 2;
        } else {
#define F2Q_LAUNCH_AN2(NW_, KB_, SQ_)                                                                                \
        do {                                                                                                         \
            if (ecm) hipLaunchKernelGGL((k_count_anchor<NW_, KB_, true, false, SQ_>), dim3(grid), dim3(F2Q_AN_THREADS),   \
                                        shmem, c->stream, c->run_d, c->lib_d, c->ec, pb, acc, c->reads_seen);     \
            else if (lds) hipLaunchKernelGGL((k_count_anchor<NW_, KB_, false, true, SQ_>), dim3(grid),               \
                                             dim3(F2Q_AN_THREADS), shmem, c->stream, c->run_d, c->lib_d, c->ec,      \
                                             pb, acc, c->reads_seen);                                             \
            else hipLaunchKernelGGL((k_count_anchor<NW_, KB_, false, false, SQ_>), dim3(grid), dim3(F2Q_AN_THREADS),  \
                                    shmem, c->stream, c->run_d, c->lib_d, c->ec, pb, acc, c->reads_seen);         \
        } while (0)
#define F2Q_LAUNCH_AN(NW_, KB_) do { if (sameq) F2Q_LAUNCH_AN2(NW_, KB_, true); else F2Q_LAUNCH_AN2(NW_, KB_, false); } while (0)
        if (nw == 10) F2Q_LAUNCH_AN2(10, 3, false);      // reads of 161 .. 320 bases: the general instantiation
        else if (nw == 3 && kb == 0) F2Q_LAUNCH_AN(3, 0);
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
        }
    } else if (pb.n_tiles && c->prm.mode == 1) {
        const uint32_t wgs = (pb.n_tiles + F2Q_V2_WAVES - 1) / F2Q_V2_WAVES;
        const uint32_t grid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu * 4u);
        c->last_path = F2Q_PATH_EXTRACT;
        hipLaunchKernelGGL(k_extract_fixed4, dim3(grid), dim3(F2Q_V2_THREADS), 0, c->stream, c->run_d, c->ec, pb, acc, c->reads_seen);
        HIPC(c, hipGetLastError());
        launches++;
    } else if (pb.n_tiles) {
        const bool lds = c->lib_h.n_features <= F2Q_HIST_MAX;
        const bool v2 = !c->force_v1 && c->lib_h.pk.len == (uint32_t)c->run_h.length && c->lib_h.pk.len > 0 &&
                        c->lib_h.n_irregular == 0;
        const int mw_total = c->run_h.n_iter * c->run_h.length;
        const bool mw_lt = c->plan.multi && lds && !c->no_lt && c->lib_h.lt.ok && c->lib_h.lt.len == (uint32_t)mw_total && c->run_h.miss <= 1 &&
                           pb.len != nullptr && (uint32_t)((mw_total + 3) / 4) <= pb.wq && (uint32_t)((mw_total + 15) / 16) <= pb.wb;
        if (mw_lt) {
            // several windows per read, every feature with one part per window: the joined keys on the library-in-LDS kernel
            // (the tiles hold the windows back to back: one window of n_iter * length bases, Phred rule per part)
            c->last_path = F2Q_PATH_MULTI_LDS;
            const uint32_t wgs = (pb.n_tiles + F2Q_LT_WAVES - 1) / F2Q_LT_WAVES;
            const uint32_t grid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu);
            const bool near = c->run_h.miss > 0;
            const size_t shmem = ((near ? 2u : 1u) * (size_t)F2Q_LT_SLOTS + F2Q_LT_BUCKETS) * 4;
            const FixedGeom fg = fixed_geom_at(0, mw_total, c->run_h.thr);
            const bool spec52 = !c->force_generic && fg.nq == 5 && fg.nb == 2 && c->run_h.thr >= 33;
            const bool a20 = spec52 && mw_total == 20;              // (two 10-base or four 5-base windows: the usual 20-base geometry)
            auto kern = near ? (a20 ? k_count_fixed4_lds<5, 2, true, true, true> : spec52 ? k_count_fixed4_lds<5, 2, true, false, true> : k_count_fixed4_lds<0, 0, true, false, true>)
                             : (a20 ? k_count_fixed4_lds<5, 2, false, true, true> : spec52 ? k_count_fixed4_lds<5, 2, false, false, true> : k_count_fixed4_lds<0, 0, false, false, true>);
            const uint32_t nf_ = c->lib_h.n_features;
            const size_t need = (size_t)grid * nf_;
            if (need > c->slab_n || (size_t)grid > c->stat_slab_n) {
                if (c->slab_d) (void)hipFree(c->slab_d);
                if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
                c->slab_d = nullptr; c->slab_n = 0; c->stat_slab_d = nullptr; c->stat_slab_n = 0;
                HIPC(c, hipMalloc((void **)&c->slab_d, std::max<size_t>(need, 1) * sizeof(uint32_t)));
                HIPC(c, hipMalloc((void **)&c->stat_slab_d, (size_t)grid * 8 * sizeof(unsigned long long)));
                c->slab_n = need; c->stat_slab_n = grid;
            }
            acc.slab = c->slab_d; acc.stat_slab = c->stat_slab_d;
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(F2Q_LT_THREADS), shmem, c->stream, c->run_d, c->lib_d, pb, acc);
            HIPC(c, hipGetLastError());
            hipLaunchKernelGGL(k_reduce_slabs, dim3((nf_ + 63) / 64, F2Q_RED_SPLIT), dim3(256), 0, c->stream,
                               c->slab_d, grid, nf_, acc.counts, c->stat_slab_d, grid, acc.stats);
            launches++;
        } else if (c->plan.multi) {
            // several windows per read (--st a,b,...): k-part keys against the k-part features
            c->last_path = F2Q_PATH_MULTI;
            const uint32_t wgs = (pb.n_tiles + F2Q_V2_WAVES - 1) / F2Q_V2_WAVES;
            const uint32_t grid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu * 2u);
            const size_t shmem = (size_t)F2Q_V2_WAVES * F2Q_V2_QCAP * 12 + (lds ? (size_t)c->lib_h.n_features * 4 : 0);
            const uint32_t nf_ = c->lib_h.n_features;
            const size_t need = lds ? (size_t)grid * nf_ : 0;
            if (need > c->slab_n || (size_t)grid > c->stat_slab_n) {
                if (c->slab_d) (void)hipFree(c->slab_d);
                if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
                c->slab_d = nullptr; c->slab_n = 0; c->stat_slab_d = nullptr; c->stat_slab_n = 0;
                HIPC(c, hipMalloc((void **)&c->slab_d, std::max<size_t>(need, 1) * sizeof(uint32_t)));
                HIPC(c, hipMalloc((void **)&c->stat_slab_d, (size_t)grid * 8 * sizeof(unsigned long long)));
                c->slab_n = need; c->stat_slab_n = grid;
            }
            acc.slab = c->slab_d; acc.stat_slab = c->stat_slab_d;
            if (lds) hipLaunchKernelGGL(k_count_multi4<true>, dim3(grid), dim3(F2Q_V2_THREADS), shmem, c->stream, c->run_d, c->lib_d, pb, acc, c->plan.need);
            else hipLaunchKernelGGL(k_count_multi4<false>, dim3(grid), dim3(F2Q_V2_THREADS), shmem, c->stream, c->run_d, c->lib_d, pb, acc, c->plan.need);
            HIPC(c, hipGetLastError());
            hipLaunchKernelGGL(k_reduce_slabs, dim3(std::max<uint32_t>(1u, (nf_ + 63) / 64), F2Q_RED_SPLIT), dim3(256), 0, c->stream,
                               c->slab_d, lds ? grid : 0u, nf_, acc.counts, c->stat_slab_d, grid, acc.stats);
            launches++;
        } else {
        const FixedGeom fgeo = fixed_geom(c->run_h);
        // the library in LDS: uniform 14..21-base library, --m <= 1, and tiles that hold every row under the window
        const bool use_lt = v2 && lds && !c->no_lt && c->lib_h.lt.ok && c->lib_h.lt.len == (uint32_t)c->run_h.length &&
                            c->run_h.miss <= 1 && (uint32_t)(fgeo.qw0 + fgeo.nq) <= pb.wq && (uint32_t)(fgeo.bw0 + fgeo.nb) <= pb.wb &&
                            pb.len != nullptr;
        // a library beyond one workgroup's LDS, dealt into partitions (or F2Q_PT_PARTS set: that path whatever the size)
        const bool use_pt = v2 && !c->no_pt && c->lib_h.pt.ok && c->lib_h.pt.len == (uint32_t)c->run_h.length && c->run_h.miss <= 1 &&
                            (uint32_t)(fgeo.qw0 + fgeo.nq) <= pb.wq && (uint32_t)(fgeo.bw0 + fgeo.nb) <= pb.wb && pb.len != nullptr &&
                            (c->ix.pt_force_parts > 0 || (!use_lt && (uint64_t)pb.n_tiles * F2Q_TILE >= c->pt_min_reads));
        if (use_pt) {
            c->last_path = F2Q_PATH_FIXED_PART;
            int prc = launch_part(c, pb, acc, launches);
            if (prc) return prc;
            launches--;                                  // (the common tail below counts one)
        } else if (use_lt) {
            c->last_path = F2Q_PATH_FIXED_LDS;
            const uint32_t wgs = (pb.n_tiles + F2Q_LT_WAVES - 1) / F2Q_LT_WAVES;
            const uint32_t grid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu);
            const bool near = c->run_h.miss > 0;
            const size_t shmem = ((near ? 2u : 1u) * (size_t)F2Q_LT_SLOTS + F2Q_LT_BUCKETS) * 4;
            const bool spec52 = !c->force_generic && fgeo.nq == 5 && fgeo.nb == 2 && c->run_h.thr >= 33;
            const bool a20 = spec52 && fgeo.L == 20 && (fgeo.st & 15) == 0;
            auto kern = near ? (a20 ? k_count_fixed4_lds<5, 2, true, true> : spec52 ? k_count_fixed4_lds<5, 2, true, false> : k_count_fixed4_lds<0, 0, true, false>)
                             : (a20 ? k_count_fixed4_lds<5, 2, false, true> : spec52 ? k_count_fixed4_lds<5, 2, false, false> : k_count_fixed4_lds<0, 0, false, false>);
            const uint32_t nf_ = c->lib_h.n_features;
            const size_t need = (size_t)grid * nf_;
            if (need > c->slab_n || (size_t)grid > c->stat_slab_n) {
                if (c->slab_d) (void)hipFree(c->slab_d);
                if (c->stat_slab_d) (void)hipFree(c->stat_slab_d);
                c->slab_d = nullptr; c->slab_n = 0; c->stat_slab_d = nullptr; c->stat_slab_n = 0;
                HIPC(c, hipMalloc((void **)&c->slab_d, std::max<size_t>(need, 1) * sizeof(uint32_t)));
                HIPC(c, hipMalloc((void **)&c->stat_slab_d, (size_t)grid * 8 * sizeof(unsigned long long)));
                c->slab_n = need; c->stat_slab_n = grid;
            }
            acc.slab = c->slab_d; acc.stat_slab = c->stat_slab_d;
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(F2Q_LT_THREADS), shmem, c->stream, c->run_d, c->lib_d, pb, acc);
            HIPC(c, hipGetLastError());
            hipLaunchKernelGGL(k_reduce_slabs, dim3((nf_ + 63) / 64, F2Q_RED_SPLIT), dim3(256), 0, c->stream,
                               c->slab_d, grid, nf_, acc.counts, c->stat_slab_d, grid, acc.stats);
            launches++;
        } else if (v2) {
            c->last_path = F2Q_PATH_FIXED_PACKED;
            const uint32_t wgs = (pb.n_tiles + F2Q_V2_WAVES - 1) / F2Q_V2_WAVES;
            uint32_t v2_mult = 2u; { const char *e = getenv("F2Q_V2_GRID"); if (e && atoi(e) > 0) v2_mult = (uint32_t)atoi(e); }
            const uint32_t grid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu * v2_mult);
            const size_t shmem = (size_t)F2Q_V2_WAVES * F2Q_V2_QCAP * 12 + (lds ? (size_t)c->lib_h.n_features * 4 : (size_t)F2Q_V2_WAVES * F2Q_V2_QCAP * 4);
            const FixedGeom fg = fixed_geom(c->run_h);
            const bool spec52 = !c->force_generic && fg.nq == 5 && fg.nb == 2 && c->run_h.thr >= 33;
            auto kern = lds ? (spec52 ? k_count_fixed4<true, 5, 2> : k_count_fixed4<true, 0, 0>)
                            : (spec52 ? k_count_fixed4<false, 5, 2> : k_count_fixed4<false, 0, 0>);
            const uint32_t nf_ = c->lib_h.n_features;
            const uint32_t n_ranges = lds ? 1u : (nf_ + F2Q_HIST_RANGE - 1) / F2Q_HIST_RANGE;
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
                if (pb.n_slots > c->hit_buf_n) {
                    if (c->hit_buf_d) (void)hipFree(c->hit_buf_d);
                    c->hit_buf_d = nullptr; c->hit_buf_n = 0;
                    HIPC(c, hipMalloc((void **)&c->hit_buf_d, pb.n_slots * sizeof(uint32_t)));
                    c->hit_buf_n = pb.n_slots;
                }
                acc.hit_buf = c->hit_buf_d;                          // (the kernel writes every slot: no clearing)
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(F2Q_V2_THREADS), shmem, c->stream, c->run_d, c->lib_d, pb, acc);
            HIPC(c, hipGetLastError());
            if (!lds && nf_) {
                (void)hipFuncSetAttribute((const void *)k_hist_ranges, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F2Q_HIST_RANGE * 4));
                hipLaunchKernelGGL(k_hist_ranges, dim3(n_ranges * n_parts), dim3(1024), (size_t)F2Q_HIST_RANGE * 4, c->stream,
                                   c->hit_buf_d, (uint64_t)pb.n_slots, nf_, n_parts, c->slab_d);
                HIPC(c, hipGetLastError());
                launches++;
            }
            if (nf_) {
                hipLaunchKernelGGL(k_reduce_slabs, dim3((nf_ + 63) / 64, F2Q_RED_SPLIT), dim3(256), 0, c->stream,
                                   c->slab_d, n_parts, nf_, acc.counts, c->stat_slab_d, grid, acc.stats);
                launches++;
            }
        } else {
            c->last_path = F2Q_PATH_FIXED_V1;
            const uint32_t grid = std::min<uint32_t>(pb.n_tiles, (uint32_t)c->n_cu * 8u);
            if (lds) {
                size_t shmem = std::max<size_t>(4, (size_t)c->lib_h.n_features * 4);
                hipLaunchKernelGGL(k_count_fixed<true>, dim3(grid), dim3(F2Q_TILE), shmem, c->stream, c->run_d, c->lib_d, pb, acc);
            } else {
                hipLaunchKernelGGL(k_count_fixed<false>, dim3(grid), dim3(F2Q_TILE), 0, c->stream, c->run_d, c->lib_d, pb, acc);
            }
        }
        }
        HIPC(c, hipGetLastError());
        launches++;
    }
    if (rbv.n) {
        RawBlock rb = rbv;
        rb.first_index += c->reads_seen;
        const uint64_t wg = (rb.n + F2Q_GEN_THREADS - 1) / F2Q_GEN_THREADS;
        uint32_t gmul = 64u; { const char *e = getenv("F2Q_GEN_GRID"); if (e && atoi(e) > 0) gmul = (uint32_t)atoi(e); }
        const uint32_t grid = (uint32_t)std::min<uint64_t>(wg, (uint64_t)c->n_cu * gmul);
        hipLaunchKernelGGL(k_count_general, dim3(grid), dim3(F2Q_GEN_THREADS), 0, c->stream, c->run_d, c->lib_d, c->ec, rb, acc);
        HIPC(c, hipGetLastError());
        launches++;
        EC_POINT(c, "k_count_general");
    }
    return F2Q_OK;
}

// ---- Extract+Count, anchored tiles, hot keys in LDS (EcHot, k_extract_anchor_hot) ---------------------------------
static int hot_arrays(f2q_ctx *c)
{
    if (c->hot.keys) return F2Q_OK;
    int rc;
    if ((rc = dev_alloc(c, (size_t)F2Q_HOT_SLOTS, &c->hot.keys, c->hot_allocs, 0xFF))) return rc;
    if ((rc = dev_alloc(c, (size_t)F2Q_HOT_SLOTS, &c->hot.slot, c->hot_allocs))) return rc;
    if ((rc = dev_alloc(c, (size_t)2, &c->hot.meta, c->hot_allocs, 0))) return rc;
    return dev_alloc(c, (size_t)F2Q_HOT_CAND, &c->hot.cand, c->hot_allocs);
}
// the set from the candidates the learning launches noted
static int hot_build(f2q_ctx *c)
{
    int rc = hot_arrays(c);
    if (rc) return rc;
    HIPC(c, hipMemsetAsync(c->hot.keys, 0xFF, (size_t)F2Q_HOT_SLOTS * 8, c->stream));
    HIPC(c, hipMemsetAsync(c->hot.meta, 0, 16, c->stream));
    if (c->ec.k64_slots) {
        hipLaunchKernelGGL(k_ec_hot_build, dim3((F2Q_HOT_CAP + 255) / 256), dim3(256), 0, c->stream, c->ec, c->hot);
        HIPC(c, hipGetLastError());
        EC_POINT(c, "hot set: build");
    }
    c->hot_valid = true;
    return F2Q_OK;
}

// the raw records of a block on the second stream (after ev_aux0: tables reserved, block resident)
static int launch_aux_general(f2q_ctx *c, const f2q_block *b, Accum &acc, uint32_t &launches)
{
    HIPC(c, hipStreamWaitEvent(c->aux_stream, c->ev_aux0, 0));
    RawBlock rb = b->rb;
    rb.first_index += c->reads_seen;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((rb.n + F2Q_GEN_THREADS - 1) / F2Q_GEN_THREADS, (uint64_t)c->n_cu * 32u);
    hipLaunchKernelGGL(k_count_general, dim3(grid), dim3(F2Q_GEN_THREADS), 0, c->aux_stream, c->run_d, c->lib_d, c->ec, rb, acc);
    HIPC(c, hipGetLastError());
    launches++;
    c->aux_busy = true;
    return F2Q_OK;
}

// one launch of the hot-key kernel over a view of the block's anchored tiles (it starts slot_base slots into the
// block); the tables have room (the caller reserved).  learning: the hot set is not built yet, the kernel runs with an
// empty one and notes the keys that come up F2Q_HOT_MINCOUNT times.  ctr: the counters after the launch.
static int launch_hot(f2q_ctx *c, const PackedBlock &v, uint64_t slot_base, Accum &acc, uint32_t &launches, bool learning,
                      unsigned long long ctr[F2Q_CTR_WORDS], const f2q_block *aux_block)
{
    int rc = hot_arrays(c);
    if (rc) return rc;
    if (learning) HIPC(c, hipMemsetAsync(c->hot.keys, 0xFF, (size_t)F2Q_HOT_SLOTS * 8, c->stream));   // an empty set
    const int nw = (int)v.planar_nw, kb = c->plan.kb;
    const bool sameq = c->run_h.thr_up == c->run_h.thr && c->run_h.thr_down == c->run_h.thr;
    const size_t shmem = (size_t)F2Q_HOT_SLOTS * 12;             // key words + counters
    if (!v.planar_nw) {
        // fixed window: one wave per tile
        const uint32_t wgs = (v.n_tiles + F2Q_FH_WAVES - 1) / F2Q_FH_WAVES;
        const uint32_t fgrid = std::min<uint32_t>(wgs, (uint32_t)c->n_cu);
        auto kern = learning ? k_extract_fixed4_hot<true> : k_extract_fixed4_hot<false>;
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(kern, dim3(fgrid), dim3(F2Q_FH_THREADS), shmem, c->stream, c->run_d, c->ec, c->hot, v, acc, c->reads_seen,
                           c->defer_d, slot_base, (uint64_t)c->defer_cap);
    } else {
    const uint32_t groups = (v.n_tiles + F2Q_HOT_GROUPS - 1) / F2Q_HOT_GROUPS;
    const uint32_t grid = std::min<uint32_t>(groups, (uint32_t)c->n_cu);
#define F2Q_LAUNCH_HOT2(NW_, KB_, SQ_)                                                                                 \
    do {                                                                                                               \
        auto kern = learning ? k_extract_anchor_hot<NW_, KB_, SQ_, true> : k_extract_anchor_hot<NW_, KB_, SQ_, false>; \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(F2Q_HOT_THREADS), shmem, c->stream, c->run_d, c->ec, c->hot, v, acc, \
                           c->reads_seen, c->defer_d, slot_base, (uint64_t)c->defer_cap);                              \
    } while (0)
#define F2Q_LAUNCH_HOT(NW_, KB_) do { if (sameq) F2Q_LAUNCH_HOT2(NW_, KB_, true); else F2Q_LAUNCH_HOT2(NW_, KB_, false); } while (0)
    if (nw == 3 && kb == 0) F2Q_LAUNCH_HOT(3, 0);
    else if (nw == 3 && kb == 1) F2Q_LAUNCH_HOT(3, 1);
    else if (nw == 3) F2Q_LAUNCH_HOT(3, 3);
    else if (kb == 0) F2Q_LAUNCH_HOT(5, 0);
    else if (kb == 1) F2Q_LAUNCH_HOT(5, 1);
    else F2Q_LAUNCH_HOT(5, 3);
#undef F2Q_LAUNCH_HOT
#undef F2Q_LAUNCH_HOT2
    }
    HIPC(c, hipGetLastError());
    launches++;
    if (aux_block && (rc = launch_aux_general(c, aux_block, acc, launches))) return rc;
    EC_POINT(c, learning ? "k_extract_anchor_hot (learning)" : "k_extract_anchor_hot");
    HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, F2Q_CTR_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error, code " + std::to_string(ctr[2]) + ")");
    return F2Q_OK;
}

// the reads the launches over a block set aside (ctr: the counters as just read): windows the single-word table cannot
// hold go to the byte-string table from the planes, the rest through the byte-exact routine
static int hot_aside(f2q_ctx *c, const PackedBlock &blk, Accum &acc, uint32_t &launches, const unsigned long long ctr[F2Q_CTR_WORDS])
{
    const unsigned long long n_def = ctr[F2Q_CTR_ASIDE], n_slow = ctr[F2Q_CTR_ASIDE_SLOW];
    if (!n_def) return F2Q_OK;
    // the raw records' kernel may still be adding entries: the counters read after the launch say too little then
    const unsigned long long *known = ctr;
    if (c->aux_busy) { HIPC(c, hipStreamSynchronize(c->aux_stream)); c->aux_busy = false; known = nullptr; }
    int rc = ec_reserve(c, n_slow, n_def, n_def * ((uint64_t)blk.rmax + F2Q_MAX_ITER), known);
    if (rc) return rc;
    const uint32_t g = (uint32_t)std::min<uint64_t>((n_def + 255) / 256, (uint64_t)c->n_cu * 8u);
    if (n_def > n_slow) {
        hipLaunchKernelGGL(k_ec_deferred_keys, dim3(g), dim3(256), 0, c->stream, c->ec, blk, acc, c->reads_seen, c->defer_d);
        HIPC(c, hipGetLastError());
        launches++;
        EC_POINT(c, "k_ec_deferred_keys");
    }
    if (n_slow && !blk.planar_nw) {
        hipLaunchKernelGGL(k_ec_deferred_fixed, dim3(g), dim3(256), 0, c->stream, c->run_d, c->ec, blk, acc, c->reads_seen, c->defer_d);
        HIPC(c, hipGetLastError());
        launches++;
        EC_POINT(c, "k_ec_deferred_fixed");
    } else if (n_slow) {
        hipLaunchKernelGGL(k_ec_deferred_slow, dim3(g), dim3(256), 0, c->stream, c->run_d, c->lib_d, c->ec, blk, acc, c->reads_seen, c->defer_d);
        HIPC(c, hipGetLastError());
        launches++;
        EC_POINT(c, "k_ec_deferred_slow");
    }
    return F2Q_OK;
}

static int launch_block(f2q_ctx *c, const f2q_block *b, f2q_timing *t, hipEvent_t k0 = nullptr, hipEvent_t k1 = nullptr)
{
    if (!k0) { k0 = c->ev_k0; k1 = c->ev_k1; }          // (a queued step brings its own pair)
    if (c->prm.mode == 0 && !c->have_lib) return fail(c, F2Q_ESTATE, "f2q_set_features must be called before counting in Counter mode");
    Accum acc{c->acc_d, c->acc_d + (c->acc_n - 5), nullptr, nullptr, nullptr, nullptr};
#ifdef F2Q_STAMP
    static unsigned long long *stamp_d = nullptr;
    if (!stamp_d) { (void)hipMalloc((void **)&stamp_d, 64); (void)hipMemset(stamp_d, 0, 64); }
    acc.stamp = stamp_d;
#endif
    uint32_t launches = 0;
    HIPC(c, hipEventRecord(k0, c->stream));
    if (c->prm.mode == 0) {
        int rc = launch_view(c, b->pb, b->rb, acc, launches);
        if (rc) return rc;
    } else if (b->n_reads) {
        // Extract+Count: the tables must have room for "every read of the launch is a new key".  Sizing them for the
        // whole block (50 M reads -> 2^28 slots) makes the table many GiB and every probe a DRAM access, so the block
        // is walked in steps of F2Q_EC_STEP reads: room for one step beyond the keys already there is enough, and the
        // tables grow (device rehash) only when the number of distinct keys does.
        uint64_t step = (uint64_t)8 << 20;
        { const char *e = getenv("F2Q_EC_STEP"); if (e && atol(e) >= F2Q_TILE) step = (uint64_t)atol(e); }
        const uint32_t tiles_per = (uint32_t)std::max<uint64_t>(1, step / F2Q_TILE);
        const RawBlock none{};
        auto view_of = [&](uint32_t t0, uint32_t nt) {
            PackedBlock v = b->pb;
            v.n_tiles = nt; v.n_slots = (uint64_t)nt * F2Q_TILE;
            v.bases += (size_t)t0 * v.wb * F2Q_TILE; v.qual += (size_t)t0 * v.wq * F2Q_TILE;
            if (v.len) v.len += (size_t)t0 * F2Q_TILE;
            if (v.index) v.index += (size_t)t0 * F2Q_TILE; else v.first_index += (uint64_t)t0 * F2Q_TILE;
            return v;
        };
        // anchored tiles (one pair) and fixed-window tiles take the hot-key kernels
        const bool hot_path = b->pb.n_tiles && !c->no_hot && !c->plan.multi_pair && (b->pb.planar_nw ? b->pb.len != nullptr && b->pb.planar_nw != 10 : true);
        if (hot_path) {
            // anchored tiles: hot keys in LDS.  The first hot_learn reads of a sample go through the same kernel with an
            // empty hot set (every key takes the table's insert); then the set is built and serves the rest of the sample.
            // The raw records are decided on a second stream while the packed tiles are counted.
            const uint64_t n = (uint64_t)b->pb.n_tiles * F2Q_TILE;
            const uint64_t to_learn = c->hot_valid || c->ec_learned >= c->hot_learn ? 0 : std::min<uint64_t>(n, c->hot_learn - c->ec_learned);
            unsigned long long ctr[F2Q_CTR_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
            int rc;
            if (c->ec.ctr) {
                HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
                HIPC(c, hipStreamSynchronize(c->stream));
                if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error, code " + std::to_string(ctr[2]) + ")");
            }
            // new single-word keys to expect: at the rate the sample has shown once the learning reads are in (a table that
            // fills up all the same only moves reads to the deferred pass), a guess of one key per six reads before that
            auto expect_of = [&](uint64_t reads) {
                const double rate = std::min(1.0, (double)ctr[3] / (double)std::max<uint64_t>(c->ec_learned, 1));
                return std::min<uint64_t>(reads, (uint64_t)(rate * (double)reads) + 4096);
            };
            const uint64_t aside = 4096 + n / 2048;                                 // reads the packed kernel may set aside
            const uint64_t raw_bytes = b->raw_key_bytes + b->rb.n * F2Q_MAX_ITER + aside * ((uint64_t)b->pb.rmax + F2Q_MAX_ITER);
            if ((rc = ec_reserve(c, (to_learn ? to_learn + (n - to_learn) / 6 : expect_of(n)) + b->rb.n + aside, b->rb.n + aside, raw_bytes, ctr))) return rc;
            // the list of reads set aside: room for every slot of the block, cleared once per block
            if (n > c->defer_cap || !c->defer_d) {
                if (c->defer_d) (void)hipFree(c->defer_d);
                c->defer_d = nullptr; c->defer_cap = 0;
                HIPC(c, hipMalloc((void **)&c->defer_d, (size_t)n * sizeof(unsigned long long)));
                c->defer_cap = n;
            }
            HIPC(c, hipMemsetAsync(c->ec.ctr + F2Q_CTR_ASIDE, 0, 16, c->stream));
            uint32_t t0 = 0;
            if (to_learn) {
                const uint32_t nt = (uint32_t)((to_learn + F2Q_TILE - 1) / F2Q_TILE);
                if ((rc = launch_hot(c, view_of(0, nt), 0, acc, launches, true, ctr, nullptr))) return rc;
                c->ec_learned += (uint64_t)nt * F2Q_TILE;
                t0 = nt;
                // the sample's own rate replaces the guess (the reads set aside so far are still to come: `aside` covers them)
                if (t0 < b->pb.n_tiles && (rc = ec_reserve(c, expect_of(n - (uint64_t)t0 * F2Q_TILE) + b->rb.n + aside, b->rb.n + aside, raw_bytes, ctr))) return rc;
            }
            if (t0 < b->pb.n_tiles && !c->hot_valid && (rc = hot_build(c))) return rc;
            if (b->rb.n) {
                if (!c->aux_stream) {
                    HIPC(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
                    HIPC(c, hipEventCreateWithFlags(&c->ev_aux0, hipEventDisableTiming));
                    HIPC(c, hipEventCreateWithFlags(&c->ev_aux1, hipEventDisableTiming));
                }
                HIPC(c, hipEventRecord(c->ev_aux0, c->stream));                     // the tables and the block are ready
            }
            // the long launch goes first; the raw records' kernel fills the wave slots it leaves free
            if (t0 < b->pb.n_tiles) {
                if ((rc = launch_hot(c, view_of(t0, b->pb.n_tiles - t0), (uint64_t)t0 * F2Q_TILE, acc, launches, false, ctr, b->rb.n ? b : nullptr))) return rc;
            } else if (b->rb.n && (rc = launch_aux_general(c, b, acc, launches))) return rc;
            if ((rc = hot_aside(c, view_of(0, b->pb.n_tiles), acc, launches, ctr))) return rc;
            if (c->aux_busy) {                                                       // join: the block is done when both streams are
                HIPC(c, hipEventRecord(c->ev_aux1, c->aux_stream));
                HIPC(c, hipStreamWaitEvent(c->stream, c->ev_aux1, 0));
                c->aux_busy = false;
            }
        } else
        for (uint32_t t0 = 0; t0 < b->pb.n_tiles; t0 += tiles_per) {
            const PackedBlock v = view_of(t0, std::min<uint32_t>(tiles_per, b->pb.n_tiles - t0));
            // packed reads give single-window keys of at most rmax bytes (several pairs: one window each, at most F2Q_PAIRS_KEYMAX in all)
            const uint64_t key_max = c->plan.multi_pair ? (uint64_t)F2Q_PAIRS_KEYMAX : (uint64_t)v.rmax + F2Q_MAX_ITER;
            int rc = ec_reserve(c, v.n_slots, v.n_slots, v.n_slots * key_max);
            if (rc) return rc;
            if ((rc = launch_view(c, v, none, acc, launches))) return rc;
        }
        const PackedBlock nop{};
        for (uint64_t r0 = 0; r0 < b->rb.n && !hot_path; r0 += step) {
            RawBlock v = b->rb;
            v.n = std::min<uint64_t>(step, b->rb.n - r0);
            v.off += r0; v.len += r0; v.qlen += r0;
            if (v.qoff) v.qoff += r0;
            if (v.index) v.index += r0; else v.first_index += r0;
            // no key is longer than its record's bytes + separators (block-wide bound: the arena is never cleared)
            int rc = ec_reserve(c, v.n, v.n, b->raw_key_bytes + v.n * F2Q_MAX_ITER);
            if (rc) return rc;
            if ((rc = launch_view(c, nop, v, acc, launches))) return rc;
        }
    }
    HIPC(c, hipEventRecord(k1, c->stream));
    c->reads_seen += b->n_reads;
#ifdef F2Q_STAMP
    { unsigned long long h[8]; (void)hipStreamSynchronize(c->stream); (void)hipMemcpy(h, acc.stamp, 64, hipMemcpyDeviceToHost);
      (void)hipMemset(acc.stamp, 0, 64);
      unsigned long long tot = 0;
      for (int i = 0; i < 8; i++) tot += h[i];
      if (tot) {
          fprintf(stderr, "[stamp]");
          for (int i = 0; i < 8; i++) if (h[i]) fprintf(stderr, " phase %d %.1f%%", i, 100.0 * h[i] / tot);
          fprintf(stderr, "  (%.0f clock ticks/wave-tile)\n", (double)tot / ((double)b->pb.n_tiles * 4));
      } }
#endif
    if (t) {
        HIPC(c, hipEventSynchronize(k1));
        float ms = 0;
        HIPC(c, hipEventElapsedTime(&ms, k0, k1));
        t->kernel_ms = ms; t->reads = b->n_reads; t->general_reads = b->n_general;
        t->fast_reads = b->n_reads - b->n_general; t->launches = launches; t->path = c->last_path;
    }
    if (c->prm.mode == 1 && b->n_reads && !(b->pb.n_tiles && !c->no_hot && !c->plan.multi_pair && (b->pb.planar_nw ? b->pb.len != nullptr && b->pb.planar_nw != 10 : true))) {
        // (the hot-key path has looked at the counters after its last launch; what its deferred passes could still
        // report is seen by the next call that reads them)
        unsigned long long ctr[4];
        HIPC(c, hipMemcpyAsync(ctr, c->ec.ctr, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error, code " + std::to_string(ctr[2]) + ")");
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

extern "C" int f2q_count_resident_queued(f2q_ctx *c, const f2q_block *b)
{
    if (!c || !b) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    while (c->q_ev.size() < 2 * ((size_t)c->q_n + 1)) {
        hipEvent_t e = nullptr;
        HIPC(c, hipEventCreate(&e));
        c->q_ev.push_back(e);
    }
    int rc = launch_block(c, b, nullptr, c->q_ev[2 * (size_t)c->q_n], c->q_ev[2 * (size_t)c->q_n + 1]);
    if (rc) return rc;
    c->q_n++;
    return F2Q_OK;
}

extern "C" int f2q_queued_times(f2q_ctx *c, float *kernel_ms, uint32_t cap, uint32_t *n)
{
    if (!c || !n) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    *n = c->q_n;
    for (uint32_t i = 0; i < c->q_n && i < cap && kernel_ms; i++)
        HIPC(c, hipEventElapsedTime(&kernel_ms[i], c->q_ev[2 * (size_t)i], c->q_ev[2 * (size_t)i + 1]));
    c->q_n = 0;
    return F2Q_OK;
}

extern "C" void f2q_block_free(f2q_ctx *c, f2q_block *b)
{
    if (!b) return;
    if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    free_all(c, b->allocs);
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
            b->raw_key_bytes = hp.raw.size();
        }
        hipError_t e = hipStreamSynchronize(c->stream);      // host staging vectors die with this frame
        if (e != hipSuccess) { rc = fail(c, F2Q_EHIP, hipGetErrorString(e)); break; }
    } while (0);
    if (rc) { free_all(c, b->allocs); delete b; return rc; }
    *out = b;
    return F2Q_OK;
}


// out[i] = in[0] + ... + in[i-1] on the context's stream (k_scan_blocks / k_scan_sums / k_scan_add)
static int exclusive_scan(f2q_ctx *c, const uint32_t *in, uint32_t *out, uint32_t n, std::vector<void *> &tmp)
{
    if (!n) return F2Q_OK;
    const uint32_t n_blocks = (n + F2Q_SCAN_BLOCK - 1u) / F2Q_SCAN_BLOCK;
    uint32_t *sums;
    int rc = dev_alloc(c, (size_t)n_blocks, &sums, tmp);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_blocks, dim3(n_blocks), dim3(F2Q_SCAN_THREADS), 0, c->stream, in, out, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(F2Q_SCAN_THREADS), 0, c->stream, sums, n_blocks);
    hipLaunchKernelGGL(k_scan_add, dim3(n_blocks), dim3(F2Q_SCAN_THREADS), 0, c->stream, out, n, sums);
    HIPC(c, hipGetLastError());
    return F2Q_OK;
}

// FASTQ text -> resident block, framing and packing done by the device (k_nl_count .. k_pack).  Handles up to
// 2 GiB of text per call; *consumed = bytes up to the end of the last complete record.
// text that is already (on its way) in device memory: `buf` is the allocation (it becomes the block's), `text` the
// 16-byte aligned start of the FASTQ bytes inside it, with room for the census padding behind them
struct DevText { void *buf = nullptr; size_t cap = 0; uint8_t *text = nullptr; uint8_t last_byte = 0; bool borrowed = false; };   // borrowed: the caller keeps the allocation (f2q_text)

static int block_from_text_device(f2q_ctx *c, const uint8_t *fastq, size_t nbytes, size_t *consumed, f2q_block **out,
                                  const DevText *pre = nullptr, uint64_t max_records = ~0ull)
{
    *out = nullptr; *consumed = 0;
    f2q_block *b = new f2q_block();
    if (pre && !pre->borrowed) b->allocs.push_back(pre->buf);
    if (nbytes == 0) { if (pre) free_all(c, b->allocs); *out = b; return F2Q_OK; }    // an empty buffer is an empty block
    std::vector<void *> tmp;                         // scratch freed before returning
    int rc = F2Q_OK;
    auto bail = [&](int code) { free_all(c, tmp); free_all(c, b->allocs); delete b; return code; };
    const uint32_t n_chunks = (uint32_t)((nbytes + F2Q_NL_CHUNK - 1) / F2Q_NL_CHUNK);
    const size_t padded = (size_t)n_chunks * F2Q_NL_CHUNK + 16;
    uint8_t *d_text; uint32_t *d_cc, *d_cp;
    if (pre) {
        d_text = pre->text;
        if ((size_t)(d_text - (uint8_t *)pre->buf) + padded > pre->cap) { fail(c, F2Q_EINVAL, "staged text buffer too small"); return bail(F2Q_EINVAL); }
    } else if ((rc = dev_alloc(c, padded, &d_text, b->allocs))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_chunks + 1, &d_cc, tmp, 0))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)n_chunks + 1, &d_cp, tmp))) return bail(rc);
#define ING(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(c, F2Q_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); return bail(F2Q_EHIP); } } while (0)
    ING(hipMemsetAsync(d_text + nbytes, 0, padded - nbytes, c->stream));
    const double tc0 = now_ms();
    if (!pre) ING(hipMemcpyAsync(d_text, fastq, nbytes, hipMemcpyHostToDevice, c->stream));
    if (c->trace) { ING(hipStreamSynchronize(c->stream)); c->tr_copy += now_ms() - tc0; }
    hipLaunchKernelGGL(k_nl_count, dim3(n_chunks), dim3(256), 0, c->stream, d_text, (uint64_t)nbytes, d_cc);
    ING(hipGetLastError());
    if ((rc = exclusive_scan(c, d_cc, d_cp, (uint32_t)n_chunks + 1u, tmp))) return bail(rc);
    uint32_t n_newlines = 0;
    ING(hipMemcpyAsync(&n_newlines, d_cp + n_chunks, 4, hipMemcpyDeviceToHost, c->stream));
    ING(hipStreamSynchronize(c->stream));
    const bool open_tail = nbytes > 0 && (pre ? pre->last_byte : fastq[nbytes - 1]) != '\n';
    const uint64_t n_lines = (uint64_t)n_newlines + (open_tail ? 1 : 0);
    const uint32_t n_rec = (uint32_t)std::min<uint64_t>(n_lines / 4, max_records);   // (a piece of a sharded file owns only the records that start in it)
    uint32_t *d_ls;
    if ((rc = dev_alloc(c, (size_t)n_newlines + 2, &d_ls, tmp))) return bail(rc);
    hipLaunchKernelGGL(k_line_starts, dim3(n_chunks), dim3(256), 0, c->stream, d_text, (uint64_t)nbytes, d_cp, d_ls);
    ING(hipGetLastError());
    const uint32_t sentinel = (uint32_t)nbytes + 1u;
    ING(hipMemcpyAsync(d_ls + n_newlines + 1, &sentinel, 4, hipMemcpyHostToDevice, c->stream));
    b->n_reads = n_rec;
    if (n_rec == 0) { ING(hipStreamSynchronize(c->stream)); free_all(c, tmp); *out = b; return F2Q_OK; }
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
    const unsigned igrid = (unsigned)((n_rec + F2Q_ING_THREADS - 1u) / F2Q_ING_THREADS);
    hipLaunchKernelGGL(k_classify, dim3(igrid), dim3(F2Q_ING_THREADS), 0, c->stream, ing, c->plan);
    ING(hipGetLastError());
    if ((rc = exclusive_scan(c, ing.clean, d_before, n_rec + 1u, tmp))) return bail(rc);
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
        b->dev_bytes += nbytes;
        b->raw_key_bytes = nbytes;                    // the records point into the text: no key is longer than its record
    }
    b->n_general = n_dirty;
    hipLaunchKernelGGL(k_pack, dim3(igrid), dim3(F2Q_ING_THREADS), 0, c->stream, ing, c->plan, d_before, o);
    ING(hipGetLastError());
    ING(hipStreamSynchronize(c->stream));             // scratch dies with this frame
#undef ING
    free_all(c, tmp);
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

// one window of text (at most 1 GiB: the device packer indexes it with 32 bits): frame, pack, count, free
static int count_window(f2q_ctx *c, const uint8_t *fastq, size_t take, const DevText *pre, size_t *used_out, f2q_timing *one,
                        uint64_t max_records = ~0ull)
{
    f2q_block *b = nullptr; size_t used = 0;
    int rc;
    const double t0 = now_ms();
    if (!c->host_pack || pre || max_records != ~0ull) rc = block_from_text_device(c, fastq, take, &used, &b, pre, max_records);
    else {
        std::vector<Rec> recs;
        used = frame_fastq(fastq, take, recs);
        rc = recs.empty() ? F2Q_OK : block_from_records(c, recs, &b);
    }
    if (rc) return rc;
    const double t1 = now_ms();
    if (b && b->n_reads) rc = launch_block(c, b, one);
    if (c->trace) (void)hipStreamSynchronize(c->stream);
    const double t2 = now_ms();
    if (b) f2q_block_free(c, b);
    c->tr_frame += t1 - t0; c->tr_count += t2 - t1; c->tr_free += now_ms() - t2;
    *used_out = used;
    return rc;
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
        size_t used = 0;
        f2q_timing one; memset(&one, 0, sizeof one);
        rc = count_window(c, fastq + pos, take, nullptr, &used, t ? &one : nullptr);
        if (rc) return rc;
        sum.kernel_ms += one.kernel_ms; sum.reads += one.reads; sum.fast_reads += one.fast_reads;
        sum.general_reads += one.general_reads; sum.launches += one.launches; if (one.path) sum.path = one.path;
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

// ---- FASTQ text resident in device memory ------------------------------------------------------------------
struct f2q_text { void *buf = nullptr; size_t cap = 0, nbytes = 0; uint8_t last = 0; };

extern "C" int f2q_text_upload(f2q_ctx *c, const uint8_t *fastq, size_t nbytes, f2q_text **out)
{
    if (!c || !out || (!fastq && nbytes)) return F2Q_EINVAL;
    if (nbytes > ((size_t)1 << 30)) return fail(c, F2Q_EINVAL, "f2q_text_upload: at most 1 GiB of text (the device framing indexes a text with 32 bits)");
    HIPC(c, hipSetDevice(c->device));
    f2q_text *t = new f2q_text();
    const size_t n_chunks = (nbytes + F2Q_NL_CHUNK - 1) / F2Q_NL_CHUNK;
    t->cap = n_chunks * F2Q_NL_CHUNK + 16; t->nbytes = nbytes; t->last = nbytes ? fastq[nbytes - 1] : 0;
    hipError_t e = hipMalloc(&t->buf, t->cap);
    if (e == hipSuccess && nbytes) e = hipMemcpyAsync(t->buf, fastq, nbytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { if (t->buf) (void)hipFree(t->buf); delete t; return fail(c, F2Q_EHIP, std::string("f2q_text_upload: ") + hipGetErrorString(e)); }
    *out = t;
    return F2Q_OK;
}

extern "C" int f2q_count_text(f2q_ctx *c, f2q_text *txt, size_t *consumed, f2q_timing *t)
{
    if (!c || !txt) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    if (t) { memset(t, 0, sizeof *t); HIPC(c, hipEventRecord(c->ev_a, c->stream)); }
    if (consumed) *consumed = 0;
    DevText pre; pre.buf = txt->buf; pre.cap = txt->cap; pre.text = (uint8_t *)txt->buf; pre.last_byte = txt->last; pre.borrowed = true;
    size_t used = 0;
    f2q_timing one; memset(&one, 0, sizeof one);
    int rc = txt->nbytes ? count_window(c, nullptr, txt->nbytes, &pre, &used, t ? &one : nullptr) : F2Q_OK;
    if (rc) return rc;
    if (consumed) *consumed = used;
    if (t) {
        hipError_t e = hipEventRecord(c->ev_b, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev_b);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_a, c->ev_b);
        if (e != hipSuccess) return fail(c, F2Q_EHIP, hipGetErrorString(e));
        *t = one; t->total_ms = ms;
    }
    return F2Q_OK;
}

extern "C" void f2q_text_free(f2q_ctx *c, f2q_text *t)
{
    if (!t) return;
    if (c) { (void)hipSetDevice(c->device); if (c->stream) (void)hipStreamSynchronize(c->stream); }
    if (t->buf) (void)hipFree(t->buf);
    delete t;
}

// ---- file streaming -------------------------------------------------------------------------------
// Page-locked staging buffers are expensive to create (tens of ms per 256 MiB) and every file needs two, so
// they are kept in a small process-wide pool between files (and between contexts: --cp runs several at once).
struct PinBuf { uint8_t *p = nullptr; size_t cap = 0; bool pageable = false; };
struct PinnedPool {
    std::mutex mu;
    std::vector<PinBuf> idle;
    size_t idle_bytes = 0;
    static constexpr size_t KEEP_BYTES = (size_t)1600 << 20;
    bool acquire(size_t cap, PinBuf &out)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].cap >= cap && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
            if (best < idle.size()) { out = idle[best]; idle_bytes -= out.cap; idle.erase(idle.begin() + (ptrdiff_t)best); return true; }
        }
        out.p = nullptr; out.cap = cap; out.pageable = false;
        if (hipHostMalloc((void **)&out.p, cap, hipHostMallocPortable) == hipSuccess) return true;
        (void)hipGetLastError();                          // no page-locked memory to be had: ordinary memory works too,
        out.pageable = true;                              // the copies to the device are just staged by the runtime
        out.p = (uint8_t *)malloc(cap);
        return out.p != nullptr;
    }
    void release(PinBuf &b)
    {
        if (!b.p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            if (!b.pageable && idle_bytes + b.cap <= KEEP_BYTES && idle.size() < 8) { idle.push_back(b); idle_bytes += b.cap; b = PinBuf(); return; }
        }
        if (b.pageable) free(b.p); else (void)hipHostFree(b.p);
        b = PinBuf();
    }
};
static PinnedPool g_pinned;

// reads_counter's file half (fast2q.py:560-578).  A reader thread (f2q_reader.h: parallel pread / parallel BGZF
// inflate / gzread) fills one pinned buffer while the device frames, packs and counts the other; whole lines only
// are handed over, the unconsumed tail (a partial record) is carried in front of the next piece.
// The framing of a piece that another rank counts: how many records it holds and where the last complete one ends
// (the same verdict block_from_text_device reaches on the device), from a newline census by the worker pool.
static size_t skip_piece(const uint8_t *p, size_t n, int threads, uint64_t *n_records)
{
    *n_records = 0;
    if (n == 0) return 0;
    const size_t slice_min = (size_t)1 << 20;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n / slice_min));
    std::vector<uint64_t> cnt((size_t)T, 0);
    auto bounds = [&](int t, size_t &a, size_t &b) { a = n * (size_t)t / (size_t)T; b = n * (size_t)(t + 1) / (size_t)T; };
    auto work = [&](int t) {
        size_t a, b; bounds(t, a, b);
        uint64_t k = 0;
        for (size_t i = a; i < b; i++) k += (p[i] == 0x0a);
        cnt[(size_t)t] = k;
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    uint64_t n_nl = 0; for (uint64_t k : cnt) n_nl += k;
    const uint64_t n_lines = n_nl + (p[n - 1] != 0x0a ? 1 : 0);
    const uint64_t n_rec = n_lines / 4;
    *n_records = n_rec;
    if (n_rec == 0) return 0;
    if (4 * n_rec > n_nl) return n;                       // the last record's last line has no newline: everything
    // position after the (4 * n_rec)-th newline
    uint64_t want = 4 * n_rec, seen = 0;
    for (int t = 0; t < T; t++) {
        if (seen + cnt[(size_t)t] >= want) {
            size_t a, b; bounds(t, a, b);
            for (size_t i = a; i < b; i++) if (p[i] == 0x0a && ++seen == want) return i + 1;
        }
        seen += cnt[(size_t)t];
    }
    return n;
}

static int count_file_impl(f2q_ctx *c, const char *path, uint32_t rank, uint32_t world, f2q_timing *t)
{
    if (!c || !path || world == 0 || rank >= world) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    size_t CH = (size_t)256 << 20;                 // bytes of text per piece; F2Q_FILE_CHUNK overrides (tests)
    { const char *e = getenv("F2Q_FILE_CHUNK"); if (e && atol(e) >= 4096) CH = (size_t)atol(e); }
    TextSource src;
    { std::string err; if (src.open(path, err) != 0) return fail(c, F2Q_EIO, err); }
    const size_t HEAD = 64 << 10;                  // room in front of each piece for the carried tail
    if (src.kind == TextSource::PLAIN && src.regular) CH = std::min<size_t>(CH, std::max<size_t>(src.file_size, 4096));
    else if (src.regular) CH = std::min<size_t>(CH, std::max<size_t>(src.file_size * 16, (size_t)4 << 20));
    const double tr_a = now_ms();
    // three staging buffers: one being counted, one ready (its text already travelling to the device), one being read
    constexpr int NSLOT = 3;
    PinBuf buf[NSLOT];
    auto drop = [&]() { for (auto &b : buf) g_pinned.release(b); };
    for (auto &b : buf) if (!g_pinned.acquire(HEAD + CH, b)) { drop(); return fail(c, F2Q_ENOMEM, "cannot allocate the read buffers"); }
    const double tr_b = now_ms();

    // the second stream is made before the reader thread exists: no early return may leave a joinable thread behind
    const bool can_stage = world == 1 && !c->host_pack && !getenv("F2Q_NO_STAGING");
    const bool force_stage = getenv("F2Q_FORCE_STAGING") != nullptr;
    if (can_stage && !c->copy_stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming);
        if (e != hipSuccess) { drop(); return fail(c, F2Q_EHIP, std::string("copy stream: ") + hipGetErrorString(e)); }
    }

    struct Piece { int slot; size_t n; };
    std::mutex mu; std::condition_variable cv;
    std::deque<Piece> ready; bool slot_free[NSLOT]; for (bool &f : slot_free) f = true; bool stop = false;
    double read_ms = 0;
    std::thread reader([&]() {
        for (int slot = 0;; slot = (slot + 1) % NSLOT) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return slot_free[slot] || stop; }); if (stop) return; slot_free[slot] = false; }
            const double r0 = now_ms();
            const size_t n = src.read(buf[slot].p + HEAD, CH);
            read_ms += now_ms() - r0;
            { std::lock_guard<std::mutex> g(mu); ready.push_back(Piece{slot, n}); }
            cv.notify_all();
            if (n == 0) return;
        }
    });

    // The text of piece k+1 is copied to the device (second stream) while piece k is framed, packed and counted.  Its
    // carried tail is only known once piece k is framed, so the piece lands HEAD bytes into its device buffer and the
    // tail is put in front of it later; the text then starts wherever HEAD - tail falls, and the up to 15 bytes between
    // the 16-byte boundary below it and the text are filled with 'x': they lengthen the first line, which is a record's
    // header line and is never looked at (fast2q.py:324-328 takes lines 2 and 4 only).
    struct Staged { void *buf = nullptr; size_t cap = 0; int slot = -1; size_t n = 0; } staged;
    auto unstage = [&]() {                         // give up a staged copy (after it has landed)
        if (!staged.buf) return;
        (void)hipStreamSynchronize(c->copy_stream);
        std::vector<void *> v{staged.buf}; free_all(c, v);
        staged = Staged();
    };
    auto stage_next = [&]() {
        if (!can_stage || staged.buf) return;
        Piece nx{-1, 0};
        {
            std::unique_lock<std::mutex> lk(mu);
            if (force_stage) cv.wait(lk, [&] { return !ready.empty(); });      // tests: every piece takes this path
            if (!ready.empty()) nx = ready.front();
        }
        if (nx.slot < 0 || nx.n == 0) return;
        void *d = nullptr;
        const size_t cap = HEAD + nx.n + 2 * (size_t)F2Q_NL_CHUNK + 64;
        if (dev_get(c, cap, &d) != F2Q_OK) return;     // no memory for it: the piece takes the ordinary path
        if (hipMemcpyAsync((uint8_t *)d + HEAD, buf[nx.slot].p + HEAD, nx.n, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
            hipEventRecord(c->ev_copy, c->copy_stream) != hipSuccess) {
            (void)hipGetLastError(); (void)hipStreamSynchronize(c->copy_stream);
            std::vector<void *> v{d}; free_all(c, v);
            return;
        }
        staged.buf = d; staged.cap = cap; staged.slot = nx.slot; staged.n = nx.n;
    };

    f2q_timing sum; memset(&sum, 0, sizeof sum);
    int rc = F2Q_OK;
    std::vector<uint8_t> carry, big;
    double wait_ms = 0;
    uint32_t piece_no = 0;                         // pieces are dealt to the ranks round robin
    for (;;) {
        Piece pc;
        { const double w0 = now_ms(); std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !ready.empty(); }); pc = ready.front(); ready.pop_front(); wait_ms += now_ms() - w0; }
        const bool eof = (pc.n == 0);
        Staged mine;                               // this piece's text, if it was sent ahead
        if (staged.buf && staged.slot == pc.slot && staged.n == pc.n) {
            mine = staged; staged = Staged();
            if (hipStreamWaitEvent(c->stream, c->ev_copy, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->copy_stream); }
        } else unstage();
        if (!eof) stage_next();                    // the piece after this one, if the reader has it already (after the end marker nothing follows)
        uint8_t *base; size_t have;
        if (carry.size() <= HEAD) {
            base = buf[pc.slot].p + HEAD - carry.size();
            if (!carry.empty()) memcpy(base, carry.data(), carry.size());
            have = carry.size() + pc.n;
        } else {                                   // a tail longer than the head room (very long lines): pageable detour
            big.resize(carry.size() + pc.n);
            memcpy(big.data(), carry.data(), carry.size());
            if (pc.n) memcpy(big.data() + carry.size(), buf[pc.slot].p + HEAD, pc.n);
            base = big.data(); have = big.size();
        }
        size_t used = 0;
        if (have) {
            f2q_timing one; memset(&one, 0, sizeof one);
            size_t cut = have;                     // only whole lines: a line is never split between blocks
            // (at the end of a damaged archive too: readline raised there instead of returning the cut-off line, :405-407)
            if (!eof || src.truncated()) while (cut > 0 && base[cut - 1] != 0x0a) cut--;
            if (cut) {
                if (mine.buf && carry.size() <= HEAD) {
                    const size_t c_len = carry.size(), textoff = HEAD - c_len, al = textoff & ~(size_t)15, lead = textoff - al;
                    uint8_t *d = (uint8_t *)mine.buf;
                    hipError_t e = hipSuccess;
                    if (c_len) e = hipMemcpyAsync(d + textoff, carry.data(), c_len, hipMemcpyHostToDevice, c->stream);
                    if (e == hipSuccess && lead) e = hipMemsetAsync(d + al, 'x', lead, c->stream);
                    if (e != hipSuccess) { rc = fail(c, F2Q_EHIP, hipGetErrorString(e)); (void)hipStreamSynchronize(c->stream); std::vector<void *> v{mine.buf}; free_all(c, v); }
                    else {
                        DevText pre; pre.buf = mine.buf; pre.cap = mine.cap; pre.text = d + al; pre.last_byte = base[cut - 1];
                        size_t used_dev = 0;
                        rc = count_window(c, nullptr, lead + cut, &pre, &used_dev, &one);      // the buffer now belongs to the block
                        used = used_dev > lead ? used_dev - lead : 0;
                    }
                    mine = Staged();
                } else if (piece_no % world == rank) rc = f2q_count_block(c, base, cut, &used, &one);
                else {                             // another rank's piece: only its framing matters here
                    uint64_t n_rec = 0;
                    used = skip_piece(base, cut, src.n_threads, &n_rec);
                    c->reads_seen += n_rec;
                }
                piece_no++;
            }
            if (mine.buf) { (void)hipStreamSynchronize(c->stream); std::vector<void *> v{mine.buf}; free_all(c, v); mine = Staged(); }   // not used after all
            if (eof) used = have;                  // trailing partial record is dropped (:392)
            sum.kernel_ms += one.kernel_ms; sum.total_ms += one.total_ms; sum.reads += one.reads;
            sum.fast_reads += one.fast_reads; sum.general_reads += one.general_reads; sum.launches += one.launches; if (one.path) sum.path = one.path;
        }
        if (!rc) { std::vector<uint8_t> rest(base + used, base + have); carry.swap(rest); }
        { std::lock_guard<std::mutex> g(mu); slot_free[pc.slot] = true; if (rc || eof) stop = true; }
        cv.notify_all();
        if (rc || eof) break;
    }
    unstage();
    reader.join();
    drop();
    if (t) *t = sum;
    if (c->trace) fprintf(stderr, "[f2q trace] %s (%s, %d io threads): pinned %.1f ms, reader busy %.1f ms, waited for reader %.1f ms, frame+pack %.1f ms (H2D copy %.1f), count %.1f ms, free %.1f ms\n",
                          path, src.kind_name(), src.n_threads, tr_b - tr_a, read_ms, wait_ms, c->tr_frame, c->tr_copy, c->tr_count, c->tr_free);
    if (rc) return rc;
    if (src.truncated()) return fail(c, F2Q_ETRUNCATED, std::string(path) + " is an incomplete or corrupted gzip file");
    return F2Q_OK;
}

extern "C" int f2q_count_file(f2q_ctx *c, const char *path, f2q_timing *t) { return count_file_impl(c, path, 0, 1, t); }

// one process per GPU on the same file: every rank streams the whole file (the framing is global), counts the
// pieces k with k % world == rank on its device and only takes a newline census of the others
extern "C" int f2q_count_file_shard(f2q_ctx *c, const char *path, uint32_t rank, uint32_t world, f2q_timing *t)
{
    return count_file_impl(c, path, rank, world, t);
}

// ---- one plain file counted by several processes without anybody reading foreign bytes ------------------------------
// The 4-line framing is global (fast2q.py:324-328: a record is lines 4i..4i+3 of the FILE), so a rank can only frame its
// share once it knows how many lines precede it.  The file is cut into pieces of piece_bytes, piece k belongs to rank
// k % world, and the work is split in two steps around ONE small all-reduce that the caller does (it owns the
// communicator): (1) every rank counts the newlines of ITS pieces (f2q_census_pieces), (2) with the summed census every
// rank knows the line index at which each of its pieces starts and counts the records whose first line STARTS in
// its pieces, reading on past the piece's end only to finish its last record (f2q_count_pieces).
struct PieceSpan { uint64_t base, size; };
// a piece and its 1 MiB look-ahead are framed in one window, and the device framing indexes a window with 32 bits
// (count_file_impl and f2q_count_block cap their windows at 1 GiB for the same reason)
static const uint64_t F2Q_MAX_PIECE_BYTES = ((uint64_t)1 << 30) - ((uint64_t)2 << 20);
static inline bool piece_bytes_ok(uint64_t piece_bytes) { return piece_bytes >= 4096 && piece_bytes <= F2Q_MAX_PIECE_BYTES; }
static inline PieceSpan piece_span(uint64_t file_size, uint64_t piece_bytes, uint64_t k)
{
    const uint64_t base = k * piece_bytes;
    return PieceSpan{base, base < file_size ? std::min<uint64_t>(piece_bytes, file_size - base) : 0};
}

// how a file is cut: plain files by byte ranges; BGZF files by runs of whole members whose text adds up to at most
// piece_bytes (a member is <= 64 KiB of text and carries its text size in its trailer, so the cut needs no inflating)
struct PieceMap {
    bool bgzf = false;
    uint64_t n = 0, max_text = 0;
    std::vector<uint64_t> c_off, text;                     // BGZF: compressed offset of the piece's first member, its text bytes
};
static int piece_map(TextSource &src, const char *path, uint64_t piece_bytes, PieceMap &pm)
{
    pm = PieceMap();
    if (src.kind == TextSource::PLAIN && src.regular) {
        pm.n = (src.file_size + piece_bytes - 1) / piece_bytes; pm.max_text = std::min<uint64_t>(piece_bytes, src.file_size);
        return F2Q_OK;
    }
    if (!(src.kind == TextSource::BGZF && src.regular)) return F2Q_EUNSUPPORTED;
    // (the three calls of a run ask for the same map: keep the last one)
    static std::mutex mu; static std::string last_key; static PieceMap last;
    struct stat sb; if (stat(path, &sb) != 0) return F2Q_EIO;
    const std::string key = std::string(path) + "|" + std::to_string((unsigned long long)sb.st_size) + "|" + std::to_string((long long)sb.st_mtime) +
                            "|" + std::to_string((unsigned long long)piece_bytes);
    { std::lock_guard<std::mutex> g(mu); if (key == last_key) { pm = last; return F2Q_OK; } }
    std::vector<uint64_t> off; std::vector<uint32_t> isz;
    if (!src.bgzf_index(off, isz)) return F2Q_EUNSUPPORTED;
    pm.bgzf = true;
    uint64_t acc = 0;
    for (size_t i = 0; i < off.size(); i++) {
        if (pm.c_off.empty() || acc + isz[i] > piece_bytes) {
            if (!pm.c_off.empty() && acc == 0) { pm.c_off.back() = pm.c_off.back(); }      // (an empty run keeps its place)
            pm.c_off.push_back(off[i]); pm.text.push_back(0); acc = 0;
        }
        acc += isz[i]; pm.text.back() = acc;
    }
    pm.n = pm.c_off.size();
    for (uint64_t t : pm.text) pm.max_text = std::max(pm.max_text, t);
    { std::lock_guard<std::mutex> g(mu); last_key = key; last = pm; }
    return F2Q_OK;
}

extern "C" int f2q_file_pieces(const char *path, uint64_t piece_bytes, uint64_t *n_pieces, int *shardable)
{
    if (!path || !n_pieces || !shardable || !piece_bytes_ok(piece_bytes)) return F2Q_EINVAL;
    *n_pieces = 0; *shardable = 0;
    TextSource src; std::string err;
    if (src.open(path, err) != 0) { g_create_err = err; return F2Q_EIO; }
    PieceMap pm;
    if (piece_map(src, path, piece_bytes, pm) == F2Q_OK) { *shardable = pm.bgzf ? 2 : 1; *n_pieces = pm.n; }
    return F2Q_OK;
}

// census[2k] = newlines in piece k, census[2k+1] = 1 if its last byte is a newline; only this rank's pieces are written
extern "C" int f2q_census_pieces(const char *path, uint32_t rank, uint32_t world, uint64_t piece_bytes, uint64_t *census, uint64_t n_pieces)
{
    if (!path || !census || world == 0 || rank >= world || !piece_bytes_ok(piece_bytes)) return F2Q_EINVAL;
    TextSource src; std::string err;
    if (src.open(path, err) != 0) { g_create_err = err; return F2Q_EIO; }
    PieceMap pm;
    if (piece_map(src, path, piece_bytes, pm) != F2Q_OK || pm.n != n_pieces) { g_create_err = "not a plain or BGZF regular file (or the pieces changed)"; return F2Q_EUNSUPPORTED; }
    if (pm.bgzf) {
        // inflate this rank's runs of members (the reader's member-parallel decoder) and count
        std::vector<uint8_t> b((size_t)std::min<uint64_t>(std::max<uint64_t>(pm.max_text, 1 << 16), (uint64_t)64 << 20));
        for (uint64_t k = rank; k < n_pieces; k += world) {
            uint64_t left = pm.text[k], nl = 0; uint8_t lastb = 0;
            if (!src.seek_bgzf(pm.c_off[k])) { g_create_err = "seek"; return F2Q_EIO; }
            while (left) {
                const size_t n = src.read(b.data(), (size_t)std::min<uint64_t>(b.size(), left));
                if (n == 0) { g_create_err = "BGZF member damaged or cut off"; return F2Q_EIO; }
                for (size_t j = 0; j < n; j++) nl += (b[j] == 0x0a);
                lastb = b[n - 1]; left -= n;
            }
            census[2 * k] = nl; census[2 * k + 1] = (pm.text[k] && lastb == 0x0a) ? 1 : 0;
        }
        return F2Q_OK;
    }
    const int T = src.n_threads;
    const size_t SL = (size_t)4 << 20;                                  // bytes per pread
    std::vector<std::vector<uint8_t>> bufs((size_t)T);
    for (uint64_t k = rank; k < n_pieces; k += world) {
        const PieceSpan sp = piece_span(src.file_size, piece_bytes, k);
        if (sp.size == 0) { census[2 * k] = 0; census[2 * k + 1] = 0; continue; }
        const uint64_t n_sl = (sp.size + SL - 1) / SL;
        std::vector<uint64_t> cnt((size_t)T, 0);
        std::atomic<uint64_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&](int t) {
            std::vector<uint8_t> &b = bufs[(size_t)t];
            if (b.size() < SL) b.resize(SL);
            for (;;) {
                const uint64_t i = next.fetch_add(1);
                if (i >= n_sl) return;
                const uint64_t off = sp.base + i * SL, len = std::min<uint64_t>(SL, sp.base + sp.size - off);
                uint64_t o = 0;
                while (o < len) { ssize_t r = pread(src.fd, b.data() + o, len - o, (off_t)(off + o)); if (r <= 0) { bad = true; return; } o += (uint64_t)r; }
                uint64_t n = 0;
                for (uint64_t j = 0; j < len; j++) n += (b[j] == 0x0a);
                cnt[(size_t)t] += n;
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        if (bad) { g_create_err = "short read"; return F2Q_EIO; }
        uint64_t nl = 0; for (uint64_t v : cnt) nl += v;
        uint8_t lastb = 0;
        if (pread(src.fd, &lastb, 1, (off_t)(sp.base + sp.size - 1)) != 1) { g_create_err = "short read"; return F2Q_EIO; }
        census[2 * k] = nl; census[2 * k + 1] = lastb == 0x0a ? 1 : 0;
    }
    return F2Q_OK;
}

extern "C" int f2q_count_pieces(f2q_ctx *c, const char *path, uint32_t rank, uint32_t world, uint64_t piece_bytes,
                                const uint64_t *census, uint64_t n_pieces, f2q_timing *t)
{
    if (!c || !path || !census || world == 0 || rank >= world || !piece_bytes_ok(piece_bytes)) return F2Q_EINVAL;
    HIPC(c, hipSetDevice(c->device));
    TextSource src;
    { std::string err; if (src.open(path, err) != 0) return fail(c, F2Q_EIO, err); }
    PieceMap pm;
    if (piece_map(src, path, piece_bytes, pm) != F2Q_OK) return fail(c, F2Q_EUNSUPPORTED, "f2q_count_pieces takes a plain or a BGZF regular file");
    if (n_pieces != pm.n) return fail(c, F2Q_EINVAL, "census does not fit the file");
    // what this rank owns: for every piece the text offset of its first record start, the number of records, the read index
    struct Job { uint64_t k, skip_lines, n_records, first_read; bool at_line_start; };
    std::vector<Job> jobs;
    {
        uint64_t lines_before = 0;                         // newlines before the piece = index of the line its first byte is in
        bool prev_ends_nl = true;                          // (the text before the piece ends with a newline, or there is none)
        for (uint64_t k = 0; k < n_pieces; k++) {
            const bool at_start = prev_ends_nl;
            const uint64_t nl = census[2 * k], last_nl = census[2 * k + 1];
            if (pm.bgzf ? pm.text[k] != 0 : piece_span(src.file_size, piece_bytes, k).size != 0) prev_ends_nl = last_nl != 0;
            else continue;                                 // an empty piece starts no line
            const uint64_t starts = (at_start ? 1 : 0) + nl - (last_nl ? 1 : 0);          // lines that START in the piece
            const uint64_t i0 = at_start ? lines_before : lines_before + 1;              // index of the first of them
            const uint64_t r0 = (i0 + 3) / 4 * 4;                                        // first record start at or after it
            if (k % world == rank && starts > r0 - i0) {
                const uint64_t left = starts - (r0 - i0);
                jobs.push_back(Job{k, r0 - i0, (left + 3) / 4, r0 / 4, at_start});
            }
            lines_before += nl;
        }
    }
    const size_t HEAD = 64, MARGIN = (size_t)1 << 20;       // a record that starts in the piece ends within the margin behind it
    constexpr int NSLOT = 3;
    PinBuf buf[NSLOT];
    auto drop = [&]() { for (auto &b : buf) g_pinned.release(b); };
    const size_t cap = HEAD + (size_t)std::max<uint64_t>(pm.max_text, 4096) + MARGIN + (pm.bgzf ? (size_t)1 << 16 : 0);
    for (auto &b : buf) if (!g_pinned.acquire(cap, b)) { drop(); return fail(c, F2Q_ENOMEM, "cannot allocate the read buffers"); }
    const bool can_stage = !c->host_pack && !getenv("F2Q_NO_STAGING");
    if (can_stage && !c->copy_stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming);
        if (e != hipSuccess) { drop(); return fail(c, F2Q_EHIP, std::string("copy stream: ") + hipGetErrorString(e)); }
    }
    struct Piece { int slot; size_t n, a; uint64_t max_rec, first_read; bool ok; };
    std::mutex mu; std::condition_variable cv;
    std::deque<Piece> ready; bool slot_free[NSLOT]; for (bool &f : slot_free) f = true; bool stop = false;
    std::thread reader([&]() {
        int slot = 0;
        for (size_t j = 0; j <= jobs.size(); j++, slot = (slot + 1) % NSLOT) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return slot_free[slot] || stop; }); if (stop) return; slot_free[slot] = false; }
            Piece pc{slot, 0, 0, 0, 0, true};
            if (j < jobs.size()) {
                const Job &jb = jobs[j];
                PieceSpan sp = piece_span(src.file_size, piece_bytes, jb.k);
                uint64_t want = std::min<uint64_t>(sp.size + MARGIN, src.file_size - sp.base);
                uint8_t *p = buf[slot].p + HEAD;
                size_t n = 0;
                if (!pm.bgzf) { n = src.read_at(sp.base, p, (size_t)want); pc.ok = n == want; }
                else {
                    // the run of members, then members behind it until the last record is whole (4 newlines) or the margin is full
                    sp.base = 0; sp.size = pm.text[jb.k];
                    pc.ok = src.seek_bgzf(pm.c_off[jb.k]);
                    while (pc.ok && n < sp.size) { const size_t g = src.read(p + n, (size_t)(sp.size - n)); if (!g) pc.ok = false; n += g; }
                    size_t seen = 0, o = n;
                    bool more = true;
                    while (pc.ok && more && seen < 4 && n < sp.size + MARGIN) {
                        const size_t g = src.read(p + n, (size_t)std::min<uint64_t>((uint64_t)1 << 16, sp.size + MARGIN - n));
                        if (!g) { more = false; if (src.truncated()) pc.ok = false; break; }
                        n += g;
                        while (seen < 4 && o < n) { const uint8_t *q = (const uint8_t *)memchr(p + o, 0x0a, n - o); if (!q) { o = n; break; } o = (size_t)(q - p) + 1; seen++; }
                    }
                    // (for the check below: "the file goes on behind what was read" <=> the margin filled up without 4 newlines)
                    want = n; sp.base = 0;
                    const bool file_goes_on = more && seen < 4;
                    if (file_goes_on) pc.ok = false;
                }
                // the first record start: the line after the one cut by the piece's start, then `skip_lines` more
                size_t a = 0;
                uint64_t skip = jb.skip_lines + (jb.at_line_start ? 0 : 1);
                while (skip && a < n) { const uint8_t *q = (const uint8_t *)memchr(p + a, 0x0a, n - a); if (!q) { a = n; break; } a = (size_t)(q - p) + 1; skip--; }
                // the margin must hold the rest of the last record (4 more newlines, or the end of the file)
                if (!pm.bgzf && sp.base + want < src.file_size) {
                    size_t seen = 0, o = (size_t)sp.size;
                    while (seen < 4 && o < n) { const uint8_t *q = (const uint8_t *)memchr(p + o, 0x0a, n - o); if (!q) break; o = (size_t)(q - p) + 1; seen++; }
                    if (seen < 4) pc.ok = false;           // lines too long for the margin: the caller falls back
                }
                pc.n = n; pc.a = a; pc.max_rec = jb.n_records; pc.first_read = jb.first_read;
            }
            { std::lock_guard<std::mutex> g(mu); ready.push_back(pc); }
            cv.notify_all();
        }
    });
    struct Staged { void *buf = nullptr; size_t cap = 0; int slot = -1; size_t n = 0; } staged;
    auto unstage = [&]() {
        if (!staged.buf) return;
        (void)hipStreamSynchronize(c->copy_stream);
        std::vector<void *> v{staged.buf}; free_all(c, v);
        staged = Staged();
    };
    auto stage_next = [&]() {
        if (!can_stage || staged.buf) return;
        Piece nx{-1, 0, 0, 0, 0, true};
        { std::unique_lock<std::mutex> lk(mu); if (!ready.empty()) nx = ready.front(); }
        if (nx.slot < 0 || nx.n == 0 || !nx.ok) return;
        void *d = nullptr;
        const size_t dcap = HEAD + nx.n + 2 * (size_t)F2Q_NL_CHUNK + 64;
        if (dev_get(c, dcap, &d) != F2Q_OK) return;
        if (hipMemcpyAsync((uint8_t *)d + HEAD, buf[nx.slot].p + HEAD, nx.n, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
            hipEventRecord(c->ev_copy, c->copy_stream) != hipSuccess) {
            (void)hipGetLastError(); (void)hipStreamSynchronize(c->copy_stream);
            std::vector<void *> v{d}; free_all(c, v);
            return;
        }
        staged.buf = d; staged.cap = dcap; staged.slot = nx.slot; staged.n = nx.n;
    };
    f2q_timing sum; memset(&sum, 0, sizeof sum);
    int rc = F2Q_OK;
    for (size_t j = 0; j <= jobs.size(); j++) {
        Piece pc;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !ready.empty(); }); pc = ready.front(); ready.pop_front(); }
        Staged mine;
        if (staged.buf && staged.slot == pc.slot && staged.n == pc.n) {
            mine = staged; staged = Staged();
            if (hipStreamWaitEvent(c->stream, c->ev_copy, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->copy_stream); }
        } else unstage();
        if (j < jobs.size()) stage_next();
        if (!pc.ok) rc = fail(c, F2Q_EUNSUPPORTED, "a line longer than the piece margin (or a short read): count this file unsharded");
        if (!rc && j < jobs.size() && pc.n > pc.a) {
            f2q_timing one; memset(&one, 0, sizeof one);
            size_t used = 0;
            c->reads_seen = pc.first_read;
            uint8_t *base = buf[pc.slot].p + HEAD + pc.a;
            if (mine.buf) {
                // the text starts `a` bytes into the staged copy; the bytes between the 16-byte boundary below it and the
                // text lengthen the first line, a header line, which is never looked at (fast2q.py:324-328)
                const size_t textoff = HEAD + pc.a, al = textoff & ~(size_t)15, lead = textoff - al;
                uint8_t *d = (uint8_t *)mine.buf;
                hipError_t e = lead ? hipMemsetAsync(d + al, 'x', lead, c->stream) : hipSuccess;
                if (e != hipSuccess) { rc = fail(c, F2Q_EHIP, hipGetErrorString(e)); }
                else {
                    DevText pre; pre.buf = mine.buf; pre.cap = mine.cap; pre.text = d + al; pre.last_byte = base[pc.n - pc.a - 1];
                    rc = count_window(c, nullptr, lead + (pc.n - pc.a), &pre, &used, &one, pc.max_rec);   // the buffer now belongs to the block
                    mine = Staged();
                }
            } else rc = count_window(c, base, pc.n - pc.a, nullptr, &used, &one, pc.max_rec);
            sum.kernel_ms += one.kernel_ms; sum.total_ms += one.total_ms; sum.reads += one.reads;
            sum.fast_reads += one.fast_reads; sum.general_reads += one.general_reads; sum.launches += one.launches; if (one.path) sum.path = one.path;
        }
        if (mine.buf) { (void)hipStreamSynchronize(c->stream); std::vector<void *> v{mine.buf}; free_all(c, v); }
        { std::lock_guard<std::mutex> g(mu); slot_free[pc.slot] = true; if (rc) stop = true; }
        cv.notify_all();
        if (rc) break;
    }
    unstage();
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv.notify_all();
    reader.join();
    drop();
    if (t) *t = sum;
    return rc;
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
    o.planar_nw = planar ? (R <= 96 ? 3u : R <= 160 ? 5u : 10u) : 0u;
    if (c->plan.fast_fixed && c->plan.n_win) {                     // several windows: the tiles hold the windows only
        o.n_win = c->plan.n_win; o.win_len = c->plan.win_len; o.win_end = c->plan.win_end;
        for (int i = 0; i < c->plan.n_win; i++) o.win_start[i] = c->plan.win_start[i];
    }
    const int Rs = o.n_win ? o.n_win * o.win_len : R;               // bases a tile stores per read
    const uint64_t n_tiles = (s->n_reads + F2Q_TILE - 1) / F2Q_TILE;
    const uint64_t n_slots = n_tiles * F2Q_TILE;
    // general-path capacity: everything, or the expected 'N' share with a wide margin
    double pn = (double)d.t_n / 4294967296.0;
    uint64_t gcap = fast ? (uint64_t)((double)s->n_reads * pn * 1.5 + 6.0 * sqrt((double)s->n_reads * pn + 1.0) + 1024.0) : s->n_reads;
    if (gcap > s->n_reads) gcap = s->n_reads;
    if (gcap == 0) gcap = 1;
    do {
        if (fast) {
            o.wb = planar ? 2 * o.planar_nw : (uint32_t)((Rs + 15) / 16);
            o.wq = planar ? 8 * o.planar_nw : (uint32_t)((Rs + 3) / 4);
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
            b->pb.n_slots = n_slots; b->pb.n_tiles = (uint32_t)n_tiles; b->pb.wb = o.wb; b->pb.wq = o.wq; b->pb.rmax = (uint32_t)Rs;
            b->pb.planar_nw = o.planar_nw;
            b->pb.bases = o.bases; b->pb.qual = o.qual; b->pb.len = o.len;
        }
        b->rb.n = g; b->rb.raw = o.raw; b->rb.off = o.off; b->rb.len = o.glen; b->rb.qlen = o.gqlen; b->rb.index = o.gindex;
        b->rb.first_index = 0;
        b->dev_bytes += g * (uint64_t)(2 * R + 20);
        b->raw_key_bytes = g * (uint64_t)R;
    } while (0);
    if (rc) { free_all(c, b->allocs); delete b; return rc; }
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
    if (ctr[2]) return fail(c, F2Q_ENOMEM, "Extract+Count table overflow (internal sizing error, code " + std::to_string(ctr[2]) + ")");
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
            if (off[e] + ((unsigned long long)len[e] + 3) / 4 > ctr[1])
                return fail(c, F2Q_ESTATE, "Extract+Count entry " + std::to_string(e) + " of " + std::to_string(n) + " is not filled in (offset " +
                            std::to_string(off[e]) + ", length " + std::to_string(len[e]) + ", arena " + std::to_string(ctr[1]) + " words)");
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
            char text[32];
            const uint32_t len = ec64_text(ks[i], text);
            h.keys.emplace_back(text, len); h.cnt.push_back(kc[i] + 1ull); h.first.push_back(kf[i]);      // (the word holds n - 1)
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
