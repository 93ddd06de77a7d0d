/*
 * oracle/f2q_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU-only restatement of the read-counting hot path of 2FAST2Q
 * v2.8.1 (reference: fast2q/fast2q.py).  It exists so that the HIP path can be
 * checked bit-for-bit against an independent statement of the reference's
 * semantics.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this file; nothing under 2fast2q_amd/ links or calls it.
 *
 * Pinning (see tests/test_oracle_*.py):
 *   - the reference's own known-answer tests (tests/test_mainfunctions.py:4-78)
 *     are restated against orc_border_finder / orc_sequence_tinder;
 *   - golden vectors under tests/golden/ were produced by calling the
 *     reference's reads_counter() (fast2q.py:514) in the build container
 *     (tests/golden/make_golden.py) and are replayed against orc_count_fastq.
 *
 * Every function cites the reference lines it follows.  The algorithms are the
 * reference's (dict hit, then early-exit all-vs-all scan per mismatch level,
 * with the passed/failed memo caches) so that timing this file is a fair
 * "port" CPU baseline; data structures are plain C.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_ITER 16

typedef struct {
    int32_t mode;            /* 0 = Counter ("C"), 1 = Extract+Count ("EC")   fast2q.py:364,382 */
    int32_t miss;            /* --m                                           fast2q.py:1266    */
    int32_t phred;           /* --ph  (raw CLI value; <=0 coerced to 1)       fast2q.py:1118    */
    int32_t qual_up;         /* --qsu                                         fast2q.py:1121    */
    int32_t qual_down;       /* --qsd                                         fast2q.py:1124    */
    int32_t length;          /* --l                                           fast2q.py:1250    */
    int32_t fixed;           /* 1: --st mode, 0: --us/--ds anchored mode      fast2q.py:536-545 */
    int32_t n_iter;          /* search_iterations                             fast2q.py:541,558 */
    int32_t starts[ORC_MAX_ITER];      /* start_positioning                   fast2q.py:539     */
    int32_t n_up, n_down;    /* 0 = that anchor list is absent (None)                           */
    int32_t msu, msd;        /* --msu / --msd                                 fast2q.py:1278-84 */
    const char *up[ORC_MAX_ITER];      /* upstream anchors (any case; upper-cased here :547)    */
    const char *down[ORC_MAX_ITER];    /* downstream anchors                                    */
} orc_params;

/* ------------------------------------------------------------------------- */
/* small byte-string hash map (insertion ordered) -- stands in for the Python */
/* dict `features` (fast2q.py:174) and the memo dict/set (fast2q.py:1628).    */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint8_t *bytes;      /* arena of key bytes                   */
    size_t   bytes_len, bytes_cap;
    size_t  *off;        /* per entry: offset into arena         */
    uint32_t *len;       /* per entry: key length                */
    int64_t *val;        /* per entry: value (count / feature id)*/
    size_t   n, cap;     /* entries                              */
    int64_t *slot;       /* open addressing: entry index or -1   */
    size_t   nslot;      /* power of two                         */
} orc_map;

static uint64_t fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ULL; }
    return h ^ (h >> 29);
}

static void map_init(orc_map *m)
{
    memset(m, 0, sizeof *m);
    m->nslot = 1024;
    m->slot = (int64_t *)malloc(m->nslot * sizeof(int64_t));
    for (size_t i = 0; i < m->nslot; i++) m->slot[i] = -1;
}

static void map_free(orc_map *m)
{
    free(m->bytes); free(m->off); free(m->len); free(m->val); free(m->slot);
    memset(m, 0, sizeof *m);
}

static int64_t map_find(const orc_map *m, const uint8_t *k, size_t n)
{
    size_t mask = m->nslot - 1, h = (size_t)fnv1a(k, n) & mask;
    for (;;) {
        int64_t e = m->slot[h];
        if (e < 0) return -1;
        if (m->len[e] == n && memcmp(m->bytes + m->off[e], k, n) == 0) return e;
        h = (h + 1) & mask;
    }
}

static void map_rehash(orc_map *m)
{
    size_t ns = m->nslot * 2;
    int64_t *s = (int64_t *)malloc(ns * sizeof(int64_t));
    for (size_t i = 0; i < ns; i++) s[i] = -1;
    for (size_t e = 0; e < m->n; e++) {
        size_t h = (size_t)fnv1a(m->bytes + m->off[e], m->len[e]) & (ns - 1);
        while (s[h] >= 0) h = (h + 1) & (ns - 1);
        s[h] = (int64_t)e;
    }
    free(m->slot); m->slot = s; m->nslot = ns;
}

/* insert (key must be absent); returns entry index */
static int64_t map_add(orc_map *m, const uint8_t *k, size_t n, int64_t v)
{
    if ((m->n + 1) * 2 > m->nslot) map_rehash(m);
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 1024;
        m->off = (size_t *)realloc(m->off, m->cap * sizeof(size_t));
        m->len = (uint32_t *)realloc(m->len, m->cap * sizeof(uint32_t));
        m->val = (int64_t *)realloc(m->val, m->cap * sizeof(int64_t));
    }
    if (m->bytes_len + n + 1 > m->bytes_cap) {
        m->bytes_cap = (m->bytes_cap ? m->bytes_cap * 2 : 65536) + n;
        m->bytes = (uint8_t *)realloc(m->bytes, m->bytes_cap);
    }
    size_t e = m->n++;
    m->off[e] = m->bytes_len; m->len[e] = (uint32_t)n; m->val[e] = v;
    memcpy(m->bytes + m->bytes_len, k, n); m->bytes_len += n;
    size_t mask = m->nslot - 1, h = (size_t)fnv1a(k, n) & mask;
    while (m->slot[h] >= 0) h = (h + 1) & mask;
    m->slot[h] = (int64_t)e;
    return (int64_t)e;
}

/* ------------------------------------------------------------------------- */
/* the oracle context                                                         */
/* ------------------------------------------------------------------------- */
typedef struct {
    orc_params p;
    uint8_t *up[ORC_MAX_ITER];   int up_len[ORC_MAX_ITER];
    uint8_t *down[ORC_MAX_ITER]; int down_len[ORC_MAX_ITER];
    orc_map features;       /* Counter: library (val = count); EC: de-novo dict (val = count) */
    orc_map passed;         /* memo: seq -> feature entry   fast2q.py:728,741 */
    orc_map failed;         /* memo: seq set                fast2q.py:724,748 */
    int use_memo;
    int64_t stats[5];       /* reads, perfect, imperfect, non_aligned, quality_failed :310-316 */
} orc_ctx;

static uint8_t *dup_upper(const char *s, int *n)
{
    size_t l = strlen(s);
    uint8_t *r = (uint8_t *)malloc(l + 1);
    for (size_t i = 0; i < l; i++) {
        uint8_t c = (uint8_t)s[i];
        r[i] = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;   /* n.upper()  fast2q.py:547,550 */
    }
    r[l] = 0; *n = (int)l;
    return r;
}

orc_ctx *orc_create(const orc_params *p, int use_memo)
{
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
    c->p = *p;
    /* fast2q.py:1118-1125: non-positive thresholds are coerced to 1 */
    if (c->p.phred <= 0) c->p.phred = 1;
    if (c->p.qual_up <= 0) c->p.qual_up = 1;
    if (c->p.qual_down <= 0) c->p.qual_down = 1;
    for (int i = 0; i < p->n_up; i++) c->up[i] = dup_upper(p->up[i], &c->up_len[i]);
    for (int i = 0; i < p->n_down; i++) c->down[i] = dup_upper(p->down[i], &c->down_len[i]);
    if (!c->p.fixed) {
        /* fast2q.py:558 search_iterations = max(len_up, len_down) */
        c->p.n_iter = p->n_up > p->n_down ? p->n_up : p->n_down;
    }
    map_init(&c->features); map_init(&c->passed); map_init(&c->failed);
    c->use_memo = use_memo;
    return c;
}

void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    for (int i = 0; i < ORC_MAX_ITER; i++) { free(c->up[i]); free(c->down[i]); }
    map_free(&c->features); map_free(&c->passed); map_free(&c->failed);
    free(c);
}

/* Library ingest: sequences arrive already upper-cased / de-duplicated by the
 * harness side (features_loader, fast2q.py:148-166); a repeated sequence is
 * ignored here too (first one wins, :160-165). Returns the feature's index or
 * -1 when it was a duplicate. */
int64_t orc_add_feature(orc_ctx *c, const uint8_t *seq, uint32_t n)
{
    if (map_find(&c->features, seq, n) >= 0) return -1;
    return map_add(&c->features, seq, n, 0);
}

/* ------------------------------------------------------------------------- */
/* Phred rule -- initializer(), fast2q.py:1112-1129.                           */
/* quality_list = chr(33)..chr(126); fail set = its first (ph-1) characters,   */
/* i.e. a byte fails iff 33 <= c <= min(ph+31, 126).  Q(ph-1) passes.          */
/* ------------------------------------------------------------------------- */
static int phred_fails(const uint8_t *q, long a, long b, int ph)
{
    int hi = ph + 31; if (hi > 126) hi = 126;
    for (long i = a; i < b; i++) if (q[i] >= 33 && q[i] <= hi) return 1;
    return 0;
}

/* Python slice bounds x[a:b] on a sequence of length n (negative indices wrap
 * once, then clamp) -- needed because down-only anchoring can produce a
 * negative start (fast2q.py:282 then :354). */
static void py_slice(long n, long a, long b, long *oa, long *ob)
{
    if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
    if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
    if (b < a) b = a;
    *oa = a; *ob = b;
}

/* binary_subtract, fast2q.py:601-626: 1 iff the zip()-paired bytes differ in at
 * most `mismatch` places (zip stops at the shorter operand). */
static int binary_subtract(const uint8_t *a, long na, const uint8_t *b, long nb, int mismatch)
{
    long n = na < nb ? na : nb;
    int miss = 0;
    for (long i = 0; i < n; i++) {
        if (a[i] != b[i]) miss++;
        if (miss > mismatch) return 0;
    }
    return 1;
}

/* border_finder, fast2q.py:628-658: first position p in [start_place, r-s]
 * whose s-byte window is within `mismatch` of seq; -1 when none. */
long orc_border_finder(const uint8_t *seq, long s, const uint8_t *read, long r,
                       int mismatch, long start_place)
{
    long fall_over_index = r - s;
    /* enumerate(read[start_place:]) -- an out-of-range start gives an empty slice */
    long a, b; py_slice(r, start_place, r, &a, &b);
    for (long i = 0; i < b - a; i++) {
        long p = start_place + i;            /* the reference indexes with start_place+i (:653) */
        long ca, cb; py_slice(r, p, s + p, &ca, &cb);
        int finder = binary_subtract(seq, s, read + ca, cb - ca, mismatch);
        if (p > fall_over_index) return -1;  /* :655-656 */
        if (finder) return p;                /* :657-658 */
    }
    return -1;
}

/* sequence_tinder, fast2q.py:215-285.  Returns 1 and (start,end) or 0 for
 * (None,None).  `read` is the raw-case sequence line (:337), `qual` the quality
 * line; anchors were upper-cased at :547/:550. */
int orc_sequence_tinder(const orc_ctx *c, const uint8_t *read, long r,
                        const uint8_t *qual, long qn, int i, long *ostart, long *oend)
{
    const orc_params *p = &c->p;
    long start, end, a, b;
    if (p->n_up && p->n_down) {                                   /* :240 */
        start = orc_border_finder(c->up[i], c->up_len[i], read, r, p->msu, 0);
        if (start < 0) return 0;
        end = orc_border_finder(c->down[i], c->down_len[i], read, r, p->msd,
                                start + c->up_len[i]);           /* :246-249 */
        if (end < 0) return 0;
        py_slice(qn, start, start + c->up_len[i], &a, &b);        /* :252 */
        if (phred_fails(qual, a, b, p->qual_up)) return 0;
        py_slice(qn, end, end + c->down_len[i], &a, &b);          /* :253 */
        if (phred_fails(qual, a, b, p->qual_down)) return 0;
        *ostart = start + c->up_len[i]; *oend = end;              /* :257-258 */
        return 1;
    } else if (p->n_up) {                                         /* :260 */
        start = orc_border_finder(c->up[i], c->up_len[i], read, r, p->msu, 0);
        if (start < 0) return 0;
        py_slice(qn, start, start + c->up_len[i], &a, &b);        /* :266 */
        if (phred_fails(qual, a, b, p->qual_up)) return 0;
        *ostart = start + c->up_len[i];                           /* :269 */
        *oend = *ostart + p->length;                              /* :270 */
        return 1;
    } else if (p->n_down) {                                       /* :273 */
        end = orc_border_finder(c->down[i], c->down_len[i], read, r, p->msd, 0);
        if (end < 0) return 0;
        py_slice(qn, end, end + c->down_len[i], &a, &b);          /* :279 */
        if (phred_fails(qual, a, b, p->qual_down)) return 0;
        *ostart = end - p->length; *oend = end;                   /* :282-283 (start may be < 0) */
        return 1;
    }
    return 0;
}

/* features_all_vs_all, fast2q.py:660-690: scan every same-length feature in
 * dict order; returns the entry iff exactly one is within `mismatch`, bailing
 * out at the second hit. */
static int64_t features_all_vs_all(const orc_map *f, const uint8_t *read, uint32_t r, int mismatch)
{
    int found = 0; int64_t found_guide = -1;
    for (size_t e = 0; e < f->n; e++) {
        if (f->len[e] != r) continue;                             /* :683 */
        if (binary_subtract(f->bytes + f->off[e], r, read, r, mismatch)) {
            found++; found_guide = (int64_t)e;
            if (found >= 2) return -1;                            /* :687-688 */
        }
    }
    return found == 1 ? found_guide : -1;
}

/* mismatch_search_handler, fast2q.py:692-750 */
static void mismatch_search(orc_ctx *c, const uint8_t *seq, uint32_t n)
{
    if (c->use_memo) {
        if (map_find(&c->failed, seq, n) >= 0) { c->stats[3]++; return; }     /* :724-726 */
        int64_t pe = map_find(&c->passed, seq, n);
        if (pe >= 0) { c->features.val[c->passed.val[pe]]++; c->stats[2]++; return; } /* :728-731 */
    }
    for (int miss = 1; miss <= c->p.miss; miss++) {                            /* :734 */
        int64_t f = features_all_vs_all(&c->features, seq, n, miss);
        if (f >= 0) {                                                          /* :738-742 */
            c->features.val[f]++; c->stats[2]++;
            if (c->use_memo) map_add(&c->passed, seq, n, f);
            return;
        }
    }
    if (c->use_memo) map_add(&c->failed, seq, n, 0);                           /* :747-748 */
    c->stats[3]++;                                                             /* :749 */
}

static long rstrip_len(const uint8_t *p, long n)
{
    /* bytes.rstrip(): trailing ASCII whitespace  (fast2q.py:326) */
    while (n > 0) {
        uint8_t ch = p[n - 1];
        if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == 0x0b || ch == 0x0c) n--;
        else break;
    }
    return n;
}

/* one FASTQ record (seq line, quality line) -- the body of the `len(reading)==4`
 * block, fast2q.py:328-393 */
static void process_record(orc_ctx *c, const uint8_t *seq, long r, const uint8_t *qual, long qn,
                           uint8_t *scratch)
{
    const orc_params *p = &c->p;
    int all_failed = 1;
    long flen = 0; int have_feature = 0;          /* full_feature (:332) as bytes in scratch */
    for (int i = 0; i < p->n_iter; i++) {
        long start = 0, end = 0; int ok = 1;
        if (!p->fixed) {
            ok = orc_sequence_tinder(c, seq, r, qual, qn, i, &start, &end);    /* :337 */
            if (ok && end < start) ok = 0;                                     /* :343-345 */
        } else {
            start = p->starts[i]; end = (long)p->starts[i] + p->length;        /* :350-351, :540 */
        }
        if (!ok) continue;                                                     /* flag stays 1 (:347) */
        long a, b, qa, qb;
        py_slice(r, start, end, &a, &b);                                       /* :354 */
        py_slice(qn, start, end, &qa, &qb);                                    /* :355 */
        if (phred_fails(qual, qa, qb, p->phred)) continue;                     /* :357-360 */
        all_failed = 0;
        scratch[flen++] = ':';                                                 /* :358 */
        for (long k = a; k < b; k++) {
            uint8_t ch = seq[k];
            scratch[flen++] = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch;  /* .upper() :354 */
        }
        have_feature = 1;
    }
    if (have_feature) {                                                        /* :362 */
        const uint8_t *key = scratch + 1; uint32_t kn = (uint32_t)(flen - 1);  /* :363 */
        int64_t e = map_find(&c->features, key, kn);
        if (p->mode == 0) {
            if (e >= 0) { c->features.val[e]++; c->stats[1]++; }               /* :365-367 */
            else if (p->miss > 0) mismatch_search(c, key, kn);                 /* :369-378 */
            else c->stats[3]++;                                                /* :380 */
        } else {
            if (e < 0) map_add(&c->features, key, kn, 1);                      /* :383-384 */
            else c->features.val[e]++;                                         /* :386 */
            c->stats[1]++;                                                     /* :387 */
        }
    }
    if (all_failed) c->stats[4]++;                                             /* :389-390 */
    c->stats[0]++;                                                             /* :393 */
}

/* fastq_parser framing, fast2q.py:324-328,392: every 4 rstrip()-ed lines are a
 * record; lines 2 and 4 are used; a trailing partial record is dropped.
 * Returns the number of bytes consumed up to the last complete record so a
 * caller can stream blocks. */
int64_t orc_count_fastq(orc_ctx *c, const uint8_t *buf, int64_t nbytes)
{
    const uint8_t *lines[4]; long lens[4]; int nl = 0;
    int64_t pos = 0, consumed = 0;
    size_t scratch_cap = 0; uint8_t *scratch = NULL;
    while (pos < nbytes) {
        const uint8_t *nlp = (const uint8_t *)memchr(buf + pos, '\n', (size_t)(nbytes - pos));
        int64_t eol = nlp ? (int64_t)(nlp - buf) : nbytes;
        lines[nl] = buf + pos; lens[nl] = rstrip_len(buf + pos, (long)(eol - pos)); nl++;
        pos = nlp ? eol + 1 : nbytes;
        if (nl == 4) {
            size_t need = (size_t)(lens[1] + 2) * (size_t)(c->p.n_iter > 0 ? c->p.n_iter : 1) + 16;
            if (need > scratch_cap) { scratch_cap = need * 2; scratch = (uint8_t *)realloc(scratch, scratch_cap); }
            process_record(c, lines[1], lens[1], lines[3], lens[3], scratch);
            nl = 0; consumed = pos;
        }
    }
    free(scratch);
    return consumed;
}

/* ------------------------------------------------------------------------- */
/* result access                                                              */
/* ------------------------------------------------------------------------- */
void orc_get_stats(const orc_ctx *c, int64_t out[5]) { memcpy(out, c->stats, sizeof c->stats); }
int64_t orc_n_keys(const orc_ctx *c) { return (int64_t)c->features.n; }
int64_t orc_key_bytes(const orc_ctx *c) { return (int64_t)c->features.bytes_len; }
void orc_get_counts(const orc_ctx *c, int64_t *out)
{
    for (size_t e = 0; e < c->features.n; e++) out[e] = c->features.val[e];
}
/* keys in insertion order: concatenated bytes + n+1 offsets */
void orc_get_keys(const orc_ctx *c, uint8_t *bytes, int64_t *offs)
{
    memcpy(bytes, c->features.bytes, c->features.bytes_len);
    for (size_t e = 0; e < c->features.n; e++) offs[e] = (int64_t)c->features.off[e];
    offs[c->features.n] = (int64_t)c->features.bytes_len;
}
void orc_reset(orc_ctx *c)
{
    memset(c->stats, 0, sizeof c->stats);
    if (c->p.mode == 0) { for (size_t e = 0; e < c->features.n; e++) c->features.val[e] = 0; }
    else { map_free(&c->features); map_init(&c->features); }
}
void orc_merge_counts(orc_ctx *dst, const orc_ctx *src)   /* Counter mode only: sum shard results */
{
    for (size_t e = 0; e < dst->features.n && e < src->features.n; e++) dst->features.val[e] += src->features.val[e];
    for (int i = 0; i < 5; i++) dst->stats[i] += src->stats[i];
}

/* helper for the multi-threaded baseline driver: byte offset just after the
 * n_lines-th '\n' at or after `from` (or nbytes when the buffer ends first) */
int64_t orc_skip_lines(const uint8_t *buf, int64_t nbytes, int64_t from, int64_t n_lines)
{
    int64_t pos = from;
    while (n_lines > 0 && pos < nbytes) {
        const uint8_t *nlp = (const uint8_t *)memchr(buf + pos, '\n', (size_t)(nbytes - pos));
        if (!nlp) return nbytes;
        pos = (int64_t)(nlp - buf) + 1; n_lines--;
    }
    return pos;
}
int64_t orc_count_lines(const uint8_t *buf, int64_t nbytes)
{
    int64_t n = 0, pos = 0;
    while (pos < nbytes) {
        const uint8_t *nlp = (const uint8_t *)memchr(buf + pos, '\n', (size_t)(nbytes - pos));
        if (!nlp) { n++; break; }
        pos = (int64_t)(nlp - buf) + 1; n++;
    }
    return n;
}
