/*
 * include/f2q.h -- C ABI of libf2q_hip.so, the MI355X-native read-counting path of 2FAST2Q.
 *
 * The reference (afombravo/2FAST2Q v2.8.1) is a pure-Python program with no FFI; the seam this
 * library sits behind is the Python call
 *     reads_counter(i, raw, features, param, reads_stats) -> (features, reads_stats, local_read_stats)
 * (fast2q/fast2q.py:514-582) made once per FASTQ file by aligner() (:762).  Everything below
 * that call -- fastq_parser (:306-409), sequence_tinder (:215-285), border_finder (:628-658),
 * binary_subtract (:601-626), features_all_vs_all (:660-690), mismatch_search_handler (:692-750),
 * the Features counters (:21-44) and the chunk pool single_file_reads_binner (:411-512) -- is
 * replaced by the entry points declared here.  2fast2q_amd/fast2q.py binds them with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * F2Q_E* code, with text from f2q_last_error(); the caller owns every buffer it passes; the
 * library owns the context and all device memory; a context is bound to one HIP device and is not
 * thread-safe; there is no global state, so several contexts may coexist.  There is NO CPU
 * fallback: without a usable HIP device f2q_create fails with F2Q_ENODEVICE.
 */
#ifndef F2Q_H
#define F2Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F2Q_ABI_VERSION 1
#define F2Q_MAX_ITER 16          /* --st values / --us,--ds pairs per run                     */

enum {
    F2Q_OK = 0,
    F2Q_EINVAL = -1,             /* bad argument                                              */
    F2Q_ENODEVICE = -2,          /* no HIP device / kernel image not loadable                 */
    F2Q_EHIP = -3,               /* a HIP runtime call failed                                 */
    F2Q_ENOMEM = -4,
    F2Q_EIO = -5,                /* file could not be opened / read                           */
    F2Q_ETRUNCATED = -6,         /* corrupted or truncated gzip stream (partial counts kept,
                                    as fast2q.py:405-407,580-582)                             */
    F2Q_ESTATE = -7,             /* call order (e.g. counting before f2q_set_features)        */
    F2Q_EUNSUPPORTED = -8        /* input outside what the device path implements             */
};

/* index into the stats vector == the keys of local_read_stats (fast2q.py:310-316) */
enum { F2Q_READS = 0, F2Q_PERFECT = 1, F2Q_IMPERFECT = 2, F2Q_NON_ALIGNED = 3, F2Q_QUALITY_FAILED = 4 };

typedef struct f2q_ctx f2q_ctx;

/* The subset of the reference's `param` dict (fast2q.py:1226-1309) that the path reads. */
typedef struct {
    int32_t mode;                        /* 0 = "C" Counter, 1 = "EC" Extract+Count (:364,:382) */
    int32_t miss;                        /* --m   allowed mismatches per feature  (:1266)       */
    int32_t phred;                       /* --ph  raw CLI value; <=0 behaves as 1 (:1118)       */
    int32_t length;                      /* --l   feature length                  (:1250)       */
    int32_t n_start;                     /* number of --st values (fixed mode)    (:539)        */
    int32_t start[F2Q_MAX_ITER];         /* --st values                                         */
    int32_t n_upstream;                  /* number of --us sequences, 0 = None    (:546-548)    */
    int32_t n_downstream;                /* number of --ds sequences, 0 = None    (:549-551)    */
    const char *upstream[F2Q_MAX_ITER];  /* NUL-terminated, any case (upper-cased inside, :547) */
    const char *downstream[F2Q_MAX_ITER];
    int32_t miss_search_up;              /* --msu (:1278) */
    int32_t miss_search_down;            /* --msd (:1282) */
    int32_t qual_up;                     /* --qsu (:1286) */
    int32_t qual_down;                   /* --qsd (:1290) */
    int32_t device;                      /* HIP device ordinal this context drives              */
    int32_t reserved[7];
} f2q_params;

/* Device-side synthetic workload of SURVEY.md §8(d) (spec: tests/synth.py). Probabilities are
 * cumulative 32-bit thresholds (p * 2^32). */
typedef struct {
    uint64_t seed;
    uint64_t n_reads;
    uint64_t first_read;                 /* global index of read 0 of this block (sharding)     */
    int32_t read_len;
    int32_t start;                       /* window position when cassette == 0                  */
    int32_t cassette;                    /* 1: up+guide+down at a uniform offset in [0,max_offset] */
    int32_t max_offset;
    const char *up;                      /* cassette flanks (may be NULL when cassette == 0)    */
    const char *down;
    uint32_t t_sub, t_rand, t_n, t_lowq, t_q29, t_q28;
    int32_t reserved[4];
} f2q_synth;

typedef struct {
    double kernel_ms;                    /* HIP-event time of the counting kernels of the call  */
    double total_ms;                     /* HIP-event time of the whole call on its stream      */
    uint64_t reads;                      /* reads the call processed                            */
    uint64_t fast_reads;                 /* ... of which went through the packed fast path      */
    uint64_t general_reads;              /* ... of which went through the general path          */
    uint32_t launches;                   /* kernel launches in the call                         */
    uint32_t path;                       /* kernel family that counted the packed tiles of the call (F2Q_PATH_*, diagnostics) */
} f2q_timing;
/* f2q_timing.path */
enum {
    F2Q_PATH_NONE = 0,                   /* no packed tiles / not recorded                        */
    F2Q_PATH_FIXED_V1 = 1,               /* one read per lane, wide tables                        */
    F2Q_PATH_FIXED_PACKED = 2,           /* k_count_fixed4: packed tables in L2                   */
    F2Q_PATH_FIXED_LDS = 3,              /* k_count_fixed4_lds: the library in LDS                */
    F2Q_PATH_FIXED_PART = 4,             /* k_part_*: a large library dealt into LDS-sized partitions */
    F2Q_PATH_MULTI = 5,                  /* k_count_multi4                                        */
    F2Q_PATH_ANCHOR = 6,                 /* k_count_anchor                                        */
    F2Q_PATH_ANCHOR_LDS = 7,             /* k_count_anchor_lt                                     */
    F2Q_PATH_PAIRS = 8,                  /* k_count_anchor_pairs                                  */
    F2Q_PATH_EXTRACT = 9,                /* the Extract+Count kernels                             */
    F2Q_PATH_MULTI_LDS = 10              /* k_count_fixed4_lds<.., MW>: several windows, joined keys in the LDS tables */
};

typedef struct f2q_block f2q_block;      /* a device-resident block of reads (opaque)           */

/* ---- lifetime --------------------------------------------------------------------------- */
int f2q_version(void);
/* 16 hex digits identifying the sources this binary was compiled from (SHA-256 over csrc/ and this header, computed
 * by __graft_entry__.build() and passed as -DF2Q_BUILD_ID); "unknown" for a build made any other way.  smoke() and
 * the GPU tests compare it with the tree they run from, so a stale binary cannot pass for HEAD. */
const char *f2q_build_id(void);
/* Replaces the per-file set-up half of reads_counter (fast2q.py:536-558): resolves fixed vs
 * anchored mode and search_iterations, builds the Phred fail thresholds (initializer :1112-1129). */
int f2q_create(const f2q_params *p, f2q_ctx **out);
void f2q_destroy(f2q_ctx *ctx);
const char *f2q_last_error(const f2q_ctx *ctx);   /* ctx may be NULL: last f2q_create error   */

/* ---- library (Counter mode) ---------------------------------------------------------------
 * Replaces binary_converter (fast2q.py:188-213) and the `features` dict as a lookup structure.
 * seqs/offs: n sequences concatenated, offs has n+1 entries; already upper-cased and
 * de-duplicated in loader order by features_loader (:148-166), which stays in Python.
 * Feature i <-> row i of the count vector. */
int f2q_set_features(f2q_ctx *ctx, const char *seqs, const uint32_t *offs, uint32_t n);

/* ---- counting ------------------------------------------------------------------------------
 * All counting entry points ACCUMULATE into the context's device-resident count vector and
 * 5 stats (like Features.counts += 1, fast2q.py:366); read them with f2q_read_counts. */

/* fastq_parser over an in-memory FASTQ buffer (fast2q.py:324-393): 4-line framing with
 * rstrip(), trailing partial record ignored.  *consumed (optional) = bytes up to the end of the
 * last complete record so a caller can stream a file in blocks. */
int f2q_count_block(f2q_ctx *ctx, const uint8_t *fastq, size_t nbytes, size_t *consumed, f2q_timing *t);
/* The same for FASTQ text that is ALREADY in device memory: f2q_text_upload copies at most 1 GiB of text once;
 * f2q_count_text frames, packs and counts it on the device (the ingest kernels + the counting kernels, no host-to-device
 * copy in the call; accumulating like f2q_count_block, *consumed = bytes up to the end of the last complete record) and
 * may be called any number of times.  What the tile layout costs to produce, without PCIe in the way (bench.py:
 * end_to_end.device_text_to_counts); a producer that fills device memory itself would enter here.  The read-side half of
 * reads_counter, fast2q.py:560-578, once the bytes are on the device. */
typedef struct f2q_text f2q_text;
int f2q_text_upload(f2q_ctx *ctx, const uint8_t *fastq, size_t nbytes, f2q_text **out);
int f2q_count_text(f2q_ctx *ctx, f2q_text *text, size_t *consumed, f2q_timing *t);
void f2q_text_free(f2q_ctx *ctx, f2q_text *text);
/* reads_counter's file half (fast2q.py:560-578): plain or .gz FASTQ by path (gzip by content; blocked gzip --
 * BGZF -- is inflated member-parallel).  The file is streamed in pieces by a reader thread while the device
 * frames, packs and counts.  F2Q_ETRUNCATED: the archive is cut off or damaged; every complete line before the damage has
 * been counted, as the reference does (its parser keeps what it counted when readline raises, fast2q.py:405-407, and
 * reads_counter returns those partial counts with a warning; the cut-off last line is never seen).  The harness too. */
int f2q_count_file(f2q_ctx *ctx, const char *path, f2q_timing *t);
/* The same file counted by `world` processes, one per GPU (replaces the chunk pool of
 * single_file_reads_binner, fast2q.py:447-512): every rank streams the whole file -- the 4-line framing is
 * global -- counts the pieces k with k % world == rank on its device and only frames the others (a newline
 * census on the host); read indices stay global, so Extract+Count tables merge by key with min(first).
 * The sum over the ranks of counts and stats equals f2q_count_file's. */
int f2q_count_file_shard(f2q_ctx *ctx, const char *path, uint32_t rank, uint32_t world, f2q_timing *t);

/* The same job without any rank reading -- or inflating -- another rank's share (regular files, plain or BGZF): the
 * file is cut into pieces, piece k belongs to rank k % world.  Plain: byte ranges of piece_bytes (4096 ... 1 GiB - 2 MiB:
 * a piece and its look-ahead are framed with 32-bit offsets; F2Q_EINVAL outside that range).  BGZF: runs
 * of whole members whose text adds up to at most piece_bytes (every member carries its compressed size in the header
 * and its text size in the trailer, so the cut needs no inflating); a rank inflates only its own runs -- once for the
 * census, once to count -- plus the few members behind a run that finish its last record.  The 4-line framing is
 * global, so the ranks first exchange how many lines each piece holds:
 *   1. f2q_file_pieces   -- number of pieces; *shardable = 1 plain, 2 BGZF, 0 for ordinary gzip / pipes / BGZF with
 *                           other members inside (use f2q_count_file_shard there)
 *   2. f2q_census_pieces -- census[2k] = newlines of piece k, census[2k+1] = 1 if it ends with one, for THIS rank's pieces
 *                           (the other entries are left as they are: pass a zeroed vector)
 *   3. the caller sums the census vectors over the ranks (one all-reduce of 2 * n_pieces uint64)
 *   4. f2q_count_pieces  -- counts the records whose first line starts in this rank's pieces (global read indices)
 * F2Q_EUNSUPPORTED from step 4: a line longer than the 1 MiB look-ahead behind a piece; nothing usable was counted on
 * this rank -- reset and use f2q_count_file_shard.  Errors of the two context-free calls: f2q_last_error(NULL). */
int f2q_file_pieces(const char *path, uint64_t piece_bytes, uint64_t *n_pieces, int *shardable);
int f2q_census_pieces(const char *path, uint32_t rank, uint32_t world, uint64_t piece_bytes, uint64_t *census, uint64_t n_pieces);
int f2q_count_pieces(f2q_ctx *ctx, const char *path, uint32_t rank, uint32_t world, uint64_t piece_bytes,
                     const uint64_t *census, uint64_t n_pieces, f2q_timing *t);

/* Device-resident blocks: the roofline entry points.  f2q_synth_create generates the §8(d)
 * reads on the device straight into the packed tile layout; f2q_block_from_fastq packs a host
 * FASTQ buffer the same way f2q_count_block does but keeps it resident; f2q_count_resident runs
 * the hot path over a resident block (this is what bench.py times). */
int f2q_synth_create(f2q_ctx *ctx, const f2q_synth *spec, f2q_block **out);
int f2q_block_from_fastq(f2q_ctx *ctx, const uint8_t *fastq, size_t nbytes, f2q_block **out);
int f2q_count_resident(f2q_ctx *ctx, const f2q_block *blk, f2q_timing *t);
/* The same without waiting: the launches are queued on the context's stream and the step gets its own pair of HIP
 * events; f2q_queued_times waits for the stream, returns the kernel time of every step queued since the last call
 * (kernel_ms[0 .. min(*n, cap))) and forgets them.  For callers that queue block after block (bench.py's timed loop). */
int f2q_count_resident_queued(f2q_ctx *ctx, const f2q_block *blk);
int f2q_queued_times(f2q_ctx *ctx, float *kernel_ms, uint32_t cap, uint32_t *n);
int f2q_block_info(const f2q_block *blk, uint64_t *n_reads, uint64_t *n_general, uint64_t *device_bytes);
void f2q_block_free(f2q_ctx *ctx, f2q_block *blk);

/* The guide set the generator plants into reads: by default the library given to f2q_set_features;
 * Extract+Count contexts (which take no library) and tests set it explicitly. n*length ACGT bytes. */
int f2q_synth_guides(f2q_ctx *ctx, const char *seqs, uint32_t n, uint32_t length);
/* Host-side twin of the device generator: FASTQ text of reads [lo,hi) of `spec` against the
 * library given to f2q_set_features. Two-call pattern: buf == NULL returns the size in *nbytes. */
int f2q_synth_fastq(f2q_ctx *ctx, const f2q_synth *spec, uint64_t lo, uint64_t hi, uint8_t *buf, size_t *nbytes);
/* n unique uniform ACGT strings of `length` bases (tests/synth.py make_library); out = n*length bytes */
int f2q_synth_library(uint64_t seed, uint32_t n, uint32_t length, char *out);

/* Global index of the next block's first read (default: running count of reads seen).  Only the
 * first-occurrence order of Extract+Count keys depends on it; sharded callers set it per block. */
int f2q_set_read_base(f2q_ctx *ctx, uint64_t first_read_index);

/* ---- results -------------------------------------------------------------------------------- */
int f2q_reset_counts(f2q_ctx *ctx);
/* Counter mode: counts[n_features] + stats[5] (device -> host, synchronises the stream). */
int f2q_read_counts(f2q_ctx *ctx, int64_t *counts, int64_t stats[5]);
/* Device address of the int64[n_features + 5] accumulator (counts then stats), so a caller can
 * all-reduce it in place over RCCL (torch.distributed) before reading it back.  No synchronisation is done here:
 * work on the accumulator must be ordered after the context's stream (f2q_stream), e.g. by issuing it on that stream. */
int f2q_counts_device_ptr(f2q_ctx *ctx, void **dptr, uint64_t *n_int64);
/* The HIP stream (hipStream_t) all work of this context is launched on. */
void *f2q_stream(f2q_ctx *ctx);

/* Extract+Count results (the de-novo dict of fast2q.py:382-387). Two-call pattern: sizes first,
 * then the caller allocates keys[n_bytes], offs[n_keys+1], counts[n_keys], first_read[n_keys]
 * (index of the first read that produced the key, so the caller can restore dict order). */
int f2q_ec_size(f2q_ctx *ctx, uint64_t *n_keys, uint64_t *n_bytes);
int f2q_ec_fetch(f2q_ctx *ctx, char *keys, uint64_t *offs, int64_t *counts, uint64_t *first_read);

#ifdef __cplusplus
}
#endif
#endif /* F2Q_H */
