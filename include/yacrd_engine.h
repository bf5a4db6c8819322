/*
 * yacrd_engine.h — C ABI of the MI355X bad-region engine (libyacrd_hip.so).
 *
 * This is the drop-in boundary for natir/yacrd's `trait BadPart` (reference src/stack.rs:35-41):
 * a Rust `struct FromGpu: BadPart` calls yacrd_engine_run() from
 * `compute_all_bad_part` (src/stack.rs:143-162, called once from src/main.rs:78) and answers
 * `get_bad_part` (src/stack.rs:164-169) / `get_reads` (:171-173) from the returned CSR.
 *
 * Data contract (CSR over reads, the device-side replacement of
 * `MapReads2Ovl = FxHashMap<String,(Vec<(u32,u32)>,usize)>`, src/reads2ovl/mod.rs:41):
 *   offsets   u64[R+1]  prefix sums of intervals per read (offsets[0] = 0)
 *   intervals u32[2*I]  (start,end) pairs in any order within a read (src/io.rs:23-34 cols
 *                       3-4 / 8-9; both sides of each overlap line, src/reads2ovl/mod.rs:108-109)
 *   lengths   u32[R]    first length seen per read (src/reads2ovl/fullmemory.rs:82-90)
 * Results:
 *   bad_offsets u64[R+1], bad_regions u32[2*G] (begin,end) pairs in the reference's order
 *   (src/stack.rs:61-139), read_type u8[R] (src/editor/mod.rs:85-100).
 *
 * Conventions: every function returns 0 on success or a YACRD_E* code; the message for the
 * last error on the calling thread is yacrd_last_error().  Calls on one engine must not
 * overlap; different engines (one per GPU) may be driven from different threads/processes.
 * Plain C types only — no torch, no HIP types in the signatures.
 *
 * Deviations from the reference, all loud: read lengths must fit u32 (the reference keeps
 * usize and truncates with `as u32` when emitting, src/stack.rs:112); `coverage` is u32 (the
 * CLI's u64, src/cli.rs:53-54, saturates — a read never has 2^32 intervals here either).
 */
#ifndef YACRD_ENGINE_H
#define YACRD_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YACRD_ABI_VERSION 7 /* 7: yacrd_timing.predicted / prediction_misses / build_switches / sorting_build; 6: yacrd_engines_ingest_overlaps[_mem]; 5: YACRD_F_ONE_LAUNCH, yacrd_timing.one_launch; 4: yacrd_timing.screen_items, yacrd_engine_ingest_overlaps_mem */

/* src/editor/mod.rs:42-59 ReadType; numeric encoding is ours, names are the reference's. */
enum { YACRD_NOT_BAD = 0, YACRD_CHIMERIC = 1, YACRD_NOT_COVERED = 2 };

enum {
    YACRD_OK = 0,
    YACRD_EINVAL = 1,  /* bad argument / malformed CSR */
    YACRD_ENODEV = 2,  /* no usable gfx950 device, or HIP runtime error */
    YACRD_ENOMEM = 3,  /* host or device allocation failed */
    YACRD_EINTERNAL = 4,
    YACRD_EFALLBACK = 5 /* yacrd_engine_ingest_paf: the input is not for the device parser; use the host parser */
};

typedef struct yacrd_engine yacrd_engine;

typedef struct {
    int32_t device_id;  /* HIP device ordinal; -1 = the calling thread's current device */
    uint32_t flags;     /* YACRD_F_* */
} yacrd_engine_cfg;

#define YACRD_F_DEFAULT 0u
/* record HIP events around every phase and every class kernel (costs ~3 us of stream time per
 * event); by default only the dominant class kernel is timed */
#define YACRD_F_TIMING_FULL 32u
/* record no HIP events at all (yacrd_timing stays 0): what a caller that only wants results uses */
#define YACRD_F_NO_TIMING 1024u
/* the run's final wait polls an event and sleeps in between instead of spinning in
 * hipStreamSynchronize: for several engines per CPU core (pipelined batches, many GPUs).
 * Engines created on the same device pipeline their batches (yacrd_engine_submit / _collect from
 * one host thread, or one host thread each). */
#define YACRD_F_BLOCKING_WAIT 2048u
/* the dominant kernel carries its start / stop events on every 8th run of the engine only (counted
 * from its creation or the last yacrd_engine_timing_total(reset), whose next run is a timed one): attached
 * events cost ~10 us per batch (host + stream) against a 20 us kernel; yacrd_timing.timed_runs says how
 * many runs were timed */
#define YACRD_F_TIMING_SAMPLED 32768u
/* Latency over throughput, for callers that run ONE short batch at a time (the CLI once its input is in HBM): a batch of
 * fewer than 400 000 reads whose reads all have at most 256 intervals runs as one kernel launch — no size-class plan, no
 * class counts, no prediction, one dispatch instead of three (csrc/one_batch.h: 100 000 reads in 51-55 us instead of 64-66).
 * Any other batch takes the default path, as does a batch in which that launch met a read it does not handle (found on
 * the device; the batch is then run again).  Callers that keep several batches in flight are better off without. */
#define YACRD_F_ONE_LAUNCH 4194304u
/* (the A/B and test switches that pin a kernel family or a build live in yacrd_engine_debug.h; the
 * product path is flags = 0) */

/* Host-side result, allocated by the engine, released with yacrd_result_free(). */
typedef struct {
    uint64_t n_reads;
    uint64_t n_regions;    /* G */
    uint64_t *bad_offsets; /* R+1 */
    uint32_t *bad_regions; /* 2*G */
    uint8_t *read_type;    /* R */
} yacrd_result;

/* Device-side result: pointers into engine-owned HBM, valid until the next run / destroy. */
typedef struct {
    uint64_t n_reads;
    uint64_t n_regions;
    const void *d_bad_offsets; /* u64[R+1] */
    const void *d_bad_regions; /* u32[2*G] */
    const void *d_read_type;   /* u8[R]   */
} yacrd_device_result;

/* Wall times of the last run, measured with HIP events on the engine's stream.  An event costs
 * ~3 us of stream time, so by default only the dominant kernel is timed (class_ms of the class
 * with the most intervals, or fused_ms — the latter with start / stop events attached to the
 * launch, i.e. the dispatch's own timestamps); plan_ms, sweep_*_ms, compact_ms, total_ms and class_ms
 * of the other classes need YACRD_F_TIMING_FULL and are 0 without it. */
typedef struct {
    float h2d_ms;
    float plan_ms;          /* size-class binning */
    float sweep_small_ms;   /* one read per wavefront (dominant kernel) */
    float sweep_medium_ms;  /* one read per workgroup, LDS resident */
    float sweep_general_ms; /* global-memory path: huge / degenerate reads */
    float compact_ms;       /* scan + compact + classify */
    float d2h_ms;
    float total_ms;         /* first kernel start -> last kernel end */
    uint64_t n_small, n_medium, n_general;     /* reads per class */
    uint64_t iv_small, iv_medium, iv_general;  /* intervals per class */
    /* per size class (YACRD_CLASS_NAMES order): own kernel time, reads, intervals */
    float class_ms[12];
    uint64_t class_reads[12];
    uint64_t class_intervals[12];
    /* the classes R2..H16 run as ONE launch (sweep_small_fused_kernel): its time and what it
     * processed; class_ms of those classes is then 0 */
    float fused_ms;
    uint64_t fused_reads, fused_intervals;
    /* reads whose events were thinned by the coverage pre-filter before the sort (counted only
     * under YACRD_F_COUNT_PREFILTERED) */
    uint64_t prefiltered_reads;
    /* reads (and their intervals) the fused kernel's screen could not finish and left to the follow-on
     * kernel's sort (they are part of fused_reads / fused_intervals: the screen loads and bins them);
     * screened: 1 when the run's fused launch was the screening build (the count of such runs in
     * yacrd_engine_timing_total) */
    uint64_t deferred_reads;
    uint64_t deferred_intervals;
    uint32_t screened;
    /* runs in which the dominant kernel carried its start / stop events: 1 or 0 for one run, the count
     * in yacrd_engine_timing_total (fused_ms / class_ms are sums over these runs: with
     * YACRD_F_TIMING_SAMPLED not every run is one) */
    uint32_t timed_runs;
    /* groups of list entries per wavefront the screen ran with (1 or 2; the last run's): 2 for long launches unless the one
     * before left more than a tenth of its reads to the sort — then the one-item build with the sliding windows runs */
    uint32_t screen_items;
    /* 1: the screen ran in the build with the second looks (sliding windows; always one item); the count of such runs in
     * yacrd_engine_timing_total */
    uint32_t screen_wide;
    /* 1: the batch ran as ONE launch (YACRD_F_ONE_LAUNCH and every read within 256 intervals); the count of such runs in
     * yacrd_engine_timing_total */
    uint32_t one_launch;
    /* 1: the batch was run a second time with the workgroup classes down the three-launch chain, because a workgroup of the
     * persistent screen + fallback kernel ran out of looks at its queue slot (its grid was not resident as a whole: a device
     * shared with another process, a CU mask); the count of such runs in yacrd_engine_timing_total.  (Was reserved0: ABI 5.) */
    uint32_t fused_reruns;
    /* Which path a run took — what makes a short batch's latency depend on the engine's history (ABI 7; all four: 0 / 1 for
     * one run, the count of such runs in yacrd_engine_timing_total):
     * predicted: the sweeps were launched without waiting for the plan's class counts, their grids sized from the engine's
     *   previous run of the same shape;
     * prediction_misses: the run's prediction did not hold at the final sync (a class beyond its grid, a read for the
     *   device-wide or the exact path, a region overflow) and the batch was run again down the synchronous path;
     * build_switches: the register classes' launch took another build than the engine's previous run (sorting / screening,
     *   one item / two items / second looks: see screen_items, screen_wide);
     * sorting_build: that launch was the sorting build (the screen switched off by size or by the last batches' deferral rate). */
    uint32_t predicted;
    uint32_t prediction_misses;
    uint32_t build_switches;
    uint32_t sorting_build;
} yacrd_timing;

/* size classes, in the order of yacrd_timing.class_*: R<K> = four reads per wavefront (16-lane
 * rows, K keys per lane), H16 = two per wavefront, W<K> = one per wavefront, M1/M2 = one per
 * workgroup (LDS), BIG = device-wide path (> 16384 intervals) */
#define YACRD_CLASS_NAMES "R2,R4,R8,R16,H16,W2,W4,W8,W16,M1,M2,BIG"

int yacrd_abi_version(void);
const char *yacrd_last_error(void);

int yacrd_engine_create(const yacrd_engine_cfg *cfg, yacrd_engine **out);
void yacrd_engine_destroy(yacrd_engine *e);

/* Blocking: H2D, kernels, D2H.  Replaces FromOverlap::compute_all_bad_part
 * (src/stack.rs:143-162) + the per-read type_of_read of the report loop
 * (src/main.rs:80-84, src/editor/mod.rs:71). */
int yacrd_engine_run(yacrd_engine *e, const uint64_t *offsets, const uint32_t *intervals,
                     const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                     double not_coverage, yacrd_result *out);
void yacrd_result_free(yacrd_result *r);

/* Same computation on inputs already resident in HBM (device pointers, same layout).
 * n_intervals must equal offsets[n_reads].  Blocking (synchronises the engine's stream). */
int yacrd_engine_run_device(yacrd_engine *e, const void *d_offsets, const void *d_intervals,
                            const void *d_lengths, uint64_t n_reads, uint64_t n_intervals,
                            uint32_t coverage, double not_coverage, yacrd_device_result *out);

/* yacrd_engine_run_device in two halves, so that one host thread can keep batches in flight on
 * several engines of a device (submit on engine A, submit on engine B, wait on A, submit on A ...):
 * the plan / follow-on kernels and the launch gaps of one batch hide behind the
 * sweep of another.  submit enqueues the
 * whole run when the previous run on this engine had the same shape (its class counts size the
 * launches; wait validates them and, in the rare case they do not hold, runs the batch again the
 * synchronous way) and otherwise simply runs it to the end.  The inputs must stay valid until
 * wait returns; between submit and wait the engine accepts no other call. */
int yacrd_engine_submit_device(yacrd_engine *e, const void *d_offsets, const void *d_intervals,
                               const void *d_lengths, uint64_t n_reads, uint64_t n_intervals,
                               uint32_t coverage, double not_coverage);
int yacrd_engine_wait(yacrd_engine *e, yacrd_device_result *out);

/* The submit / wait loop over a LIST of device-resident batches, kept on this side of the ABI (a
 * streaming host that hands over many batches pays one call, not two per batch): batch i runs on
 * engines[i % n_engines], all on one device, up to n_engines batches in flight.  `done`, when given,
 * is called from the calling thread once per batch, in order, after its wait and before its engine
 * is used again — the place to consume the batch's device-resident result (yacrd_engine_fetch) —
 * and a non-zero return stops the loop (YACRD_EINVAL).  The replacement for the batch loop of
 * FromOverlap::compute_all_bad_part (src/stack.rs:143-162) when the overlaps already sit in HBM. */
typedef struct yacrd_device_batch {
    const void *d_offsets, *d_intervals, *d_lengths;
    uint64_t n_reads, n_intervals;
    uint32_t coverage;
    double not_coverage;
} yacrd_device_batch;
typedef int (*yacrd_batch_done_fn)(void *user, uint32_t batch, yacrd_engine *e, const yacrd_device_result *res);
int yacrd_engines_run_device_batches(yacrd_engine *const *engines, uint32_t n_engines,
                                     const yacrd_device_batch *batches, uint32_t n_batches,
                                     yacrd_batch_done_fn done, void *user, yacrd_device_result *last);

/* yacrd_engine_run in two halves for HOST inputs: submit validates the CSR, enqueues H2D + the whole
 * run on the engine's stream and returns; collect waits and brings the result home.  With two
 * engines per GPU a caller keeps PCIe and the kernels busy at the same time (batch k+1 crosses PCIe
 * while batch k is swept) — the batch loop of FromOverlap::compute_all_bad_part
 * (src/stack.rs:143-162) as a pipeline.  Inputs should come from yacrd_pinned_alloc (direct DMA,
 * ~55 GB/s; they must then stay valid until collect returns); pageable inputs are staged through
 * the engine's pinned bounce buffers before submit returns.  Like submit_device, the run is only
 * left in flight when the previous run on this engine had the same shape; otherwise submit runs it
 * to the end and collect just fetches. */
int yacrd_engine_submit(yacrd_engine *e, const uint64_t *offsets, const uint32_t *intervals,
                        const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                        double not_coverage);
int yacrd_engine_collect(yacrd_engine *e, yacrd_result *out);

/* Page-locked host memory (hipHostMalloc) for inputs: the engine recognises it and moves it over
 * PCIe by direct DMA instead of staging.  NULL on failure. */
void *yacrd_pinned_alloc(size_t bytes);
void yacrd_pinned_free(void *p);

/* ---- streaming ingest: overlap records cross PCIe while the parser is still running -------------
 * The north star's "streams it to HBM via pinned hipMemcpyAsync".  The reference hands the engine
 * whole reads (MapReads2Ovl, src/reads2ovl/mod.rs:41) only after the last line is parsed
 * (FullMemory::get_overlaps, src/reads2ovl/fullmemory.rs:46-50); here the parser threads fill
 * pinned buffers with overlap RECORDS (both reads of a PAF/M4 line, src/reads2ovl/mod.rs:108-109)
 * and every full buffer goes to HBM on a copy stream at once, double-buffered.  When the parser is
 * done, yacrd_stream_finish builds the CSR on the GPU (count -> scan -> scatter, the grouping
 * FullMemory::add_overlap_and_length does per line, src/reads2ovl/fullmemory.rs:82-90) and runs the
 * engine on it: parse || H2D, then GPU CSR build + kernels + D2H (all sub-millisecond per 10 M
 * intervals).  Reads are named by 32-bit handles the caller chooses while parsing; `handle_map`
 * translates them to final read ids at finish (NULL = handles are read ids). */
#ifndef YACRD_OVL_REC_DEFINED
#define YACRD_OVL_REC_DEFINED
typedef struct {
    uint32_t a, b;   /* handles of the two reads of an overlap line */
    uint32_t sa, ea; /* interval on read a (PAF cols 3-4, src/io.rs:23-34) */
    uint32_t sb, eb; /* interval on read b (PAF cols 8-9) */
} yacrd_ovl_rec;
/* Where a parser puts its records (implemented by yacrd_stream; consumed by
 * yacrd_ingest_stream in yacrd_host.h).  Both calls may come from several threads at once. */
typedef struct {
    void *ctx;
    /* a buffer of *capacity records to fill; blocks while all buffers are in flight */
    int (*acquire)(void *ctx, yacrd_ovl_rec **buf, uint64_t *capacity);
    /* hand a buffer back with n_records filled (0 allowed): its copy to HBM starts now */
    int (*commit)(void *ctx, yacrd_ovl_rec *buf, uint64_t n_records);
} yacrd_rec_sink;
#endif

typedef struct yacrd_stream yacrd_stream;
/* chunk_records: records per pinned buffer (0 = 131072, 3 MiB); n_buffers: 0 = 2 per usable CPU + 2 */
int yacrd_stream_open(yacrd_engine *e, uint64_t chunk_records, uint32_t n_buffers, yacrd_stream **out);
int yacrd_stream_sink(yacrd_stream *s, yacrd_rec_sink *sink);
int yacrd_stream_acquire(yacrd_stream *s, yacrd_ovl_rec **buf, uint64_t *capacity);
int yacrd_stream_commit(yacrd_stream *s, yacrd_ovl_rec *buf, uint64_t n_records);
/* All records are in.  handle_map[n_handles] (or NULL), lengths[n_reads]: builds the CSR in HBM,
 * runs the engine (blocking) and returns the host result like yacrd_engine_run.  The stream is
 * empty again afterwards — on success AND on every error return — and can take the next file; the two
 * exceptions are calls that are refused before anything is touched (YACRD_EINVAL: a batch pending on the
 * engine, a buffer a parser still holds): fix the cause, then yacrd_stream_reset (the group's finish
 * clears every device when any of its streams failed). */
int yacrd_stream_finish(yacrd_stream *s, const uint32_t *handle_map, uint64_t n_handles,
                        const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                        double not_coverage, yacrd_result *out);
typedef struct {
    uint64_t n_records;  /* overlap records of the last finished stream */
    uint64_t h2d_bytes;  /* bytes the copy stream moved */
    float h2d_busy_ms;   /* sum of the DMA durations (HIP events around each buffer's copy) */
    float build_ms;      /* CSR build on the GPU: count + scan + scatter */
    float run_ms;        /* engine run on the built CSR, wall clock incl. its sync */
    float d2h_ms;
} yacrd_stream_stats;
int yacrd_stream_last_stats(const yacrd_stream *s, yacrd_stream_stats *st);
/* Discard every record committed so far (an ingest that failed half way through its file leaves its
 * records in the stream: call this before the stream takes another file).  No buffer may be held. */
int yacrd_stream_reset(yacrd_stream *s);
void yacrd_stream_close(yacrd_stream *s);

/* ---- streaming ingest over several GPUs --------------------------------------------------------
 * The read-partitioned form of the stream (the reference's equivalent is its batch loop over reads,
 * src/stack.rs:143-162, fed by src/reads2ovl/mod.rs:83-113): a read belongs to device
 * yacrd_stream_device_of(handle, N) = handle mod N, a fact known the moment the parser has interned
 * the id, so the group's sink routes every record while the parse is still running — to the device of
 * its read a and, when that is another one, also to the device of its read b (every record crosses
 * PCIe at most twice, whatever N is) — through one yacrd_stream per device.  At the end each device
 * gets its own handle map (its reads numbered densely in first-appearance order, every other read
 * YACRD_HANDLE_ELSEWHERE: csr_build takes only the half of a record that names a read of its own),
 * builds its CSR, runs; the results are merged back into first-appearance order.  No collective.
 * With one engine the sink is the stream's own (no routing, no extra copy). */
#define YACRD_HANDLE_ELSEWHERE 0xFFFFFFFEu /* in a handle map: this read lives on another device */
typedef struct yacrd_stream_group yacrd_stream_group;
uint32_t yacrd_stream_device_of(uint32_t handle, uint32_t n_devices);
int yacrd_stream_group_open(yacrd_engine *const *engines, uint32_t n_engines, uint64_t chunk_records,
                            uint32_t n_buffers, yacrd_stream_group **out);
int yacrd_stream_group_sink(yacrd_stream_group *g, yacrd_rec_sink *sink);
/* like yacrd_stream_finish; handle_map / lengths are the GLOBAL ones (handle -> first-appearance id) */
int yacrd_stream_group_finish(yacrd_stream_group *g, const uint32_t *handle_map, uint64_t n_handles,
                              const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                              double not_coverage, yacrd_result *out);
/* stats of device `index`'s stream and the number of reads it owned in the last finish */
int yacrd_stream_group_last_stats(const yacrd_stream_group *g, uint32_t index, yacrd_stream_stats *st,
                                  uint64_t *n_reads_owned);
int yacrd_stream_group_reset(yacrd_stream_group *g);
void yacrd_stream_group_close(yacrd_stream_group *g);

/* ---- PAF text -> read types with the parse on the GPU ------------------------------------------------
 * Reads2Ovl::init_paf (src/reads2ovl/mod.rs:83-113) + FullMemory (src/reads2ovl/fullmemory.rs:82-90) +
 * compute_all_bad_part in one call: the host only moves the text (pread chunks -> pinned buffers ->
 * hipMemcpyAsync), the device parses it (nine tab-separated columns as src/io.rs:23-34 names them), interns the
 * ids, numbers the reads by first appearance (a read's length = the first one seen), builds the CSR and runs
 * the engine.  `reads` receives what the report and the editors need to name the reads (host arrays, released
 * with yacrd_reads_free).  Plain PAF files only; returns YACRD_EFALLBACK (nothing else happened) for whatever
 * only the host parser handles — a '"' or a lone CR anywhere (csv quoting / record rules), a 0x integer, a
 * malformed line (the host parser words the error), a read length beyond u32, a file that is not regular:
 * the caller then takes yacrd_ingest_stream + yacrd_stream_finish. */
typedef struct {
    uint64_t n_reads;
    uint64_t n_records;  /* overlap lines */
    uint32_t *lengths;   /* [n_reads] */
    uint64_t *name_off;  /* [n_reads + 1] */
    char *names;         /* name_off[n_reads] bytes, reads in first-appearance order */
} yacrd_reads;
typedef struct {
    uint64_t text_bytes, n_records, n_reads;
    float text_ms;   /* file -> pinned -> HBM (wall clock) */
    float parse_ms;  /* scan + parse + id table on the device */
    float build_ms;  /* numbering, names, CSR */
    float run_ms;    /* the engine */
    float d2h_ms;
} yacrd_ingest_stats;
int yacrd_engine_ingest_paf(yacrd_engine *e, const char *path, int n_threads, uint32_t coverage, double not_coverage,
                            yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats /* may be NULL */);
/* The same for either overlap format — format: 0 = by file name like util::get_file_type (src/util.rs:39-55), 1 = PAF,
 * 2 = M4 / MHAP (Reads2Ovl::init_m4, src/reads2ovl/mod.rs:115-145; M4Record, src/io.rs:36-50: twelve space-separated
 * columns; an error rate not written as plain decimal digits is the host parser's: YACRD_EFALLBACK). */
int yacrd_engine_ingest_overlaps(yacrd_engine *e, const char *path, int format, int n_threads, uint32_t coverage,
                                 double not_coverage, yacrd_result *out, yacrd_reads *reads,
                                 yacrd_ingest_stats *stats /* may be NULL */);
/* The same over text that already lies in host memory (format: 1 = PAF, 2 = M4) — what a compressed overlap file is
 * once libyacrd_host has inflated it (yacrd_text_from_file: the reference reads .gz / .bz2 / .xz through niffler like
 * any other file, src/util.rs:57-70).  yacrd_engine_ingest_overlaps itself answers YACRD_EFALLBACK for a file that
 * begins with a gzip / bzip2 / xz magic.  `text` needs no padding; it may be pageable. */
int yacrd_engine_ingest_overlaps_mem(yacrd_engine *e, const char *text, uint64_t n_bytes, int format, int n_threads,
                                     uint32_t coverage, double not_coverage, yacrd_result *out, yacrd_reads *reads,
                                     yacrd_ingest_stats *stats /* may be NULL */);
/* The same with SEVERAL engines (one per GPU; engines that share a device take shares of it) — the N-GPU form of
 * Reads2Ovl::init_paf / init_m4 (src/reads2ovl/mod.rs:83-145) + compute_all_bad_part (src/stack.rs:143-162: reads are
 * independent).  Engine d moves and parses a byte range of the text — over its own PCIe link —, cut on 4 MiB boundaries; a
 * line belongs to the range it STARTS in (the range's mirror reaches one chunk beyond it).  The ranges' read lists (name,
 * first position in the file, first length, intervals: a few dozen bytes per read and range) go to engines[0], which numbers
 * the reads of the whole file by first appearance; every engine rewrites its records to those numbers, the reads are dealt
 * out as contiguous ranges of numbers with about the same number of intervals each, every engine builds the CSR of its
 * reads from the records of all ranges (copied device to device where they live elsewhere: the one exchange step the
 * path has) and sweeps it.  Results, names and lengths as from one engine, bit for bit.  YACRD_EFALLBACK as there (any
 * range's text may ask for the host parser).  A text of fewer than two chunks is engines[0]'s alone. */
int yacrd_engines_ingest_overlaps(yacrd_engine *const *engines, uint32_t n_engines, const char *path, int format, int n_threads,
                                  uint32_t coverage, double not_coverage, yacrd_result *out, yacrd_reads *reads,
                                  yacrd_ingest_stats *stats /* may be NULL */);
int yacrd_engines_ingest_overlaps_mem(yacrd_engine *const *engines, uint32_t n_engines, const char *text, uint64_t n_bytes,
                                      int format, int n_threads, uint32_t coverage, double not_coverage, yacrd_result *out,
                                      yacrd_reads *reads, yacrd_ingest_stats *stats /* may be NULL */);
void yacrd_reads_free(yacrd_reads *r);
/* The device parser keeps its buffers between calls (the text's mirror, the id table, the records: about twice the
 * file's size; a call into warm buffers is 2-3 times faster than one that has to allocate them): this gives them back. */
int yacrd_engine_trim(yacrd_engine *e);

/* Copy the last device result to host (allocates like yacrd_engine_run). */
int yacrd_engine_fetch(yacrd_engine *e, yacrd_result *out);

int yacrd_engine_last_timing(const yacrd_engine *e, yacrd_timing *t);
/* Sums of the float fields over the runs since the last reset (count fields are those of the
 * last run), and how many runs: lets a caller time K runs without K round trips. */
int yacrd_engine_timing_total(yacrd_engine *e, yacrd_timing *sum, uint64_t *n_runs, int reset);

/* Elapsed time of an EMPTY event bracket on the engine's stream (mean of 32): what the two
 * hipEventRecord calls add to a class_ms / fused_ms value.  Lets a caller report the kernel's own
 * duration (bracket - overhead) next to the raw bracket; rocprofv3 --kernel-trace gives the same
 * figure from the dispatch timestamps. */
int yacrd_engine_event_overhead(yacrd_engine *e, float *ms);

/* Read-id partitioning for multi-GPU: cuts[n_parts+1], contiguous read
 * ranges balanced by interval count; reads are independent so there is no exchange step. */
int yacrd_partition_reads(const uint64_t *offsets, uint64_t n_reads, uint32_t n_parts,
                          uint64_t *cuts);

/* Read-partitioned run over several GPUs of one node: contiguous read ranges from
 * yacrd_partition_reads(), one host thread per engine (one engine per device; the same device
 * may appear twice, useful for testing), no collective — results are concatenated in read order,
 * so the output is identical to a single-engine run. */
int yacrd_engines_run_partitioned(yacrd_engine *const *engines, uint32_t n_engines,
                                  const uint64_t *offsets, const uint32_t *intervals,
                                  const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                                  double not_coverage, yacrd_result *out);

/* Standalone classification of an existing region CSR (the reference calls type_of_read
 * again in every editor, e.g. src/editor/scrubbing.rs:181).  Host buffers in, host out. */
int yacrd_engine_classify(yacrd_engine *e, const uint64_t *bad_offsets,
                          const uint32_t *bad_regions, const uint32_t *lengths,
                          uint64_t n_reads, double not_coverage, uint8_t *read_type);

#ifdef __cplusplus
}
#endif
#endif
