/*
 * yacrd_host.h — C ABI of the host side (libyacrd_host.so, plain C++17, no GPU code):
 * overlap ingest into the CSR the engine consumes, the .yacrd report writer, and the
 * deterministic synthetic workload generator used by bench.py and the scale tests.
 *
 * Reference interfaces replaced (paths relative to the reference root):
 *   yacrd_csr_from_file   <- Reads2Ovl::init / init_paf / init_m4  (src/reads2ovl/mod.rs:43-145)
 *                            + FullMemory::add_overlap_and_length  (src/reads2ovl/fullmemory.rs:82-90)
 *   yacrd_report_write    <- editor::report loop of main           (src/main.rs:80-84,
 *                            src/editor/mod.rs:61-107)
 */
#ifndef YACRD_HOST_H
#define YACRD_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *yacrd_host_last_error(void);

/* ---- CSR built from an overlap file --------------------------------------------------------- */
typedef struct yacrd_csr yacrd_csr; /* opaque, owns its arrays */

typedef struct {
    uint64_t n_reads;
    uint64_t n_intervals;
    uint64_t n_records;        /* overlap lines ingested */
    const uint64_t *offsets;   /* R+1; NULL after yacrd_ingest_stream (the CSR lives in HBM) */
    const uint32_t *intervals; /* 2*I; NULL after yacrd_ingest_stream */
    const uint32_t *lengths;   /* R (first length seen) */
    const uint64_t *name_off;  /* R+1 offsets into names */
    const char *names;         /* concatenated read ids, first-appearance order */
} yacrd_csr_view;

/* format: 0 = by file name like util::get_file_type (src/util.rs:39-55), 1 = PAF, 2 = M4.
 * n_threads: parser threads (0 = hardware concurrency). */
int yacrd_csr_from_file(const char *path, int format, int n_threads, yacrd_csr **out);
int yacrd_csr_from_memory(const char *text, size_t len, int format, int n_threads,
                          yacrd_csr **out);
int yacrd_csr_get(const yacrd_csr *c, yacrd_csr_view *v);

/* ---- streaming ingest ----------------------------------------------------------------------------
 * Same parse, same numbering, but the overlap records leave for a sink buffer by buffer while the
 * parse is still running instead of being grouped on the host: with the engine's yacrd_stream as
 * the sink (include/yacrd_engine.h) they cross PCIe from pinned memory during the parse and the CSR
 * is built in HBM.  Records name reads by HANDLES (32-bit numbers unique per id, dense enough to
 * index an array) because the first-appearance numbering only exists once the last line is read
 * (the reference has the same barrier: FullMemory::get_overlaps hands the map over at the end,
 * src/reads2ovl/fullmemory.rs:46-50); yacrd_csr_handle_map gives handle -> read id.
 * The returned yacrd_csr holds names and lengths; its view has offsets == intervals == NULL. */
#ifndef YACRD_OVL_REC_DEFINED
#define YACRD_OVL_REC_DEFINED
typedef struct {
    uint32_t a, b;   /* handles of the two reads of an overlap line */
    uint32_t sa, ea; /* interval on read a (PAF cols 3-4, src/io.rs:23-34) */
    uint32_t sb, eb; /* interval on read b (PAF cols 8-9) */
} yacrd_ovl_rec;
typedef struct {
    void *ctx;
    /* a buffer of *capacity records to fill; may block; called from several threads */
    int (*acquire)(void *ctx, yacrd_ovl_rec **buf, uint64_t *capacity);
    /* hand a buffer back with n_records filled (0 allowed) */
    int (*commit)(void *ctx, yacrd_ovl_rec *buf, uint64_t n_records);
} yacrd_rec_sink;
#endif
int yacrd_ingest_stream(const char *path, int format, int n_threads, const yacrd_rec_sink *sink,
                        yacrd_csr **out);
int yacrd_ingest_stream_memory(const char *text, size_t len, int format, int n_threads,
                               const yacrd_rec_sink *sink, yacrd_csr **out);
/* handle -> read id (0xFFFFFFFF for handles no record uses); valid while the csr lives */
int yacrd_csr_handle_map(const yacrd_csr *c, const uint32_t **map, uint64_t *n_handles);
/* index of a read id, or -1 (BadPart::get_bad_part answers unknown ids with an empty list,
 * src/stack.rs:164-169) */
int64_t yacrd_csr_find(const yacrd_csr *c, const char *name, size_t name_len);
void yacrd_csr_free(yacrd_csr *c);

/* ---- report ---------------------------------------------------------------------------------- */
/* One line per read: "{type}\t{id}\t{len}\t{len_i,begin_i,end_i;...}\n" (src/editor/mod.rs:61-107),
 * reads in CSR (first-appearance) order. */
int yacrd_report_write(const char *path, const yacrd_csr_view *reads, const uint64_t *bad_offsets,
                       const uint32_t *bad_regions, const uint8_t *read_type);

/* ---- editors and the .yacrd re-reader ----------------------------------------------------------- */
/* What BadPart::get_bad_part answers (src/stack.rs:164-169) plus the engine's read type, over
 * all reads: names (concatenated, name_off[R+1]), lengths, region CSR, read_type. */
typedef struct {
    uint64_t n_reads;
    const uint64_t *name_off;
    const char *names;
    const uint32_t *lengths;
    const uint64_t *bad_offsets;
    const uint32_t *bad_regions;
    const uint8_t *read_type; /* from the engine: yacrd_engine_run / yacrd_engine_classify */
} yacrd_badparts_view;

enum { YACRD_OP_SCRUBB = 0, YACRD_OP_FILTER = 1, YACRD_OP_EXTRACT = 2, YACRD_OP_SPLIT = 3 };

/* editor::{scrubbing,filter,extract,split} (src/editor/ scrubbing.rs, filter.rs, extract.rs, split.rs): FASTA/FASTQ for all four, PAF/M4
 * for filter and extract; gzip in -> gzip out.  Reads unknown to `bp` are NotBad with no region. */
int yacrd_edit_file(int op, const char *in_path, const char *out_path, const yacrd_badparts_view *bp);
/* The same with a thread count (0 = the usable CPUs, at most three: what yacrd_edit_file passes; YACRD_EDIT_THREADS overrides):
 * plain FASTA / FASTQ files are cut at record boundaries and edited chunk-parallel — the threads parse side by side, one of
 * them at a time writes, chunk after chunk in order —, output byte-identical to one thread's; compressed files and overlap
 * files take the one-thread loop. */
int yacrd_edit_file_mt(int op, const char *in_path, const char *out_path, const yacrd_badparts_view *bp, int n_threads);

/* FromReport (src/stack.rs:176-257): a .yacrd report back into the BadPart table.  read_type is
 * left NULL: classify with yacrd_engine_classify() and the -n of the current invocation. */
typedef struct yacrd_report yacrd_report;
int yacrd_report_read(const char *path, yacrd_report **out);
int yacrd_report_get(const yacrd_report *r, yacrd_badparts_view *v);
void yacrd_report_free(yacrd_report *r);

/* ---- a compressed overlap file as text in memory (for yacrd_engine_ingest_overlaps_mem: the parse on the GPU) ----
 * gzip / bzip2 / xz, sniffed from the magic bytes like the reference's niffler reader (src/util.rs:57-70).  Returns
 * 0 = `out` holds the text (free with yacrd_text_free), 2 = the file is not compressed (or not a regular file): read
 * it where it lies, 1 = error (truncated / corrupt stream, like the reference's readers).  BGZF files (bgzip) are
 * inflated member-parallel on n_threads threads (0 = every usable CPU); any other stream is one thread's work. */
typedef struct {
    char *data;        /* n bytes of text (+ at least 64 readable bytes behind them) */
    uint64_t n, cap;   /* cap: size of the mapping behind `data` */
    uint64_t members;  /* BGZF members inflated in parallel (1 = one stream) */
    uint32_t threads;  /* threads that inflated */
    int32_t compression; /* 1 gzip, 2 bzip2, 3 xz */
} yacrd_text;
int yacrd_text_from_file(const char *path, int n_threads, yacrd_text *out);
void yacrd_text_free(yacrd_text *t);

/* ---- synthetic workloads ------------------------------------------------------ */
enum { YACRD_SYNTH_ONT = 0, YACRD_SYNTH_SEQUEL = 1, YACRD_SYNTH_SKEWED = 2 };
/* flags.  NO_INJECTION: no abutting / degenerate intervals.  JITTER: the N(0, sigma) offset of a
 * dovetail end is REFLECTED into the read instead of clamped onto 0 / len (SURVEY.md §8d's clamp puts
 * 15 % of all starts on exactly 0 and 15 % of all ends on exactly len; a real overlapper's chain ends
 * are spread over a few dozen positions).  Bits 8..15: sigma of that offset in positions (0 = 30).
 * Bits 16..23: per cent of the reads that are chimeras — a junction no overlap crosses — instead of
 * SURVEY.md §8d's 2 (0 = 2): the reads yacrd looks for, i.e. the ones the healthy-read screen defers. */
enum { YACRD_SYNTH_F_NO_INJECTION = 1u, YACRD_SYNTH_F_JITTER = 2u, YACRD_SYNTH_F_SIGMA_X4 = 4u /* the sigma field counts 4 positions */ };
#define YACRD_SYNTH_F_SIGMA(s) ((s) > 255 ? (YACRD_SYNTH_F_SIGMA_X4 | ((((uint32_t)(s) / 4u) & 0xFFu) << 8)) : (((uint32_t)(s) & 0xFFu) << 8))
#define YACRD_SYNTH_F_CHIMERA_PCT(p) (((uint32_t)(p) & 0xFFu) << 16)

typedef struct {
    uint32_t profile;       /* YACRD_SYNTH_* */
    uint32_t flags;         /* YACRD_SYNTH_F_* */
    uint64_t n_reads;
    uint64_t n_overlaps;    /* PAF lines; intervals = 2 * n_overlaps */
    uint64_t seed;
} yacrd_synth_cfg;

/* Fills caller-allocated arrays: offsets[R+1], intervals[4*n_overlaps], lengths[R]. */
int yacrd_synth_csr(const yacrd_synth_cfg *cfg, uint64_t *offsets, uint32_t *intervals,
                    uint32_t *lengths);
/* Same overlaps as PAF text (12 columns + tp:A:S), read ids r%09u. */
int yacrd_synth_paf(const yacrd_synth_cfg *cfg, const char *path);
/* Matching reads as FASTQ (bases from the PRNG, quality '?', a description on every record),
 * plus `extra_reads` reads x%09u that appear in no overlap (they must pass through editors
 * untouched, src/stack.rs:164-169). */
int yacrd_synth_fastq(const yacrd_synth_cfg *cfg, uint64_t extra_reads, const char *path);

#ifdef __cplusplus
}
#endif
#endif
