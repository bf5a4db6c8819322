/*
 * yacrd_engine_debug.h — A/B and test switches for yacrd_engine_cfg.flags.  Not part of the drop-in
 * boundary (include/yacrd_engine.h): they pin one kernel family or one build of a launch so that the
 * tests can run every path against the oracle and tools/ can measure one against the other.  The product
 * path is flags = 0 (plus the timing flags of yacrd_engine.h).
 */
#ifndef YACRD_ENGINE_DEBUG_H
#define YACRD_ENGINE_DEBUG_H

#include "yacrd_engine.h"

/* route every read through the fully general (arbitrary-input) kernel; testing only */
#define YACRD_F_FORCE_GENERAL 1u
/* use the LDS-sort kernel for the small class instead of the register-sort kernel; A/B only */
#define YACRD_F_FORCE_LDS_SORT 2u
/* cross-lane exchanges of the register sort all through the LDS crossbar (ds_swizzle); A/B only */
#define YACRD_F_XLANE_DS 4u
/* no four-reads-per-wavefront row layout: every small read gets a whole wavefront; A/B only */
#define YACRD_F_WAVE_ONLY 8u
/* no two-reads-per-wavefront layout for reads of 129..256 intervals; A/B only */
#define YACRD_F_NO_HALVES 16u
/* always wait for the plan's class counts (no prediction from the previous run); A/B only */
#define YACRD_F_NO_PREDICTION 64u
/* one launch per register-sort class instead of the fused launch; A/B only */
#define YACRD_F_NO_FUSED_LAUNCH 128u
/* sort every event: skip the coverage pre-filter (register-sort and LDS classes; A/B, tests) */
#define YACRD_F_NO_PREFILTER 256u
/* count the reads the pre-filter thinned (yacrd_timing.prefiltered_reads); one global atomic per
 * read, so only for tests */
#define YACRD_F_COUNT_PREFILTERED 512u
/* the fused launch of the register-sort classes never runs the healthy-read screen (by default it does
 * from 4 M intervals in the classes R16 + H16 on, unless the previous batches failed the screen too
 * often); 8192: it always does; A/B, tests */
#define YACRD_F_NO_DEFER 4096u
#define YACRD_F_ALWAYS_DEFER 8192u
/* engines that share a device take turns with the dominant sweep launch (a GPU-side event wait:
 * the launch's start / stop events then time that kernel alone); A/B only */
#define YACRD_F_SWEEP_TURNS 16384u
/* the screen takes one / two groups of list entries per wavefront whatever the
 * launch's size (default: two from 40 M intervals on, i.e. inputs outside the Infinity Cache); tests, A/B */
#define YACRD_F_SCREEN_ITEMS_1 262144u
#define YACRD_F_SCREEN_ITEMS_2 524288u
/* the workgroup classes (513 .. 16 384 intervals) run the screen and its fallback as separate launches (round 3's
 * chain) instead of the persistent screen_wg_fused_kernel; A/B, tests */
#define YACRD_F_NO_FUSED_SCREEN 1048576u
/* the workgroup classes go through a screen of one read per WAVEFRONT first, the read streamed twice (screen_stream.h,
 * round 6: measured, 2 x the traffic and no faster — DESIGN.md), and the persistent screen + fallback kernel takes only what
 * that leaves; A/B, tests */
#define YACRD_F_STREAM_SCREEN 65536u
/* the screen always runs in its one-item build with the second looks (sliding windows; by default only after a batch that
 * deferred more than a tenth of what it screened); tests, A/B */
#define YACRD_F_SCREEN_WIDE 2097152u

/* the device parser's sort on its own (yacrd_amd/csrc/radix_sort.h): (key, value) pairs in host memory, sorted in place by key,
 * stable; keys must be below key_bound (it decides the number of passes); tests only */
#ifdef __cplusplus
extern "C" {
#endif
int yacrd_debug_sort_pairs(yacrd_engine *e, uint64_t *keys, uint32_t *vals, uint64_t n, uint64_t key_bound);
/* the device-side counter block of the engine's last run as it came home (csrc/device_common.h: struct Counters — class counts,
 * rejection / fallback list lengths, deferred reads): at most `bytes` of it are copied to dst; returns the block's size.
 * tools/ and tests only: the layout is not part of any ABI. */
uint64_t yacrd_debug_last_counters(const yacrd_engine *e, void *dst, uint64_t bytes);
/* cross-engine copies of the N-engine device parser (yacrd_engines_ingest_overlaps) since the library was loaded, by route:
 * [0] same device (hipMemcpyAsync), [1] hipMemcpyPeerAsync, [2] staged through pinned host buffers.
 * YACRD_TEST_FORCE_PEER_COPY=peer|staged (environment, tests) forces route 1 / 2 for EVERY such copy, also between engines of
 * one device, and makes every engine gather the other engines' records instead of reading them in place: the multi-device
 * branch on a one-GPU box. */
void yacrd_debug_peer_copy_counts(uint64_t out[3]);
#ifdef __cplusplus
}
#endif

#endif
