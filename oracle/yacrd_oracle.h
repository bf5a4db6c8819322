/*
 * yacrd_oracle.h — CPU restatement of natir/yacrd's bad-region path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under yacrd_amd/ (the product) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / reported CPU baseline.
 *
 * Parity pin: this restatement is checked (tests/test_oracle.py) against every golden vector
 * the reference holds for the path: tests/reads.paf -> tests/truth.yacrd (230 lines,
 * reference tests/run.rs:95-117), the seven known answers of src/stack.rs:311-390, the six
 * classifications of src/editor/mod.rs:113-128 and the ingest vectors of
 * src/reads2ovl/mod.rs:173-237.  The reference itself (Rust) cannot be built in this image
 * (no cargo/rustc, un-vendored crates), so there is no oracle/_ref.
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 */
#ifndef YACRD_ORACLE_H
#define YACRD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Read classes, same numeric encoding as include/yacrd_engine.h. */
enum { YO_NOT_BAD = 0, YO_CHIMERIC = 1, YO_NOT_COVERED = 2 };

/* src/stack.rs:61-139  FromOverlap::compute_bad_part.
 * `iv` holds n (start,end) pairs and is sorted in place (the reference sorts its own Vec).
 * `out` must have room for n+2 pairs; returns the number of pairs written.
 * `heap` is scratch for n u32 (the reference's BinaryHeap<Reverse<u32>>). */
size_t yo_compute_bad_part(uint32_t *iv, size_t n, uint64_t len, uint64_t coverage,
                           uint32_t *out, uint32_t *heap);

/* src/editor/mod.rs:85-100  type_of_read. */
int yo_type_of_read(uint64_t len, const uint32_t *regions, size_t n_regions, double not_covered);

/* src/stack.rs:143-162 driver over a CSR batch (FullMemory hands one batch, fullmemory.rs:46-50)
 * followed by the per-read classification of src/editor/mod.rs:61-83.
 * offsets[R+1] (in intervals), intervals[2*I], lengths[R].
 * bad_offsets[R+1] written; *bad_regions is malloc'ed (2*G u32), caller frees with free();
 * read_type[R] written.  n_threads >= 1 (pthread pool over reads, dynamic chunks).
 * Returns 0, or -1 on allocation failure. */
int yo_run(const uint64_t *offsets, const uint32_t *intervals, const uint64_t *lengths,
           uint64_t n_reads, uint64_t coverage, double not_covered, int n_threads,
           uint64_t *bad_offsets, uint32_t **bad_regions, uint8_t *read_type);

#ifdef __cplusplus
}
#endif
#endif
