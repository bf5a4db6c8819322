/*
 * yacrd_oracle.c — CPU restatement of natir/yacrd's bad-region path (see yacrd_oracle.h).
 * TEST INFRASTRUCTURE ONLY: the checker and the reported CPU baseline, never the product path.
 *
 * The sweep keeps the reference's own shape on purpose (sort + binary min-heap + post-merge):
 * the GPU path uses a different, parallel formulation and is compared against this one.
 */
#include "yacrd_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ---- lexicographic order on (start,end): Rust tuple Ord used by sort_unstable, stack.rs:66 */
static int cmp_pair(const void *a, const void *b)
{
    const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}

/* ---- binary min-heap of u32 == BinaryHeap<Reverse<u32>>, stack.rs:63-64 ---- */
static void heap_push(uint32_t *h, size_t *sz, uint32_t v)
{
    size_t i = (*sz)++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (h[p] <= v) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = v;
}

static void heap_pop(uint32_t *h, size_t *sz)
{
    size_t n = --(*sz);
    if (n == 0) return;
    uint32_t v = h[n];
    size_t i = 0;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && h[c + 1] < h[c]) c++;
        if (h[c] >= v) break;
        h[i] = h[c];
        i = c;
    }
    h[i] = v;
}

size_t yo_compute_bad_part(uint32_t *iv, size_t n, uint64_t len, uint64_t coverage,
                           uint32_t *out, uint32_t *heap)
{
    size_t ng = 0; /* raw gaps, written to `out` (at most n + 2) */
    size_t hs = 0; /* heap size */

    qsort(iv, n, 2 * sizeof(uint32_t), cmp_pair); /* stack.rs:66 */

    uint32_t first_covered = 0; /* stack.rs:68 */
    uint32_t last_covered = 0;  /* stack.rs:69 */

    /* The prepend of (0, first_covered) (stack.rs:107-109, Vec::insert(0, ..)) is done by
     * leaving slot 0 free and shifting afterwards if it is not needed. */
    uint32_t *raw = out + 2;

    for (size_t i = 0; i < n; i++) { /* stack.rs:71 */
        uint32_t s = iv[2 * i], e = iv[2 * i + 1];
        while (hs > 0) {            /* stack.rs:72 */
            uint32_t head = heap[0];
            if (head > s) break;    /* stack.rs:73-75 */
            if (hs > coverage) last_covered = head; /* stack.rs:77-79 */
            heap_pop(heap, &hs);    /* stack.rs:80 */
        }
        if (hs <= coverage) {       /* stack.rs:83 */
            if (last_covered != 0) { /* stack.rs:84 */
                raw[2 * ng] = last_covered;
                raw[2 * ng + 1] = s;
                ng++;
            } else {
                first_covered = s;  /* stack.rs:87 */
            }
        }
        heap_push(heap, &hs, e);    /* stack.rs:90 */
    }

    while (hs > coverage) {         /* stack.rs:93 */
        last_covered = heap[0];     /* stack.rs:94-100 */
        if ((uint64_t)last_covered >= len) break; /* stack.rs:101-103 */
        heap_pop(heap, &hs);        /* stack.rs:104 */
    }

    size_t total;
    uint32_t *g;
    if (first_covered != 0) {       /* stack.rs:107-109 */
        out[0] = 0;
        out[1] = first_covered;
        g = out;
        total = ng + 1;
    } else {
        g = raw;
        total = ng;
    }
    if ((uint64_t)last_covered != len) { /* stack.rs:111-113 */
        g[2 * total] = last_covered;
        g[2 * total + 1] = (uint32_t)len; /* `len as u32` */
        total++;
    }

    if (total == 0) return 0;       /* stack.rs:115-117 */

    /* stack.rs:119-136: merge runs of equal begin.  Output index never passes input index,
     * so compacting into `out` in place is safe. */
    size_t w = 0;
    uint32_t begin = g[0], end = g[1];
    for (size_t k = 0; k + 1 < total; k++) {
        uint32_t g1b = g[2 * k], g1e = g[2 * k + 1];
        uint32_t g2b = g[2 * k + 2], g2e = g[2 * k + 3];
        if (g1b == g2b) {
            begin = g1b;
            end = g1e > g2e ? g1e : g2e;
        } else {
            out[2 * w] = begin;
            out[2 * w + 1] = end;
            w++;
            begin = g2b;
            end = g2e;
        }
    }
    out[2 * w] = begin;
    out[2 * w + 1] = end;
    w++;
    return w;
}

int yo_type_of_read(uint64_t len, const uint32_t *regions, size_t n_regions, double not_covered)
{
    /* editor/mod.rs:86: fold in u32, wrapping (release profile: overflow-checks = false,
     * Cargo.toml:40) */
    uint32_t bad = 0;
    for (size_t i = 0; i < n_regions; i++) bad += regions[2 * i + 1] - regions[2 * i];

    /* editor/mod.rs:88: f64 divide, strict >, NaN (0/0) compares false */
    if ((double)bad / (double)len > not_covered) return YO_NOT_COVERED;

    /* editor/mod.rs:92-97 */
    for (size_t i = 0; i < n_regions; i++)
        if (regions[2 * i] != 0 && regions[2 * i + 1] != (uint32_t)len) return YO_CHIMERIC;

    return YO_NOT_BAD;
}

/* ------------------------------------------------------------------------------------- */

typedef struct {
    const uint64_t *offsets;
    const uint32_t *intervals;
    const uint64_t *lengths;
    uint64_t n_reads;
    uint64_t coverage;
    double not_covered;
    uint32_t *stage;     /* per-read slot of 2*(n+2) u32 at 2*(offsets[r] + 2r) */
    uint32_t *counts;    /* regions per read */
    uint8_t *read_type;
    atomic_ullong next;
    int failed;
} job_t;

#define YO_CHUNK 256

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    size_t cap = 0;
    uint32_t *iv = NULL, *heap = NULL;
    for (;;) {
        uint64_t r0 = atomic_fetch_add(&j->next, YO_CHUNK);
        if (r0 >= j->n_reads) break;
        uint64_t r1 = r0 + YO_CHUNK < j->n_reads ? r0 + YO_CHUNK : j->n_reads;
        for (uint64_t r = r0; r < r1; r++) {
            uint64_t o = j->offsets[r];
            size_t n = (size_t)(j->offsets[r + 1] - o);
            if (n > cap) {
                cap = n * 2 + 64;
                free(iv);
                free(heap);
                iv = (uint32_t *)malloc(cap * 2 * sizeof(uint32_t));
                heap = (uint32_t *)malloc(cap * sizeof(uint32_t));
                if (!iv || !heap) {
                    j->failed = 1;
                    free(iv);
                    free(heap);
                    return NULL;
                }
            }
            /* the reference moves the Vec out of the map (stack.rs:152-154); we copy so the
             * caller's CSR stays untouched */
            memcpy(iv, j->intervals + 2 * o, n * 2 * sizeof(uint32_t));
            uint32_t *slot = j->stage + 2 * (o + 2 * r);
            size_t g = yo_compute_bad_part(iv, n, j->lengths[r], j->coverage, slot, heap);
            j->counts[r] = (uint32_t)g;
            j->read_type[r] = (uint8_t)yo_type_of_read(j->lengths[r], slot, g, j->not_covered);
        }
    }
    free(iv);
    free(heap);
    return NULL;
}

int yo_run(const uint64_t *offsets, const uint32_t *intervals, const uint64_t *lengths,
           uint64_t n_reads, uint64_t coverage, double not_covered, int n_threads,
           uint64_t *bad_offsets, uint32_t **bad_regions, uint8_t *read_type)
{
    uint64_t total_iv = offsets[n_reads];
    job_t j;
    memset(&j, 0, sizeof j);
    j.offsets = offsets;
    j.intervals = intervals;
    j.lengths = lengths;
    j.n_reads = n_reads;
    j.coverage = coverage;
    j.not_covered = not_covered;
    j.read_type = read_type;
    j.stage = (uint32_t *)malloc((size_t)(2 * (total_iv + 2 * n_reads) + 2) * sizeof(uint32_t));
    j.counts = (uint32_t *)malloc((size_t)(n_reads + 1) * sizeof(uint32_t));
    atomic_init(&j.next, 0);
    *bad_regions = NULL;
    if (!j.stage || !j.counts) {
        free(j.stage);
        free(j.counts);
        return -1;
    }

    if (n_threads <= 1) {
        worker(&j);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &j);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    if (j.failed) {
        free(j.stage);
        free(j.counts);
        return -1;
    }

    uint64_t g = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        bad_offsets[r] = g;
        g += j.counts[r];
    }
    bad_offsets[n_reads] = g;
    uint32_t *reg = (uint32_t *)malloc((size_t)(2 * g + 2) * sizeof(uint32_t));
    if (!reg) {
        free(j.stage);
        free(j.counts);
        return -1;
    }
    for (uint64_t r = 0; r < n_reads; r++)
        memcpy(reg + 2 * bad_offsets[r], j.stage + 2 * (offsets[r] + 2 * r),
               (size_t)j.counts[r] * 2 * sizeof(uint32_t));
    *bad_regions = reg;
    free(j.stage);
    free(j.counts);
    return 0;
}
