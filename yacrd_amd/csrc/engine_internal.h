// engine_internal.h — host-side internals shared by the translation units of libyacrd_hip.so
// (engine.hip: batch runs; stream.hip: streaming ingest + CSR build on the GPU).
#pragma once
#include "../../include/yacrd_engine_debug.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <mutex>
#include <string>

#include "device_common.h"

namespace yke {

// message of the last error on the calling thread (yacrd_last_error)
std::string &err_slot();
inline int fail(int code, const std::string &msg)
{
    err_slot() = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return yke::fail(_e == hipErrorOutOfMemory ? YACRD_ENOMEM : YACRD_ENODEV,          \
                             std::string(#expr) + ": " + hipGetErrorString(_e));              \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
};

enum { EV_START = 0, EV_PLAN, EV_S0, EV_SMALL, EV_MED, EV_GEN, EV_COMPACT, EV_X0, EV_X1, EV_COUNT };


struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

inline float ev_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}

} // namespace yke


// One per device: whose dominant sweep went out last (see launch_sweeps).
struct BigLane {
    std::mutex mu;
    hipEvent_t last = nullptr;
    struct yacrd_engine *owner = nullptr;
    int n_engines = 0;
};
static BigLane g_big_lane[64];
// (Rounds 4-5 had a FusedLane here: the persistent screen + fallback launches of the engines that share a device took turns.
// Since round 6 no workgroup of that launch waits for another — screen_wg.h — and engines launch it side by side.)

// A batch that was submitted without waiting for it (yacrd_engine_submit_device).
struct Pending {
    bool active = false;
    const u64 *d_off = nullptr;
    const uint2 *d_iv = nullptr;
    const u32 *d_len = nullptr;
    uint64_t n_reads = 0, n_iv = 0;
    uint32_t cov = 0;
    double not_cov = 0;
    u32 grid_n[12] = {};      // reads each class's grid covers
    u32 big_n = 0;            // reads / intervals beyond the workgroup classes the device-wide screen was launched for (a prediction)
    u64 big_iv = 0;
    bool fused_marked = false, screened = false;
    bool one_launch = false;  // the batch went out as one_batch_kernel (one_batch.h)
    int cls_b[12] = {}, cls_e[12] = {};
};

struct yacrd_engine {
    Pending pending;
    bool in_lane = false;
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;                       // the device-wide screen's launches, beside the workgroup classes' (run_on_device)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // stream -> side (behind the plan), side -> stream (in front of the follow-on step)
    hipEvent_t ev[yke::EV_COUNT] = {};
    hipEvent_t ev_h2d0 = nullptr, ev_h2d1 = nullptr, ev_d2h0 = nullptr, ev_d2h1 = nullptr;
    hipEvent_t ev_cls[24] = {}; // brackets around class kernels
    hipEvent_t ev_done = nullptr; // hipEventBlockingSync: the final wait of YACRD_F_BLOCKING_WAIT
    int num_cu = 256;
    bool fused_off = false; // this run: the workgroup classes down the three-launch chain (a fused launch gave up: Counters::fused_gave_up)
    int num_xcc = 0; // XCDs of this device / partition (one_batch_kernel's read-to-XCD map assumes 8)
    int screen_fused_wgs_per_cu = 0; // workgroups of screen_wg_fused_kernel a CU holds at once (its grid must be resident as a whole)

    // inputs staged by yacrd_engine_run
    yke::DevBuf in_off, in_iv, in_len;
    // work buffers
    yke::DevBuf dlist; // the follow-on step's list of marked reads (long batches: finish_compact.h, mark_list_kernel)
    uint32_t compact_calls = 0; // launch_compact calls of the current run (a redo starts the list over)
    yke::DevBuf mrec; // the workgroup classes' records (plan_compact.h)
    yke::DevBuf lists, ctrl2[2], stage, counts, closed, gen_sizes, gen_scratch_off, gen_scratch, big_tab, big_keys, big_redo, bt_tab, bt_hist, bt_cur, bt_keys, bs_seg, bs_chunk, bs_hist;
    // two control blocks (counters + scan state), used alternately: the plan kernel of a run zeroes
    // the other one for the next run.  ctrl_clean[i] = leading bytes of block i known to be zero.
    size_t ctrl_clean[2] = {0, 0};
    int ctrl_cur = 0;
    // results
    yke::DevBuf bad_offsets, bad_regions, read_type;
    yk::Counters *h_ctr = nullptr; // pinned

    uint64_t last_reads = 0, last_regions = 0;
    size_t last_list_stride = 0; // reads per class list of the current run (lists = [class][read])
    bool has_result = false;
    // class counts of the previous run: the prediction that lets the next one skip the plan sync
    yk::Counters pred{};
    uint64_t pred_reads = 0, pred_iv = 0;
    bool pred_valid = false;
    yacrd_timing timing = {};
    yacrd_timing timing_sum = {};
    uint64_t timing_runs = 0;
    uint32_t run_seq = 0; // runs since creation (YACRD_F_TIMING_SAMPLED times every 8th)
    uint32_t wide_left = 0;    // batches the screen still takes in its one-item build WITH the second looks (sliding windows) before the default builds are tried again
    uint32_t last_items = 1;   // groups of list entries per wavefront the last screen ran with
    bool last_wide = false;    // ... and whether it was the build with the second looks
    bool one_launch_off = false; // (while a batch the one-launch form could not take is run again on the default path)
    int last_build = -1, prev_build = -1; // build of the register classes' launch in the last run / the one before (-1: none; 0 sorting, 1 / 2 screening with one / two items, 3 second looks)
    bool miss_pending = false;            // the run in progress is the synchronous re-run of a batch whose prediction did not hold
    uint32_t nodefer_left = 0; // batches the sorting build of the fused launch still takes before the screen is tried again
    // pinned bounce buffers for pageable inputs (yke::h2d), allocated on first use; an event per
    // buffer says when its DMA is done and it may be refilled
    static constexpr int kBounce = 12;
    static constexpr size_t kBounceBytes = (size_t)4 << 20;
    void *bounce[kBounce] = {};
    hipEvent_t bounce_ev[kBounce] = {};
    bool bounce_busy[kBounce] = {};
    // pinned staging for the results on their way home (fetch_result), grow-only
    void *h_out = nullptr;
    size_t h_out_cap = 0;
    // a batch submitted from host buffers (yacrd_engine_submit): collect() fetches the result
    bool host_pending = false;
    // pinned buffers the PAF text passes through on its way to HBM (gpu_paf.hip), grow-only
    void *paf_arena = nullptr;
    size_t paf_arena_cap = 0;
    void *paf_scratch = nullptr;                 // gpu_paf.hip's device buffers (its type), kept between calls
    void (*paf_scratch_free)(void *) = nullptr;
};


namespace yke {
// the whole launch sequence over a CSR resident in HBM (engine.hip)
int run_on_device(yacrd_engine *e, const u64 *d_off, const uint2 *d_iv, const u32 *d_len,
                  uint64_t n_reads64, uint64_t n_iv, uint32_t cov, double not_cov, bool defer = false);
// D2H of the last result into a freshly allocated yacrd_result
int fetch_result(yacrd_engine *e, yacrd_result *out);
// overlap records in HBM -> the engine's input CSR (stream.hip; blocking), and a u32 -> u64 exclusive scan
struct RecSlab {
    const yk::OvlRec *recs;
    uint64_t n;
};
int csr_from_records(yacrd_engine *e, const RecSlab *slabs, size_t n_slabs, const u32 *d_map, u64 n_handles, u64 n_reads,
                     DevBuf &cnt, DevBuf &part, DevBuf &err, hipEvent_t done, u64 *n_intervals = nullptr,
                     bool counted = false, u64 iv_bound = 0);
int scan_u32_to_u64(yacrd_engine *e, const u32 *in, u64 n, u64 *out, DevBuf &part);
// host -> HBM at PCIe rate: direct DMA when `src` is pinned, otherwise through the engine's pinned
// bounce buffers filled by a few copy threads; asynchronous on e->stream only for pinned sources
int h2d(yacrd_engine *e, void *dst, const void *src, size_t bytes);
} // namespace yke
