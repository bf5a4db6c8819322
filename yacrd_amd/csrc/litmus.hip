// litmus.hip — message passing across XCDs with the EXACT store / wait / atomic / load sequences the device-side
// hand-overs of one_batch.h and finish_compact.h rely on.  Test infrastructure:
// libyacrd_litmus.so is loaded by tests/test_gpu_litmus.py only; nothing in the product links it.
//
// Those hand-overs are NOT release / acquire pairs (an agent-scope fence writes back / invalidates a whole L2 on gfx950:
// 111 us for a 41 us kernel, profiles/r04): they are relaxed agent-scope accesses — global_store / global_load with
// sc1, performed at the XCD's L2 write-through / read-through to memory — ordered by s_waitcnt vmcnt(0) on the
// writer's side and by the data dependence on a RETURNING atomic's result (or on a polled word) on the reader's side:
//   kind 0 (one_batch.h: verdicts -> arrival):  agent store payload; s_waitcnt vmcnt(0); returning agent fetch_add on
//           the arrival word.  The last arriver then reads every payload with agent loads: none may be stale.
//   kind 1 (finish_compact.h: counters -> scan word):  returning agent fetch_add on a counter, result consumed;
//           agent store of the flag word.  A reader polls the flags with agent loads, then reads the counter with an
//           agent load: it must hold every writer's contribution.
//   kind 2 (control): kind 0 with PLAIN stores and loads (no sc1) — what the scope bits are there for; stale reads
//           are counted, not judged.
// Eight one-wavefront workgroups form a group, blockIdx = 8 * group + member: consecutive workgroups are dealt out to
// the eight XCDs round robin, so every member sits behind another L2.  A group runs its iterations in lock step
// (`go` word), every iteration passes eight messages.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

namespace {
typedef uint32_t u32;
typedef uint64_t u64;

struct LitmusArgs {
    u32 *data;   // [groups][8] payloads
    u32 *arr;    // [groups][iters] arrival words (kind 0 / 2), zero
    u32 *cnt;    // [groups] counters (kind 1)
    u32 *flag;   // [groups][8] (kind 1)
    u32 *go;     // [groups]
    u64 *stale;  // [1]
    u64 *done;   // [1] messages checked
    u32 iters;
    int kind;
    u32 *xcc;    // [groups][8] XCC_ID of each member (s_getreg), for the test's report
};

__device__ __forceinline__ void st_agent(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_agent(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64) void litmus_kernel(LitmusArgs a)
{
    if (threadIdx.x != 0) return;
    const u32 g = blockIdx.x >> 3, w = blockIdx.x & 7u;
    {
        u32 id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[g * 8u + w] = id & 0xFu;
    }
    u32 *data = a.data + g * 8u, *flag = a.flag + g * 8u;
    u64 stale = 0, checked = 0;
    for (u32 i = 0; i < a.iters; i++) {
        const u32 tag = i + 1u;
        if (a.kind == 1) {
            const u32 t = __hip_atomic_fetch_add(&a.cnt[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(t)); // (the returning atomic is waited for: finish_compact.h does the same)
            st_agent(&flag[w], tag);
            if (w == (i & 7u)) { // this iteration's reader
                for (u32 k = 0; k < 8u; k++)
                    while (ld_agent(&flag[k]) != tag) __builtin_amdgcn_s_sleep(1);
                const u32 c = ld_agent(&a.cnt[g]);
                stale += c < 8u * tag ? 1u : 0u;
                checked += 8;
                st_agent(&a.go[g], tag);
            }
        } else {
            if (a.kind == 0) st_agent(&data[w], tag);
            else *(volatile u32 *)&data[w] = tag;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every store above has been acknowledged
            const u32 t = __hip_atomic_fetch_add(&a.arr[(u64)g * a.iters + i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == 7u) { // the last to arrive reads what the others wrote
                for (u32 k = 0; k < 8u; k++) {
                    const u32 v = a.kind == 0 ? ld_agent(&data[k]) : *(volatile u32 *)&data[k];
                    stale += v != tag ? 1u : 0u;
                }
                checked += 8;
                st_agent(&a.go[g], tag);
            }
        }
        while (ld_agent(&a.go[g]) < tag) __builtin_amdgcn_s_sleep(1);
    }
    if (stale) atomicAdd((unsigned long long *)a.stale, (unsigned long long)stale);
    if (checked) atomicAdd((unsigned long long *)a.done, (unsigned long long)checked);
}
} // namespace

// kind: 0 / 1 / 2 (above); groups x 8 workgroups must be resident together (the members wait for each other): groups <= 512.
// out[0] = stale reads, out[1] = messages checked, out[2] = distinct XCCs seen among one group's members (8 on an SPX MI355X).
extern "C" int yacrd_litmus_run(int kind, unsigned groups, unsigned iters, unsigned long long *out)
{
    if (kind < 0 || kind > 2 || groups == 0 || groups > 512 || iters == 0 || !out) return 1;
    LitmusArgs a{};
    a.iters = iters, a.kind = kind;
    u64 *res = nullptr;
    auto bytes = [&](size_t n) { return n * sizeof(u32); };
    if (hipMalloc(&a.data, bytes((size_t)groups * 8)) != hipSuccess) return 2;
    if (hipMalloc(&a.flag, bytes((size_t)groups * 8)) != hipSuccess) return 2;
    if (hipMalloc(&a.xcc, bytes((size_t)groups * 8)) != hipSuccess) return 2;
    if (hipMalloc(&a.arr, bytes((size_t)groups * iters)) != hipSuccess) return 2;
    if (hipMalloc(&a.cnt, bytes(groups)) != hipSuccess) return 2;
    if (hipMalloc(&a.go, bytes(groups)) != hipSuccess) return 2;
    if (hipMalloc(&res, 2 * sizeof(u64)) != hipSuccess) return 2;
    a.stale = res, a.done = res + 1;
    (void)hipMemset(a.data, 0, bytes((size_t)groups * 8));
    (void)hipMemset(a.flag, 0, bytes((size_t)groups * 8));
    (void)hipMemset(a.xcc, 0xFF, bytes((size_t)groups * 8));
    (void)hipMemset(a.arr, 0, bytes((size_t)groups * iters));
    (void)hipMemset(a.cnt, 0, bytes(groups));
    (void)hipMemset(a.go, 0, bytes(groups));
    (void)hipMemset(res, 0, 2 * sizeof(u64));
    hipLaunchKernelGGL(litmus_kernel, dim3(groups * 8), dim3(64), 0, 0, a);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 3;
    u64 h[2] = {0, 0};
    u32 x[8] = {};
    if (rc == 0 && hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = 3;
    if (rc == 0 && hipMemcpy(x, a.xcc, sizeof(x), hipMemcpyDeviceToHost) != hipSuccess) rc = 3;
    unsigned seen = 0;
    for (int i = 0; i < 8; i++) seen |= 1u << (x[i] & 15u);
    out[0] = h[0], out[1] = h[1], out[2] = (unsigned long long)__builtin_popcount(seen);
    (void)hipFree(a.data), (void)hipFree(a.flag), (void)hipFree(a.xcc), (void)hipFree(a.arr), (void)hipFree(a.cnt), (void)hipFree(a.go), (void)hipFree(res);
    return rc;
}
