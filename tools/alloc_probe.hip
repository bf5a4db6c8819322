// tools/alloc_probe.hip — what hipMalloc / hipFree / first touch of large HBM buffers cost (the device parser's first
// call on a 15 GB file spends more time allocating than moving and parsing).  hipcc --offload-arch=gfx950 -O2 -o alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(nullptr);
    for (size_t gb : {1, 4, 16}) {
        for (int rep = 0; rep < 2; rep++) {
            void *p = nullptr;
            const size_t bytes = gb << 30;
            double t0 = now();
            if (hipMalloc(&p, bytes) != hipSuccess) { printf("malloc failed\n"); return 1; }
            double t1 = now();
            hipMemset(p, 0, bytes);
            hipDeviceSynchronize();
            double t2 = now();
            hipMemset(p, 1, bytes);
            hipDeviceSynchronize();
            double t3 = now();
            hipFree(p);
            double t4 = now();
            printf("%2zu GB rep %d: hipMalloc %.1f ms, first memset %.1f ms, second memset %.1f ms, hipFree %.1f ms\n", gb, rep, t1 - t0, t2 - t1,
                   t3 - t2, t4 - t3);
        }
    }
    // virtual memory management: reserve, then map physical chunks one by one
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess) {
        const size_t chunk = ((size_t)1 << 30) / gran * gran, total = 8 * chunk;
        void *va = nullptr;
        double t0 = now();
        hipError_t e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
        double t1 = now();
        printf("granularity %zu, reserve 8 GB: %s %.2f ms\n", gran, hipGetErrorString(e), t1 - t0);
        if (e == hipSuccess) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            for (int k = 0; k < 8; k++) {
                hipMemGenericAllocationHandle_t h;
                double a = now();
                hipError_t e1 = hipMemCreate(&h, chunk, &prop, 0);
                double b = now();
                hipError_t e2 = e1 == hipSuccess ? hipMemMap((char *)va + k * chunk, chunk, 0, h, 0) : e1;
                hipError_t e3 = e2 == hipSuccess ? hipMemSetAccess((char *)va + k * chunk, chunk, &acc, 1) : e2;
                double c = now();
                if (k < 3 || e3 != hipSuccess) printf("  chunk %d: create %.1f ms (%s), map + access %.1f ms (%s)\n", k, b - a, hipGetErrorString(e1), c - b, hipGetErrorString(e3));
                if (e3 != hipSuccess) break;
            }
            double a = now();
            hipMemset(va, 0, total);
            hipDeviceSynchronize();
            printf("  memset of the mapped 8 GB: %.1f ms\n", now() - a);
        }
    }
    return 0;
}
