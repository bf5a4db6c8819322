// tools/launch_probe.hip — how fast can the MI355X start wavefronts?  Kernels that do nothing (or hold
// a register / LDS footprint like sweep_small_fused_defer_kernel's) over grids of 25 234 one-wavefront
// workgroups and 6 309 four-wavefront workgroups; dispatch timestamps via hipExtLaunchKernelGGL.
//   hipcc --offload-arch=gfx950 -O3 -o tools/launch_probe tools/launch_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_empty(unsigned *out) {}

template <int LDSW>
__global__ __launch_bounds__(256) void k_foot(unsigned *out, unsigned n)
{
    __shared__ unsigned s[LDSW];
    // ~70 live registers: values that depend on a runtime argument and all meet at the end
    unsigned v[64];
#pragma unroll
    for (int i = 0; i < 64; i++) v[i] = n * (i + 1) + threadIdx.x;
    s[threadIdx.x % LDSW] = v[3];
    __syncthreads();
    unsigned acc = s[(threadIdx.x + 1) % LDSW];
#pragma unroll
    for (int i = 0; i < 64; i++) acc ^= v[i] * (acc | 1u);
    if (acc == 0x12345u) out[0] = acc; // never
}

// a wavefront that waits for `spin` clock ticks (s_memtime) — lifetime without issue pressure
__global__ void k_sleep(unsigned *out, unsigned spin)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(1);
}

template <typename F>
static float timed(F launch, int reps = 30)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ms;
    for (int i = 0; i < reps + 3; i++) {
        launch(a, b);
        hipEventSynchronize(b);
        float t;
        hipEventElapsedTime(&t, a, b);
        if (i >= 3) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1000.f;
}

int main()
{
    unsigned *d;
    hipMalloc(&d, 4096);
    const unsigned W = 25234;
    struct { const char *name; unsigned grid, block; } shapes[] = {
        {"25234 x 64", W, 64}, {"12617 x 128", W / 2, 128}, {"6309 x 256", (W + 3) / 4, 256}, {"7168 x 64", 7168, 64},
        {"1024 x 64", 1024, 64}, {"100936 x 64", 4 * W, 64}};
    for (auto &s : shapes) {
        float e = timed([&](hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(k_empty, dim3(s.grid), dim3(s.block), 0, 0, a, b, 0, d);
        });
        float f = timed([&](hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL((k_foot<816>), dim3(s.grid), dim3(s.block), 0, 0, a, b, 0, d, 0u);
        });
        float z1 = timed([&](hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(k_sleep, dim3(s.grid), dim3(s.block), 0, 0, a, b, 0, d, 500u);
        });
        printf("%-14s empty %7.2f us   footprint(64+ VGPR, 3.2 KB LDS) %7.2f us   sleep(5 us) %7.2f us\n", s.name, e, f, z1);
    }
    return 0;
}
