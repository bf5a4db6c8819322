// tools/xq_probe.hip — what does a cross-queue dependency cost in front of a kernel?
// hipcc --offload-arch=gfx950 -O2 tools/xq_probe.hip -o /tmp/xq_probe && /tmp/xq_probe
// A ~45 us kernel is launched N times back to back in stream A: (1) plain, (2) each launch
// preceded by hipStreamWaitEvent on an event of stream B that completed long ago, (3) preceded by
// a wait on an event recorded in stream B right after a tiny kernel there (a live dependency).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long cycles, unsigned *out)
{
    const unsigned long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeef) *out = x;
}
__global__ void tiny(unsigned *out) { if (threadIdx.x == 1234567) *out = 1; }
int main()
{
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned *d;
    hipMalloc(&d, 4);
    hipEvent_t old_ev, ev[64];
    hipEventCreateWithFlags(&old_ev, hipEventDisableTiming);
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    tiny<<<1, 64, 0, b>>>(d);
    hipEventRecord(old_ev, b);
    hipDeviceSynchronize();
    const unsigned long long cyc = 4500; // wall_clock64 ticks at 100 MHz -> 45 us
    const int N = 400;
    hipEvent_t st[64], en[64];
    for (int i = 0; i < 64; i++) { hipEventCreate(&st[i]); hipEventCreate(&en[i]); }
    // mode 3: start / stop events attached to the launch; mode 4: hipEventRecord before and after
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (mode == 1) hipStreamWaitEvent(a, old_ev, 0);
                if (mode == 2) {
                    tiny<<<1, 64, 0, b>>>(d);
                    hipEventRecord(ev[i & 63], b);
                    hipStreamWaitEvent(a, ev[i & 63], 0);
                }
                if (mode == 3) hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, a, st[i & 63], en[i & 63], 0, cyc, d);
                else if (mode == 4) { hipEventRecord(st[i & 63], a); spin<<<2048, 256, 0, a>>>(cyc, d); hipEventRecord(en[i & 63], a); }
                else spin<<<2048, 256, 0, a>>>(cyc, d);
            }
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("mode %d: %.2f us per iteration (kernel ~45 us)\n", mode, us / N);
        }
    }
    return 0;
}
