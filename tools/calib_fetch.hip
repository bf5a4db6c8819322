// calib_fetch.hip — calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on known byte counts, in the
// access patterns of the sweep kernel (8 B/lane coalesced loads) and of a wide copy (16 B/lane).
// MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 and is
// "uncalibrated" for other widths, so we measure the factor instead of assuming it.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read8(const uint2 *p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint2 v = p[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void read16(const uint4 *p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void write8(uint2 *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint2((unsigned)i, 7u);
}

int main()
{
    const size_t bytes = (size_t)1 << 30; // 1 GiB: well past the 256 MiB Infinity Cache
    void *buf;
    unsigned *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, (const uint2 *)buf, bytes / 8, sink);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, (uint2 *)buf, bytes / 8);
    }
    hipDeviceSynchronize();
    std::printf("each kernel moves %zu bytes (%.1f KiB)\n", bytes, bytes / 1024.0);
    return 0;
}
