// sweep_wave.h — small class: one read per wavefront.
#pragma once
#include "device_common.h"
#include "sweep_lds.h"

namespace yk {

inline void launch_sweep_wave(const SweepArgs &sa, u32 n_reads, int num_cu, hipStream_t stream)
{
    const u32 grid = (u32)((u64)n_reads < (u64)num_cu * 32 ? (u64)n_reads : (u64)num_cu * 32);
    hipLaunchKernelGGL((sweep_lds_kernel<64, (int)kSmallEvents>), dim3(grid), dim3(64), 0, stream,
                       sa);
}

} // namespace yk
