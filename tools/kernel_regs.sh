#!/bin/bash
# usage: tools/kernel_regs.sh [pattern] [extra hipcc flags...] : VGPRs / SGPRs / LDS / scratch of the device kernels
# of engine.hip (device-only assembly, gfx950), e.g. tools/kernel_regs.sh fused -DYK_SCREEN_WINDOW=64
pat=${1:-.}; shift
cd "$(dirname "$0")/../yacrd_amd/csrc" || exit 1
out=$(mktemp /tmp/engine_XXXX.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math --offload-device-only -S "$@" -o "$out" engine.hip || exit 1
awk '/^[ \t]*\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_next_free_sgpr/{s=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /\.amdhsa_accum_offset/{a=$2} /^[ \t]*\.end_amdhsa_kernel/{printf "%-110s vgpr %3d (accum_offset %3d) sgpr %3d lds %6d scratch %5d\n", k, v, a, s, l, p}' "$out" | grep -E "$pat"
echo "asm: $out"
