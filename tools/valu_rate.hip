// valu_rate.hip — issue rate of the VALU instructions the register sort is made of, on gfx950.
// Each wavefront runs ITER x 16 independent instructions of one kind (8 waves per SIMD resident);
// cycles per instruction per SIMD = time * clock * SIMDs / (waves * ITER * 16).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ __launch_bounds__(256) void rate(unsigned *out, int iters, unsigned seed)
{
    unsigned x[16], y = seed + threadIdx.x, z = seed * 3u + 1u;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = seed + i * 977u + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#define OP_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MINF(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MED3U(i) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_MED3F(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y));
#define OP_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y));
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_PKMIN(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MAX3(i) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_MINDPP(i) asm volatile("v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y));
#define OP_SUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]));
#define OP_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[i]), "v"(y) : "vcc");
#define OP_CMPS(i) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(x[i]), "v"(y) : "s20", "s21");
#define OP_CNDS(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(y));
#define OP_MAXU(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MINI(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MIN3(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_BFI(i) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_MINU16(i) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_SAD(i) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_PKMAXI(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MAXU16(i) asm volatile("v_max_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MED3U16(i) asm volatile("v_med3_u16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define OP_MINU32E64(i) asm volatile("v_min_u32_e64 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y));
#define OP_MINU16DPP(i) asm volatile("v_min_u16_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y));
#define OP_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x[i]) : "v"(y) : "vcc");
#define OP_ASHR(i) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x[i]));
#define OP_SUBREV(i) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define OP_MINU16SDWA(i) asm volatile("v_min_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(x[i]) : "v"(y));
        if (KIND == 27) { REP16(OP_AND) }
        if (KIND == 28) { REP16(OP_OR) }
        if (KIND == 29) { REP16(OP_MAXU16) }
        if (KIND == 30) { REP16(OP_MED3U16) }
        if (KIND == 31) { REP16(OP_MINU32E64) }
        if (KIND == 32) { REP16(OP_ADDDPP) }
        if (KIND == 33) { REP16(OP_MINU16DPP) }
        if (KIND == 34) { REP16(OP_ADDCO) }
        if (KIND == 35) { REP16(OP_ASHR) }
        if (KIND == 36) { REP16(OP_SUBREV) }
        if (KIND == 37) { REP16(OP_MINU16SDWA) }
        if (KIND == 11) { REP16(OP_SUB) }
        if (KIND == 12) { REP16(OP_XOR) }
        if (KIND == 13) { REP16(OP_LSHL) }
        if (KIND == 14) { REP16(OP_CMP) }
        if (KIND == 15) { REP16(OP_CMPS) }
        if (KIND == 16) { REP16(OP_CNDS) }
        if (KIND == 17) { REP16(OP_MAXU) }
        if (KIND == 18) { REP16(OP_MINI) }
        if (KIND == 19) { REP16(OP_MIN3) }
        if (KIND == 20) { REP16(OP_BFI) }
        if (KIND == 21) { REP16(OP_ADD3) }
        if (KIND == 22) { REP16(OP_MINU16) }
        if (KIND == 23) { REP16(OP_MOV) }
        if (KIND == 24) { REP16(OP_SAD) }
        if (KIND == 25) { REP16(OP_PKMAXI) }
        if (KIND == 26) { REP16(OP_PKADD) }
        if (KIND == 0) { REP16(OP_MINU) }
        if (KIND == 1) { REP16(OP_MINF) }
        if (KIND == 2) { REP16(OP_MED3U) }
        if (KIND == 3) { REP16(OP_MED3F) }
        if (KIND == 4) { REP16(OP_DPP) }
        if (KIND == 5) { REP16(OP_CND) }
        if (KIND == 6) { REP16(OP_FMA) }
        if (KIND == 7) { REP16(OP_ADD) }
        if (KIND == 8) { REP16(OP_PKMIN) }
        if (KIND == 9) { REP16(OP_MAX3) }
        if (KIND == 10) { REP16(OP_MINDPP) }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= x[i];
    if (acc == 0x12345u) out[0] = acc;
}

template <int KIND>
double run(const char *name, unsigned *d)
{
    const int iters = 4096, blocks = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, 64, 7u);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = (double)blocks * 4 / (256.0 * 4) * iters * 16; // waves per SIMD * per-wave count
    const double cyc = ms * 1e-3 * 2.4e9 / insts_per_simd;
    std::printf("%-14s %8.3f ms  %.2f cycles/inst/SIMD at 2.4 GHz\n", name, ms, cyc);
    return cyc;
}

int main()
{
    unsigned *d;
    hipMalloc(&d, 64);
    run<0>("v_min_u32", d);
    run<1>("v_min_f32", d);
    run<2>("v_med3_u32", d);
    run<3>("v_med3_f32", d);
    run<4>("v_mov_dpp", d);
    run<5>("v_cndmask", d);
    run<6>("v_fma_f32", d);
    run<7>("v_add_u32", d);
    run<8>("v_pk_min_u16", d);
    run<9>("v_max3_u32", d);
    run<10>("v_min_u32_dpp", d);
    run<11>("v_sub_u32", d);
    run<12>("v_xor_b32", d);
    run<13>("v_lshlrev_b32", d);
    run<14>("v_cmp vcc", d);
    run<15>("v_cmp sgpr", d);
    run<16>("v_cndmask sgpr", d);
    run<17>("v_max_u32", d);
    run<18>("v_min_i32", d);
    run<19>("v_min3_u32", d);
    run<20>("v_bfi_b32", d);
    run<21>("v_add3_u32", d);
    run<22>("v_min_u16", d);
    run<23>("v_mov_b32", d);
    run<24>("v_sad_u32", d);
    run<25>("v_pk_max_i16", d);
    run<26>("v_pk_add_u16", d);
    run<27>("v_and_b32", d);
    run<28>("v_or_b32", d);
    run<29>("v_max_u16", d);
    run<30>("v_med3_u16", d);
    run<31>("v_min_u32_e64", d);
    run<32>("v_add_u32_dpp", d);
    run<33>("v_min_u16_dpp", d);
    run<34>("v_add_co_u32", d);
    run<35>("v_ashrrev_i32", d);
    run<36>("v_subrev_u32", d);
    run<37>("v_min_u16_sdwa", d);
    return 0;
}
