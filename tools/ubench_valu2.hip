// Micro-benchmark 2: sustained issue cost of the instructions the Harris / octave kernels are built
// from (f64 path vs f32/int alternatives, packed f32, SDWA, DPP, 3-operand forms) on gfx950.
// 16 independent chains per lane, 8 blocks of 256 threads per CU.  Reports ns and shader cycles per
// wave64 instruction per SIMD (clock = s_memtime ticks / s_memrealtime ticks * 100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu2 ubench_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL32(NAME, ASM)                                                                             \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, unsigned long long* cyc, uint32_t s0, uint32_t s1, int iters) { \
        uint32_t a[16];                                                                                 \
        uint32_t x = threadIdx.x * 2654435761u + s0, y = x ^ 0x5555u;                                  \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = x + i;                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();    \
        for (int it = 0; it < iters; ++it) {                                                            \
            REP16(ASM)                                                                                  \
        }                                                                                               \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();    \
        uint32_t r = 0;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) r ^= a[i];                                       \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                        \
        if (threadIdx.x == 0) cyc[2 * blockIdx.x] = t1 - t0, cyc[2 * blockIdx.x + 1] = r1 - r0;        \
    }
#define KERNEL64(NAME, ASM)                                                                             \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, unsigned long long* cyc, uint32_t s0, uint32_t s1, int iters) { \
        double a[16];                                                                                   \
        uint32_t x = threadIdx.x * 2654435761u + s0, y = x ^ 0x5555u;                                  \
        double dx = 1.0 + 1e-9 * threadIdx.x, dy = 0.999999;                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = dx + i;                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();    \
        for (int it = 0; it < iters; ++it) {                                                            \
            REP16(ASM)                                                                                  \
        }                                                                                               \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();    \
        double r = 0;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) r += a[i];                                       \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(long long)r + x + y;                           \
        if (threadIdx.x == 0) cyc[2 * blockIdx.x] = t1 - t0, cyc[2 * blockIdx.x + 1] = r1 - r0;        \
    }

#define A1(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_add_u32, A1)
#define A2(i) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_add3_u32, A2)
#define A3(i) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_mul_i32_i24, A3)
#define A4(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_mad_i32_i24, A4)
#define A5(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_mul_lo_u32, A5)
#define A6(i) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_mul_hi_u32, A6)
#define A7(i) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_mul_hi_u32_u24, A7)
#define A8(i) asm volatile("v_mul_i32_i24_sdwa %0, sext(%1), sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_mul_i32_i24_sdwa, A8)
#define A9(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i]));
KERNEL32(k_cvt_f32_i32_sdwa, A9)
#define A10(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
KERNEL32(k_cvt_f32_i32, A10)
#define A11(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_mul_f32, A11)
#define A12(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_sub_f32, A12)
#define A13(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_max_f32, A13)
#define A14(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_max3_f32, A14)
#define A15(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
KERNEL32(k_rndne_f32, A15)
#define A16(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
KERNEL32(k_mov_dpp_wave_shr, A16)
#define A17(i) asm volatile("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
KERNEL32(k_add_u32_dpp_wave_shr, A17)
#define A18(i) asm volatile("v_max_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
KERNEL32(k_max_f32_dpp_row_shr, A18)
#define A19(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
KERNEL32(k_cmp_cndmask_pair, A19)
#define A20(i) asm volatile("v_alignbit_b32 %0, %1, %0, 16" : "+v"(a[i]) : "v"(x));
KERNEL32(k_alignbit, A20)
#define A21(i) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_pk_mul_lo_u16, A21)
#define A22(i) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_pk_mad_u16, A22)
#define A23(i) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_pk_sub_i16, A23)
#define A24(i) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_dot2_i32_i16, A24)
#define A25(i) asm volatile("v_bfe_i32 %0, %0, 16, 16" : "+v"(a[i]));
KERNEL32(k_bfe_i32, A25)
#define A26(i) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_and_or_b32, A26)
#define A27(i) asm volatile("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_lshl_add_u32, A27)
#define A28(i) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_med3_f32, A28)
#define A29(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_min_f32, A29)
#define A30(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_fma_f32, A30)
#define A31(i) asm volatile("v_ashrrev_i32 %0, 16, %0" : "+v"(a[i]));
KERNEL32(k_ashrrev_i32, A31)
#define A32(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_mad_u32_u24, A32)
#define A33(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_sub_u32, A33)
#define A34(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
KERNEL32(k_lshlrev_b32, A34)
#define A35(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_and_b32, A35)
#define A36(i) asm volatile("v_cvt_f32_ubyte2 %0, %0" : "+v"(a[i]));
KERNEL32(k_cvt_f32_ubyte2, A36)
#define A37(i) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_pk_add_u16, A37)
#define A38(i) asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(a[i]) : "v"(x));
KERNEL32(k_pk_max_u16, A38)
#define A39(i) asm volatile("v_pk_lshrrev_b16 %0, 4, %0" : "+v"(a[i]));
KERNEL32(k_pk_lshrrev_b16, A39)
#define A40(i) asm volatile("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_mad_u32_u16_opsel, A40)
#define A41(i) asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,1,0,0]" : "+v"(a[i]) : "v"(x), "v"(y));
KERNEL32(k_mad_i32_i16_opsel, A41)

#define D1(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(dx), "v"(dy));
KERNEL64(k_fma_f64, D1)
#define D2(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "v"(dy));
KERNEL64(k_mul_f64, D2)
#define D3(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[i]) : "v"(dx));
KERNEL64(k_add_f64, D3)
#define D4(i) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a[i]) : "v"(x));
KERNEL64(k_cvt_f64_i32, D4)
#define D5(i) asm volatile("v_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %0, %1" : "+v"(a[i]), "+v"(x));
KERNEL64(k_cvt_f32_f64_plus_back, D5)
#define D6(i) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(a[i]) : "v"(x));
KERNEL64(k_cvt_f64_f32, D6)
#define D7(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
KERNEL64(k_mad_u64_u32, D7)
#define D8(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(dx));
KERNEL64(k_pk_mul_f32, D8)
#define D9(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(dx));
KERNEL64(k_pk_add_f32, D9)
#define D10(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(dx), "v"(dy));
KERNEL64(k_pk_fma_f32, D10)
#define D11(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
KERNEL64(k_lshlrev_b64, D11)

typedef void (*kfn)(uint32_t*, unsigned long long*, uint32_t, uint32_t, int);
static void run(kfn k, const char* name, int per_iter = 16) {
    const int iters = 4096, blocks = 256 * 8;
    uint32_t* d;
    unsigned long long* c;
    (void)hipMalloc(&d, blocks * 256 * 4);
    (void)hipMalloc(&c, blocks * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, 3u, 0x01020304u, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, 3u, 0x01020304u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[4096];
    (void)hipMemcpy(h, c, blocks * 16, hipMemcpyDeviceToHost);
    double clk = 0;
    for (int b = 0; b < blocks; ++b) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 100e6;
    clk /= blocks;
    const double wave_instr_per_simd = (double)blocks * 4 * iters * per_iter / 1024.0;
    const double ns = ms * 1e6 / wave_instr_per_simd;
    printf("%-28s wall %7.3f ms  clock %5.2f GHz  %5.2f ns = %5.2f cycles per wave-instr per SIMD\n", name, ms, clk / 1e9, ns, ns * clk / 1e9);
    (void)hipFree(d);
    (void)hipFree(c);
}
#define RUN(k) run(k, &#k[2])
int main() {
    RUN(k_add_u32); RUN(k_sub_u32); RUN(k_and_b32); RUN(k_lshlrev_b32); RUN(k_ashrrev_i32); RUN(k_add3_u32); RUN(k_lshl_add_u32); RUN(k_and_or_b32);
    RUN(k_mul_i32_i24); RUN(k_mad_i32_i24); RUN(k_mad_u32_u24); RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_mul_hi_u32_u24);
    RUN(k_mul_i32_i24_sdwa); RUN(k_mad_u32_u16_opsel); RUN(k_mad_i32_i16_opsel); RUN(k_bfe_i32);
    RUN(k_cvt_f32_i32); RUN(k_cvt_f32_i32_sdwa); RUN(k_cvt_f32_ubyte2);
    RUN(k_mul_f32); RUN(k_sub_f32); RUN(k_fma_f32); RUN(k_max_f32); RUN(k_min_f32); RUN(k_max3_f32); RUN(k_med3_f32); RUN(k_rndne_f32);
    run(k_cmp_cndmask_pair, "cmp_lt_f32 + cndmask (2)", 32);
    RUN(k_mov_dpp_wave_shr); RUN(k_add_u32_dpp_wave_shr); RUN(k_max_f32_dpp_row_shr);
    RUN(k_alignbit); RUN(k_pk_add_u16); RUN(k_pk_sub_i16); RUN(k_pk_max_u16); RUN(k_pk_lshrrev_b16); RUN(k_pk_mul_lo_u16); RUN(k_pk_mad_u16); RUN(k_dot2_i32_i16);
    RUN(k_fma_f64); RUN(k_mul_f64); RUN(k_add_f64); RUN(k_cvt_f64_i32); RUN(k_cvt_f64_f32);
    run(k_cvt_f32_f64_plus_back, "cvt_f32_f64 + cvt_f64_f32 (2)", 32);
    RUN(k_mad_u64_u32); RUN(k_lshlrev_b64); RUN(k_pk_mul_f32); RUN(k_pk_add_f32); RUN(k_pk_fma_f32);
    return 0;
}
