// Micro-benchmark: sustained issue cost (cycles per wave64 instruction per SIMD) of candidate
// instructions for the Gaussian passes on gfx950, 8 waves per SIMD, 16 independent chains.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cyc, uint32_t s0, uint32_t s1, int iters) {
    uint32_t a[16];
    uint32_t x = threadIdx.x * 2654435761u + s0, y = x ^ 0x5555u;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define ONE(i)                                                                                         \
    if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));                          \
    if (OP == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));                          \
    if (OP == 2) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(x));                      \
    if (OP == 3) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));          \
    if (OP == 4) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(s1));         \
    if (OP == 5) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(s1));        \
    if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(s1));            \
    if (OP == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));                 \
    if (OP == 8) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(s1));             \
    if (OP == 9) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(x));                       \
    if (OP == 10) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(x));                  \
    if (OP == 11) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));                        \
    if (OP == 12) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));                               \
    if (OP == 13) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));            \
    if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x));                             \
    if (OP == 15) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));           \
    if (OP == 16) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));         \
    if (OP == 17) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(a[i]) : "s"(s1), "v"(y));           \
    if (OP == 18) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "s"(s1), "v"(y));
        OPS(ONE)
        x += s0;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name) {
    const int iters = 4096, blocks = 256 * 8;
    uint32_t* d; unsigned long long* c; (void)hipMalloc(&d, blocks * 256 * 4); (void)hipMalloc(&c, blocks * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, c, 3u, 0x01020304u, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, c, 3u, 0x01020304u, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; (void)hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    // all 8 blocks of a CU are co-resident (8 waves/SIMD): cycles the CU spent / wave-instrs per SIMD
    double wave_instr_per_simd = 8.0 * iters * 16;
    printf("%-26s wall %7.3f ms  block0 %9llu cyc  -> %5.2f cyc per wave-instr per SIMD (if 8 waves co-resident)\n", name, ms, hc,
           (double)hc / wave_instr_per_simd);
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    run<0>("v_add_u32 (VOP2)");
    run<1>("v_xor_b32 (VOP2)");
    run<11>("v_and_b32 v,s (VOP2)");
    run<14>("v_mov_b32 (VOP1)");
    run<12>("v_cvt_f32_ubyte1 (VOP1)");
    run<2>("v_mul_u32_u24 (VOP2)");
    run<3>("v_mad_u32_u24 (VOP3)");
    run<10>("v_lshl_or_b32 (VOP3)");
    run<6>("v_perm_b32 (VOP3)");
    run<4>("v_dot4_u32_u8 (VOP3P)");
    run<13>("v_dot4c_i32_i8 (VOP2)");
    run<16>("v_dot4_u32_u8 v,v (VOP3P)");
    run<17>("v_dot4c_i32_i8 s,v (VOP2)");
    run<5>("v_dot2_u32_u16 (VOP3P)");
    run<15>("v_dot2c_i32_i16 (VOP2)");
    run<18>("v_dot2c_i32_i16 s,v (VOP2)");
    run<9>("v_pk_add_u16 (VOP3P)");
    run<7>("v_fmac_f32 (VOP2)");
    run<8>("v_fma_f32 (VOP3)");
    return 0;
}
