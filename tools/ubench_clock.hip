// What clock and how many CUs do we really get?  (s_memtime = shader clock, wall via events)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cyc, unsigned long long* rt, uint32_t s0, uint32_t s1, int iters) {
    uint32_t a[16];
    uint32_t x = threadIdx.x * 2654435761u + s0;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_udot4(x, s1, a[i], false);
        x += s0;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clockRate=%d kHz memClock=%d kHz busWidth=%d l2=%d MB\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.memoryBusWidth, p.l2CacheSize >> 20);
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        const int blocks = 256 * bpc, iters = 8192;
        uint32_t* d; unsigned long long *c, *r; (void)hipMalloc(&d, blocks * 256 * 4); (void)hipMalloc(&c, blocks * 8); (void)hipMalloc(&r, blocks * 8);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, r, 3u, 0x01020304u, 16); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, r, 3u, 0x01020304u, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[4], hr[4]; (void)hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, r, 32, hipMemcpyDeviceToHost);
        double clk_ghz = (double)hc[0] / ((double)hr[0] / 100e6) / 1e9;
        double inst_per_wave = (double)iters * 16;
        printf("blocks/CU=%d wall=%.3f ms  block0: %llu cycles, %.3f ms realtime -> clock %.2f GHz; cycles per wave-instr per SIMD-resident-wave-set: %.2f\n", bpc, ms, hc[0], hr[0] / 100e3, clk_ghz,
               (double)hc[0] / (inst_per_wave * bpc));
    }
    return 0;
}
