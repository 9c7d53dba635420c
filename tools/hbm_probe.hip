// HBM rates of the box by access mix: read-only, write-only, copy (16 bytes per lane, 4 GiB buffers, grid-stride).
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_probe tools/hbm_probe.hip && tools/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ a, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(a) + i);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ a, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = make_uint4(seed, (unsigned)i, seed, seed);
}
__global__ __launch_bounds__(256) void k_write_nt(uint4* __restrict__ a, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { u4 v = {seed, (unsigned)i, seed, seed}; __builtin_nontemporal_store(v, reinterpret_cast<u4*>(a) + i); }
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// 1 read : R writes (the pyramid kernels write 11 bytes per byte read)
template <int R>
__global__ __launch_bounds__(256) void k_fan(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = a[i];
#pragma unroll
        for (int r = 0; r < R; ++r) b[(size_t)r * n + i] = make_uint4(v.x + r, v.y, v.z, v.w);
    }
}
int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    uint4 *a, *b;
    unsigned* out;
    CK(hipMalloc((void**)&a, bytes));
    CK(hipMalloc((void**)&b, bytes));
    CK(hipMalloc((void**)&out, 4));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto t = [&](auto f) {
        f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / 5;
    };
    const int g = 256 * 32;
    const float tr = t([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, out); });
    const float tw = t([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, a, n, 7u); });
    const float twn = t([&] { hipLaunchKernelGGL(k_write_nt, dim3(g), dim3(256), 0, 0, a, n, 7u); });
    const float tc = t([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); });
    const size_t n8 = n / 8;
    const float tf = t([&] { hipLaunchKernelGGL(k_fan<8>, dim3(g), dim3(256), 0, 0, a, b, n8); });
    std::printf("{\"read_GBps\": %.0f, \"write_GBps\": %.0f, \"write_nontemporal_GBps\": %.0f, \"copy_GBps_read_plus_write\": %.0f, \"fan_1_read_8_writes_GBps\": %.0f}\n",
                bytes / tr / 1e6, bytes / tw / 1e6, bytes / twn / 1e6, 2.0 * bytes / tc / 1e6, 9.0 * n8 * 16 / tf / 1e6);
    return 0;
}
