// Probe (NOT part of the product): does a low-priority stream's hardware queue get starved when the normal-priority
// main stream's queue is parked on a barrier that waits for it?  (one side queue ran its kernels
// 2-5x slower in some GPU_MAX_HW_QUEUES settings.)  For each of N low-priority streams: time a many-workgroup kernel on
// it (a) with the main stream idle, (b) with the main stream waiting for its end event and more work queued behind.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/queue_probe && GPU_MAX_HW_QUEUES=3 tools/queue_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

__global__ void k_touch(uint4* p, size_t n, int rounds) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 v = p[i];
    for (int r = 0; r < rounds; ++r) v.x = v.x * 1664525u + 1013904223u + v.y;
    p[i] = v;
}

int main(int argc, char** argv) {
    const int n_side = argc > 1 ? std::atoi(argv[1]) : 6;
    const size_t n = (size_t)64 << 20;  // 1 GiB of uint4: 262144 workgroups of 256
    uint4* buf;
    CK(hipMalloc((void**)&buf, n * sizeof(uint4)));
    CK(hipMemset(buf, 1, n * sizeof(uint4)));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t main_s;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    std::vector<hipStream_t> side((size_t)n_side);
    for (auto& s : side) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));
    hipEvent_t a, b, fork;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    const dim3 grid((unsigned)(n / 256)), block(256);
    auto run = [&](hipStream_t s, bool park_main) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(fork, main_s));
            CK(hipStreamWaitEvent(s, fork, 0));
            CK(hipEventRecord(a, s));
            k_touch<<<grid, block, 0, s>>>(buf, n, 8);
            CK(hipEventRecord(b, s));
            if (park_main) {  // the join of a batch call: main waits for the side stream, the next batch's kernels behind it
                CK(hipStreamWaitEvent(main_s, b, 0));
                k_touch<<<dim3(1024), block, 0, main_s>>>(buf, 1024 * 256, 1);
            }
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        return best;
    };
    k_touch<<<grid, block, 0, main_s>>>(buf, n, 8);  // first use of every stream binds its hardware queue
    for (auto& s : side) k_touch<<<dim3(64), block, 0, s>>>(buf, 64 * 256, 1);
    CK(hipDeviceSynchronize());
    std::printf("main stream alone: %.3f ms\n", run(main_s, false));
    for (int i = 0; i < n_side; ++i) {
        const float free_ms = run(side[(size_t)i], false), parked_ms = run(side[(size_t)i], true);
        std::printf("low-priority stream %d: %.3f ms with main idle, %.3f ms with main parked on its end event%s\n", i, free_ms, parked_ms,
                    parked_ms > 1.5f * free_ms ? "   <-- starved" : "");
    }
    return 0;
}
