// PCIe copy rates of the GPU box: H2D and D2H alone and concurrently on two streams, pinned host memory,
// sizes of one 256-frame 1080p batch (531 MB up, 645 MB of packed lists down).  Host-only HIP API.
//   g++ -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/pcie_probe.cpp -L/opt/rocm/lib -lamdhip64 -o tools/pcie_probe
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(e)                                                                      \
    do {                                                                           \
        hipError_t r_ = (e);                                                       \
        if (r_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));           \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

int main() {
    const size_t up = 530841600, down = 645000000;
    void *hu, *hd, *du, *dd;
    CK(hipHostMalloc(&hu, up, hipHostMallocDefault));
    CK(hipHostMalloc(&hd, down, hipHostMallocDefault));
    CK(hipMalloc(&du, up));
    CK(hipMalloc(&dd, down));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto run = [&](bool do_up, bool do_down, int reps) {
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) {
            if (do_up) CK(hipMemcpyAsync(du, hu, up, hipMemcpyHostToDevice, s1));
            if (do_down) CK(hipMemcpyAsync(hd, dd, down, hipMemcpyDeviceToHost, s2));
        }
        CK(hipDeviceSynchronize());
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    run(true, true, 2);
    const double tu = run(true, false, 5), td = run(false, true, 5), tb = run(true, true, 5);
    std::printf("{\"h2d_alone_GBps\": %.1f, \"d2h_alone_GBps\": %.1f, \"both_ms\": %.2f, \"both_h2d_GBps\": %.1f, \"both_d2h_GBps\": %.1f, \"alone_ms\": [%.2f, %.2f]}\n",
                up / tu / 1e9, down / td / 1e9, tb * 1e3, up / tb / 1e9, down / tb / 1e9, tu * 1e3, td * 1e3);
    return 0;
}
