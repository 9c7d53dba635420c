// Exhaustive check of sqrt_rn_small and sqrt_rn_small_pk (visualslam_amd/csrc/kernels_generic.hip.h): for every argument the
// library can form - x*x + y*y of integer Sobel differences, integers 0 .. 2*255^2 - it must equal the
// correctly rounded f32 square root (f64 sqrt rounded once: 53 >= 2*24 + 2 bits).  Prints the number of
// mismatches; exit code 0 iff none.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/sqrt_check tools/sqrt_check.hip && /tmp/sqrt_check
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../include/vslam.h"
#include "../visualslam_amd/csrc/kernels_generic.hip.h"

__global__ void k_check(int n, unsigned int* bad, float* first) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    const float x = (float)i;
    const float a = vslam::sqrt_rn_small(x), b = (float)sqrt((double)x);
    // the packed form (k_orient_survivors_pk), the argument in either half beside another one
    const vslam::vslam_f2 p = vslam::sqrt_rn_small_pk(vslam::vslam_f2{x, (float)(n - i)});
    const float c = (float)sqrt((double)(n - i));
    const float d = vslam::sqrt_rn_small_nr(x);
    if (__float_as_uint(d) != __float_as_uint(b) || __float_as_uint(a) != __float_as_uint(b) || __float_as_uint(p.x) != __float_as_uint(b) || __float_as_uint(p.y) != __float_as_uint(c)) {
        if (atomicAdd(bad, 1u) == 0) *first = x;
    }
}

int main() {
    const int n = 2 * 255 * 255;
    unsigned int *d_bad, bad = 0;
    float *d_first, first = -1;
    if (hipMalloc((void**)&d_bad, 4) != hipSuccess || hipMalloc((void**)&d_first, 4) != hipSuccess) return 2;
    (void)hipMemset(d_bad, 0, 4);
    hipLaunchKernelGGL(k_check, dim3(n / 256 + 1), dim3(256), 0, 0, n, d_bad, d_first);
    if (hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    (void)hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    std::printf("{\"checked\": %d, \"mismatches\": %u, \"first\": %g}\n", n + 1, bad, bad ? first : -1.0);
    return bad ? 1 : 0;
}
