// K-H1: fused Harris front end for gfx950 -- 8-bit frame in, f32 corner response out.
// Replaces GaussianBlur(3x3) -> Sobel x/y (ksize 1) -> copyMakeBorder -> StructureMatrix
// -> determinant/trace -> response of Harris_corners.cpp:158-172 / :10-68 in one pass:
// the frame is read from HBM once (plus a 3-pixel halo) and only the response is written.
//
// Exactness: every intermediate is an integer (|Ix|,|Iy| <= 255, 3x3
// sums <= 585,225 < 2^24), so the reference's f32 accumulations are exact and are done
// here in int32; det = Sxx*Syy - Sxy^2 is formed exactly in f64 (< 2^53) and rounded once
// to f32 like cv::determinant's double result narrowed at Harris_corners.cpp:54; the
// three f32 operations of :57 are individually rounded (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_generic.hip.h"

namespace vslam {

constexpr int HT_W = 64;  // output tile width  (one wave = one tile row)
constexpr int HT_H = 16;  // output tile height

// Border semantics inside the tile:
//  * raw/B cells outside the image hold the BORDER_REFLECT_101 extension (filled at load;
//    B of a reflect-extended image is again reflect-symmetric because [1 2 1] is symmetric),
//    so GaussianBlur and Sobel borders come out right with no special cases;
//  * the structure tensor reads gradients with BORDER_REPLICATE, i.e. clamped coordinates.
__global__ __launch_bounds__(256) void k_harris_fused(const uint8_t* __restrict__ img, size_t step, size_t frame,
                                                       int rows, int cols, float k, float* __restrict__ resp,
                                                       size_t rstep_elems, size_t rframe) {
    __shared__ uint8_t raw[HT_H + 6][HT_W + 8];
    __shared__ uint8_t blur[HT_H + 4][HT_W + 4];
    __shared__ short2 grad[HT_H + 2][HT_W + 2];

    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * HT_H, c0 = blockIdx.x * HT_W;
    const uint8_t* src = img + blockIdx.z * frame;

    for (int idx = tid; idx < (HT_H + 6) * (HT_W + 6); idx += 256) {
        const int i = idx / (HT_W + 6), j = idx - i * (HT_W + 6);
        raw[i][j] = src[(size_t)reflect101(r0 - 3 + i, rows) * step + reflect101(c0 - 3 + j, cols)];
    }
    __syncthreads();
    // blur cell (i,j) <-> image (r0-2+i, c0-2+j): (S + 8) >> 4 with S the 16-weight sum
    for (int idx = tid; idx < (HT_H + 4) * (HT_W + 4); idx += 256) {
        const int i = idx / (HT_W + 4), j = idx - i * (HT_W + 4);
        const int s = raw[i][j] + 2 * raw[i][j + 1] + raw[i][j + 2] +
                      2 * (raw[i + 1][j] + 2 * raw[i + 1][j + 1] + raw[i + 1][j + 2]) +
                      raw[i + 2][j] + 2 * raw[i + 2][j + 1] + raw[i + 2][j + 2];
        blur[i][j] = (uint8_t)((s + 8) >> 4);
    }
    __syncthreads();
    // grad cell (i,j) <-> image (r0-1+i, c0-1+j)
    for (int idx = tid; idx < (HT_H + 2) * (HT_W + 2); idx += 256) {
        const int i = idx / (HT_W + 2), j = idx - i * (HT_W + 2);
        short2 g;
        g.x = (short)((int)blur[i + 1][j + 2] - (int)blur[i + 1][j]);
        g.y = (short)((int)blur[i + 2][j + 1] - (int)blur[i][j + 1]);
        grad[i][j] = g;
    }
    __syncthreads();
    const int tx = tid & (HT_W - 1);
    const int c = c0 + tx;
#pragma unroll
    for (int ty = tid / HT_W; ty < HT_H; ty += 256 / HT_W) {
        const int r = r0 + ty;
        if (r >= rows || c >= cols) continue;
        int sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int du = -1; du <= 1; ++du) {
            const int gi = clampi(r + du, 0, rows - 1) - (r0 - 1);
#pragma unroll
            for (int dv = -1; dv <= 1; ++dv) {
                const short2 g = grad[gi][clampi(c + dv, 0, cols - 1) - (c0 - 1)];
                sxx += g.x * g.x;
                syy += g.y * g.y;
                sxy += g.x * g.y;
            }
        }
        const float det = (float)((double)sxx * (double)syy - (double)sxy * (double)sxy);
        const float tr = (float)(sxx + syy);
        const float trtr = tr * tr;
        const float ktr = k * trtr;
        const float response = det - ktr;
        resp[blockIdx.z * rframe + (size_t)r * rstep_elems + c] = response > 0 ? response : 0.0f;
    }
}

}  // namespace vslam
