/*
 * vslam_oracle.c -- CPU restatement of the reference hot path (see vslam_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no OpenCV here, no golden vectors in
 * the reference).  Build with:  gcc -O2 -ffp-contract=off -fPIC -shared
 * (-ffp-contract=off: the reference is built without FMA).
 *
 * Citations are relative to /root/reference/KeyPointDetection/.
 */
#include "vslam_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* OpenCV primitive semantics                             */
/* ------------------------------------------------------------------------- */

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101). */
int vo_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cvRound(double): round half to even (SSE2 cvtsd2si under the default mode). */
static inline long cv_round(double v) { return lrint(v); }

int vo_gauss_ksize_u8(double sigma) { return (int)cv_round(sigma * 3 * 2 + 1) | 1; }

/* getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (OpenCV >= 4.5.1
 * smooth.dispatch.cpp), double arithmetic with libm exp instead of softfloat. */
static void gauss_kernel_f64(int n, double sigma, double* kf) {
    int n2 = (n - 1) / 2;
    if (sigma <= 0 && n == 1) {
        kf[0] = 1.0;
    } else if (sigma <= 0 && n == 3) {
        kf[0] = kf[2] = 0.25;
        kf[1] = 0.5;
    } else if (sigma <= 0 && n == 5) {
        kf[0] = kf[4] = 0.0625;
        kf[1] = kf[3] = 0.25;
        kf[2] = 0.375;
    } else if (sigma <= 0 && n == 7) {
        kf[0] = kf[6] = 0.03125;
        kf[1] = kf[5] = 0.109375;
        kf[2] = kf[4] = 0.21875;
        kf[3] = 0.28125;
    } else {
        double sigmaX = sigma > 0 ? sigma : fma((double)n, 0.15, 0.35);
        double scale2X = -0.125 / (sigmaX * sigmaX);
        double sum = 0.0;
        for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
            double t = exp((double)(x * x) * scale2X);
            kf[i] = t;
            sum += t;
        }
        sum *= 2.0;
        sum += 1.0;
        double mul1 = 1.0 / sum;
        for (int i = 0; i < n2; i++) {
            double t = kf[i] * mul1;
            kf[i] = t;
            kf[n - 1 - i] = t;
        }
        kf[n2] = 1.0 * mul1;
    }
}

int vo_gauss_ksize_f32(double sigma) { return (int)cv_round(sigma * 4 * 2 + 1) | 1; }

/* getGaussianKernel(n, sigma, CV_32F): the bit-exact f64 kernel narrowed to f32. */
int vo_gauss_kernel_f32(int n, double sigma, float* k) {
    if (n <= 0 || (n & 1) == 0 || n > (1 << 20) || !k) return -1;
    double* kf = (double*)malloc(sizeof(double) * (size_t)n);
    if (!kf) return -1;
    gauss_kernel_f64(n, sigma, kf);
    for (int i = 0; i < n; i++) k[i] = (float)kf[i];
    free(kf);
    return 0;
}

int vo_gauss_taps_q8(int n, double sigma, uint16_t* taps) {
    if (n <= 0 || (n & 1) == 0 || n > 4096) return -1;
    double* kf = (double*)malloc(sizeof(double) * (size_t)n);
    if (!kf) return -1;
    gauss_kernel_f64(n, sigma, kf);
    /* error diffusion from the outermost tap inwards; centre takes the remainder */
    double err = 0.0;
    long isum = 0;
    for (int i = 0; i < n / 2; i++) {
        double adj = kf[i] * 256.0 + err;
        long v0 = cv_round(adj);
        err = adj - (double)v0;
        taps[i] = (uint16_t)v0;
        taps[n - 1 - i] = (uint16_t)v0;
        isum += v0;
    }
    taps[n / 2] = (uint16_t)(256 - 2 * isum);
    free(kf);
    return 0;
}

int vo_gaussian_blur_u8(const uint8_t* src, int rows, int cols, size_t step, int ksize, double sigma,
                        uint8_t* dst, size_t dst_step) {
    if (!src || !dst || rows <= 0 || cols <= 0) return -1;
    int n = ksize > 0 ? ksize : (sigma > 0 ? vo_gauss_ksize_u8(sigma) : -1);
    if (n <= 0 || (n & 1) == 0) return -1;
    uint16_t* taps = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n);
    if (!taps || vo_gauss_taps_q8(n, sigma, taps) != 0) {
        free(taps);
        return -1;
    }
    int r = n / 2;
    /* horizontal pass: h = sum tap*pixel, exact (<= 255*256), SURVEY A2-iv */
    uint16_t* H = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)rows * (size_t)cols);
    uint8_t* prow = (uint8_t*)malloc((size_t)cols + 2 * (size_t)r);
    int* xi = (int*)malloc(sizeof(int) * ((size_t)cols + 2 * (size_t)r));
    uint32_t* acc = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cols);
    if (!H || !prow || !xi || !acc) {
        free(taps), free(H), free(prow), free(xi), free(acc);
        return -1;
    }
    for (int x = 0; x < cols + 2 * r; x++) xi[x] = vo_reflect101(x - r, cols);
    for (int y = 0; y < rows; y++) {
        const uint8_t* s = src + (size_t)y * step;
        for (int x = 0; x < cols + 2 * r; x++) prow[x] = s[xi[x]];
        memset(acc, 0, sizeof(uint32_t) * (size_t)cols);
        for (int k = 0; k < n; k++) {
            uint32_t t = taps[k];
            if (!t) continue;
            const uint8_t* p = prow + k;
            for (int x = 0; x < cols; x++) acc[x] += t * p[x];
        }
        uint16_t* h = H + (size_t)y * cols;
        for (int x = 0; x < cols; x++) h[x] = (uint16_t)acc[x];
    }
    /* vertical pass: 32-bit accumulate, one round-half-up at the end */
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) acc[x] = 32768u;
        for (int k = 0; k < n; k++) {
            uint32_t t = taps[k];
            if (!t) continue;
            const uint16_t* h = H + (size_t)vo_reflect101(y - r + k, rows) * cols;
            for (int x = 0; x < cols; x++) acc[x] += t * h[x];
        }
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < cols; x++) d[x] = (uint8_t)(acc[x] >> 16);
    }
    free(taps), free(H), free(prow), free(xi), free(acc);
    return 0;
}

int vo_sobel_k1_u8_f32(const uint8_t* src, int rows, int cols, size_t step, int dx, int dy, float* dst,
                       size_t dst_step_bytes) {
    if (!src || !dst || rows <= 0 || cols <= 0) return -1;
    if (!((dx == 1 && dy == 0) || (dx == 0 && dy == 1))) return -1;
    for (int r = 0; r < rows; r++) {
        float* d = (float*)((char*)dst + (size_t)r * dst_step_bytes);
        for (int c = 0; c < cols; c++) {
            int a, b;
            if (dx) {
                a = src[(size_t)r * step + vo_reflect101(c + 1, cols)];
                b = src[(size_t)r * step + vo_reflect101(c - 1, cols)];
            } else {
                a = src[(size_t)vo_reflect101(r + 1, rows) * step + c];
                b = src[(size_t)vo_reflect101(r - 1, rows) * step + c];
            }
            d[c] = (float)(a - b);
        }
    }
    return 0;
}

/* cv::resize(..., Size(), 2, 2, INTER_LINEAR) on CV_8UC1: HResizeLinear into int
 * with 11-bit coefficients, then VResizeLinear<uchar,int,short,FixedPtCast<..22>>. */
int vo_resize_linear2x_u8(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst,
                          size_t dst_step) {
    if (!src || !dst || rows <= 0 || cols <= 0) return -1;
    int dcols = cols * 2, drows = rows * 2;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dcols);
    short* alpha = (short*)malloc(sizeof(short) * 2 * (size_t)dcols);
    int* h0 = (int*)malloc(sizeof(int) * (size_t)dcols);
    int* h1 = (int*)malloc(sizeof(int) * (size_t)dcols);
    if (!xofs || !alpha || !h0 || !h1) {
        free(xofs), free(alpha), free(h0), free(h1);
        return -1;
    }
    for (int dx = 0; dx < dcols; dx++) {
        float fx = (float)((dx + 0.5) * 0.5 - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) fx = 0.f, sx = 0;
        if (sx >= cols - 1) fx = 0.f, sx = cols - 1;
        xofs[dx] = sx;
        alpha[2 * dx] = (short)lrintf((1.f - fx) * 2048.f);
        alpha[2 * dx + 1] = (short)lrintf(fx * 2048.f);
    }
    for (int dy = 0; dy < drows; dy++) {
        float fy = (float)((dy + 0.5) * 0.5 - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        int b0 = (short)lrintf((1.f - fy) * 2048.f);
        int b1 = (short)lrintf(fy * 2048.f);
        int r0 = clampi(sy, 0, rows - 1), r1 = clampi(sy + 1, 0, rows - 1);
        const uint8_t* s0 = src + (size_t)r0 * step;
        const uint8_t* s1 = src + (size_t)r1 * step;
        for (int dx = 0; dx < dcols; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < cols ? sx + 1 : sx; /* alpha1 == 0 there */
            h0[dx] = s0[sx] * alpha[2 * dx] + s0[sx1] * alpha[2 * dx + 1];
            h1[dx] = s1[sx] * alpha[2 * dx] + s1[sx1] * alpha[2 * dx + 1];
        }
        uint8_t* d = dst + (size_t)dy * dst_step;
        for (int dx = 0; dx < dcols; dx++) {
            int v = (((b0 * (h0[dx] >> 4)) >> 16) + ((b1 * (h1[dx] >> 4)) >> 16) + 2) >> 2;
            d[dx] = (uint8_t)clampi(v, 0, 255);
        }
    }
    free(xofs), free(alpha), free(h0), free(h1);
    return 0;
}

void vo_half_size(int rows, int cols, int* out_rows, int* out_cols) {
    *out_rows = (int)cv_round(rows * 0.5);
    *out_cols = (int)cv_round(cols * 0.5);
}

int vo_resize_nearest_half_u8(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst,
                              size_t dst_step) {
    int dr, dc;
    vo_half_size(rows, cols, &dr, &dc);
    if (!src || !dst || dr <= 0 || dc <= 0) return -1;
    for (int y = 0; y < dr; y++) {
        int sy = 2 * y < rows - 1 ? 2 * y : rows - 1;
        for (int x = 0; x < dc; x++) {
            int sx = 2 * x < cols - 1 ? 2 * x : cols - 1;
            dst[(size_t)y * dst_step + x] = src[(size_t)sy * step + sx];
        }
    }
    return 0;
}

/* saturate_cast<uchar>(cvRound(|x|)) as the x86-64 OpenCV build the reference runs on evaluates it
 * (Harris_corners.cpp:176,181; README.md:29-31: Ubuntu 22.04 / GCC 11): cvRound is cvtss2si
 * (scalar tail) / cvtps2dq (v_round in cvtabs_32f), round half even, and returns INT_MIN -- the
 * "integer indefinite" value -- for NaN and for |x| >= 2^31; saturate_cast<uchar>(INT_MIN), like
 * v_pack + v_pack_u_store of it, is 0.  So responses in [255.5, 2^31) read 255 in the 8-bit view and
 * responses >= 2^31 read 0: literal reference behaviour (an AArch64 build would saturate). */
static inline uint8_t cvt_abs_u8(float x) {
    float a = fabsf(x);
    if (!(a < 2147483648.0f)) return 0; /* INT_MIN -> 0; NaN -> 0 */
    if (a >= 255.5f) return 255;
    return (uint8_t)lrintf(a);
}

int vo_convert_scale_abs_f32(const float* src, int rows, int cols, size_t step_bytes, uint8_t* dst,
                             size_t dst_step) {
    if (!src || !dst || rows <= 0 || cols <= 0) return -1;
    for (int r = 0; r < rows; r++) {
        const float* s = (const float*)((const char*)src + (size_t)r * step_bytes);
        for (int c = 0; c < cols; c++) dst[(size_t)r * dst_step + c] = cvt_abs_u8(s[c]);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Harris path                                                                */
/* ------------------------------------------------------------------------- */

int vo_harris_from_grad_f32(const float* ix, const float* iy, int rows, int cols, size_t step_bytes, float k,
                            int window, float* resp, size_t resp_step_bytes) {
    if (!ix || !iy || !resp || rows <= 0 || cols <= 0 || window < 1 || (window & 1) == 0) return -1;
    int pad = (window - 1) / 2; /* Harris_corners.cpp:35 */
    for (int r = 0; r < rows; r++) {
        float* out = (float*)((char*)resp + (size_t)r * resp_step_bytes);
        for (int c = 0; c < cols; c++) {
            /* StructureMatrix, Harris_corners.cpp:10-29, on BORDER_REPLICATE-padded
             * gradients (:42-43): clamp addressing */
            float Ix2 = 0.0f, Iy2 = 0.0f, IxIy = 0.0f;
            for (int u = r - pad; u <= r + pad; u++) {
                int uu = clampi(u, 0, rows - 1);
                const float* px = (const float*)((const char*)ix + (size_t)uu * step_bytes);
                const float* py = (const float*)((const char*)iy + (size_t)uu * step_bytes);
                for (int v = c - pad; v <= c + pad; v++) {
                    int vv = clampi(v, 0, cols - 1);
                    Ix2 += px[vv] * px[vv];
                    Iy2 += py[vv] * py[vv];
                    IxIy += px[vv] * py[vv];
                }
            }
            /* cv::determinant 2x2 CV_32F: double products (A6); cv::trace: double sum */
            float det = (float)((double)Ix2 * (double)Iy2 - (double)IxIy * (double)IxIy);
            float tr = (float)((double)Ix2 + (double)Iy2);
            float trtr = tr * tr;
            float ktr = k * trtr;
            float response = det - ktr; /* Harris_corners.cpp:57 */
            out[c] = response > 0 ? response : 0.0f; /* :60-62 over a zeroed Mat (:40) */
        }
    }
    return 0;
}

int vo_harris_response_u8(const uint8_t* img, int rows, int cols, size_t step, float k, int window,
                          float* resp, size_t resp_step_bytes) {
    if (!img || !resp || rows <= 0 || cols <= 0) return -1;
    size_t n = (size_t)rows * cols;
    uint8_t* blurred = (uint8_t*)malloc(n);
    float* gx = (float*)malloc(n * sizeof(float));
    float* gy = (float*)malloc(n * sizeof(float));
    int rc = -1;
    if (blurred && gx && gy) {
        rc = vo_gaussian_blur_u8(img, rows, cols, step, 3, 0.0, blurred, (size_t)cols); /* :158 */
        if (!rc) rc = vo_sobel_k1_u8_f32(blurred, rows, cols, (size_t)cols, 1, 0, gx, (size_t)cols * 4);
        if (!rc) rc = vo_sobel_k1_u8_f32(blurred, rows, cols, (size_t)cols, 0, 1, gy, (size_t)cols * 4);
        if (!rc)
            rc = vo_harris_from_grad_f32(gx, gy, rows, cols, (size_t)cols * 4, k, window, resp,
                                         resp_step_bytes); /* :172 */
    }
    free(blurred), free(gx), free(gy);
    return rc;
}

int vo_nms_strict_u8(const uint8_t* src, int rows, int cols, size_t step, int window, uint8_t* mask,
                     size_t mask_step) {
    if (!src || !mask || rows <= 0 || cols <= 0 || window < 1 || (window & 1) == 0) return -1;
    int p = (window - 1) / 2;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            int m = 0; /* dilate border value for CV_8U: never wins (A8) */
            for (int u = r - p; u <= r + p; u++)
                for (int v = c - p; v <= c + p; v++) {
                    if ((u == r && v == c) || u < 0 || u >= rows || v < 0 || v >= cols) continue;
                    int t = src[(size_t)u * step + v];
                    if (t > m) m = t;
                }
            mask[(size_t)r * mask_step + c] = src[(size_t)r * step + c] > m ? 255 : 0;
        }
    return 0;
}

int vo_nms_strict_f32(const float* src, int rows, int cols, size_t step_bytes, int window, uint8_t* mask,
                      size_t mask_step) {
    if (!src || !mask || rows <= 0 || cols <= 0 || window < 1 || (window & 1) == 0) return -1;
    int p = (window - 1) / 2;
#define AT(r, c) (((const float*)((const char*)src + (size_t)(r) * step_bytes))[c])
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            float m = -FLT_MAX;
            for (int u = r - p; u <= r + p; u++)
                for (int v = c - p; v <= c + p; v++) {
                    if ((u == r && v == c) || u < 0 || u >= rows || v < 0 || v >= cols) continue;
                    float t = AT(u, v);
                    if (t > m) m = t;
                }
            mask[(size_t)r * mask_step + c] = AT(r, c) > m ? 255 : 0;
        }
#undef AT
    return 0;
}

int vo_nms2_f32(const float* resp, int rows, int cols, size_t step_bytes, int window, float* out,
                size_t out_step_bytes, float* true_max_out) {
    if (!resp || !out || rows <= 0 || cols <= 0 || window < 1) return -1;
    int padding = (window - 1) / 2; /* Harris_corners.cpp:91 */
    float true_max = 0;
#define AT(r, c) (((const float*)((const char*)resp + (size_t)(r) * step_bytes))[c])
    for (int r = 0; r < rows; r++) memset((char*)out + (size_t)r * out_step_bytes, 0, sizeof(float) * (size_t)cols);
    for (int i = padding; i < rows - padding; i++) {
        for (int j = padding; j < cols - padding; j++) {
            float max = 0;
            for (int u = i - padding; u < i + padding; u++) {     /* half-open, :100 */
                for (int v = j - padding; v < j + padding; v++) { /* :101 */
                    if (AT(u, v) > max) {
                        max = AT(u, v);
                        if (max > true_max) true_max = max;
                    }
                }
            }
            if (AT(i, j) >= max) /* :116 */
                ((float*)((char*)out + (size_t)i * out_step_bytes))[j] = max;
        }
    }
#undef AT
    if (true_max_out) *true_max_out = true_max;
    return 0;
}

size_t vo_harris_keypoints(const float* nms2, int rows, int cols, size_t step_bytes, vo_kp* out, size_t cap) {
    size_t n = 0;
    for (int r = 0; r < rows; r++) {
        const float* s = (const float*)((const char*)nms2 + (size_t)r * step_bytes);
        for (int c = 0; c < cols; c++) {
            if (cvt_abs_u8(s[c]) > 253) { /* Harris_corners.cpp:139 on the :181 view */
                if (out && n < cap) {
                    out[n].row = r;
                    out[n].col = c;
                    out[n].response = s[c];
                }
                n++;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* DoG pyramid path                                                           */
/* ------------------------------------------------------------------------- */

int vo_auto_num_octaves(int rows, int cols) {
    int m = rows < cols ? rows : cols;
    return (int)floor(log2((double)m)) - 4; /* GaussPyramid.cpp:151 */
}

double vo_sigma(double sigma0, int octave, int level) {
    double k_ = pow(2.0f, 1.0f / (double)3); /* GaussPyramid.hpp:69, scaleSamples_ = 3 */
    return pow(2, octave) * sigma0 * pow(k_, level); /* GaussPyramid.cpp:157,161 */
}

void vo_pyramid_free(vo_pyramid* p) {
    if (!p) return;
    for (int o = 0; o < VO_MAX_OCTAVES; o++) {
        free(p->base[o]);
        for (int l = 0; l < VO_NUM_LEVELS; l++) free(p->gauss[o][l]);
        for (int l = 0; l < VO_NUM_DOGS; l++) free(p->dog[o][l]);
    }
    free(p);
}

vo_pyramid* vo_pyramid_build_u8(const uint8_t* img, int rows, int cols, size_t step, int n_octaves,
                                double sigma0) {
    if (!img || rows <= 0 || cols <= 0 || n_octaves < 1 || n_octaves > VO_MAX_OCTAVES || !(sigma0 > 0))
        return NULL;
    vo_pyramid* p = (vo_pyramid*)calloc(1, sizeof(vo_pyramid));
    if (!p) return NULL;
    p->n_octaves = n_octaves;
    p->sigma0 = sigma0;
    int r = rows * 2, c = cols * 2;
    uint8_t* base = (uint8_t*)malloc((size_t)r * c);
    if (!base || vo_resize_linear2x_u8(img, rows, cols, step, base, (size_t)c) != 0) { /* GaussPyramid.cpp:110 */
        free(base);
        vo_pyramid_free(p);
        return NULL;
    }
    for (int o = 0; o < n_octaves; o++) {
        size_t n = (size_t)r * c;
        p->rows[o] = r;
        p->cols[o] = c;
        p->base[o] = base; /* :119 */
        for (int l = 0; l < VO_NUM_LEVELS; l++) { /* GaussVector, :166-185: each level from the base */
            double s = vo_sigma(sigma0, o, l);
            p->sigma[o][l] = s;
            p->ksize[o][l] = vo_gauss_ksize_u8(s);
            p->gauss[o][l] = (uint8_t*)malloc(n);
            if (!p->gauss[o][l] ||
                vo_gaussian_blur_u8(base, r, c, (size_t)c, 0, s, p->gauss[o][l], (size_t)c) != 0) {
                vo_pyramid_free(p);
                return NULL;
            }
        }
        for (int l = 0; l < VO_NUM_DOGS; l++) { /* Diff_of_Gauss, :191-200: saturating u8 subtract */
            uint8_t* d = (uint8_t*)malloc(n);
            if (!d) {
                vo_pyramid_free(p);
                return NULL;
            }
            const uint8_t* a = p->gauss[o][l + 1];
            const uint8_t* b = p->gauss[o][l];
            for (size_t i = 0; i < n; i++) d[i] = a[i] > b[i] ? (uint8_t)(a[i] - b[i]) : 0;
            p->dog[o][l] = d;
        }
        base = NULL;
        if (o + 1 < n_octaves) { /* :123-126 */
            int nr, nc;
            vo_half_size(r, c, &nr, &nc);
            if (nr <= 0 || nc <= 0) {
                vo_pyramid_free(p);
                return NULL;
            }
            base = (uint8_t*)malloc((size_t)nr * nc);
            if (!base || vo_resize_nearest_half_u8(p->gauss[o][3], r, c, (size_t)c, base, (size_t)nc) != 0) {
                free(base);
                vo_pyramid_free(p);
                return NULL;
            }
            r = nr;
            c = nc;
        }
    }
    return p;
}

/* ---- the one UNPINNED arithmetic choice of the f32 stages: contraction of a*b+c -------------------------------------
 * The reference is compiled without -march flags, but OpenCV itself dispatches at run time: on an x86-64 with
 * AVX2 + FMA3 its f32 row / column filters (filter.simd.hpp: RowVec_32f / SymmColumnVec_32f use v_muladd = v_fma) and
 * hal::fastAtan32f (mathfuncs_core.simd.hpp: v_fma(v_fma(v_fma(cc, p7, p5), cc, p3), cc, p1) * c) run variants whose
 * v_fma IS one fused instruction, where the SSE2 baseline rounds the product and the sum separately.  Which one a given
 * OpenCV build runs cannot be known here (no OpenCV), so the oracle carries BOTH: variant 0 (default, what the GPU kernels
 * compute: every op rounded) and the fused form in exactly these three places and nowhere else - a mask, because the two
 * OpenCV modules dispatch independently: bit 0 = the arctangent polynomial, bit 1 = the filter's two passes (3 = both).
 * tools/fma_risk_report.py runs rows (f)1 / (f)3 / (f)4 under both and counts what changes; tools/pin_with_opencv.sh
 * tries both against a real OpenCV.  Test infrastructure only, like the rest of this file. */
static int g_fma_variant = 0; /* bit 0: fastAtan32f's polynomial, bit 1: the separable f32 filter's row and column passes */
void vo_set_fma_variant(int mask) { g_fma_variant = mask & 3; }
int vo_get_fma_variant(void) { return g_fma_variant; }

/* cv::fastAtan2 / hal::fastAtan32f (OpenCV 4.x mathfuncs_core.simd.hpp): scalar form, or with the polynomial's three
 * multiply-adds fused (variant 1: the AVX2 / FMA3 dispatch of the vector loop). */
float vo_fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)DBL_EPSILON;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (g_fma_variant & 1) ? fmaf(fmaf(fmaf(c2, p7, p5), c2, p3), c2, p1) * c : (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - ((g_fma_variant & 1) ? fmaf(fmaf(fmaf(c2, p7, p5), c2, p3), c2, p1) * c : (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c);
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

int vo_level_gradients(const uint8_t* g, int rows, int cols, size_t step, float* gx, float* gy, float* mag,
                       float* orient, size_t out_step_bytes) {
    if (!g || rows <= 0 || cols <= 0) return -1;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            const float x = (float)((int)g[(size_t)r * step + vo_reflect101(c + 1, cols)] -
                                    (int)g[(size_t)r * step + vo_reflect101(c - 1, cols)]);
            const float y = (float)((int)g[(size_t)vo_reflect101(r + 1, rows) * step + c] -
                                    (int)g[(size_t)vo_reflect101(r - 1, rows) * step + c]);
            const size_t o = (size_t)r * out_step_bytes / sizeof(float) + c;
            if (gx) gx[o] = x;
            if (gy) gy[o] = y;
            if (mag) {
                const float xx = x * x, yy = y * y;
                mag[o] = sqrtf(xx + yy);
            }
            if (orient) orient[o] = vo_fast_atan2_deg(y, x);
        }
    return 0;
}

void vo_extrema_lattice(int rows, int cols, int window, int* lat_rows, int* lat_cols) {
    int pad = (window - 1) / 2;
    *lat_rows = rows > pad ? (rows - pad + window - 1) / window : 0;
    *lat_cols = cols > pad ? (cols - pad + window - 1) / window : 0;
}

/* (int)float the way the reference's x86-64 build performs it (cvttss2si): out-of-range and NaN
 * inputs give INT_MIN ("integer indefinite"); the C standard leaves them undefined. */
static int cvtt_f32_i32(float x) {
    if (!(x > -2147483904.0f && x < 2147483648.0f)) return INT32_MIN;
    return (int)x;
}

/* FeaturePointLocalization, Diff_of_Gauss.cpp:223-251, on the three finite differences it takes
 * at :226-228 and the candidate's own DoG value.  OpenCV arithmetic as recalled (SURVEY
 * Appendix B-6, all of it unverifiable here):
 *   :233  A = (d_x, d_y, d_scale) / 255.0f, three f32 divisions
 *   :238  B = A * A^T is an outer product: gemm's n == 1 branch rounds each exact product once
 *   :239  Mat::inv() = cv::invert(DECOMP_LU), closed form for 3x3 CV_32F: determinant by cofactor
 *         expansion in f64 on the f32 entries; exactly 0 -> all-zero result, otherwise each f64
 *         cofactor * (1/det) narrowed to f32.  B has rank 1, so whenever one difference is 0 the
 *         determinant is exactly 0; with three non-zero differences it is rounding noise and the
 *         "inverse" is large and arbitrary - the reference's behaviour, reproduced as is.
 *   :239  unary minus: exact
 *   :240  z_hat = B_inverse * A takes gemm's small-matrix path (len 3 == rows of the result):
 *         f32 products summed left to right in f32
 *   :241  0.5f * A_T * z_hat is gemm with alpha 0.5: products and sum in f64, * 0.5 in f64,
 *         narrowed to f32; then the f32 addition of value / 255.0f
 *   :245  keep iff dog_zhat > 0.03f; :246 value = (int)(dog_zhat * 255.0f)
 * Returns 1 and stores the new value if the point is kept, else 0. */
int vo_feature_point_localization(int d_x, int d_y, int d_scale, int value, int* new_value) {
    const float a[3] = {(float)d_x / 255.0f, (float)d_y / 255.0f, (float)d_scale / 255.0f};
    float B[3][3], Bi[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) B[i][j] = (float)((double)a[i] * (double)a[j]);
#define M(i, j) ((double)B[i][j])
    double d = M(0, 0) * (M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) - M(0, 1) * (M(1, 0) * M(2, 2) - M(1, 2) * M(2, 0)) +
               M(0, 2) * (M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0));
    if (d != 0.) {
        d = 1. / d;
        Bi[0][0] = (float)((M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) * d);
        Bi[0][1] = (float)((M(0, 2) * M(2, 1) - M(0, 1) * M(2, 2)) * d);
        Bi[0][2] = (float)((M(0, 1) * M(1, 2) - M(0, 2) * M(1, 1)) * d);
        Bi[1][0] = (float)((M(1, 2) * M(2, 0) - M(1, 0) * M(2, 2)) * d);
        Bi[1][1] = (float)((M(0, 0) * M(2, 2) - M(0, 2) * M(2, 0)) * d);
        Bi[1][2] = (float)((M(0, 2) * M(1, 0) - M(0, 0) * M(1, 2)) * d);
        Bi[2][0] = (float)((M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0)) * d);
        Bi[2][1] = (float)((M(0, 1) * M(2, 0) - M(0, 0) * M(2, 1)) * d);
        Bi[2][2] = (float)((M(0, 0) * M(1, 1) - M(0, 1) * M(1, 0)) * d);
    } else {
        memset(Bi, 0, sizeof Bi);
    }
#undef M
    float z[3];
    for (int i = 0; i < 3; i++) {
        float t = (-Bi[i][0]) * a[0];
        t = t + (-Bi[i][1]) * a[1];
        t = t + (-Bi[i][2]) * a[2];
        z[i] = t;
    }
    double s = 0.;
    for (int k = 0; k < 3; k++) s += (double)z[k] * (double)a[k];
    const float half_quad = (float)(s * 0.5);
    const float dog_zhat = (float)value / 255.0f + half_quad;
    if (dog_zhat > 0.03f) {
        if (new_value) *new_value = cvtt_f32_i32(dog_zhat * 255.0f);
        return 1;
    }
    return 0;
}

/* computeEdgeResponse, Diff_of_Gauss.cpp:79-109: f32 sums of gradient products over the
 * half-open window [y-p, y+p) x [x-p, x+p) of the UNPADDED gradient images addressed with the
 * keypoint's PADDED coordinates (:93-94; Appendix B), cv::determinant of the 2x2 CV_32F matrix
 * (f64 products, :105), cv::trace (f64 sum, :106), tr*tr/det in f32 (:107).  With the
 * reference's windowSize 3 every access is inside the image; for larger paddings the reference
 * reads out of bounds, which is clamped here. */
float vo_compute_edge_response(const float* gx, const float* gy, int rows, int cols, size_t step_elems, int row,
                               int col, int padding) {
    float Ix2 = 0, Iy2 = 0, IxIy = 0;
    for (int u = row - padding; u < row + padding; u++)
        for (int v = col - padding; v < col + padding; v++) {
            const size_t o = (size_t)clampi(u, 0, rows - 1) * step_elems + clampi(v, 0, cols - 1);
            Ix2 += gx[o] * gx[o];
            Iy2 += gy[o] * gy[o];
            IxIy += gx[o] * gy[o];
        }
    const float det = (float)((double)Ix2 * (double)Iy2 - (double)IxIy * (double)IxIy);
    const float tr = (float)(0.0 + (double)Ix2 + (double)Iy2);
    return (tr * tr) / det;
}

/* GaussianBlur(roi, dst, Size(0,0), sigma, 0, BORDER_DEFAULT) on a CV_32F window that is a ROI
 * of a larger Mat (Diff_of_Gauss.cpp:341-348).  BORDER_ISOLATED is not set, so pixels outside the
 * window come from the parent and reflect-101 applies at the PARENT's edges.  Separable f32
 * filter in OpenCV's order: row filter s = k[0]*S[0]; s += k[i]*S[i] left to right; symmetric
 * column filter s = k[c]*S[0]; s += k[c+i]*(S[+i] + S[-i]) (mul and add rounded separately: SSE
 * baseline; variant 1 - vo_set_fma_variant - fuses each multiply-add as the AVX2/FMA3 dispatch does). */
static int blur_f32_roi(const float* parent, int prows, int pcols, int x0, int y0, int w, int h, const float* k,
                        int n, float* dst) {
    const int R = n / 2;
    float* rb = (float*)malloc(sizeof(float) * (size_t)(h + 2 * R) * (size_t)w);
    int* cx = (int*)malloc(sizeof(int) * (size_t)(w + 2 * R));
    if (!rb || !cx) {
        free(rb);
        free(cx);
        return -1;
    }
    for (int i = 0; i < w + 2 * R; i++) cx[i] = vo_reflect101(x0 + i - R, pcols);
    for (int rr = 0; rr < h + 2 * R; rr++) {
        const float* S = parent + (size_t)vo_reflect101(y0 + rr - R, prows) * pcols;
        for (int c = 0; c < w; c++) {
            float s0 = k[0] * S[cx[c]];
            if (g_fma_variant & 2)
                for (int i = 1; i < n; i++) s0 = fmaf(S[cx[c + i]], k[i], s0);
            else
                for (int i = 1; i < n; i++) s0 += k[i] * S[cx[c + i]];
            rb[(size_t)rr * w + c] = s0;
        }
    }
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            float s0 = k[R] * rb[(size_t)(r + R) * w + c];
            if (g_fma_variant & 2)
                for (int i = 1; i <= R; i++) s0 = fmaf(rb[(size_t)(r + R + i) * w + c] + rb[(size_t)(r + R - i) * w + c], k[R + i], s0);
            else
                for (int i = 1; i <= R; i++) s0 += k[R + i] * (rb[(size_t)(r + R + i) * w + c] + rb[(size_t)(r + R - i) * w + c]);
            dst[(size_t)r * w + c] = s0;
        }
    free(rb);
    free(cx);
    return 0;
}

/* The same blur with the kernel formed from sigma (getGaussianKernel(cvRound(8 sigma + 1) | 1, sigma, CV_32F)): the entry
 * point tests/test_opencv_crosscheck.py compares with a real cv::GaussianBlur on a ROI (tools/opencv_pin/pin_harness.cpp). */
int vo_blur_f32_roi(const float* parent, int prows, int pcols, int x0, int y0, int w, int h, double sigma, float* dst) {
    if (!parent || !dst || prows <= 0 || pcols <= 0 || w <= 0 || h <= 0 || x0 < 0 || y0 < 0 || x0 + w > pcols || y0 + h > prows || !(sigma > 0)) return -1;
    const int n = vo_gauss_ksize_f32(sigma);
    float* k = (float*)malloc(sizeof(float) * (size_t)n);
    if (!k) return -1;
    int rc = vo_gauss_kernel_f32(n, sigma, k);
    if (rc == 0) rc = blur_f32_roi(parent, prows, pcols, x0, y0, w, h, k, n, dst);
    free(k);
    return rc;
}

static float* pad_replicate_f32(const float* img, int rows, int cols, int pad) {
    const int pr = rows + 2 * pad, pc = cols + 2 * pad;
    float* o = (float*)malloc(sizeof(float) * (size_t)pr * pc);
    if (!o) return NULL;
    for (int r = 0; r < pr; r++)
        for (int c = 0; c < pc; c++)
            o[(size_t)r * pc + c] = img[(size_t)clampi(r - pad, 0, rows - 1) * cols + clampi(c - pad, 0, cols - 1)];
    return o;
}

/* filterKeypoints, Diff_of_Gauss.cpp:301-372 (+ orientationHistogram :112-133): edge rejection
 * tr^2/det < 12.1 on the level's Sobel gradients, then the 36-bin orientation histogram of the
 * 16x16 window Rect(x, y, 16, 16) of the 8-padded magnitude / orientation images, magnitudes
 * weighted by GaussianBlur(sigma = 1.5 * sigma(octave, level)); every bin above 0.8 * max
 * appends SLAM::point{y, x, bin*10, 0, octave, level}.  Returns the total (writes <= cap), or
 * (size_t)-1 for a keypoint the reference could not process (level / window outside the data). */
size_t vo_filter_keypoints(const vo_pyramid* p, int octave, const vo_point* kps, size_t n, vo_point* out, size_t cap) {
    if (!p || octave < 0 || octave >= p->n_octaves || (!kps && n)) return (size_t)-1;
    const int rows = p->rows[octave], cols = p->cols[octave];
    const int windowSize = 16, padding = windowSize / 2; /* :304-305 */
    const size_t P = (size_t)rows * cols;
    float *gx[VO_NUM_LEVELS] = {0}, *gy[VO_NUM_LEVELS] = {0}, *pmag[VO_NUM_LEVELS] = {0}, *porient[VO_NUM_LEVELS] = {0};
    float* kern[VO_NUM_LEVELS] = {0};
    int kn[VO_NUM_LEVELS] = {0};
    size_t total = 0;
    int bad = 0;
    const float r = 10.0f, threshold = ((r + 1.0f) * (r + 1.0f)) / r; /* :331-332 */
    const int size = 36;                                              /* :352 */
    const float reductionCoeff = (float)size / 360.0f;                /* :114 */
    for (size_t q = 0; q < n && !bad; q++) {
        const int x = kps[q].col, y = kps[q].row, level = kps[q].level;
        if (level < 0 || level >= VO_NUM_LEVELS || x < 0 || y < 0 || x > cols || y > rows || kps[q].octave != octave) {
            bad = 1; /* vector::at / Rect outside the padded Mat: the reference throws */
            break;
        }
        if (!gx[level]) { /* processGradients for this level (GaussPyramid.cpp:65-104), padOctave (:322-323) */
            gx[level] = (float*)malloc(4 * P);
            gy[level] = (float*)malloc(4 * P);
            float* mag = (float*)malloc(4 * P);
            float* ori = (float*)malloc(4 * P);
            if (!gx[level] || !gy[level] || !mag || !ori) abort();
            vo_level_gradients(p->gauss[octave][level], rows, cols, (size_t)cols, gx[level], gy[level], mag, ori, 4 * (size_t)cols);
            pmag[level] = pad_replicate_f32(mag, rows, cols, padding);
            porient[level] = pad_replicate_f32(ori, rows, cols, padding);
            free(mag);
            free(ori);
            const double sigma = 1.5 * p->sigma[octave][level]; /* :346 */
            kn[level] = vo_gauss_ksize_f32(sigma);
            kern[level] = (float*)malloc(4 * (size_t)kn[level]);
            if (!pmag[level] || !porient[level] || !kern[level]) abort();
            vo_gauss_kernel_f32(kn[level], sigma, kern[level]);
        }
        const float response = vo_compute_edge_response(gx[level], gy[level], rows, cols, (size_t)cols, y, x, kps[q].padding);
        if (!(response < threshold)) continue; /* :335 */
        float magWeighted[16 * 16];
        if (blur_f32_roi(pmag[level], rows + 2 * padding, cols + 2 * padding, x, y, windowSize, windowSize, kern[level], kn[level],
                         magWeighted) != 0)
            abort();
        float histo[36];
        for (int b = 0; b < size; b++) histo[b] = 0.0f;
        const int pc = cols + 2 * padding;
        for (int i = 0; i < windowSize && !bad; i++)
            for (int j = 0; j < windowSize; j++) {
                const float orientation = porient[level][(size_t)(y + i) * pc + (x + j)];
                const int index = (int)(orientation * reductionCoeff); /* :126 */
                if (index < 0 || index >= size) {
                    bad = 1; /* histo.at(index) throws */
                    break;
                }
                histo[index] += magWeighted[i * windowSize + j];
            }
        if (bad) break;
        float maxPeak = histo[0];
        for (int b = 1; b < size; b++)
            if (histo[b] > maxPeak) maxPeak = histo[b];
        const float peakThreshold = maxPeak * 0.8f; /* :358 */
        for (int b = 0; b < size; b++)
            if (histo[b] > peakThreshold) { /* :362 */
                if (out && total < cap) {
                    out[total].row = y;
                    out[total].col = x;
                    out[total].value = b * 10;
                    out[total].padding = 0;
                    out[total].octave = kps[q].octave;
                    out[total].level = level;
                }
                total++;
            }
    }
    for (int l = 0; l < VO_NUM_LEVELS; l++) {
        free(gx[l]);
        free(gy[l]);
        free(pmag[l]);
        free(porient[l]);
        free(kern[l]);
    }
    return bad ? (size_t)-1 : total;
}

/* initialKeypointDetection as the reference runs it, Diff_of_Gauss.cpp:254-297 including the
 * FeaturePointLocalization call at :290: the keypoints vector it appends to. */
size_t vo_dog_keypoints(const vo_pyramid* p, int octave, int window, vo_point* out, size_t cap) {
    if (!p || octave < 0 || octave >= p->n_octaves || window < 3 || (window & 1) == 0) return 0;
    int rows = p->rows[octave], cols = p->cols[octave];
    int padding = (window - 1) / 2;
    size_t n = 0;
#define DP(l, u, v) \
    ((int)p->dog[octave][l][(size_t)clampi((u) - padding, 0, rows - 1) * cols + clampi((v) - padding, 0, cols - 1)])
    for (int level = 1; level < VO_NUM_DOGS - 1; level++)
        for (int i = padding; i < rows; i += window)
            for (int j = padding; j < cols; j += window) {
                int this_pixel = DP(level, i, j);
                int mn = 256, mx = -1;
                for (int u = i - padding; u < i + padding; u++)
                    for (int v = j - padding; v < j + padding; v++)
                        for (int l = level - 1; l <= level + 1; l++) {
                            int t = DP(l, u, v);
                            if (t < mn) mn = t;
                            if (t > mx) mx = t;
                        }
                if (this_pixel != mn && this_pixel != mx) continue;
                int d_x = DP(level, i, j - 1) - DP(level, i, j + 1);         /* :226 */
                int d_y = DP(level, i - 1, j) - DP(level, i + 1, j);         /* :227 */
                int d_scale = DP(level - 1, i, j) - DP(level + 1, i, j);     /* :228 */
                int nv;
                if (!vo_feature_point_localization(d_x, d_y, d_scale, this_pixel, &nv)) continue;
                if (out && n < cap) {
                    out[n].row = i;
                    out[n].col = j;
                    out[n].value = nv; /* :246 */
                    out[n].padding = padding;
                    out[n].octave = octave;
                    out[n].level = level;
                }
                n++;
            }
#undef DP
    return n;
}

size_t vo_dog_extrema(const vo_pyramid* p, int octave, int window, int min_contrast, uint8_t* mask,
                      vo_point* out, size_t cap) {
    if (!p || octave < 0 || octave >= p->n_octaves || window < 3 || (window & 1) == 0) return 0;
    int rows = p->rows[octave], cols = p->cols[octave];
    int padding = (window - 1) / 2; /* Diff_of_Gauss.cpp:259 */
    int lr, lc;
    vo_extrema_lattice(rows, cols, window, &lr, &lc);
    size_t n = 0;
    /* padOctave (:260, GaussPyramid.cpp:133-141) = BORDER_REPLICATE: clamp addressing */
#define DP(l, u, v) \
    ((int)p->dog[octave][l][(size_t)clampi((u) - padding, 0, rows - 1) * cols + clampi((v) - padding, 0, cols - 1)])
    for (int level = 1; level < VO_NUM_DOGS - 1; level++) { /* :264 */
        int li = 0;
        for (int i = padding; i < rows; i += window, li++) { /* :267 */
            int lj = 0;
            for (int j = padding; j < cols; j += window, lj++) { /* :268 */
                int this_pixel = DP(level, i, j);
                int mn = 256, mx = -1;
                for (int u = i - padding; u < i + padding; u++)     /* :273 */
                    for (int v = j - padding; v < j + padding; v++) /* :274 */
                        for (int l = level - 1; l <= level + 1; l++) {
                            int t = DP(l, u, v);
                            if (t < mn) mn = t;
                            if (t > mx) mx = t;
                        }
                int cand = (this_pixel == mn || this_pixel == mx); /* :287 */
                if (mask) mask[((size_t)(level - 1) * lr + li) * lc + lj] = (uint8_t)cand;
                if (cand && this_pixel >= min_contrast) {
                    if (out && n < cap) { /* SLAM::point(i, j, value, padding, octave, level), :289 */
                        out[n].row = i;
                        out[n].col = j;
                        out[n].value = this_pixel;
                        out[n].padding = padding;
                        out[n].octave = octave;
                        out[n].level = level;
                    }
                    n++;
                }
            }
        }
    }
#undef DP
    return n;
}

/* Extension, NOT a restatement of reference code (note under the table): the
 * dense 3x3x3 scale-space test the north star's wording names.  Same rule as :282-287 (a site is a
 * candidate iff its value equals the minimum or the maximum of its window, ties included), same
 * replicate border as padOctave (:260), but on EVERY pixel of levels 1..3 with the full 3x3x3
 * neighbourhood instead of the half-open 2x2x3 window on the stride-3 lattice.  Points carry padded
 * coordinates (row+1, col+1, padding 1) like :289, in (level, row, col) order. */
size_t vo_dog_extrema_dense(const vo_pyramid* p, int octave, int min_contrast, uint8_t* mask, vo_point* out, size_t cap) {
    if (!p || octave < 0 || octave >= p->n_octaves) return 0;
    int rows = p->rows[octave], cols = p->cols[octave];
    size_t n = 0;
    for (int level = 1; level < VO_NUM_DOGS - 1; level++)
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) {
                int v = p->dog[octave][level][(size_t)y * cols + x], mn = 256, mx = -1;
                for (int l = level - 1; l <= level + 1; l++)
                    for (int u = y - 1; u <= y + 1; u++)
                        for (int w = x - 1; w <= x + 1; w++) {
                            int t = p->dog[octave][l][(size_t)clampi(u, 0, rows - 1) * cols + clampi(w, 0, cols - 1)];
                            if (t < mn) mn = t;
                            if (t > mx) mx = t;
                        }
                int cand = (v == mn || v == mx);
                if (mask) mask[((size_t)(level - 1) * rows + y) * cols + x] = (uint8_t)cand;
                if (cand && v >= min_contrast) {
                    if (out && n < cap) {
                        out[n].row = y + 1;
                        out[n].col = x + 1;
                        out[n].value = v;
                        out[n].padding = 1;
                        out[n].octave = octave;
                        out[n].level = level;
                    }
                    n++;
                }
            }
    return n;
}

/* ------------------------------------------------------------------------- */
/* CPU baseline driver                                                        */
/* ------------------------------------------------------------------------- */

int vo_baseline_frames(const uint8_t* frames, int n, int rows, int cols, int n_octaves, int threads,
                       unsigned long long* keypoints) {
    if (!frames || n <= 0 || rows <= 0 || cols <= 0 || threads <= 0) return -1;
    unsigned long long total = 0;
    int bad = 0;
    const size_t N = (size_t)rows * cols;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total) reduction(| : bad)
    for (int f = 0; f < n; f++) {
        const uint8_t* img = frames + (size_t)f * N;
        float* R = (float*)malloc(sizeof(float) * N);
        float* n2 = (float*)malloc(sizeof(float) * N);
        uint8_t* u8 = (uint8_t*)malloc(N);
        uint8_t* mask = (uint8_t*)malloc(N);
        if (!R || !n2 || !u8 || !mask) {
            bad |= 1;
        } else {
            float tm;
            bad |= vo_harris_response_u8(img, rows, cols, (size_t)cols, 0.04f, 3, R, sizeof(float) * (size_t)cols) != 0;
            bad |= vo_convert_scale_abs_f32(R, rows, cols, sizeof(float) * (size_t)cols, u8, (size_t)cols) != 0;
            bad |= vo_nms_strict_u8(u8, rows, cols, (size_t)cols, 3, mask, (size_t)cols) != 0;
            bad |= vo_nms2_f32(R, rows, cols, sizeof(float) * (size_t)cols, 5, n2, sizeof(float) * (size_t)cols, &tm) != 0;
            total += vo_harris_keypoints(n2, rows, cols, sizeof(float) * (size_t)cols, NULL, 0);
            if (n_octaves > 0) {
                vo_pyramid* p = vo_pyramid_build_u8(img, rows, cols, (size_t)cols, n_octaves, 1.6);
                if (!p) {
                    bad |= 1;
                } else {
                    for (int o = 0; o < p->n_octaves; o++) total += vo_dog_extrema(p, o, 3, 8, NULL, NULL, 0);
                    vo_pyramid_free(p);
                }
            }
        }
        free(R);
        free(n2);
        free(u8);
        free(mask);
    }
    if (keypoints) *keypoints = total;
    return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------- */
/* SIFT descriptor stage                            */
/* ------------------------------------------------------------------------- */

/* Rotation::cos_sin_of_angle (rotation.cpp:5-17): theta * (CV_PI / 180.0f) is evaluated in double
 * (CV_PI is a double literal) and returned as float (:6); cos / sin of that float.  Whether the
 * unqualified cos(angle) binds to ::cos(double) or to a float overload depends on what OpenCV's
 * headers include; for the angles the pipeline produces (multiples of 10 degrees) both give the
 * same float with glibc (checked in tests/test_sift_cpu.py), the double form is used here. */
void vo_cos_sin_deg(float theta_deg, float* c, float* s) {
    const float angle = (float)((double)theta_deg * (3.1415926535897932384626433832795 / (double)180.0f));
    *c = (float)cos((double)angle);
    *s = (float)sin((double)angle);
}

/* Rotation::rotate_pt_CW (rotation.cpp:19-27): int * float products and their difference / sum in
 * f32 (x86-64 SSE, no FMA: CMakeLists.txt sets no -march), truncated to int (cvttss2si). */
static void rotate_pt_cw(int px, int py, int cx, int cy, float c, float s, int* ox, int* oy) {
    const int rx = px - cx, ry = py - cy;
    const float a = (float)rx * c, b = (float)ry * s;
    const int x_r = cvtt_f32_i32(a - b);
    const float d = (float)rx * s, e = (float)ry * c;
    const int y_r = cvtt_f32_i32(d + e);
    *ox = x_r + cx;
    *oy = y_r + cy;
}

/* Rotation::getRotatedWindowPoints (rotation.cpp:112-130): the (window+1)^2 points of the square
 * [c - window/2, c + window/2]^2 rotated clockwise about the centre, rows (y) outer; xy[2q] = x,
 * xy[2q+1] = y.  The Mat argument of the reference is unused. */
int vo_rotated_window_points(int cx, int cy, int window, float theta_deg, int32_t* xy) {
    if (!xy || window <= 0) return -1;
    float c, s;
    vo_cos_sin_deg(theta_deg, &c, &s);
    const int padding = window / 2;
    int q = 0;
    for (int i = cy - padding; i <= cy + padding; i++)
        for (int j = cx - padding; j <= cx + padding; j++, q++) {
            int x, y;
            rotate_pt_cw(j, i, cx, cy, c, s, &x, &y);
            xy[2 * q] = x;
            xy[2 * q + 1] = y;
        }
    return 0;
}

/* SIFT(), Diff_of_Gauss.cpp:561-693, with rotateImageSection (:528-559), for the oriented
 * keypoints of one octave (output of vo_filter_keypoints: value = angle in degrees).
 *   :578-580  level images padded by maxPadding = 20 (replicate): magnitude, orientation
 *   :591      rotated window points about (col + 20, row + 20), window 16 -> 17 x 17 points
 *   :545      rotatedPoints[i * imgROI.rows + j]: the 16 x 16 ROI walks the 17-wide list with stride
 *             16 (literal; the list has 289 entries)
 *   :549-554  Mat::at<>(rotatedPoint.x, rotatedPoint.y): x is used as the ROW and y as the COLUMN
 *             (literal).  Mat::at does no range check in a release build, so the element read is
 *             linear index x * (cols + 40) + y of the continuous padded Mat; a keypoint is DEFINED
 *             iff all 256 indices lie inside the Mat's buffer -- otherwise the reference reads
 *             foreign memory (its own comment at :541 mentions the segmentation fault), reported
 *             here as defined[k] = 0 with an all-zero descriptor
 *   :616-618  GaussianBlur of the 16 x 16 magnitude ROI (its own Mat: borders reflect inside it),
 *             sigma = 1.5 * sigma(octave, level), CV_32F kernel width cvRound(8 sigma + 1) | 1
 *   :629-653  sixteen 4 x 4 sub-regions, row-major, 8 orientation bins each, nearest-bin indexing
 *             (int)(orientation * (8 / 360.0f)), magnitudes accumulated in pixel order
 *   :659-675  divide by the max, clip at 0.2f with std::min semantics, divide by the new max (an
 *             all-zero histogram gives 0/0 = NaN throughout, kept)
 * desc: n x 128 floats; defined: n bytes (may be NULL).  Returns the number of undefined
 * keypoints, or (size_t)-1 for invalid input (level / octave mismatch, histogram index out of range
 * = the reference's vector::at would throw). */
size_t vo_sift_descriptors(const vo_pyramid* p, int octave, const vo_point* kps, size_t n, float* desc, uint8_t* defined) {
    if (!p || octave < 0 || octave >= p->n_octaves || (!kps && n) || (!desc && n)) return (size_t)-1;
    const int rows = p->rows[octave], cols = p->cols[octave];
    const int windowSize = 16, maxPadding = 20;
    const int pr = rows + 2 * maxPadding, pc = cols + 2 * maxPadding;
    const size_t P = (size_t)rows * cols;
    float *pmag[VO_NUM_LEVELS] = {0}, *porient[VO_NUM_LEVELS] = {0}, *kern[VO_NUM_LEVELS] = {0};
    int kn[VO_NUM_LEVELS] = {0};
    size_t undefined = 0;
    int bad = 0;
    const int histoSize = 8, subregion = 4;
    const float reductionCoeff = (float)histoSize / 360.0f;
    for (size_t q = 0; q < n && !bad; q++) {
        const int level = kps[q].level;
        float* out = desc + q * 128;
        if (level < 0 || level >= VO_NUM_LEVELS || kps[q].octave != octave) {
            bad = 1;
            break;
        }
        if (!pmag[level]) {
            float* mag = (float*)malloc(4 * P);
            float* ori = (float*)malloc(4 * P);
            if (!mag || !ori) abort();
            vo_level_gradients(p->gauss[octave][level], rows, cols, (size_t)cols, NULL, NULL, mag, ori, 4 * (size_t)cols);
            pmag[level] = pad_replicate_f32(mag, rows, cols, maxPadding);
            porient[level] = pad_replicate_f32(ori, rows, cols, maxPadding);
            free(mag);
            free(ori);
            const double sigma = 1.5 * p->sigma[octave][level]; /* :616 */
            kn[level] = vo_gauss_ksize_f32(sigma);
            kern[level] = (float*)malloc(4 * (size_t)kn[level]);
            if (!pmag[level] || !porient[level] || !kern[level]) abort();
            vo_gauss_kernel_f32(kn[level], sigma, kern[level]);
        }
        int32_t pts[2 * 17 * 17];
        vo_rotated_window_points(kps[q].col + maxPadding, kps[q].row + maxPadding, windowSize, (float)kps[q].value, pts);
        float magROI[16 * 16], orientROI[16 * 16], magWeighted[16 * 16];
        int ok = 1;
        for (int i = 0; i < windowSize && ok; i++)
            for (int j = 0; j < windowSize; j++) {
                const int32_t* rp = pts + 2 * (i * windowSize + j); /* :545 */
                const long long e = (long long)rp[0] * pc + rp[1];   /* at<>(rp.x, rp.y), :549-554 */
                if (e < 0 || e >= (long long)pr * pc) {
                    ok = 0;
                    break;
                }
                magROI[i * windowSize + j] = pmag[level][e];
                orientROI[i * windowSize + j] = porient[level][e];
            }
        if (defined) defined[q] = (uint8_t)ok;
        if (!ok) {
            undefined++;
            for (int b = 0; b < 128; b++) out[b] = 0.0f;
            continue;
        }
        if (blur_f32_roi(magROI, windowSize, windowSize, 0, 0, windowSize, windowSize, kern[level], kn[level], magWeighted) != 0) abort();
        int nf = 0;
        for (int r0 = 0; r0 < windowSize && !bad; r0 += subregion)
            for (int c0 = 0; c0 < windowSize && !bad; c0 += subregion) { /* :637-652: columns advance first */
                float histo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = r0; i < r0 + subregion && !bad; i++)
                    for (int j = c0; j < c0 + subregion; j++) {
                        const int index = (int)(orientROI[i * windowSize + j] * reductionCoeff); /* :126 */
                        if (index < 0 || index >= histoSize) {
                            bad = 1;
                            break;
                        }
                        histo[index] += magWeighted[i * windowSize + j];
                    }
                for (int b = 0; b < histoSize; b++) out[nf++] = histo[b];
            }
        if (bad) break;
        float maxPeak = out[0]; /* *max_element: operator< scan, first element kept on NaN */
        for (int b = 1; b < 128; b++)
            if (maxPeak < out[b]) maxPeak = out[b];
        for (int b = 0; b < 128; b++) out[b] = out[b] / maxPeak; /* :661 */
        const float threshold = 0.2f;
        for (int b = 0; b < 128; b++) out[b] = threshold < out[b] ? threshold : out[b]; /* std::min(c, threshold), :668 */
        maxPeak = out[0];
        for (int b = 1; b < 128; b++)
            if (maxPeak < out[b]) maxPeak = out[b];
        for (int b = 0; b < 128; b++) out[b] = out[b] / maxPeak; /* :675 */
    }
    for (int l = 0; l < VO_NUM_LEVELS; l++) {
        free(pmag[l]);
        free(porient[l]);
        free(kern[l]);
    }
    return bad ? (size_t)-1 : undefined;
}
