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
int vo_gauss_taps_q8(int n, double sigma, uint16_t* taps) {
    if (n <= 0 || (n & 1) == 0 || n > 4096) return -1;
    double* kf = (double*)malloc(sizeof(double) * (size_t)n);
    if (!kf) return -1;
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

static inline uint8_t cvt_abs_u8(float x) {
    float a = fabsf(x);
    if (!(a < 255.5f)) return a != a ? 0 : 255; /* saturate; NaN -> 0 */
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

/* cv::fastAtan2 / hal::fastAtan32f scalar form (OpenCV 4.x mathfuncs_core.simd.hpp). */
float vo_fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)DBL_EPSILON;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
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
