// Host-side parameter math of the C ABI (no GPU involved): Gaussian tap quantisation,
// pyramid geometry and the batch output layout.  Each rule that restates OpenCV
// behaviour is isolated in one function so it can be corrected if an OpenCV build ever
// becomes available (SURVEY.md section 7 "No external truth").
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/vslam.h"
#include "vslam_internal.h"

namespace vslam {

// cvRound(double): round half to even under the default rounding mode.
static inline long cv_round(double v) { return std::lrint(v); }

int gauss_ksize_u8(double sigma) { return (int)cv_round(sigma * 3 * 2 + 1) | 1; }

// cv::getGaussianKernel as used by the CV_8U fixed-point GaussianBlur (OpenCV >= 4.5.1):
// normalised double kernel, then 8.8 quantisation with error diffusion from the outermost
// tap inwards; the centre tap takes the remainder so the taps sum to exactly 256.
static void gauss_kernel_f64(int n, double sigma, std::vector<double>& kf) {
    kf.assign((size_t)n, 0.0);
    const int half = (n - 1) / 2;
    static const double k3[] = {0.25, 0.5, 0.25};
    static const double k5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double k7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    if (sigma <= 0 && n == 1) {
        kf[0] = 1.0;
    } else if (sigma <= 0 && n == 3) {
        kf.assign(k3, k3 + 3);
    } else if (sigma <= 0 && n == 5) {
        kf.assign(k5, k5 + 5);
    } else if (sigma <= 0 && n == 7) {
        kf.assign(k7, k7 + 7);
    } else {
        const double s = sigma > 0 ? sigma : std::fma((double)n, 0.15, 0.35);
        const double scale = -0.125 / (s * s);
        double sum = 0.0;
        for (int i = 0, x = 1 - n; i < half; ++i, x += 2) {
            kf[i] = std::exp((double)(x * x) * scale);
            sum += kf[i];
        }
        sum = sum * 2.0 + 1.0;
        const double inv = 1.0 / sum;
        for (int i = 0; i < half; ++i) kf[n - 1 - i] = kf[i] = kf[i] * inv;
        kf[half] = inv;
    }
}

int gauss_ksize_f32(double sigma) { return (int)cv_round(sigma * 4 * 2 + 1) | 1; }

// cv::getGaussianKernel(n, sigma, CV_32F): the same normalised f64 kernel narrowed to f32 (the
// kernel GaussianBlur builds for CV_32F images, Diff_of_Gauss.cpp:348).
bool gauss_kernel_f32(int n, double sigma, std::vector<float>& out) {
    if (n <= 0 || (n & 1) == 0 || n > (1 << 20)) return false;
    std::vector<double> kf;
    gauss_kernel_f64(n, sigma, kf);
    out.resize((size_t)n);
    for (int i = 0; i < n; ++i) out[(size_t)i] = (float)kf[(size_t)i];
    return true;
}

bool gauss_taps_q8(int n, double sigma, uint16_t* taps) {
    if (n <= 0 || (n & 1) == 0 || n > VSLAM_MAX_KSIZE) return false;
    std::vector<double> kf;
    gauss_kernel_f64(n, sigma, kf);
    double err = 0.0;
    long acc = 0;
    for (int i = 0; i < n / 2; ++i) {
        const double adj = kf[i] * 256.0 + err;
        const long q = cv_round(adj);
        err = adj - (double)q;
        taps[i] = taps[n - 1 - i] = (uint16_t)q;
        acc += q;
    }
    taps[n / 2] = (uint16_t)(256 - 2 * acc);
    return true;
}

// Quantised Gaussian taps with their all-zero tails removed.  Error diffusion leaves the
// outermost taps of every sigma >= 1.6 kernel at 0 (sigma = 5.08: 31 -> 29 taps; sigma = 40.6:
// 245 -> 223), and a zero tap contributes nothing to the exact integer sum, so the kernels may
// use the trimmed window: same result, fewer MACs, smaller halos.
bool gauss_taps_q8_trimmed(int n, double sigma, std::vector<uint16_t>& out) {
    std::vector<uint16_t> t((size_t)(n > 0 ? n : 0));
    if (!gauss_taps_q8(n, sigma, t.data())) return false;
    int z = 0;
    while (2 * z + 1 < n && t[z] == 0 && t[n - 1 - z] == 0) ++z;
    out.assign(t.begin() + z, t.end() - z);
    return true;
}

double sigma_at(double sigma0, int octave, int level) {
    const double k = std::pow(2.0f, 1.0f / (double)3);  // GaussPyramid.hpp:69
    return std::pow(2, octave) * sigma0 * std::pow(k, level);
}

int auto_num_octaves(int rows, int cols) {
    const int m = rows < cols ? rows : cols;
    return (int)std::floor(std::log2((double)m)) - 4;
}

void half_size(int rows, int cols, int* r, int* c) {
    *r = (int)cv_round(rows * 0.5);
    *c = (int)cv_round(cols * 0.5);
}

void extrema_lattice(int rows, int cols, int window, int* lr, int* lc) {
    const int pad = (window - 1) / 2;
    *lr = rows > pad ? (rows - pad + window - 1) / window : 0;
    *lc = cols > pad ? (cols - pad + window - 1) / window : 0;
}

int make_layout(const vslam_params* p, vslam_batch_layout* L) {
    if (!p || !L || p->rows <= 0 || p->cols <= 0 || p->n_octaves < 0 || p->n_octaves > VSLAM_MAX_OCTAVES)
        return VSLAM_ERR_INVALID;
    if (p->n_octaves > 0 && (!(p->sigma0 > 0) || p->extrema_window < 3 || (p->extrema_window & 1) == 0))
        return VSLAM_ERR_INVALID;
    if (p->n_octaves > 0 && p->extrema_dense && (p->extrema_window != 3 || p->localize || p->orient)) return VSLAM_ERR_INVALID;
    std::memset(L, 0, sizeof(*L));
    L->n_octaves = p->n_octaves;
    int r = p->rows * 2, c = p->cols * 2;
    size_t off = 0, woff = 0, sum_p = 0;
    for (int o = 0; o < p->n_octaves; ++o) {
        if (r <= 0 || c <= 0) return VSLAM_ERR_INVALID;
        L->rows[o] = r;
        L->cols[o] = c;
        if (p->extrema_dense)  // dense 3x3x3 test: one site per pixel
            L->lat_rows[o] = r, L->lat_cols[o] = c;
        else
            extrema_lattice(r, c, p->extrema_window, &L->lat_rows[o], &L->lat_cols[o]);
        L->lat_words[o] = (L->lat_cols[o] + 63) / 64;
        L->octave_offset[o] = off;
        L->bits_offset[o] = woff;
        L->pitch[o] = (c + 15) & ~15;
        const size_t P = (size_t)r * c;
        sum_p += P;
        off += (VSLAM_NUM_LEVELS + VSLAM_NUM_DOGS) * ((size_t)r * L->pitch[o]);
        woff += (size_t)3 * L->lat_rows[o] * L->lat_words[o];
        half_size(r, c, &r, &c);
    }
    L->pyramid_frame_bytes = (off + 255) & ~(size_t)255;
    L->bits_frame_words = woff;
    const size_t N = (size_t)p->rows * p->cols;
    L->algorithmic_bytes_harris = 6 * N;            // u8 in + f32 response + u8 mask
    L->algorithmic_bytes_dog = N + 11 * sum_p;      // u8 in + 6 Gaussian + 5 DoG stacks
    return VSLAM_OK;
}

}  // namespace vslam

extern "C" {

int vslam_version(void) { return VSLAM_VERSION; }

const char* vslam_status_string(int s) {
    switch (s) {
        case VSLAM_OK: return "ok";
        case VSLAM_ERR_INVALID: return "invalid argument";
        case VSLAM_ERR_HIP: return "HIP runtime error";
        case VSLAM_ERR_NOMEM: return "out of memory";
        case VSLAM_ERR_UNSUPPORTED: return "unsupported parameter combination";
        case VSLAM_ERR_RANGE: return "octave/level out of range";
        default: return "unknown status";
    }
}

int vslam_gauss_ksize_u8(double sigma) { return vslam::gauss_ksize_u8(sigma); }

int vslam_gauss_taps_q8(int n, double sigma, uint16_t* taps) {
    if (!taps) return VSLAM_ERR_INVALID;
    return vslam::gauss_taps_q8(n, sigma, taps) ? VSLAM_OK : VSLAM_ERR_INVALID;
}

double vslam_sigma_at(double sigma0, int octave, int level) { return vslam::sigma_at(sigma0, octave, level); }
int vslam_auto_num_octaves(int rows, int cols) { return vslam::auto_num_octaves(rows, cols); }

void vslam_half_size(int rows, int cols, int* out_rows, int* out_cols) {
    vslam::half_size(rows, cols, out_rows, out_cols);
}

void vslam_extrema_lattice(int rows, int cols, int window, int* lat_rows, int* lat_cols) {
    vslam::extrema_lattice(rows, cols, window, lat_rows, lat_cols);
}

void vslam_params_default(vslam_params* p, int rows, int cols) {
    if (!p) return;
    p->rows = rows;
    p->cols = cols;
    p->n_octaves = 4;        // Diff_of_Gauss.cpp:742
    p->sigma0 = 1.6;         // Diff_of_Gauss.cpp:743
    p->harris_k = 0.04f;     // Harris_corners.cpp:36
    p->do_harris = 1;
    p->extrema_window = 3;   // Diff_of_Gauss.cpp:772
    p->min_contrast = 8;
    p->localize = 0;
    p->orient = 0;
    p->extrema_dense = 0;
    // list capacities follow the frame area: 1 record per 8 pixels (Harris, DoG), 1 per 32 (oriented),
    // rounded up to 4096 -> 262144 / 262144 / 65536 at 1920x1080 (the synthetic 1080p frame gives ~62k + ~74k);
    // small frames keep a floor that holds even an all-noise frame's lists (65536 / 65536 / 16384)
    const unsigned long long N = rows > 0 && cols > 0 ? (unsigned long long)rows * (unsigned long long)cols : 0ull;
    auto cap_of = [](unsigned long long v, unsigned long long lo) {
        v = (v + 4095ull) & ~4095ull;
        return (uint32_t)(v < lo ? lo : v > 0x40000000ull ? 0x40000000ull : v);
    };
    p->oriented_cap = cap_of(N / 32, 1ull << 14);
    p->harris_cap = cap_of(N / 8, 1ull << 16);
    p->dog_cap = cap_of(N / 8, 1ull << 16);
}

int vslam_batch_out_required(const vslam_params* p, int n_frames, vslam_batch_out* z) {
    vslam_batch_layout L;
    if (!z || n_frames <= 0 || vslam::make_layout(p, &L) != VSLAM_OK) return VSLAM_ERR_INVALID;
    const size_t n = (size_t)n_frames, N = (size_t)p->rows * p->cols;
    z->struct_size = sizeof(vslam_batch_out);
    z->response_bytes = n * N * sizeof(float);
    z->nms_mask_bytes = n * N;
    z->nms2_bytes = n * N * sizeof(float);
    z->harris_kps_bytes = n * p->harris_cap * sizeof(vslam_kp);
    z->harris_counts_bytes = n * sizeof(uint32_t);
    z->pyramid_bytes = n * L.pyramid_frame_bytes;
    z->extrema_bits_bytes = n * L.bits_frame_words * sizeof(uint64_t);
    z->dog_points_bytes = n * p->dog_cap * sizeof(vslam_point);
    z->dog_counts_bytes = n * sizeof(uint32_t);
    z->oriented_points_bytes = n * p->oriented_cap * sizeof(vslam_point);
    z->oriented_counts_bytes = n * sizeof(uint32_t);
    z->oriented_survivors_bytes = n * sizeof(uint32_t);
    z->descriptors_bytes = n * p->oriented_cap * 128 * sizeof(float);
    z->descriptor_defined_bytes = n * p->oriented_cap;
    return VSLAM_OK;
}

int vslam_batch_layout_query(const vslam_params* p, vslam_batch_layout* out) { return vslam::make_layout(p, out); }

void vslam_cos_sin_deg(float theta_deg, float* c, float* s) {
    // rotation.cpp:5-7: theta * (CV_PI / 180.0f) -- double arithmetic, float result; :16 cos / sin
    const float angle = (float)((double)theta_deg * (3.1415926535897932384626433832795 / (double)180.0f));
    if (c) *c = (float)std::cos((double)angle);
    if (s) *s = (float)std::sin((double)angle);
}

int vslam_rotated_window_points(int cx, int cy, int window, float theta_deg, int32_t* xy) {
    if (!xy || window <= 0) return VSLAM_ERR_INVALID;
    float c, s;
    vslam_cos_sin_deg(theta_deg, &c, &s);
    const int padding = window / 2;  // rotation.cpp:114
    int q = 0;
    for (int i = cy - padding; i <= cy + padding; ++i)      // :123
        for (int j = cx - padding; j <= cx + padding; ++j, ++q) {  // :124
            const int rx = j - cx, ry = i - cy;             // rotate_pt_CW, :19-27
            const float a = (float)rx * c, b = (float)ry * s, d = (float)rx * s, e = (float)ry * c;
            xy[2 * q] = (int)(a - b) + cx;
            xy[2 * q + 1] = (int)(d + e) + cy;
        }
    return VSLAM_OK;
}

int vslam_descriptor_file_write(const char* path, const float* desc, size_t n) {
    if (!path || (!desc && n) || n > 0x7fffffff) return VSLAM_ERR_INVALID;
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return VSLAM_ERR_INVALID;
    const int32_t head[3] = {(int32_t)n, 128, 24};  // vecSize, histoLength, sizeof(std::vector<float>) (:843-845)
    bool ok = std::fwrite(head, sizeof(int32_t), 3, f) == 3;
    if (n) ok = ok && std::fwrite(desc, sizeof(float) * 128, n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? VSLAM_OK : VSLAM_ERR_INVALID;
}

void vslam_points16_expand(const vslam_point16* in, size_t n, vslam_point* out) {
    if (!in || !out) return;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t t = in[i].tag;
        out[i].row = in[i].row, out[i].col = in[i].col, out[i].value = in[i].value;
        out[i].padding = (int32_t)(t >> 16), out[i].octave = (int32_t)((t >> 8) & 0xffu), out[i].level = (int32_t)(t & 0xffu);
    }
}

}  // extern "C"
