// MFMA feasibility probe - NOT part of the product library, NOT on the default path:
// BASELINE.json's north star rules MFMA out ("no dense contraction here"); the round-2 accounting says the
// pyramid kernels are bound by the issue rate of the packed dot instructions.  This
// standalone program measures what the same arithmetic costs as a banded-Toeplitz product on the matrix
// cores, so that the rule's owner can decide.
//
// One Gaussian level of one octave, G = (sum_y sum_x ty*tx*p + 32768) >> 16 (exact integer
// arithmetic, so any evaluation order gives the reference result), as two chained v_mfma_i32_32x32x32_i8:
//
//   pass 1 (vertical)    C1^T[x, y'] = sum_y  P'^T[x, y] * Tv^T[y, y']      A = pixels (lane = column x, 16
//                         consecutive rows per lane: the byte-transposed LDS image of k_pyr_octave), B = taps as
//                         a 32 x 32 band matrix (constant), P' = P - 128 (i8 is signed; the taps sum to 256, so
//                         C1 = H - 32768 with H the 16-bit row sum).
//   hand-off             C1^T has the column y' on the LANE and the rows x in its 16 registers - exactly the
//                         shape of a B operand whose K index is x.  No lane movement, no LDS: the 16 registers
//                         are split into a signed high byte plane and a low byte plane (4 v_perm + 1 v_xor per
//                         4 values) because the matrix cores take 8-bit operands and H has 16 bits.
//   pass 2 (horizontal)  C2^T[x', y'] = sum_x Th'[x', x] * H^T[x, y']       A = taps band matrix with its K
//                         columns permuted to the register order of C1 (constant), B = the two byte planes:
//                         two MFMAs per K step, G = ((C2hi << 8) + C2lo) >> 16 with the rounding constant and
//                         the biases folded into C2lo's initial value.
//   output                C2^T has the image ROW y' on the lane and 16 columns in registers: after packing
//                         and one v_permlane32_swap pair each lane owns 16 consecutive bytes of its row.
//
// Per 32 x 32 output block and level: 3 + 6 MFMAs with kernels of up to 65 taps (FORM 3: K windows of 96),
// 2 + 4 with up to 33 taps (FORM 2: K windows of 64, the output block sits between two input blocks).
//
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o tools/mfma_probe tools/mfma_probe.hip
//   tools/mfma_probe <in.bin> <out.bin> [planes] [reps]      (driver + bit-exact check: tools/mfma_probe.py)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(e)                                                              \
    do {                                                                   \
        hipError_t r_ = (e);                                               \
        if (r_ != hipSuccess) {                                            \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));   \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

constexpr int TW = 128, TH = 128;   // output tile of one 256-thread workgroup: 4 waves x 32 rows
constexpr int R = 32;               // staged halo (rows and columns)
constexpr int RQ = (TH + 2 * R) / 4;
constexpr int RW = TW + 2 * R;
constexpr int RWP = RW + 8;         // dword pitch of a row quad: 4 * RWP = 32 mod 64 -> the two lane halves hit different banks

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

struct Tables {       // operand fragments in lane order: [step][lane] 16 bytes
    v4i b1[3][64];    // pass 1 B: taps band matrix, K = rows y
    v4i a2[3][64];    // pass 2 A: taps band matrix, K = columns x in the register order of the C layout
};

// LEVELS > 1 repeats the level on the staged tile (same taps, one output plane set per repetition): the
// difference to LEVELS = 1 is the marginal cost of a level in a fused octave kernel (staging paid once).
template <int FORM, int LEVELS>   // FORM = 3: K windows [-32, 64) (kernels up to 65 taps); 2: [-16, 48) (up to 33 taps)
__global__ __launch_bounds__(256) void k_mfma_level(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols,
                                                     int pitch, size_t plane, const Tables* __restrict__ tab, size_t level_stride) {
    __shared__ __attribute__((aligned(16))) uint32_t rp[RQ * RWP];  // [row quad][column]: dword = 4 vertically adjacent pixels - 128
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int X0 = blockIdx.x * TW, Y0 = blockIdx.y * TH;
    const uint8_t* img = src + blockIdx.z * plane;
    uint8_t* out = dst + blockIdx.z * plane;

    // ---- stage the tile + halo, byte-transposed, reflect-101 resolved, biased by -128 (x ^ 0x80) -------------
    const bool interior = X0 - R >= 0 && X0 + TW + R <= cols && Y0 - R >= 0 && Y0 + TH + R <= rows;
    if (interior) {
        for (int it = tid; it < RQ * (RW / 16); it += 256) {
            const int yq = it / (RW / 16), xs = it - yq * (RW / 16);
            const uint8_t* p = img + (size_t)(Y0 - R + 4 * yq) * pitch + (X0 - R + 16 * xs);
            uint4 a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const uint4*>(p + (size_t)k * pitch);
            const uint32_t* aw[4] = {&a[0].x, &a[1].x, &a[2].x, &a[3].x};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t r0 = aw[0][q], r1 = aw[1][q], r2 = aw[2][q], r3 = aw[3][q];
                const uint32_t p01l = __builtin_amdgcn_perm(r1, r0, 0x05010400), p01h = __builtin_amdgcn_perm(r1, r0, 0x07030602);
                const uint32_t p23l = __builtin_amdgcn_perm(r3, r2, 0x05010400), p23h = __builtin_amdgcn_perm(r3, r2, 0x07030602);
                uint4 t;
                t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100) ^ 0x80808080u;
                t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302) ^ 0x80808080u;
                t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100) ^ 0x80808080u;
                t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302) ^ 0x80808080u;
                *reinterpret_cast<uint4*>(rp + yq * RWP + 16 * xs + 4 * q) = t;
            }
        }
    } else {
        for (int it = tid; it < RQ * RW; it += 256) {
            const int yq = it / RW, x = it - yq * RW;
            const int gx = reflect101(X0 - R + x, cols);
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) w |= (uint32_t)img[(size_t)reflect101(Y0 - R + 4 * yq + k, rows) * pitch + gx] << (8 * k);
            rp[yq * RWP + x] = w ^ 0x80808080u;
        }
    }
    __syncthreads();

    constexpr int NS = FORM;                     // K steps per pass
    constexpr int OFF = FORM == 3 ? 32 : 16;     // K window starts OFF before the output block
    const int m = lane & 31, h = lane >> 5;
    v4i b1[NS], a2[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) b1[s] = tab->b1[s][lane], a2[s] = tab->a2[s][lane];
    const int Yw = 32 * wave;                    // this wave's 32 output rows inside the tile
    // G = ((C2hi << 8) + C2lo) >> 16 with C2lo starting at 256 * (128 + 32768) + 32768 (biases of the two byte planes + round half up)
    constexpr int kLoInit = 256 * (128 + 32768) + 32768;

    // input x blocks: FORM 3: tile columns [-32, TW + 32), output block o uses inputs o-1, o, o+1;
    //                 FORM 2: tile columns [-16, TW + 16), output block o uses inputs o, o+1 (it sits between them)
    constexpr int NIN = TW / 32 + (FORM == 3 ? 2 : 1);
#pragma unroll 1
    for (int lev = 0; lev < LEVELS; ++lev) {
    if (LEVELS > 1) {  // a fused kernel loads each level's fragments: keep the loads inside the loop
#pragma unroll
        for (int s = 0; s < NS; ++s) b1[s] = tab->b1[s][lane], a2[s] = tab->a2[s][lane];
    }
    // streaming over the input blocks of the wave's row: block ib goes through pass 1 and the byte split into a
    // ring of NS converted blocks; as soon as the ring holds the NS blocks of output block ob = ib - (NS - 1) its
    // pass 2 runs.  Live state: the ring (NS x 8 registers), one pass-1 and two pass-2 accumulator tiles.
    v4i hi[NS], lo[NS];
#pragma unroll
    for (int ib = 0; ib < NIN; ++ib) {
        const int xcol = R - OFF + 32 * ib + m;  // LDS column of this lane's image column
        v16i c1 = {};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int q0 = (R - OFF + Yw + 32 * s + 16 * h) >> 2;  // first of the 4 row quads = 16 rows of this K half-step
            v4i a;
            a[0] = (int)rp[(q0 + 0) * RWP + xcol];
            a[1] = (int)rp[(q0 + 1) * RWP + xcol];
            a[2] = (int)rp[(q0 + 2) * RWP + xcol];
            a[3] = (int)rp[(q0 + 3) * RWP + xcol];
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1[s], c1, 0, 0, 0);
        }
        // C1 = H - 32768 in [-32768, 32512]: signed high byte as it is, low byte - 128 (x ^ 0x80)
        const int slot = ib % NS;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)c1[4 * d + 1], (uint32_t)c1[4 * d + 0], 0x05010400);  // (lo0, lo1, hi0, hi1)
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)c1[4 * d + 3], (uint32_t)c1[4 * d + 2], 0x05010400);
            lo[slot][d] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100) ^ 0x80808080u);
            hi[slot][d] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302);
        }
        if (ib < NS - 1) continue;
        const int ob = ib - (NS - 1);
        v16i chi = {}, clo;
#pragma unroll
        for (int v = 0; v < 16; ++v) clo[v] = kLoInit;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            chi = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2[s], hi[(ob + s) % NS], chi, 0, 0, 0);
            clo = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2[s], lo[(ob + s) % NS], clo, 0, 0, 0);
        }
        // lane = image row, registers = 16 columns in groups of 4: (v & 3) + 8 (v >> 2) + 4 h
        uint32_t g[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = ((uint32_t)chi[4 * d + k] << 8) + (uint32_t)clo[4 * d + k];
            const uint32_t e = __builtin_amdgcn_perm(w[1], w[0], 0x0c0c0602), o = __builtin_amdgcn_perm(w[3], w[2], 0x06020c0c);
            g[d] = e | o;  // bytes 2 of the four accumulators
        }
        // lane (row, h) holds the dwords of columns 8d + 4h; swap so that h = 0 owns columns 0..15 and h = 1 columns 16..31
        {
            auto r0 = __builtin_amdgcn_permlane32_swap(g[0], g[2], false, false);  // g[0]: lower lanes keep D0, upper lanes get D2; g[2]: lower E0, upper E2
            auto r1 = __builtin_amdgcn_permlane32_swap(g[1], g[3], false, false);
            g[0] = r0[0], g[2] = r0[1], g[1] = r1[0], g[3] = r1[1];
        }
        const int y = Y0 + Yw + m, x = X0 + 32 * ob + 16 * h;
        if (y < rows && x < cols)  // cols is a multiple of 16 in this probe
            *reinterpret_cast<uint4*>(out + lev * level_stride + (size_t)y * pitch + x) = make_uint4(g[0], g[2], g[1], g[3]);
    }
    }  // lev
}

// ------------------------------------------------------------------------------------------------------------
static int rho(int h, int j) { return (j & 3) + 8 * (j >> 2) + 4 * h; }  // row of C-layout register j in lane half h

static void build_tables(const std::vector<uint8_t>& taps, int form, Tables& T) {
    const int n = (int)taps.size(), r = n / 2, off = form == 3 ? 32 : 16;
    std::memset(&T, 0, sizeof(T));
    auto tap = [&](int idx) -> int { return idx >= 0 && idx < n ? taps[idx] : 0; };
    for (int s = 0; s < form; ++s)
        for (int l = 0; l < 64; ++l) {
            const int m = l & 31, h = l >> 5;
            int8_t b[16], a[16];
            for (int j = 0; j < 16; ++j) {
                const int y = -off + 32 * s + 16 * h + j;   // input row relative to the output block, output row = m
                b[j] = (int8_t)tap(y - m + r);
                // FORM 3: input block s covers columns [-32 + 32 s, ..), output column = m
                // FORM 2: input block s covers columns [-16 + 32 s, ..)
                const int x = -off + 32 * s + rho(h, j);
                a[j] = (int8_t)tap(x - m + r);
            }
            std::memcpy(&T.b1[s][l], b, 16);
            std::memcpy(&T.a2[s][l], a, 16);
        }
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: mfma_probe <in.bin> <out.bin> [planes] [reps]\n");
        return 2;
    }
    const int planes = argc > 3 ? std::atoi(argv[3]) : 16, reps = argc > 4 ? std::atoi(argv[4]) : 10;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t head[3];  // rows, cols, ntaps
    if (std::fread(head, 4, 3, f) != 3) return 2;
    const int rows = head[0], cols = head[1], n = head[2];
    std::vector<uint8_t> taps(n), img((size_t)rows * cols);
    if (std::fread(taps.data(), 1, n, f) != (size_t)n || std::fread(img.data(), 1, img.size(), f) != img.size()) return 2;
    std::fclose(f);
    if (cols % 16 || (n & 1) == 0 || n > 65) {
        std::fprintf(stderr, "probe limits: cols %% 16 == 0, odd kernels up to 65 taps\n");
        return 2;
    }
    const int form = n <= 33 ? 2 : 3;
    Tables T;
    build_tables(taps, form, T);
    const size_t plane = (size_t)rows * cols;
    uint8_t *d_src, *d_dst;
    Tables* d_tab;
    CK(hipMalloc((void**)&d_src, plane * planes));
    const int max_levels = 6;
    CK(hipMalloc((void**)&d_dst, plane * planes * max_levels));
    CK(hipMalloc((void**)&d_tab, sizeof(Tables)));
    for (int p = 0; p < planes; ++p) CK(hipMemcpy(d_src + p * plane, img.data(), plane, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, &T, sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemset(d_dst, 0xEE, plane * planes * max_levels));
    const dim3 grid((cols + TW - 1) / TW, (rows + TH - 1) / TH, planes);
    const size_t lstride = plane * planes;
    auto launch = [&](int levels) {
        if (form == 3 && levels == 1)
            hipLaunchKernelGGL((k_mfma_level<3, 1>), grid, dim3(256), 0, 0, d_src, d_dst, rows, cols, cols, plane, d_tab, lstride);
        else if (form == 3)
            hipLaunchKernelGGL((k_mfma_level<3, 6>), grid, dim3(256), 0, 0, d_src, d_dst, rows, cols, cols, plane, d_tab, lstride);
        else if (levels == 1)
            hipLaunchKernelGGL((k_mfma_level<2, 1>), grid, dim3(256), 0, 0, d_src, d_dst, rows, cols, cols, plane, d_tab, lstride);
        else
            hipLaunchKernelGGL((k_mfma_level<2, 6>), grid, dim3(256), 0, 0, d_src, d_dst, rows, cols, cols, plane, d_tab, lstride);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_ms = [&](int levels) {
        launch(levels);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch(levels);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        return t / reps;
    };
    const float ms6 = time_ms(6);
    std::vector<uint8_t> lvl5(plane);
    CK(hipMemcpy(lvl5.data(), d_dst + 5 * lstride, plane, hipMemcpyDeviceToHost));
    CK(hipMemset(d_dst, 0xEE, plane * planes));
    float ms = time_ms(1);
    std::vector<uint8_t> outp(plane), last(plane);
    CK(hipMemcpy(outp.data(), d_dst, plane, hipMemcpyDeviceToHost));
    CK(hipMemcpy(last.data(), d_dst + (size_t)(planes - 1) * plane, plane, hipMemcpyDeviceToHost));
    f = std::fopen(argv[2], "wb");
    std::fwrite(outp.data(), 1, plane, f);
    std::fclose(f);
    const double px = (double)plane * planes;
    const int mfma_per_block = form == 3 ? 9 : 6;
    // useful MACs: two passes of n taps per pixel; issued MACs: 32768 per MFMA (halo blocks of pass 1 included)
    const double blocks_out = (double)grid.x * grid.y * planes * (TW / 32) * 4;
    const double mfmas = (double)grid.x * grid.y * planes * 4 * ((TW / 32 + (form == 3 ? 2 : 1)) * form + (TW / 32) * 2 * form);
    std::printf("{\"rows\": %d, \"cols\": %d, \"taps\": %d, \"form\": %d, \"planes\": %d, \"ms_per_launch\": %.4f, \"gpixel_per_s\": %.2f, "
                "\"useful_mac_per_clk_per_cu\": %.1f, \"issued_mfma\": %.0f, \"issued_mac_per_clk_per_cu\": %.1f, \"mfma_per_32x32_block_interior\": %d, "
                "\"hbm_GBps_in_plus_out\": %.1f, \"planes_identical\": %s, \"blocks_out\": %.0f, \"ms_six_levels\": %.4f, \"ms_marginal_per_level\": %.4f, "
                "\"six_levels_GBps_in_plus_6out\": %.1f, \"level5_identical\": %s}\n",
                rows, cols, n, form, planes, ms, px / ms / 1e6, 2.0 * n * px / (ms * 1e-3) / 2.4e9 / 256, mfmas, mfmas * 32768 / (ms * 1e-3) / 2.4e9 / 256,
                mfma_per_block, 2 * px / ms / 1e6, std::memcmp(outp.data(), last.data(), plane) == 0 ? "true" : "false", blocks_out, ms6, (ms6 - ms) / 5,
                7 * px / ms6 / 1e6, std::memcmp(outp.data(), lvl5.data(), plane) == 0 ? "true" : "false");
    return 0;
}
