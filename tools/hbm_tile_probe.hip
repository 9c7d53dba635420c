// What the octave kernels' STORE PATTERN sustains (round 5): eleven u8 planes of 3840 x 2160 per frame written tile by
// tile, each wave of a 256-thread workgroup owning a strip and writing it plane after plane with 16-byte stores, workgroups
// in the XCD-contiguous tile order of k_pyr_octave(_mx).  Nothing is read.  Variants of the piece one store instruction covers:
//   A  8 rows x 128 B   (the matrix kernel's flush: tile 128 x 128, wave strip 32 rows x 128 columns, 4 stores per plane)
//   B  4 rows x 256 B   (tile 256 x 64,  wave strip 16 rows x 256 columns)
//   C  2 rows x 512 B   (tile 512 x 32,  wave strip  8 rows x 512 columns)
//   D  1 row  x 1024 B  (tile 1024 x 16, wave strip  4 rows x 1024 columns)
//   E  as A, but the eleven planes of a frame interleaved row by row in ONE buffer (row pitch 11 x 3840): same bytes, one stream
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_tile_probe tools/hbm_tile_probe.hip && tools/hbm_tile_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); std::exit(1); } } while (0)

constexpr int W = 3840, H = 2160, NP = 11;

template <int PW, bool INTERLEAVE>  // PW: bytes of a row one store instruction covers (128 .. 1024)
__global__ __launch_bounds__(256) void k_tiles(uint8_t* __restrict__ out, size_t frame_bytes, int tiles_x, int tiles_y, unsigned seed) {
    constexpr int LPR = PW / 16, RPS = 64 / LPR;        // lanes per row piece, rows per store
    constexpr int TW = PW, SROWS = 4 * RPS, TH = 4 * SROWS;  // wave strip: SROWS rows x PW bytes (4 stores per plane), tile = 4 strips
    unsigned int bid = blockIdx.x;
    const unsigned int per_xcd = gridDim.x >> 3;
    if (bid < (per_xcd << 3)) bid = (bid & 7u) * per_xcd + (bid >> 3);
    const unsigned int tpf = tiles_x * tiles_y, fz = bid / tpf, rem = bid - fz * tpf;
    const int by = rem / tiles_x, bx = rem - by * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = bx * TW + 16 * (lane % LPR), y0 = by * TH + wave * SROWS + lane / LPR;
    if (x >= W) return;
    uint8_t* f = out + fz * frame_bytes;
    const uint4 v = make_uint4(seed, bid, lane, wave);
#pragma unroll 1
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 + i * RPS;
            if (y < H) {
                uint8_t* d = INTERLEAVE ? f + ((size_t)y * NP + p) * W + x : f + (size_t)p * W * H + (size_t)y * W + x;
                *reinterpret_cast<uint4*>(d) = v;
            }
        }
}

template <int PW, bool IL>
static float run(uint8_t* buf, size_t frame_bytes, int frames, hipEvent_t e0, hipEvent_t e1) {
    constexpr int RPS = 64 / (PW / 16), TH = 16 * RPS;
    const int tx = (W + PW - 1) / PW, ty = (H + TH - 1) / TH;
    auto go = [&] { hipLaunchKernelGGL((k_tiles<PW, IL>), dim3(tx * ty * frames), dim3(256), 0, 0, buf, frame_bytes, tx, ty, 7u); };
    go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    const int frames = 64;
    const size_t frame_bytes = (size_t)NP * W * H;
    uint8_t* buf;
    CK(hipMalloc((void**)&buf, frame_bytes * frames));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double gb = (double)frame_bytes * frames / 1e9;
    const float a = run<128, false>(buf, frame_bytes, frames, e0, e1), b = run<256, false>(buf, frame_bytes, frames, e0, e1);
    const float c = run<512, false>(buf, frame_bytes, frames, e0, e1), d = run<1024, false>(buf, frame_bytes, frames, e0, e1);
    const float e = run<128, true>(buf, frame_bytes, frames, e0, e1);
    std::printf("{\"GB_written\": %.2f, \"A_8x128B_GBps\": %.0f, \"B_4x256B_GBps\": %.0f, \"C_2x512B_GBps\": %.0f, \"D_1x1024B_GBps\": %.0f, \"E_8x128B_interleaved_planes_GBps\": %.0f}\n", gb,
                gb / a * 1e3, gb / b * 1e3, gb / c * 1e3, gb / d * 1e3, gb / e * 1e3);
    return 0;
}
