// Matrix-core (v_mfma_i32_32x32x32_i8) form of the LDS-tiled octave kernels: OPT-IN, never the default path
// (kernels_pyramid_mx.hip.h says why).  Its own translation unit because of its compiler flag (Makefile).
#include "vslam_mx.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels_pyramid_mx.hip.h"
// The 16 x 16 x 64 form of octaves 0-1 (round 6: measured at parity with the 32 x 32 x 32 form - both run within 30 % of the time
// the planes' stores alone take, profiles/r06_mx16_*.txt) is compiled into the diagnostics build only (VSLAM_MX_FORM=16).
#ifdef VSLAM_DIAGNOSTICS
#include "kernels_pyramid_mx16.hip.h"
#define VSLAM_MX16 1
#else
#define VSLAM_MX16 0
#endif

namespace vslam {

// Octave 0's kernel (configuration 1: fused x2 upsample + lattice scan) is compiled in a translation unit of its own
// (vslam_mx0.hip = this file with VSLAM_MX_OCT0_TU defined) under -amdgpu-sched-strategy=max-ilp: same box, four
// alternations, that scheduling takes octave 0 from 6.8-7.1 to 6.4-6.6 ms per 256-frame step and costs the other three
// configurations 4-5 % (profiles/r05_mx_maxilp_ab.txt) - so only octave 0 gets it.
hipError_t mx_prepare_oct0();
hipError_t mx_launch_oct0(hipStream_t stream, const void* d_table, const uint8_t* base, size_t bframe, uint8_t* oct_out, size_t pframe, int rows,
                          int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch, const MxScan* scan, int up2_step);

#ifndef VSLAM_MX_OCT0_TU
template <class CFG>
static bool widths_match(const int ke[6]) {
    for (int l = 0; l < 6; ++l)
        if (ke[l] != CFG::n(l)) return false;
    return true;
}

// Octaves 0 and 1 exist in two MFMA shapes: 32 x 32 x 32 (configurations 1, 2: rounds 4-5, the form in use) and 16 x 16 x 64
// (5, 6: round 6, kernels_pyramid_mx16.hip.h: a quarter of the accumulator registers, four waves per SIMD - and no faster: the
// kernel's time is the planes' stores).  The diagnostics build carries the second form for A/B runs (VSLAM_MX_FORM=16 ->
// mx_set_form, process-wide, before the first batch call).
static int g_mx_form = 32;
void mx_set_form(int form) { g_mx_form = (VSLAM_MX16 && form == 16) ? 16 : 32; }
static bool form16() { return VSLAM_MX16 && g_mx_form == 16; }

int mx_config_for(const int ke[6]) {
    if (widths_match<MxCfgOct0>(ke)) return form16() ? 5 : 1;
    if (widths_match<MxCfgOct1>(ke)) return form16() ? 6 : 2;
    if (widths_match<MxCfgOct2>(ke)) return 3;
    if (widths_match<MxCfgOct3>(ke)) return 4;
    return 0;
}

int mx_strip_rows(int cfg) { return cfg == 5 || cfg == 6 ? 16 : 32; }

size_t mx_taps_bytes(int cfg) {
#if VSLAM_MX16
    if (cfg == 5) return sizeof(Mx16Taps<Mx16CfgOct0>);
    if (cfg == 6) return sizeof(Mx16Taps<Mx16CfgOct1>);
#endif
    return cfg == 1 ? sizeof(MxTaps<MxCfgOct0>) : cfg == 2 ? sizeof(MxTaps<MxCfgOct1>) : cfg == 3 ? sizeof(MxTaps<MxCfgOct2>) : cfg == 4 ? sizeof(MxTaps<MxCfgOct3>) : 0;
}

bool mx_pack(int cfg, const uint16_t* const taps[6], void* host_table) {
    if (cfg == 1) return mx_pack_taps<MxCfgOct0>(taps, *static_cast<MxTaps<MxCfgOct0>*>(host_table));
    if (cfg == 2) return mx_pack_taps<MxCfgOct1>(taps, *static_cast<MxTaps<MxCfgOct1>*>(host_table));
    if (cfg == 3) return mx_pack_taps<MxCfgOct2>(taps, *static_cast<MxTaps<MxCfgOct2>*>(host_table));
    if (cfg == 4) return mx_pack_taps<MxCfgOct3>(taps, *static_cast<MxTaps<MxCfgOct3>*>(host_table));
#if VSLAM_MX16
    if (cfg == 5) return mx16_pack_taps<Mx16CfgOct0>(taps, *static_cast<Mx16Taps<Mx16CfgOct0>*>(host_table));
    if (cfg == 6) return mx16_pack_taps<Mx16CfgOct1>(taps, *static_cast<Mx16Taps<Mx16CfgOct1>*>(host_table));
#endif
    return false;
}

bool mx_scan_supported(int cfg) { return cfg == 1 || cfg == 2 || cfg == 5 || cfg == 6; }  // the configurations with a D buffer in LDS (MxCfg::DBUF)
bool mx_up2_supported(int cfg) { return cfg == 1 || cfg == 5; }                            // the reference's pyramid upsamples in front of octave 0 only

#endif  // !VSLAM_MX_OCT0_TU

template <class CFG>
static hipError_t prepare() {
    if (CFG::DBUF) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx<CFG, CFG::DBUF != 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    if constexpr (std::is_same<CFG, MxCfgOct0>::value) {  // (constexpr: the other configurations' instantiations must not pull octave 0's kernels into their translation unit)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx<MxCfgOct0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx<MxCfgOct0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx<CFG, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
}

#if !defined(VSLAM_MX_OCT0_TU) && VSLAM_MX16
template <class CFG, bool UP2_TOO>
static hipError_t prepare16() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx16<CFG, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx16<CFG, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    if constexpr (UP2_TOO) {
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx16<CFG, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_octave_mx16<CFG, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    }
    return e;
}

template <class CFG, bool UP2_OK>
static hipError_t launch16(hipStream_t stream, const void* d_table, const uint8_t* base, size_t bframe, uint8_t* oct_out, size_t pframe, int rows, int cols, int pitch,
                           int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch, const MxScan* scan, int up2_step) {
    const dim3 grid((cols + CFG::TW - 1) / CFG::TW, (rows + CFG::TH - 1) / CFG::TH, nf);
    const Mx16Taps<CFG>* tp = static_cast<const Mx16Taps<CFG>*>(d_table);
    MxExtArgs ext{};
    if (scan) ext = MxExtArgs{scan->sitemap, scan->mframe, scan->lat_rows, scan->lat_cols, scan->mpitch, scan->min_contrast, scan->colmap, scan->cframe, scan->nseams};
    if (up2_step > 0) {
        if constexpr (!UP2_OK) {
            return hipErrorInvalidValue;
        } else {
            if ((rows & 1) || (cols & 1)) return hipErrorInvalidValue;
            if (scan)
                hipLaunchKernelGGL((k_pyr_octave_mx16<CFG, true, true>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch, tp,
                                   next_base, nframe, nrows, ncols, npitch, ext, up2_step);
            else
                hipLaunchKernelGGL((k_pyr_octave_mx16<CFG, false, true>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch, tp,
                                   next_base, nframe, nrows, ncols, npitch, ext, up2_step);
            return hipGetLastError();
        }
    }
    if (scan)
        hipLaunchKernelGGL((k_pyr_octave_mx16<CFG, true, false>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch, tp, next_base,
                           nframe, nrows, ncols, npitch, ext, 0);
    else
        hipLaunchKernelGGL((k_pyr_octave_mx16<CFG, false, false>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch, tp, next_base,
                           nframe, nrows, ncols, npitch, ext, 0);
    return hipGetLastError();
}
#endif

#ifdef VSLAM_MX_OCT0_TU
hipError_t mx_prepare_oct0() { return prepare<MxCfgOct0>(); }
#else
hipError_t mx_prepare(int cfg) {
#if VSLAM_MX16
    if (cfg == 5) return prepare16<Mx16CfgOct0, true>();
    if (cfg == 6) return prepare16<Mx16CfgOct1, false>();
#endif
    return cfg == 1 ? mx_prepare_oct0() : cfg == 2 ? prepare<MxCfgOct1>() : cfg == 3 ? prepare<MxCfgOct2>() : cfg == 4 ? prepare<MxCfgOct3>() : hipErrorInvalidValue;
}
#endif

template <class CFG>
static hipError_t launch(hipStream_t stream, const void* d_table, const uint8_t* base, size_t bframe, uint8_t* oct_out, size_t pframe, int rows,
                         int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch, const MxScan* scan, int up2_step) {
    const dim3 grid((cols + CFG::TW - 1) / CFG::TW, (rows + CFG::TH - 1) / CFG::TH, nf);
    MxExtArgs ext{};
    if (up2_step > 0) {
      if constexpr (!std::is_same<CFG, MxCfgOct0>::value) {
        return hipErrorInvalidValue;
      } else {
        if ((rows & 1) || (cols & 1)) return hipErrorInvalidValue;
        if (scan) {
            ext = MxExtArgs{scan->sitemap, scan->mframe, scan->lat_rows, scan->lat_cols, scan->mpitch, scan->min_contrast, scan->colmap, scan->cframe, scan->nseams};
            hipLaunchKernelGGL((k_pyr_octave_mx<MxCfgOct0, true, true>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch,
                               static_cast<const MxTaps<MxCfgOct0>*>(d_table), next_base, nframe, nrows, ncols, npitch, ext, up2_step);
        } else {
            hipLaunchKernelGGL((k_pyr_octave_mx<MxCfgOct0, false, true>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch,
                               static_cast<const MxTaps<MxCfgOct0>*>(d_table), next_base, nframe, nrows, ncols, npitch, ext, up2_step);
        }
        return hipGetLastError();
      }
    }
    if (scan && CFG::DBUF) {
        ext = MxExtArgs{scan->sitemap, scan->mframe, scan->lat_rows, scan->lat_cols, scan->mpitch, scan->min_contrast, scan->colmap, scan->cframe, scan->nseams};
        hipLaunchKernelGGL((k_pyr_octave_mx<CFG, CFG::DBUF != 0, false>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch,
                           static_cast<const MxTaps<CFG>*>(d_table), next_base, nframe, nrows, ncols, npitch, ext, 0);
    } else {
        if (scan) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_pyr_octave_mx<CFG, false, false>), grid, dim3(CFG::NT), CFG::LDS_BYTES, stream, base, bframe, oct_out, pframe, rows, cols, pitch,
                           static_cast<const MxTaps<CFG>*>(d_table), next_base, nframe, nrows, ncols, npitch, ext, 0);
    }
    return hipGetLastError();
}

#ifdef VSLAM_MX_OCT0_TU
hipError_t mx_launch_oct0(hipStream_t stream, const void* d_table, const uint8_t* base, size_t bframe, uint8_t* oct_out, size_t pframe, int rows,
                          int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch, const MxScan* scan, int up2_step) {
    return launch<MxCfgOct0>(stream, d_table, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
}
#else
hipError_t mx_launch_pack(hipStream_t stream, const MxScan& scan, int rows, int wpr, int nf, unsigned long long* bits, unsigned long long* lflags,
                          size_t bframe, int strip_rows) {
    // strips of 32 (or, diagnostics build, 16) rows x 128 columns in every configuration with a fused scan
    static_assert(MxCfgOct0::SW == 128 && MxCfgOct1::SW == 128, "k_extrema_pack's straddle rule, mx_seams");
    hipLaunchKernelGGL(k_extrema_pack, dim3((8 * wpr + 255) / 256, (scan.lat_rows + MX_PACK_ROWS - 1) / MX_PACK_ROWS, nf), dim3(256), 0, stream, scan.sitemap,
                       scan.mframe, scan.mpitch, scan.colmap, scan.cframe, scan.nseams, rows, scan.lat_rows, scan.lat_cols, wpr, strip_rows, 128, scan.min_contrast, bits,
                       lflags, bframe);
    return hipGetLastError();
}

hipError_t mx_launch(int cfg, hipStream_t stream, const void* d_table, const uint8_t* base, size_t bframe, uint8_t* oct_out, size_t pframe,
                     int rows, int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch, const MxScan* scan, int up2_step) {
    if (cfg == 1) return mx_launch_oct0(stream, d_table, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
    if (cfg == 2) return launch<MxCfgOct1>(stream, d_table, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
    if (cfg == 3) return launch<MxCfgOct2>(stream, d_table, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
    if (cfg == 4) return launch<MxCfgOct3>(stream, d_table, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
#if VSLAM_MX16
    if (cfg == 5) return launch16<Mx16CfgOct0, true>(stream, d_table, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
    if (cfg == 6) return launch16<Mx16CfgOct1, false>(stream, d_table, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
#endif
    return hipErrorInvalidValue;
}
#endif  // VSLAM_MX_OCT0_TU

}  // namespace vslam
