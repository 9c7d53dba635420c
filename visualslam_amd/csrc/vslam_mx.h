// Interface between the C-ABI translation unit (vslam_hip.hip) and the matrix-core octave kernels
// (vslam_mx.hip, compiled on its own with -mllvm -amdgpu-mfma-vgpr-form: the accumulators then live in
// VGPRs and the epilogue reads them without v_accvgpr_read).  OPT-IN path, see kernels_pyramid_mx.hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace vslam {

// Which matrix-core configuration runs an octave with these zero-trimmed kernel widths: 0 = none.
int mx_config_for(const int ke[6]);
// Rows of a wave's strip in configuration `cfg` (32: the 32 x 32 x 32 kernels; 16: the diagnostics build's 16 x 16 x 64 kernels): the
// lattice rows a fused scan leaves to k_extrema_w3 are the multiples of it.
int mx_strip_rows(int cfg);
// 32 (default) or - diagnostics build only - 16: which MFMA shape mx_config_for hands out for octaves 0-1 (process-wide; VSLAM_MX_FORM)
void mx_set_form(int form);
// Bytes of the device table of configuration `cfg`; mx_pack fills a host copy (false: a tap does not fit a signed byte).
size_t mx_taps_bytes(int cfg);
bool mx_pack(int cfg, const uint16_t* const taps[6], void* host_table);
// The lattice scan fused into the octave kernel (kernels_pyramid_mx.hip.h): the kernel leaves one byte per lattice site in
// `sitemap` ([frames][lat_rows][mpitch]) and mx_launch_pack turns them - and the few sites the octave kernel cannot own -
// into the candidate / list-flag words.  Window 3 only, candidates + min_contrast list (no localization).
struct MxScan {
    uint8_t* sitemap;
    size_t mframe;
    int lat_rows, lat_cols, mpitch, min_contrast;
    uint8_t* colmap;  // [frames][5][nseams][rows][2]: the DoG columns on either side of every seam (image column 384 s)
    size_t cframe;
    int nseams;
};
// Seams of an octave `cols` wide: lattice columns whose window straddles two 128-column strips (3b = 384 s <= cols - 2).
inline int mx_seams(int cols) { return cols >= 2 ? (cols - 2) / 384 : 0; }
bool mx_scan_supported(int cfg);
// Raises the kernel's dynamic-LDS ceiling (once per device) and launches it on `stream`; scan = nullptr: no fused scan.
hipError_t mx_prepare(int cfg);
// up2_step > 0: `base` / `bframe` are the SOURCE frames (rows / 2 x cols / 2 pixels, `up2_step` bytes per row) and the octave's base is
// their 2x bilinear upsample, formed while each tile is staged (configuration 1 only: octave 0 of createPyramid).
hipError_t mx_launch(int cfg, hipStream_t stream, const void* d_table, const uint8_t* base, size_t bframe, uint8_t* oct_out, size_t pframe,
                     int rows, int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch,
                     const MxScan* scan, int up2_step = 0);
bool mx_up2_supported(int cfg);
// bits / lflags: word 0 of frame 0's octave (bits may be nullptr).
hipError_t mx_launch_pack(hipStream_t stream, const MxScan& scan, int rows, int wpr, int nf, unsigned long long* bits, unsigned long long* lflags,
                          size_t bframe, int strip_rows);

}  // namespace vslam
