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
// Bytes of the device table of configuration `cfg`; mx_pack fills a host copy (false: a tap does not fit a signed byte).
size_t mx_taps_bytes(int cfg);
bool mx_pack(int cfg, const uint16_t* const taps[6], void* host_table);
// Raises the kernel's dynamic-LDS ceiling (once per device) and launches it on `stream`.
hipError_t mx_prepare(int cfg);
hipError_t mx_launch(int cfg, hipStream_t stream, const void* d_table, const uint8_t* base, size_t bframe, uint8_t* oct_out, size_t pframe,
                     int rows, int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch);

}  // namespace vslam
