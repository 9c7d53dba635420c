// Octave 0's matrix-core kernel in a translation unit of its own (see vslam_mx.hip: compiled with
// -amdgpu-sched-strategy=max-ilp, which pays for this configuration only).
#define VSLAM_MX_OCT0_TU 1
#include "vslam_mx.hip"
