// Internal declarations shared by the host-math TU and the HIP TU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/vslam.h"

#define VSLAM_MAX_KSIZE 4095

namespace vslam {
int gauss_ksize_u8(double sigma);
bool gauss_taps_q8(int n, double sigma, uint16_t* taps);
bool gauss_taps_q8_trimmed(int n, double sigma, std::vector<uint16_t>& out);
int gauss_ksize_f32(double sigma);
bool gauss_kernel_f32(int n, double sigma, std::vector<float>& out);
double sigma_at(double sigma0, int octave, int level);
int auto_num_octaves(int rows, int cols);
void half_size(int rows, int cols, int* r, int* c);
void extrema_lattice(int rows, int cols, int window, int* lr, int* lc);
int make_layout(const vslam_params* p, vslam_batch_layout* L);
}  // namespace vslam
