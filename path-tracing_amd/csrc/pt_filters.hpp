// Launch interface of the post-filter kernels (pt_filters.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace pt {

constexpr size_t kGaussLdsBytes = 48 * 1024;   // halo tile of the LDS Gaussian; larger radii (-GAUSS > 9) read global memory
constexpr int kMedianMaxRank = 63;   // window_size * window_size / 2 must not exceed this (window_size <= 11)

hipError_t launch_gauss(const float *d_in, float *d_out, const float *d_weights, int width, int height, int rs, hipStream_t stream);
hipError_t launch_median(const float *d_in, float *d_out, int width, int height, int ws, hipStream_t stream);

}  // namespace pt
