// Post filters of the reference on the GPU: GaussBlur (main.cpp:11-33) and MedianFilter (main.cpp:49-80).
// They act on the tonemapped float image (main.cpp:179-182), one lane per pixel; results are identical to the CPU
// loops: the Gaussian taps are accumulated in the reference's order with unfused float arithmetic, the weights come
// from the host (libm expf, as the reference computes them), and the "median" is the reference's order statistic.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pt_filters.hpp"

#pragma clang fp contract(off)

namespace pt {

namespace {

__global__ __launch_bounds__(256) void gauss_kernel(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ weights,
                                                    int width, int height, int rs) {
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);   // x
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4);   // y
    if (j >= width || i >= height) return;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, wsum = 0.0f;
    const int side = 2 * rs + 1;
    for (int iy = i - rs; iy <= i + rs; ++iy) {
        const int y = min(height - 1, max(0, iy));
        for (int ix = j - rs; ix <= j + rs; ++ix) {
            const int x = min(width - 1, max(0, ix));
            const float w = weights[(iy - i + rs) * side + (ix - j + rs)];
            const float *p = in + 3 * (static_cast<size_t>(y) * width + x);
            v0 += p[0] * w;
            v1 += p[1] * w;
            v2 += p[2] * w;
            wsum += w;
        }
    }
    float *o = out + 3 * (static_cast<size_t>(i) * width + j);
    o[0] = __builtin_roundf(v0 / wsum);   // glm::round -> std::round: half away from zero
    o[1] = __builtin_roundf(v1 / wsum);
    o[2] = __builtin_roundf(v2 / wsum);
}

// Element k = w*w/2 (0-based) of the sorted (2w+1)^2 window: keep the k+1 smallest values seen so far, sorted.
__global__ __launch_bounds__(256) void median_kernel(const float *__restrict__ in, float *__restrict__ out, int width, int height, int ws) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15);
    const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;
    const int k = ws * ws / 2;
    for (int c = 0; c < 3; ++c) {
        float small[kMedianMaxRank + 1];
        int have = 0;
        for (int wx = -ws; wx <= ws; ++wx) {
            const int i = max(min(wx + x, width - 1), 0);
            for (int wy = -ws; wy <= ws; ++wy) {
                const int j = max(min(wy + y, height - 1), 0);
                const float v = in[3 * (static_cast<size_t>(j) * width + i) + c];
                if (have <= k) {
                    int pos = have++;
                    while (pos > 0 && small[pos - 1] > v) { small[pos] = small[pos - 1]; --pos; }
                    small[pos] = v;
                } else if (v < small[k]) {
                    int pos = k;
                    while (pos > 0 && small[pos - 1] > v) { small[pos] = small[pos - 1]; --pos; }
                    small[pos] = v;
                }
            }
        }
        out[3 * (static_cast<size_t>(y) * width + x) + c] = small[k];
    }
}

}  // namespace

hipError_t launch_gauss(const float *d_in, float *d_out, const float *d_weights, int width, int height, int rs, hipStream_t stream) {
    const dim3 grid((width + 15) / 16, (height + 15) / 16);
    hipLaunchKernelGGL(gauss_kernel, grid, dim3(256), 0, stream, d_in, d_out, d_weights, width, height, rs);
    return hipGetLastError();
}
hipError_t launch_median(const float *d_in, float *d_out, int width, int height, int ws, hipStream_t stream) {
    const dim3 grid((width + 15) / 16, (height + 15) / 16);
    hipLaunchKernelGGL(median_kernel, grid, dim3(256), 0, stream, d_in, d_out, width, height, ws);
    return hipGetLastError();
}

}  // namespace pt
