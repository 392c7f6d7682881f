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

// GaussBlur, one lane per pixel of a 16x16 tile.  The tile and its halo of `rs` pixels (clamped to the image edge as
// the reference clamps its tap coordinates, main.cpp:19-24) are staged in LDS as three planes, so a tap costs three
// conflict-free LDS reads instead of three global loads; the weight of a tap is wave-uniform (scalar load).
// The taps are accumulated in the reference's order (rows outer, columns inner) with unfused multiply and add.
__global__ __launch_bounds__(256) void gauss_lds_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        const float *__restrict__ weights, int width, int height, int rs) {
    extern __shared__ float tile[];   // 3 planes of tw x tw
    const int tw = 16 + 2 * rs, plane = tw * tw;
    const int x0 = blockIdx.x * 16 - rs, y0 = blockIdx.y * 16 - rs;
    for (int idx = threadIdx.x; idx < plane; idx += 256) {
        const int ty = idx / tw, tx = idx - ty * tw;
        const int y = min(height - 1, max(0, y0 + ty)), x = min(width - 1, max(0, x0 + tx));
        const float *p = in + 3 * (static_cast<size_t>(y) * width + x);
        tile[idx] = p[0];
        tile[plane + idx] = p[1];
        tile[2 * plane + idx] = p[2];
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + lx, i = blockIdx.y * 16 + ly;
    if (j >= width || i >= height) return;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, wsum = 0.0f;
    const int side = 2 * rs + 1;
    for (int dy = 0; dy < side; ++dy) {
        const float *row = tile + (ly + dy) * tw + lx;
        const float *wrow = weights + dy * side;
#pragma unroll 8
        for (int dx = 0; dx < side; ++dx) {
            const float w = wrow[dx];
            v0 += row[dx] * w;
            v1 += row[plane + dx] * w;
            v2 += row[2 * plane + dx] * w;
            wsum += w;
        }
    }
    float *o = out + 3 * (static_cast<size_t>(i) * width + j);
    o[0] = __builtin_roundf(v0 / wsum);   // glm::round -> std::round: half away from zero
    o[1] = __builtin_roundf(v1 / wsum);
    o[2] = __builtin_roundf(v2 / wsum);
}

// The same from global memory, for radii whose halo tile would not fit in LDS.
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
    o[0] = __builtin_roundf(v0 / wsum);
    o[1] = __builtin_roundf(v1 / wsum);
    o[2] = __builtin_roundf(v2 / wsum);
}

// MedianFilter for the small windows (-MEDIAN 1..3): element K = WS*WS/2 of the sorted (2 WS + 1)^2 window.  The K+1
// smallest values seen so far live in registers; a new value is passed down the list with one min/max pair per slot,
// which leaves the K+1 smallest of everything seen, sorted -- the same order statistic the reference reads from its
// sorted vector (main.cpp:62-74).  No NaNs can reach this point (the inputs are tonemapped colours).
template <int WS>
__global__ __launch_bounds__(256) void median_small_kernel(const float *__restrict__ in, float *__restrict__ out, int width, int height) {
    constexpr int K = WS * WS / 2;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15);
    const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;
    float small[3][K + 1];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s <= K; ++s) small[c][s] = __builtin_inff();
#pragma unroll
    for (int wx = -WS; wx <= WS; ++wx) {
        const int i = max(min(wx + x, width - 1), 0);
#pragma unroll
        for (int wy = -WS; wy <= WS; ++wy) {
            const int j = max(min(wy + y, height - 1), 0);
            const float *p = in + 3 * (static_cast<size_t>(j) * width + i);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = p[c];
#pragma unroll
                for (int s = 0; s <= K; ++s) {
                    const float lo = __builtin_fminf(small[c][s], v);
                    v = __builtin_fmaxf(small[c][s], v);
                    small[c][s] = lo;
                }
            }
        }
    }
    float *o = out + 3 * (static_cast<size_t>(y) * width + x);
    o[0] = small[0][K];
    o[1] = small[1][K];
    o[2] = small[2][K];
}

// Any window up to 11: keep the K+1 smallest values seen so far, sorted by insertion.
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
    const size_t lds = static_cast<size_t>(16 + 2 * rs) * (16 + 2 * rs) * 3 * sizeof(float);
    if (lds <= kGaussLdsBytes)
        hipLaunchKernelGGL(gauss_lds_kernel, grid, dim3(256), lds, stream, d_in, d_out, d_weights, width, height, rs);
    else
        hipLaunchKernelGGL(gauss_kernel, grid, dim3(256), 0, stream, d_in, d_out, d_weights, width, height, rs);
    return hipGetLastError();
}
hipError_t launch_median(const float *d_in, float *d_out, int width, int height, int ws, hipStream_t stream) {
    const dim3 grid((width + 15) / 16, (height + 15) / 16);
    switch (ws) {
    case 1: hipLaunchKernelGGL(median_small_kernel<1>, grid, dim3(256), 0, stream, d_in, d_out, width, height); break;
    case 2: hipLaunchKernelGGL(median_small_kernel<2>, grid, dim3(256), 0, stream, d_in, d_out, width, height); break;
    case 3: hipLaunchKernelGGL(median_small_kernel<3>, grid, dim3(256), 0, stream, d_in, d_out, width, height); break;
    default: hipLaunchKernelGGL(median_kernel, grid, dim3(256), 0, stream, d_in, d_out, width, height, ws);
    }
    return hipGetLastError();
}

}  // namespace pt
