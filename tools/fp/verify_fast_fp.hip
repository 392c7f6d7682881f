// Exhaustive check of the lean correctly-rounded float sequences used by the integrator (pt_fastfp.hpp) against IEEE
// results, over EVERY float in the range they are used for.  Reference values are computed in double precision and
// rounded once: for sqrt and for a quotient of floats that is the correctly rounded float result (53 >= 2*24 + 2).
//   hipcc -O2 -ffp-contract=off --offload-arch=gfx950 -I path-tracing_amd/csrc tools/fp/verify_fast_fp.hip -o verify_fast_fp && ./verify_fast_fp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "pt_fastfp.hpp"

__global__ void check(uint32_t first, uint32_t count, unsigned long long *bad, uint32_t *example) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t bits = first + i;
    const float x = __uint_as_float(bits);
    const float s_ref = static_cast<float>(sqrt(static_cast<double>(x)));
    const float r_ref = static_cast<float>(1.0 / static_cast<double>(x));
    const float s = pt::sqrt_rn_normal(x), r = pt::rcp_rn_normal(x);
    if (__float_as_uint(s) != __float_as_uint(s_ref)) { if (atomicAdd(&bad[0], 1ull) == 0) example[0] = bits; }
    if (__float_as_uint(r) != __float_as_uint(r_ref)) { if (atomicAdd(&bad[1], 1ull) == 0) example[1] = bits; }
}

int main() {
    unsigned long long *bad; uint32_t *ex;
    if (hipMalloc(&bad, 16) != hipSuccess || hipMalloc(&ex, 8) != hipSuccess || hipMemset(bad, 0, 16) != hipSuccess || hipMemset(ex, 0, 8) != hipSuccess) return 2;
    // every positive float with exponent in [kFastExpLo, kFastExpHi]
    const uint32_t lo = static_cast<uint32_t>(pt::kFastExpLo + 127) << 23, hi = (static_cast<uint32_t>(pt::kFastExpHi + 127) << 23) | 0x7FFFFFu;
    const unsigned long long total = static_cast<unsigned long long>(hi) - lo + 1;
    for (unsigned long long done = 0; done < total;) {
        const uint32_t n = static_cast<uint32_t>(total - done > (1ull << 28) ? (1ull << 28) : total - done);
        hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, 0, static_cast<uint32_t>(lo + done), n, bad, ex);
        done += n;
    }
    unsigned long long h[2]; uint32_t e[2];
    if (hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(e, ex, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    std::printf("checked %llu floats in [2^%d, 2^%d): sqrt mismatches %llu (first 0x%08x), rcp mismatches %llu (first 0x%08x)\n", total,
                pt::kFastExpLo, pt::kFastExpHi + 1, h[0], e[0], h[1], e[1]);
    return (h[0] || h[1]) ? 1 : 0;
}
