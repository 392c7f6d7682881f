// Where a fresh process spends its time before and after its first kernel: one line per HIP call, seconds.
//   hipcc -O2 -o hip_startup_probe tools/hip_startup_probe.cpp && ./hip_startup_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(float *p) { p[threadIdx.x] = 1.0f; }
int main() {
    using clk = std::chrono::steady_clock;
    auto t = clk::now();
    auto lap = [&](const char *what) {
        const auto n = clk::now();
        std::printf("%-28s %.4f\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    };
    int n = 0;
    hipGetDeviceCount(&n); lap("hipGetDeviceCount");
    hipSetDevice(0); lap("hipSetDevice");
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0); lap("hipDeviceGetAttribute");
    float *d = nullptr;
    hipMalloc(reinterpret_cast<void **>(&d), 64); lap("hipMalloc 64 B");
    float *big = nullptr;
    hipMalloc(reinterpret_cast<void **>(&big), 58u << 20); lap("hipMalloc 58 MB");
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking); lap("hipStreamCreate");
    hipMemsetAsync(big, 0, 58u << 20, s); lap("hipMemsetAsync enqueue");
    hipStreamSynchronize(s); lap("... sync");
    hipMemsetAsync(big, 0, 58u << 20, s); hipStreamSynchronize(s); lap("second memset + sync");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d); lap("first kernel enqueue");
    hipStreamSynchronize(s); lap("... sync");
    void *h = nullptr;
    hipHostMalloc(&h, 116u << 20, hipHostMallocDefault); lap("hipHostMalloc 116 MB");
    hipMemcpy(h, big, 58u << 20, hipMemcpyDeviceToHost); lap("hipMemcpy D2H 58 MB");
    hipEvent_t e;
    hipEventCreate(&e); lap("hipEventCreate");
    std::printf("devices %d, CUs %d\n", n, cus);
    return 0;
}
