// How many one-wave workgroups of a kernel a compute unit holds, by LDS bytes per workgroup: the runtime's occupancy calculator
// (hipOccupancyMaxActiveBlocksPerMultiprocessor) AND a measurement -- a kernel in which every workgroup spins until all workgroups of a
// grid of exactly cu_count * n have started; it only terminates if n workgroups per CU are co-resident (bounded by a timeout counter).
//   hipcc --offload-arch=gfx950 -O2 -o lds_granule_probe tools/lds_granule_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void probe(unsigned *arrived, unsigned total, unsigned long long spin_limit, unsigned *timed_out) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;   // (touch the allocation)
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(arrived, 1u);
        unsigned long long spins = 0;
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > spin_limit) { atomicAdd(timed_out, 1u); break; }
        }
    }
    __syncthreads();
    if (lds[threadIdx.x] == 0xFFFFFFFFu) arrived[1] = 1;
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    std::printf("# %s, %d CUs, %zu B of LDS per workgroup at most\n", p.gcnArchName, p.multiProcessorCount, static_cast<size_t>(p.sharedMemPerBlock));
    unsigned *d = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d), 16) != hipSuccess) return 1;
    std::printf("# LDS bytes per one-wave workgroup | runtime's occupancy (workgroups per CU) | largest n for which cu_count * n workgroups were all running at once\n");
    const int sizes[] = {5120, 6400, 6401, 6656, 7168, 7376, 7680, 7681, 7888, 8192, 8960, 8961, 10240, 10241};
    for (int bytes : sizes) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 64, static_cast<size_t>(bytes)) != hipSuccess) occ = -1;
        int measured = 0;
        for (int n = 12; n <= 32; ++n) {   // (a one-wave workgroup with no registers to speak of: LDS is the only limit below 32 per CU)
            unsigned zero[4] = {0, 0, 0, 0};
            hipMemcpy(d, zero, sizeof zero, hipMemcpyHostToDevice);
            const unsigned total = static_cast<unsigned>(p.multiProcessorCount) * n;
            hipLaunchKernelGGL(probe, dim3(total), dim3(64), static_cast<size_t>(bytes), 0, d, total, 200000ull, d + 2);
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            unsigned out[4];
            hipMemcpy(out, d, sizeof out, hipMemcpyDeviceToHost);
            if (out[2] == 0) measured = n; else break;
        }
        std::printf("%6d  %3d  %3d\n", bytes, occ, measured);
    }
    hipFree(d);
    return 0;
}
