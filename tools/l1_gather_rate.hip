// Micro-benchmark: how many gather requests per clock does one compute unit's vector L1 serve?  (profiles/r03_ab_logs.txt ab63)
//   hipcc --offload-arch=gfx950 -O3 -o l1_gather_rate tools/l1_gather_rate.hip && ./l1_gather_rate
// 24 waves per CU walk a table of 64-byte nodes with data-dependent indices (like a tree walk) in six access patterns:
//   0  one 16-byte load per lane and step            3  a quad reads ONE node, each lane another quarter (coalesced 64 bytes)
//   1  four 16-byte loads of the lane's own node     4  one 4-byte load per lane and step
//   2  four lanes share a node, numbers by ds_bpermute   5  quads read four nodes in four instructions, node numbers by DPP
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64, 6) void k(const uint4 *__restrict__ tab, int n_nodes, int iters, uint32_t *out) {
    const int lane = threadIdx.x;
    uint32_t idx = (blockIdx.x * 977u + lane * 131u) % n_nodes;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // one 16-byte load per lane, every lane another node
            const uint4 a = tab[idx * 4u];
            acc += a.x + a.w;
        } else if (MODE == 1) {   // the node pattern: four 16-byte loads of the lane's own node (64 bytes)
            const uint4 a = tab[idx * 4u], b = tab[idx * 4u + 1], c = tab[idx * 4u + 2], d = tab[idx * 4u + 3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 3) {   // a quad reads ONE node, each lane another quarter (coalesced 64 bytes), one instruction
            const uint32_t node = (idx & ~3u) == 0xFFFFFFFFu ? 0u : ((blockIdx.x * 977u + (lane >> 2) * 131u + acc) % n_nodes);
            const uint4 a = tab[node * 4u + (lane & 3)];
            acc = (acc + a.x) & 0xFFFFu;
        } else if (MODE == 4) {   // one 4-byte load per lane, every lane another node
            const uint32_t *t = reinterpret_cast<const uint32_t *>(tab);
            acc += t[idx * 16u];
        } else if (MODE == 5) {   // a quad reads four nodes in four instructions, quarter (lane & 3) of each: node numbers by DPP
            uint32_t s = 0;
            const uint32_t n0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xF, 0xF, true), n1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xF, 0xF, true);
            const uint32_t n2 = __builtin_amdgcn_mov_dpp(idx, 0xAA, 0xF, 0xF, true), n3 = __builtin_amdgcn_mov_dpp(idx, 0xFF, 0xF, 0xF, true);
            const uint4 a0 = tab[n0 * 4u + (lane & 3)], a1 = tab[n1 * 4u + (lane & 3)], a2 = tab[n2 * 4u + (lane & 3)], a3 = tab[n3 * 4u + (lane & 3)];
            s = a0.x + a1.y + a2.z + a3.w;
            acc += s;
        } else {                  // four lanes share a node: each loads another quarter (one line per quad and instruction), 4 instructions
            uint32_t s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t node = __shfl(idx, (lane & ~3) + j);
                const uint4 a = tab[node * 4u + (lane & 3)];
                s += a.x + a.w;
            }
            acc += s;
        }
        idx = (idx * 5u + acc % 7u + 1u) % n_nodes;   // next node depends on the data: like a tree walk
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main() {
    const int n_nodes_list[] = {256, 4096, 65536};
    for (int n_nodes : n_nodes_list) {
        std::vector<uint4> h(n_nodes * 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = {uint32_t(i * 2654435761u), uint32_t(i), uint32_t(i * 40503u), uint32_t(i ^ 0x5555u)};
        uint4 *d; uint32_t *o;
        CHECK(hipMalloc(&d, h.size() * sizeof(uint4)));
        CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
        const int blocks = 256 * 24 * 4, iters = 2000;
        CHECK(hipMalloc(&o, blocks * 64 * 4));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, n_nodes, iters, o);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, n_nodes, iters, o);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, n_nodes, iters, o);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, n_nodes, iters, o);
                else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d, n_nodes, iters, o);
                else hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, d, n_nodes, iters, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double lane_loads = double(blocks) * 64 * iters * ((mode == 0 || mode == 3 || mode == 4) ? 1 : 4);
            const double cu_cycles = best * 1e-3 * 2.4e9;   // per CU
            printf("table %6d nodes (%5d KB)  mode %d: %.3f ms  lane-loads of 16 B per CU-clock %.3f   (node visits per CU-clock %.3f)\n", n_nodes, n_nodes * 64 / 1024, mode,
                   best, lane_loads / 256 / cu_cycles, double(blocks) * 64 * iters / 256 / cu_cycles);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
