// Issue cost of single gfx950 instructions, measured the way the integrator meets them: several waves per SIMD, each
// running a long straight-line run of ONE instruction (four independent destination registers, so a wave's own
// dependencies do not matter once a few waves share the SIMD).  Prints SIMD cycles per wave-instruction at 1, 2 and 6
// waves per SIMD, absolute (at the clock rocm-smi reports under load, given with --mhz) and relative to v_add_u32.
//
//   make -C path-tracing_amd/csrc issue-cost && path-tracing_amd/lib/tools/issue_cost [--mhz 2400]
//
// tools/asm_profile.py weights the instruction profile with this table ("pipe cycles"); profiles/r02_issue_cost.txt holds
// a run and says what the table explained about the integrator and what it did not.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

#define REP4(s) s s s s
#define REP16(s) REP4(REP4(s))
#define REP64(s) REP4(REP16(s))

// One kernel per instruction: `iters` times 64 groups of 4 instructions (256 per iteration).
#define OP_KERNEL(name, body4, clobbers...)                                                               \
    __global__ __launch_bounds__(256) void k_##name(uint32_t *out, int iters) {                           \
        uint32_t a = threadIdx.x + 1, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3, x = a ^ 0x3f800000u, y = 0x3f900000u + a; \
        double e = 1.0 + a, f = 2.0 + a, g = 3.0 + a, h = 4.0 + a;                                        \
        __shared__ uint32_t lds[1024];                                                                    \
        lds[threadIdx.x] = a;                                                                             \
        uint32_t la = threadIdx.x * 4;                                                                    \
        for (int i = 0; i < iters; ++i) {                                                                 \
            asm volatile(REP64(body4)                                                                     \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)         \
                         : "v"(x), "v"(y), "v"(la)                                                        \
                         : clobbers);                                                                     \
        }                                                                                                 \
        if (a + b + c + d == 0x12345 && e + f + g + h == 1.5) out[0] = a;                                 \
    }

// %0..%3 = a..d (32-bit), %4..%7 = e..h (64-bit), %8 = x, %9 = y, %10 = LDS address
OP_KERNEL(v_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n", "memory")
OP_KERNEL(v_mov_b32, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n", "memory")
OP_KERNEL(v_mul_f32, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n", "memory")
OP_KERNEL(v_fma_f32, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_fmac_f32, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n", "memory")
OP_KERNEL(v_pk_fma_f32, "v_pk_fma_f32 %4, %4, %5, %6\n v_pk_fma_f32 %5, %5, %6, %7\n v_pk_fma_f32 %6, %6, %7, %4\n v_pk_fma_f32 %7, %7, %4, %5\n", "memory")
OP_KERNEL(v_pk_mul_f32, "v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %6\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %4\n", "memory")
OP_KERNEL(v_fma_f64, "v_fma_f64 %4, %4, %5, %6\n v_fma_f64 %5, %5, %6, %7\n v_fma_f64 %6, %6, %7, %4\n v_fma_f64 %7, %7, %4, %5\n", "memory")
OP_KERNEL(v_mul_f64, "v_mul_f64 %4, %4, %5\n v_mul_f64 %5, %5, %6\n v_mul_f64 %6, %6, %7\n v_mul_f64 %7, %7, %4\n", "memory")
OP_KERNEL(v_add_f64, "v_add_f64 %4, %4, %5\n v_add_f64 %5, %5, %6\n v_add_f64 %6, %6, %7\n v_add_f64 %7, %7, %4\n", "memory")
OP_KERNEL(v_rcp_f32, "v_rcp_f32 %0, %8\n v_rcp_f32 %1, %8\n v_rcp_f32 %2, %8\n v_rcp_f32 %3, %8\n", "memory")
OP_KERNEL(v_rsq_f32, "v_rsq_f32 %0, %8\n v_rsq_f32 %1, %8\n v_rsq_f32 %2, %8\n v_rsq_f32 %3, %8\n", "memory")
OP_KERNEL(v_sqrt_f32, "v_sqrt_f32 %0, %8\n v_sqrt_f32 %1, %8\n v_sqrt_f32 %2, %8\n v_sqrt_f32 %3, %8\n", "memory")
OP_KERNEL(v_mad_u64_u32, "v_mad_u64_u32 %4, vcc, %8, %9, 0\n v_mad_u64_u32 %5, vcc, %8, %9, 0\n v_mad_u64_u32 %6, vcc, %8, %9, 0\n v_mad_u64_u32 %7, vcc, %8, %9, 0\n", "memory", "vcc")
OP_KERNEL(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n", "memory")
OP_KERNEL(v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n", "memory")
OP_KERNEL(v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n", "memory")
OP_KERNEL(v_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte0 %1, %8\n v_cvt_f32_ubyte0 %2, %8\n v_cvt_f32_ubyte0 %3, %8\n", "memory")
OP_KERNEL(v_cvt_f32_ubyte3, "v_cvt_f32_ubyte3 %0, %8\n v_cvt_f32_ubyte3 %1, %8\n v_cvt_f32_ubyte3 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n", "memory")
OP_KERNEL(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %1, %8\n v_cvt_f32_u32 %2, %8\n v_cvt_f32_u32 %3, %8\n", "memory")
OP_KERNEL(v_cvt_f64_f32, "v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %8\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %8\n", "memory")
OP_KERNEL(v_cvt_f32_f64, "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n", "memory")
OP_KERNEL(v_max3_f32, "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_min_f32, "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n", "memory")
OP_KERNEL(v_cndmask_vcc, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n", "memory")
OP_KERNEL(v_cndmask_sgpr, "v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %8, %9, s[20:21]\n v_cndmask_b32 %2, %8, %9, s[20:21]\n v_cndmask_b32 %3, %8, %9, s[20:21]\n", "memory")
OP_KERNEL(v_cmp_vcc, "v_cmp_gt_f32 vcc, %8, %9\n v_cmp_gt_f32 vcc, %8, %9\n v_cmp_gt_f32 vcc, %8, %9\n v_cmp_gt_f32 vcc, %8, %9\n", "memory", "vcc")
OP_KERNEL(v_cmp_sgpr, "v_cmp_gt_f32 s[20:21], %8, %9\n v_cmp_gt_f32 s[22:23], %8, %9\n v_cmp_gt_f32 s[24:25], %8, %9\n v_cmp_gt_f32 s[26:27], %8, %9\n", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
OP_KERNEL(v_bfe_u32, "v_bfe_u32 %0, %8, 3, 5\n v_bfe_u32 %1, %8, 3, 5\n v_bfe_u32 %2, %8, 3, 5\n v_bfe_u32 %3, %8, 3, 5\n", "memory")
OP_KERNEL(v_perm_b32, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_bitop3_b32, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n", "memory")
OP_KERNEL(v_lshlrev_b64, "v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7\n", "memory")
OP_KERNEL(v_readlane, "v_readlane_b32 s20, %8, 5\n v_readlane_b32 s21, %8, 6\n v_readlane_b32 s22, %8, 7\n v_readlane_b32 s23, %8, 8\n", "memory", "s20", "s21", "s22", "s23")
OP_KERNEL(v_readfirstlane, "v_readfirstlane_b32 s20, %8\n v_readfirstlane_b32 s21, %8\n v_readfirstlane_b32 s22, %8\n v_readfirstlane_b32 s23, %8\n", "memory", "s20", "s21", "s22", "s23")
OP_KERNEL(v_mov_dpp, "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n", "memory")
OP_KERNEL(v_add_dpp, "v_add_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n", "memory")
OP_KERNEL(v_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, 0\n v_mbcnt_lo_u32_b32 %1, -1, 0\n v_mbcnt_lo_u32_b32 %2, -1, 0\n v_mbcnt_lo_u32_b32 %3, -1, 0\n", "memory")
OP_KERNEL(s_add_u32, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n", "memory", "s20", "s21", "s22", "s23", "scc")
OP_KERNEL(s_and_b64, "s_and_b64 s[20:21], s[20:21], exec\n s_and_b64 s[22:23], s[22:23], exec\n s_and_b64 s[24:25], s[24:25], exec\n s_and_b64 s[26:27], s[26:27], exec\n", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc")
OP_KERNEL(s_mov_b32, "s_mov_b32 s20, 1\n s_mov_b32 s21, 1\n s_mov_b32 s22, 1\n s_mov_b32 s23, 1\n", "memory", "s20", "s21", "s22", "s23")
OP_KERNEL(s_bcnt1, "s_bcnt1_i32_b64 s20, exec\n s_bcnt1_i32_b64 s21, exec\n s_bcnt1_i32_b64 s22, exec\n s_bcnt1_i32_b64 s23, exec\n", "memory", "s20", "s21", "s22", "s23", "scc")
OP_KERNEL(s_nop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n", "memory")
OP_KERNEL(s_waitcnt, "s_waitcnt vmcnt(0) lgkmcnt(0)\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_waitcnt vmcnt(0) lgkmcnt(0)\n", "memory")
OP_KERNEL(ds_read_b32, "ds_read_b32 %0, %10\n ds_read_b32 %1, %10\n ds_read_b32 %2, %10\n ds_read_b32 %3, %10\n s_waitcnt lgkmcnt(0)\n", "memory")
OP_KERNEL(ds_write_b32, "ds_write_b32 %10, %8\n ds_write_b32 %10, %8\n ds_write_b32 %10, %8\n ds_write_b32 %10, %8\n s_waitcnt lgkmcnt(0)\n", "memory")
OP_KERNEL(ds_bpermute, "ds_bpermute_b32 %0, %10, %8\n ds_bpermute_b32 %1, %10, %8\n ds_bpermute_b32 %2, %10, %8\n ds_bpermute_b32 %3, %10, %8\n s_waitcnt lgkmcnt(0)\n", "memory")
OP_KERNEL(mix_valu_salu, "v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %8\n s_add_u32 s21, s21, 1\n", "memory", "s20", "s21", "scc")
OP_KERNEL(mix_fma_cvt, "v_fma_f32 %0, %0, %8, %9\n v_cvt_f32_ubyte0 %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_cvt_f32_ubyte0 %3, %8\n", "memory")

OP_KERNEL(v_sub_f32, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n", "memory")
OP_KERNEL(v_add_f32, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n", "memory")
OP_KERNEL(v_mul_f32_sgpr, "v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s20, %1\n v_mul_f32 %2, s20, %2\n v_mul_f32 %3, s20, %3\n", "memory")
OP_KERNEL(v_mul_f32_lit, "v_mul_f32 %0, 0x40490fdb, %0\n v_mul_f32 %1, 0x40490fdb, %1\n v_mul_f32 %2, 0x40490fdb, %2\n v_mul_f32 %3, 0x40490fdb, %3\n", "memory")
OP_KERNEL(v_mul_f32_abs, "v_mul_f32_e64 %0, |%0|, %8\n v_mul_f32_e64 %1, |%1|, %8\n v_mul_f32_e64 %2, |%2|, %8\n v_mul_f32_e64 %3, |%3|, %8\n", "memory")
OP_KERNEL(v_fma_f32_sgpr, "v_fma_f32 %0, %0, s20, %9\n v_fma_f32 %1, %1, s20, %9\n v_fma_f32 %2, %2, s20, %9\n v_fma_f32 %3, %3, s20, %9\n", "memory")
OP_KERNEL(v_fma_f32_neg, "v_fma_f32 %0, -%0, %8, %9\n v_fma_f32 %1, -%1, %8, %9\n v_fma_f32 %2, -%2, %8, %9\n v_fma_f32 %3, -%3, %8, %9\n", "memory")
OP_KERNEL(v_mov_b32_sgpr, "v_mov_b32 %0, s20\n v_mov_b32 %1, s20\n v_mov_b32 %2, s20\n v_mov_b32 %3, s20\n", "memory")
OP_KERNEL(v_mov_b32_lit, "v_mov_b32 %0, 0x12345678\n v_mov_b32 %1, 0x12345678\n v_mov_b32 %2, 0x12345678\n v_mov_b32 %3, 0x12345678\n", "memory")
OP_KERNEL(v_and_b32, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n", "memory")
OP_KERNEL(v_or_b32, "v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n", "memory")
OP_KERNEL(v_xor_b32, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n", "memory")
OP_KERNEL(v_not_b32, "v_not_b32 %0, %8\n v_not_b32 %1, %8\n v_not_b32 %2, %8\n v_not_b32 %3, %8\n", "memory")
OP_KERNEL(v_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n", "memory")
OP_KERNEL(v_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n", "memory")
OP_KERNEL(v_sub_u32, "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n", "memory")
OP_KERNEL(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n", "memory")
OP_KERNEL(v_add3_u32, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_or3_b32, "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %8\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %8\n", "memory")
OP_KERNEL(v_and_or_b32, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_add_u32_sdwa, "v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n", "memory")
OP_KERNEL(v_max_f32, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n", "memory")
OP_KERNEL(v_min3_f32, "v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_med3_f32, "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_min_u32, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n", "memory")
OP_KERNEL(v_div_scale_f32, "v_div_scale_f32 %0, vcc, %8, %9, %8\n v_div_scale_f32 %1, vcc, %8, %9, %8\n v_div_scale_f32 %2, vcc, %8, %9, %8\n v_div_scale_f32 %3, vcc, %8, %9, %8\n", "memory", "vcc")
OP_KERNEL(v_div_fmas_f32, "v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_div_fixup_f32, "v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_cmp_lt_f32_e64, "v_cmp_lt_f32_e64 s[20:21], %8, %9\n v_cmp_lt_f32_e64 s[20:21], %8, %9\n v_cmp_lt_f32_e64 s[20:21], %8, %9\n v_cmp_lt_f32_e64 s[20:21], %8, %9\n", "memory", "s20", "s21")
OP_KERNEL(v_cmp_ne_u32_vcc, "v_cmp_ne_u32 vcc, %8, %9\n v_cmp_ne_u32 vcc, %8, %9\n v_cmp_ne_u32 vcc, %8, %9\n v_cmp_ne_u32 vcc, %8, %9\n", "memory", "vcc")
OP_KERNEL(v_cmp_class_f32, "v_cmp_class_f32 vcc, %8, %9\n v_cmp_class_f32 vcc, %8, %9\n v_cmp_class_f32 vcc, %8, %9\n v_cmp_class_f32 vcc, %8, %9\n", "memory", "vcc")
OP_KERNEL(v_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %8, %9, vcc\n v_cndmask_b32_e64 %2, %8, %9, vcc\n v_cndmask_b32_e64 %3, %8, %9, vcc\n", "memory")
OP_KERNEL(v_writelane, "v_writelane_b32 %0, s20, 5\n v_writelane_b32 %1, s20, 5\n v_writelane_b32 %2, s20, 5\n v_writelane_b32 %3, s20, 5\n", "memory")
OP_KERNEL(v_bcnt_u32, "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n", "memory")
OP_KERNEL(v_ffbl_b32, "v_ffbl_b32 %0, %8\n v_ffbl_b32 %1, %8\n v_ffbl_b32 %2, %8\n v_ffbl_b32 %3, %8\n", "memory")
OP_KERNEL(v_mbcnt_hi, "v_mbcnt_hi_u32_b32 %0, -1, %0\n v_mbcnt_hi_u32_b32 %1, -1, %1\n v_mbcnt_hi_u32_b32 %2, -1, %2\n v_mbcnt_hi_u32_b32 %3, -1, %3\n", "memory")
OP_KERNEL(v_mul_lo_u16, "v_mul_lo_u16 %0, %0, %8\n v_mul_lo_u16 %1, %1, %8\n v_mul_lo_u16 %2, %2, %8\n v_mul_lo_u16 %3, %3, %8\n", "memory")
OP_KERNEL(v_cvt_u32_f32, "v_cvt_u32_f32 %0, %8\n v_cvt_u32_f32 %1, %8\n v_cvt_u32_f32 %2, %8\n v_cvt_u32_f32 %3, %8\n", "memory")
OP_KERNEL(v_lshl_add_u64, "v_lshl_add_u64 %4, %4, 3, %5\n v_lshl_add_u64 %4, %4, 3, %5\n v_lshl_add_u64 %4, %4, 3, %5\n v_lshl_add_u64 %4, %4, 3, %5\n", "memory")
OP_KERNEL(cmp_cndmask_vcc, "v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32 %0, %8, %9, vcc\n v_cmp_gt_f32 vcc, %9, %8\n v_cndmask_b32 %1, %8, %9, vcc\n", "memory", "vcc")
OP_KERNEL(cmp_cndmask_sgpr, "v_cmp_gt_f32 s[20:21], %8, %9\n v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cmp_gt_f32 s[22:23], %9, %8\n v_cndmask_b32 %1, %8, %9, s[22:23]\n", "memory", "s20", "s21", "s22", "s23")
OP_KERNEL(cndmask_vcc_after_salu, "s_mov_b64 vcc, exec\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n", "memory", "vcc")
OP_KERNEL(mix_fma_min, "v_fma_f32 %0, %0, %8, %9\n v_min_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_min_f32 %3, %3, %8\n", "memory")
OP_KERNEL(mix_fma_salu_lds, "v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %2, %2, %8, %9\n ds_read_b32 %3, %10\n", "memory", "s20", "scc")

OP_KERNEL(cmp_3cndmask_vcc, "v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n", "memory", "vcc")
OP_KERNEL(cmp_add_2cndmask_vcc, "v_cmp_gt_f32 vcc, %8, %9\n v_add_u32 %0, %0, %8\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n", "memory", "vcc")
OP_KERNEL(cmp_3cndmask_e64_vcc, "v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32_e64 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %8, %9, vcc\n v_cndmask_b32_e64 %2, %8, %9, vcc\n", "memory", "vcc")
OP_KERNEL(cmp_3cndmask_sgpr, "v_cmp_gt_f32 s[20:21], %8, %9\n v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %8, %9, s[20:21]\n v_cndmask_b32 %2, %8, %9, s[20:21]\n", "memory", "s20", "s21")
OP_KERNEL(addc_vcc, "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n", "memory", "vcc")
OP_KERNEL(div_fmas_vcc, "v_div_scale_f32 %0, vcc, %8, %9, %8\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_scale_f32 %2, vcc, %8, %9, %8\n v_div_fmas_f32 %3, %3, %8, %9\n", "memory", "vcc")
OP_KERNEL(cbranch_vccz, "v_cmp_gt_f32 vcc, %8, %9\n s_cbranch_vccz 0\n v_cmp_gt_f32 vcc, %8, %9\n s_cbranch_vccz 0\n", "memory", "vcc")
OP_KERNEL(s_and_saveexec, "s_and_saveexec_b64 s[20:21], exec\n s_mov_b64 exec, s[20:21]\n s_and_saveexec_b64 s[22:23], exec\n s_mov_b64 exec, s[22:23]\n", "memory", "s20", "s21", "s22", "s23", "scc")
OP_KERNEL(s_cbranch_scc, "s_cmp_eq_u32 s20, s21\n s_cbranch_scc1 0\n s_cmp_eq_u32 s20, s21\n s_cbranch_scc1 0\n", "memory", "scc")
OP_KERNEL(s_cbranch_execz, "s_cbranch_execz 0\n s_cbranch_execz 0\n s_cbranch_execz 0\n s_cbranch_execz 0\n", "memory")
OP_KERNEL(v_mul_f32_dep_sgprmix, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, s20, %1\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, s21, %3\n", "memory")
OP_KERNEL(v_lshlrev_b32_v, "v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n", "memory")
OP_KERNEL(v_add_u32_const, "v_add_u32 %0, 5, %0\n v_add_u32 %1, 5, %1\n v_add_u32 %2, 5, %2\n v_add_u32 %3, 5, %3\n", "memory")
OP_KERNEL(v_mul_f32_const, "v_mul_f32 %0, 2.0, %0\n v_mul_f32 %1, 2.0, %1\n v_mul_f32 %2, 2.0, %2\n v_mul_f32 %3, 2.0, %3\n", "memory")
OP_KERNEL(v_fma_f32_const, "v_fma_f32 %0, %0, 2.0, %9\n v_fma_f32 %1, %1, 2.0, %9\n v_fma_f32 %2, %2, 2.0, %9\n v_fma_f32 %3, %3, 2.0, %9\n", "memory")
OP_KERNEL(v_ashrrev_i32, "v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3\n", "memory")
OP_KERNEL(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_mul_legacy_f32, "v_mul_legacy_f32 %0, %0, %8\n v_mul_legacy_f32 %1, %1, %8\n v_mul_legacy_f32 %2, %2, %8\n v_mul_legacy_f32 %3, %3, %8\n", "memory")
OP_KERNEL(v_max_f32_via_fma_test, "v_fma_f32 %0, %0, %8, %9 clamp\n v_fma_f32 %1, %1, %8, %9 clamp\n v_fma_f32 %2, %2, %8, %9 clamp\n v_fma_f32 %3, %3, %8, %9 clamp\n", "memory")
OP_KERNEL(v_mul_f32_omod, "v_mul_f32_e64 %0, %0, %8 mul:2\n v_mul_f32_e64 %1, %1, %8 mul:2\n v_mul_f32_e64 %2, %2, %8 mul:2\n v_mul_f32_e64 %3, %3, %8 mul:2\n", "memory")

// round 4: the packed 16-bit forms a half-precision slab test of the box tree would use
OP_KERNEL(v_pk_fma_f16, "v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_pk_max_f16, "v_pk_max_f16 %0, %0, %8\n v_pk_max_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_max_f16 %3, %3, %8\n", "memory")
OP_KERNEL(v_pk_min_f16, "v_pk_min_f16 %0, %0, %8\n v_pk_min_f16 %1, %1, %8\n v_pk_min_f16 %2, %2, %8\n v_pk_min_f16 %3, %3, %8\n", "memory")
OP_KERNEL(v_pk_add_f16, "v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n", "memory")
OP_KERNEL(v_pk_mul_f16, "v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n", "memory")
OP_KERNEL(v_pk_mad_i16, "v_pk_mad_i16 %0, %0, %8, %9\n v_pk_mad_i16 %1, %1, %8, %9\n v_pk_mad_i16 %2, %2, %8, %9\n v_pk_mad_i16 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_pk_max_i16, "v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n", "memory")
OP_KERNEL(v_pk_sub_i16, "v_pk_sub_i16 %0, %0, %8\n v_pk_sub_i16 %1, %1, %8\n v_pk_sub_i16 %2, %2, %8\n v_pk_sub_i16 %3, %3, %8\n", "memory")
OP_KERNEL(v_pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 8, %8\n v_pk_lshrrev_b16 %1, 8, %8\n v_pk_lshrrev_b16 %2, 8, %8\n v_pk_lshrrev_b16 %3, 8, %8\n", "memory")
OP_KERNEL(v_and_b32_lit, "v_and_b32 %0, 0xff00ff, %8\n v_and_b32 %1, 0xff00ff, %8\n v_and_b32 %2, 0xff00ff, %8\n v_and_b32 %3, 0xff00ff, %8\n", "memory")
OP_KERNEL(v_cvt_pkrtz_f16_f32, "v_cvt_pkrtz_f16_f32 %0, %8, %9\n v_cvt_pkrtz_f16_f32 %1, %8, %9\n v_cvt_pkrtz_f16_f32 %2, %8, %9\n v_cvt_pkrtz_f16_f32 %3, %8, %9\n", "memory")
OP_KERNEL(v_max_f16, "v_max_f16 %0, %0, %8\n v_max_f16 %1, %1, %8\n v_max_f16 %2, %2, %8\n v_max_f16 %3, %3, %8\n", "memory")
OP_KERNEL(v_fma_f16, "v_fma_f16 %0, %0, %8, %9\n v_fma_f16 %1, %1, %8, %9\n v_fma_f16 %2, %2, %8, %9\n v_fma_f16 %3, %3, %8, %9\n", "memory")
OP_KERNEL(v_alignbyte_b32, "v_alignbyte_b32 %0, %8, %8, 1\n v_alignbyte_b32 %1, %8, %8, 1\n v_alignbyte_b32 %2, %8, %8, 1\n v_alignbyte_b32 %3, %8, %8, 1\n", "memory")
OP_KERNEL(v_lshrrev_b32_imm, "v_lshrrev_b32 %0, 8, %8\n v_lshrrev_b32 %1, 8, %8\n v_lshrrev_b32 %2, 8, %8\n v_lshrrev_b32 %3, 8, %8\n", "memory")
OP_KERNEL(mix_pkfma_and, "v_pk_fma_f16 %0, %0, %8, %9\n v_and_b32 %1, 0xff00ff, %8\n v_pk_fma_f16 %2, %2, %8, %9\n v_and_b32 %3, 0xff00ff, %8\n", "memory")

struct Entry { const char *name; void (*fn)(uint32_t *, int); int per_group; };
#define E(n) {#n, k_##n, 4}

int main(int argc, char **argv) {
    double mhz = 2400.0;
    const char *only = nullptr;   // --only a,b,c: just these (and v_add_u32, the reference)
    for (int i = 1; i + 1 < argc; ++i) {
        if (!std::strcmp(argv[i], "--mhz")) mhz = std::atof(argv[i + 1]);
        if (!std::strcmp(argv[i], "--only")) only = argv[i + 1];
    }
    const std::vector<Entry> ops = {
        E(v_add_u32), E(v_mov_b32), E(v_mul_f32), E(v_fma_f32), E(v_fmac_f32), E(v_pk_fma_f32), E(v_pk_mul_f32), E(v_fma_f64), E(v_mul_f64),
        E(v_add_f64), E(v_rcp_f32), E(v_rsq_f32), E(v_sqrt_f32), E(v_mad_u64_u32), E(v_mul_lo_u32), E(v_mul_hi_u32), E(v_mul_u32_u24),
        E(v_cvt_f32_ubyte0), E(v_cvt_f32_ubyte3), E(v_cvt_f32_u32), E(v_cvt_f64_f32), E(v_cvt_f32_f64), E(v_max3_f32), E(v_min_f32),
        E(v_cndmask_vcc), E(v_cndmask_sgpr), E(v_cmp_vcc), E(v_cmp_sgpr), E(v_bfe_u32), E(v_perm_b32), E(v_bitop3_b32), E(v_lshlrev_b64),
        E(v_readlane), E(v_readfirstlane), E(v_mov_dpp), E(v_add_dpp), E(v_mbcnt), E(s_add_u32), E(s_and_b64), E(s_mov_b32), E(s_bcnt1),
        E(s_nop), E(s_waitcnt), E(ds_read_b32), E(ds_write_b32), E(ds_bpermute), E(mix_valu_salu), E(mix_fma_cvt),
        E(v_sub_f32), E(v_add_f32), E(v_mul_f32_sgpr), E(v_mul_f32_lit), E(v_mul_f32_abs), E(v_fma_f32_sgpr), E(v_fma_f32_neg), E(v_mov_b32_sgpr), E(v_mov_b32_lit), E(v_and_b32), E(v_or_b32), E(v_xor_b32), E(v_not_b32), E(v_lshlrev_b32), E(v_lshrrev_b32), E(v_sub_u32), E(v_lshl_add_u32), E(v_add3_u32), E(v_or3_b32), E(v_lshl_or_b32), E(v_and_or_b32), E(v_add_u32_sdwa), E(v_max_f32), E(v_min3_f32), E(v_med3_f32), E(v_min_u32), E(v_div_scale_f32), E(v_div_fmas_f32), E(v_div_fixup_f32), E(v_cmp_lt_f32_e64), E(v_cmp_ne_u32_vcc), E(v_cmp_class_f32), E(v_cndmask_e64_vcc), E(v_writelane), E(v_bcnt_u32), E(v_ffbl_b32), E(v_mbcnt_hi), E(v_mul_lo_u16), E(v_cvt_u32_f32), E(v_lshl_add_u64), E(cmp_cndmask_vcc), E(cmp_cndmask_sgpr), E(cndmask_vcc_after_salu), E(mix_fma_min), E(mix_fma_salu_lds),
        E(cmp_3cndmask_vcc), E(cmp_add_2cndmask_vcc), E(cmp_3cndmask_e64_vcc), E(cmp_3cndmask_sgpr), E(addc_vcc), E(div_fmas_vcc), E(cbranch_vccz), E(s_and_saveexec), E(s_cbranch_scc), E(s_cbranch_execz), E(v_mul_f32_dep_sgprmix), E(v_lshlrev_b32_v), E(v_add_u32_const), E(v_mul_f32_const), E(v_fma_f32_const), E(v_ashrrev_i32), E(v_mad_u32_u24), E(v_mul_legacy_f32), E(v_max_f32_via_fma_test), E(v_mul_f32_omod),
        E(v_pk_fma_f16), E(v_pk_max_f16), E(v_pk_min_f16), E(v_pk_add_f16), E(v_pk_mul_f16), E(v_pk_mad_i16), E(v_pk_max_i16), E(v_pk_sub_i16), E(v_pk_lshrrev_b16), E(v_and_b32_lit), E(v_cvt_pkrtz_f16_f32), E(v_max_f16), E(v_fma_f16), E(v_alignbyte_b32), E(v_lshrrev_b32_imm), E(mix_pkfma_and)};
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    const int iters = 2000;
    std::printf("# %s, %d CUs, clock taken as %.0f MHz; SIMD cycles per wave-instruction (256-thread groups = one wave per SIMD each)\n", prop.name, cus, mhz);
    std::printf("%-18s %10s %10s %10s %12s\n", "instruction", "1 wave", "2 waves", "6 waves", "vs v_add_u32");
    double ref = 0;
    for (const Entry &e : ops) {
        if (only && std::strcmp(e.name, "v_add_u32") != 0) {
            const std::string list = std::string(",") + only + ",", key = std::string(",") + e.name + ",";
            if (list.find(key) == std::string::npos) continue;
        }
        double cyc[3];
        const int waves[3] = {1, 2, 6};
        for (int w = 0; w < 3; ++w) {
            const int groups = cus * waves[w];
            hipLaunchKernelGGL(e.fn, dim3(groups), dim3(256), 0, 0, out, 10);   // warm-up
            CHECK(hipEventRecord(t0));
            hipLaunchKernelGGL(e.fn, dim3(groups), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(t1));
            CHECK(hipEventSynchronize(t1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, t0, t1));
            // every SIMD runs `waves` waves of iters * 256 instructions (the ds_* groups carry one s_waitcnt per four)
            cyc[w] = ms * 1e-3 * mhz * 1e6 / (static_cast<double>(waves[w]) * iters * 256.0);
        }
        if (!std::strcmp(e.name, "v_add_u32")) ref = cyc[2];
        std::printf("%-18s %10.2f %10.2f %10.2f %12.2f\n", e.name, cyc[0], cyc[1], cyc[2], cyc[2] / ref);
        std::fflush(stdout);
    }
    return 0;
}
