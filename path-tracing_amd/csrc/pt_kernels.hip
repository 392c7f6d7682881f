// Radiance-integrator kernel for gfx950 (MI355X).  One lane = one pixel's path, one wave = an 8x8 pixel tile,
// all passes x segments x triangles of a row band in one launch.
//
// Per path segment a wave does three things:
//   1. CULL   triangles are listed in SLOT order (pt_scene.hpp): the table builder groups them spatially, whatever order the
//             OBJ lists them in.  Large triangles (walls) get a barycentric test, wave-uniformly, records through scalar
//             loads and used straight from SGPRs.  Small triangles hang under a hierarchy: in scenes of up to 2048 triangles
//             an 8-ary tree of bounding spheres per connected group, in bigger ones ONE tree of quantised boxes (64-byte
//             nodes) over all of them.  Either tree is walked with a wave-wide LIFO of (ray, node) work items in LDS, dealt out
//             evenly: lane l expands the l-th item whoever's ray it is; the queues are filled through a DPP prefix sum over the
//             lanes' survivor counts (one LDS store per entry, totals for the capacity checks for free).  The box walk also drops nodes entered beyond the
//             ray's best hit so far.  Every test is CONSERVATIVE with respect to the reference's Triangle::Intersect
//             (triangles.h:48-73): it may only say "cannot be a hit".  Survivors become (ray, slot) pairs in a second LDS queue.
//   2. EXACT  the pairs (about 1.5 per ray) are dealt out evenly as well; each runs the reference's arithmetic operation
//             for operation (same association, no FMA contraction, IEEE divide and sqrt), and the closest hit per ray is
//             taken with one LDS atomic-min on (distance, original triangle index), so `t`, the hit decision and the
//             closest-hit choice are bit-identical to the CPU path.
//   3. SHADE  Material::Process + the three lobes (material.h:36-102), Ray::Reflect (ray.h:45-50), the tile's accumulators
//             (material.h:74-77) in LDS, counter-based Philox4x32-10 randoms keyed by (seed | pixel, pass, segment).
//
// The kernel is bound by instruction issue -- its time follows the number of instructions its waves execute, vector or scalar
//  -- so the code below is written for few instructions, not for few memory accesses or many waves.
//
// Everything that decides a result is plain IEEE binary32/binary64 arithmetic (float sqrt and reciprocal through
// pt_fastfp.hpp: shorter sequences, verified against the correctly rounded result for every float of their range); only
// step 1 uses fused multiply-adds and raw v_rcp_f32 results, and step 1 cannot change a result (DESIGN.md "Culling: why it
// cannot reject a hit"; the verification build, -DPT_VERIFY_BRUTE, re-checks every segment against the all-triangles loop).
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "pt_fastfp.hpp"
#include "pt_kernels.hpp"

#pragma clang fp contract(off)

namespace pt {

namespace {

#ifndef PT_WAVES_PER_SIMD
#define PT_WAVES_PER_SIMD 6
#endif
constexpr int kBlock = 64;                // one wave = one 8x8 pixel tile per workgroup (all LDS below is wave-private)
// Capacity of the two wave-private work queues.  Small scenes (Tor.obj) keep them small so that 5 KB of LDS per wave
// leaves room for 6+ waves per SIMD; scenes with thousands of triangles get deep queues (fuller rounds) and pay with
// occupancy, which matters less there.
#ifndef PT_SMALL_NODES
#define PT_SMALL_NODES 160   // 96 / 144: 4 x the rounds that do not fit (partial commits), -0.9 %; LDS still allows 6 waves per SIMD
#endif
#ifndef PT_SMALL_PAIRS
#define PT_SMALL_PAIRS 256
#endif
struct SmallQueues { static constexpr int kNodeStack = PT_SMALL_NODES, kPairQueue = PT_SMALL_PAIRS, kFiltered = 1; };
// the same for the small-scene kernel with two rays per lane (twice the items per wave-segment)
#ifndef PT_SMALL2_NODES
#define PT_SMALL2_NODES 160
#endif
#ifndef PT_SMALL2_PAIRS
#define PT_SMALL2_PAIRS 256
#endif
struct SmallQueues2 { static constexpr int kNodeStack = PT_SMALL2_NODES, kPairQueue = PT_SMALL2_PAIRS, kFiltered = 1; };
#ifndef PT_BIG_NODES
#define PT_BIG_NODES 384      // measured on the 16 398- and 49 934-triangle scenes: 832 / 512 (4 waves per SIMD) is 4 % slower,
#endif
#ifndef PT_BIG_PAIRS
#define PT_BIG_PAIRS 256      // 256 / 160 (6 waves per SIMD) 7 % slower than this (5 waves per SIMD)
#endif
#ifndef PT_BIG_FILTERED
#define PT_BIG_FILTERED 128
#endif
#ifndef PT_BIG_EXACT_AT
#define PT_BIG_EXACT_AT 64    // big scenes: pre-filtered pairs waiting before an exact round runs (fewer = earlier pruning, emptier rounds)
#endif
#ifndef PT_BOX_SCHED
#define PT_BOX_SCHED 1        // box tree's child test: a scheduling barrier after every PT_BOX_SCHED children (9 = none)
#endif
#ifndef PT_BOX_SPREAD
#define PT_BOX_SPREAD 0       // 1: under-filled box-tree rounds spread a node's children over 2 / 4 / 8 lanes (measured: +-0 on the x64
#endif                        // replica, -1 % on x195 -- the tail rounds wait for their node loads, not for issue slots; ab62)
#ifndef PT_BIG_WAVES
#define PT_BIG_WAVES (PT_WAVES_PER_SIMD - 2)   // waves per SIMD the big-scene instantiations are compiled for (they fit 6: 77 VGPRs)
#endif
#ifndef PT_SKY_WAVES
// ... and the skybox instantiations: 5 since round 4 (96 VGPRs, no scratch; at 4 the compiler took 102-117): +9.6 % on the open
// scenes (Tor.obj without its back wall 6 950 -> 7 620 Msamples/s at 64 spp, the torus x9 3 585 -> 3 915; r04_ab_logs.txt sky5).
// At 6 (80 VGPRs) the lookup's double arithmetic spills 30 registers.
#define PT_SKY_WAVES (PT_WAVES_PER_SIMD - 1)
#endif
struct BigQueues { static constexpr int kNodeStack = PT_BIG_NODES, kPairQueue = PT_BIG_PAIRS, kFiltered = PT_BIG_FILTERED; };
// the box-tree kernel's adaptive instantiations keep one 16-bit word per pixel of their tile (128 / 256 pixels) in LDS: the node stack
// gives up that much, so that the wave stays within five 1 280-byte granules (six waves per SIMD)
#ifndef PT_BIG_ADAPT_NODES
#define PT_BIG_ADAPT_NODES(pixels_per_lane) ((PT_BIG_NODES + 46 - 32 * (pixels_per_lane)) > 64 ? (PT_BIG_NODES + 46 - 32 * (pixels_per_lane)) : 64)   // (the tiny-queue test build)
#endif
template <int OWN> struct BigQueuesAdapt { static constexpr int kNodeStack = PT_BIG_ADAPT_NODES(OWN), kPairQueue = PT_BIG_PAIRS, kFiltered = PT_BIG_FILTERED; };

// ---------------------------------------------------------------------------------------------------------------
// Counter RNG (layout shared with the CPU oracle; see DESIGN.md "Counter RNG")
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
    // The ten round keys are wave-uniform and loop-invariant; hoisted out of the path loops they would sit in ten
    // spilled SGPRs (a v_readlane + hazard nop each).  Opaque here, they are rebuilt with one scalar add per round.
    asm volatile("" : "+s"(k0));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = static_cast<unsigned long long>(c0) * 0xD2511F53u;   // one v_mad_u64_u32 each
        const unsigned long long p1 = static_cast<unsigned long long>(c2) * 0xCD9E8D57u;
        const uint32_t h0 = static_cast<uint32_t>(p0 >> 32), l0 = static_cast<uint32_t>(p0);
        const uint32_t h1 = static_cast<uint32_t>(p1 >> 32), l1 = static_cast<uint32_t>(p1);
        c0 = __builtin_amdgcn_bitop3_b32(h1, c1, k0, 0x96);   // three-way xor in one instruction (the compiler emits two v_xor_b32)
        c1 = l1;
        c2 = h0 ^ c3 ^ k1;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
__device__ __forceinline__ float unit_float(uint32_t w) {   // (0,1), never 0 or 1
    return static_cast<float>(((w >> 9) << 1) | 1u) * 5.9604644775390625e-08f;
}
__device__ __forceinline__ double jitter_double(uint32_t w) {   // (-0.5,0.5)
    return (static_cast<double>(w) + 0.5) * 2.3283064365386962890625e-10 - 0.5;
}

// sin/cos of a float angle in [0, 2pi]: double +,-,* only, so the result is the same on every IEEE machine.
// The sixteen double constants come from one table through the scalar cache (two s_load_dwordx16): as immediates each
// costs two s_mov_b32, a third of this function's instructions.
__constant__ double kSinCosTab[16] = {
    0.63661977236758138, 1.57079632673412561417e+00, 6.07710050650619224932e-11, 0.5,
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
    -2.50507602534068634195e-08, 1.58969099521155010221e-10,
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07,
    2.08757232129817482790e-09, -1.13596475577881948265e-11};
__device__ __forceinline__ void portable_sincos(float a, float &s_out, float &c_out) {
    typedef const __attribute__((address_space(4))) double *ConstD;
    ConstD t = (ConstD)reinterpret_cast<uintptr_t>(kSinCosTab);
    asm volatile("" : "+s"(t));   // keeps the compiler from folding the table back into immediates
    const double x = static_cast<double>(a);
    const int k = static_cast<int>(x * t[0] + t[3]);
    const double kd = static_cast<double>(k);
    const double r = (x - kd * t[1]) - kd * t[2];
    const double z = r * r;
    const double ps = t[4] + z * (t[5] + z * (t[6] + z * (t[7] + z * (t[8] + z * t[9]))));
    const double sn = r + r * (z * ps);
    const double pc = t[10] + z * (t[11] + z * (t[12] + z * (t[13] + z * (t[14] + z * t[15]))));
    const double cs = (1.0 - t[3] * z) + (z * z) * pc;
    // Quadrant k & 3 = 0: (sn, cs), 1: (cs, -sn), 2: (-sn, -cs), 3: (-cs, sn).  Rounding to float commutes with negation, so the
    // pair is narrowed first and then swapped / sign-flipped with integer operations.
    const uint32_t q = static_cast<uint32_t>(k);
    const float sf = static_cast<float>(sn), cf = static_cast<float>(cs);
    const bool odd = (q & 1u) != 0u;
    const uint32_t ua = __float_as_uint(odd ? cf : sf), ub = __float_as_uint(odd ? sf : cf);
    s_out = __uint_as_float(ua ^ ((q & 2u) << 30));
    c_out = __uint_as_float(ub ^ (((q + 1u) & 2u) << 30));
}

// Portable atan / atan2 / acos for the skybox lookup (scene.cpp:127-128): double +,-,*,/,sqrt only, the same
// sequence as the CPU oracle's, so that texel coordinates agree bit for bit.
__device__ __forceinline__ double portable_atan_pos(double x) {   // x >= 0, finite or +inf
    int id;
    double hi = 0.0, lo = 0.0;
    if (x < 0.4375) {
        id = -1;
    } else if (x < 1.1875) {
        if (x < 0.6875) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
        else { id = 1; x = (x - 1.0) / (x + 1.0); hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
    } else if (x < 2.4375) {
        id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17;
    } else {
        id = 3; x = -1.0 / x; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17;
    }
    const double z = x * x, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02
                    + w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02
                    + w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    if (id < 0) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}
__device__ __forceinline__ float portable_atan2f(float yf, float xf) {
    const double y = static_cast<double>(yf), x = static_cast<double>(xf);
    if (y != y || x != x) return __builtin_nanf("");
    const double pi = 3.14159265358979311600e+00, pi_2 = 1.57079632679489655800e+00;
    const double ay = y < 0 ? -y : y, ax = x < 0 ? -x : x;
    double r;
    if (ay == 0.0) r = (x < 0 || (x == 0 && __builtin_signbit(xf))) ? pi : 0.0;
    else if (ax == 0.0) r = pi_2;
    else {
        const double t = portable_atan_pos(ay / ax);
        r = x < 0 ? pi - t : t;
    }
    return static_cast<float>((y < 0 || (y == 0 && __builtin_signbit(yf))) ? -r : r);
}
__device__ __forceinline__ float portable_acosf(float vf) {
    const double v = static_cast<double>(vf);
    const double s = __builtin_sqrt((1.0 - v) * (1.0 + v));   // NaN for |v| > 1, like acosf
    if (s != s) return __builtin_nanf("");
    const double pi = 3.14159265358979311600e+00, pi_2 = 1.57079632679489655800e+00;
    double r;
    if (v == 0.0) r = pi_2;
    else {
        const double t = portable_atan_pos(s / (v < 0 ? -v : v));
        r = v < 0 ? pi - t : t;
    }
    return static_cast<float>(r);
}

// ---------------------------------------------------------------------------------------------------------------
// 1. CULL: may only answer "this triangle cannot be accepted by Triangle::Intersect for this ray".
// ---------------------------------------------------------------------------------------------------------------
struct Ray {
    float ox, oy, oz, dx, dy, dz;
};

typedef const __attribute__((address_space(4))) float *ConstF;
__device__ __forceinline__ CullRec load_cull(ConstF p) {
    CullRec r;
    r.n[0] = p[0]; r.n[1] = p[1]; r.n[2] = p[2]; r.w = p[3];
    r.au[0] = p[4]; r.au[1] = p[5]; r.au[2] = p[6]; r.cu = p[7];
    r.av[0] = p[8]; r.av[1] = p[9]; r.av[2] = p[10]; r.cv = p[11];
    return r;
}
typedef const __attribute__((address_space(4))) uint32_t *ConstU;

// Keep the node iff the ray (not the whole line) comes within sqrt(r2) of the centre.  Written so that a NaN keeps.
__device__ __forceinline__ bool sphere_keep(float cx, float cy, float cz, float r2, const Ray &q) {
    const float mx = cx - q.ox, my = cy - q.oy, mz = cz - q.oz;
    const float b = __builtin_fmaf(mx, q.dx, __builtin_fmaf(my, q.dy, mz * q.dz));
    const float m2 = __builtin_fmaf(mx, mx, __builtin_fmaf(my, my, mz * mz));
    const float bb = __builtin_fmaxf(b, 0.0f);
    const float disc = __builtin_fmaf(-bb, bb, m2);
    return !(disc > r2);
}

// A quad record = the two halves of a parallelogram in one stored plane (pt_scene.hpp: ClusterDesc): the plane rows
// as in a triangle record, then alpha and beta rows.  Half A keeps iff min(beta, alpha-beta, 1-alpha) >= -mg,
// half B iff min(alpha, beta-alpha, 1-beta) >= -mg.  Returns bit 0 = reject A, bit 1 = reject B.
__device__ __forceinline__ uint32_t cull_reject_quad(const CullRec r, const Ray &q, float k1, float k2, float a_max, float m0q,
                                                     float t_guard) {
    const float num = __builtin_fmaf(q.ox, r.n[0], __builtin_fmaf(q.oy, r.n[1], __builtin_fmaf(q.oz, r.n[2], r.w)));
    const float den = __builtin_fmaf(q.dx, r.n[0], __builtin_fmaf(q.dy, r.n[1], q.dz * r.n[2]));
    const float rden = __builtin_amdgcn_rcpf(den);
    const float t = -num * rden;
    const float px = __builtin_fmaf(t, q.dx, q.ox), py = __builtin_fmaf(t, q.dy, q.oy), pz = __builtin_fmaf(t, q.dz, q.oz);
    const float al = __builtin_fmaf(px, r.au[0], __builtin_fmaf(py, r.au[1], __builtin_fmaf(pz, r.au[2], r.cu)));
    const float be = __builtin_fmaf(px, r.av[0], __builtin_fmaf(py, r.av[1], __builtin_fmaf(pz, r.av[2], r.cv)));
    const float d = al - be;
    const float ea = __builtin_fminf(__builtin_fminf(be, d), 1.0f - al);
    const float eb = __builtin_fminf(__builtin_fminf(al, -d), 1.0f - be);
    const float et = __builtin_fmaf(k1, __builtin_fabsf(t), k2) * __builtin_fabsf(rden);
    const float mg = __builtin_fmaf(a_max, et, m0q);
    const bool behind = t < -et;
    const bool trusted = __builtin_fabsf(t) < t_guard;
    return ((trusted & ((ea < -mg) | behind)) ? 1u : 0u) | ((trusted & ((eb < -mg) | behind)) ? 2u : 0u);
}

__device__ __forceinline__ bool cull_reject(const CullRec r, const Ray &q, float k1, float k2, float a_max, float m0,
                                            float t_guard) {
    const float num = __builtin_fmaf(q.ox, r.n[0], __builtin_fmaf(q.oy, r.n[1], __builtin_fmaf(q.oz, r.n[2], r.w)));
    const float den = __builtin_fmaf(q.dx, r.n[0], __builtin_fmaf(q.dy, r.n[1], q.dz * r.n[2]));
    const float rden = __builtin_amdgcn_rcpf(den);
    const float t = -num * rden;
    const float px = __builtin_fmaf(t, q.dx, q.ox), py = __builtin_fmaf(t, q.dy, q.oy), pz = __builtin_fmaf(t, q.dz, q.oz);
    const float u = __builtin_fmaf(px, r.au[0], __builtin_fmaf(py, r.au[1], __builtin_fmaf(pz, r.au[2], r.cu)));
    const float v = __builtin_fmaf(px, r.av[0], __builtin_fmaf(py, r.av[1], __builtin_fmaf(pz, r.av[2], r.cv)));
    const float w = (1.0f - u) - v;
    const float e = __builtin_fminf(__builtin_fminf(u, v), w);
    // |t - t_reference| <= et ; a point the reference accepts has every barycentric coordinate >= -mg
    const float et = __builtin_fmaf(k1, __builtin_fabsf(t), k2) * __builtin_fabsf(rden);
    const float mg = __builtin_fmaf(a_max, et, m0);
    const bool outside = e < -mg;
    const bool behind = t < -et;                           // then t_reference < 0 < eps (triangles.h:51)
    const bool trusted = __builtin_fabsf(t) < t_guard;     // false for NaN/inf and for grazing rays
    return trusted & (outside | behind);                   // bitwise on purpose: no divergent branches in the cull loop
}

// ---------------------------------------------------------------------------------------------------------------
// 2. EXACT: Triangle::Intersect (triangles.h:48-73) with PlaneIntersect (:10-13) and ParallelogramSquare (:15-17).
// Every operation is written in the reference's order; GLM's cross/length association is kept.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float area2_of(float ax, float ay, float az, float bx, float by, float bz) {   // |a x b|^2
    const float cx = ay * bz - by * az;
    const float cy = az * bx - bz * ax;
    const float cz = ax * by - bx * ay;
    return cx * cx + cy * cy + cz * cz;
}
// Stages B-D of Triangle::Intersect for one (ray, triangle) pair, without the running `distance`:
// returns new_distance if the point passes the area tests, -inf otherwise (then stage A rejects it: -inf < eps).
__device__ __forceinline__ float exact_inside(const ExactRec *__restrict__ rec, const Ray &q, float eps, uint32_t &orig) {
    const float4 r0 = reinterpret_cast<const float4 *>(rec)[0];   // plane
    const float4 r1 = reinterpret_cast<const float4 *>(rec)[1];   // v0, square
    const float4 r2 = reinterpret_cast<const float4 *>(rec)[2];   // v1, material
    const float4 r3 = reinterpret_cast<const float4 *>(rec)[3];   // v2, original triangle index
    orig = __float_as_uint(r3.w);
    const float signed_dist = q.dx * r0.x + q.dy * r0.y + q.dz * r0.z;
    const float nd = -(q.ox * r0.x + q.oy * r0.y + q.oz * r0.z + r0.w) / signed_dist;
    const float px = q.ox + q.dx * nd, py = q.oy + q.dy * nd, pz = q.oz + q.dz * nd;
    const float f0x = px - r1.x, f0y = py - r1.y, f0z = pz - r1.z;
    const float f1x = px - r2.x, f1y = py - r2.y, f1z = pz - r2.z;
    const float f2x = px - r3.x, f2y = py - r3.y, f2z = pz - r3.z;
    const float sq = r1.w;
    const float q1 = area2_of(f0x, f0y, f0z, f1x, f1y, f1z);
    const float q2 = area2_of(f0x, f0y, f0z, f2x, f2y, f2z);
    const float q3 = area2_of(f2x, f2y, f2z, f1x, f1y, f1z);
    float s1, s2, s3;
    // one range check for the three roots: min and max of the squared areas decide for all of them
    if (__all(fast_fp_ok(__builtin_fminf(__builtin_fminf(q1, q2), q3)) & fast_fp_ok(__builtin_fmaxf(__builtin_fmaxf(q1, q2), q3)))) {
        s1 = sqrt_rn_normal(q1); s2 = sqrt_rn_normal(q2); s3 = sqrt_rn_normal(q3);
    } else {
        s1 = __builtin_sqrtf(q1); s2 = __builtin_sqrtf(q2); s3 = __builtin_sqrtf(q3);
    }
    const bool stage_b = !(s1 > sq + eps);
    const bool stage_c = !(s1 + s2 > sq + eps);
    const bool stage_d = !(__builtin_fabsf(sq - s1 - s2 - s3) > eps);
    return (stage_b && stage_c && stage_d) ? nd : -__builtin_inff();
}

// Inclusive prefix sum over the 64 lanes in six DPP additions: within rows of 16 lanes (row_shr 1, 2, 4, 8), then row 0
// into row 1 and row 2 into row 3 (row_bcast:15), then rows 0-1 into rows 2-3 (row_bcast:31).  Lane 63 holds the total.
// The work queues are filled with it: a lane's entries go to [base + exclusive prefix, ...), one ds_write per entry,
// instead of one wave-wide ballot round per bit plane (which cost a fifth of the big-scene kernel's instructions,
// profiles/r02_blockprof_*.txt) -- and the total, which the capacity checks need, comes for free.
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v) {
    int x = static_cast<int>(v);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true);
    return static_cast<uint32_t>(x);
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63)); }
// Minimum over the 64 lanes, wave-uniform: the same six DPP steps with min instead of + (a lane without a source keeps ~0).
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
    int x = static_cast<int>(v);
    auto step = [&](int moved) { x = static_cast<int>(min(static_cast<uint32_t>(x), static_cast<uint32_t>(moved))); };
    step(__builtin_amdgcn_update_dpp(-1, x, 0x111, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp(-1, x, 0x112, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp(-1, x, 0x114, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp(-1, x, 0x118, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp(-1, x, 0x142, 0xA, 0xF, false));
    step(__builtin_amdgcn_update_dpp(-1, x, 0x143, 0xC, 0xF, false));
    return wave_last(static_cast<uint32_t>(x));
}
// One LDS store per set bit of `bits`: entry = ebase + (bit index << shift), at queue[pos], queue[pos + 1], ...
__device__ __forceinline__ void emit_bits(uint32_t *queue, uint32_t pos, uint32_t bits, uint32_t ebase, uint32_t shift = 0) {
    while (bits != 0) {
        const uint32_t j = __builtin_ctz(bits);
        bits &= bits - 1;
        queue[pos++] = ebase + (j << shift);
    }
}
// LDS traffic between lanes of ONE wave: the hardware executes a wave's LDS instructions in order, so only the
// compiler has to be stopped from moving them across the hand-off.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {   // glm::normalize(vec4(x,y,z,0))
    const float d = (x * x + y * y) + z * z;
    float inv;
    if (__all(fast_fp_ok(d))) inv = rcp_rn_normal(sqrt_rn_normal(d));   // the root of an in-range number is in range
    else inv = 1.0f / __builtin_sqrtf(d);
    x = x * inv; y = y * inv; z = z * inv;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Closest hit of every lane's ray: Scene::TraceRay's loop (scene.cpp:114-120) for one wave.  Wave-uniform control
// flow: all 64 lanes must call it together; lanes with valid == false take part in the shared work only.
// ---------------------------------------------------------------------------------------------------------------
// R = rays per lane.  A wave owns 64 R pixels (an 8R x 8 tile) and a wave-segment searches 64 R rays: lane l carries rays
// l, l + 64, ...  Everything wave-uniform about a segment (the scalar half of the large-class loop, the cluster loop, queue
// bookkeeping) is then shared by R times the rays, and the lane-balanced rounds (sphere-tree items, exact pairs) are that much
// fuller -- for Tor.obj, where 9 % of the rays come near the torus and a segment yields 1.5 pairs per ray, that is where a sixth
// of the instructions went.  Big scenes keep R = 1 (their rounds are full anyway, their registers are not).
template <int N> struct AccPlanes { float v[7][N]; };
template <> struct AccPlanes<0> {};
template <int N> struct FlagWords { uint16_t v[N]; };
template <> struct FlagWords<0> {};
#ifndef PT_MATS_IN_LDS
#define PT_MATS_IN_LDS 16   // two-pixel kernels: a scene's materials, if it has at most this many, are read from a copy in LDS
#endif
template <int N> struct InvPlanes { float v[3][N]; };
template <> struct InvPlanes<0> {};
template <int N> struct MatCache { float4 v[3 * N]; };
template <> struct MatCache<0> {};
template <class Q, int R, int FLAGS = 0>
struct WaveLds {
    static constexpr int kRays = R, kSlots = 64 * R;
    static constexpr int kNodeStack = Q::kNodeStack, kPairQueue = Q::kPairQueue;
    static constexpr bool kPrefilter = Q::kFiltered > 1;   // big scenes: thin the pairs with the barycentric test first
    static_assert(R == 1 || R == 2, "ray ids take 6 or 7 bits of a work item");
    // work-item layouts: sphere-tree stack entries = ray << kNodeSrcShift | level << kNodeLevShift | node index within its level;
    // pairs = slot | ray << 24 (| kUnfiltered << 24 with the pre-filter)
    static constexpr uint32_t kSrcMask = kSlots - 1, kNodeSrcShift = R == 1 ? 26 : 25, kNodeLevShift = kNodeSrcShift - 3;
    // The code that fills the queues relies on these minima (see push_pairs_any, drain_pairs and the tree walk):
    static_assert(Q::kPairQueue >= 128, "push_pairs_any publishes slices of up to 128 pairs");
    static_assert(!kPrefilter || Q::kFiltered >= 128, "the pre-filter appends up to 64 survivors to up to 63 waiting ones");
    static_assert(!kPrefilter || Q::kFiltered >= PT_BIG_EXACT_AT - 1 + 64, "up to PT_BIG_EXACT_AT - 1 pairs wait when 64 survivors are appended");
    static_assert(Q::kNodeStack >= 64, "a round pops up to 64 nodes");
    uint32_t filtered[Q::kFiltered];   // pairs that survived the pre-filter, waiting for a full exact round
    unsigned long long best[kSlots];   // per ray: (order-preserving bits of t) << 32 | triangle index; smaller is closer
    float ray[6][kSlots];          // this segment's rays, readable by every lane
    // big scenes: 1 / direction of every ray, taken once per segment instead of at every node visit (three quarter-rate v_rcp_f32
    // per visit, six visits per ray: +1.4 % / +0.9 % on the replicas, ab68)
    InvPlanes<kPrefilter ? kSlots : 0> rinv;
    uint32_t nodes[kNodeStack + 64];// LIFO of tree nodes to expand (layout above; the box tree: ray << 26 | node)
    uint32_t pairs[kPairQueue];    // (ray, triangle) work items
    uint32_t level_off[kMaxLevels];// sphere offset of each level of the cluster being walked
    uint32_t level_cnt[kMaxLevels];// number of real nodes of each level
    // The tile's accumulators (sum rgb, sum2 rgb, count as int bits) live here for the small-scene kernels with one ray per lane;
    // the two-rays-per-lane kernel and the big-scene kernels leave them in memory (integrate_kernel: "Where the tile's
    // accumulators live"): the LDS is worth a wave per SIMD to them.
    static constexpr bool kAccInLds = R == 1 && !kPrefilter;
    AccPlanes<kAccInLds ? kSlots : 0> acc;
    // adaptive-sampling instantiations of the two-pixel kernel: one 16-bit word per pixel of the tile (FLAGS per lane: pixel lane +
    // 64 j) -- the pixel's next pass << 1 | the cached "variance is low" answer (launches whose passes end beyond kMaxBatchPass
    // run the plain kernel); any lane may be the one that traces it, see "Batches" in integrate_kernel.  (16 bits: with 32-bit
    // words the 32 x 8 instantiation's 7 888 B of LDS round up to seven 1 280-byte granules and cost it its fifth wave per SIMD:
    // +13 % frame time, profiles/r04_ab_logs.txt adapt2)
    FlagWords<64 * FLAGS> low;
    // Shading reads the hit's record and then, through its material index, the material: two dependent loads.  The two-pixel
    // kernels take the second from a copy in LDS when the scene has at most kMatCache materials (Tor.obj: 5): +0.8 % (256 spp:
    // 73.2 -> 72.6 ms; with adaptive sampling +1.3 %); the box-tree kernel gains nothing from it (ab64) and does without.
    static constexpr int kMatCache = (R == 2 && !kPrefilter) ? PT_MATS_IN_LDS : 0;
    MatCache<kMatCache> mat;
};
struct WaveStats {
    uint32_t n_exact = 0, w_segments = 0, w_node_rounds = 0, w_exact_iters = 0, w_partial = 0;   // wave-uniform, live in SGPRs
#ifdef PT_VERIFY_BRUTE
    uint32_t v_checked = 0, v_bad = 0;   // verification build: segments compared with the brute-force loop / differing
#endif
#ifdef PT_PHASE_TIMERS
    // diagnostic build only: shader-clock cycles per phase (never compiled into the shipped library)
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last = 0;
#endif
};
// Launches that do not ask for pt_render_stats run an instantiation without any of the counters (they cost ~3 %:
// nine more wave-uniform values alive across the whole kernel push the scalar register file into spilling).
struct Ignored {
    __host__ __device__ Ignored() {}
    __host__ __device__ Ignored(uint32_t) {}
    template <class T> __host__ __device__ Ignored &operator+=(T) { return *this; }
    __host__ __device__ Ignored &operator++() { return *this; }
};
struct NoStats {
    Ignored n_exact, w_segments, w_node_rounds, w_exact_iters, w_partial;
#ifdef PT_VERIFY_BRUTE
    Ignored v_checked, v_bad;   // keeps the verification build compiling; it always launches the STATS instantiations
#endif
#ifdef PT_PHASE_TIMERS
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;   // keeps the diagnostic build compiling; never launched there
#endif
};
#ifdef PT_PHASE_TIMERS
#define PT_STAMP(st, idx)                                              \
    do {                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        (st).phase[idx] += now_ - (st).last;                           \
        (st).last = now_;                                              \
    } while (0)
#else
#define PT_STAMP(st, idx) do { } while (0)
#endif

constexpr uint32_t kUnfiltered = 128u;  // added to the ray id of a pair (bit 31 of the work item): the pre-filter must not judge it

// float -> uint32 whose unsigned order is the float order (for ds_min_u64 keys)
__device__ __forceinline__ uint32_t ordered_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t b) {
    return __uint_as_float(b ^ ((b >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// hides a value from loop-invariant code motion (see the accumulator addresses and the camera ray)
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}


// ---------------------------------------------------------------------------------------------------------------
// Box tree of big scenes (pt_scene.hpp: BvhNode): which of a node's (up to 8) children can hold a hit of ray r that
// beats t_best?  Slab test against the 8-bit child boxes, dequantised on the fly: along axis x the planes of child c are
// t = (org.x + q step - o.x) / d.x = A q + B with A = step / d.x, B = (org.x - o.x) / d.x, one fma per plane.
// CONSERVATIVE: every computed t is within E = err (|B| + 255 |A|) of its exact value (rcp 1 ulp, one product, the
// subtraction of the allowance, one fma: < 3.6e-7 relative to the operands' magnitudes; err = 5e-7), the boxes were
// rounded outward on the host, and a child is dropped only if its interval misses [0, t_best] by more than 2E.  A direction
// component of exactly zero (or too small for its reciprocal) makes A and B infinite: the allowance becomes infinite, every entry
// plane -inf or NaN, and every child of the node is kept (max / min skip NaNs, the final comparison is written so that a NaN keeps)
// -- conservative, and rare enough (no camera ray, no sampled direction has an exactly zero component) not to be worth the nine
// instructions per node that clamping the component to 1e-30 cost.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float byte_to_float(uint32_t w, int k) { return static_cast<float>((w >> (8 * k)) & 0xFFu); }   // v_cvt_f32_ubyteK
// PER = 8: all children of the node.  PER = 4, 2, 1: the node's children are spread over 2, 4, 8 lanes (under-filled rounds) and
// this lane tests children sub * PER ... sub * PER + PER - 1; the result has its bits at those positions.
template <int PER = 8>
__device__ __forceinline__ uint32_t box_children_kept(const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3, const Ray &r,
                                                      float ix, float iy, float iz, float t_best, float err, uint32_t sub = 0) {
    // ix, iy, iz = v_rcp_f32 of the ray's direction components (the caller holds them per ray)
    const float step = __uint_as_float((q0.w & 0xFFu) << 23);
    const float ax = step * ix, ay = step * iy, az = step * iz;
    const float bx = (__uint_as_float(q0.x) - r.ox) * ix, by = (__uint_as_float(q0.y) - r.oy) * iy, bz = (__uint_as_float(q0.z) - r.oz) * iz;
    const float bmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(bx), __builtin_fabsf(by)), __builtin_fabsf(bz));
    const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ax), __builtin_fabsf(ay)), __builtin_fabsf(az));
    const float e2 = 2.0f * err * __builtin_fmaf(255.0f, amax, bmax);
    // The allowance goes into the ENTRY planes once per node (entry - 2E against exit) instead of into every comparison.
    const float nbx = bx - e2, nby = by - e2, nbz = bz - e2, t_min = -e2;
    // The ray enters a slab through its lower plane if it travels upwards along that axis, else through the upper one:
    // pick the byte rows of the entry (n) and exit (f) planes once per node.  q -> fma(A, q, B) is monotone, so
    // entry <= exit per axis holds in float arithmetic too.
    const bool sx = ix < 0.0f, sy = iy < 0.0f, sz = iz < 0.0f;
    uint32_t nx0 = sx ? q2.z : q1.x, nx1 = sx ? q2.w : q1.y, fx0 = sx ? q1.x : q2.z, fx1 = sx ? q1.y : q2.w;
    uint32_t ny0 = sy ? q3.x : q1.z, ny1 = sy ? q3.y : q1.w, fy0 = sy ? q1.z : q3.x, fy1 = sy ? q1.w : q3.y;
    uint32_t nz0 = sz ? q3.z : q2.x, nz1 = sz ? q3.w : q2.y, fz0 = sz ? q2.x : q3.z, fz1 = sz ? q2.y : q3.w;
    if constexpr (PER < 8) {
        // this lane's children sit in one word per row (children 0-3 in the first, 4-7 in the second), from byte (sub * PER) & 3 on:
        // bring them to byte 0 of the "first" words, so that the byte indices below stay compile-time constants
        const uint32_t c0 = sub * PER;
        const bool up = c0 >= 4u;
        const uint32_t sh = 8u * (c0 & 3u);
        nx0 = (up ? nx1 : nx0) >> sh; fx0 = (up ? fx1 : fx0) >> sh;
        ny0 = (up ? ny1 : ny0) >> sh; fy0 = (up ? fy1 : fy0) >> sh;
        nz0 = (up ? nz1 : nz0) >> sh; fz0 = (up ? fz1 : fz0) >> sh;
    }
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        const int k = c & 3;
        const bool up = c >= 4;
        const float tnx = __builtin_fmaf(ax, byte_to_float(up ? nx1 : nx0, k), nbx), tfx = __builtin_fmaf(ax, byte_to_float(up ? fx1 : fx0, k), bx);
        const float tny = __builtin_fmaf(ay, byte_to_float(up ? ny1 : ny0, k), nby), tfy = __builtin_fmaf(ay, byte_to_float(up ? fy1 : fy0, k), by);
        const float tnz = __builtin_fmaf(az, byte_to_float(up ? nz1 : nz0, k), nbz), tfz = __builtin_fmaf(az, byte_to_float(up ? fz1 : fz0, k), bz);
        const float t_in = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, t_min));
        const float t_out = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, t_best));
        m |= !(t_in > t_out) ? (1u << c) : 0u;   // a NaN keeps
        // One child at a time.  Left alone, the compiler interleaves the eight children's conversions, multiply-adds and
        // comparisons into one long schedule; a scheduling barrier after every child -- the same instructions, 222 per node visit,
        // the same registers -- makes the x64 / x195 replicas 3.8 % / 2.9 % faster (2 152 -> 2 234, 1 447 -> 1 492 Msamples/s;
        // after 2 or 4 children: +0.9 % / 0; r04_ab_logs.txt sched).  Nothing else in the kernels reacted to such barriers.
        if ((c + 1) % PT_BOX_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PER < 8) m <<= sub * PER;
    return m;
}

#ifndef PT_BOX_MIX
#define PT_BOX_MIX 0   // 1: the float test with the byte -> float conversion folded into the multiply-add (box_children_kept_mix)
#endif
// ---------------------------------------------------------------------------------------------------------------
// The float test again, bit for bit -- with the conversions gone (round 4; profiles/r04_ab_logs.txt slabmix).  A plane byte q,
// zero-extended to 16 bits, is the half-precision subnormal q 2^-24; v_fma_mix_f32 takes one operand as a half, widens it exactly
// and does the fused multiply-add in float32.  With A 2^24 for coefficient (folded into the step's exponent) the product is the
// same real number A q, the sum and its single rounding the same: fma(A 2^24, q 2^-24, B) == fma(A, float(q), B) for every input.
// So the masks, the rounds, the frames and every argument of DESIGN.md section 5 are those of box_children_kept; what changes is
// the instruction count: per child six v_fma_mix_f32 instead of six v_cvt_f32_ubyteN + six v_fma_f32, and per node 24
// instructions that unpack the bytes of the twelve row words into halves (one unpacked register serves two children).
// ---------------------------------------------------------------------------------------------------------------
// (Written as inline assembly on purpose.  Left to the compiler -- __builtin_fmaf(a, float(half), b) selects the same
// v_fma_mix_f32, 203 instructions per node visit instead of 218 with the waits it puts around inline assembly -- the kernel is
// 2.5 % SLOWER than the float test on the x64 replica instead of 0.8 % faster: its scheduler hoists all twelve unpackings and
// lengthens the dependent stretch of a visit; r04_ab_logs.txt slabmix.)
template <int HI>
__device__ __forceinline__ float fma_mix_h(float a, uint32_t q2, float b) {   // a * half(q2.lo or q2.hi) + b, one rounding, float32
    float d;
    if constexpr (HI == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(q2), "v"(b));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(q2), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t box_children_kept_mix(const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3, const Ray &r,
                                                          float ix, float iy, float iz, float t_best, float err) {
    // the coefficients times 2^24: the factor goes into the step's exponent (pt_scene.cpp keeps it at or below 2^100), everything
    // derived from them scales back by an exact power of two, so e2 and the planes are those of box_children_kept bit for bit
    const float step24 = __uint_as_float(((q0.w & 0xFFu) + 24u) << 23);
    const float cx = step24 * ix, cy = step24 * iy, cz = step24 * iz;
    const float bx = (__uint_as_float(q0.x) - r.ox) * ix, by = (__uint_as_float(q0.y) - r.oy) * iy, bz = (__uint_as_float(q0.z) - r.oz) * iz;
    const float bmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(bx), __builtin_fabsf(by)), __builtin_fabsf(bz));
    const float cmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cx), __builtin_fabsf(cy)), __builtin_fabsf(cz));
    const float e2 = 2.0f * err * __builtin_fmaf(255.0f * 5.9604644775390625e-08f, cmax, bmax);   // 255 |A|max + |B|max
    const float nbx = bx - e2, nby = by - e2, nbz = bz - e2, t_min = -e2;
    const bool sx = ix < 0.0f, sy = iy < 0.0f, sz = iz < 0.0f;
    const uint32_t nx[2] = {sx ? q2.z : q1.x, sx ? q2.w : q1.y}, fx[2] = {sx ? q1.x : q2.z, sx ? q1.y : q2.w};
    const uint32_t ny[2] = {sy ? q3.x : q1.z, sy ? q3.y : q1.w}, fy[2] = {sy ? q1.z : q3.x, sy ? q1.w : q3.y};
    const uint32_t nz[2] = {sz ? q3.z : q2.x, sz ? q3.w : q2.y}, fz[2] = {sz ? q2.x : q3.z, sz ? q2.y : q3.w};
    uint32_t m = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            // bytes (0, 2) or (1, 3) of a row word as two 16-bit numbers: children 4 w + par and 4 w + par + 2
            auto un = [&](uint32_t v) {
                if (par == 0) return v & 0x00FF00FFu;
                uint32_t d;
                asm("v_pk_lshrrev_b16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(v));   // (op_sel_hi: the inline 8 shifts the high half too)
                return d;
            };
            const uint32_t unx = un(nx[w]), uny = un(ny[w]), unz = un(nz[w]), ufx = un(fx[w]), ufy = un(fy[w]), ufz = un(fz[w]);
            auto child = [&](auto hi_c) {
                constexpr int HI = decltype(hi_c)::value;
                const float tnx = fma_mix_h<HI>(cx, unx, nbx), tfx = fma_mix_h<HI>(cx, ufx, bx);
                const float tny = fma_mix_h<HI>(cy, uny, nby), tfy = fma_mix_h<HI>(cy, ufy, by);
                const float tnz = fma_mix_h<HI>(cz, unz, nbz), tfz = fma_mix_h<HI>(cz, ufz, bz);
                // (max / min by name: on the outputs of inline assembly the compiler would first quiet possible signalling NaNs,
                // one v_max_f32 x, x per value)
                float t_in, t_out;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t_in) : "v"(tnx), "v"(tny), "v"(tnz));
                asm("v_max_f32 %0, %1, %2" : "=v"(t_in) : "v"(t_in), "v"(t_min));
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t_out) : "v"(tfx), "v"(tfy), "v"(tfz));
                asm("v_min_f32 %0, %1, %2" : "=v"(t_out) : "v"(t_out), "v"(t_best));
                m |= !(t_in > t_out) ? (1u << (4 * w + par + 2 * HI)) : 0u;   // a NaN keeps
            };
            child(std::integral_constant<int, 0>());
            child(std::integral_constant<int, 1>());
            // one pair of children at a time: the unpacked rows of all four would cost a wave per SIMD
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return m;
}

#ifndef PT_BOX_F16
#define PT_BOX_F16 0   // 1: the box tree's child boxes are tested two at a time in packed half precision (box_children_kept_h)
#endif
// ---------------------------------------------------------------------------------------------------------------
// The same test in PACKED HALF PRECISION, two children per instruction (round 4; profiles/r04_ab_logs.txt slab16).
//   * The ray is recentred on the point where it enters the node's frame (t_enter = the latest of the three near planes of the
//     frame [0, 255]^3) and scaled by a power of two S: T = (t - t_enter) S, with S chosen from the exponent of the smallest
//     |A| so that the frame's extent along the ray, at most 255 min|A|, is below 2^-10.  What the test compares then lies in
//     [0, 2^-10), where half precision resolves 2^-21 or better: an eighth to a quarter of the smallest quantisation step.
//   * A child plane byte q is used AS IT IS: zero-extended to 16 bits it is the half-precision subnormal q 2^-24, and
//     v_pk_fma_f16 multiplies it exactly (half-precision denormals are on: the kernel descriptor's default); the factor 2^24 goes
//     into the coefficient.  A ray travelling down an axis sees the frame mirrored (q -> 255 - q = ~q), so every coefficient is
//     non-negative and every offset non-positive, and the DIRECTED roundings are free: v_cvt_pkrtz rounds the entry planes'
//     coefficient down and the exit planes' offset up, one added to the bit pattern gives the other two.
//   * Both roundings of a comparison (the two fused multiply-adds, half an ulp of a value below 2^-10 each: 2^-21 (1 + 2^-6)) and
//     the float32 stage's own error (3 err (|B| + 255 |A|)_max S) go into the entry planes' offset once per node, like the
//     allowance of the float version.
//   * Overflow is conservative by construction: a coefficient beyond the half-precision range saturates at 65504 (entry: a lower
//     bound) and its successor is +inf (exit: an upper bound); inf * 0 and inf - inf give NaNs, which max / min skip and whose
//     difference has a clear sign bit: kept.
// CONSERVATIVE like the float test (tests/bvh_emulation.py: children_kept_f16 restates it operation for operation;
// tests/test_cull_tables_host.py holds it to the chain test), a little looser: 2 % more node visits and 8 % more (ray, triangle)
// pairs on the x64 replica (tools/slab_f16_study.py).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_fma_h(uint32_t a, uint32_t q, uint32_t b) {
    uint32_t d;
    asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(q), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_max_h(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_min_h(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_sub_h(uint32_t a, uint32_t b) {   // a - b in both halves
    uint32_t d;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_rtz2(float x) {   // (x, x) as two halves, rounded toward zero
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x, x));
}
__device__ __forceinline__ uint32_t box_children_kept_h(const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3, const Ray &r,
                                                        float ix, float iy, float iz, float t_best, float err) {
    const float step = __uint_as_float((q0.w & 0xFFu) << 23);
    const float ax = step * ix, ay = step * iy, az = step * iz;
    const float bx = (__uint_as_float(q0.x) - r.ox) * ix, by = (__uint_as_float(q0.y) - r.oy) * iy, bz = (__uint_as_float(q0.z) - r.oz) * iz;
    // the frame's near plane along every axis, the point of entry, the scale
    const float lx = __builtin_fminf(bx, __builtin_fmaf(255.0f, ax, bx)), ly = __builtin_fminf(by, __builtin_fmaf(255.0f, ay, by)),
                lz = __builtin_fminf(bz, __builtin_fmaf(255.0f, az, bz));
    const float t_enter = __builtin_fmaxf(__builtin_fmaxf(lx, ly), lz);
    const float amin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(ax), __builtin_fabsf(ay)), __builtin_fabsf(az));
    const float s24 = __uint_as_float(0x81800000u - (__float_as_uint(amin) & 0x7F800000u));   // 2^(5 - exponent of amin): amin s24 in [2^5, 2^6)
    const float s = s24 * 5.9604644775390625e-08f;                                            // 2^-24
    // the allowance, in T units: the float32 stage's own rounding (reciprocal to an ulp, products, differences: within err of
    // |B| + 255 |A| per plane as in the float test, taken with half as much again for the recentring's extra difference) + the two
    // half-precision roundings of a comparison, 2^-21 (1 + 2^-6); both scale with err (test hooks: CullMutation::box_err)
    const float bmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(bx), __builtin_fabsf(by)), __builtin_fabsf(bz));
    const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ax), __builtin_fabsf(ay)), __builtin_fabsf(az));
    const float m = __builtin_fmaf(3.0f * err * __builtin_fmaf(255.0f, amax, bmax), s, err * (4.842877388e-07f / 5.0e-07f));
    const uint32_t one = 0x00010001u;                      // one ulp away from zero, both halves
    const uint32_t anx = pk_rtz2(__builtin_fabsf(ax) * s24), any_ = pk_rtz2(__builtin_fabsf(ay) * s24), anz = pk_rtz2(__builtin_fabsf(az) * s24);
    const uint32_t afx = anx + one, afy = any_ + one, afz = anz + one;
    const float dx = lx - t_enter, dy = ly - t_enter, dz = lz - t_enter;   // <= 0
    const uint32_t bfx = pk_rtz2(dx * s), bfy = pk_rtz2(dy * s), bfz = pk_rtz2(dz * s);
    const uint32_t bnx = pk_rtz2(__builtin_fmaf(dx, s, -m)) + one, bny = pk_rtz2(__builtin_fmaf(dy, s, -m)) + one, bnz = pk_rtz2(__builtin_fmaf(dz, s, -m)) + one;
    const float tm = __builtin_fmaf(-t_enter, s, -m), tb = (t_best - t_enter) * s;
    const uint32_t tmin = pk_rtz2(__builtin_fmaf(__builtin_fabsf(tm), -1.953125e-03f, tm));    // rounded down whatever its sign (2^-9 of slack)
    const uint32_t tbest = pk_rtz2(__builtin_fmaf(__builtin_fabsf(tb), 1.953125e-03f, tb));    // rounded up
    // rows: near = lower planes for a ray going up the axis, else the mirrored upper planes (~hi); far likewise.  One v_bitop3 each.
    const uint32_t mx = static_cast<uint32_t>(__float_as_int(ix) >> 31), my = static_cast<uint32_t>(__float_as_int(iy) >> 31), mz = static_cast<uint32_t>(__float_as_int(iz) >> 31);
    auto sel = [](uint32_t lo, uint32_t hi, uint32_t mk) { return static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(lo, hi, mk, 0x72)); };   // mk ? ~hi : lo
    const uint32_t nx[2] = {sel(q1.x, q2.z, mx), sel(q1.y, q2.w, mx)}, fx[2] = {sel(q2.z, q1.x, mx), sel(q2.w, q1.y, mx)};
    const uint32_t ny[2] = {sel(q1.z, q3.x, my), sel(q1.w, q3.y, my)}, fy[2] = {sel(q3.x, q1.z, my), sel(q3.y, q1.w, my)};
    const uint32_t nz[2] = {sel(q2.x, q3.z, mz), sel(q2.y, q3.w, mz)}, fz[2] = {sel(q3.z, q2.x, mz), sel(q3.w, q2.y, mz)};
    uint32_t dd[2][2];   // [word][parity]: t_out - t_in of children (4 word + parity, 4 word + parity + 2) in the (low, high) half
#pragma unroll
    for (int w = 0; w < 2; ++w) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            // bytes (0, 2) or (1, 3) of a row word as two 16-bit numbers: one v_and_b32 / one v_pk_lshrrev_b16
            auto un = [&](uint32_t v) {
                if (par == 0) return v & 0x00FF00FFu;
                uint32_t d;
                asm("v_pk_lshrrev_b16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(v));   // (op_sel_hi: the inline 8 shifts the high half too)
                return d;
            };
            const uint32_t tnx = pk_fma_h(anx, un(nx[w]), bnx), tny = pk_fma_h(any_, un(ny[w]), bny), tnz = pk_fma_h(anz, un(nz[w]), bnz);
            const uint32_t tfx = pk_fma_h(afx, un(fx[w]), bfx), tfy = pk_fma_h(afy, un(fy[w]), bfy), tfz = pk_fma_h(afz, un(fz[w]), bfz);
            const uint32_t t_in = pk_max_h(pk_max_h(tnx, tny), pk_max_h(tnz, tmin));
            const uint32_t t_out = pk_min_h(pk_min_h(tfx, tfy), pk_min_h(tfz, tbest));
            dd[w][par] = pk_sub_h(t_out, t_in);
        }
    }
    // sign bits -> mask of the children to DROP: child 4 w + par + 2 h sits in bit 15 + 16 h of dd[w][par]
    const uint32_t acc = ((dd[0][0] >> 15) & 0x00010001u) | ((dd[0][1] >> 14) & 0x00020002u) | ((dd[1][0] >> 11) & 0x00100010u) | ((dd[1][1] >> 10) & 0x00200020u);
    return ~(acc | (acc >> 14)) & 0xFFu;
}

// Scene::TraceRay's triangle loop exactly as the reference runs it (scene.cpp:116-120): every triangle, in index order,
// through Triangle::Intersect for this lane's ray; returns the closest-hit key (~0 = miss).  Wave-uniform control flow.
// Only the verification build calls it.
__device__ __forceinline__ unsigned long long brute_force_key(const RenderArgs &a, const Ray &q, float eps) {
    unsigned long long best = ~0ull;
    for (int i = 0; i < a.n_tri; ++i) {
        uint32_t orig;
        const float nd = exact_inside(a.exact + i, q, eps, orig);
        if (nd >= eps && nd < __builtin_inff()) {
            const unsigned long long k = (static_cast<unsigned long long>(ordered_bits(nd)) << 32) | static_cast<uint32_t>(i);
            best = k < best ? k : best;
        }
    }
    return best;
}

// emis_flag (wave-uniform, honoured only by instantiations with EMIS): search only where emitters are (RenderArgs::emis_*) --
// the closest hit among a SUPERSET of the emitters, which is all the caller needs to know whether the ray's closest hit can be
// one (see the last segment in the kernel).
// DYN (the adaptive-sampling instantiation of the two-rays-per-lane kernel): `two` (wave-uniform) says whether any lane's second
// ray slot is in use this pass; when it is not, everything per-ray about slot 1 is skipped by scalar branches and the segment
// costs about what a one-ray-per-lane segment costs (PT_SLOT_ON).  Without DYN the guards fold away.
#define PT_SLOT_ON(k) ((k) == 0 || k1_on)
template <bool ENV, bool EMIS, bool DYN = false, class Lds, class Stats>
__device__ __forceinline__ void closest_hit(const RenderArgs &a, Lds &lds, const Ray (&q)[Lds::kRays], const bool (&live)[Lds::kRays],
                                            const bool (&in_envelope)[Lds::kRays], int lane, float eps, float (&best)[Lds::kRays],
                                            int (&hit)[Lds::kRays], const ExactRec *(&hit_rec)[Lds::kRays], Stats &st, bool emis_flag = false,
                                            bool two = true) {
    constexpr int R = Lds::kRays;   // rays per lane: ray k of lane l has the id l + 64 k
    const bool emis_only = EMIS && emis_flag;
    const bool k1_on = !DYN || two;   // (slot 1 of a lane is never live when `two` is false: the guards only save its instructions)
    // `valid` below = rays that go through the culling hierarchy; live rays outside the envelope its margins were derived
    // for get every slot as a candidate instead (rare: see the caller).
    bool valid[R];
#pragma unroll
    for (int k = 0; k < R; ++k) valid[k] = live[k] && in_envelope[k];
    auto any_ray = [&](const bool (&b)[R]) {   // wave-uniform: does any ray of the wave-segment have the flag?
        bool x = b[0];
#pragma unroll
        for (int k = 1; k < R; ++k) x = x || b[k];
        return __any(x);
    };
    constexpr uint32_t kNodeStack = Lds::kNodeStack, kPairQueue = Lds::kPairQueue;
    // Small scenes (CullTables::big == false: at most kSmallSceneMaxTriangles triangles and slots, so 16 bits each): the closest-hit key carries the SLOT
    // below the original index, and shading reads the slot-ordered record the exact test has just pulled through the
    // caches.  Big scenes look the hit up in the table kept in the original order.
    constexpr bool kPackSlot = !Lds::kPrefilter;
    static_assert(kSmallSceneMaxTriangles < 32768, "packed (original index, slot) key: 16 bits each, slots padded to at most twice the triangles");
    ++st.w_segments;
    PT_STAMP(st, 0);   // everything since the last stamp: ray generation / loop control
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (!PT_SLOT_ON(k)) continue;
        const int id = lane + 64 * k;
        lds.best[id] = ~0ull;
        lds.ray[0][id] = q[k].ox; lds.ray[1][id] = q[k].oy; lds.ray[2][id] = q[k].oz;
        lds.ray[3][id] = q[k].dx; lds.ray[4][id] = q[k].dy; lds.ray[5][id] = q[k].dz;
        if constexpr (Lds::kPrefilter) {
            lds.rinv.v[0][id] = __builtin_amdgcn_rcpf(q[k].dx); lds.rinv.v[1][id] = __builtin_amdgcn_rcpf(q[k].dy); lds.rinv.v[2][id] = __builtin_amdgcn_rcpf(q[k].dz);
        }
    }
    uint32_t n_pairs = 0;   // wave-uniform fill level of lds.pairs
    wave_sync();

    // ---- 2. exact.  (ray, triangle) pairs are spread evenly over the lanes (a lane works on other lanes' rays).
    // The reference keeps, in triangle order, every accepted triangle with new_distance < distance
    // (scene.cpp:116-120, triangles.h:51), i.e. the lexicographic minimum of (new_distance, index) over the triangles
    // that pass Triangle::Intersect with eps <= new_distance < inf: that minimum is taken with one LDS atomic per pair.
    uint32_t n_filtered = 0;   // wave-uniform fill level of lds.filtered (big scenes)
    auto exact_round = [&](uint32_t e, bool active, uint32_t cnt) {
        ++st.w_exact_iters;
        st.n_exact += cnt;
        if (active) {
            const uint32_t src = (e >> 24) & Lds::kSrcMask, tri = e & 0xFFFFFFu;
            Ray r;
            r.ox = lds.ray[0][src]; r.oy = lds.ray[1][src]; r.oz = lds.ray[2][src];
            r.dx = lds.ray[3][src]; r.dy = lds.ray[4][src]; r.dz = lds.ray[5][src];
            uint32_t orig;   // the pair names a SLOT; the closest-hit key carries the original triangle index (tie-break)
            const float nd = exact_inside(a.exact_slot + tri, r, eps, orig);   // -inf unless stages B-D pass
            if (nd >= eps && nd < __builtin_inff())
                atomicMin(&lds.best[src], (static_cast<unsigned long long>(ordered_bits(nd)) << 32) | (kPackSlot ? (orig << 16) | tri : orig));
        }
    };
    auto drain_pairs = [&](uint32_t keep_below) {
        while (n_pairs > keep_below) {
            const uint32_t cnt = min(64u, n_pairs);
            n_pairs -= cnt;
            const bool active = static_cast<uint32_t>(lane) < cnt;
            const uint32_t e = active ? lds.pairs[n_pairs + lane] : 0u;
            if constexpr (!Lds::kPrefilter) {
                exact_round(e, active, cnt);
            } else {
                // A scene with thousands of small triangles yields ~8 sphere survivors per ray; the 30-instruction
                // barycentric test (conservative, like every cull) removes most of them before the 170-instruction exact
                // test, and the survivors are re-packed so that exact rounds stay full.
                bool keep = false;
                if (active) {
                    const uint32_t src = (e >> 24) & Lds::kSrcMask, tri = e & 0xFFFFFFu;
                    Ray r;
                    r.ox = lds.ray[0][src]; r.oy = lds.ray[1][src]; r.oz = lds.ray[2][src];
                    r.dx = lds.ray[3][src]; r.dy = lds.ray[4][src]; r.dz = lds.ray[5][src];
                    const float4 *rp = reinterpret_cast<const float4 *>(a.bary_all + tri);
                    const float4 c0 = rp[0], c1 = rp[1], c2 = rp[2];
                    CullRec rec;
                    rec.n[0] = c0.x; rec.n[1] = c0.y; rec.n[2] = c0.z; rec.w = c0.w;
                    rec.au[0] = c1.x; rec.au[1] = c1.y; rec.au[2] = c1.z; rec.cu = c1.w;
                    rec.av[0] = c2.x; rec.av[1] = c2.y; rec.av[2] = c2.z; rec.cv = c2.w;
                    keep = !cull_reject(rec, r, a.k1, a.k2, a.a_max_all, a.m0_all, a.t_guard_all) || (e >> 31) != 0u;   // bit 31: never filtered
#ifdef PT_DBG_NO_PREFILTER
                    keep = true;
#endif
                }
                const unsigned long long ball = __ballot(keep);
                if (keep) lds.filtered[n_filtered + lanes_below(ball)] = e;
                n_filtered += __builtin_popcountll(ball);
                wave_sync();
                if (n_filtered >= static_cast<uint32_t>(PT_BIG_EXACT_AT)) {
                    const uint32_t take = min(64u, n_filtered);
                    n_filtered -= take;
                    const bool on = static_cast<uint32_t>(lane) < take;
                    exact_round(on ? lds.filtered[n_filtered + lane] : 0u, on, take);
                    wave_sync();
                }
            }
            wave_sync();
        }
    };
    // Append one pair per set bit of `bits[k]` (bit j = triangle tri0 + j of this lane's ray k): a prefix sum over the lanes'
    // counts gives every lane its place in the queue (and the total).  `flag` is or-ed into the ray id (kUnfiltered).
    auto emit_lane_pairs = [&](const uint32_t (&bits)[R], uint32_t tri0, uint32_t flag, uint32_t excl, uint32_t total) {
        uint32_t pos = n_pairs + excl;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (!PT_SLOT_ON(k)) continue;
            emit_bits(lds.pairs, pos, bits[k], tri0 | ((static_cast<uint32_t>(lane + 64 * k) | flag) << 24));
            pos += __builtin_popcount(bits[k]);
        }
        n_pairs += total;
        wave_sync();
    };
    // the same for entries of other lanes' rays (root round): one mask per lane, ray id `src`
    auto emit_pairs = [&](uint32_t bits, uint32_t tri0, uint32_t src, uint32_t excl, uint32_t total) {
        emit_bits(lds.pairs, n_pairs + excl, bits, tri0 | (src << 24));
        n_pairs += total;
        wave_sync();
    };
    // for masks of any width: makes room first, and slices the masks if one batch could exceed the queue
    auto push_pairs_any = [&](const uint32_t (&bits)[R], uint32_t tri0, uint32_t flag) {
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (PT_SLOT_ON(k)) cnt += __builtin_popcount(bits[k]);
        const uint32_t incl = wave_scan_inclusive(cnt), total = wave_last(incl);
        if (total == 0) return;
        if (n_pairs + total > kPairQueue) drain_pairs(0);
        if (total <= kPairQueue) {
            emit_lane_pairs(bits, tri0, flag, incl - cnt, total);
            return;
        }
        constexpr uint32_t kSlice = 2u / R;   // bits of every ray's mask per slice: <= 128 pairs
        for (uint32_t lo = 0; lo < 32u; lo += kSlice) {
            uint32_t part[R], pc2 = 0;
            bool some = false;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                part[k] = PT_SLOT_ON(k) ? bits[k] & (((1u << kSlice) - 1u) << lo) : 0u;
                pc2 += __builtin_popcount(part[k]);
                some = some || part[k] != 0;
            }
            if (!__any(some)) continue;
            if (n_pairs + 128u > kPairQueue) drain_pairs(0);
            const uint32_t in2 = wave_scan_inclusive(pc2);
            emit_lane_pairs(part, tri0, flag, in2 - pc2, wave_last(in2));
        }
    };

    // Candidate masks of runs that need no tree walk (large-triangle words, runs of at most 8 small triangles) are
    // collected in a window of 32 consecutive triangle indices and published together: adjacent short runs (the walls
    // and the light of a room, split by the file order into three clusters) then cost one publication, not three.
    uint32_t pend[R], pend_tri0 = 0;    // per-ray bits; wave-uniform window start
#pragma unroll
    for (int k = 0; k < R; ++k) pend[k] = 0;
    bool pend_open = false;             // wave-uniform
    auto flush_pending = [&]() {
        if (pend_open) push_pairs_any(pend, pend_tri0, 0u);
#pragma unroll
        for (int k = 0; k < R; ++k) pend[k] = 0;
        pend_open = false;
    };
    auto add_pending = [&](const uint32_t (&bits)[R], uint32_t tri0, uint32_t width) {   // tri0, width wave-uniform
        if (pend_open && tri0 >= pend_tri0 && tri0 + width <= pend_tri0 + 32u) {
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (PT_SLOT_ON(k)) pend[k] |= bits[k] << (tri0 - pend_tri0);
        } else {
            flush_pending();
#pragma unroll
            for (int k = 0; k < R; ++k) pend[k] = PT_SLOT_ON(k) ? bits[k] : 0u;
            pend_tri0 = tri0;
            pend_open = true;
        }
    };

    // ---- 0. rays outside the envelope: all slots are candidates (padding slots carry NaN planes and are never accepted);
    // their pairs carry bit 30 = "skip the barycentric pre-filter", whose margins assume the envelope as well
    if constexpr (ENV) {
        bool far[R];
#pragma unroll
        for (int k = 0; k < R; ++k) far[k] = live[k] && !in_envelope[k];
        if (__builtin_expect(any_ray(far), 0)) {
            for (uint32_t base = 0; base < a.n_slots; base += 32u) {
                const uint32_t left = a.n_slots - base;
                const uint32_t all = left >= 32u ? 0xFFFFFFFFu : ((1u << left) - 1u);
                uint32_t bits[R];
#pragma unroll
                for (int k = 0; k < R; ++k) bits[k] = far[k] ? all : 0u;
                push_pairs_any(bits, base, Lds::kPrefilter ? kUnfiltered : 0u);
            }
        }
    }

    // ---- 1. cull
    const ConstF clusters = (ConstF)reinterpret_cast<uintptr_t>(a.clusters);
    const ConstF spheres = (ConstF)reinterpret_cast<uintptr_t>(a.spheres);
    const ConstF bary = (ConstF)reinterpret_cast<uintptr_t>(a.bary);
    constexpr int kDescWords = sizeof(ClusterDesc) / 4;
    for (int cl = 0; cl < a.n_clusters; ++cl) {
        const ConstF cp = clusters + kDescWords * cl;
        if (emis_only && cl < 32 && !((a.emis_clusters >> cl) & 1u)) continue;   // no emitter in this cluster
        const uint32_t first_tri = ((ConstU)cp)[4], n_tri = ((ConstU)cp)[5], kind = ((ConstU)cp)[6], off = ((ConstU)cp)[7];
        // (the large class is tested triangle by triangle anyway, and its bounding sphere is the scene's: nothing to gain from it)
        // (big scenes have no sphere-tree clusters -- all their small triangles sit under the box tree, pt_scene.cpp --: their
        // instantiations leave that code out, which spares them its registers)
        constexpr bool kSphereTrees = !Lds::kPrefilter;
        if (!kSphereTrees && kind == 0) continue;
        bool pc[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            pc[k] = false;
            if (PT_SLOT_ON(k)) pc[k] = (kSphereTrees && kind == 0) ? valid[k] & sphere_keep(cp[0], cp[1], cp[2], cp[3], q[k]) : valid[k];
        }
        if (!any_ray(pc)) continue;
        if (kSphereTrees && kind == 0) {
            // ---- small triangles: an 8-ary tree of bounding spheres, walked with a wave-wide LIFO of (ray, node) items
            const uint32_t n_levels = ((ConstU)cp)[8];
            const uint32_t top = n_levels - 1;
            if (lane < static_cast<int>(n_levels)) {
                lds.level_off[lane] = lane == 0 ? 0u : a.clusters[cl].level_off[lane - 1];
                lds.level_cnt[lane] = (n_tri + (1u << (3 * lane)) - 1u) >> (3 * lane);
            }
            wave_sync();
            uint32_t n_nodes = 0;   // wave-uniform fill level of lds.nodes
            uint32_t tmask[R];      // per ray: top-level nodes still to be pushed
#pragma unroll
            for (int k = 0; k < R; ++k) tmask[k] = 0;
            auto any_tmask = [&]() {
                uint32_t x = tmask[0];
#pragma unroll
                for (int k = 1; k < R; ++k) x |= tmask[k];
                return __any(x != 0);
            };
            // pushes one top-level node per ray and step until nothing is left or the stack is full (the rest follows once it has drained)
            auto push_top = [&](uint32_t top) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if (!PT_SLOT_ON(k)) continue;
                    while (__any(tmask[k] != 0)) {   // at most 8 x 64 R items > capacity: drained in the expansion loop before overflow
                        const bool has = tmask[k] != 0;
                        const unsigned long long ball = __ballot(has);
                        if (n_nodes + __builtin_popcountll(ball) > kNodeStack) return;
                        if (has) {
                            const uint32_t j = __builtin_ctz(tmask[k]);
                            tmask[k] &= tmask[k] - 1;
                            lds.nodes[n_nodes + lanes_below(ball)] = (static_cast<uint32_t>(lane + 64 * k) << Lds::kNodeSrcShift) | (top << Lds::kNodeLevShift) | j;
                        }
                        n_nodes += __builtin_popcountll(ball);
                    }
                }
            };
            // (a') Few rays near a small tree: skip its top level.  The (at most 16) lanes that passed the cluster sphere are
            // compacted, each gets 4 to 64 lanes, and together they test ALL nodes of the level below the top (at most 8
            // per lane) in one round -- instead of the wave-uniform top-level tests plus a round that is mostly empty.
            bool rooted = false;
            if (top >= 1) {
                unsigned long long pcb[R];
                uint32_t rcnt = 0;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    pcb[k] = 0;
                    if (!PT_SLOT_ON(k)) continue;
                    pcb[k] = __ballot(pc[k]);
                    rcnt += __builtin_popcountll(pcb[k]);
                }
                const uint32_t clev = top - 1;
                const uint32_t nchild = (n_tri + (1u << (3 * clev)) - 1u) >> (3 * clev);
                const uint32_t sh = rcnt <= 1u ? 6u : rcnt <= 2u ? 5u : rcnt <= 4u ? 4u : rcnt <= 8u ? 3u : rcnt <= 16u ? 2u : 0u;   // lanes per ray
                const uint32_t per = (nchild + (1u << sh) - 1u) >> sh;                                                           // nodes per lane
                if (rcnt <= 16u && per <= 8u) {
                    ++st.w_node_rounds;
                    {
                        uint32_t at = 0;   // the stack is empty here: the compacted ray ids go to its bottom
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            if (!PT_SLOT_ON(k)) continue;
                            if (pc[k]) lds.nodes[at + lanes_below(pcb[k])] = static_cast<uint32_t>(lane + 64 * k);
                            at += __builtin_popcountll(pcb[k]);
                        }
                    }
                    wave_sync();
                    // Lane `sub` of a ray's group takes nodes sub, sub + group, sub + 2 group, ...: one load instruction then
                    // reads consecutive records across the group (whole cache lines) instead of one line per lane.  When
                    // the nodes are the triangles themselves the slices stay contiguous (pairs are published as bit masks).
                    const uint32_t item = static_cast<uint32_t>(lane) >> sh, sub = static_cast<uint32_t>(lane) & ((1u << sh) - 1u);
                    const uint32_t c0 = clev == 0 ? sub * per : sub, cstep = clev == 0 ? 1u : (1u << sh);
                    uint32_t m = 0, src = 0;
                    if (item < rcnt) {
                        src = lds.nodes[item];
                        Ray r;
                        r.ox = lds.ray[0][src]; r.oy = lds.ray[1][src]; r.oz = lds.ray[2][src];
                        r.dx = lds.ray[3][src]; r.dy = lds.ray[4][src]; r.dz = lds.ray[5][src];
                        const float4 *cs = reinterpret_cast<const float4 *>(a.spheres) + off + lds.level_off[clev] + c0;
                        auto test = [&](auto per_c) {
                            constexpr uint32_t kPer = decltype(per_c)::value;
#pragma unroll
                            for (uint32_t i = 0; i < kPer; ++i) {   // no bounds test per sphere: the table is padded (pt_scene.cpp) ...
                                const float4 sp = cs[i * cstep];
                                m |= sphere_keep(sp.x, sp.y, sp.z, sp.w, r) ? (1u << i) : 0u;
                            }
                            // ... and the results beyond the level's last node are dropped here
                            const uint32_t left = nchild > c0 ? nchild - c0 : 0u;
                            const uint32_t n_ok = min((left + cstep - 1u) >> (clev == 0 ? 0u : sh), kPer);
                            m &= (1u << n_ok) - 1u;
                        };
                        if (per == 1u) test(std::integral_constant<uint32_t, 1>());
                        else if (per == 2u) test(std::integral_constant<uint32_t, 2>());
                        else if (per <= 4u) test(std::integral_constant<uint32_t, 4>());
                        else test(std::integral_constant<uint32_t, 8>());
                    }
                    wave_sync();   // every lane has read its ray's lane number before the stack is written
                    const uint32_t mc = __builtin_popcount(m);
                    const uint32_t incl = wave_scan_inclusive(mc), tot = wave_last(incl);
                    if (clev == 0) {   // the level below the top is the triangles themselves
                        if (n_pairs + tot > kPairQueue) drain_pairs(0);
                        if (tot <= kPairQueue) {
                            emit_pairs(m, first_tri + c0, src, incl - mc, tot);
                            rooted = true;
                        }
                    } else if (tot <= kNodeStack) {
                        // bit j = node c0 + j * cstep of level clev (cstep = 1 << sh here)
                        emit_bits(lds.nodes, n_nodes + incl - mc, m, (src << Lds::kNodeSrcShift) | (clev << Lds::kNodeLevShift) | c0, sh);
                        n_nodes += tot;
                        wave_sync();
                        rooted = true;
                    }
                    // (more survivors than the queue holds: fall through to the general path below)
                }
            }
            if (!rooted) {
            // (a) wave-uniform: every lane against the (at most 8) top-level spheres, records in SGPRs
            const uint32_t top_off = top == 0 ? 0u : ((ConstU)cp)[8 + top];
            const uint32_t top_cnt = (n_tri + (1u << (3 * top)) - 1u) >> (3 * top);
            const ConstF tp = spheres + 4 * (static_cast<size_t>(off) + top_off);
            for (uint32_t j = 0; j < top_cnt; ++j) {
                const float sx = tp[4 * j], sy = tp[4 * j + 1], sz = tp[4 * j + 2], sr = tp[4 * j + 3];
#pragma unroll
                for (int k = 0; k < R; ++k)
                    if (PT_SLOT_ON(k)) tmask[k] |= sphere_keep(sx, sy, sz, sr, q[k]) ? (1u << j) : 0u;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) tmask[k] = pc[k] ? tmask[k] : 0u;
            PT_STAMP(st, 1);   // cluster + top-level sphere tests
            if (top == 0) {
                add_pending(tmask, first_tri, n_tri);   // the run has at most 8 triangles
#pragma unroll
                for (int k = 0; k < R; ++k) tmask[k] = 0;
            } else {
                push_top(top);
                wave_sync();
            }
            }
            // (b) lane-balanced expansion: lane l takes the l-th item from the top of the stack, tests the node's 8
            // children against that item's ray and pushes the survivors (tree nodes back on the stack, triangles as
            // (ray, triangle) pairs).  A round is committed only for the top k lanes whose children fit.
            while (n_nodes > 0 || any_tmask()) {
                if (n_nodes == 0) {   // top-level items that did not fit earlier
                    push_top(top);
                    wave_sync();
                }
                ++st.w_node_rounds;
                const uint32_t cnt = min(64u, n_nodes);
                // A round with few items spreads each item's 8 children over 2, 4 or 8 lanes (wave-uniform choice).
                uint32_t shift = cnt <= 8u ? 3u : cnt <= 16u ? 2u : cnt <= 32u ? 1u : 0u;
                uint32_t m8, src, level, child0, keep, packed, incl, tot;
                bool leaf;
                for (;;) {
                    m8 = 0; src = 0; level = 1; child0 = 0;
                    const uint32_t item = static_cast<uint32_t>(lane) >> shift, sub = static_cast<uint32_t>(lane) & ((1u << shift) - 1u);
                    if (item < cnt) {
                        const uint32_t e = lds.nodes[n_nodes - 1 - item];
                        src = e >> Lds::kNodeSrcShift;
                        level = (e >> Lds::kNodeLevShift) & 7u;
                        child0 = (e & ((1u << Lds::kNodeLevShift) - 1u)) * kFan;   // index of the first child within level-1
                        Ray r;
                        r.ox = lds.ray[0][src]; r.oy = lds.ray[1][src]; r.oz = lds.ray[2][src];
                        r.dx = lds.ray[3][src]; r.dy = lds.ray[4][src]; r.dz = lds.ray[5][src];
                        const float4 *cs = reinterpret_cast<const float4 *>(a.spheres) + off + lds.level_off[level - 1] + child0;
                        auto test = [&](auto per_c) {
                            constexpr int per = decltype(per_c)::value;
#pragma unroll
                            for (int i = 0; i < per; ++i) {
                                const uint32_t c = sub + (static_cast<uint32_t>(i) << shift);   // interleaved: the item's lanes read consecutive records
                                const float4 sp = cs[c];
                                m8 |= sphere_keep(sp.x, sp.y, sp.z, sp.w, r) ? (1u << c) : 0u;
                            }
                        };
                        if (shift == 0u) test(std::integral_constant<int, 8>());
                        else if (shift == 1u) test(std::integral_constant<int, 4>());
                        else if (shift == 2u) test(std::integral_constant<int, 2>());
                        else test(std::integral_constant<int, 1>());
                        const uint32_t real = lds.level_cnt[level - 1];   // padding spheres are never children
                        const uint32_t left = real > child0 ? real - child0 : 0u;
                        m8 &= left >= 8u ? 0xFFu : ((1u << left) - 1u);
                    }
                    leaf = level == 1;   // children are triangles
                    uint32_t kids = __builtin_popcount(m8);
                    keep = cnt;          // items [0, keep) (from the top of the stack) are committed this round
                    // one prefix sum for both queues: tree nodes count in the low half of the word, triangles in the high half
                    packed = leaf ? kids << 16 : kids;
                    incl = wave_scan_inclusive(packed);
                    tot = wave_last(incl);
                    if (n_pairs + (tot >> 16) > kPairQueue) drain_pairs(0);
                    if (n_pairs + (tot >> 16) <= kPairQueue && n_nodes - cnt + (tot & 0xFFFFu) <= kNodeStack) break;
                    if (shift != 0u) { shift = 0u; continue; }   // rare: redo the round one lane per item
                    // Rare: not everything fits.  Commit the longest prefix of lanes (= the top of the stack) whose
                    // children do; the other items stay where they are.  If not even the top item fits it is committed
                    // anyway: its (at most 8) children replace it, and since they are one level deeper the stack can
                    // outgrow kNodeStack by at most 7 per level, which is what the 64 slots of slack are for.
                    const bool fits = static_cast<uint32_t>(lane) < cnt && n_pairs + (incl >> 16) <= kPairQueue &&
                                      (n_nodes - (lane + 1)) + (incl & 0xFFFFu) <= kNodeStack;
                    const unsigned long long fb = __ballot(fits);
                    keep = (fb == ~0ull) ? 64u : static_cast<uint32_t>(__builtin_ctzll(~fb));
                    if (keep == 0) keep = 1;
                    ++st.w_partial;
                    if (static_cast<uint32_t>(lane) >= keep) m8 = 0;
                    kids = __builtin_popcount(m8);
                    packed = leaf ? kids << 16 : kids;
                    incl = wave_scan_inclusive(packed);
                    tot = wave_last(incl);
                    break;
                }
                n_nodes -= keep;
                wave_sync();
                // node children back on the stack, triangle children into the pair queue: one pass
                {
                    const uint32_t excl = incl - packed;
                    emit_bits(leaf ? lds.pairs : lds.nodes, leaf ? n_pairs + (excl >> 16) : n_nodes + (excl & 0xFFFFu), m8,
                              leaf ? ((first_tri + child0) | (src << 24)) : ((src << Lds::kNodeSrcShift) | ((level - 1) << Lds::kNodeLevShift) | child0));
                    n_nodes += tot & 0xFFFFu;
                    n_pairs += tot >> 16;
                }
                wave_sync();
            }
            PT_STAMP(st, 2);   // balanced tree walk
        } else {
            // ---- large triangles: barycentric cull, wave-uniform over the triangles.
            // The margins go to VGPRs here: a VALU instruction can name only one SGPR, so an SGPR-resident
            // constant next to an SGPR-resident triangle coefficient would cost a v_mov per use.
            float k1 = a.k1, k2 = a.k2, a_max = a.a_max, m0 = a.m0, m0q = a.m0_quad, t_guard = a.t_guard;
            asm volatile("" : "+v"(k1), "+v"(k2), "+v"(a_max), "+v"(m0), "+v"(m0q), "+v"(t_guard));
            const int n_words = static_cast<int>((n_tri + kChunk - 1) / kChunk);
            for (int w = 0; w < n_words; ++w) {
                const uint32_t left = n_tri - kChunk * w;
                const ConstF bp = bary + 12 * (static_cast<size_t>(off) + kChunk * w);
                const uint32_t quads = w < kMaxLevels - 1 ? ((ConstU)cp)[9 + w] : 0u;   // bit k: slots k, k+1 are one quad record
                uint32_t m[R];
#pragma unroll
                for (int k = 0; k < R; ++k) m[k] = 0;
                const uint32_t cnt32 = min(left, 32u);
                const uint32_t pair_bits = 0x55555555u & (cnt32 >= 32u ? 0xFFFFFFFFu : ((1u << cnt32) - 1u));
                // one record (wave-uniform, in scalar registers) against this lane's R rays
                auto quad = [&](uint32_t k0) {
                    const CullRec rec = load_cull(bp + 12 * k0);
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        if (PT_SLOT_ON(k)) m[k] |= (~cull_reject_quad(rec, q[k], k1, k2, a_max, m0q, t_guard) & 3u) << k0;
                };
                auto single = [&](uint32_t slot) {
                    const CullRec rec = load_cull(bp + 12 * slot);
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        if (PT_SLOT_ON(k)) m[k] |= cull_reject(rec, q[k], k1, k2, a_max, m0, t_guard) ? 0u : (1u << slot);
                };
                if (emis_only && n_words == 1) {   // only the records that hold an emitter (the light of a room: one quad)
                    for (uint32_t rest = (a.emis_large_w0 | (a.emis_large_w0 >> 1)) & pair_bits; rest != 0; rest &= rest - 1) {
                        const uint32_t k0 = __builtin_ctz(rest);
                        if ((quads >> k0) & 1u) {
                            quad(k0);
                        } else {
                            single(k0);
                            single(k0 + 1);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k) m[k] &= a.emis_large_w0;
                } else
                if (quads == pair_bits) {   // every record of the word is a quad (the walls of a room): no per-record dispatch
                    // Big scenes take the records two at a time -- both records' scalar loads named before the first one's
                    // arithmetic.  The compiler still waits per record, but this form leaves the box-tree kernel with 31 instead of
                    // 37 spilled scalar registers: +1.7 % on the x64 replica (ab58).  For the small-scene kernels it changes nothing
                    // in pairs and costs 5 % in fours (spills).  The table is padded by more than a record: the load past an odd
                    // class's end is harmless.
                    constexpr int kWallGroup = Lds::kPrefilter ? 2 : 1;
                    if constexpr (kWallGroup > 1) {
                        for (uint32_t k0 = 0; k0 < cnt32; k0 += 2 * kWallGroup) {
                            CullRec r[kWallGroup];
#pragma unroll
                            for (int g = 0; g < kWallGroup; ++g) r[g] = load_cull(bp + 12 * (k0 + 2 * g));
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int g = 0; g < kWallGroup; ++g) {
                                if (k0 + 2 * g < cnt32) {
#pragma unroll
                                    for (int k = 0; k < R; ++k) m[k] |= (~cull_reject_quad(r[g], q[k], k1, k2, a_max, m0q, t_guard) & 3u) << (k0 + 2 * g);
                                }
                            }
                        }
                    } else {
                        for (uint32_t k0 = 0; k0 < cnt32; k0 += 2) quad(k0);
                    }
                } else
                for (uint32_t k0 = 0; k0 < cnt32; k0 += 2) {   // records are padded to whole words
                    if ((quads >> k0) & 1u) {   // wave-uniform
                        quad(k0);
                    } else {
                        single(k0);
                        single(k0 + 1);
                    }
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    m[k] = pc[k] ? m[k] : 0u;
                    m[k] &= left >= 32u ? 0xFFFFFFFFu : ((1u << left) - 1u);
                }
                PT_STAMP(st, 3);   // barycentric cull of the large triangles
                add_pending(m, first_tri + kChunk * w, min(left, 32u));
                PT_STAMP(st, 4);   // pair publication
            }
        }
    }
    PT_STAMP(st, 1);
    if constexpr (Lds::kPrefilter) {
        // ---- big scenes: ONE box tree over all small triangles, walked with the wave-wide LIFO of (ray, node) items.
        // The large class above went first and is tested right away, so that lds.best already holds a hit (the wall behind
        // everything, in a closed room) when the walk starts: a node whose box the ray enters beyond its best hit so far is
        // dropped, and because pairs are tested as soon as 64 are waiting, closer hits keep shrinking the rest of the walk.
        if (a.n_bvh > 0 && !(emis_only && a.emis_bvh == 0u)) {
            flush_pending();
            drain_pairs(0);
            if (n_filtered > 0) {
                const bool active = static_cast<uint32_t>(lane) < n_filtered;
                exact_round(active ? lds.filtered[lane] : 0u, active, n_filtered);
                n_filtered = 0;
                wave_sync();
            }
            uint32_t n_nodes;
            {
                n_nodes = 0;
#pragma unroll
                for (int k = 0; k < R; ++k) {   // the root, for every live ray
                    const unsigned long long vb = __ballot(valid[k]);
                    if (valid[k]) lds.nodes[n_nodes + lanes_below(vb)] = static_cast<uint32_t>(lane + 64 * k) << Lds::kNodeSrcShift;
                    n_nodes += __builtin_popcountll(vb);
                }
                wave_sync();
            }
            while (n_nodes > 0) {
                ++st.w_node_rounds;
                const uint32_t cnt = min(64u, n_nodes);
                // (PT_BOX_SPREAD, off: a round with few items -- the tail of every walk: 1.3 of the x64 replica's 9.1 rounds per
                // wave-segment hold at most 32 -- spreads each item's 8 children over 2, 4 or 8 lanes, like the sphere-tree walk)
                uint32_t shift = PT_BOX_SPREAD ? (cnt <= 8u ? 3u : cnt <= 16u ? 2u : cnt <= 32u ? 1u : 0u) : 0u;
                uint32_t m8, src, base, kids, keep, packed, incl, tot;
                bool leaf;
                for (;;) {
                m8 = 0; src = 0; base = 0;
                leaf = false;
                const uint32_t item = static_cast<uint32_t>(lane) >> shift, sub = static_cast<uint32_t>(lane) & ((1u << shift) - 1u);
                if (item < cnt) {
                    const uint32_t e = lds.nodes[n_nodes - 1 - item];
                    src = e >> Lds::kNodeSrcShift;
                    const uint32_t node = e & ((1u << Lds::kNodeSrcShift) - 1u);
                    Ray r;
                    r.ox = lds.ray[0][src]; r.oy = lds.ray[1][src]; r.oz = lds.ray[2][src];
                    r.dx = lds.ray[3][src]; r.dy = lds.ray[4][src]; r.dz = lds.ray[5][src];
                    const uint32_t bh = reinterpret_cast<const uint32_t *>(&lds.best[src])[1];   // high word: ordered bits of t
#ifdef PT_DBG_NO_PRUNE
                    const float t_best = bh == 0x12345u ? 0.0f : __builtin_inff();
#else
                    const float t_best = bh == 0xFFFFFFFFu ? __builtin_inff() : from_ordered_bits(bh);
#endif
                    const uint4 *np = reinterpret_cast<const uint4 *>(a.bvh + node);
                    const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                    const float ix = lds.rinv.v[0][src], iy = lds.rinv.v[1][src], iz = lds.rinv.v[2][src];
#if PT_BOX_F16
                    static_assert(!PT_BOX_SPREAD, "the half-precision test handles a node's eight children in one lane");
                    if (shift == 0u) m8 = box_children_kept_h(q0, q1, q2, q3, r, ix, iy, iz, t_best, a.bvh_err);
#elif PT_BOX_MIX
                    static_assert(!PT_BOX_SPREAD, "the mixed-precision form handles a node's eight children in one lane");
                    if (shift == 0u) m8 = box_children_kept_mix(q0, q1, q2, q3, r, ix, iy, iz, t_best, a.bvh_err);
#else
                    if (shift == 0u) m8 = box_children_kept<8>(q0, q1, q2, q3, r, ix, iy, iz, t_best, a.bvh_err);
#endif
                    else if (shift == 1u) m8 = box_children_kept<4>(q0, q1, q2, q3, r, ix, iy, iz, t_best, a.bvh_err, sub);
                    else if (shift == 2u) m8 = box_children_kept<2>(q0, q1, q2, q3, r, ix, iy, iz, t_best, a.bvh_err, sub);
                    else m8 = box_children_kept<1>(q0, q1, q2, q3, r, ix, iy, iz, t_best, a.bvh_err, sub);
                    m8 &= (2u << ((q0.w >> 8) & 7u)) - 1u;   // children that exist
                    leaf = ((q0.w >> 11) & 1u) != 0u;                       // BvhNode::meta
                    base = leaf ? (q0.w >> 12) * kFan : (q0.w >> 12);      // a leaf's first slot / an inner node's first child
                }
                kids = __builtin_popcount(m8);
                keep = cnt;
                // Children of inner nodes go back on the stack, triangles of leaves into the pair queue.  One prefix sum serves both
                // (inner lanes count in the low half of the word, leaf lanes in the high half): its last lane says whether everything
                // fits, its other lanes where each lane's entries go.
                packed = leaf ? kids << 16 : kids;
                incl = wave_scan_inclusive(packed);
                tot = wave_last(incl);
                if (n_pairs + (tot >> 16) > kPairQueue) drain_pairs(0);
                if (n_pairs + (tot >> 16) <= kPairQueue && n_nodes - cnt + (tot & 0xFFFFu) <= kNodeStack) break;
                if (shift != 0u) { shift = 0u; continue; }   // rare: redo the round one lane per item
                {
                    // Rare: not everything fits.  Commit the longest prefix of lanes (= the top of the stack) whose children do;
                    // the top item always commits (its children are one level deeper; the 64 slots of slack absorb them).
                    const bool fits = static_cast<uint32_t>(lane) < cnt && n_pairs + (incl >> 16) <= kPairQueue &&
                                      (n_nodes - (lane + 1)) + (incl & 0xFFFFu) <= kNodeStack;
                    const unsigned long long fb = __ballot(fits);
                    keep = (fb == ~0ull) ? 64u : static_cast<uint32_t>(__builtin_ctzll(~fb));
                    if (keep == 0) keep = 1;
                    ++st.w_partial;
                    if (static_cast<uint32_t>(lane) >= keep) m8 = 0;
                    kids = __builtin_popcount(m8);
                    packed = leaf ? kids << 16 : kids;
                    incl = wave_scan_inclusive(packed);
                    tot = wave_last(incl);
                }
                break;
                }
                n_nodes -= keep;
                wave_sync();   // every lane has read its item before the stack is written
                // (Measured and dropped: pushing the children of the far half of a node first, right-aligned over the push steps so
                // that the next round is the near front of all rays.  Pruning is already within 10 % of what knowing the final hit
                // from the start would give -- the walls are tested first and pairs as soon as 64 wait -- and the ordering's 25
                // instructions per round cost as much as it saved.)
                {
                    const uint32_t excl = incl - packed;
                    emit_bits(leaf ? lds.pairs : lds.nodes, leaf ? n_pairs + (excl >> 16) : n_nodes + (excl & 0xFFFFu), m8,
                              leaf ? (base | (src << 24)) : ((src << Lds::kNodeSrcShift) | base));
                    n_nodes += tot & 0xFFFFu;
                    n_pairs += tot >> 16;
                }
                wave_sync();
                PT_STAMP(st, 2);   // box-tree rounds
                if (n_pairs >= 64u) drain_pairs(63);
                PT_STAMP(st, 6);   // pre-filter + exact rounds inside the walk
            }
            PT_STAMP(st, 2);
        }
    }
    flush_pending();
    drain_pairs(0);
    if constexpr (Lds::kPrefilter) {
        if (n_filtered > 0) {
            const bool active = static_cast<uint32_t>(lane) < n_filtered;
            exact_round(active ? lds.filtered[lane] : 0u, active, n_filtered);
            n_filtered = 0;
            wave_sync();
        }
    }
    PT_STAMP(st, 5);   // exact rounds
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (!PT_SLOT_ON(k)) {   // (nobody's ray)
            hit[k] = -1;
            hit_rec[k] = a.exact;
            best[k] = __builtin_inff();
            continue;
        }
        const unsigned long long key = lds.best[lane + 64 * k];
        const uint32_t key_lo = static_cast<uint32_t>(key);
        hit[k] = (key == ~0ull) ? -1 : static_cast<int>(kPackSlot ? key_lo >> 16 : key_lo);
        hit_rec[k] = kPackSlot ? a.exact_slot + (key_lo & 0xFFFFu) : a.exact + key_lo;   // dereferenced only if hit >= 0
#ifdef PT_VERIFY_BRUTE
        // Verification build (libpt_verify.so, never the shipped library): Scene::TraceRay's loop as written -- every
        // triangle through Triangle::Intersect for this lane's own ray -- and a comparison of the two closest hits.
        {
            const unsigned long long brute = brute_force_key(a, q[k], eps);
            st.v_checked += static_cast<uint32_t>(__builtin_popcountll(__ballot(live[k])));
            const unsigned long long mine = (key == ~0ull) ? ~0ull : ((key & 0xFFFFFFFF00000000ull) | static_cast<uint32_t>(hit[k]));
            st.v_bad += static_cast<uint32_t>(__builtin_popcountll(__ballot(live[k] && brute != mine)));
            if (live[k] && brute != mine && a.stats) {   // one example for the host to print (any of them)
                a.stats[11] = brute;
                a.stats[12] = mine;
                a.stats[13] = (static_cast<unsigned long long>(__float_as_uint(q[k].ox)) << 32) | __float_as_uint(q[k].oy);
                a.stats[14] = (static_cast<unsigned long long>(__float_as_uint(q[k].oz)) << 32) | __float_as_uint(q[k].dx);
                a.stats[15] = (static_cast<unsigned long long>(__float_as_uint(q[k].dy)) << 32) | __float_as_uint(q[k].dz);
            }
        }
#endif
        best[k] = (key == ~0ull) ? __builtin_inff() : from_ordered_bits(static_cast<uint32_t>(key >> 32));
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------------------------
// SKY = the scene has a skybox (scene.cpp:126-154).  A separate instantiation: the lookup's double arithmetic raises
// the register peak, and the no-skybox kernel (every BASELINE configuration) should not pay for it.
// BIG = deep work queues for scenes with thousands of triangles (see SmallQueues / BigQueues).
// STATS = the caller asked for pt_render_stats.
// ENV = some triangle of the scene can be "hit" outside the envelope the culling margins are derived for (near-degenerate
// triangles; CullTables::may_leave_envelope): every segment's origin is then checked.  A compile-time choice because the
// mere presence of the rare path costs the common scenes 2 % (measured), whether or not it ever runs.
#ifndef PT_RAYS_PER_LANE
#define PT_RAYS_PER_LANE 2   // pixels (rays) per lane of the small-scene, statistics-free, skybox-free kernel; 1 = one 8 x 8 tile per wave
#endif
constexpr int kMaxBatchPass = 32766;   // adaptive instantiations: a pixel's next pass (at most one past the launch's last) << 1 | a flag in 16 bits
#ifndef PT_ADAPT4_DYN
#define PT_ADAPT4_DYN 0   // 32 x 8 adaptive kernel: 1 = batches of at most 64 switch the second ray slots off (scalar branches around every per-ray piece)
#endif
#ifndef PT_ADAPT_TWO_AT
#define PT_ADAPT_TWO_AT 100   // adaptive kernels: pixels with a pass to run from which a batch takes up to 128, two per lane (80 ... 112 within 1 %, adapt3)
#endif
#ifndef PT_BIG_RAYS_PER_LANE
#define PT_BIG_RAYS_PER_LANE 1   // the same for the statistics-free, skybox-free big-scene kernel
#endif
// rays per lane of an instantiation: the launch geometry (tile width) follows from it on the host as well
template <bool SKY, bool BIG, bool STATS>
constexpr int rays_per_lane() { return (!SKY && !STATS) ? (BIG ? PT_BIG_RAYS_PER_LANE : PT_RAYS_PER_LANE) : 1; }

// NARROW = the statistics-free small-scene kernel with ONE pixel per lane (8 x 8 tiles): for launches with too few pixels to
// fill the chip with 16 x 8 tiles (small previews, thin row bands) -- half as many waves would leave wave slots empty.
// waves per SIMD an instantiation is compiled for
template <bool SKY, bool BIG, bool STATS, bool ENV, bool NARROW>
constexpr int integrator_waves() {
    return SKY ? ((BIG && STATS && ENV) ? PT_BIG_WAVES /* (that one spills at 5) */ : PT_SKY_WAVES)
               : BIG ? PT_BIG_WAVES : (STATS || (!NARROW && rays_per_lane<SKY, BIG, STATS>() > 1)) ? PT_WAVES_PER_SIMD - 1 : PT_WAVES_PER_SIMD;
}
// ADAPT = the two-pixel kernel for launches with adaptive sampling on (error >= 0), with tiles of 64 ADAPT pixels (2: 16 x 8,
// 4: 32 x 8): see "Batches" in the pass loop.
template <bool SKY, bool BIG, bool STATS, bool ENV, bool NARROW = false, int ADAPT = 0>
__global__ __launch_bounds__(kBlock, (integrator_waves<SKY, BIG, STATS, ENV, NARROW>())) void integrate_kernel(const RenderArgs a) {
    static_assert(!NARROW || (!SKY && !STATS), "only the statistics-free, skybox-free kernels have a narrow variant");
    constexpr int R = NARROW ? 1 : rays_per_lane<SKY, BIG, STATS>();   // pixels per lane: the wave's tile is kTileW * R x kTileH
    static_assert(ADAPT == 0 || ((ADAPT == 2 || ADAPT == 4) && (R == 2 || BIG) && !STATS && !SKY), "batches of the tile's pixels (a pixel's number takes 8 bits): the two-pixel kernel and the box-tree kernel");
    // REGEN = path regeneration: a lane whose path has ended starts its pixel's NEXT pass at once instead of idling until the
    // longest path of the wave is done.  The skybox instantiations run this way: a scene with a skybox is an open scene, most
    // paths end on their first or second segment (scene.cpp:125-155: a miss ends the path) -- Tor.obj without its back wall has
    // 32.7 of 64 rays alive per wave-segment at -MRR 8 (profiles/r04_open_scene_probe_before.jsonl), the closed room 63.6.  The
    // frame cannot change: a pixel's passes still run in order on its own lane (its contributions are added in pass order, its
    // adaptive-sampling answer is the one main.cpp:118-125 computes before that pass), and the RNG counter is (pixel, pass,
    // segment) whatever the other lanes are doing.  The closed-room kernels keep the pass loop: there regeneration gains
    // nothing and would cost the last-segment filter, which needs the wave's rays to reach their last segment together.
    constexpr bool REGEN = SKY;
    static_assert(!REGEN || !ADAPT, "the compacting instantiation keeps the pass loop");
    // kDynSlots: batches of at most 64 pixels run with the second ray slots switched off (scalar branches around every per-ray piece of
    // the search and of shading).  The 16 x 8 kernel has them -- a third of its batches are such -- the 32 x 8 kernel does not: one batch
    // in eleven is, and the branches cost every batch 3-4 % (1080p 56.0 -> 54.1 ms, 3840 x 2160 215.5 -> 207.2, r04_ab_logs.txt adapt5).
    constexpr bool kDynSlots = ADAPT != 0 && R == 2 && (ADAPT != 4 || PT_ADAPT4_DYN);
    constexpr int kOwn = ADAPT ? ADAPT : R;   // pixels of the tile per lane: pixel j of the tile = column (j % 8) + 8 (j / 64), row (j % 64) / 8
    constexpr int kTW = kTileW * kOwn;
    __shared__ WaveLds<std::conditional_t<BIG, std::conditional_t<ADAPT != 0, BigQueuesAdapt<ADAPT>, BigQueues>, std::conditional_t<(R > 1), SmallQueues2, SmallQueues>>, R, ADAPT> lds;   // one wave per workgroup: all wave-private

    const int lane = threadIdx.x;
    if constexpr (decltype(lds)::kMatCache > 0) {
        if (a.n_mats <= decltype(lds)::kMatCache) {
            const float4 *src = reinterpret_cast<const float4 *>(a.mats);
            for (int i = lane; i < 3 * a.n_mats; i += 64) lds.mat.v[i] = src[i];
            wave_sync();
        }
    }
    // Work item = (pixel tile, chunk of passes), claimed from a ticket counter in chunk-major order: all tiles' first
    // chunk, then all tiles' second chunk, ...  Cutting the pass range into chunks gives the tail of the launch small
    // items to balance with (at 1080p x 64 spp one tile per wave left the last of 5.3 rounds a quarter full).
    // A chunk may only start once the tile's previous chunk has published its accumulators; because tickets are
    // taken in execution order, that chunk was claimed earlier by a wave that is running or done, so the wait below
    // cannot deadlock, and with thousands of tiles between two chunks of one tile it practically never waits.
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(&a.sched[0], 1u);
    item = __builtin_amdgcn_readfirstlane(item);
    const uint32_t tile = item % a.n_tiles, chunk = item / a.n_tiles;
    // chunk_passes > 0: equal chunks.  chunk_passes == 0: chunk c covers passes [P - (P >> 2c), P - (P >> 2(c+1))) of the launch's P
    // (3/4 of what is left each time, the last chunk takes the rest): the tail of the launch is balanced with small items while a
    // tile's accumulators make few round trips to memory (5 chunks at 256 passes: 192 + 48 + 12 + 3 + 1).
    int pass_first, pass_last;
    if (a.chunk_passes > 0) {
        pass_first = a.pass_begin + static_cast<int>(chunk) * a.chunk_passes;
        pass_last = min(a.pass_begin + a.pass_count, pass_first + a.chunk_passes);
    } else {
        pass_first = a.pass_begin + (a.pass_count - (a.pass_count >> (2u * chunk)));
        pass_last = a.pass_begin + (chunk + 1 < a.n_chunks ? a.pass_count - (a.pass_count >> (2u * (chunk + 1u))) : a.pass_count);
    }
    if (chunk > 0) {
        if (lane == 0) {
            while (__hip_atomic_load(&a.sched[1 + tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunk) __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // one poll, one agent-scope acquire, then plain loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // a.blocks_x counts tiles of THIS instantiation's width (the host asks integrator_tile_width)
    // A band is rows [row_begin, row_end) of the image -- or, with row_stride n > 1, every n-th TILE ROW (kTileH image rows) from row_begin
    // on, packed in the band's planes: tile row j of the band is image rows row_begin + j n kTileH ..., plane rows j kTileH ...  (the
    // interleaved split of a frame over several devices, pt_frame.cpp).  acc_y0 = the tile's first row in the band's planes; its first
    // image row follows from it (and is what the camera ray and the RNG's pixel index take).
    int tile_x0 = static_cast<int>(tile % a.blocks_x) * kTW, acc_y0 = static_cast<int>(tile / a.blocks_x) * kTileH;   // wave-uniform
    if constexpr (ADAPT != 0) {   // (the division runs on the vector unit: say that its results are scalars, or the 16 x 8 kernel spills one of them)
        tile_x0 = __builtin_amdgcn_readfirstlane(tile_x0);
        acc_y0 = __builtin_amdgcn_readfirstlane(acc_y0);
    }
    const int tile_y0 = a.row_begin + acc_y0 * a.row_stride;
    // pixel k of the lane: column (lane % 8) + 8 k of the tile, row lane / 8; its slot in the wave's LDS arrays is lane + 64 k
    int x[R];
    const int y = tile_y0 + (lane / kTileW);
    bool in_image[R];
    uint32_t gpix[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        x[k] = tile_x0 + (lane % kTileW) + kTileW * k;
        in_image[k] = x[k] < a.width && y < a.row_end;
        gpix[k] = static_cast<uint32_t>(static_cast<size_t>(y) * a.width + x[k]);
    }

    // The tile's accumulators live in LDS for the whole launch (read once, written once: exactly the algorithmic
    // 56 B/pixel of HBM traffic).  Keeping them in VGPRs costs a wave per SIMD; read-modify-writing them in HBM at every
    // emitter hit moved 10x the algorithmic bytes, because each hit touches three sparse cache lines.
    // Where the tile's accumulators live.  Small scenes, one ray per lane: in LDS for the whole work item (read once, written
    // once: exactly the algorithmic 56 B/pixel of HBM traffic; keeping them in VGPRs costs a wave per SIMD).  Two rays per lane,
    // and big scenes: in memory, read-modify-written when a path reaches an emitter (1 % of the samples): the LDS they would
    // take (3.5 KB / 1.8 KB per wave) is what separates 4 from 5 waves per SIMD in the first and 5 from 6 in the second, worth
    // 8 % and 3 % of the frame time (profiles/r03_ab_logs.txt, ab52 / ab54), while the extra traffic -- three sparse cache lines
    // in and out per contribution, ~0.4 GB per 256-spp frame -- is 0.08 % of the HBM peak.
    constexpr bool kAccInLds = decltype(lds)::kAccInLds;
    if constexpr (kAccInLds) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (in_image[k]) {
            const size_t p = static_cast<size_t>(acc_y0 + lane / kTileW) * a.width + x[k];
            const int id = lane + 64 * k;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lds.acc.v[c][id] = a.sum[3 * p + c];
                lds.acc.v[3 + c][id] = a.sum2[3 * p + c];
            }
            lds.acc.v[6][id] = __int_as_float(a.count[p]);
        }
    }
    }
    // Adaptive sampling (main.cpp:118-125) asks, before every pass > 10, whether the variance estimate of all three
    // channels is below `error`.  That is a pure function of the accumulators, which change only when this pixel's
    // path reaches an emitter, so the answer is cached in one bit and refreshed there: no per-pass re-reads.
    auto low_variance = [&](float c0, float c1, float c2, float q0, float q1, float q2, int n) {
        const float sc = static_cast<float>(n);
        if (!(sc > 0)) return false;
        const float mr = c0 / sc, mg_ = c1 / sc, mb = c2 / sc;
        const float dr = q0 / sc - mr * mr, dg = q1 / sc - mg_ * mg_, db = q2 / sc - mb * mb;
        return dr < a.error && dg < a.error && db < a.error;
    };
    bool lowvar[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        lowvar[k] = false;
        if constexpr (ADAPT) {
            // (the answers of this instantiation live in LDS, by pixel: below)
        } else if constexpr (!kAccInLds) {
            if (in_image[k] && a.error >= 0.0f) {   // (adaptive sampling off: nobody asks, and a work item need not read its tile at all)
                const size_t p = static_cast<size_t>(acc_y0 + lane / kTileW) * a.width + x[k];
                lowvar[k] = low_variance(a.sum[3 * p], a.sum[3 * p + 1], a.sum[3 * p + 2], a.sum2[3 * p], a.sum2[3 * p + 1], a.sum2[3 * p + 2], a.count[p]);
            }
        } else {
            const int id = lane + 64 * k;
            if (in_image[k]) lowvar[k] = low_variance(lds.acc.v[0][id], lds.acc.v[1][id], lds.acc.v[2][id], lds.acc.v[3][id], lds.acc.v[4][id],
                                                      lds.acc.v[5][id], __float_as_int(lds.acc.v[6][id]));
        }
    }
    // ADAPT: the pixels of the tile (number j = owner lane | column block << 6) the lane's two ray slots work on in the current
    // batch: slot 0's in bits 0-7, slot 1's in bits 8-15 (see "Batches" in the pass loop).  Everything about a traced pixel --
    // camera ray, RNG counter, accumulator address, pass -- is derived from its number where it is needed.
    uint32_t jobs = 0;
    auto job_of = [&](int k) { return (opaque(jobs) >> (8 * k)) & 0xFFu; };
    auto tile_x_of = [&](uint32_t j) { return tile_x0 + static_cast<int>((j & 63u) % kTileW) + kTileW * static_cast<int>(j >> 6); };
    auto tile_y_of = [&](uint32_t j) { return tile_y0 + static_cast<int>((j & 63u) / kTileW); };
    if constexpr (ADAPT) {
#pragma unroll
        for (int kb = 0; kb < kOwn; ++kb) {   // the lane's own pixels: next pass | answer
            const uint32_t j = static_cast<uint32_t>(lane) + 64u * kb;
            bool low = false;
            if (tile_x_of(j) < a.width && y < a.row_end) {
                const size_t p = static_cast<size_t>(acc_y0 + lane / kTileW) * a.width + tile_x_of(j);
                low = low_variance(a.sum[3 * p], a.sum[3 * p + 1], a.sum[3 * p + 2], a.sum2[3 * p], a.sum2[3 * p + 1], a.sum2[3 * p + 2], a.count[p]);
            }
            lds.low.v[j] = static_cast<uint16_t>((static_cast<uint32_t>(pass_first) << 1) | (low ? 1u : 0u));
        }
        wave_sync();
    }
    // adds one contribution (material.h:74-77) to pixel k of the lane and refreshes its cached adaptive-sampling answer
    auto contribute = [&](int k, float cr, float cg, float cb) {
        float n0, n1, n2, p0, p1, p2;
        int nn;
        if constexpr (!kAccInLds) {
            // (the pixel's index is rebuilt from the lane number and the tile's wave-uniform corner, like the camera ray's x, y)
            // (opaque: or the addresses are formed at the head of the pass and held -- spilled -- until here)
            uint32_t le = opaque(static_cast<uint32_t>(lane));
            int kb = k;
            if constexpr (ADAPT) {
                const uint32_t j = job_of(k);
                le = j & 63u;
                kb = static_cast<int>(j >> 6);
            }
            const size_t p = static_cast<size_t>(acc_y0 + static_cast<int>(le / kTileW)) * a.width + (tile_x0 + static_cast<int>(le % kTileW) + kTileW * kb);
            n0 = a.sum[3 * p] + cr; n1 = a.sum[3 * p + 1] + cg; n2 = a.sum[3 * p + 2] + cb;
            p0 = a.sum2[3 * p] + cr * cr; p1 = a.sum2[3 * p + 1] + cg * cg; p2 = a.sum2[3 * p + 2] + cb * cb;
            nn = a.count[p] + 1;
            a.sum[3 * p] = n0; a.sum[3 * p + 1] = n1; a.sum[3 * p + 2] = n2;
            a.sum2[3 * p] = p0; a.sum2[3 * p + 1] = p1; a.sum2[3 * p + 2] = p2;
            a.count[p] = nn;
        } else {
            const int id = lane + 64 * k;   // the pixel's slot in the tile's accumulators
            n0 = lds.acc.v[0][id] + cr; n1 = lds.acc.v[1][id] + cg; n2 = lds.acc.v[2][id] + cb;
            p0 = lds.acc.v[3][id] + cr * cr; p1 = lds.acc.v[4][id] + cg * cg; p2 = lds.acc.v[5][id] + cb * cb;
            nn = __float_as_int(lds.acc.v[6][id]) + 1;
            lds.acc.v[0][id] = n0; lds.acc.v[1][id] = n1; lds.acc.v[2][id] = n2;
            lds.acc.v[3][id] = p0; lds.acc.v[4][id] = p1; lds.acc.v[5][id] = p2;
            lds.acc.v[6][id] = __int_as_float(nn);
        }
        const bool now_low = low_variance(n0, n1, n2, p0, p1, p2, nn);
        if constexpr (ADAPT) {
            // (the answers live in LDS, by pixel, whoever traces it -- bit 0 of the pixel's word, its next pass above --; the owners
            // read them at the head of every batch)
            const uint32_t pj = job_of(k);
            lds.low.v[pj] = static_cast<uint16_t>((lds.low.v[pj] & ~1u) | (now_low ? 1u : 0u));
        } else {
            lowvar[k] = now_low;
        }
    };
    // statistics are wave-level (uniform) counts: they live in SGPRs
    using Count = std::conditional_t<STATS, uint32_t, Ignored>;
    Count n_traced = 0, n_segments = 0, n_contrib = 0, n_miss = 0;
    std::conditional_t<STATS, WaveStats, NoStats> wst;
#ifdef PT_PHASE_TIMERS
    wst.last = __builtin_amdgcn_s_memtime();
#endif

    const int mrr = a.mrr;
    const float eps = a.eps;
#ifdef PT_VERIFY_SHIPPED
    uint32_t v_checked = 0, v_bad = 0;   // wave-uniform
    uint32_t v_compacted = 0;            // passes a wave ran compacted (ADAPT): reported where this build has no other use for a field
#endif
#ifdef PT_ADAPT_COUNT
    // diagnostic build (make variant DEFS=-DPT_ADAPT_COUNT): what the batches of the adaptive instantiations held, reported in the
    // statistics block's fields: samples_traced = rays, wave_node_rounds / wave_exact_iterations = one- / two-slot batches,
    // segments / wave_segments = segment-loop iterations inside one- / two-slot batches
    uint32_t c_rays = 0, c_one = 0, c_two = 0, c_seg1 = 0, c_seg2 = 0;
#endif
    auto any_of = [&](const bool (&b)[R]) {   // wave-uniform: any ray of the wave
        bool v = b[0];
#pragma unroll
        for (int k = 1; k < R; ++k) v = v || b[k];
        return __any(v);
    };

    // this lane's path state (REGEN: across passes -- a lane is in a pass of its own)
    Ray q[R];
    float tr[R], tg[R], tb[R];   // Ray::color_ (throughput), ray.h:17
    int depth[R];
    int cur_pass[R], next_pass[R];   // REGEN: the pass slot k's path belongs to / the next one its pixel has to run
#pragma unroll
    for (int k = 0; k < R; ++k) {
        tr[k] = tg[k] = tb[k] = 1.0f;
        depth[k] = mrr;
        cur_pass[k] = next_pass[k] = pass_first;
        q[k].ox = q[k].oy = q[k].oz = 0.0f; q[k].dx = q[k].dy = 0.0f; q[k].dz = 1.0f;
    }
    for (int pass = pass_first; ADAPT ? true : REGEN ? pass == pass_first : pass < pass_last; ++pass) {   // (REGEN: one trip, the loop inside runs all passes; ADAPT: one trip per batch)
        // Adaptive skip, main.cpp:118-125.
        bool skip[R], traced[R];
        bool two = true;   // wave-uniform: some lane's second ray slot is in use this pass
        if constexpr (!ADAPT) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                skip[k] = !in_image[k] || (pass > 10 && (pass % 4) && lowvar[k]);
                if constexpr (REGEN) skip[k] = true;   // (paths are started inside the segment loop)
                traced[k] = !skip[k];
            }
            if constexpr (!REGEN) {
                if (!any_of(traced)) continue;
            }
        }
        // Batches (ADAPT).  With adaptive sampling on, most pixels of a tile sit out three passes of four in the second half of a
        // frame (Tor.obj, -ERR 0.001, 256 spp: a third of the pixels are still traced at the end), scattered over the tile: a wave
        // that runs the tile pass by pass, two pixels per lane, pays nearly every pass in full for a third of the rays.  So this
        // instantiation does not run passes, it runs BATCHES.  Every pixel of the tile (64 ADAPT of them: twice as many as ray
        // slots with 32 x 8 tiles) has its own next pass (lds.low: pass << 1 | "variance is low"), stepped over the passes it
        // sits out; a batch takes pixels that still have a pass to run in this work item, each at ITS next pass: up to 128, two
        // per lane, if at least PT_ADAPT_TWO_AT wait; otherwise up to 64, one per lane, with the second ray slots switched off
        // (`two`: 0.65 of the cost of a two-slot batch, profiles/r04_ab_logs.txt adapt1).  When more wait than a batch takes,
        // those that are furthest behind go first (all at the smallest next pass, then the others in pixel order), so that
        // the tile's pixels finish together.  The chosen pixels' numbers are compacted into a list (ranks by ballot and prefix
        // over the column blocks); ray slot k of lane l traces entry l + 64 k.
        // A pixel's passes still run in order, one per batch at most: its contributions are added in pass order and its
        // adaptive-sampling answer is the one main.cpp:118-125 computes before that pass (it changes only when the pixel's own
        // path contributes).  A sample does not depend on the lane or the batch that traces it (the counter RNG is keyed by pixel
        // and pass, the search returns the minimum over the same candidates), so the frame does not change.
        // the RNG's pixel index of ray slot k
        auto rng_pixel = [&](int k) {
            if constexpr (ADAPT) {
                const uint32_t j = job_of(k);
                return static_cast<uint32_t>(static_cast<size_t>(tile_y_of(j)) * a.width + tile_x_of(j));
            }
            return opaque(gpix[k]);
        };
        if constexpr (ADAPT) {
            wave_sync();   // (the answers written while shading the last batch, by whichever lane traced the pixel)
            uint32_t word[kOwn], np[kOwn];
            bool pend[kOwn], sel[kOwn];
            unsigned long long pb[kOwn];
            const uint32_t le = opaque(static_cast<uint32_t>(lane));   // (or the words' addresses are kept -- spilled -- across the batch)
            uint32_t n_pend = 0, behind = ~0u;
#pragma unroll
            for (int kb = 0; kb < kOwn; ++kb) {
                const uint32_t j = le + 64u * kb;
                word[kb] = lds.low.v[j];
                np[kb] = word[kb] >> 1;
                if (np[kb] > 10u && (np[kb] & 3u) && (word[kb] & 1u)) np[kb] = (np[kb] + 3u) & ~3u;   // sits out until the next multiple of 4
                pend[kb] = tile_x_of(j) < a.width && tile_y_of(j) < a.row_end && static_cast<int>(np[kb]) < pass_last;
                sel[kb] = pend[kb];
                pb[kb] = __ballot(pend[kb]);
                n_pend += __builtin_popcountll(pb[kb]);
                behind = min(behind, pend[kb] ? np[kb] : ~0u);
            }
            if (n_pend == 0) break;
            // (the box-tree kernel has one ray slot per lane; without kDynSlots a batch costs the same however few it holds)
            const uint32_t quota = (R == 2 && (!kDynSlots || n_pend >= static_cast<uint32_t>(PT_ADAPT_TWO_AT))) ? 128u : 64u;
            if (n_pend > quota) {
                const uint32_t m = wave_min(behind);
                uint32_t at_a = 0, at_b = 0, rank_a[kOwn], rank_b[kOwn];
                bool is_a[kOwn];
#pragma unroll
                for (int kb = 0; kb < kOwn; ++kb) {
                    is_a[kb] = pend[kb] && np[kb] == m;
                    const unsigned long long ab = __ballot(is_a[kb]), bb = pb[kb] & ~ab;
                    rank_a[kb] = at_a + lanes_below(ab);
                    rank_b[kb] = at_b + lanes_below(bb);
                    at_a += __builtin_popcountll(ab);
                    at_b += __builtin_popcountll(bb);
                }
                const uint32_t quota_b = at_a >= quota ? 0u : quota - at_a;
#pragma unroll
                for (int kb = 0; kb < kOwn; ++kb) sel[kb] = is_a[kb] ? rank_a[kb] < quota : (pend[kb] && rank_b[kb] < quota_b);
            }
            uint32_t n_sel = 0;
#pragma unroll
            for (int kb = 0; kb < kOwn; ++kb) {
                const unsigned long long sb = __ballot(sel[kb]);
                if (sel[kb]) lds.pairs[n_sel + lanes_below(sb)] = le + 64u * kb;   // (the queues are empty between two searches)
                n_sel += __builtin_popcountll(sb);
            }
            wave_sync();
            two = n_sel > 64u;
#ifdef PT_ADAPT_COUNT
            c_rays += n_sel;
            if (two) ++c_two; else ++c_one;
#endif
            jobs = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                skip[k] = le + 64u * k >= n_sel;
                traced[k] = !skip[k];
                if (traced[k]) jobs |= lds.pairs[le + 64u * k] << (8 * k);
            }
#ifdef PT_VERIFY_SHIPPED
            if (__any((traced[0] && (jobs & 0xFFu) != le) || (traced[R - 1] && R == 2 && (jobs >> 8) != le + 64u))) ++v_compacted;
#endif
            wave_sync();   // (the list was in the pair queue: read before the search fills that)
#pragma unroll
            for (int kb = 0; kb < kOwn; ++kb)   // the owners: the chosen pixels' next pass
                lds.low.v[le + 64u * kb] = static_cast<uint16_t>(((np[kb] + (sel[kb] ? 1u : 0u)) << 1) | (word[kb] & 1u));
        }
        const bool k1_on = kDynSlots ? two : true;   // (PT_SLOT_ON)

        // Primary ray, main.cpp:126-129 + Ray ctor ray.h:21-25 (double arithmetic, then narrowed).
        auto primary_dir = [&](int k, int pass_k, float &out_dx, float &out_dy, float &out_dz) {
            {
                uint32_t w0, w1, w2, w3;
                philox4x32_10(rng_pixel(k), static_cast<uint32_t>(pass_k), 0xFFFFFFFFu, 0u, a.seed, kPhiloxKey1, w0, w1, w2, w3);
                const double jx = jitter_double(w0), jy = jitter_double(w1);
                // x, y are made opaque once per pass so that their int->double conversions (and the doubles of width and
                // height) are redone here instead of being hoisted out of the pass loop, where they would occupy eight
                // VGPRs for the whole kernel (the compiler spilled them to scratch: ~1 GB of memory traffic per frame).
                // (x and y themselves are rebuilt from the lane number and the tile's wave-uniform corner: kept in two VGPRs for the
                // whole kernel they were spilled to scratch, a launch-time cost that doubled the time of a 256 x 256 frame)
                // (the big-scene kernel has the registers to keep them: there the recomputation costs 2.6 %)
                int xi = x[k], yi = y;
                if constexpr (!BIG) {
                    xi = tile_x0 + static_cast<int>(opaque(static_cast<uint32_t>(lane)) % kTileW) + kTileW * k;
                    yi = tile_y0 + static_cast<int>(opaque(static_cast<uint32_t>(lane)) / kTileW);
                }
                if constexpr (ADAPT) {
                    xi = tile_x_of(job_of(k));
                    yi = tile_y_of(job_of(k));
                }
                int wi = a.width, hi = a.height;
                asm volatile("" : "+v"(xi), "+v"(yi), "+s"(wi), "+s"(hi));
                const float ddx = static_cast<float>((xi + jx) / wi - 0.5f);
                const float ddy = static_cast<float>(-(yi + jy) / hi + 0.5f);
                const float ddz = 1.0f;
                const float inv = rcp_rn_normal(sqrt_rn_normal((ddx * ddx + ddy * ddy) + (1.0f * 1.0f + 0.0f * 0.0f)));   // 1 <= argument < 2
                out_dx = ddx * inv; out_dy = ddy * inv; out_dz = ddz * inv;
            }
        };
        // starts slot k's path along a primary direction (eye fixed at (0, 0, -20), main.cpp:129; Ray::color_ = 1, ray.h:17)
        auto start_path = [&](int k, float ddx, float ddy, float ddz) {
            q[k].dx = ddx; q[k].dy = ddy; q[k].dz = ddz;
            q[k].ox = 0.0f; q[k].oy = 0.0f; q[k].oz = -20.0f;
            tr[k] = tg[k] = tb[k] = 1.0f;
            depth[k] = 0;
        };
        auto primary_ray = [&](int k, int pass_k) {
            float ddx, ddy, ddz;
            primary_dir(k, pass_k, ddx, ddy, ddz);
            start_path(k, ddx, ddy, ddz);
        };
        // the pass of ray slot k's path: the wave's (a scalar) -- or, with regeneration or batches, the slot's own
        auto pass_of = [&](int k) {
            if constexpr (ADAPT) {   // (a traced pixel's word holds the pass AFTER this one: read where needed, a register would spill)
                return static_cast<int>(lds.low.v[job_of(k)] >> 1) - 1;
            } else if constexpr (REGEN) {
                return cur_pass[k];
            } else {
                return pass;
            }
        };
        if constexpr (!REGEN) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                tr[k] = tg[k] = tb[k] = 1.0f;
                depth[k] = mrr;
                if (PT_SLOT_ON(k) && !skip[k]) {
                    primary_ray(k, pass_of(k));
                } else {
                    q[k].ox = q[k].oy = q[k].oz = 0.0f; q[k].dx = q[k].dy = 0.0f; q[k].dz = 1.0f;
                }
                if constexpr (STATS) n_traced += __builtin_popcountll(__ballot(!skip[k]));
            }
        }
        for (;;) {
            bool valid[R];
#pragma unroll
            for (int k = 0; k < R; ++k) valid[k] = depth[k] < mrr && (tr[k] != 0.0f || tg[k] != 0.0f || tb[k] != 0.0f);   // Ray::IsValid, ray.h:52-54
            if constexpr (REGEN) {
                // Regeneration: a slot whose path is over takes its pixel's next pass -- after the adaptive skip of
                // main.cpp:118-125, which sits out passes > 10 that are not multiples of 4 while the variance is low: the next
                // one that runs is then the next multiple of 4.
                // The primary-ray code (Philox, two double-precision divisions, a normalisation) costs about a third of a segment
                // however few lanes run it, so it runs only once a.regen_min_dead slots of the wave wait for a path, or when no
                // ray of the wave is alive (a.regen_min_dead = 64: the wave's passes stay in step, as without regeneration).  The
                // host sets it per launch (enqueue_render): 4 from -MRR 5 up, 64 below -- measured on Tor.obj without its back
                // wall, 1080p x 64 spp (profiles/r04_regen_sweep.jsonl; Msamples/s at 1 / 16 / 32 / 64): -MRR 8 6 980 / 6 650 / 6 210 /
                // 5 350, -MRR 5 8 250 / 8 030 / 7 710 / 8 100, -MRR 3 11 260 / 11 360 / 12 740 / 13 390.  Making the rays in advance and
                // in batches (a ray in store per slot) was built and measured too: +1 % at -MRR 8, -16 % at -MRR 3 still -- a slot
                // that ends two paths in a row has nothing in store, and some slot of 64 nearly always does (r04_ab_logs.txt, regen2).
                // Which iteration a path starts in changes nothing about it: the frames are the same for every setting.
                uint32_t n_wait = 0;
                bool wants[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    wants[k] = false;
                    if (!valid[k] && in_image[k]) {
                        int np = next_pass[k];
                        if (np > 10 && (np & 3) && lowvar[k]) np = (np + 3) & ~3;
                        next_pass[k] = np;                 // (the skip is final: lowvar only changes when this slot's own path contributes)
                        wants[k] = np < pass_last;
                    }
                    n_wait += __builtin_popcountll(__ballot(wants[k]));
                }
                if (n_wait >= a.regen_min_dead || (n_wait > 0 && !any_of(valid))) {
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        if (wants[k]) {
                            cur_pass[k] = next_pass[k];
                            primary_ray(k, next_pass[k]);
                            next_pass[k] = next_pass[k] + 1;
                        }
                        valid[k] = valid[k] || wants[k];
                        if constexpr (STATS) n_traced += __builtin_popcountll(__ballot(wants[k]));
                    }
                }
            }
            if (!any_of(valid)) break;
#ifdef PT_ADAPT_COUNT
            if (two) ++c_seg2; else ++c_seg1;
#endif
            if constexpr (STATS) {
#pragma unroll
                for (int k = 0; k < R; ++k) n_segments += __builtin_popcountll(__ballot(valid[k]));
            }

            float best[R];
            int hit[R];
            const ExactRec *hit_rec[R];
            // The culling margins hold for origins within r_org of the scene (pt_scene.cpp).  A path can leave that envelope:
            // the reference accepts a near-degenerate triangle for points that have nothing to do with it, at any distance
            // (all three computed sub-areas can vanish), and the next segment then starts millions of units away.  For such
            // a ray nothing is culled: every triangle goes through the exact test.  (A NaN origin counts as outside.)
            // (one v_max3_f32 with |.| modifiers and one compare; an origin here is never NaN: it is o + d t + N eps of finite terms)
            bool inside[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                inside[k] = true;
                if constexpr (ENV)
                    inside[k] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(q[k].ox), __builtin_fabsf(q[k].oy)), __builtin_fabsf(q[k].oz)) <= a.r_org;
            }
            // A path's last segment (depth + 1 == mrr; the live rays of a wave reach it together) can only contribute by hitting
            // an emitter, and whatever else it hits is never looked at (no next ray, statistics not requested, no skybox).  So
            // the search runs among the emitters alone first -- for Tor.obj one quad record instead of seven walls and a torus --
            // and the full search, which decides whether the emitter really is the closest hit, only for rays that hit one.
            bool searched[R], last_or_dead[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                searched[k] = valid[k];
                last_or_dead[k] = !valid[k] || depth[k] + 1 >= mrr;
            }
            bool all_last = last_or_dead[0];
#pragma unroll
            for (int k = 1; k < R; ++k) all_last = all_last && last_or_dead[k];
            constexpr bool kLastSegmentFilter = !STATS && !SKY && !BIG;   // (big scenes: the flag's scalar registers cost more than it saves)
            bool emis_phase = false;
            if constexpr (kLastSegmentFilter) emis_phase = a.last_segment_filter != 0u && __all(all_last);
#ifdef PT_VERIFY_SHIPPED
            bool filtered_last = emis_phase;   // wave-uniform: this segment's search only has to find emitters
#endif
            if constexpr (BIG && !STATS && !SKY) {
                // Big scenes: the same idea without a second search.  The table builder keeps a big scene's emitters in the
                // large class (pt_scene.cpp), so the conservative test of those records alone says which rays of the last
                // segment can reach an emitter at all; only those are searched.
                if (a.last_segment_filter != 0u && a.emis_bvh == 0u && a.n_clusters == 1 && __all(all_last)) {
                    const ConstF cp = (ConstF)reinterpret_cast<uintptr_t>(a.clusters) + (sizeof(ClusterDesc) / 4) * (a.n_clusters - 1);
                    const uint32_t n_large = ((ConstU)cp)[5], kind = ((ConstU)cp)[6], off = ((ConstU)cp)[7], quads = ((ConstU)cp)[9];
                    if (kind == 1u && n_large <= static_cast<uint32_t>(kChunk)) {
                        float k1 = a.k1, k2 = a.k2, a_max = a.a_max, m0 = a.m0, m0q = a.m0_quad, t_guard = a.t_guard;
                        asm volatile("" : "+v"(k1), "+v"(k2), "+v"(a_max), "+v"(m0), "+v"(m0q), "+v"(t_guard));
                        const ConstF bp = (ConstF)reinterpret_cast<uintptr_t>(a.bary) + 12 * static_cast<size_t>(off);
                        uint32_t m = 0;
                        for (uint32_t rest = (a.emis_large_w0 | (a.emis_large_w0 >> 1)) & 0x55555555u; rest != 0; rest &= rest - 1) {
                            const uint32_t k0 = __builtin_ctz(rest);
                            if ((quads >> k0) & 1u) {
                                m |= (~cull_reject_quad(load_cull(bp + 12 * k0), q[0], k1, k2, a_max, m0q, t_guard) & 3u) << k0;
                            } else {
                                for (uint32_t j = 0; j < 2; ++j)
                                    m |= cull_reject(load_cull(bp + 12 * (k0 + j)), q[0], k1, k2, a_max, m0, t_guard) ? 0u : (1u << (k0 + j));
                            }
                        }
                        bool can_reach = (m & a.emis_large_w0) != 0u;
                        if constexpr (ENV) can_reach = can_reach || !inside[0];   // (outside the margins' envelope nothing is culled)
                        searched[0] = valid[0] && can_reach;
#ifdef PT_VERIFY_SHIPPED
                        filtered_last = true;
#endif
                    }
                }
            }
            for (;;) {
                if (BIG && !any_of(searched)) break;
                closest_hit<ENV, kLastSegmentFilter, kDynSlots>(a, lds, q, searched, inside, lane, eps, best, hit, hit_rec, wst, emis_phase, two);
                if (!emis_phase) break;
                emis_phase = false;
#pragma unroll
                for (int k = 0; k < R; ++k) searched[k] = searched[k] && hit[k] >= 0;
                if (!any_of(searched)) break;
            }
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (!searched[k]) hit[k] = -1;   // (a ray of the last segment that met no emitter ends like a miss, contributing nothing)
#ifdef PT_VERIFY_SHIPPED
            // Verification of the path that SHIPS (libpt_verify_shipped.so, never the product): this is the statistics-free
            // instantiation with the emitter-first last segment and the big scenes' can-reach filter compiled in.  Every segment's
            // result is compared with Scene::TraceRay's loop as written (scene.cpp:116-120) for the lane's own ray: the same
            // (distance bits, triangle index) -- except that a FILTERED last segment may report a miss where the reference hits
            // something, if and only if that something has no emissive lobe (nothing else of a last segment is ever looked at:
            // Ray::IsValid ray.h:52-54, material.h:67-80).
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const unsigned long long brute = brute_force_key(a, q[k], eps);
                const unsigned long long mine = hit[k] < 0 ? ~0ull : ((static_cast<unsigned long long>(ordered_bits(best[k])) << 32) | static_cast<uint32_t>(hit[k]));
                bool ok = brute == mine;
                if (!ok && filtered_last && mine == ~0ull) {
                    bool emissive = false;
                    if (brute != ~0ull) {
                        const MatRec m = a.mats[a.exact[static_cast<uint32_t>(brute)].material];
                        emissive = (m.n_lobes >= 1 && m.kind0 == 0) || (m.n_lobes >= 2 && m.kind1 == 0);
                    }
                    ok = !emissive;
                }
                v_checked += static_cast<uint32_t>(__builtin_popcountll(__ballot(valid[k])));
                v_bad += static_cast<uint32_t>(__builtin_popcountll(__ballot(valid[k] && !ok)));
                if (valid[k] && !ok && a.stats) {   // one example for the host to print (any of them)
                    a.stats[11] = brute;
                    a.stats[12] = mine;
                    a.stats[13] = (static_cast<unsigned long long>(__float_as_uint(q[k].ox)) << 32) | __float_as_uint(q[k].oy);
                    a.stats[14] = (static_cast<unsigned long long>(__float_as_uint(q[k].oz)) << 32) | __float_as_uint(q[k].dx);
                    a.stats[15] = (static_cast<unsigned long long>(__float_as_uint(q[k].dy)) << 32) | __float_as_uint(q[k].dz);
                }
            }
#endif

            // ---- 3. shade (Scene::TraceRay scene.cpp:121-156, Material::Process material.h:36-50), ray after ray of the lane
#pragma unroll
            for (int k = 0; k < R; ++k) {
            if constexpr (STATS) n_miss += __builtin_popcountll(__ballot(valid[k] && hit[k] < 0));
            bool contributed = false;
            if (PT_SLOT_ON(k) && valid[k]) {
                if (hit[k] < 0) {
                    if (SKY) {   // skybox miss shader, scene.cpp:126-154 (note: the path throughput is NOT applied)
                        const float pi = 3.141593f;
                        const float theta = portable_acosf(q[k].dy) / pi;
                        const float phi = portable_atan2f(q[k].dz, -q[k].dx) / pi / 2 + 0.5f;
                        const uint32_t sw = static_cast<uint32_t>(a.sky_w), sh = static_cast<uint32_t>(a.sky_h);
                        const float sx = phi * static_cast<float>(sw), sy = theta * static_cast<float>(sh);
                        // float -> unsigned is undefined for NaN / out of range in the reference; clamp into the image
                        uint32_t x1 = (sx >= 0.0f) ? (sx < 4294967040.0f ? static_cast<uint32_t>(sx) : 0xFFFFFFFFu) : 0u;
                        uint32_t y1 = (sy >= 0.0f) ? (sy < 4294967040.0f ? static_cast<uint32_t>(sy) : 0xFFFFFFFFu) : 0u;
                        x1 = min(x1, sw - 1u);
                        y1 = min(y1, sh - 1u);
                        const uint32_t x2 = (x1 + 1u) % sw, y2 = (y1 + 1u) % sh;
                        const uint8_t *t1 = a.sky + (static_cast<size_t>(y1) * sw + x1) * 3, *t2 = a.sky + (static_cast<size_t>(y1) * sw + x2) * 3;
                        const uint8_t *t3 = a.sky + (static_cast<size_t>(y2) * sw + x1) * 3, *t4 = a.sky + (static_cast<size_t>(y2) * sw + x2) * 3;
                        const float ax = 1 - sx + static_cast<float>(x1), ay = 1 - sy + static_cast<float>(y1);
                        float c[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {   // r,g,b = bytes 2,1,0
                            const float c1 = static_cast<float>(t1[2 - ch]), c2 = static_cast<float>(t2[2 - ch]);
                            const float c3 = static_cast<float>(t3[2 - ch]), c4 = static_cast<float>(t4[2 - ch]);
                            const float c12 = c1 * (1.0f - ax) + c2 * ax;   // glm::mix(x, y, a) = x*(1-a) + y*a
                            const float c34 = c3 * (1.0f - ax) + c4 * ax;
                            c[ch] = (c12 * (1.0f - ay) + c34 * ay) / 256.f;
                        }
                        contribute(k, c[0], c[1], c[2]);
                        contributed = true;
                    }
                    depth[k] = mrr;   // MakeInvalid
                } else {
                    const ExactRec *__restrict__ rec = hit_rec[k];
                    const float4 pl = reinterpret_cast<const float4 *>(rec)[0];
                    const int mi = rec->material;
                    const float px = q[k].ox + q[k].dx * best[k], py = q[k].oy + q[k].dy * best[k], pz = q[k].oz + q[k].dz * best[k];
                    float4 m0v, m1v;   // kd, chance0; ks, chance1
                    int4 m2v;          // n_lobes, kind0, kind1
                    bool mat_cached = false;
                    if constexpr (decltype(lds)::kMatCache > 0) mat_cached = a.n_mats <= decltype(lds)::kMatCache;
                    if (mat_cached) {
                        if constexpr (decltype(lds)::kMatCache > 0) {
                            m0v = lds.mat.v[3 * mi]; m1v = lds.mat.v[3 * mi + 1];
                            const float4 t = lds.mat.v[3 * mi + 2];
                            m2v = make_int4(__float_as_int(t.x), __float_as_int(t.y), __float_as_int(t.z), __float_as_int(t.w));
                        }
                    } else {
                        m0v = reinterpret_cast<const float4 *>(a.mats + mi)[0];
                        m1v = reinterpret_cast<const float4 *>(a.mats + mi)[1];
                        m2v = reinterpret_cast<const int4 *>(a.mats + mi)[2];
                    }
                    // On a path's last segment the random words only matter where they choose between an emissive lobe and
                    // another one (see below: nothing else of that segment survives it).
                    uint32_t w0 = 0, w1 = 0, w2 = 0, w3;
                    if (depth[k] + 1 < mrr || (m2v.x >= 2 && (m2v.y == 0 || m2v.z == 0)))
                        philox4x32_10(rng_pixel(k), static_cast<uint32_t>(pass_of(k)), static_cast<uint32_t>(depth[k]), 0u, a.seed, kPhiloxKey1,
                                      w0, w1, w2, w3);
                    int kind;
                    if (m2v.x == 0) {
                        kind = -1;
                    } else if (m2v.x == 1) {
                        kind = m2v.y;
                    } else {
                        // `while (sample > 0) { ++i; sample -= chance_[i]; }` with sample in (0,1); past the last lobe
                        // the reference reads out of bounds, here the last lobe is kept.
                        const float sample = unit_float(w0);
                        kind = (sample - m0v.w > 0) ? m2v.z : m2v.y;
                    }
                    if (kind < 0) {
                        depth[k] = mrr;
                    } else if (kind == 0) {   // emissive, material.h:68-79
                        if (!((q[k].dx * pl.x + q[k].dy * pl.y) + q[k].dz * pl.z > 0)) {
                            const float cr = tr[k] * m0v.x, cg = tg[k] * m0v.y, cb = tb[k] * m0v.z;
                            contribute(k, cr, cg, cb);
                            contributed = true;
                        }
                        depth[k] = mrr;
                    } else {
                        // Both scattering lobes end in Ray::Reflect (ray.h:45-50): the lobe-specific part leaves the new direction
                        // (not yet normalised by Reflect) and the throughput factor, the common tail runs once per wave.
                        // A path's last segment (depth + 1 == mrr: all live lanes of a wave reach it together) can only contribute
                        // through the emissive lobe above: the ray a scattering lobe would produce is never traced
                        // (Ray::IsValid, ray.h:52-54), so it is not computed either.
                        if (depth[k] + 1 < mrr) {
                        float rx, ry, rz, fr, fg, fb;
                        if (kind == 1) {   // glossy, material.h:83-85
                            const float dn = (pl.x * q[k].dx + pl.y * q[k].dy) + pl.z * q[k].dz;
                            rx = q[k].dx - pl.x * dn * 2.0f; ry = q[k].dy - pl.y * dn * 2.0f; rz = q[k].dz - pl.z * dn * 2.0f;
                            fr = m1v.x; fg = m1v.y; fb = m1v.z;
                        } else {   // diffuse, material.h:90-100
                            const float xi1 = unit_float(w1), xi2 = unit_float(w2);
                            const float ang = 2 * 3.141593f * xi2;
                            float sn, cs;
                            portable_sincos(ang, sn, cs);
                            const float sq = sqrt_rn_normal(xi1);   // xi1 and 1 - xi1 are multiples of 2^-24 in [2^-24, 1)
                            rx = sq * cs; ry = sq * sn; rz = sqrt_rn_normal(1 - xi1);
                            normalize3(rx, ry, rz);
                            if ((pl.x * rx + pl.y * ry) + pl.z * rz < 0) { rx *= -1; ry *= -1; rz *= -1; }
                            float dt = (pl.x * rx + pl.y * ry) + pl.z * rz;
                            dt = dt > 0.0f ? dt : 0.0f;
                            fr = m0v.x * dt; fg = m0v.y * dt; fb = m0v.z * dt;
                        }
                        normalize3(rx, ry, rz);   // Ray::Reflect normalises (again), ray.h:47
                        q[k].ox = px + pl.x * eps; q[k].oy = py + pl.y * eps; q[k].oz = pz + pl.z * eps;
                        q[k].dx = rx; q[k].dy = ry; q[k].dz = rz;
                        tr[k] *= fr; tg[k] *= fg; tb[k] *= fb;
                        }
                        ++depth[k];
                    }
                }
            }
            if constexpr (STATS) n_contrib += __builtin_popcountll(__ballot(contributed));
            }
            PT_STAMP(wst, 7);   // shading
        }
    }

    // Write-back.  A tile whose rows are 16-byte aligned in the caller's planes is written with 16-byte stores (a row of
    // the tile is kTW*12 contiguous bytes of sum / sum2 and kTW*4 of count): dword stores at a 12-byte stride made the
    // memory side see about twice the bytes.  Column c of the tile's row r lives in LDS slot r*kTileW + c%kTileW + 64*(c/kTileW).
    if constexpr (kAccInLds) {
    const int tile_x = static_cast<int>(tile % a.blocks_x) * kTW, tile_y = a.row_begin + static_cast<int>(tile / a.blocks_x) * kTileH * a.row_stride;
    const int wb_row0 = a.row_begin + static_cast<int>(tile / a.blocks_x) * kTileH * (a.row_stride - 1);
    const bool whole = a.vec_ok && tile_x + kTW <= a.width && tile_y + kTileH <= a.row_end;   // wave-uniform
    wave_sync();
    auto slot_of = [](int row, int col) { return row * kTileW + (col % kTileW) + 64 * (col / kTileW); };
    if (whole) {
        constexpr int kRowVec = kTW * 3 / 4;          // float4 per tile row of a colour plane
#pragma unroll
        for (int v0 = 0; v0 < kRowVec * kTileH; v0 += kBlock) {
            const int vv = v0 + lane;
            if (vv < kRowVec * kTileH) {
                const int row = vv / kRowVec, v = vv % kRowVec;
                const size_t base = (static_cast<size_t>(tile_y - wb_row0 + row) * a.width + tile_x) * 3 + 4 * v;
                float4 o1, o2;
                float *p1 = &o1.x, *p2 = &o2.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * v + j;                   // element of the row: pixel e/3, channel e%3
                    p1[j] = lds.acc.v[e % 3][slot_of(row, e / 3)];
                    p2[j] = lds.acc.v[3 + e % 3][slot_of(row, e / 3)];
                }
                *reinterpret_cast<float4 *>(a.sum + base) = o1;
                *reinterpret_cast<float4 *>(a.sum2 + base) = o2;
            }
        }
        constexpr int kCntVec = kTW / 4;              // int4 per tile row of the count plane
        if (lane < kCntVec * kTileH) {
            const int row = lane / kCntVec, v = lane % kCntVec;
            const size_t base = static_cast<size_t>(tile_y - wb_row0 + row) * a.width + tile_x + 4 * v;
            int4 oc;
            oc.x = __float_as_int(lds.acc.v[6][slot_of(row, 4 * v)]);
            oc.y = __float_as_int(lds.acc.v[6][slot_of(row, 4 * v + 1)]);
            oc.z = __float_as_int(lds.acc.v[6][slot_of(row, 4 * v + 2)]);
            oc.w = __float_as_int(lds.acc.v[6][slot_of(row, 4 * v + 3)]);
            *reinterpret_cast<int4 *>(a.count + base) = oc;
        }
    } else {
        // recomputed from the tile's corner and the lane number: p, x or y kept across the kernel would be spilled
        const uint32_t le = opaque(static_cast<uint32_t>(lane));
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int xe = tile_x + static_cast<int>(le % kTileW) + kTileW * k, ye = tile_y + static_cast<int>(le / kTileW);
            if (xe < a.width && ye < a.row_end) {
                const size_t pe = static_cast<size_t>(ye - wb_row0) * a.width + xe;
                const int id = lane + 64 * k;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a.sum[3 * pe + c] = lds.acc.v[c][id];
                    a.sum2[3 * pe + c] = lds.acc.v[3 + c][id];
                }
                a.count[pe] = __float_as_int(lds.acc.v[6][id]);
            }
        }
    }
    }
    if (chunk + 1 < a.n_chunks) {   // publish the tile's accumulators to the wave that takes its next chunk
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the fence's own wait can be dropped by the compiler (guide, G16)
        if (lane == 0) __hip_atomic_store(&a.sched[1 + tile], chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef PT_ADAPT_COUNT
    if (a.stats && lane == 0) {
        atomicAdd(&a.stats[0], static_cast<unsigned long long>(c_rays));
        atomicAdd(&a.stats[6], static_cast<unsigned long long>(c_one));
        atomicAdd(&a.stats[7], static_cast<unsigned long long>(c_two));
        atomicAdd(&a.stats[1], static_cast<unsigned long long>(c_seg1));
        atomicAdd(&a.stats[5], static_cast<unsigned long long>(c_seg2));
    }
#endif
#ifdef PT_VERIFY_SHIPPED
    if (a.stats && lane == 0) {
        atomicAdd(&a.stats[9], static_cast<unsigned long long>(v_checked));
        if (v_bad) atomicAdd(&a.stats[10], static_cast<unsigned long long>(v_bad));
        if (v_compacted) atomicAdd(&a.stats[8], static_cast<unsigned long long>(v_compacted));   // (pt_render_stats::partial_commit_rounds)
    }
#endif
    if constexpr (STATS) if (a.stats && lane == 0) {
        atomicAdd(&a.stats[0], static_cast<unsigned long long>(n_traced));
        atomicAdd(&a.stats[1], static_cast<unsigned long long>(n_segments));
        atomicAdd(&a.stats[2], static_cast<unsigned long long>(n_contrib));
        atomicAdd(&a.stats[3], static_cast<unsigned long long>(wst.n_exact));
        atomicAdd(&a.stats[4], static_cast<unsigned long long>(n_miss));
        atomicAdd(&a.stats[5], static_cast<unsigned long long>(wst.w_segments));
        atomicAdd(&a.stats[6], static_cast<unsigned long long>(wst.w_node_rounds));
        atomicAdd(&a.stats[7], static_cast<unsigned long long>(wst.w_exact_iters));
        if (wst.w_partial) atomicAdd(&a.stats[8], static_cast<unsigned long long>(wst.w_partial));
#ifdef PT_VERIFY_BRUTE
        atomicAdd(&a.stats[9], static_cast<unsigned long long>(wst.v_checked));
        if (wst.v_bad) atomicAdd(&a.stats[10], static_cast<unsigned long long>(wst.v_bad));
#endif
#ifdef PT_PHASE_TIMERS
        for (int k = 0; k < 8; ++k) atomicAdd(&a.stats[16 + k], wst.phase[k]);
#endif
    }
}

// Closest hit for caller-supplied rays (the intersection half of Scene::TraceRay, scene.cpp:114-120).
template <bool BIG>
__global__ __launch_bounds__(kBlock, BIG ? PT_BIG_WAVES : PT_WAVES_PER_SIMD) void trace_rays_kernel(const RenderArgs a, const float *__restrict__ origins,
                                                                              const float *__restrict__ directions, int n_rays,
                                                                              int32_t *__restrict__ hit_index, float *__restrict__ hit_t) {
    __shared__ WaveLds<std::conditional_t<BIG, BigQueues, SmallQueues>, 1> lds;
    const int lane = threadIdx.x;
    const int i = blockIdx.x * kBlock + lane;
    const bool valid = i < n_rays;
    Ray q;
    q.ox = q.oy = q.oz = 0.0f; q.dx = q.dy = 0.0f; q.dz = 1.0f;
    if (valid) {
        q.ox = origins[3 * i]; q.oy = origins[3 * i + 1]; q.oz = origins[3 * i + 2];
        q.dx = directions[3 * i]; q.dy = directions[3 * i + 1]; q.dz = directions[3 * i + 2];
    }
    // Rays outside the envelope the culling margins were derived for (pt_hip.h: pt_trace_rays_host) get every triangle as
    // a candidate for the exact test instead; written so that NaNs count as outside.
    const float d2 = (q.dx * q.dx + q.dy * q.dy) + q.dz * q.dz;
    const bool inside = __builtin_fabsf(q.ox) <= a.r_org && __builtin_fabsf(q.oy) <= a.r_org && __builtin_fabsf(q.oz) <= a.r_org &&
                        __builtin_fabsf(d2 - 1.0f) <= 1.0e-5f;
    float best[1];
    int hit[1];
    const ExactRec *hit_rec[1];
    WaveStats st;
    const Ray qs[1] = {q};
    const bool lives[1] = {valid}, insides[1] = {inside};
    closest_hit<true, false>(a, lds, qs, lives, insides, lane, a.eps, best, hit, hit_rec, st);
    if (valid) {
        hit_index[i] = hit[0];
        hit_t[i] = best[0];
    }
}

// Diagnostic (test builds call it through pt_test_box_masks): both forms of the box tree's child test on caller-supplied
// (node, ray, t_best) items, one item per lane -- out[2 i] = box_children_kept<8>, out[2 i + 1] = box_children_kept_h.
__global__ __launch_bounds__(kBlock) void box_masks_kernel(const BvhNode *__restrict__ nodes, const float *__restrict__ rays,
                                                           const float *__restrict__ t_best, float err, int n, uint32_t *__restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint4 *np = reinterpret_cast<const uint4 *>(nodes + i);
    const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    Ray r;
    r.ox = rays[6 * i]; r.oy = rays[6 * i + 1]; r.oz = rays[6 * i + 2];
    r.dx = rays[6 * i + 3]; r.dy = rays[6 * i + 4]; r.dz = rays[6 * i + 5];
    const float ix = __builtin_amdgcn_rcpf(r.dx), iy = __builtin_amdgcn_rcpf(r.dy), iz = __builtin_amdgcn_rcpf(r.dz);
    const uint32_t exists = (2u << ((q0.w >> 8) & 7u)) - 1u;
    out[2 * i] = box_children_kept<8>(q0, q1, q2, q3, r, ix, iy, iz, t_best[i], err) & exists;
    // (bits 8-15 of the second word: the mixed-precision form, which must equal the float form bit for bit)
    out[2 * i + 1] = (box_children_kept_h(q0, q1, q2, q3, r, ix, iy, iz, t_best[i], err) & exists) |
                     ((box_children_kept_mix(q0, q1, q2, q3, r, ix, iy, iz, t_best[i], err) & exists) << 8);
}
hipError_t launch_box_masks(const BvhNode *d_nodes, const float *d_rays, const float *d_t_best, float err, int n, uint32_t *d_out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(box_masks_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_nodes, d_rays, d_t_best, err, n, d_out);
    return hipGetLastError();
}

hipError_t launch_trace_rays(const RenderArgs &args, const float *d_origins, const float *d_directions, int n_rays,
                             int32_t *d_hit_index, float *d_hit_t, hipStream_t stream) {
    if (n_rays <= 0) return hipSuccess;
    const unsigned grid = static_cast<unsigned>((n_rays + kBlock - 1) / kBlock);
    if ((args.big != 0))
        hipLaunchKernelGGL(trace_rays_kernel<true>, dim3(grid), dim3(kBlock), 0, stream, args, d_origins, d_directions, n_rays, d_hit_index, d_hit_t);
    else
        hipLaunchKernelGGL(trace_rays_kernel<false>, dim3(grid), dim3(kBlock), 0, stream, args, d_origins, d_directions, n_rays, d_hit_index, d_hit_t);
    return hipGetLastError();
}

#ifdef PT_BLOCK_PROFILE
// the instantiations the instrumented code object must contain (nothing in this build references them)
#define PT_INST(S, B) \
    template __global__ void integrate_kernel<S, B, false, false, false>(const RenderArgs); \
    template __global__ void integrate_kernel<S, B, true, false, false>(const RenderArgs);  \
    template __global__ void integrate_kernel<S, B, false, true, false>(const RenderArgs);  \
    template __global__ void integrate_kernel<S, B, true, true, false>(const RenderArgs);
PT_INST(false, false) PT_INST(false, true) PT_INST(true, false) PT_INST(true, true)
#undef PT_INST
}  // namespace pt
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
namespace pt {
// Diagnostic build (libpt_blockprof.so): the integrator is launched from an INSTRUMENTED copy of this file's code object
// (tools/asm_profile.py inserts an execution counter in front of every straight-line run of instructions of the compiler's
// own assembly) and the counters are written to $PT_BLOCKPROF_OUT.<kernel> after every launch.  Never part of the product.
hipError_t launch_integrator(const RenderArgs &args0, hipStream_t stream) {
    constexpr size_t kCounters = 4096;
    static hipModule_t mod = nullptr;
    static uint32_t *d_cnt = nullptr;
    const int rows = args0.row_end - args0.row_begin;
    if (rows <= 0 || args0.width <= 0) return hipSuccess;
    if (!mod) {
        const char *path = std::getenv("PT_BLOCKPROF_HSACO");
        if (!path) return hipErrorInvalidValue;
        hipError_t e = hipModuleLoad(&mod, path);
        if (e != hipSuccess) return e;
        e = hipMalloc(reinterpret_cast<void **>(&d_cnt), kCounters * sizeof(uint32_t));
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipMemsetAsync(d_cnt, 0, kCounters * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    RenderArgs args = args0;
    args.blockprof = d_cnt;
    char name[128];
    std::snprintf(name, sizeof name, "_ZN2pt16integrate_kernelILb%dELb%dELb%dELb%dELb0ELi0EEEvNS_10RenderArgsE", args.sky ? 1 : 0,
                  (args.big != 0) ? 1 : 0, args.stats ? 1 : 0, args.may_leave_envelope ? 1 : 0);
    hipFunction_t f;
    e = hipModuleGetFunction(&f, mod, name);
    if (e != hipSuccess) return e;
    size_t size = sizeof args;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    e = hipModuleLaunchKernel(f, args.n_tiles * args.n_chunks, 1, 1, kBlock, 1, 1, 0, stream, nullptr, config);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    std::vector<uint32_t> h(kCounters);
    e = hipMemcpy(h.data(), d_cnt, kCounters * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    if (const char *out = std::getenv("PT_BLOCKPROF_OUT")) {
        if (FILE *fp = std::fopen((std::string(out) + "." + name + ".txt").c_str(), "w")) {
            for (size_t i = 0; i < kCounters; ++i) if (h[i]) std::fprintf(fp, "%zu %u\n", i, h[i]);
            std::fclose(fp);
        }
    }
    return hipSuccess;
}
hipError_t integrator_waves_per_cu(const RenderArgs &, int *waves) {
    *waves = 24;
    return hipSuccess;
}
void integrator_plan_tiles(RenderArgs &args, int, int) {   // (the instrumented code object holds the wide variant only)
    const int rays = (!args.sky && !args.stats) ? ((args.big != 0) ? PT_BIG_RAYS_PER_LANE : PT_RAYS_PER_LANE) : 1;
    args.narrow = 0;
    args.adapt_pool = 0;
    args.blocks_x = (args.width + kTileW * rays - 1) / (kTileW * rays);
    args.n_tiles = static_cast<uint32_t>(args.blocks_x) * static_cast<uint32_t>((args.band_rows + kTileH - 1) / kTileH);
}
#else
namespace {
// statistics instantiation or not, for a launch with these arguments
bool launch_with_stats(const RenderArgs &args) {
#if defined(PT_PHASE_TIMERS) || defined(PT_VERIFY_BRUTE)
    (void)args;
    return true;
#elif defined(PT_VERIFY_SHIPPED) || defined(PT_ADAPT_COUNT)
    (void)args;
    return false;   // the instantiations a caller without pt_render_stats gets; args.stats only receives the verdict
#else
    return args.stats != nullptr;
#endif
}
// Calls f(kernel) with the instantiation a launch with these arguments runs.
template <class F>
void with_instantiation(const RenderArgs &args, F &&f) {
    const bool big = (args.big != 0);
    const bool stats = launch_with_stats(args);
    auto pick = [&](auto sky, auto bg, auto st) {
        constexpr bool S = decltype(sky)::value, B = decltype(bg)::value, T = decltype(st)::value;
        if constexpr (!S && !T && rays_per_lane<S, B, T>() > 1) {
            if (args.narrow) {
                if (args.may_leave_envelope) f(integrate_kernel<false, B, false, true, true>, 17 + 2 * B);
                else f(integrate_kernel<false, B, false, false, true>, 16 + 2 * B);
                return;
            }
            if constexpr (!B) {
                // adaptive sampling on: the instantiations that run batches (not built with the rare envelope test: one more
                // spilled register there)
                if (args.adapt_pool == 4) {
                    f(integrate_kernel<false, false, false, false, false, 4>, 21);
                    return;
                }
                if (args.adapt_pool == 2) {
                    f(integrate_kernel<false, false, false, false, false, 2>, 20);
                    return;
                }
            }
        }
        if constexpr (!S && !T && B) {   // the box-tree kernel with adaptive sampling on: batches over 16 x 8 / 32 x 8 tiles, one ray slot per lane
            if (args.adapt_pool == 4) {
                f(integrate_kernel<false, true, false, false, false, 4>, 23);
                return;
            }
            if (args.adapt_pool == 2) {
                f(integrate_kernel<false, true, false, false, false, 2>, 22);
                return;
            }
        }
        if (args.may_leave_envelope) f(integrate_kernel<S, B, T, true>, ((S * 2 + B) * 2 + T) * 2 + 1);
        else f(integrate_kernel<S, B, T, false>, ((S * 2 + B) * 2 + T) * 2);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    if (stats) {
        if (args.sky && big) pick(Yes(), Yes(), Yes());
        else if (args.sky) pick(Yes(), No(), Yes());
        else if (big) pick(No(), Yes(), Yes());
        else pick(No(), No(), Yes());
    } else {
        if (args.sky && big) pick(Yes(), Yes(), No());
        else if (args.sky) pick(Yes(), No(), No());
        else if (big) pick(No(), Yes(), No());
        else pick(No(), No(), No());
    }
}
}  // namespace

hipError_t launch_integrator(const RenderArgs &args, hipStream_t stream) {
    const int rows = args.row_end - args.row_begin;
    if (rows <= 0 || args.width <= 0) return hipSuccess;
    const unsigned grid = args.n_tiles * args.n_chunks;
    with_instantiation(args, [&](auto kernel, int) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, args); });
    return hipGetLastError();
}

// Cuts the launch's row band into the tiles of the instantiation it will run: fills narrow, blocks_x, n_tiles.  The
// statistics-free small-scene kernel owns 16 x 8 tiles (two pixels per lane) -- unless that would leave the chip's wave slots
// underfilled, in which case its 8 x 8 variant runs (a tile's passes are a serial chain: fewer tiles than slots means idle SIMDs).
void integrator_plan_tiles(RenderArgs &args, int cu_count, int force) {
    const bool sky = args.sky != nullptr, big = (args.big != 0), stats = launch_with_stats(args);
    const uint32_t rows = static_cast<uint32_t>((args.band_rows + kTileH - 1) / kTileH);      // tile rows of the band (its planes hold band_rows rows)
    int rays = (!sky && !stats) ? (big ? PT_BIG_RAYS_PER_LANE : PT_RAYS_PER_LANE) : 1;
    args.narrow = 0;
    args.adapt_pool = 0;
    int tile_px = rays;   // tile width in 8-pixel column blocks
    if (rays > 1) {
        const uint32_t wide_tiles = static_cast<uint32_t>((args.width + kTileW * rays - 1) / (kTileW * rays)) * rows;
        // adaptive sampling on: the instantiations that run batches (not built with the rare envelope test: one more spilled
        // register there)
        const bool batches = !big && args.error >= 0.0f && !args.may_leave_envelope && args.pass_begin >= 0 && args.pass_begin + args.pass_count <= kMaxBatchPass;
        // one and a half rounds of its waves (measured, profiles/r03_ab_logs.txt ab53: 7 200 tiles -19 %, 8 160 tiles +3 %, 16 200 +6.6 %);
        // the batch kernel beats the 8 x 8 kernel's sitting out from 1.2 rounds on (1280 x 720: 29.0 against 31.7 ms, r04_ab_logs.txt adapt6)
        const uint32_t slots = static_cast<uint32_t>(cu_count) * 4u * static_cast<uint32_t>(PT_WAVES_PER_SIMD - 1);
        uint32_t min_tiles = slots * 3u / 2u, min_wide = batches ? slots * 6u / 5u : min_tiles;
        if (force == 1) min_tiles = min_wide = 0xFFFFFFFFu;   // (test builds: always 8 x 8 / always 16 x 8 / always 32 x 8 with adaptive sampling on)
        if (force == 2 || force == 3) min_tiles = min_wide = 0;
        if (wide_tiles < min_wide) {
            rays = tile_px = 1;
            args.narrow = 1;
        } else if (batches) {   // over 32 x 8 tiles if there are enough of those as well, else over 16 x 8 tiles
            const uint32_t pool4_tiles = static_cast<uint32_t>((args.width + kTileW * 4 - 1) / (kTileW * 4)) * rows;
            args.adapt_pool = (force != 2 && pool4_tiles >= min_tiles) ? 4 : 2;
            tile_px = args.adapt_pool;
        }
    }
    if (rays == 1 && big && !sky && !stats && args.error >= 0.0f && !args.may_leave_envelope && args.pass_begin >= 0 &&
        args.pass_begin + args.pass_count <= kMaxBatchPass && force != 1) {
        // the box-tree kernel (one ray slot per lane, six waves per SIMD) with adaptive sampling on: batches of 64 over 16 x 8 tiles,
        // over 32 x 8 tiles where the frame has one and a half rounds of those
        const uint32_t min_tiles = (force == 2 || force == 3) ? 0u : static_cast<uint32_t>(cu_count) * 4u * static_cast<uint32_t>(PT_WAVES_PER_SIMD) * 3u / 2u;
        const uint32_t pool2_tiles = static_cast<uint32_t>((args.width + kTileW * 2 - 1) / (kTileW * 2)) * rows;
        const uint32_t pool4_tiles = static_cast<uint32_t>((args.width + kTileW * 4 - 1) / (kTileW * 4)) * rows;
        if (force != 2 && pool4_tiles >= min_tiles) args.adapt_pool = 4;
        else if (pool2_tiles >= min_tiles) args.adapt_pool = 2;
        if (args.adapt_pool) tile_px = args.adapt_pool;
    }
    args.blocks_x = (args.width + kTileW * tile_px - 1) / (kTileW * tile_px);
    args.n_tiles = static_cast<uint32_t>(args.blocks_x) * rows;
}

// Waves (= workgroups: one wave each) of that instantiation one compute unit holds at a time, from the runtime's occupancy
// calculation (registers, LDS, launch bounds): the scheduler's count of wave slots.  Asked once per instantiation and device.
hipError_t integrator_waves_per_cu(const RenderArgs &args, int *waves) {
    constexpr int kDevices = 16;
    static std::atomic<int> cache[kDevices][24];   // 0 = not asked yet
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipError_t result = hipSuccess;
    with_instantiation(args, [&](auto kernel, int id) {
        int n = dev < kDevices ? cache[dev][id].load(std::memory_order_relaxed) : 0;
        if (n == 0) {
            result = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlock, 0);
            if (result == hipSuccess && n > 0) {
                // The runtime's calculator divides the CU's LDS by the kernel's bytes; the hardware hands LDS out in granules of 1 280
                // bytes (tools/lds_granule_probe.hip, profiles/r04_lds_granule.txt: 7 888 B -> 18 workgroups run, the runtime says 20).
                hipFuncAttributes fa;
                hipDeviceProp_t prop;
                if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kernel)) == hipSuccess && fa.sharedSizeBytes > 0 &&
                    hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.maxSharedMemoryPerMultiProcessor > 0) {
                    constexpr size_t kLdsGranule = 1280;
                    const size_t per_wave = (fa.sharedSizeBytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule;
                    const size_t by_lds = std::max<size_t>(prop.maxSharedMemoryPerMultiProcessor, prop.sharedMemPerBlock) / per_wave;
                    if (by_lds >= 1 && by_lds < static_cast<size_t>(n)) n = static_cast<int>(by_lds);
                }
            }
            if (result != hipSuccess || n <= 0) n = 0;
            else if (dev < kDevices) cache[dev][id].store(n, std::memory_order_relaxed);
        }
        if (n > 0) *waves = n;
    });
    return result;
}
#endif

}  // namespace pt
