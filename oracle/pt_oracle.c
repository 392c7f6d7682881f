/*
 * pt_oracle.c -- CPU ORACLE for the radiance-integrator hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * smoke() check in __graft_entry__.py and the `cpu_baseline` leg of bench.py
 * may load it.  The shipped path (path-tracing_amd/csrc) never links, calls
 * or falls back to anything in this directory.
 *
 * What it is: a plain-C, scalar, one-ray-at-a-time restatement of the
 * reference's algorithm (Andareon/Path-Tracing), each function citing the
 * reference file:line it follows.  It deliberately shares NO code and no
 * headers with the HIP product so that it is an independent checker.
 *
 * Pinning status.  The reference ships no tests, golden images or fixtures,
 * and it cannot be compiled in this image (its only third-party dependency,
 * GLM -- un-vendored, version unpinned, CMakeLists.txt:13 -- is absent), so
 * by the strict definition: PARITY UNPINNED at the GLM boundary.  What this
 * oracle IS anchored on (tests/test_oracle_known_answers.py):
 *   - libstdc++ minstd_rand0 / uniform_real_distribution draw values and the
 *     whole-frame BMP md5s + dispersion statistics recorded in SURVEY.md
 *     section 8(c) for five configurations of the reference (sequential RNG,
 *     one thread, seed 42).  Reproducing a 196 662-byte BMP bit for bit
 *     exercises loader, plane set-up, intersection, all three lobes, the
 *     accumulator estimator, adaptive sampling, tonemap and BMP writer.
 *   - Random123's published known-answer vectors for Philox4x32-10.
 * GLM semantics restated here (GLM 0.9.9 scalar path): dot3=(x+y)+z of the
 * products, dot4=(x+y)+(z+w), cross as in geometric.inl, length=sqrt(dot),
 * normalize=v*(1/sqrt(dot(v,v))), reflect=I-N*dot(N,I)*2.
 *
 * Two RNG policies:
 *   ORC_RNG_SEQUENTIAL  the reference's own streams (two minstd_rand0 engines
 *                       consumed in single-thread order, material.h:16-20,
 *                       main.cpp:91-92) -- used only to pin against the md5s.
 *   ORC_RNG_COUNTER     stateless Philox4x32-10 keyed by (seed | pixel, pass,
 *                       segment) -- the policy the HIP path implements; a
 *                       parallel machine can reproduce it for any tiling.
 * Two trig policies for the diffuse lobe (material.h:92-94):
 *   ORC_TRIG_LIBM       cosf/sinf from libm, as the reference calls them.
 *   ORC_TRIG_PORTABLE   a fixed double-precision polynomial (+,-,* only) whose
 *                       result is reproducible bit for bit on any IEEE machine.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * Float model: IEEE binary32, no FMA contraction (CMakeLists.txt:5 sets only
 * -fopenmp => baseline x86-64 SSE2 arithmetic).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_RNG_SEQUENTIAL 0
#define ORC_RNG_COUNTER 1
#define ORC_TRIG_LIBM 0
#define ORC_TRIG_PORTABLE 1

typedef struct {
    float plane[4];   /* triangles.h:21  plane_ (n.xyz, w)        */
    float v[3][3];    /* triangles.h:23  vertices_                 */
    float square;     /* triangles.h:24  parallelogram area        */
    int material;     /* index into materials (scene.cpp:105-106)  */
} orc_triangle;

typedef struct {
    float Kd[3], Ke[3], Ks[3], Ns;   /* material.h:22-28 */
    /* lobe table built by Factory, material.h:58-106 */
    int n_lobes;
    int lobe_kind[2];                /* 0 emissive, 1 glossy, 2 diffuse */
    float chance[2];
} orc_material;

typedef struct {
    orc_triangle *tri;
    int n_tri;
    orc_material *mat;
    int n_mat;
    unsigned char *sky;          /* skybox_ (scene.h:15): BGR bytes, top-down rows, no padding; NULL = no skybox */
    int sky_w, sky_h;
} orc_scene;

typedef struct {
    int width, height;
    int row_begin, row_end;      /* rows traced by this call (counter RNG only; sequential needs the full frame) */
    int pass_begin, pass_count;  /* passes = samples per pixel (main.cpp:110) */
    int max_ray_reflections;     /* -MRR, config.h:19 */
    float eps;                   /* -EPS, config.h:22 */
    float error;                 /* -ERR, config.h:23 */
    uint32_t seed;               /* -SEED, config.h:10 */
    int rng_policy;
    int trig_policy;
    int threads;                 /* counter policy only; <=0 -> all cores */
} orc_params;

typedef struct {
    uint64_t samples_traced;     /* primary rays actually generated (adaptive skip excluded) */
    uint64_t segments;           /* TraceRay calls */
    uint64_t contributing;       /* samples that reached an emitter */
    uint64_t stage_exit[5];      /* Intersect exits: A, B, C, D-reject, accept */
    uint64_t misses;             /* segments with no triangle found */
} orc_stats;

/* ------------------------------------------------------------------ */
/* GLM-equivalent scalar helpers                                        */
/* ------------------------------------------------------------------ */
static inline float dot3f(const float *a, const float *b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline void cross3f(const float *x, const float *y, float *o) {
    o[0] = x[1] * y[2] - y[1] * x[2];
    o[1] = x[2] * y[0] - y[2] * x[0];
    o[2] = x[0] * y[1] - y[0] * x[1];
}
static inline float length3f(const float *a) { return sqrtf(dot3f(a, a)); }
/* normalize(vec4 (x,y,z,0)): dot4 = (xx+yy)+(zz+0) == (xx+yy)+zz */
static inline void normalize_dir(float *d) {
    float inv = 1.0f / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    d[0] = d[0] * inv; d[1] = d[1] * inv; d[2] = d[2] * inv;
}

/* ------------------------------------------------------------------ */
/* RNG: libstdc++ restatement (sequential policy)                       */
/* ------------------------------------------------------------------ */
/* std::default_random_engine == minstd_rand0: x <- 16807 x mod (2^31-1) */
typedef struct { uint32_t x; } minstd0;
static void minstd0_seed(minstd0 *g, uint32_t seed) {
    uint32_t s = seed % 2147483647u;
    g->x = s ? s : 1u;
}
static inline uint32_t minstd0_next(minstd0 *g) {
    g->x = (uint32_t)(((uint64_t)g->x * 16807u) % 2147483647u);
    return g->x;
}
/* uniform_real_distribution<float>(0,1) over minstd_rand0: one draw,
 * generate_canonical<float,24>: float(x-1) / 2147483648.0f, clamped below 1
 * (material.h:16-20). */
static inline float libstd_canonical_float(minstd0 *g) {
    float sum = (float)(uint64_t)(minstd0_next(g) - 1u);
    float ret = sum / 2147483648.0f;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}
/* uniform_real_distribution<double>(-0.5f,0.5f): two draws (main.cpp:92). */
static inline double libstd_jitter_double(minstd0 *g) {
    const double r = 2147483646.0;
    double sum = (double)(uint64_t)(minstd0_next(g) - 1u);
    double tmp = r;
    sum += (double)(uint64_t)(minstd0_next(g) - 1u) * tmp;
    tmp *= r;
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret * (0.5 - (-0.5)) + (-0.5);
}

/* ------------------------------------------------------------------ */
/* RNG: Philox4x32-10 (counter policy)                                  */
/* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC11). */
/* ------------------------------------------------------------------ */
static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* Stream layout shared with the HIP path (DESIGN.md "Counter RNG"):
 *   key  = (seed, 0x50544831 "PTH1")
 *   ctr  = (global pixel index y*W+x, pass, segment, 0)       for a bounce
 *          (global pixel index,       pass, 0xFFFFFFFF, 0)    for the camera jitter
 *   bounce: word0 -> lobe choice, word1 -> xi1, word2 -> xi2
 *   camera: word0 -> jx, word1 -> jy
 * u32 -> float in (0,1): ((w>>9)*2+1) * 2^-24   (never 0, never 1)
 * u32 -> jitter double in (-0.5,0.5): (w + 0.5) * 2^-32 - 0.5 */
#define ORC_PHILOX_KEY1 0x50544831u
static inline float u32_to_unit_float(uint32_t w) {
    return (float)(((w >> 9) << 1) | 1u) * 5.9604644775390625e-08f;
}
static inline double u32_to_jitter(uint32_t w) {
    return ((double)w + 0.5) * 2.3283064365386962890625e-10 - 0.5;
}

/* ------------------------------------------------------------------ */
/* Portable sin/cos of a float angle in [0, ~6.2832] (double +,-,* only) */
/* ------------------------------------------------------------------ */
static void portable_sincosf(float a, float *s_out, float *c_out) {
    const double x = (double)a;
    const int k = (int)(x * 0.63661977236758138 + 0.5);          /* nearest multiple of pi/2 */
    const double kd = (double)k;
    const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04
                    + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double sn = r + r * (z * ps);
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05
                    + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

/* Portable atan / atan2 / acos for the skybox lookup (scene.cpp:127-128): double +,-,*,/,sqrt only.
 * atan follows the classic argument reduction to [0, 7/16] around 0.5, 1, 1.5, inf with an odd polynomial. */
static double portable_atan_pos(double x) {       /* x >= 0, finite or +inf */
    static const double hi[4] = { 4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00 };
    static const double lo[4] = { 2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17 };
    int id;
    if (x < 0.4375) id = -1;
    else if (x < 1.1875) { if (x < 0.6875) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); } else { id = 1; x = (x - 1.0) / (x + 1.0); } }
    else if (x < 2.4375) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else { id = 3; x = -1.0 / x; }
    const double z = x * x, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02
                    + w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02
                    + w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    if (id < 0) return x - x * (s1 + s2);
    return hi[id] - ((x * (s1 + s2) - lo[id]) - x);
}
static float portable_atan2f(float yf, float xf) {
    const double y = (double)yf, x = (double)xf;
    if (y != y || x != x) return NAN;
    const double pi = 3.14159265358979311600e+00, pi_2 = 1.57079632679489655800e+00;
    const double ay = y < 0 ? -y : y, ax = x < 0 ? -x : x;
    double r;
    if (ay == 0.0) r = (x < 0 || (x == 0 && signbit(xf))) ? pi : 0.0;
    else if (ax == 0.0) r = pi_2;
    else {
        const double t = portable_atan_pos(ay / ax);
        r = x < 0 ? pi - t : t;
    }
    return (float)((y < 0 || (y == 0 && signbit(yf))) ? -r : r);
}
static float portable_acosf(float vf) {
    const double v = (double)vf;
    const double s = sqrt((1.0 - v) * (1.0 + v));       /* NaN for |v| > 1, like acosf */
    if (s != s) return NAN;
    const double pi = 3.14159265358979311600e+00, pi_2 = 1.57079632679489655800e+00;
    double r;
    if (v == 0.0) r = pi_2;
    else {
        const double t = portable_atan_pos(s / (v < 0 ? -v : v));
        r = v < 0 ? pi - t : t;
    }
    return (float)r;
}

/* ------------------------------------------------------------------ */
/* Scene set-up                                                         */
/* ------------------------------------------------------------------ */
/* Triangle::SetNormal, triangles.h:40-44 */
static void tri_set_normal(orc_triangle *t, const float *n_in) {
    float n[3] = { n_in[0], n_in[1], n_in[2] };
    float inv = 1.0f / sqrtf(dot3f(n, n));
    n[0] = n[0] * inv; n[1] = n[1] * inv; n[2] = n[2] * inv;
    t->plane[0] = n[0]; t->plane[1] = n[1]; t->plane[2] = n[2];
    t->plane[3] = -dot3f(n, t->v[0]);
}
/* Triangle ctor, triangles.h:27-36 */
static void tri_init(orc_triangle *t, const float *v0, const float *v1, const float *v2, int material) {
    memcpy(t->v[0], v0, 12); memcpy(t->v[1], v1, 12); memcpy(t->v[2], v2, 12);
    float ab[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] };
    float ac[3] = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] };
    float c[3];
    cross3f(ab, ac, c);
    tri_set_normal(t, c);
    t->square = length3f(c);
    t->material = material;
}
/* Factory, material.h:58-106: which lobes exist and with what probability. */
static void mat_build(orc_material *m) {
    m->n_lobes = 0;
    int ke_nonzero = (m->Ke[0] != 0.0f) || (m->Ke[1] != 0.0f) || (m->Ke[2] != 0.0f);
    int ks_nonzero = (m->Ks[0] != 0.0f) || (m->Ks[1] != 0.0f) || (m->Ks[2] != 0.0f);
    if (ke_nonzero) {
        m->lobe_kind[0] = 0; m->chance[0] = 1.0f; m->n_lobes = 1;
    } else {
        if (m->Ns != 0.0f && ks_nonzero) {
            m->lobe_kind[m->n_lobes] = 1; m->chance[m->n_lobes] = m->Ns / 1000; m->n_lobes++;
        }
        if (1 - m->Ns / 1000 > 0) {
            m->lobe_kind[m->n_lobes] = 2; m->chance[m->n_lobes] = 1 - m->Ns / 1000; m->n_lobes++;
        }
    }
}

void orc_scene_free(orc_scene *s) {
    if (!s) return;
    free(s->tri); free(s->mat); free(s->sky); free(s);
}

/* bitmap_image(filename) -> load_bitmap (bitmap_image.hpp:1508-1603), as Scene's constructor uses it for -SKYBOX
 * (scene.cpp:20-22).  Returns 0 on success; on any of the reference's load errors the scene keeps no skybox
 * (the reference would go on with a 0x0 image and divide by zero at the first miss). */
int orc_scene_set_skybox(orc_scene *s, const char *path) {
    free(s->sky); s->sky = NULL; s->sky_w = s->sky_h = 0;
    if (!path || !*path) return 0;
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    unsigned char h[54];
    if (fread(h, 1, 54, f) != 54) { fclose(f); return 2; }
    uint16_t type, bit_count; uint32_t bih_size, w, hgt;
    memcpy(&type, h, 2); memcpy(&bih_size, h + 14, 4); memcpy(&w, h + 18, 4); memcpy(&hgt, h + 22, 4); memcpy(&bit_count, h + 28, 2);
    if (type != 19778 || bit_count != 24 || bih_size != 40) { fclose(f); return 3; }
    const unsigned pad = (4 - ((3 * w) % 4)) % 4;
    fseek(f, 0, SEEK_END);
    const size_t physical = (size_t)ftell(f);
    const size_t logical = (size_t)hgt * w * 3 + (size_t)hgt * pad + 40 + 14;
    if (physical != logical || w == 0 || hgt == 0) { fclose(f); return 4; }
    fseek(f, 54, SEEK_SET);
    s->sky = (unsigned char *)malloc((size_t)w * hgt * 3);
    unsigned char padbuf[4];
    for (uint32_t i = 0; i < hgt; ++i) {                                  /* rows are stored bottom-up */
        if (fread(s->sky + (size_t)(hgt - i - 1) * w * 3, 1, (size_t)w * 3, f) != (size_t)w * 3) { fclose(f); free(s->sky); s->sky = NULL; return 5; }
        if (pad && fread(padbuf, 1, pad, f) != pad) { fclose(f); free(s->sky); s->sky = NULL; return 5; }
    }
    fclose(f);
    s->sky_w = (int)w; s->sky_h = (int)hgt;
    return 0;
}

/* Build from flat arrays: tri14 = plane[4], v0,v1,v2[9], square[1]; mats = Kd,Ke,Ks,Ns (10 floats). */
orc_scene *orc_scene_from_arrays(const float *tri14, const int *tri_mat, int n_tri, const float *mats10, int n_mat) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->n_tri = n_tri; s->n_mat = n_mat;
    s->tri = (orc_triangle *)calloc((size_t)(n_tri > 0 ? n_tri : 1), sizeof(orc_triangle));
    s->mat = (orc_material *)calloc((size_t)(n_mat > 0 ? n_mat : 1), sizeof(orc_material));
    for (int i = 0; i < n_tri; ++i) {
        const float *p = tri14 + 14 * (size_t)i;
        memcpy(s->tri[i].plane, p, 16);
        memcpy(s->tri[i].v, p + 4, 36);
        s->tri[i].square = p[13];
        s->tri[i].material = tri_mat[i];
    }
    for (int i = 0; i < n_mat; ++i) {
        const float *p = mats10 + 10 * (size_t)i;
        memcpy(s->mat[i].Kd, p, 12); memcpy(s->mat[i].Ke, p + 3, 12); memcpy(s->mat[i].Ks, p + 6, 12);
        s->mat[i].Ns = p[9];
        mat_build(&s->mat[i]);
    }
    return s;
}

/* --- whitespace-token reader equivalent to `stream >> std::string` --- */
typedef struct { char *buf; size_t len, pos; int failed; } tokstream;
static int ts_open(tokstream *ts, const char *path) {
    memset(ts, 0, sizeof(*ts));
    FILE *f = fopen(path, "rb");
    if (!f) { ts->failed = 1; return 0; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    ts->buf = (char *)malloc((size_t)n + 1);
    ts->len = fread(ts->buf, 1, (size_t)n, f);
    ts->buf[ts->len] = 0;
    fclose(f);
    return 1;
}
static int ts_next(tokstream *ts, char *out, size_t cap) {
    while (ts->pos < ts->len && isspace((unsigned char)ts->buf[ts->pos])) ts->pos++;
    if (ts->pos >= ts->len) { out[0] = 0; return 0; }
    size_t n = 0;
    while (ts->pos < ts->len && !isspace((unsigned char)ts->buf[ts->pos])) {
        if (n + 1 < cap) out[n++] = ts->buf[ts->pos];
        ts->pos++;
    }
    out[n] = 0;
    return 1;
}
/* istream::eof(): set once a read has touched the end of the file (also by a token that ends exactly there). */
static int ts_eof(const tokstream *ts) { return ts->failed || ts->pos >= ts->len; }
static float ts_float(tokstream *ts) {   /* `stream >> float` */
    char tok[128];
    if (!ts_next(ts, tok, sizeof tok)) return 0.0f;
    return strtof(tok, NULL);
}

/* Scene::LoadModel, scene.cpp:26-109.  `dir` is Config::model_path (with trailing slash). */
orc_scene *orc_scene_load(const char *dir, const char *name) {
    char path[4096];
    snprintf(path, sizeof path, "%s%s", dir, name);
    tokstream ts;
    if (!ts_open(&ts, path)) return NULL;                       /* scene.cpp:32-35 (reference exits) */
    size_t cap_v = 1024, cap_n = 1024, cap_t = 1024, cap_m = 16;
    size_t nv = 0, nn = 0, nt = 0, nm = 0;
    float *V = (float *)malloc(cap_v * 12), *N = (float *)malloc(cap_n * 12);
    orc_triangle *T = (orc_triangle *)malloc(cap_t * sizeof(orc_triangle));
    orc_material *M = (orc_material *)calloc(cap_m, sizeof(orc_material));
    int current_material = 0;
    char tok[4096];
    while (!ts_eof(&ts)) {                                      /* scene.cpp:37-41 */
        ts_next(&ts, tok, sizeof tok);
        if (ts_eof(&ts)) break;
        if (!strcmp(tok, "mtllib")) {                           /* scene.cpp:41-71 */
            char mtl[2048];
            ts_next(&ts, mtl, sizeof mtl);
            snprintf(path, sizeof path, "%s%s", dir, mtl);
            tokstream ms;
            if (!ts_open(&ms, path)) continue;                  /* reference would spin forever */
            char mt[256] = "1";
            /* scene.cpp:45-71, including its end-of-file behaviour: a material is appended
             * every time the outer loop body runs, even when no `newmtl` was found. */
            while (!ts_eof(&ms)) {
                orc_material c; memset(&c, 0, sizeof c);
                while (!ts_eof(&ms) && strcmp(mt, "newmtl")) ts_next(&ms, mt, sizeof mt);
                ts_next(&ms, mt, sizeof mt);                      /* the material's name (handled like any token below) */
                while (!ts_eof(&ms) && strcmp(mt, "newmtl")) {
                    if (!strcmp(mt, "Kd")) { c.Kd[0] = ts_float(&ms); c.Kd[1] = ts_float(&ms); c.Kd[2] = ts_float(&ms); }
                    else if (!strcmp(mt, "Ke")) { c.Ke[0] = ts_float(&ms); c.Ke[1] = ts_float(&ms); c.Ke[2] = ts_float(&ms); }
                    else if (!strcmp(mt, "Ks")) { c.Ks[0] = ts_float(&ms); c.Ks[1] = ts_float(&ms); c.Ks[2] = ts_float(&ms); }
                    else if (!strcmp(mt, "Ns")) { c.Ns = ts_float(&ms); }
                    ts_next(&ms, mt, sizeof mt);
                }
                if (nm == cap_m) { cap_m *= 2; M = (orc_material *)realloc(M, cap_m * sizeof(orc_material)); }
                mat_build(&c);
                M[nm++] = c;
            }
            free(ms.buf);
        } else if (!strcmp(tok, "v")) {                         /* scene.cpp:73-76 */
            if (nv == cap_v) { cap_v *= 2; V = (float *)realloc(V, cap_v * 12); }
            V[3 * nv] = ts_float(&ts); V[3 * nv + 1] = ts_float(&ts); V[3 * nv + 2] = ts_float(&ts); nv++;
        } else if (!strcmp(tok, "vt")) {                        /* scene.cpp:77-80 */
            (void)ts_float(&ts); (void)ts_float(&ts);
        } else if (!strcmp(tok, "vn")) {                        /* scene.cpp:81-84 */
            if (nn == cap_n) { cap_n *= 2; N = (float *)realloc(N, cap_n * 12); }
            N[3 * nn] = ts_float(&ts); N[3 * nn + 1] = ts_float(&ts); N[3 * nn + 2] = ts_float(&ts); nn++;
        } else if (!strcmp(tok, "f")) {                         /* scene.cpp:85-104 */
            int vi[3], ni[3];
            for (int i = 0; i < 3; ++i) {
                char cr[256];
                ts_next(&ts, cr, sizeof cr);
                /* Split(cr,'/') then resize(3): fields 0 and 2, atoi()-1 */
                const char *f0 = cr, *f2 = "";
                char *s1 = strchr(cr, '/');
                if (s1) { *s1 = 0; char *s2 = strchr(s1 + 1, '/'); if (s2) { *s2 = 0; f2 = s2 + 1; char *s3 = strchr(s2 + 1, '/'); if (s3) *s3 = 0; } }
                vi[i] = atoi(f0) - 1;
                ni[i] = atoi(f2) - 1;
            }
            int ok = 1;
            for (int i = 0; i < 3; ++i) if (vi[i] < 0 || (size_t)vi[i] >= nv) ok = 0;
            if (!ok || current_material < 0 || (size_t)current_material >= nm) continue;   /* UB in the reference */
            if (nt == cap_t) { cap_t *= 2; T = (orc_triangle *)realloc(T, cap_t * sizeof(orc_triangle)); }
            tri_init(&T[nt], V + 3 * vi[0], V + 3 * vi[1], V + 3 * vi[2], current_material);
            if (ni[0] >= 0 && (size_t)ni[0] < nn) tri_set_normal(&T[nt], N + 3 * ni[0]);    /* scene.cpp:102-104 */
            nt++;
        } else if (!strcmp(tok, "usemtl")) {                    /* scene.cpp:105-106: `file >> int` */
            char mt[256];
            ts_next(&ts, mt, sizeof mt);
            current_material = atoi(mt);
        }
    }
    free(ts.buf); free(V); free(N);
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->tri = T; s->n_tri = (int)nt; s->mat = M; s->n_mat = (int)nm;
    return s;
}

int orc_scene_num_triangles(const orc_scene *s) { return s->n_tri; }
int orc_scene_num_materials(const orc_scene *s) { return s->n_mat; }
void orc_scene_get_triangles(const orc_scene *s, float *tri14, int *tri_mat) {
    for (int i = 0; i < s->n_tri; ++i) {
        float *p = tri14 + 14 * (size_t)i;
        memcpy(p, s->tri[i].plane, 16); memcpy(p + 4, s->tri[i].v, 36); p[13] = s->tri[i].square;
        tri_mat[i] = s->tri[i].material;
    }
}
void orc_scene_get_materials(const orc_scene *s, float *mats10) {
    for (int i = 0; i < s->n_mat; ++i) {
        float *p = mats10 + 10 * (size_t)i;
        memcpy(p, s->mat[i].Kd, 12); memcpy(p + 3, s->mat[i].Ke, 12); memcpy(p + 6, s->mat[i].Ks, 12); p[9] = s->mat[i].Ns;
    }
}

/* ------------------------------------------------------------------ */
/* Intersection                                                         */
/* ------------------------------------------------------------------ */
/* Triangle::Intersect, triangles.h:48-73 (PlaneIntersect :10-13, ParallelogramSquare :15-17).
 * Returns the stage at which the test ended: 0=A 1=B 2=C 3=D-reject 4=accept. */
static inline int tri_intersect(const orc_triangle *t, const float *o, const float *d, float eps, float *distance) {
    const float *p = t->plane;
    const float signed_dist = d[0] * p[0] + d[1] * p[1] + d[2] * p[2];
    const float nd = -(o[0] * p[0] + o[1] * p[1] + o[2] * p[2] + p[3]) / signed_dist;
    if (nd >= *distance || nd < eps) return 0;
    const float P[3] = { o[0] + d[0] * nd, o[1] + d[1] * nd, o[2] + d[2] * nd };
    const float f0[3] = { P[0] - t->v[0][0], P[1] - t->v[0][1], P[2] - t->v[0][2] };
    const float f1[3] = { P[0] - t->v[1][0], P[1] - t->v[1][1], P[2] - t->v[1][2] };
    const float f2[3] = { P[0] - t->v[2][0], P[1] - t->v[2][1], P[2] - t->v[2][2] };
    float c[3];
    cross3f(f0, f1, c);
    const float s1 = length3f(c);
    if (s1 > t->square + eps) return 1;
    cross3f(f0, f2, c);
    const float s2 = length3f(c);
    if (s1 + s2 > t->square + eps) return 2;
    cross3f(f2, f1, c);
    const float s3 = length3f(c);
    if (fabsf(t->square - s1 - s2 - s3) > eps) return 3;
    *distance = nd;
    return 4;
}

/* Closest hit, scene.cpp:114-120.  Exposed for the intersection-vector tests. */
int orc_closest_hit(const orc_scene *s, const float *o, const float *d, float eps, float *t_out) {
    float distance = INFINITY;
    int cur = -1;
    for (int i = 0; i < s->n_tri; ++i)
        if (tri_intersect(&s->tri[i], o, d, eps, &distance) == 4) cur = i;
    *t_out = distance;
    return cur;
}

/* 1 if some triangle's plane evaluation is not a finite number for this ray (0/0: the ray lies exactly in the stored
 * plane).  The reference then accepts that triangle whatever the geometry (every comparison with NaN is false). */
static int ray_sees_nan(const orc_scene *s, const float *o, const float *d) {
    for (int i = 0; i < s->n_tri; ++i) {
        const float *p = s->tri[i].plane;
        const float sd = d[0] * p[0] + d[1] * p[1] + d[2] * p[2];
        const float nd = -(o[0] * p[0] + o[1] * p[1] + o[2] * p[2] + p[3]) / sd;
        if (nd != nd) return 1;
    }
    return 0;
}

void orc_closest_hit_batch(const orc_scene *s, int n, const float *o, const float *d, float eps, int *idx, float *t,
                           unsigned char *nan_seen, int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel for schedule(static, 256) num_threads(threads)
    for (int i = 0; i < n; ++i) {
        idx[i] = orc_closest_hit(s, o + 3 * (size_t)i, d + 3 * (size_t)i, eps, t + i);
        if (nan_seen) nan_seen[i] = (unsigned char)ray_sees_nan(s, o + 3 * (size_t)i, d + 3 * (size_t)i);
    }
}

/* ------------------------------------------------------------------ */
/* Path state and shading                                               */
/* ------------------------------------------------------------------ */
typedef struct {
    float o[3], d[3];    /* ray.h:13-14 (w components are constant 1 / 0) */
    int depth;           /* ray.h:15 */
    float color[3];      /* ray.h:17 throughput */
} orc_ray;

typedef struct {
    const orc_scene *sc;
    const orc_params *pr;
    minstd0 *mat_engine;       /* sequential policy: material.h:17 */
    uint32_t rnd[3];           /* counter policy: words for this segment */
    float *sum, *sum2;         /* this pixel's accumulators */
    int *count;
    orc_stats *st;
} orc_ctx;

static inline int ray_valid(const orc_ray *r, int mrr) {   /* ray.h:52-54 */
    return r->depth < mrr && (r->color[0] != 0.0f || r->color[1] != 0.0f || r->color[2] != 0.0f);
}
/* Ray::Reflect, ray.h:45-50 */
static inline void ray_reflect(orc_ray *r, const float *begin, const float *dir, const float *w) {
    r->o[0] = begin[0]; r->o[1] = begin[1]; r->o[2] = begin[2];
    r->d[0] = dir[0]; r->d[1] = dir[1]; r->d[2] = dir[2];
    normalize_dir(r->d);
    r->color[0] *= w[0]; r->color[1] *= w[1]; r->color[2] *= w[2];
    r->depth++;
}

/* The skybox lookup of Scene::TraceRay's miss branch (scene.cpp:126-149): direction -> (r, g, b) / 256 of the bilinear
 * sample, with the weights as written there. */
static void sky_sample(const orc_scene *sc, const float *d, int trig_policy, float *out) {
    const float pi = 3.141593f;
    float ac, at;
    if (trig_policy == ORC_TRIG_LIBM) { ac = acosf(d[1]); at = atan2f(d[2], -d[0]); }
    else { ac = portable_acosf(d[1]); at = portable_atan2f(d[2], -d[0]); }
    const float theta = ac / pi;
    const float phi = at / pi / 2 + 0.5f;
    const float x = phi * (float)(unsigned)sc->sky_w, y = theta * (float)(unsigned)sc->sky_h;
    /* static_cast<unsigned>(float): undefined for NaN / out of range in the reference; here such values,
     * and a coordinate that lands on the last row/column + 1, are clamped into the image */
    unsigned x1 = (x >= 0.0f) ? (x < 4294967040.0f ? (unsigned)x : 0xFFFFFFFFu) : 0u;
    unsigned y1 = (y >= 0.0f) ? (y < 4294967040.0f ? (unsigned)y : 0xFFFFFFFFu) : 0u;
    if (x1 > (unsigned)sc->sky_w - 1u) x1 = (unsigned)sc->sky_w - 1u;
    if (y1 > (unsigned)sc->sky_h - 1u) y1 = (unsigned)sc->sky_h - 1u;
    const unsigned x2 = (x1 + 1) % (unsigned)sc->sky_w, y2 = (y1 + 1) % (unsigned)sc->sky_h;
    const unsigned char *t1 = sc->sky + ((size_t)y1 * sc->sky_w + x1) * 3, *t2 = sc->sky + ((size_t)y1 * sc->sky_w + x2) * 3;
    const unsigned char *t3 = sc->sky + ((size_t)y2 * sc->sky_w + x1) * 3, *t4 = sc->sky + ((size_t)y2 * sc->sky_w + x2) * 3;
    const float ax = 1 - x + (float)x1, ay = 1 - y + (float)y1;   /* the weights of scene.cpp:146-149, as written */
    for (int k = 0; k < 3; ++k) {                                   /* k: r,g,b = bytes 2,1,0 */
        const float c1 = (float)t1[2 - k], c2 = (float)t2[2 - k], c3 = (float)t3[2 - k], c4 = (float)t4[2 - k];
        const float c12 = c1 * (1.0f - ax) + c2 * ax;             /* glm::mix(x, y, a) = x*(1-a) + y*a */
        const float c34 = c3 * (1.0f - ax) + c4 * ax;
        out[k] = (c12 * (1.0f - ay) + c34 * ay) / 256.f;
    }
}
/* the same for caller-supplied directions (tests: tests/test_double_entry.py); returns 0, or -1 without a skybox */
int orc_probe_skybox(const orc_scene *sc, int n, const float *dirs, int trig_policy, float *rgb) {
    if (!sc->sky) return -1;
    for (int i = 0; i < n; ++i) sky_sample(sc, dirs + 3 * i, trig_policy, rgb + 3 * i);
    return 0;
}

/* Scene::TraceRay (scene.cpp:113-157, no-skybox branch) + Material::Process (material.h:36-50)
 * + the three lobes (material.h:67-102). */
static void trace_segment(orc_ctx *cx, orc_ray *r) {
    const orc_scene *sc = cx->sc;
    const float eps = cx->pr->eps;
    const int mrr = cx->pr->max_ray_reflections;
    float distance = INFINITY;
    int cur = -1;
    for (int i = 0; i < sc->n_tri; ++i) {
        int stage = tri_intersect(&sc->tri[i], r->o, r->d, eps, &distance);
        cx->st->stage_exit[stage]++;
        if (stage == 4) cur = i;
    }
    cx->st->segments++;
    if (cur < 0) {                                                          /* scene.cpp:125-156 */
        cx->st->misses++;
        if (sc->sky) {
            float c3[3];
            sky_sample(sc, r->d, cx->pr->trig_policy, c3);
            for (int k = 0; k < 3; ++k) { cx->sum[k] += c3[k]; cx->sum2[k] += c3[k] * c3[k]; }
            ++*cx->count;
            cx->st->contributing++;
        }
        r->depth = mrr;
        return;
    }
    const orc_triangle *t = &sc->tri[cur];
    const float P[3] = { r->o[0] + r->d[0] * distance, r->o[1] + r->d[1] * distance, r->o[2] + r->d[2] * distance };
    const float *Nn = t->plane;
    const orc_material *m = &sc->mat[t->material];
    int kind;
    int use_counter = cx->pr->rng_policy == ORC_RNG_COUNTER;
    if (m->n_lobes == 0) { r->depth = mrr; return; }                      /* material.h:37-38 */
    if (m->n_lobes == 1) {
        kind = m->lobe_kind[0];                                            /* material.h:39-40: no random consumed */
    } else {
        float sample = use_counter ? u32_to_unit_float(cx->rnd[0]) : libstd_canonical_float(cx->mat_engine);
        int i = -1;                                                        /* material.h:42-47 */
        while (sample > 0) {
            ++i;
            if (i >= m->n_lobes) { i = m->n_lobes - 1; break; }           /* reference: UB (reads past chance_) */
            sample -= m->chance[i];
        }
        if (i < 0) i = 0;                                                  /* sample == 0: reference indexes [-1] (UB) */
        kind = m->lobe_kind[i];
    }
    if (kind == 0) {                                                       /* emissive, material.h:68-79 */
        if ((r->d[0] * Nn[0] + r->d[1] * Nn[1]) + r->d[2] * Nn[2] > 0) { r->depth = mrr; return; }
        float c[3] = { r->color[0] * m->Kd[0], r->color[1] * m->Kd[1], r->color[2] * m->Kd[2] };
        for (int k = 0; k < 3; ++k) { cx->sum[k] += c[k]; cx->sum2[k] += c[k] * c[k]; }
        ++*cx->count;
        cx->st->contributing++;
        r->depth = mrr;
    } else if (kind == 1) {                                                /* glossy, material.h:83-85 */
        float dn = (Nn[0] * r->d[0] + Nn[1] * r->d[1]) + Nn[2] * r->d[2];
        float dir[3], begin[3];
        for (int k = 0; k < 3; ++k) {
            dir[k] = r->d[k] - Nn[k] * dn * 2.0f;
            begin[k] = P[k] + Nn[k] * eps;
        }
        ray_reflect(r, begin, dir, m->Ks);
    } else {                                                               /* diffuse, material.h:90-100 */
        float xi1, xi2;
        if (use_counter) { xi1 = u32_to_unit_float(cx->rnd[1]); xi2 = u32_to_unit_float(cx->rnd[2]); }
        else { xi1 = libstd_canonical_float(cx->mat_engine); xi2 = libstd_canonical_float(cx->mat_engine); }
        const float pi = 3.141593f;                                        /* material.h:12 */
        const float ang = 2 * pi * xi2;
        float sn, cs;
        if (cx->pr->trig_policy == ORC_TRIG_LIBM) { cs = cosf(ang); sn = sinf(ang); }
        else portable_sincosf(ang, &sn, &cs);
        const float sq = sqrtf(xi1);
        float rnd[3] = { sq * cs, sq * sn, sqrtf(1 - xi1) };
        normalize_dir(rnd);
        if ((Nn[0] * rnd[0] + Nn[1] * rnd[1]) + Nn[2] * rnd[2] < 0) { rnd[0] *= -1; rnd[1] *= -1; rnd[2] *= -1; }
        float dt = (Nn[0] * rnd[0] + Nn[1] * rnd[1]) + Nn[2] * rnd[2];
        dt = dt > 0.0f ? dt : 0.0f;                                        /* std::max(0.0f, dot) */
        float w[3] = { m->Kd[0] * dt, m->Kd[1] * dt, m->Kd[2] * dt };
        float begin[3] = { P[0] + Nn[0] * eps, P[1] + Nn[1] * eps, P[2] + Nn[2] * eps };
        ray_reflect(r, begin, rnd, w);
    }
}

/* Adaptive-sampling skip test, main.cpp:118-125. */
static inline int adaptive_skip(int pass, const float *c, const float *c2, int n, float error) {
    const float sc = (float)n;
    if (pass > 10 && sc > 0) {
        float D[3];
        for (int k = 0; k < 3; ++k) { float m = c[k] / sc; D[k] = c2[k] / sc - m * m; }
        if ((pass % 4) && D[0] < error && D[1] < error && D[2] < error) return 1;
    }
    return 0;
}
/* Primary ray, main.cpp:126-129 + Ray ctor ray.h:21-25. */
static inline void primary_ray(orc_ray *r, int x, int y, double jx, double jy, int W, int H) {
    r->o[0] = 0; r->o[1] = 0; r->o[2] = -20;
    r->d[0] = (float)((x + jx) / W - 0.5f);
    r->d[1] = (float)(-(y + jy) / H + 0.5f);
    r->d[2] = 1.0f;
    float inv = 1.0f / sqrtf((r->d[0] * r->d[0] + r->d[1] * r->d[1]) + (1.0f * 1.0f + 0.0f * 0.0f));
    r->d[0] *= inv; r->d[1] *= inv; r->d[2] *= inv;
    r->depth = 0;
    r->color[0] = r->color[1] = r->color[2] = 1.0f;
}

static void stats_add(orc_stats *a, const orc_stats *b) {
    a->samples_traced += b->samples_traced; a->segments += b->segments; a->contributing += b->contributing;
    a->misses += b->misses;
    for (int i = 0; i < 5; ++i) a->stage_exit[i] += b->stage_exit[i];
}

/* Frame loop, main.cpp:110-140.  sum/sum2: [(row_end-row_begin)*W*3] row-major, count: [(row_end-row_begin)*W].
 * Buffers are accumulated into (caller zeroes them for a fresh frame). */
int orc_render(const orc_scene *sc, const orc_params *pr, float *sum, float *sum2, int *count, orc_stats *stats_out) {
    const int W = pr->width, H = pr->height;
    orc_stats total; memset(&total, 0, sizeof total);
    if (W <= 0 || H <= 0 || pr->row_begin < 0 || pr->row_end > H || pr->row_begin > pr->row_end) return 1;
    if (pr->rng_policy == ORC_RNG_SEQUENTIAL) {
        if (pr->row_begin != 0 || pr->row_end != H || pr->pass_begin != 0) return 2;   /* stream order is whole-frame */
        minstd0 jitter, material;
        minstd0_seed(&jitter, pr->seed);      /* main.cpp:91 */
        minstd0_seed(&material, pr->seed);    /* material.h:17 */
        orc_ray *rays = (orc_ray *)calloc((size_t)W * H, sizeof(orc_ray));
        /* main.cpp:107-108: the default-constructed rays are all overwritten on pass 0 (the adaptive
         * test needs pass > 10), so their initial state is never observed. */
        for (size_t i = 0; i < (size_t)W * H; ++i) rays[i].depth = pr->max_ray_reflections;
        for (int pass = 0; pass < pr->pass_count; ++pass) {
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {            /* main.cpp:116-131 */
                size_t p = (size_t)y * W + x;
                if (adaptive_skip(pass, sum + 3 * p, sum2 + 3 * p, count[p], pr->error)) continue;
                /* g++ evaluates the vec4 ctor arguments right to left: the y jitter is drawn first */
                double jy = libstd_jitter_double(&jitter);
                double jx = libstd_jitter_double(&jitter);
                primary_ray(&rays[p], x, y, jx, jy, W, H);
                total.samples_traced++;
            }
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {            /* main.cpp:132-140 */
                size_t p = (size_t)y * W + x;
                orc_ctx cx = { sc, pr, &material, {0, 0, 0}, sum + 3 * p, sum2 + 3 * p, count + p, &total };
                while (ray_valid(&rays[p], pr->max_ray_reflections)) trace_segment(&cx, &rays[p]);
            }
        }
        free(rays);
    } else {
        int nthreads = pr->threads;
#ifdef _OPENMP
        if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
        nthreads = 1;
#endif
        const uint32_t key[2] = { pr->seed, ORC_PHILOX_KEY1 };
        const int rows = pr->row_end - pr->row_begin;
#pragma omp parallel num_threads(nthreads)
        {
            orc_stats local; memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
            for (int ry = 0; ry < rows; ++ry) {
                const int y = pr->row_begin + ry;
                for (int x = 0; x < W; ++x) {
                    const size_t p = (size_t)ry * W + x;
                    const uint32_t gpix = (uint32_t)((size_t)y * W + x);
                    /* pixel-major order: legal because every quantity is pixel-local under the counter RNG */
                    for (int pass = pr->pass_begin; pass < pr->pass_begin + pr->pass_count; ++pass) {
                        if (adaptive_skip(pass, sum + 3 * p, sum2 + 3 * p, count[p], pr->error)) continue;
                        uint32_t ctr[4] = { gpix, (uint32_t)pass, 0xFFFFFFFFu, 0u }, w[4];
                        philox4x32_10(ctr, key, w);
                        orc_ray r;
                        primary_ray(&r, x, y, u32_to_jitter(w[0]), u32_to_jitter(w[1]), W, H);
                        local.samples_traced++;
                        orc_ctx cx = { sc, pr, NULL, {0, 0, 0}, sum + 3 * p, sum2 + 3 * p, count + p, &local };
                        while (ray_valid(&r, pr->max_ray_reflections)) {
                            ctr[2] = (uint32_t)r.depth;
                            philox4x32_10(ctr, key, w);
                            cx.rnd[0] = w[0]; cx.rnd[1] = w[1]; cx.rnd[2] = w[2];
                            trace_segment(&cx, &r);
                        }
                    }
                }
            }
#pragma omp critical
            stats_add(&total, &local);
        }
    }
    if (stats_out) *stats_out = total;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Resolve + BMP                                                        */
/* ------------------------------------------------------------------ */
/* main.cpp:162-201 + bitmap_image::set_pixel (bitmap_image.hpp:194-206, float->uchar truncation).
 * bgr: H*W*3 bytes, top-down rows, zero-initialised here (image.clear(), main.cpp:106).
 * disp[0..2] = max, min, average dispersion (main.cpp:162-185). */
void orc_resolve(int W, int H, const float *sum, const float *sum2, const int *count, float gamma,
                 unsigned char *bgr, float *disp) {
    float max_d = 0, min_d = INFINITY, avg_d = 0;
    memset(bgr, 0, (size_t)W * H * 3);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        size_t p = (size_t)y * W + x;
        if (!count[p]) { avg_d += 1; continue; }
        const float n = (float)count[p];
        float D[3], c[3];
        for (int k = 0; k < 3; ++k) {
            float m = sum[3 * p + k] / n;
            D[k] = sum2[3 * p + k] / n - m * m;
            c[k] = powf(sum[3 * p + k] / n, gamma) * 255.0f;
        }
        const float d = D[0] + D[1] + D[2];
        if (d > max_d) max_d = d;
        if (d < min_d) min_d = d;
        avg_d += d;
        bgr[3 * p + 0] = (unsigned char)(int)c[2];
        bgr[3 * p + 1] = (unsigned char)(int)c[1];
        bgr[3 * p + 2] = (unsigned char)(int)c[0];
    }
    avg_d /= W * H;
    disp[0] = max_d; disp[1] = min_d; disp[2] = avg_d;
}

/* main.cpp:162-185 without the byte conversion: the tonemapped float image the post filters work on.
 * rgb: H*W*3 floats, row-major; pixels without samples keep their raw sums (zero), as color_map does. */
void orc_resolve_float(int W, int H, const float *sum, const float *sum2, const int *count, float gamma, float *rgb, float *disp) {
    float max_d = 0, min_d = INFINITY, avg_d = 0;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        size_t p = (size_t)y * W + x;
        for (int k = 0; k < 3; ++k) rgb[3 * p + k] = sum[3 * p + k];
        if (!count[p]) { avg_d += 1; continue; }
        const float n = (float)count[p];
        float D[3];
        for (int k = 0; k < 3; ++k) {
            float m = sum[3 * p + k] / n;
            D[k] = sum2[3 * p + k] / n - m * m;
            rgb[3 * p + k] = powf(sum[3 * p + k] / n, gamma) * 255.0f;
        }
        const float d = D[0] + D[1] + D[2];
        if (d > max_d) max_d = d;
        if (d < min_d) min_d = d;
        avg_d += d;
    }
    avg_d /= W * H;
    if (disp) { disp[0] = max_d; disp[1] = min_d; disp[2] = avg_d; }
}
/* GaussBlur, main.cpp:11-33 (clamp-to-edge taps, weights exp(-d2/(2r^2))/(2 pi r^2) in float, glm::round). */
void orc_gauss_blur(int W, int H, const float *in, float r, float *out) {
    const float pi = 3.141593f;
    const int rs = (int)ceil(r * 2.57);
    for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) {
        float val[3] = { 0, 0, 0 }, wsum = 0;
        for (int iy = i - rs; iy <= i + rs; ++iy) for (int ix = j - rs; ix <= j + rs; ++ix) {
            const int x = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix), y = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
            const int dsq = (ix - j) * (ix - j) + (iy - i) * (iy - i);
            const float wght = expf(-dsq / (2 * r * r)) / (pi * 2 * r * r);
            for (int k = 0; k < 3; ++k) val[k] += in[3 * ((size_t)y * W + x) + k] * wght;
            wsum += wght;
        }
        for (int k = 0; k < 3; ++k) out[3 * ((size_t)i * W + j) + k] = roundf(val[k] / wsum);
    }
}
static int cmp_float(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
/* MedianFilter, main.cpp:49-80: element window_size*window_size/2 of the sorted (2w+1)^2 window (sic). */
void orc_median_filter(int W, int H, const float *in, int ws, float *out) {
    const int n = (2 * ws + 1) * (2 * ws + 1);
    float *win = (float *)malloc(sizeof(float) * (size_t)n);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int k = 0; k < 3; ++k) {
        int m = 0;
        for (int wx = -ws; wx < ws + 1; ++wx) for (int wy = -ws; wy < ws + 1; ++wy) {
            const int i = wx + x > W - 1 ? W - 1 : (wx + x < 0 ? 0 : wx + x), j = wy + y > H - 1 ? H - 1 : (wy + y < 0 ? 0 : wy + y);
            win[m++] = in[3 * ((size_t)j * W + i) + k];
        }
        qsort(win, (size_t)n, sizeof(float), cmp_float);
        out[3 * ((size_t)y * W + x) + k] = win[ws * ws / 2];
    }
    free(win);
}
/* main.cpp:193-201 + set_pixel's float -> uchar conversion: only pixels with samples are written. */
void orc_quantize(int W, int H, const float *rgb, const int *count, unsigned char *bgr) {
    memset(bgr, 0, (size_t)W * H * 3);
    for (size_t p = 0; p < (size_t)W * H; ++p) {
        if (!count[p]) continue;
        bgr[3 * p + 0] = (unsigned char)(int)rgb[3 * p + 2];
        bgr[3 * p + 1] = (unsigned char)(int)rgb[3 * p + 1];
        bgr[3 * p + 2] = (unsigned char)(int)rgb[3 * p + 0];
    }
}

/* bitmap_image::save_image, bitmap_image.hpp:431-478 (headers :1302-1358). Returns bytes written, 0 on failure. */
size_t orc_write_bmp(const char *path, int W, int H, const unsigned char *bgr) {
    FILE *f = fopen(path, "wb");
    if (!f) return 0;
    uint32_t size_image = (((uint32_t)W * 3u + 3u) & 0x0000FFFCu) * (uint32_t)H;
    unsigned char h[54]; memset(h, 0, sizeof h);
    uint32_t fsize = 14u + 40u + size_image;
    h[0] = 0x42; h[1] = 0x4D;                      /* 19778 */
    memcpy(h + 2, &fsize, 4);
    uint32_t off = 54; memcpy(h + 10, &off, 4);
    uint32_t v = 40; memcpy(h + 14, &v, 4);
    v = (uint32_t)W; memcpy(h + 18, &v, 4);
    v = (uint32_t)H; memcpy(h + 22, &v, 4);
    uint16_t s = 1; memcpy(h + 26, &s, 2);
    s = 24; memcpy(h + 28, &s, 2);
    memcpy(h + 34, &size_image, 4);
    size_t n = fwrite(h, 1, 54, f);
    const unsigned pad = (4 - ((3 * (unsigned)W) % 4)) % 4;
    const unsigned char zeros[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < H; ++i) {
        n += fwrite(bgr + (size_t)(H - i - 1) * W * 3, 1, (size_t)W * 3, f);
        n += fwrite(zeros, 1, pad, f);
    }
    fclose(f);
    return n;
}

/* ------------------------------------------------------------------ */
/* Small probes for the unit tests                                      */
/* ------------------------------------------------------------------ */
void orc_probe_minstd(uint32_t seed, int n, uint32_t *raw, float *unit, double *jitter) {
    minstd0 a, b, c;
    minstd0_seed(&a, seed); minstd0_seed(&b, seed); minstd0_seed(&c, seed);
    for (int i = 0; i < n; ++i) { raw[i] = minstd0_next(&a); unit[i] = libstd_canonical_float(&b); jitter[i] = libstd_jitter_double(&c); }
}
void orc_probe_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }
float orc_probe_unit_float(uint32_t w) { return u32_to_unit_float(w); }
double orc_probe_jitter(uint32_t w) { return u32_to_jitter(w); }
void orc_probe_acos_atan2(const float *v, const float *y, const float *x, int n, int policy, float *ac, float *at) {
    for (int i = 0; i < n; ++i) {
        if (policy == ORC_TRIG_LIBM) { ac[i] = acosf(v[i]); at[i] = atan2f(y[i], x[i]); }
        else { ac[i] = portable_acosf(v[i]); at[i] = portable_atan2f(y[i], x[i]); }
    }
}
void orc_probe_sincos(const float *a, int n, int policy, float *s, float *c) {
    for (int i = 0; i < n; ++i) {
        if (policy == ORC_TRIG_LIBM) { s[i] = sinf(a[i]); c[i] = cosf(a[i]); }
        else portable_sincosf(a[i], &s[i], &c[i]);
    }
}
/* One Scene::TraceRay call per caller-supplied ray with caller-supplied random words (the counter policy's conversion to (0, 1)
 * floats; trig_policy as in orc_params): rays in and out (o, d [n][3], color [n][3], depth [n]), what the call added to the
 * pixel's accumulators (contrib [n][3]) and whether it added anything (contributed [n]).  For tests/test_double_entry.py. */
void orc_probe_segments(const orc_scene *sc, int n, float eps, int mrr, int trig_policy, float *o, float *d, float *color, int *depth,
                        const uint32_t *rnd, float *contrib, int *contributed) {
    for (int i = 0; i < n; ++i) {
        orc_params pr;
        memset(&pr, 0, sizeof pr);
        pr.eps = eps; pr.max_ray_reflections = mrr; pr.rng_policy = ORC_RNG_COUNTER; pr.trig_policy = trig_policy;
        orc_stats st;
        memset(&st, 0, sizeof st);
        float sum[3] = {0, 0, 0}, sum2[3] = {0, 0, 0};
        int count = 0;
        orc_ctx cx;
        memset(&cx, 0, sizeof cx);
        cx.sc = sc; cx.pr = &pr; cx.sum = sum; cx.sum2 = sum2; cx.count = &count; cx.st = &st;
        for (int k = 0; k < 3; ++k) cx.rnd[k] = rnd[3 * (size_t)i + k];
        orc_ray r;
        for (int k = 0; k < 3; ++k) { r.o[k] = o[3 * (size_t)i + k]; r.d[k] = d[3 * (size_t)i + k]; r.color[k] = color[3 * (size_t)i + k]; }
        r.depth = depth[i];
        trace_segment(&cx, &r);
        for (int k = 0; k < 3; ++k) { o[3 * (size_t)i + k] = r.o[k]; d[3 * (size_t)i + k] = r.d[k]; color[3 * (size_t)i + k] = r.color[k]; contrib[3 * (size_t)i + k] = sum[k]; }
        depth[i] = r.depth;
        contributed[i] = count;
    }
}
/* The primary ray's direction for pixel (x, y) and the two jitter draws (main.cpp:126-129, ray.h:21-25), and the adaptive-sampling
 * answer for a pixel's accumulators before pass `pass` (main.cpp:118-125; 1 = the pixel sits the pass out).  For tests/test_double_entry.py. */
void orc_probe_primary(int n, const int *x, const int *y, const double *jx, const double *jy, int W, int H, float *d) {
    for (int i = 0; i < n; ++i) {
        orc_ray r;
        primary_ray(&r, x[i], y[i], jx[i], jy[i], W, H);
        d[3 * (size_t)i] = r.d[0]; d[3 * (size_t)i + 1] = r.d[1]; d[3 * (size_t)i + 2] = r.d[2];
    }
}
void orc_probe_adaptive_skip(int n, const int *pass, const float *c, const float *c2, const int *count, float error, int *skip) {
    for (int i = 0; i < n; ++i) skip[i] = adaptive_skip(pass[i], c + 3 * (size_t)i, c2 + 3 * (size_t)i, count[i], error);
}
/* Per-stage test of one triangle (T2-style vectors). */
int orc_probe_intersect(const orc_scene *s, int tri, const float *o, const float *d, float eps, float best_in, float *best_out) {
    float dist = best_in;
    int st = tri_intersect(&s->tri[tri], o, d, eps, &dist);
    *best_out = dist;
    return st;
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
