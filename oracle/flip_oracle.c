/*
 * oracle/flip_oracle.c -- TEST INFRASTRUCTURE ONLY (see flip_oracle.h).
 *
 * Plain C99 restatement of the reference's FLIP substep.  Each function cites the
 * reference file:line whose arithmetic it follows, including the float/double promotions
 * (SURVEY.md Appendix A), so results are bit-comparable with the compiled reference.
 * Build with -ffp-contract=off (oracle/Makefile).
 */
#define _POSIX_C_SOURCE 200809L
#include "flip_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#define IDX(i, j, k, w, h) ((size_t)(i) + (size_t)(w) * ((size_t)(j) + (size_t)(h) * (size_t)(k)))

/* Optional all-cores build (make -C oracle oracle_omp -> libfliporacle_omp.so, -fopenmp): the loops of the reference's algorithm that
 * are data-parallel as written (SpMV, dot products, axpys, max norms, volume fractions, particle kernels) run across the host's cores;
 * what the reference's algorithm makes sequential stays sequential (the two triangular solves of MIC(0), the incomplete
 * factorisation, the scatters in particle order, the layered extrapolation).  Dot products then sum in a different order, so this
 * build is NOT the bit-pinned oracle (tests/ use the plain one); it exists for bench.py's all-cores CPU baseline. */
#ifdef _OPENMP
#include <omp.h>
#define OMP_FOR _Pragma("omp parallel for schedule(static)")
#define OMP_FOR_SUM(v) _Pragma(STRINGIFY_(omp parallel for schedule(static) reduction(+ : v)))
#define OMP_FOR_MAX(v) _Pragma(STRINGIFY_(omp parallel for schedule(static) reduction(max : v)))
#define STRINGIFY_(x) #x
int oracle_omp_threads(void) { return omp_get_max_threads(); }
#else
#define OMP_FOR
#define OMP_FOR_SUM(v)
#define OMP_FOR_MAX(v)
int oracle_omp_threads(void) { return 1; }
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------- grid helpers (reference grid3d.h) ---------------- */

/* Grid3d::positionToGridIndex(vec3, double dx) grid3d.h:60-65: float coord promoted, times 1/dx */
static int pos_to_index(float p, double dx) {
    double invdx = 1.0 / dx;
    return (int)floor((double)p * invdx);
}
/* Grid3d::GridIndexToPosition grid3d.h:81-83 */
static float index_to_pos(int i, double dx) { return (float)(i * dx); }
/* Grid3d::GridIndexToCellCenter grid3d.h:105-108 */
static float index_to_center(int i, double dx) {
    double hw = 0.5 * dx;
    return (float)(i * dx + hw);
}
static int in_range(int i, int j, int k, int w, int h, int d) {
    return i >= 0 && j >= 0 && k >= 0 && i < w && j < h && k < d;
}
static int imax2(int a, int b) { return a > b ? a : b; }
static int imin2(int a, int b) { return a < b ? a : b; }

/* Grid3d::isFaceBorderingValueU/V/W (grid3d.h:496-530) on a "phi < 0" predicate */
static int face_borders_fluid(int dir, int i, int j, int k, int I, int J, int K, const float *phi) {
    int n = dir == 0 ? I : (dir == 1 ? J : K);
    int c = dir == 0 ? i : (dir == 1 ? j : k);
    int di = dir == 0, dj = dir == 1, dk = dir == 2;
    if (c == n) return phi[IDX(i - di, j - dj, k - dk, I, J)] < 0.0f;
    if (c > 0) return phi[IDX(i, j, k, I, J)] < 0.0f || phi[IDX(i - di, j - dj, k - dk, I, J)] < 0.0f;
    return phi[IDX(i, j, k, I, J)] < 0.0f;
}

/* ---------------- level-set fractions (reference levelsetutils.{h,cpp}) ---------------- */

float oracle_fraction_inside2(float l, float r) { /* levelsetutils.cpp:15-27 */
    if (l < 0 && r < 0) return 1;
    if (l < 0 && r >= 0) return l / (l - r);
    if (l >= 0 && r < 0) return r / (r - l);
    return 0;
}

static void rotate4(float *a) {
    float t = a[0];
    a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t;
}

float oracle_fraction_inside4(float bl, float br, float tl, float tr) { /* levelsetutils.cpp:38-119 */
    int n = (bl < 0) + (tl < 0) + (br < 0) + (tr < 0);
    float l[4] = {bl, br, tr, tl};
    if (n == 4) return 1;
    if (n == 3) {
        while (l[0] < 0) rotate4(l);
        float s0 = 1 - oracle_fraction_inside2(l[0], l[3]);
        float s1 = 1 - oracle_fraction_inside2(l[0], l[1]);
        return 1.0f - 0.5f * s0 * s1;
    }
    if (n == 2) {
        while (l[0] >= 0 || !(l[1] < 0 || l[2] < 0)) rotate4(l);
        if (l[1] < 0) {
            float sl = oracle_fraction_inside2(l[0], l[3]);
            float sr = oracle_fraction_inside2(l[1], l[2]);
            return 0.5f * (sl + sr);
        }
        float mid = 0.25f * (l[0] + l[1] + l[2] + l[3]);
        if (mid < 0) {
            float area = 0;
            float s1 = 1 - oracle_fraction_inside2(l[0], l[3]);
            float s3 = 1 - oracle_fraction_inside2(l[2], l[3]);
            area += 0.5f * s1 * s3;
            float s2 = 1 - oracle_fraction_inside2(l[2], l[1]);
            float s0 = 1 - oracle_fraction_inside2(l[0], l[1]);
            area += 0.5f * s0 * s2;
            return 1.0f - area;
        } else {
            float area = 0;
            float s0 = oracle_fraction_inside2(l[0], l[1]);
            float s1 = oracle_fraction_inside2(l[0], l[3]);
            area += 0.5f * s0 * s1;
            float s2 = oracle_fraction_inside2(l[2], l[1]);
            float s3 = oracle_fraction_inside2(l[2], l[3]);
            area += 0.5f * s2 * s3;
            return area;
        }
    }
    if (n == 1) {
        while (l[0] >= 0) rotate4(l);
        float s0 = oracle_fraction_inside2(l[0], l[3]);
        float s1 = oracle_fraction_inside2(l[0], l[1]);
        return 0.5f * s0 * s1;
    }
    return 0;
}

static float tet_frac(float a, float b, float c, float d) { /* levelsetutils.h:46-49 */
    return a * a * a / ((a - b) * (a - c) * (a - d));
}
static float prism_frac(float p0, float p1, float p2, float p3) { /* levelsetutils.h:53-60 */
    float a = p0 / (p0 - p2);
    float b = p0 / (p0 - p3);
    float c = p1 / (p1 - p3);
    float d = p1 / (p1 - p2);
    return a * b * (1 - d) + b * (1 - c) * d + c * d;
}
#define CSWAP(x, y) do { if ((x) > (y)) { float t_ = (x); (x) = (y); (y) = t_; } } while (0)
static float tet_volume_fraction(float p0, float p1, float p2, float p3) { /* levelsetutils.cpp:189-202 */
    CSWAP(p0, p1); CSWAP(p2, p3); CSWAP(p0, p2); CSWAP(p1, p3); CSWAP(p1, p2);
    if (p3 <= 0) return 1;
    if (p2 <= 0) return 1 - tet_frac(p3, p2, p1, p0);
    if (p1 <= 0) return prism_frac(p0, p1, p2, p3);
    if (p0 <= 0) return tet_frac(p0, p1, p2, p3);
    return 0;
}
float oracle_volume_fraction8(const float p[8]) { /* levelsetutils.cpp:219-235 */
    float p000 = p[0], p100 = p[1], p010 = p[2], p110 = p[3], p001 = p[4], p101 = p[5], p011 = p[6], p111 = p[7];
    return (tet_volume_fraction(p000, p001, p101, p011) +
            tet_volume_fraction(p000, p101, p100, p110) +
            tet_volume_fraction(p000, p010, p011, p110) +
            tet_volume_fraction(p101, p011, p111, p110) +
            2 * tet_volume_fraction(p000, p011, p101, p110) +
            tet_volume_fraction(p100, p101, p001, p111) +
            tet_volume_fraction(p100, p001, p000, p010) +
            tet_volume_fraction(p100, p110, p111, p010) +
            tet_volume_fraction(p001, p111, p011, p010) +
            2 * tet_volume_fraction(p100, p111, p001, p010)) / 12.0f;
}

/* ---------------- scalar-field trilinear (reference interpolation.cpp:68-184) ---------------- */

/* Interpolation::trilinearInterpolate(vec3 p, double dx, Array3d<float>&): position in float,
 * weights in double, out-of-range corners contribute 0. */
static double trilerp_field(float px, float py, float pz, double dx, const float *g, int w, int h, int d) {
    int gi = pos_to_index(px, dx), gj = pos_to_index(py, dx), gk = pos_to_index(pz, dx);
    float gx = index_to_pos(gi, dx), gy = index_to_pos(gj, dx), gz = index_to_pos(gk, dx);
    double inv_dx = 1.0 / dx;
    double ix = (px - gx) * inv_dx;
    double iy = (py - gy) * inv_dx;
    double iz = (pz - gz) * inv_dx;
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (in_range(gi, gj, gk, w, h, d)) p[0] = g[IDX(gi, gj, gk, w, h)];
    if (in_range(gi + 1, gj, gk, w, h, d)) p[1] = g[IDX(gi + 1, gj, gk, w, h)];
    if (in_range(gi, gj + 1, gk, w, h, d)) p[2] = g[IDX(gi, gj + 1, gk, w, h)];
    if (in_range(gi, gj, gk + 1, w, h, d)) p[3] = g[IDX(gi, gj, gk + 1, w, h)];
    if (in_range(gi + 1, gj, gk + 1, w, h, d)) p[4] = g[IDX(gi + 1, gj, gk + 1, w, h)];
    if (in_range(gi, gj + 1, gk + 1, w, h, d)) p[5] = g[IDX(gi, gj + 1, gk + 1, w, h)];
    if (in_range(gi + 1, gj + 1, gk, w, h, d)) p[6] = g[IDX(gi + 1, gj + 1, gk, w, h)];
    if (in_range(gi + 1, gj + 1, gk + 1, w, h, d)) p[7] = g[IDX(gi + 1, gj + 1, gk + 1, w, h)];
    /* interpolation.cpp:57-66 */
    return p[0] * (1 - ix) * (1 - iy) * (1 - iz) + p[1] * ix * (1 - iy) * (1 - iz) +
           p[2] * (1 - ix) * iy * (1 - iz) + p[3] * (1 - ix) * (1 - iy) * iz + p[4] * ix * (1 - iy) * iz +
           p[5] * (1 - ix) * iy * iz + p[6] * ix * iy * (1 - iz) + p[7] * ix * iy * iz;
}

static double bilerp(double v00, double v10, double v01, double v11, double ix, double iy) { /* interpolation.cpp:114-120 */
    double l1 = (1 - ix) * v00 + ix * v10;
    double l2 = (1 - ix) * v01 + ix * v11;
    return (1 - iy) * l1 + iy * l2;
}

/* Interpolation::trilinearInterpolateGradient (interpolation.cpp:122-184), un-normalised by dx */
static void trilerp_gradient(float px, float py, float pz, double dx, const float *g, int w, int h, int d,
                             float grad[3]) {
    int gi = pos_to_index(px, dx), gj = pos_to_index(py, dx), gk = pos_to_index(pz, dx);
    float gx = index_to_pos(gi, dx), gy = index_to_pos(gj, dx), gz = index_to_pos(gk, dx);
    double inv_dx = 1.0 / dx;
    double ix = (px - gx) * inv_dx;
    double iy = (py - gy) * inv_dx;
    double iz = (pz - gz) * inv_dx;
    float v000 = 0, v001 = 0, v010 = 0, v011 = 0, v100 = 0, v101 = 0, v110 = 0, v111 = 0;
    if (in_range(gi, gj, gk, w, h, d)) v000 = g[IDX(gi, gj, gk, w, h)];
    if (in_range(gi + 1, gj, gk, w, h, d)) v100 = g[IDX(gi + 1, gj, gk, w, h)];
    if (in_range(gi, gj + 1, gk, w, h, d)) v010 = g[IDX(gi, gj + 1, gk, w, h)];
    if (in_range(gi, gj, gk + 1, w, h, d)) v001 = g[IDX(gi, gj, gk + 1, w, h)];
    if (in_range(gi + 1, gj, gk + 1, w, h, d)) v101 = g[IDX(gi + 1, gj, gk + 1, w, h)];
    if (in_range(gi, gj + 1, gk + 1, w, h, d)) v011 = g[IDX(gi, gj + 1, gk + 1, w, h)];
    if (in_range(gi + 1, gj + 1, gk, w, h, d)) v110 = g[IDX(gi + 1, gj + 1, gk, w, h)];
    if (in_range(gi + 1, gj + 1, gk + 1, w, h, d)) v111 = g[IDX(gi + 1, gj + 1, gk + 1, w, h)];
    float ddx00 = v100 - v000, ddx10 = v110 - v010, ddx01 = v101 - v001, ddx11 = v111 - v011;
    grad[0] = (float)bilerp(ddx00, ddx10, ddx01, ddx11, iy, iz);
    float ddy00 = v010 - v000, ddy10 = v110 - v100, ddy01 = v011 - v001, ddy11 = v111 - v101;
    grad[1] = (float)bilerp(ddy00, ddy10, ddy01, ddy11, ix, iz);
    float ddz00 = v001 - v000, ddz10 = v101 - v100, ddz01 = v011 - v010, ddz11 = v111 - v110;
    grad[2] = (float)bilerp(ddz00, ddz10, ddz01, ddz11, ix, iy);
}

/* ParticleLevelSet::trilinearInterpolate (particlelevelset.cpp:88-92): sample at pos - (h,h,h) */
static float liquid_phi_at(float px, float py, float pz, double dx, const float *phi, int I, int J, int K) {
    float hdx = (float)(0.5 * dx);
    return (float)trilerp_field(px - hdx, py - hdx, pz - hdx, dx, phi, I, J, K);
}

/* MeshLevelSet::getDistanceAtCellCenter (meshlevelset.cpp:66-76) */
static float solid_center_phi(int i, int j, int k, const float *s, int I, int J) {
    int w = I + 1, h = J + 1;
    return 0.125f * (s[IDX(i, j, k, w, h)] + s[IDX(i + 1, j, k, w, h)] + s[IDX(i, j + 1, k, w, h)] +
                     s[IDX(i + 1, j + 1, k, w, h)] + s[IDX(i, j, k + 1, w, h)] + s[IDX(i + 1, j, k + 1, w, h)] +
                     s[IDX(i, j + 1, k + 1, w, h)] + s[IDX(i + 1, j + 1, k + 1, w, h)]);
}

/* ---------------- K1 + K2: particle level set ---------------- */

void oracle_particle_sdf(int I, int J, int K, float dxf, const float *aos6, size_t n, const float *solid,
                         float *phi) {
    double dx = (double)dxf;
    /* fluidsimulation.cpp:36 */
    float radius_f = (float)(dxf * 1.01 * sqrt(3.0) / 2.0);
    double radius = (double)radius_f;
    float maxd = 3.0f * (float)dx; /* particlelevelset.cpp:94-96 */
    size_t ncell = (size_t)I * J * K;
    for (size_t c = 0; c < ncell; c++) phi[c] = maxd;

    /* particlelevelset.cpp:98-125 */
    for (size_t p = 0; p < n; p++) {
        float px = aos6[6 * p], py = aos6[6 * p + 1], pz = aos6[6 * p + 2];
        int gi = pos_to_index(px, dx), gj = pos_to_index(py, dx), gk = pos_to_index(pz, dx);
        int i0 = imax2(0, gi - 1), j0 = imax2(0, gj - 1), k0 = imax2(0, gk - 1);
        int i1 = imin2(gi + 1, I - 1), j1 = imin2(gj + 1, J - 1), k1 = imin2(gk + 1, K - 1);
        for (int k = k0; k <= k1; k++)
            for (int j = j0; j <= j1; j++)
                for (int i = i0; i <= i1; i++) {
                    float vx = index_to_center(i, dx) - px;
                    float vy = index_to_center(j, dx) - py;
                    float vz = index_to_center(k, dx) - pz;
                    float dist = sqrtf(vx * vx + vy * vy + vz * vz) - (float)radius;
                    size_t c = IDX(i, j, k, I, J);
                    if (dist < phi[c]) phi[c] = dist;
                }
    }

    /* particlelevelset.cpp:127-139 */
    for (int k = 0; k < K; k++)
        for (int j = 0; j < J; j++)
            for (int i = 0; i < I; i++) {
                size_t c = IDX(i, j, k, I, J);
                if (phi[c] < 0.5 * dx) {
                    if (solid_center_phi(i, j, k, solid, I, J) < 0) phi[c] = -0.5f * (float)dx;
                }
            }
}

/* ---------------- K3 + K4: particle -> grid ---------------- */

void oracle_p2g_component(int I, int J, int K, float dxf, const float *aos6, size_t n, int dir, float *field,
                          uint8_t *isset) {
    int w = I + (dir == 0), h = J + (dir == 1), d = K + (dir == 2);
    size_t nf = (size_t)w * h * d;
    float *weights = (float *)calloc(nf, sizeof(float));
    memset(field, 0, nf * sizeof(float));
    memset(isset, 0, nf);
    double dx = (double)dxf;
    float hdx = (float)(0.5 * dxf); /* fluidsimulation.cpp:370 */
    float ox = dir == 0 ? 0.0f : hdx, oy = dir == 1 ? 0.0f : hdx, oz = dir == 2 ? 0.0f : hdx;
    /* fluidsimulation.cpp:384-388 */
    float r = dxf;
    float rsq = r * r;
    float coef1 = (4.0f / 9.0f) * (1.0f / (r * r * r * r * r * r));
    float coef2 = (17.0f / 9.0f) * (1.0f / (r * r * r * r));
    float coef3 = (22.0f / 9.0f) * (1.0f / (r * r));

    for (size_t p = 0; p < n; p++) { /* fluidsimulation.cpp:391-420 */
        float px = aos6[6 * p] - ox, py = aos6[6 * p + 1] - oy, pz = aos6[6 * p + 2] - oz;
        float vel = aos6[6 * p + 3 + dir];
        int gi = pos_to_index(px, dx), gj = pos_to_index(py, dx), gk = pos_to_index(pz, dx);
        int i0 = imax2(gi - 1, 0), j0 = imax2(gj - 1, 0), k0 = imax2(gk - 1, 0);
        int i1 = imin2(gi + 1, w - 1), j1 = imin2(gj + 1, h - 1), k1 = imin2(gk + 1, d - 1);
        for (int k = k0; k <= k1; k++)
            for (int j = j0; j <= j1; j++)
                for (int i = i0; i <= i1; i++) {
                    float vx = index_to_pos(i, dx) - px;
                    float vy = index_to_pos(j, dx) - py;
                    float vz = index_to_pos(k, dx) - pz;
                    float distsq = vx * vx + vy * vy + vz * vz;
                    if (distsq < rsq) {
                        float weight = 1.0f - coef1 * distsq * distsq * distsq + coef2 * distsq * distsq -
                                       coef3 * distsq;
                        size_t f = IDX(i, j, k, w, h);
                        field[f] += weight * vel;
                        weights[f] += weight;
                    }
                }
    }
    double eps = 1e-9; /* fluidsimulation.cpp:423-437 */
    for (size_t f = 0; f < nf; f++) {
        float value = field[f], weight = weights[f];
        if (weight < eps) continue;
        field[f] = value / weight;
        isset[f] = 1;
    }
    free(weights);
}

void oracle_p2g(int I, int J, int K, float dx, const float *aos6, size_t n, const float *phi, float *U,
                float *V, float *W, uint8_t *validU, uint8_t *validV, uint8_t *validW) {
    float *out[3] = {U, V, W};
    uint8_t *val[3] = {validU, validV, validW};
    for (int dir = 0; dir < 3; dir++) { /* fluidsimulation.cpp:440-498 */
        int w = I + (dir == 0), h = J + (dir == 1), d = K + (dir == 2);
        size_t nf = (size_t)w * h * d;
        float *field = (float *)malloc(nf * sizeof(float));
        uint8_t *isset = (uint8_t *)malloc(nf);
        oracle_p2g_component(I, J, K, dx, aos6, n, dir, field, isset);
        memset(out[dir], 0, nf * sizeof(float));
        memset(val[dir], 0, nf);
        for (int k = 0; k < d; k++)
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    size_t f = IDX(i, j, k, w, h);
                    if (face_borders_fluid(dir, i, j, k, I, J, K, phi) && isset[f]) {
                        out[dir][f] = field[f];
                        val[dir][f] = 1;
                    }
                }
        free(field);
        free(isset);
    }
}

/* ---------------- K5: extrapolation ---------------- */

void oracle_extrapolate_grid(int w, int h, int d, float *grid, const uint8_t *valid, int layers) {
    /* macvelocityfield.cpp:580-687.  Status: 0 unknown, 1 waiting, 2 known, 3 frozen (unknown on the border). */
    size_t n = (size_t)w * h * d;
    uint8_t *st = (uint8_t *)malloc(n);
    for (int k = 0; k < d; k++)
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                size_t c = IDX(i, j, k, w, h);
                st[c] = valid[c] ? 2 : 0;
                int border = i == 0 || j == 0 || k == 0 || i == w - 1 || j == h - 1 || k == d - 1;
                if (st[c] == 0 && border) st[c] = 3;
            }
    size_t *list = (size_t *)malloc(n * sizeof(size_t));
    const long off[6] = {-1, 1, -(long)w, (long)w, -(long)w * h, (long)w * h};
    for (int layer = 0; layer < layers; layer++) {
        size_t cnt = 0;
        for (int k = 1; k < d - 1; k++)
            for (int j = 1; j < h - 1; j++)
                for (int i = 1; i < w - 1; i++) {
                    size_t c = IDX(i, j, k, w, h);
                    if (st[c] != 2) continue;
                    for (int q = 0; q < 6; q++) {
                        size_t nb = (size_t)((long)c + off[q]);
                        if (st[nb] == 0) {
                            st[nb] = 1;
                            list[cnt++] = nb;
                        }
                    }
                }
        for (size_t t = 0; t < cnt; t++) {
            size_t c = list[t];
            float sum = 0;
            int count = 0;
            for (int q = 0; q < 6; q++) { /* order -i,+i,-j,+j,-k,+k (macvelocityfield.cpp:671-676) */
                size_t nb = (size_t)((long)c + off[q]);
                if (st[nb] == 2) { sum += grid[nb]; count++; }
            }
            grid[c] = sum / (float)count;
        }
        for (size_t t = 0; t < cnt; t++) st[list[t]] = 2;
    }
    free(list);
    free(st);
}

/* ---------------- K6, K16, K10 ---------------- */

void oracle_body_force(int I, int J, int K, const float *phi, float *U, float *V, float *W, float gx, float gy,
                       float gz, float dt) {
    float *out[3] = {U, V, W};
    float g[3] = {gx, gy, gz};
    for (int dir = 0; dir < 3; dir++) { /* fluidsimulation.cpp:283-311 */
        int w = I + (dir == 0), h = J + (dir == 1), d = K + (dir == 2);
        float inc = g[dir] * dt;
        for (int k = 0; k < d; k++)
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++)
                    if (face_borders_fluid(dir, i, j, k, I, J, K, phi)) out[dir][IDX(i, j, k, w, h)] += inc;
    }
}

float oracle_cfl(int I, int J, int K, float dx, const float *U, const float *V, const float *W, float cfl_number) {
    float maxvel = 0; /* fluidsimulation.cpp:241-269 */
    size_t nu = (size_t)(I + 1) * J * K, nv = (size_t)I * (J + 1) * K, nw = (size_t)I * J * (K + 1);
    for (size_t c = 0; c < nu; c++) maxvel = fmaxf(maxvel, fabsf(U[c]));
    for (size_t c = 0; c < nv; c++) maxvel = fmaxf(maxvel, fabsf(V[c]));
    for (size_t c = 0; c < nw; c++) maxvel = fmaxf(maxvel, fabsf(W[c]));
    return (float)((cfl_number * dx) / maxvel);
}

static float clampf(float v, float lo, float hi) { return fmaxf(lo, fminf(v, hi)); }

void oracle_compute_weights(int I, int J, int K, const float *s, float *wU, float *wV, float *wW) {
    int nw = I + 1, nh = J + 1; /* solid node grid */
    /* fluidsimulation.cpp:549-582 with meshlevelset.cpp:92-126 argument orders */
    for (int k = 0; k < K; k++)
        for (int j = 0; j < J; j++)
            for (int i = 0; i < I + 1; i++) {
                float f = oracle_fraction_inside4(s[IDX(i, j, k, nw, nh)], s[IDX(i, j + 1, k, nw, nh)],
                                                  s[IDX(i, j, k + 1, nw, nh)], s[IDX(i, j + 1, k + 1, nw, nh)]);
                wU[IDX(i, j, k, I + 1, J)] = clampf(1.0f - f, 0.0f, 1.0f);
            }
    for (int k = 0; k < K; k++)
        for (int j = 0; j < J + 1; j++)
            for (int i = 0; i < I; i++) {
                float f = oracle_fraction_inside4(s[IDX(i, j, k, nw, nh)], s[IDX(i, j, k + 1, nw, nh)],
                                                  s[IDX(i + 1, j, k, nw, nh)], s[IDX(i + 1, j, k + 1, nw, nh)]);
                wV[IDX(i, j, k, I, J + 1)] = clampf(1.0f - f, 0.0f, 1.0f);
            }
    for (int k = 0; k < K + 1; k++)
        for (int j = 0; j < J; j++)
            for (int i = 0; i < I; i++) {
                float f = oracle_fraction_inside4(s[IDX(i, j, k, nw, nh)], s[IDX(i, j + 1, k, nw, nh)],
                                                  s[IDX(i + 1, j, k, nw, nh)], s[IDX(i + 1, j + 1, k, nw, nh)]);
                wW[IDX(i, j, k, I, J)] = clampf(1.0f - f, 0.0f, 1.0f);
            }
}

/* ---------------- K11 + K12: pressure solve ---------------- */

typedef struct { float diag, plusi, plusj, plusk; } pcell; /* pressuresolver.h:103-110 */

typedef struct {
    int n;
    int *ci, *cj, *ck; /* pressure cells in k-j-i order */
    int *key;          /* dense map, -1 = not a pressure cell */
    int I, J, K;
    pcell *A;
    double *precon;
} psys;

static int pkey(const psys *s, int i, int j, int k) { return s->key[IDX(i, j, k, s->I, s->J)]; }

static void p_apply_precon(const psys *s, const double *r, double *z, double *q) { /* pressuresolver.cpp:381-462 */
    for (int idx = 0; idx < s->n; idx++) {
        int i = s->ci[idx], j = s->cj[idx], k = s->ck[idx];
        int im1 = pkey(s, i - 1, j, k), jm1 = pkey(s, i, j - 1, k), km1 = pkey(s, i, j, k - 1);
        double pi = 0, ci_ = 0, qi = 0, pj = 0, cj_ = 0, qj = 0, pk = 0, ck_ = 0, qk = 0;
        if (im1 != -1) { pi = (double)s->A[im1].plusi; ci_ = s->precon[im1]; qi = q[im1]; }
        if (jm1 != -1) { pj = (double)s->A[jm1].plusj; cj_ = s->precon[jm1]; qj = q[jm1]; }
        if (km1 != -1) { pk = (double)s->A[km1].plusk; ck_ = s->precon[km1]; qk = q[km1]; }
        double t = r[idx] - pi * ci_ * qi - pj * cj_ * qj - pk * ck_ * qk;
        t = t * s->precon[idx];
        q[idx] = t;
    }
    for (int idx = s->n - 1; idx >= 0; idx--) {
        int i = s->ci[idx], j = s->cj[idx], k = s->ck[idx];
        int ip1 = pkey(s, i + 1, j, k), jp1 = pkey(s, i, j + 1, k), kp1 = pkey(s, i, j, k + 1);
        double zi = ip1 != -1 ? z[ip1] : 0.0, zj = jp1 != -1 ? z[jp1] : 0.0, zk = kp1 != -1 ? z[kp1] : 0.0;
        double plusi = (double)s->A[idx].plusi, plusj = (double)s->A[idx].plusj, plusk = (double)s->A[idx].plusk;
        double pv = s->precon[idx];
        double t = q[idx] - plusi * pv * zi - plusj * pv * zj - plusk * pv * zk;
        t = t * pv;
        z[idx] = t;
    }
}

static void p_apply_matrix(const psys *s, const double *x, double *y) { /* pressuresolver.cpp:464-499 */
    OMP_FOR
    for (int idx = 0; idx < s->n; idx++) {
        int i = s->ci[idx], j = s->cj[idx], k = s->ck[idx];
        double val = 0.0;
        int v;
        v = pkey(s, i - 1, j, k); if (v != -1) val += x[v] * s->A[v].plusi;
        v = pkey(s, i + 1, j, k); if (v != -1) val += x[v] * s->A[idx].plusi;
        v = pkey(s, i, j - 1, k); if (v != -1) val += x[v] * s->A[v].plusj;
        v = pkey(s, i, j + 1, k); if (v != -1) val += x[v] * s->A[idx].plusj;
        v = pkey(s, i, j, k - 1); if (v != -1) val += x[v] * s->A[v].plusk;
        v = pkey(s, i, j, k + 1); if (v != -1) val += x[v] * s->A[idx].plusk;
        val += x[idx] * s->A[idx].diag;
        y[idx] = val;
    }
}

static double vdot(const double *a, const double *b, int n) {
    double s = 0.0;
    OMP_FOR_SUM(s)
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}
static double vabsmax_neginf(const double *a, int n) { /* pressuresolver.cpp:75-84 */
    double m = -INFINITY;
    OMP_FOR_MAX(m)
    for (int i = 0; i < n; i++) if (fabs(a[i]) > m) m = fabs(a[i]);
    return m;
}

void oracle_pressure_solve(int I, int J, int K, float dxf, float dtf, const float *U, const float *V,
                           const float *W, const float *wU, const float *wV, const float *wW, const float *phi,
                           float minfrac, double tol, int maxiter, float *pressure, oracle_solve_info *info) {
    size_t ncell = (size_t)I * J * K;
    memset(pressure, 0, ncell * sizeof(float));
    oracle_solve_info li;
    memset(&li, 0, sizeof(li));
    psys s;
    s.I = I; s.J = J; s.K = K;
    s.key = (int *)malloc(ncell * sizeof(int));
    for (size_t c = 0; c < ncell; c++) s.key[c] = -1;
    int n = 0; /* pressuresolver.cpp:196-225 */
    for (int k = 1; k < K - 1; k++)
        for (int j = 1; j < J - 1; j++)
            for (int i = 1; i < I - 1; i++)
                if (phi[IDX(i, j, k, I, J)] < 0) n++;
    s.n = n;
    s.ci = (int *)malloc((size_t)(n + 1) * sizeof(int));
    s.cj = (int *)malloc((size_t)(n + 1) * sizeof(int));
    s.ck = (int *)malloc((size_t)(n + 1) * sizeof(int));
    n = 0;
    for (int k = 1; k < K - 1; k++)
        for (int j = 1; j < J - 1; j++)
            for (int i = 1; i < I - 1; i++)
                if (phi[IDX(i, j, k, I, J)] < 0) {
                    s.ci[n] = i; s.cj[n] = j; s.ck[n] = k;
                    s.key[IDX(i, j, k, I, J)] = n++;
                }
    li.rows = n;
    double dx = (double)dxf, dt = (double)dtf;
    double *b = (double *)calloc((size_t)n + 1, sizeof(double));
    for (int idx = 0; idx < n; idx++) { /* pressuresolver.cpp:227-246 */
        int i = s.ci[idx], j = s.cj[idx], k = s.ck[idx];
        double div = 0.0;
        div -= wU[IDX(i + 1, j, k, I + 1, J)] * U[IDX(i + 1, j, k, I + 1, J)];
        div += wU[IDX(i, j, k, I + 1, J)] * U[IDX(i, j, k, I + 1, J)];
        div -= wV[IDX(i, j + 1, k, I, J + 1)] * V[IDX(i, j + 1, k, I, J + 1)];
        div += wV[IDX(i, j, k, I, J + 1)] * V[IDX(i, j, k, I, J + 1)];
        div -= wW[IDX(i, j, k + 1, I, J)] * W[IDX(i, j, k + 1, I, J)];
        div += wW[IDX(i, j, k, I, J)] * W[IDX(i, j, k, I, J)];
        div /= dx;
        b[idx] = div;
    }
    if (vabsmax_neginf(b, n) < tol) { /* pressuresolver.cpp:173-175 */
        li.status = 3;
        li.residual = n ? vabsmax_neginf(b, n) : 0.0;
        goto done_b;
    }
    s.A = (pcell *)calloc((size_t)n, sizeof(pcell));
    {
        double scale = dt / (dx * dx); /* pressuresolver.cpp:248-322 */
        for (int idx = 0; idx < n; idx++) {
            int i = s.ci[idx], j = s.cj[idx], k = s.ck[idx];
            float pc = phi[IDX(i, j, k, I, J)];
            float term, pn, theta;
            pcell *a = &s.A[idx];
            term = wU[IDX(i + 1, j, k, I + 1, J)] * (float)scale;
            pn = phi[IDX(i + 1, j, k, I, J)];
            if (pn < 0) { a->diag += term; a->plusi -= term; }
            else { theta = fmaxf(oracle_fraction_inside2(pc, pn), minfrac); a->diag += term / theta; }
            term = wU[IDX(i, j, k, I + 1, J)] * (float)scale;
            pn = phi[IDX(i - 1, j, k, I, J)];
            if (pn < 0) { a->diag += term; }
            else { theta = fmaxf(oracle_fraction_inside2(pn, pc), minfrac); a->diag += term / theta; }
            term = wV[IDX(i, j + 1, k, I, J + 1)] * (float)scale;
            pn = phi[IDX(i, j + 1, k, I, J)];
            if (pn < 0) { a->diag += term; a->plusj -= term; }
            else { theta = fmaxf(oracle_fraction_inside2(pc, pn), minfrac); a->diag += term / theta; }
            term = wV[IDX(i, j, k, I, J + 1)] * (float)scale;
            pn = phi[IDX(i, j - 1, k, I, J)];
            if (pn < 0) { a->diag += term; }
            else { theta = fmaxf(oracle_fraction_inside2(pn, pc), minfrac); a->diag += term / theta; }
            term = wW[IDX(i, j, k + 1, I, J)] * (float)scale;
            pn = phi[IDX(i, j, k + 1, I, J)];
            if (pn < 0) { a->diag += term; a->plusk -= term; }
            else { theta = fmaxf(oracle_fraction_inside2(pc, pn), minfrac); a->diag += term / theta; }
            term = wW[IDX(i, j, k, I, J)] * (float)scale;
            pn = phi[IDX(i, j, k - 1, I, J)];
            if (pn < 0) { a->diag += term; }
            else { theta = fmaxf(oracle_fraction_inside2(pn, pc), minfrac); a->diag += term / theta; }
        }
    }
    s.precon = (double *)calloc((size_t)n, sizeof(double));
    {
        double tau = 0.97, sigma = 0.25; /* pressuresolver.cpp:324-379 */
        for (int idx = 0; idx < n; idx++) {
            int i = s.ci[idx], j = s.cj[idx], k = s.ck[idx];
            int im1 = pkey(&s, i - 1, j, k), jm1 = pkey(&s, i, j - 1, k), km1 = pkey(&s, i, j, k - 1);
            double diag = (double)s.A[idx].diag;
            double pi_i = im1 != -1 ? (double)s.A[im1].plusi : 0.0;
            double pi_j = jm1 != -1 ? (double)s.A[jm1].plusi : 0.0;
            double pi_k = km1 != -1 ? (double)s.A[km1].plusi : 0.0;
            double pj_i = im1 != -1 ? (double)s.A[im1].plusj : 0.0;
            double pj_j = jm1 != -1 ? (double)s.A[jm1].plusj : 0.0;
            double pj_k = km1 != -1 ? (double)s.A[km1].plusj : 0.0;
            double pk_i = im1 != -1 ? (double)s.A[im1].plusk : 0.0;
            double pk_j = jm1 != -1 ? (double)s.A[jm1].plusk : 0.0;
            double pk_k = km1 != -1 ? (double)s.A[km1].plusk : 0.0;
            double c_i = im1 != -1 ? s.precon[im1] : 0.0;
            double c_j = jm1 != -1 ? s.precon[jm1] : 0.0;
            double c_k = km1 != -1 ? s.precon[km1] : 0.0;
            double v1 = pi_i * c_i, v2 = pj_j * c_j, v3 = pk_k * c_k;
            double v4 = c_i * c_i, v5 = c_j * c_j, v6 = c_k * c_k;
            double e = diag - v1 * v1 - v2 * v2 - v3 * v3 -
                       tau * (pi_i * (pj_i + pk_i) * v4 + pj_j * (pi_j + pk_j) * v5 + pk_k * (pi_k + pj_k) * v6);
            if (e < sigma * diag) e = diag;
            if (fabs(e) > 10e-9) s.precon[idx] = 1.0 / sqrt(e);
        }
    }
    {
        /* pressuresolver.cpp:521-567 */
        double *x = (double *)calloc((size_t)n, sizeof(double));
        double *r = (double *)malloc((size_t)n * sizeof(double));
        double *z = (double *)calloc((size_t)n, sizeof(double));
        double *sv = (double *)malloc((size_t)n * sizeof(double));
        double *q = (double *)calloc((size_t)n, sizeof(double));
        memcpy(r, b, (size_t)n * sizeof(double));
        p_apply_precon(&s, r, z, q);
        memcpy(sv, z, (size_t)n * sizeof(double));
        double sigma = vdot(z, r, n);
        int it = 0;
        li.status = 1;
        while (it < maxiter) {
            p_apply_matrix(&s, sv, z);
            double alpha = sigma / vdot(z, sv, n);
            OMP_FOR
            for (int c = 0; c < n; c++) x[c] += sv[c] * alpha;
            OMP_FOR
            for (int c = 0; c < n; c++) r[c] += z[c] * (-alpha);
            if (vabsmax_neginf(r, n) < tol) { li.status = 0; break; }
            memset(q, 0, (size_t)n * sizeof(double));
            p_apply_precon(&s, r, z, q);
            double sigma_new = vdot(z, r, n);
            double beta = sigma_new / sigma;
            OMP_FOR
            for (int c = 0; c < n; c++) sv[c] = z[c] * 1.0 + sv[c] * beta;
            sigma = sigma_new;
            it++;
        }
        li.iterations = it;
        li.residual = vabsmax_neginf(r, n);
        for (int idx = 0; idx < n; idx++) /* pressuresolver.cpp:187-191 */
            pressure[IDX(s.ci[idx], s.cj[idx], s.ck[idx], I, J)] = (float)x[idx];
        free(x); free(r); free(z); free(sv); free(q);
    }
    free(s.A);
    free(s.precon);
done_b:
    free(b);
    free(s.key); free(s.ci); free(s.cj); free(s.ck);
    if (info) *info = li;
}

/* ---------------- K13, K14 ---------------- */

void oracle_apply_pressure(int I, int J, int K, float dx, float dt, const float *p, const float *phi,
                           const float *wU, const float *wV, const float *wW, float minfrac, float *U, float *V,
                           float *W, uint8_t *validU, uint8_t *validV, uint8_t *validW) {
    /* fluidsimulation.cpp:598-688 */
    memset(validU, 0, (size_t)(I + 1) * J * K);
    memset(validV, 0, (size_t)I * (J + 1) * K);
    memset(validW, 0, (size_t)I * J * (K + 1));
    for (int k = 0; k < K; k++)
        for (int j = 0; j < J; j++)
            for (int i = 1; i < I; i++) {
                size_t f = IDX(i, j, k, I + 1, J);
                if (wU[f] > 0 && face_borders_fluid(0, i, j, k, I, J, K, phi)) {
                    float p0 = p[IDX(i - 1, j, k, I, J)], p1 = p[IDX(i, j, k, I, J)];
                    float theta = fmaxf(oracle_fraction_inside2(phi[IDX(i - 1, j, k, I, J)], phi[IDX(i, j, k, I, J)]), minfrac);
                    U[f] += (float)(double)(-dt * (p1 - p0) / (dx * theta));
                    validU[f] = 1;
                }
            }
    for (int k = 0; k < K; k++)
        for (int j = 1; j < J; j++)
            for (int i = 0; i < I; i++) {
                size_t f = IDX(i, j, k, I, J + 1);
                if (wV[f] > 0 && face_borders_fluid(1, i, j, k, I, J, K, phi)) {
                    float p0 = p[IDX(i, j - 1, k, I, J)], p1 = p[IDX(i, j, k, I, J)];
                    float theta = fmaxf(oracle_fraction_inside2(phi[IDX(i, j - 1, k, I, J)], phi[IDX(i, j, k, I, J)]), minfrac);
                    V[f] += (float)(double)(-dt * (p1 - p0) / (dx * theta));
                    validV[f] = 1;
                }
            }
    for (int k = 1; k < K; k++)
        for (int j = 0; j < J; j++)
            for (int i = 0; i < I; i++) {
                size_t f = IDX(i, j, k, I, J);
                if (wW[f] > 0 && face_borders_fluid(2, i, j, k, I, J, K, phi)) {
                    float p0 = p[IDX(i, j, k - 1, I, J)], p1 = p[IDX(i, j, k, I, J)];
                    float theta = fmaxf(oracle_fraction_inside2(phi[IDX(i, j, k - 1, I, J)], phi[IDX(i, j, k, I, J)]), minfrac);
                    W[f] += (float)(double)(-dt * (p1 - p0) / (dx * theta));
                    validW[f] = 1;
                }
            }
    size_t nu = (size_t)(I + 1) * J * K, nv = (size_t)I * (J + 1) * K, nw = (size_t)I * J * (K + 1);
    for (size_t f = 0; f < nu; f++) if (!validU[f]) U[f] = 0.0f;
    for (size_t f = 0; f < nv; f++) if (!validV[f]) V[f] = 0.0f;
    for (size_t f = 0; f < nw; f++) if (!validW[f]) W[f] = 0.0f;
}

void oracle_constrain(int I, int J, int K, const float *wU, const float *wV, const float *wW, float *U, float *V,
                      float *W, float *sU, float *sV, float *sW) {
    /* fluidsimulation.cpp:696-729 */
    size_t nu = (size_t)(I + 1) * J * K, nv = (size_t)I * (J + 1) * K, nw = (size_t)I * J * (K + 1);
    for (size_t f = 0; f < nu; f++) if (wU[f] == 0) { U[f] = 0.0f; sU[f] = 0.0f; }
    for (size_t f = 0; f < nv; f++) if (wV[f] == 0) { V[f] = 0.0f; sV[f] = 0.0f; }
    for (size_t f = 0; f < nw; f++) if (wW[f] == 0) { W[f] = 0.0f; sW[f] = 0.0f; }
}

/* ---------------- K7: viscosity volumes ---------------- */

static void estimate_volume_fractions(int I, int J, int K, double dx, const float *phi, float *vol, int w, int h,
                                      int d, float csx, float csy, float csz, const uint8_t *validCells) {
    /* viscositysolver.cpp:180-270, including the first-visitor memoisation of nodal phi */
    int nw = w + 1, nh = h + 1, nd = d + 1;
    size_t nn = (size_t)nw * nh * nd;
    float *nodal = (float *)malloc(nn * sizeof(float));
    uint8_t *isset = (uint8_t *)calloc(nn, 1);
    memset(vol, 0, (size_t)w * h * d * sizeof(float));
    float hdx = 0.5f * (float)dx;
    const int vw = I + 1, vh = J + 1;
    for (int k = 0; k < d; k++)
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                if (!validCells[IDX(i, j, k, vw, vh)]) continue;
                float cx = csx + index_to_center(i, dx);
                float cy = csy + index_to_center(j, dx);
                float cz = csz + index_to_center(k, dx);
                float ph[8]; /* order of evaluation 000,001,010,011,100,101,110,111 */
                const int oi[8] = {0, 0, 0, 0, 1, 1, 1, 1};
                const int oj[8] = {0, 0, 1, 1, 0, 0, 1, 1};
                const int ok[8] = {0, 1, 0, 1, 0, 1, 0, 1};
                for (int q = 0; q < 8; q++) {
                    size_t nidx = IDX(i + oi[q], j + oj[q], k + ok[q], nw, nh);
                    if (!isset[nidx]) {
                        float sx = cx + (oi[q] ? hdx : -hdx);
                        float sy = cy + (oj[q] ? hdx : -hdx);
                        float sz = cz + (ok[q] ? hdx : -hdx);
                        nodal[nidx] = liquid_phi_at(sx, sy, sz, dx, phi, I, J, K);
                        isset[nidx] = 1;
                    }
                    ph[q] = nodal[nidx];
                }
                float p000 = ph[0], p001 = ph[1], p010 = ph[2], p011 = ph[3], p100 = ph[4], p101 = ph[5],
                      p110 = ph[6], p111 = ph[7];
                float v;
                if (p000 < 0 && p001 < 0 && p010 < 0 && p011 < 0 && p100 < 0 && p101 < 0 && p110 < 0 && p111 < 0) {
                    v = 1.0f;
                } else if (p000 >= 0 && p001 >= 0 && p010 >= 0 && p011 >= 0 && p100 >= 0 && p101 >= 0 && p110 >= 0 &&
                           p111 >= 0) {
                    v = 0.0f;
                } else {
                    float a[8] = {p000, p100, p010, p110, p001, p101, p011, p111};
                    v = oracle_volume_fraction8(a);
                }
                vol[IDX(i, j, k, w, h)] = v;
            }
    free(nodal);
    free(isset);
}

static uint8_t *viscosity_valid_cells(int I, int J, int K, const float *phi) {
    /* viscositysolver.cpp:138-168: phi<0 cells on an (I+1,J+1,K+1) mask, dilated twice (6-neighbourhood) */
    int w = I + 1, h = J + 1, d = K + 1;
    size_t n = (size_t)w * h * d;
    uint8_t *valid = (uint8_t *)calloc(n, 1);
    uint8_t *tmp = (uint8_t *)malloc(n);
    for (int k = 0; k < K; k++)
        for (int j = 0; j < J; j++)
            for (int i = 0; i < I; i++)
                if (phi[IDX(i, j, k, I, J)] < 0) valid[IDX(i, j, k, w, h)] = 1;
    for (int layer = 0; layer < 2; layer++) {
        memcpy(tmp, valid, n);
        for (int k = 0; k < d; k++)
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++)
                    if (valid[IDX(i, j, k, w, h)]) {
                        if (i > 0) tmp[IDX(i - 1, j, k, w, h)] = 1;
                        if (i < w - 1) tmp[IDX(i + 1, j, k, w, h)] = 1;
                        if (j > 0) tmp[IDX(i, j - 1, k, w, h)] = 1;
                        if (j < h - 1) tmp[IDX(i, j + 1, k, w, h)] = 1;
                        if (k > 0) tmp[IDX(i, j, k - 1, w, h)] = 1;
                        if (k < d - 1) tmp[IDX(i, j, k + 1, w, h)] = 1;
                    }
        memcpy(valid, tmp, n);
    }
    free(tmp);
    return valid;
}

typedef struct { float *c, *U, *V, *W, *eU, *eV, *eW; } volgrids;

static void compute_volume_grids(int I, int J, int K, double dx, const float *phi, volgrids *g) {
    uint8_t *valid = viscosity_valid_cells(I, J, K, phi);
    float h = (float)(0.5 * (float)dx); /* viscositysolver.cpp:170 (float _dx) */
    estimate_volume_fractions(I, J, K, dx, phi, g->c, I, J, K, h, h, h, valid);
    estimate_volume_fractions(I, J, K, dx, phi, g->U, I + 1, J, K, 0, h, h, valid);
    estimate_volume_fractions(I, J, K, dx, phi, g->V, I, J + 1, K, h, 0, h, valid);
    estimate_volume_fractions(I, J, K, dx, phi, g->W, I, J, K + 1, h, h, 0, valid);
    estimate_volume_fractions(I, J, K, dx, phi, g->eU, I, J + 1, K + 1, h, 0, 0, valid);
    estimate_volume_fractions(I, J, K, dx, phi, g->eV, I + 1, J, K + 1, 0, h, 0, valid);
    estimate_volume_fractions(I, J, K, dx, phi, g->eW, I + 1, J + 1, K, 0, 0, h, valid);
    free(valid);
}

static void alloc_volgrids(int I, int J, int K, volgrids *g) {
    g->c = (float *)malloc((size_t)I * J * K * sizeof(float));
    g->U = (float *)malloc((size_t)(I + 1) * J * K * sizeof(float));
    g->V = (float *)malloc((size_t)I * (J + 1) * K * sizeof(float));
    g->W = (float *)malloc((size_t)I * J * (K + 1) * sizeof(float));
    g->eU = (float *)malloc((size_t)I * (J + 1) * (K + 1) * sizeof(float));
    g->eV = (float *)malloc((size_t)(I + 1) * J * (K + 1) * sizeof(float));
    g->eW = (float *)malloc((size_t)(I + 1) * (J + 1) * K * sizeof(float));
}
static void free_volgrids(volgrids *g) {
    free(g->c); free(g->U); free(g->V); free(g->W); free(g->eU); free(g->eV); free(g->eW);
}

void oracle_viscosity_volumes(int I, int J, int K, float dxf, const float *phi, float *center, float *volU,
                              float *volV, float *volW, float *edgeU, float *edgeV, float *edgeW) {
    volgrids g;
    alloc_volgrids(I, J, K, &g);
    compute_volume_grids(I, J, K, (double)dxf, phi, &g);
    if (center) memcpy(center, g.c, (size_t)I * J * K * sizeof(float));
    if (volU) memcpy(volU, g.U, (size_t)(I + 1) * J * K * sizeof(float));
    if (volV) memcpy(volV, g.V, (size_t)I * (J + 1) * K * sizeof(float));
    if (volW) memcpy(volW, g.W, (size_t)I * J * (K + 1) * sizeof(float));
    if (edgeU) memcpy(edgeU, g.eU, (size_t)I * (J + 1) * (K + 1) * sizeof(float));
    if (edgeV) memcpy(edgeV, g.eV, (size_t)(I + 1) * J * (K + 1) * sizeof(float));
    if (edgeW) memcpy(edgeW, g.eW, (size_t)(I + 1) * (J + 1) * K * sizeof(float));
    free_volgrids(&g);
}

/* ---------------- K8: viscosity system (row lists, <= 15 nnz per row) ---------------- */

#define ROWCAP 16
typedef struct {
    int n;
    int *cnt;
    unsigned *col; /* n x ROWCAP, sorted ascending per row (sparsematrix.h:64-104) */
    double *val;
} rowmat;

static void rm_put(rowmat *m, int i, int j, double v, int add) {
    if (i == -1 || j == -1) return; /* sparsematrix.h:65-67 / 86-88 */
    unsigned *c = m->col + (size_t)i * ROWCAP;
    double *x = m->val + (size_t)i * ROWCAP;
    int n = m->cnt[i];
    int k;
    for (k = 0; k < n; k++) {
        if (c[k] == (unsigned)j) {
            if (add) x[k] += v; else x[k] = v;
            return;
        } else if (c[k] > (unsigned)j) {
            break;
        }
    }
    for (int t = n; t > k; t--) { c[t] = c[t - 1]; x[t] = x[t - 1]; }
    c[k] = (unsigned)j;
    x[k] = v;
    m->cnt[i] = n + 1;
}

enum { FS_AIR = 0, FS_FLUID = 1, FS_SOLID = 2 };

typedef struct {
    int I, J, K;
    uint8_t *sU, *sV, *sW; /* face states */
    int *table;            /* face flat index -> row or -1 */
    int voff, woff;
} vsys;

/* out-of-range faces: the reference would throw (Array3d::operator()); never reached with a closed
 * boundary because those rows are SOLID.  Treat them as neither FLUID nor SOLID. */
static int stU(const vsys *s, int i, int j, int k) { return in_range(i, j, k, s->I + 1, s->J, s->K) ? s->sU[IDX(i, j, k, s->I + 1, s->J)] : FS_AIR; }
static int stV(const vsys *s, int i, int j, int k) { return in_range(i, j, k, s->I, s->J + 1, s->K) ? s->sV[IDX(i, j, k, s->I, s->J + 1)] : FS_AIR; }
static int stW(const vsys *s, int i, int j, int k) { return in_range(i, j, k, s->I, s->J, s->K + 1) ? s->sW[IDX(i, j, k, s->I, s->J)] : FS_AIR; }
static int rowU(const vsys *s, int i, int j, int k) { return s->table[IDX(i, j, k, s->I + 1, s->J)]; }
static int rowV(const vsys *s, int i, int j, int k) { return s->table[s->voff + IDX(i, j, k, s->I, s->J + 1)]; }
static int rowW(const vsys *s, int i, int j, int k) { return s->table[s->woff + IDX(i, j, k, s->I, s->J)]; }

#define VISC(i, j, k) visc[IDX(i, j, k, I + 1, J + 1)]
#define VELU(i, j, k) (in_range(i, j, k, I + 1, J, K) ? U[IDX(i, j, k, I + 1, J)] : 0.0f)
#define VELV(i, j, k) (in_range(i, j, k, I, J + 1, K) ? V[IDX(i, j, k, I, J + 1)] : 0.0f)
#define VELW(i, j, k) (in_range(i, j, k, I, J, K + 1) ? W[IDX(i, j, k, I, J)] : 0.0f)

/* one coupling: matrix entry if the neighbour is FLUID, RHS term if SOLID
 * (viscositysolver.cpp:431-465 and the V/W analogues) */
#define COUPLE(STATE, ROW, VEL, coef)                                  \
    do {                                                               \
        int st_ = (STATE);                                             \
        if (st_ == FS_FLUID) rm_put(m, row, (ROW), (double)(coef), 1); \
    } while (0)
#define RHS(STATE, VEL, coef)                                \
    do {                                                     \
        if ((STATE) == FS_SOLID) rval -= (coef) * (VEL);     \
    } while (0)

static char *g_visc_dump = NULL; /* owned copy of the path (the caller's buffer may be a ctypes temporary) */
/* research hook: externally supplied control volumes (center,U,V,W,edgeU,edgeV,edgeW) and face states (U,V,W) replace
 * the ones oracle_viscosity_solve derives from phi / the solid SDF; NULL restores the normal path */
static const float *const *g_vol_override = NULL;
static const uint8_t *const *g_state_override = NULL;
void oracle_viscosity_override(const float *const *vols7, const uint8_t *const *states3) { g_vol_override = vols7; g_state_override = states3; }
/* when set (non-NULL path), the next oracle_viscosity_solve calls also write their assembled system to that file */
void oracle_viscosity_dump_to(const char *path) {
    free(g_visc_dump);
    g_visc_dump = NULL;
    if (path && path[0]) {
        size_t n = strlen(path) + 1;
        g_visc_dump = (char *)malloc(n);
        if (g_visc_dump) memcpy(g_visc_dump, path, n);
    }
}

void oracle_viscosity_solve(int I, int J, int K, float dxf, float dtf, float *U, float *V, float *W,
                            const float *phi, const float *solid, const float *visc, double tol, int maxiter,
                            double accept_tol, oracle_solve_info *info) {
    oracle_solve_info li;
    memset(&li, 0, sizeof(li));
    size_t nnode = (size_t)(I + 1) * (J + 1) * (K + 1);
    int nonzero = 0; /* fluidsimulation.cpp:171-184 */
    for (size_t c = 0; c < nnode; c++) if (visc[c] > 0.0) nonzero = 1;
    if (!nonzero) { li.status = 3; if (info) *info = li; return; }

    size_t nu = (size_t)(I + 1) * J * K, nv = (size_t)I * (J + 1) * K, nw = (size_t)I * J * (K + 1);
    vsys s;
    s.I = I; s.J = J; s.K = K;
    s.voff = (int)nu; s.woff = (int)(nu + nv);
    s.sU = (uint8_t *)malloc(nu); s.sV = (uint8_t *)malloc(nv); s.sW = (uint8_t *)malloc(nw);
    {
        /* viscositysolver.cpp:80-133 */
        float *scp = (float *)malloc((size_t)I * J * K * sizeof(float));
        for (int k = 0; k < K; k++)
            for (int j = 0; j < J; j++)
                for (int i = 0; i < I; i++) scp[IDX(i, j, k, I, J)] = solid_center_phi(i, j, k, solid, I, J);
        for (int k = 0; k < K; k++)
            for (int j = 0; j < J; j++)
                for (int i = 0; i < I + 1; i++) {
                    int edge = i == 0 || i == I;
                    s.sU[IDX(i, j, k, I + 1, J)] =
                        (edge || scp[IDX(i - 1, j, k, I, J)] + scp[IDX(i, j, k, I, J)] <= 0) ? FS_SOLID : FS_FLUID;
                }
        for (int k = 0; k < K; k++)
            for (int j = 0; j < J + 1; j++)
                for (int i = 0; i < I; i++) {
                    int edge = j == 0 || j == J;
                    s.sV[IDX(i, j, k, I, J + 1)] =
                        (edge || scp[IDX(i, j - 1, k, I, J)] + scp[IDX(i, j, k, I, J)] <= 0) ? FS_SOLID : FS_FLUID;
                }
        for (int k = 0; k < K + 1; k++)
            for (int j = 0; j < J; j++)
                for (int i = 0; i < I; i++) {
                    int edge = k == 0 || k == K;
                    s.sW[IDX(i, j, k, I, J)] =
                        (edge || scp[IDX(i, j, k - 1, I, J)] + scp[IDX(i, j, k, I, J)] <= 0) ? FS_SOLID : FS_FLUID;
                }
        free(scp);
    }
    if (g_state_override) {
        memcpy(s.sU, g_state_override[0], nu); memcpy(s.sV, g_state_override[1], nv); memcpy(s.sW, g_state_override[2], nw);
    }
    volgrids g;
    alloc_volgrids(I, J, K, &g);
    if (g_vol_override) {
        memcpy(g.c, g_vol_override[0], (size_t)I * J * K * sizeof(float));
        memcpy(g.U, g_vol_override[1], nu * sizeof(float));
        memcpy(g.V, g_vol_override[2], nv * sizeof(float));
        memcpy(g.W, g_vol_override[3], nw * sizeof(float));
        memcpy(g.eU, g_vol_override[4], (size_t)I * (J + 1) * (K + 1) * sizeof(float));
        memcpy(g.eV, g_vol_override[5], (size_t)(I + 1) * J * (K + 1) * sizeof(float));
        memcpy(g.eW, g_vol_override[6], (size_t)(I + 1) * (J + 1) * K * sizeof(float));
    } else {
        compute_volume_grids(I, J, K, (double)dxf, phi, &g);
    }
#define VC(i, j, k) g.c[IDX(i, j, k, I, J)]
#define VU(i, j, k) g.U[IDX(i, j, k, I + 1, J)]
#define VV(i, j, k) g.V[IDX(i, j, k, I, J + 1)]
#define VW(i, j, k) g.W[IDX(i, j, k, I, J)]
#define VEU(i, j, k) g.eU[IDX(i, j, k, I, J + 1)]
#define VEV(i, j, k) g.eV[IDX(i, j, k, I + 1, J)]
#define VEW(i, j, k) g.eW[IDX(i, j, k, I + 1, J + 1)]

    size_t dim = nu + nv + nw; /* viscositysolver.cpp:276-366 */
    s.table = (int *)malloc(dim * sizeof(int));
    for (size_t c = 0; c < dim; c++) s.table[c] = -1;
    for (int k = 1; k < K; k++)
        for (int j = 1; j < J; j++)
            for (int i = 1; i < I; i++) {
                if (stU(&s, i, j, k) == FS_FLUID) {
                    if (VU(i, j, k) > 0.0 || VC(i, j, k) > 0.0 || VC(i - 1, j, k) > 0.0 || VEW(i, j + 1, k) > 0.0 ||
                        VEW(i, j, k) > 0.0 || VEV(i, j, k + 1) > 0.0 || VEV(i, j, k) > 0.0)
                        s.table[IDX(i, j, k, I + 1, J)] = 0;
                }
                if (stV(&s, i, j, k) == FS_FLUID) {
                    if (VV(i, j, k) > 0.0 || VEW(i + 1, j, k) > 0.0 || VEW(i, j, k) > 0.0 || VC(i, j, k) > 0.0 ||
                        VC(i, j - 1, k) > 0.0 || VEU(i, j, k + 1) > 0.0 || VEU(i, j, k) > 0.0)
                        s.table[s.voff + IDX(i, j, k, I, J + 1)] = 0;
                }
                if (stW(&s, i, j, k) == FS_FLUID) {
                    if (VW(i, j, k) > 0.0 || VEV(i + 1, j, k) > 0.0 || VEV(i, j, k) > 0.0 || VEU(i, j + 1, k) > 0.0 ||
                        VEU(i, j, k) > 0.0 || VC(i, j, k) > 0.0 || VC(i, j, k - 1) > 0.0)
                        s.table[s.woff + IDX(i, j, k, I, J)] = 0;
                }
            }
    int n = 0;
    for (size_t c = 0; c < dim; c++) if (s.table[c] == 0) s.table[c] = n++;
    li.rows = n;

    rowmat M, *m = &M;
    M.n = n;
    M.cnt = (int *)calloc((size_t)n + 1, sizeof(int));
    M.col = (unsigned *)malloc(((size_t)n + 1) * ROWCAP * sizeof(unsigned));
    M.val = (double *)malloc(((size_t)n + 1) * ROWCAP * sizeof(double));
    double *rhs = (double *)calloc((size_t)n + 1, sizeof(double));
    /* research dump only: the diagonal summed in double (the EXACT operator's) and the rows' own volumes */
    double *dgx = g_visc_dump ? (double *)calloc((size_t)n + 1, sizeof(double)) : NULL, *dgv = g_visc_dump ? (double *)calloc((size_t)n + 1, sizeof(double)) : NULL;

    float invdx = 1.0f / dxf;
    float factor = dtf * invdx * invdx;
    /* U rows: viscositysolver.cpp:374-470 */
    for (int k = 1; k < K; k++)
        for (int j = 1; j < J; j++)
            for (int i = 1; i < I; i++) {
                if (stU(&s, i, j, k) != FS_FLUID) continue;
                int row = rowU(&s, i, j, k);
                if (row == -1) continue;
                float viscR = VISC(i, j, k), viscL = VISC(i - 1, j, k);
                float viscT = 0.25f * (VISC(i - 1, j + 1, k) + VISC(i - 1, j, k) + VISC(i, j + 1, k) + VISC(i, j, k));
                float viscB = 0.25f * (VISC(i - 1, j, k) + VISC(i - 1, j - 1, k) + VISC(i, j, k) + VISC(i, j - 1, k));
                float viscF = 0.25f * (VISC(i - 1, j, k + 1) + VISC(i - 1, j, k) + VISC(i, j, k + 1) + VISC(i, j, k));
                float viscK = 0.25f * (VISC(i - 1, j, k) + VISC(i - 1, j, k - 1) + VISC(i, j, k) + VISC(i, j, k - 1));
                float fR = 2 * factor * viscR * VC(i, j, k);
                float fL = 2 * factor * viscL * VC(i - 1, j, k);
                float fT = factor * viscT * VEW(i, j + 1, k);
                float fB = factor * viscB * VEW(i, j, k);
                float fF = factor * viscF * VEV(i, j, k + 1);
                float fK = factor * viscK * VEV(i, j, k);
                float diag = VU(i, j, k) + fR + fL + fT + fB + fF + fK;
                if (dgx) { dgx[row] = (double)VU(i, j, k) + (double)fR + (double)fL + (double)fT + (double)fB + (double)fF + (double)fK; dgv[row] = (double)VU(i, j, k); }
                rm_put(m, row, row, (double)diag, 0);
                COUPLE(stU(&s, i + 1, j, k), rowU(&s, i + 1, j, k), 0, -fR);
                COUPLE(stU(&s, i - 1, j, k), rowU(&s, i - 1, j, k), 0, -fL);
                COUPLE(stU(&s, i, j + 1, k), rowU(&s, i, j + 1, k), 0, -fT);
                COUPLE(stU(&s, i, j - 1, k), rowU(&s, i, j - 1, k), 0, -fB);
                COUPLE(stU(&s, i, j, k + 1), rowU(&s, i, j, k + 1), 0, -fF);
                COUPLE(stU(&s, i, j, k - 1), rowU(&s, i, j, k - 1), 0, -fK);
                COUPLE(stV(&s, i, j + 1, k), rowV(&s, i, j + 1, k), 0, -fT);
                COUPLE(stV(&s, i - 1, j + 1, k), rowV(&s, i - 1, j + 1, k), 0, fT);
                COUPLE(stV(&s, i, j, k), rowV(&s, i, j, k), 0, fB);
                COUPLE(stV(&s, i - 1, j, k), rowV(&s, i - 1, j, k), 0, -fB);
                COUPLE(stW(&s, i, j, k + 1), rowW(&s, i, j, k + 1), 0, -fF);
                COUPLE(stW(&s, i - 1, j, k + 1), rowW(&s, i - 1, j, k + 1), 0, fF);
                COUPLE(stW(&s, i, j, k), rowW(&s, i, j, k), 0, fK);
                COUPLE(stW(&s, i - 1, j, k), rowW(&s, i - 1, j, k), 0, -fK);
                float rval = VU(i, j, k) * VELU(i, j, k);
                RHS(stU(&s, i + 1, j, k), VELU(i + 1, j, k), -fR);
                RHS(stU(&s, i - 1, j, k), VELU(i - 1, j, k), -fL);
                RHS(stU(&s, i, j + 1, k), VELU(i, j + 1, k), -fT);
                RHS(stU(&s, i, j - 1, k), VELU(i, j - 1, k), -fB);
                RHS(stU(&s, i, j, k + 1), VELU(i, j, k + 1), -fF);
                RHS(stU(&s, i, j, k - 1), VELU(i, j, k - 1), -fK);
                RHS(stV(&s, i, j + 1, k), VELV(i, j + 1, k), -fT);
                RHS(stV(&s, i - 1, j + 1, k), VELV(i - 1, j + 1, k), fT);
                RHS(stV(&s, i, j, k), VELV(i, j, k), fB);
                RHS(stV(&s, i - 1, j, k), VELV(i - 1, j, k), -fB);
                RHS(stW(&s, i, j, k + 1), VELW(i, j, k + 1), -fF);
                RHS(stW(&s, i - 1, j, k + 1), VELW(i - 1, j, k + 1), fF);
                RHS(stW(&s, i, j, k), VELW(i, j, k), fK);
                RHS(stW(&s, i - 1, j, k), VELW(i - 1, j, k), -fK);
                rhs[row] = rval;
            }
    /* V rows: viscositysolver.cpp:472-568 */
    for (int k = 1; k < K; k++)
        for (int j = 1; j < J; j++)
            for (int i = 1; i < I; i++) {
                if (stV(&s, i, j, k) != FS_FLUID) continue;
                int row = rowV(&s, i, j, k);
                if (row == -1) continue;
                float viscR = 0.25f * (VISC(i, j - 1, k) + VISC(i + 1, j - 1, k) + VISC(i, j, k) + VISC(i + 1, j, k));
                float viscL = 0.25f * (VISC(i, j - 1, k) + VISC(i - 1, j - 1, k) + VISC(i, j, k) + VISC(i - 1, j, k));
                float viscT = VISC(i, j, k), viscB = VISC(i, j - 1, k);
                float viscF = 0.25f * (VISC(i, j - 1, k) + VISC(i, j - 1, k + 1) + VISC(i, j, k) + VISC(i, j, k + 1));
                float viscK = 0.25f * (VISC(i, j - 1, k) + VISC(i, j - 1, k - 1) + VISC(i, j, k) + VISC(i, j, k - 1));
                float fR = factor * viscR * VEW(i + 1, j, k);
                float fL = factor * viscL * VEW(i, j, k);
                float fT = 2 * factor * viscT * VC(i, j, k);
                float fB = 2 * factor * viscB * VC(i, j - 1, k);
                float fF = factor * viscF * VEU(i, j, k + 1);
                float fK = factor * viscK * VEU(i, j, k);
                float diag = VV(i, j, k) + fR + fL + fT + fB + fF + fK;
                if (dgx) { dgx[row] = (double)VV(i, j, k) + (double)fR + (double)fL + (double)fT + (double)fB + (double)fF + (double)fK; dgv[row] = (double)VV(i, j, k); }
                rm_put(m, row, row, (double)diag, 0);
                COUPLE(stV(&s, i + 1, j, k), rowV(&s, i + 1, j, k), 0, -fR);
                COUPLE(stV(&s, i - 1, j, k), rowV(&s, i - 1, j, k), 0, -fL);
                COUPLE(stV(&s, i, j + 1, k), rowV(&s, i, j + 1, k), 0, -fT);
                COUPLE(stV(&s, i, j - 1, k), rowV(&s, i, j - 1, k), 0, -fB);
                COUPLE(stV(&s, i, j, k + 1), rowV(&s, i, j, k + 1), 0, -fF);
                COUPLE(stV(&s, i, j, k - 1), rowV(&s, i, j, k - 1), 0, -fK);
                COUPLE(stU(&s, i + 1, j, k), rowU(&s, i + 1, j, k), 0, -fR);
                COUPLE(stU(&s, i + 1, j - 1, k), rowU(&s, i + 1, j - 1, k), 0, fR);
                COUPLE(stU(&s, i, j, k), rowU(&s, i, j, k), 0, fL);
                COUPLE(stU(&s, i, j - 1, k), rowU(&s, i, j - 1, k), 0, -fL);
                COUPLE(stW(&s, i, j, k + 1), rowW(&s, i, j, k + 1), 0, -fF);
                COUPLE(stW(&s, i, j - 1, k + 1), rowW(&s, i, j - 1, k + 1), 0, fF);
                COUPLE(stW(&s, i, j, k), rowW(&s, i, j, k), 0, fK);
                COUPLE(stW(&s, i, j - 1, k), rowW(&s, i, j - 1, k), 0, -fK);
                float rval = VV(i, j, k) * VELV(i, j, k);
                RHS(stV(&s, i + 1, j, k), VELV(i + 1, j, k), -fR);
                RHS(stV(&s, i - 1, j, k), VELV(i - 1, j, k), -fL);
                RHS(stV(&s, i, j + 1, k), VELV(i, j + 1, k), -fT);
                RHS(stV(&s, i, j - 1, k), VELV(i, j - 1, k), -fB);
                RHS(stV(&s, i, j, k + 1), VELV(i, j, k + 1), -fF);
                RHS(stV(&s, i, j, k - 1), VELV(i, j, k - 1), -fK);
                RHS(stU(&s, i + 1, j, k), VELU(i + 1, j, k), -fR);
                RHS(stU(&s, i + 1, j - 1, k), VELU(i + 1, j - 1, k), fR);
                RHS(stU(&s, i, j, k), VELU(i, j, k), fL);
                RHS(stU(&s, i, j - 1, k), VELU(i, j - 1, k), -fL);
                RHS(stW(&s, i, j, k + 1), VELW(i, j, k + 1), -fF);
                RHS(stW(&s, i, j - 1, k + 1), VELW(i, j - 1, k + 1), fF);
                RHS(stW(&s, i, j, k), VELW(i, j, k), fK);
                RHS(stW(&s, i, j - 1, k), VELW(i, j - 1, k), -fK);
                rhs[row] = rval;
            }
    /* W rows: viscositysolver.cpp:570-664 */
    for (int k = 1; k < K; k++)
        for (int j = 1; j < J; j++)
            for (int i = 1; i < I; i++) {
                if (stW(&s, i, j, k) != FS_FLUID) continue;
                int row = rowW(&s, i, j, k);
                if (row == -1) continue;
                float viscR = 0.25f * (VISC(i, j, k) + VISC(i, j, k - 1) + VISC(i + 1, j, k) + VISC(i + 1, j, k - 1));
                float viscL = 0.25f * (VISC(i, j, k) + VISC(i, j, k - 1) + VISC(i - 1, j, k) + VISC(i - 1, j, k - 1));
                float viscT = 0.25f * (VISC(i, j, k) + VISC(i, j, k - 1) + VISC(i, j + 1, k) + VISC(i, j + 1, k - 1));
                float viscB = 0.25f * (VISC(i, j, k) + VISC(i, j, k - 1) + VISC(i, j - 1, k) + VISC(i, j - 1, k - 1));
                float viscF = VISC(i, j, k), viscK = VISC(i, j, k - 1);
                float fR = factor * viscR * VEV(i + 1, j, k);
                float fL = factor * viscL * VEV(i, j, k);
                float fT = factor * viscT * VEU(i, j + 1, k);
                float fB = factor * viscB * VEU(i, j, k);
                float fF = 2 * factor * viscF * VC(i, j, k);
                float fK = 2 * factor * viscK * VC(i, j, k - 1);
                float diag = VW(i, j, k) + fR + fL + fT + fB + fF + fK;
                if (dgx) { dgx[row] = (double)VW(i, j, k) + (double)fR + (double)fL + (double)fT + (double)fB + (double)fF + (double)fK; dgv[row] = (double)VW(i, j, k); }
                rm_put(m, row, row, (double)diag, 0);
                COUPLE(stW(&s, i + 1, j, k), rowW(&s, i + 1, j, k), 0, -fR);
                COUPLE(stW(&s, i - 1, j, k), rowW(&s, i - 1, j, k), 0, -fL);
                COUPLE(stW(&s, i, j + 1, k), rowW(&s, i, j + 1, k), 0, -fT);
                COUPLE(stW(&s, i, j - 1, k), rowW(&s, i, j - 1, k), 0, -fB);
                COUPLE(stW(&s, i, j, k + 1), rowW(&s, i, j, k + 1), 0, -fF);
                COUPLE(stW(&s, i, j, k - 1), rowW(&s, i, j, k - 1), 0, -fK);
                COUPLE(stU(&s, i + 1, j, k), rowU(&s, i + 1, j, k), 0, -fR);
                COUPLE(stU(&s, i + 1, j, k - 1), rowU(&s, i + 1, j, k - 1), 0, fR);
                COUPLE(stU(&s, i, j, k), rowU(&s, i, j, k), 0, fL);
                COUPLE(stU(&s, i, j, k - 1), rowU(&s, i, j, k - 1), 0, -fL);
                COUPLE(stV(&s, i, j + 1, k), rowV(&s, i, j + 1, k), 0, -fT);
                COUPLE(stV(&s, i, j + 1, k - 1), rowV(&s, i, j + 1, k - 1), 0, fT);
                COUPLE(stV(&s, i, j, k), rowV(&s, i, j, k), 0, fB);
                COUPLE(stV(&s, i, j, k - 1), rowV(&s, i, j, k - 1), 0, -fB);
                float rval = VW(i, j, k) * VELW(i, j, k);
                RHS(stW(&s, i + 1, j, k), VELW(i + 1, j, k), -fR);
                RHS(stW(&s, i - 1, j, k), VELW(i - 1, j, k), -fL);
                RHS(stW(&s, i, j + 1, k), VELW(i, j + 1, k), -fT);
                RHS(stW(&s, i, j - 1, k), VELW(i, j - 1, k), -fB);
                RHS(stW(&s, i, j, k + 1), VELW(i, j, k + 1), -fF);
                RHS(stW(&s, i, j, k - 1), VELW(i, j, k - 1), -fK);
                RHS(stU(&s, i + 1, j, k), VELU(i + 1, j, k), -fR);
                RHS(stU(&s, i + 1, j, k - 1), VELU(i + 1, j, k - 1), fR);
                RHS(stU(&s, i, j, k), VELU(i, j, k), fL);
                RHS(stU(&s, i, j, k - 1), VELU(i, j, k - 1), -fL);
                RHS(stV(&s, i, j + 1, k), VELV(i, j + 1, k), -fT);
                RHS(stV(&s, i, j + 1, k - 1), VELV(i, j + 1, k - 1), fT);
                RHS(stV(&s, i, j, k), VELV(i, j, k), fB);
                RHS(stV(&s, i, j, k - 1), VELV(i, j, k - 1), -fB);
                rhs[row] = rval;
            }
    free_volgrids(&g);
    for (int r = 0; r < n; r++) li.nnz += M.cnt[r];
    if (g_visc_dump) { /* solver research hook (oracle_viscosity_dump_to): the assembled system as raw binary */
        FILE *f = fopen(g_visc_dump, "wb");
        if (f) {
            long long hdr[4] = {n, ROWCAP, (long long)dim, (dgx && dgv) ? 1 : 0};
            fwrite(hdr, sizeof(hdr), 1, f);
            fwrite(M.cnt, sizeof(int), (size_t)n, f);
            fwrite(M.col, sizeof(unsigned), (size_t)n * ROWCAP, f);
            fwrite(M.val, sizeof(double), (size_t)n * ROWCAP, f);
            fwrite(rhs, sizeof(double), (size_t)n, f);
            fwrite(s.table, sizeof(int), dim, f);
            if (dgx && dgv) { fwrite(dgx, sizeof(double), (size_t)n, f); fwrite(dgv, sizeof(double), (size_t)n, f); }
            fclose(f);
        }
    }
    free(dgx); free(dgv);

    /* ---- PCGSolver<double>::solve (pcgsolver.h:241-295) ---- */
    double *x = (double *)calloc((size_t)n + 1, sizeof(double));
    double *r = (double *)malloc(((size_t)n + 1) * sizeof(double));
    double *z = (double *)malloc(((size_t)n + 1) * sizeof(double));
    double *sv = (double *)malloc(((size_t)n + 1) * sizeof(double));
    memcpy(r, rhs, (size_t)n * sizeof(double));
    double res;
    {
        double mv = 0; /* blaswrapper.h:24-43 */
        for (int c = 0; c < n; c++) if (fabs(r[c]) > mv) mv = fabs(r[c]);
        res = mv;
    }
    int success = 0, iters = 0;
    unsigned *colstart = NULL, *rowindex = NULL;
    double *fval = NULL, *invdiag = NULL, *adiag = NULL;
    if (n == 0 || res == 0) {
        success = 1; iters = 0;
    } else {
        double tolabs = tol * res;
        /* factorModifiedIncompleteColesky0 (pcgsolver.h:62-178), parameters 0.97 / 0.25 */
        colstart = (unsigned *)malloc(((size_t)n + 1) * sizeof(unsigned));
        invdiag = (double *)calloc((size_t)n, sizeof(double));
        adiag = (double *)calloc((size_t)n, sizeof(double));
        size_t nlow = 0;
        for (int i = 0; i < n; i++)
            for (int t = 0; t < M.cnt[i]; t++) if (M.col[(size_t)i * ROWCAP + t] > (unsigned)i) nlow++;
        rowindex = (unsigned *)malloc((nlow + 1) * sizeof(unsigned));
        fval = (double *)malloc((nlow + 1) * sizeof(double));
        size_t pos = 0;
        for (int i = 0; i < n; i++) {
            colstart[i] = (unsigned)pos;
            for (int t = 0; t < M.cnt[i]; t++) {
                unsigned c = M.col[(size_t)i * ROWCAP + t];
                double v = M.val[(size_t)i * ROWCAP + t];
                if (c > (unsigned)i) { rowindex[pos] = c; fval[pos] = v; pos++; }
                else if (c == (unsigned)i) { invdiag[i] = adiag[i] = v; }
            }
        }
        colstart[n] = (unsigned)pos;
        const double modpar = 0.97, mindiag = 0.25;
        for (int k = 0; k < n; k++) {
            if (adiag[k] == 0) continue;
            if (invdiag[k] < mindiag * adiag[k]) invdiag[k] = 1 / sqrt(adiag[k]);
            else invdiag[k] = 1 / sqrt(invdiag[k]);
            for (unsigned p = colstart[k]; p < colstart[k + 1]; p++) fval[p] *= invdiag[k];
            for (unsigned p = colstart[k]; p < colstart[k + 1]; p++) {
                unsigned j = rowindex[p];
                double multiplier = fval[p];
                double missing = 0;
                unsigned a = colstart[k];
                unsigned b = 0;
                const unsigned *rj = M.col + (size_t)j * ROWCAP;
                unsigned rjn = (unsigned)M.cnt[j];
                while (a < colstart[k + 1] && rowindex[a] < j) {
                    while (b < rjn) {
                        if (rj[b] < rowindex[a]) b++;
                        else if (rj[b] == rowindex[a]) break;
                        else { missing += fval[a]; break; }
                    }
                    a++;
                }
                if (a < colstart[k + 1] && rowindex[a] == j) invdiag[j] -= multiplier * fval[a];
                a++;
                b = colstart[j];
                while (a < colstart[k + 1] && b < colstart[j + 1]) {
                    if (rowindex[b] < rowindex[a]) b++;
                    else if (rowindex[b] == rowindex[a]) { fval[b] -= multiplier * fval[a]; a++; b++; }
                    else { missing += fval[a]; a++; }
                }
                while (a < colstart[k + 1]) { missing += fval[a]; a++; }
                invdiag[j] -= modpar * multiplier * missing;
            }
        }
#define APPLY_PRECON(src, dst)                                                                     \
    do { /* solveLower + solveLowerTransposeInPlace (pcgsolver.h:184-214) */                       \
        memcpy((dst), (src), (size_t)n * sizeof(double));                                          \
        for (int i_ = 0; i_ < n; i_++) {                                                           \
            (dst)[i_] *= invdiag[i_];                                                              \
            for (unsigned j_ = colstart[i_]; j_ < colstart[i_ + 1]; j_++)                          \
                (dst)[rowindex[j_]] -= fval[j_] * (dst)[i_];                                       \
        }                                                                                          \
        for (int i_ = n - 1; i_ >= 0; i_--) {                                                      \
            for (unsigned j_ = colstart[i_]; j_ < colstart[i_ + 1]; j_++)                          \
                (dst)[i_] -= fval[j_] * (dst)[rowindex[j_]];                                       \
            (dst)[i_] *= invdiag[i_];                                                              \
        }                                                                                          \
    } while (0)
        APPLY_PRECON(r, z);
        double rho = vdot(z, r, n);
        if (rho == 0 || rho != rho) {
            success = 0; iters = 0;
        } else {
            memcpy(sv, z, (size_t)n * sizeof(double));
            int it;
            success = 0;
            const char *te_ = getenv("ORACLE_TRACE_EVERY");   /* research: the residual history of a long converged solve (test infrastructure; the library reads no environment) */
            const int trace_every = te_ ? atoi(te_) : 0;
            for (it = 0; it < maxiter; it++) {
                OMP_FOR
                for (int i = 0; i < n; i++) { /* multiply (sparsematrix.h:166-176) */
                    double acc = 0;
                    for (int t = 0; t < M.cnt[i]; t++)
                        acc += M.val[(size_t)i * ROWCAP + t] * sv[M.col[(size_t)i * ROWCAP + t]];
                    z[i] = acc;
                }
                double alpha = rho / vdot(sv, z, n);
                OMP_FOR
                for (int i = 0; i < n; i++) x[i] += alpha * sv[i];
                OMP_FOR
                for (int i = 0; i < n; i++) r[i] += (-alpha) * z[i];
                double mv = 0;
                OMP_FOR_MAX(mv)
                for (int c = 0; c < n; c++) if (fabs(r[c]) > mv) mv = fabs(r[c]);
                res = mv;
                if (trace_every > 0 && (it + 1) % trace_every == 0) { fprintf(stderr, "  oracle viscosity PCG: iteration %d, max|r| %.3e (target %.3e)\n", it + 1, res, tolabs); fflush(stderr); }
                if (res <= tolabs) { iters = it + 1; success = 1; break; }
                APPLY_PRECON(r, z);
                double rho_new = vdot(z, r, n);
                double beta = rho_new / rho;
                OMP_FOR
                for (int i = 0; i < n; i++) z[i] += beta * sv[i];
                double *tmp = sv; sv = z; z = tmp;
                rho = rho_new;
            }
            if (!success) iters = it;
        }
    }
    if (g_visc_dump) { /* research hook: the iterate the solve ended with, beside the dumped system */
        size_t L = strlen(g_visc_dump);
        char *px = (char *)malloc(L + 3);
        if (px) {
            memcpy(px, g_visc_dump, L); memcpy(px + L, ".x", 3);
            FILE *f = fopen(px, "wb");
            if (f) { fwrite(x, sizeof(double), (size_t)n, f); fclose(f); }
            free(px);
        }
    }
    li.iterations = iters;
    li.residual = res;
    /* viscositysolver.cpp:676-689 */
    int accepted = success || (iters == maxiter && res < accept_tol);
    li.status = success ? 0 : (accepted ? 1 : 2);
    if (accepted) { /* viscositysolver.cpp:692-727 */
        memset(U, 0, nu * sizeof(float));
        memset(V, 0, nv * sizeof(float));
        memset(W, 0, nw * sizeof(float));
        for (size_t c = 0; c < nu; c++) if (s.table[c] != -1) U[c] = (float)x[s.table[c]];
        for (size_t c = 0; c < nv; c++) if (s.table[s.voff + c] != -1) V[c] = (float)x[s.table[s.voff + c]];
        for (size_t c = 0; c < nw; c++) if (s.table[s.woff + c] != -1) W[c] = (float)x[s.table[s.woff + c]];
    }
    free(colstart); free(rowindex); free(fval); free(invdiag); free(adiag);
    free(x); free(r); free(z); free(sv); free(rhs);
    free(M.cnt); free(M.col); free(M.val);
    free(s.sU); free(s.sV); free(s.sW); free(s.table);
    if (info) *info = li;
}

/* ---------------- K15: grid -> particle, RK2, collision ---------------- */

/* MACVelocityField::_interpolateLinearU/V/W (macvelocityfield.cpp:455-546), everything in double */
static double mac_lerp(int dir, double x, double y, double z, double dx, int I, int J, int K, const float *g) {
    if (!(x >= 0 && y >= 0 && z >= 0 && x < dx * I && y < dx * J && z < dx * K)) return 0.0;
    int w = I + (dir == 0), h = J + (dir == 1), d = K + (dir == 2);
    if (dir != 0) x -= 0.5 * dx;
    if (dir != 1) y -= 0.5 * dx;
    if (dir != 2) z -= 0.5 * dx;
    double invdx = 1.0 / dx;
    int i = (int)floor(x * invdx), j = (int)floor(y * invdx), k = (int)floor(z * invdx);
    double gx = (double)i * dx, gy = (double)j * dx, gz = (double)k * dx;
    double inv_dx = 1 / dx;
    double ix = (x - gx) * inv_dx, iy = (y - gy) * inv_dx, iz = (z - gz) * inv_dx;
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (in_range(i, j, k, w, h, d)) p[0] = g[IDX(i, j, k, w, h)];
    if (in_range(i + 1, j, k, w, h, d)) p[1] = g[IDX(i + 1, j, k, w, h)];
    if (in_range(i, j + 1, k, w, h, d)) p[2] = g[IDX(i, j + 1, k, w, h)];
    if (in_range(i, j, k + 1, w, h, d)) p[3] = g[IDX(i, j, k + 1, w, h)];
    if (in_range(i + 1, j, k + 1, w, h, d)) p[4] = g[IDX(i + 1, j, k + 1, w, h)];
    if (in_range(i, j + 1, k + 1, w, h, d)) p[5] = g[IDX(i, j + 1, k + 1, w, h)];
    if (in_range(i + 1, j + 1, k, w, h, d)) p[6] = g[IDX(i + 1, j + 1, k, w, h)];
    if (in_range(i + 1, j + 1, k + 1, w, h, d)) p[7] = g[IDX(i + 1, j + 1, k + 1, w, h)];
    return p[0] * (1 - ix) * (1 - iy) * (1 - iz) + p[1] * ix * (1 - iy) * (1 - iz) +
           p[2] * (1 - ix) * iy * (1 - iz) + p[3] * (1 - ix) * (1 - iy) * iz + p[4] * ix * (1 - iy) * iz +
           p[5] * (1 - ix) * iy * iz + p[6] * ix * iy * (1 - iz) + p[7] * ix * iy * iz;
}

/* MACVelocityField::evaluateVelocityAtPositionLinear (macvelocityfield.cpp:564-578) */
static void mac_velocity(float px, float py, float pz, double dx, int I, int J, int K, const float *U, const float *V,
                         const float *W, float out[3]) {
    double x = px, y = py, z = pz;
    if (!(x >= 0 && y >= 0 && z >= 0 && x < dx * I && y < dx * J && z < dx * K)) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    out[0] = (float)mac_lerp(0, x, y, z, dx, I, J, K, U);
    out[1] = (float)mac_lerp(1, x, y, z, dx, I, J, K, V);
    out[2] = (float)mac_lerp(2, x, y, z, dx, I, J, K, W);
}

void oracle_update_particle_velocities(int I, int J, int K, float dxf, float *aos6, size_t n, const float *U,
                                       const float *V, const float *W, const float *sU, const float *sV,
                                       const float *sW, float ratio) {
    double dx = (double)dxf;
    OMP_FOR
    for (size_t p = 0; p < n; p++) { /* fluidsimulation.cpp:341-352 */
        float *q = aos6 + 6 * p;
        float vn[3], vo[3];
        mac_velocity(q[0], q[1], q[2], dx, I, J, K, U, V, W, vn);
        mac_velocity(q[0], q[1], q[2], dx, I, J, K, sU, sV, sW, vo);
        for (int c = 0; c < 3; c++) {
            float pic = vn[c];
            float flip = q[3 + c] + vn[c] - vo[c];
            q[3 + c] = ratio * pic + (1.0f - ratio) * flip;
        }
    }
}

void oracle_advect_particles(int I, int J, int K, float dxf, float dt, float *aos6, size_t n, const float *U,
                             const float *V, const float *W, const float *sU, const float *sV, const float *sW,
                             const float *solid, float ratio) {
    oracle_update_particle_velocities(I, J, K, dxf, aos6, n, U, V, W, sU, sV, sW, ratio);
    double dx = (double)dxf;
    /* AABB boundary(0,0,0, I*dx, J*dx, K*dx); expand(-2*dx - 1e-4)  (fluidsimulation.cpp:319-320, aabb.cpp:118-124) */
    double bw = (double)(I * dxf), bh = (double)(J * dxf), bd = (double)(K * dxf);
    double ev = -2 * dxf - 1e-4;
    double eh = 0.5 * ev;
    float bx = 0.0f - (float)eh, by = 0.0f - (float)eh, bz = 0.0f - (float)eh;
    bw += ev; bh += ev; bd += ev;
    int nw = I + 1, nh = J + 1, nd = K + 1;
    OMP_FOR
    for (size_t p = 0; p < n; p++) {
        float *q = aos6 + 6 * p;
        /* _traceRK2 (fluidsimulation.cpp:535-541) */
        float v[3];
        mac_velocity(q[0], q[1], q[2], dx, I, J, K, U, V, W, v);
        float hs = 0.5f * dt;
        mac_velocity(q[0] + hs * v[0], q[1] + hs * v[1], q[2] + hs * v[2], dx, I, J, K, U, V, W, v);
        float x = q[0] + dt * v[0], y = q[1] + dt * v[1], z = q[2] + dt * v[2];
        /* fluidsimulation.cpp:326-333 */
        float phi_val = (float)trilerp_field(x, y, z, dx, solid, nw, nh, nd);
        if (phi_val < 0) {
            float g[3];
            trilerp_gradient(x, y, z, dx, solid, nw, nh, nd, g);
            float lsq = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            if (lsq > 0) {
                float len = sqrtf(lsq);
                float inv = 1.0f / len;
                g[0] *= inv; g[1] *= inv; g[2] *= inv;
            }
            x -= phi_val * g[0]; y -= phi_val * g[1]; z -= phi_val * g[2];
        }
        /* AABB::isPointInside / getNearestPointInsideAABB (aabb.cpp:126-129, 213-234) */
        int inside = x >= bx && y >= by && z >= bz && x < bx + bw && y < by + bh && z < bz + bd;
        if (!inside) {
            float mx = bx + (float)bw, my = by + (float)bh, mz = bz + (float)bd;
            float eps = (float)1e-6;
            x = fmaxf(x, bx); y = fmaxf(y, by); z = fmaxf(z, bz);
            x = fminf(x, mx - eps); y = fminf(y, my - eps); z = fminf(z, mz - eps);
        }
        q[0] = x; q[1] = y; q[2] = z;
    }
}

/* ---------------- whole simulation ---------------- */

struct oracle_sim {
    int I, J, K;
    float dx;
    float *U, *V, *W, *sU, *sV, *sW, *phi, *solid, *wU, *wV, *wW, *visc, *pressure;
    uint8_t *vU, *vV, *vW;
    float *particles;
    size_t np;
    float gravity[3];
    float cfl_number, minfrac, pic_ratio;
    double ptol, vtol, vaccept;
    int pmaxiter, vmaxiter;
    oracle_solve_info last_visc, last_pres;
};

oracle_sim *oracle_sim_create(int I, int J, int K, float dx) {
    oracle_sim *s = (oracle_sim *)calloc(1, sizeof(oracle_sim));
    s->I = I; s->J = J; s->K = K; s->dx = dx;
    size_t nu = (size_t)(I + 1) * J * K, nv = (size_t)I * (J + 1) * K, nw = (size_t)I * J * (K + 1);
    size_t nc = (size_t)I * J * K, nn = (size_t)(I + 1) * (J + 1) * (K + 1);
    s->U = calloc(nu, 4); s->V = calloc(nv, 4); s->W = calloc(nw, 4);
    s->sU = calloc(nu, 4); s->sV = calloc(nv, 4); s->sW = calloc(nw, 4);
    s->wU = calloc(nu, 4); s->wV = calloc(nv, 4); s->wW = calloc(nw, 4);
    s->vU = calloc(nu, 1); s->vV = calloc(nv, 1); s->vW = calloc(nw, 1);
    s->phi = calloc(nc, 4); s->pressure = calloc(nc, 4);
    s->solid = calloc(nn, 4); s->visc = calloc(nn, 4);
    for (size_t c = 0; c < nc; c++) s->phi[c] = 3.0f * dx;
    for (size_t c = 0; c < nn; c++) s->visc[c] = 1.0f; /* fluidsimulation.cpp:39 */
    s->gravity[0] = 0.0f; s->gravity[1] = -9.81f; s->gravity[2] = 0.0f; /* fluidsimulation.cpp:40 */
    s->cfl_number = 5.0f; s->minfrac = 0.01f; s->pic_ratio = 0.05f;    /* fluidsimulation.h:128-130 */
    s->ptol = 1e-9; s->pmaxiter = 200;                                  /* pressuresolver.h:224-225 */
    s->vtol = 1e-6; s->vmaxiter = 700; s->vaccept = 10.0;               /* viscositysolver.h:200-202 */
    return s;
}

void oracle_sim_destroy(oracle_sim *s) {
    if (!s) return;
    free(s->U); free(s->V); free(s->W); free(s->sU); free(s->sV); free(s->sW);
    free(s->wU); free(s->wV); free(s->wW); free(s->vU); free(s->vV); free(s->vW);
    free(s->phi); free(s->pressure); free(s->solid); free(s->visc); free(s->particles);
    free(s);
}

void oracle_sim_set_solid(oracle_sim *s, const float *nodes) {
    memcpy(s->solid, nodes, (size_t)(s->I + 1) * (s->J + 1) * (s->K + 1) * 4);
}
void oracle_sim_set_viscosity(oracle_sim *s, const float *nodes) {
    memcpy(s->visc, nodes, (size_t)(s->I + 1) * (s->J + 1) * (s->K + 1) * 4);
}
void oracle_sim_set_gravity(oracle_sim *s, float gx, float gy, float gz) {
    s->gravity[0] = gx; s->gravity[1] = gy; s->gravity[2] = gz;
}
void oracle_sim_set_particles(oracle_sim *s, const float *aos6, size_t n) {
    free(s->particles);
    s->particles = (float *)malloc((n + 1) * 6 * sizeof(float));
    memcpy(s->particles, aos6, n * 6 * sizeof(float));
    s->np = n;
}
size_t oracle_sim_num_particles(oracle_sim *s) { return s->np; }
void oracle_sim_get_particles(oracle_sim *s, float *aos6) { memcpy(aos6, s->particles, s->np * 6 * sizeof(float)); }
void oracle_sim_set_solver_limits(oracle_sim *s, double ptol, int pmaxiter, double vtol, int vmaxiter) {
    if (ptol > 0) s->ptol = ptol;
    if (pmaxiter > 0) s->pmaxiter = pmaxiter;
    if (vtol > 0) s->vtol = vtol;
    if (vmaxiter > 0) s->vmaxiter = vmaxiter;
}

static int sim_grid(oracle_sim *s, int which, float **f, uint8_t **m, size_t *n) {
    int I = s->I, J = s->J, K = s->K;
    size_t nu = (size_t)(I + 1) * J * K, nv = (size_t)I * (J + 1) * K, nw = (size_t)I * J * (K + 1);
    size_t nc = (size_t)I * J * K, nn = (size_t)(I + 1) * (J + 1) * (K + 1);
    *f = NULL; *m = NULL;
    switch (which) {
        case 0: *f = s->U; *n = nu; return 0;
        case 1: *f = s->V; *n = nv; return 0;
        case 2: *f = s->W; *n = nw; return 0;
        case 3: *f = s->sU; *n = nu; return 0;
        case 4: *f = s->sV; *n = nv; return 0;
        case 5: *f = s->sW; *n = nw; return 0;
        case 6: *m = s->vU; *n = nu; return 0;
        case 7: *m = s->vV; *n = nv; return 0;
        case 8: *m = s->vW; *n = nw; return 0;
        case 9: *f = s->phi; *n = nc; return 0;
        case 10: *f = s->solid; *n = nn; return 0;
        case 11: *f = s->wU; *n = nu; return 0;
        case 12: *f = s->wV; *n = nv; return 0;
        case 13: *f = s->wW; *n = nw; return 0;
        case 14: *f = s->visc; *n = nn; return 0;
        case 15: *f = s->pressure; *n = nc; return 0;
    }
    return -1;
}
int oracle_sim_get_grid(oracle_sim *s, int which, float *out) {
    float *f; uint8_t *m; size_t n;
    if (sim_grid(s, which, &f, &m, &n)) return -1;
    if (f) memcpy(out, f, n * 4);
    else for (size_t c = 0; c < n; c++) out[c] = m[c] ? 1.0f : 0.0f;
    return 0;
}
int oracle_sim_set_grid(oracle_sim *s, int which, const float *in) {
    float *f; uint8_t *m; size_t n;
    if (sim_grid(s, which, &f, &m, &n)) return -1;
    if (f) memcpy(f, in, n * 4);
    else for (size_t c = 0; c < n; c++) m[c] = in[c] != 0.0f;
    return 0;
}

static void sim_extrapolate(oracle_sim *s) { /* fluidsimulation.cpp:690-694 */
    int layers = (int)ceil(s->cfl_number) + 2;
    oracle_extrapolate_grid(s->I + 1, s->J, s->K, s->U, s->vU, layers);
    oracle_extrapolate_grid(s->I, s->J + 1, s->K, s->V, s->vV, layers);
    oracle_extrapolate_grid(s->I, s->J, s->K + 1, s->W, s->vW, layers);
}

void oracle_sim_substep(oracle_sim *s, float dt, double *seconds, oracle_solve_info *visc, oracle_solve_info *pres) {
    int I = s->I, J = s->J, K = s->K;
    size_t nu = (size_t)(I + 1) * J * K, nv = (size_t)I * (J + 1) * K, nw = (size_t)I * J * (K + 1);
    double t0 = now_s(), t1;
#define LAP(idx) do { t1 = now_s(); if (seconds) seconds[idx] = t1 - t0; t0 = t1; } while (0)
    oracle_particle_sdf(I, J, K, s->dx, s->particles, s->np, s->solid, s->phi);
    LAP(0);
    oracle_p2g(I, J, K, s->dx, s->particles, s->np, s->phi, s->U, s->V, s->W, s->vU, s->vV, s->vW);
    sim_extrapolate(s);
    memcpy(s->sU, s->U, nu * 4); memcpy(s->sV, s->V, nv * 4); memcpy(s->sW, s->W, nw * 4);
    LAP(1);
    oracle_body_force(I, J, K, s->phi, s->U, s->V, s->W, s->gravity[0], s->gravity[1], s->gravity[2], dt);
    LAP(2);
    oracle_viscosity_solve(I, J, K, s->dx, dt, s->U, s->V, s->W, s->phi, s->solid, s->visc, s->vtol, s->vmaxiter,
                           s->vaccept, &s->last_visc);
    LAP(3);
    oracle_compute_weights(I, J, K, s->solid, s->wU, s->wV, s->wW);
    oracle_pressure_solve(I, J, K, s->dx, dt, s->U, s->V, s->W, s->wU, s->wV, s->wW, s->phi, s->minfrac, s->ptol,
                          s->pmaxiter, s->pressure, &s->last_pres);
    oracle_apply_pressure(I, J, K, s->dx, dt, s->pressure, s->phi, s->wU, s->wV, s->wW, s->minfrac, s->U, s->V, s->W,
                          s->vU, s->vV, s->vW);
    sim_extrapolate(s);
    LAP(4);
    oracle_constrain(I, J, K, s->wU, s->wV, s->wW, s->U, s->V, s->W, s->sU, s->sV, s->sW);
    LAP(5);
    oracle_advect_particles(I, J, K, s->dx, dt, s->particles, s->np, s->U, s->V, s->W, s->sU, s->sV, s->sW, s->solid,
                            s->pic_ratio);
    LAP(6);
    if (visc) *visc = s->last_visc;
    if (pres) *pres = s->last_pres;
}

int oracle_sim_advance(oracle_sim *s, float dt) { /* fluidsimulation.cpp:135-168 */
    float t = 0;
    int nsub = 0;
    while (t < dt) {
        float substep = oracle_cfl(s->I, s->J, s->K, s->dx, s->U, s->V, s->W, s->cfl_number);
        if (t + substep > dt) substep = dt - t;
        oracle_sim_substep(s, substep, NULL, NULL, NULL);
        t += substep;
        nsub++;
    }
    return nsub;
}
