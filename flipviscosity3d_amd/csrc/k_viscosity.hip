// k_viscosity.hip -- variational (Batty-Bridson) viscosity solve, matrix-free.
// Reference: ViscositySolver::applyViscosityToVelocityField (viscositysolver.cpp:41-727) on top of the
// generic CSR PCGSolver<double> (pcgsolver/pcgsolver.h).
//
// The reference assembles a double CSR matrix with <= 15 non-zeros per row (~180 B per row) and runs
// MIC(0)-PCG on it.  Here the operator is never assembled: every coefficient of a row is one of six
// "factors" f = dt/dx^2 * nu * volume that live on four lattices (cell centres and the three edge
// families), so the coupled 15-point SpMV over the U, V and W rows of one index (i,j,k) reads
//   3 diagonals + 4 factor arrays + 3 x  and writes 3 y  = 52 B per swept index in fp32,
// against ~540 B for the assembled form.  Unknown faces are exactly the faces with a
// non-zero diagonal; x is kept 0 everywhere else, which reproduces the reference's silent drop of
// couplings to faces without a matrix row (sparsematrix.h:86-88).
#include "flipv_internal.h"
#include "pcg_common.h"

#define GRID3(w, h, d) dim3(cdiv((w), 64), cdiv((h), 4), (unsigned)(d)), dim3(64, 4, 1)

enum { ST_FLUID = 1, ST_SOLID = 2 };

// ------------------------------------------------------------------ face states
// viscositysolver.cpp:80-133
__global__ void k_solid_center(const float *__restrict__ solid, float *__restrict__ scp, int I, int J, int K) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= I || j >= J) return;
    scp[DIDX(i, j, k, I, J)] = d_solid_center(solid, i, j, k, I, J);
}

__global__ void k_face_states(int dir, const float *__restrict__ scp, uint8_t *__restrict__ st, int I, int J, int K) {
    const int w = I + (dir == 0), h = J + (dir == 1);
    const int n = dir == 0 ? I : (dir == 1 ? J : K);
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    const int cd = dir == 0 ? i : (dir == 1 ? j : k);
    bool solid = cd == 0 || cd == n;
    if (!solid) {
        const float a = scp[DIDX(i - (dir == 0), j - (dir == 1), k - (dir == 2), I, J)];
        const float b = scp[DIDX(i, j, k, I, J)];
        solid = a + b <= 0.0f;
    }
    st[DIDX(i, j, k, w, h)] = solid ? ST_SOLID : ST_FLUID;
}

// ------------------------------------------------------------------ band mask
// viscositysolver.cpp:138-168: phi<0 cells on an (I+1,J+1,K+1) mask, then two 6-neighbour dilations
__global__ void k_valid_init(const float *__restrict__ phi, uint8_t *__restrict__ m, int I, int J, int K) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i > I || j > J) return;
    uint8_t v = 0;
    if (i < I && j < J && k < K) v = phi[DIDX(i, j, k, I, J)] < 0.0f;
    m[DIDX(i, j, k, I + 1, J + 1)] = v;
}
__global__ void k_valid_dilate(const uint8_t *__restrict__ a, uint8_t *__restrict__ b, int w, int h, int d) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    const size_t c = DIDX(i, j, k, w, h);
    uint8_t v = a[c];
    if (i > 0) v |= a[c - 1];
    if (i < w - 1) v |= a[c + 1];
    if (j > 0) v |= a[c - w];
    if (j < h - 1) v |= a[c + w];
    if (k > 0) v |= a[c - (size_t)w * h];
    if (k < d - 1) v |= a[c + (size_t)w * h];
    b[c] = v;
}

// ------------------------------------------------------------------ control volumes
// liquid phi sampled like ParticleLevelSet::trilinearInterpolate (particlelevelset.cpp:88-92 ->
// interpolation.cpp:68-108): float position, fp64 weights, out-of-range corners = 0
__device__ __forceinline__ float d_liquid_phi_at(float px, float py, float pz, double dx, double invdx, float hdx,
                                                 const float *__restrict__ phi, int I, int J, int K) {
    px -= hdx; py -= hdx; pz -= hdx;
    const int gi = (int)floor((double)px * invdx), gj = (int)floor((double)py * invdx), gk = (int)floor((double)pz * invdx);
    const float gx = (float)(gi * dx), gy = (float)(gj * dx), gz = (float)(gk * dx);
    const double ix = (px - gx) * invdx, iy = (py - gy) * invdx, iz = (pz - gz) * invdx;
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
    if (d_in_range(gi, gj, gk, I, J, K)) p0 = phi[DIDX(gi, gj, gk, I, J)];
    if (d_in_range(gi + 1, gj, gk, I, J, K)) p1 = phi[DIDX(gi + 1, gj, gk, I, J)];
    if (d_in_range(gi, gj + 1, gk, I, J, K)) p2 = phi[DIDX(gi, gj + 1, gk, I, J)];
    if (d_in_range(gi, gj, gk + 1, I, J, K)) p3 = phi[DIDX(gi, gj, gk + 1, I, J)];
    if (d_in_range(gi + 1, gj, gk + 1, I, J, K)) p4 = phi[DIDX(gi + 1, gj, gk + 1, I, J)];
    if (d_in_range(gi, gj + 1, gk + 1, I, J, K)) p5 = phi[DIDX(gi, gj + 1, gk + 1, I, J)];
    if (d_in_range(gi + 1, gj + 1, gk, I, J, K)) p6 = phi[DIDX(gi + 1, gj + 1, gk, I, J)];
    if (d_in_range(gi + 1, gj + 1, gk + 1, I, J, K)) p7 = phi[DIDX(gi + 1, gj + 1, gk + 1, I, J)];
    return (float)(p0 * (1 - ix) * (1 - iy) * (1 - iz) + p1 * ix * (1 - iy) * (1 - iz) + p2 * (1 - ix) * iy * (1 - iz) +
                   p3 * (1 - ix) * (1 - iy) * iz + p4 * ix * (1 - iy) * iz + p5 * (1 - ix) * iy * iz +
                   p6 * ix * iy * (1 - iz) + p7 * ix * iy * iz);
}

__device__ __forceinline__ float d_tet(float a, float b, float c, float d) { return a * a * a / ((a - b) * (a - c) * (a - d)); }
__device__ __forceinline__ float d_prism(float p0, float p1, float p2, float p3) {
    const float a = p0 / (p0 - p2), b = p0 / (p0 - p3), c = p1 / (p1 - p3), d = p1 / (p1 - p2);
    return a * b * (1 - d) + b * (1 - c) * d + c * d;
}
#define DCSWAP(x, y) do { if ((x) > (y)) { const float t_ = (x); (x) = (y); (y) = t_; } } while (0)
// LevelsetUtils::volumeFraction, tetrahedron (levelsetutils.cpp:189-202, sort network levelsetutils.h:69-77)
__device__ __forceinline__ float d_tet_fraction(float p0, float p1, float p2, float p3) {
    DCSWAP(p0, p1); DCSWAP(p2, p3); DCSWAP(p0, p2); DCSWAP(p1, p3); DCSWAP(p1, p2);
    if (p3 <= 0) return 1.0f;
    if (p2 <= 0) return 1.0f - d_tet(p3, p2, p1, p0);
    if (p1 <= 0) return d_prism(p0, p1, p2, p3);
    if (p0 <= 0) return d_tet(p0, p1, p2, p3);
    return 0.0f;
}
// cube = average of the two 5-tet decompositions (levelsetutils.cpp:219-235)
__device__ __forceinline__ float d_cube_fraction(float p000, float p100, float p010, float p110, float p001,
                                                 float p101, float p011, float p111) {
    return (d_tet_fraction(p000, p001, p101, p011) + d_tet_fraction(p000, p101, p100, p110) +
            d_tet_fraction(p000, p010, p011, p110) + d_tet_fraction(p101, p011, p111, p110) +
            2 * d_tet_fraction(p000, p011, p101, p110) + d_tet_fraction(p100, p101, p001, p111) +
            d_tet_fraction(p100, p001, p000, p010) + d_tet_fraction(p100, p110, p111, p010) +
            d_tet_fraction(p001, p111, p011, p010) + 2 * d_tet_fraction(p100, p111, p001, p010)) /
           12.0f;
}

// _estimateVolumeFractions (viscositysolver.cpp:180-270) for one lattice of dims (w,h,d) whose sample
// centre is centerStart + cellCentre(i,j,k)
__global__ void k_volume_lattice(const float *__restrict__ phi, const uint8_t *__restrict__ valid,
                                 float *__restrict__ vol, int w, int h, int d, float csx, float csy, float csz, int I,
                                 int J, int K, float dxf) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    float out = 0.0f;
    if (valid[DIDX(i, j, k, I + 1, J + 1)]) {
        const double dx = (double)dxf, invdx = 1.0 / dx, hw = 0.5 * dx;
        const float hdx = 0.5f * dxf;          // viscositysolver.cpp:188
        const float hoff = (float)(0.5 * dx);  // particlelevelset.cpp:89
        const float cx = csx + (float)(i * dx + hw), cy = csy + (float)(j * dx + hw), cz = csz + (float)(k * dx + hw);
        float p[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {  // q = 4*oi + 2*oj + ok
            const float sx = cx + ((q & 4) ? hdx : -hdx), sy = cy + ((q & 2) ? hdx : -hdx), sz = cz + ((q & 1) ? hdx : -hdx);
            p[q] = d_liquid_phi_at(sx, sy, sz, dx, invdx, hoff, phi, I, J, K);
        }
        const float p000 = p[0], p001 = p[1], p010 = p[2], p011 = p[3], p100 = p[4], p101 = p[5], p110 = p[6], p111 = p[7];
        int neg = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) neg += p[q] < 0.0f;
        if (neg == 8) out = 1.0f;
        else if (neg == 0) out = 0.0f;
        else out = d_cube_fraction(p000, p100, p010, p110, p001, p101, p011, p111);
    }
    vol[DIDX(i, j, k, w, h)] = out;
}

// ------------------------------------------------------------------ factors
// f = dt/dx^2 * nu * volume on the four coefficient lattices (viscositysolver.cpp:394-427 and the V/W
// analogues :492-525, :590-623).  nu is node-sampled; the edge lattices use the 4-node mean.
#define NU(i, j, k) visc[DIDX(i, j, k, I + 1, J + 1)]
__global__ void k_visc_factors(const float *__restrict__ visc, const float *__restrict__ volC,
                               const float *__restrict__ volEU, const float *__restrict__ volEV,
                               const float *__restrict__ volEW, float *__restrict__ fC, float *__restrict__ fEU,
                               float *__restrict__ fEV, float *__restrict__ fEW, int I, int J, int K, float factor) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i > I || j > J) return;
    if (i < I && j < J && k < K) {
        const size_t c = DIDX(i, j, k, I, J);
        fC[c] = 2 * factor * NU(i, j, k) * volC[c];
    }
    if (i < I && k <= K) {  // edgeU (I,J+1,K+1): nodes (i,j-1..j,k-1..k)
        float f = 0.0f;
        if (j >= 1 && k >= 1)
            f = factor * (0.25f * (NU(i, j - 1, k) + NU(i, j - 1, k - 1) + NU(i, j, k) + NU(i, j, k - 1))) *
                volEU[DIDX(i, j, k, I, J + 1)];
        fEU[DIDX(i, j, k, I, J + 1)] = f;
    }
    if (j < J && k <= K) {  // edgeV (I+1,J,K+1): nodes (i-1..i,j,k-1..k)
        float f = 0.0f;
        if (i >= 1 && k >= 1)
            f = factor * (0.25f * (NU(i - 1, j, k) + NU(i - 1, j, k - 1) + NU(i, j, k) + NU(i, j, k - 1))) *
                volEV[DIDX(i, j, k, I + 1, J)];
        fEV[DIDX(i, j, k, I + 1, J)] = f;
    }
    if (k < K) {  // edgeW (I+1,J+1,K): nodes (i-1..i,j-1..j,k)
        float f = 0.0f;
        if (i >= 1 && j >= 1)
            f = factor * (0.25f * (NU(i - 1, j, k) + NU(i - 1, j - 1, k) + NU(i, j, k) + NU(i, j - 1, k))) *
                volEW[DIDX(i, j, k, I + 1, J + 1)];
        fEW[DIDX(i, j, k, I + 1, J + 1)] = f;
    }
}

// ------------------------------------------------------------------ the coupled stencil
// Coefficients of the rows at index (i,j,k).  FC/FEU/FEV/FEW index the factor lattices.
#define FC(i, j, k) fC[DIDX(i, j, k, I, J)]
#define FEU(i, j, k) fEU[DIDX(i, j, k, I, J + 1)]
#define FEV(i, j, k) fEV[DIDX(i, j, k, I + 1, J)]
#define FEW(i, j, k) fEW[DIDX(i, j, k, I + 1, J + 1)]
#define XU(i, j, k) xu[DIDX(i, j, k, I + 1, J)]
#define XV(i, j, k) xv[DIDX(i, j, k, I, J + 1)]
#define XW(i, j, k) xw[DIDX(i, j, k, I, J)]

// row eligibility: the reference loops 1 <= i < I, 1 <= j < J, 1 <= k < K for all three components
// (viscositysolver.cpp:284-354).  A row whose stencil would leave the arrays (j = J-1 or k = K-1 for U,
// etc.) makes the reference throw std::out_of_range; with any closed solid boundary those faces are SOLID
// and never rows.  They are excluded here so the kernels need no bounds checks.
__device__ __forceinline__ bool d_row_range(int dir, int i, int j, int k, int I, int J, int K) {
    if (i < 1 || j < 1 || k < 1) return false;
    if (dir == 0) return i <= I - 1 && j <= J - 2 && k <= K - 2;
    if (dir == 1) return i <= I - 2 && j <= J - 1 && k <= K - 2;
    return i <= I - 2 && j <= J - 2 && k <= K - 1;
}

// K8: diagonal + right-hand side + row selection, one thread per index (i,j,k) of the (I+1,J+1,K+1) space.
// rhs follows viscositysolver.cpp:448-465 (and :546-563, :644-659): own volume * velocity minus the
// couplings to SOLID-state neighbours, accumulated in fp32 in the reference's order.
template <typename T>
__global__ void k_visc_setup(const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ W,
                             const uint8_t *__restrict__ stU, const uint8_t *__restrict__ stV,
                             const uint8_t *__restrict__ stW, const float *__restrict__ volU,
                             const float *__restrict__ volV, const float *__restrict__ volW,
                             const float *__restrict__ volC, const float *__restrict__ volEU,
                             const float *__restrict__ volEV, const float *__restrict__ volEW,
                             const float *__restrict__ fC, const float *__restrict__ fEU,
                             const float *__restrict__ fEV, const float *__restrict__ fEW, float *__restrict__ dgU,
                             float *__restrict__ dgV, float *__restrict__ dgW, PcgVecs<T> v, double *__restrict__ bmax,
                             int *__restrict__ nrows, int I, int J, int K) {
    __shared__ double lds[4];
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    double babs = 0.0;
    int rows = 0;
#define SU(i, j, k) stU[DIDX(i, j, k, I + 1, J)]
#define SV(i, j, k) stV[DIDX(i, j, k, I, J + 1)]
#define SW(i, j, k) stW[DIDX(i, j, k, I, J)]
#define VELU(i, j, k) U[DIDX(i, j, k, I + 1, J)]
#define VELV(i, j, k) V[DIDX(i, j, k, I, J + 1)]
#define VELW(i, j, k) W[DIDX(i, j, k, I, J)]
#define VC(i, j, k) volC[DIDX(i, j, k, I, J)]
#define VEU(i, j, k) volEU[DIDX(i, j, k, I, J + 1)]
#define VEV(i, j, k) volEV[DIDX(i, j, k, I + 1, J)]
#define VEW(i, j, k) volEW[DIDX(i, j, k, I + 1, J + 1)]
#define RHS(st, vel, coef) do { if ((st) == ST_SOLID) rval -= (coef) * (vel); } while (0)
    if (i <= I && j < J && k < K) {  // ---- U face
        const size_t f = DIDX(i, j, k, I + 1, J);
        float dg = 0.0f, rval = 0.0f;
        if (d_row_range(0, i, j, k, I, J, K) && SU(i, j, k) == ST_FLUID) {
            const float vol = volU[f];
            if (vol > 0.0f || VC(i, j, k) > 0.0f || VC(i - 1, j, k) > 0.0f || VEW(i, j + 1, k) > 0.0f ||
                VEW(i, j, k) > 0.0f || VEV(i, j, k + 1) > 0.0f || VEV(i, j, k) > 0.0f) {
                const float fR = FC(i, j, k), fL = FC(i - 1, j, k), fT = FEW(i, j + 1, k), fB = FEW(i, j, k),
                            fF = FEV(i, j, k + 1), fK = FEV(i, j, k);
                dg = vol + fR + fL + fT + fB + fF + fK;
                rval = vol * VELU(i, j, k);
                RHS(SU(i + 1, j, k), VELU(i + 1, j, k), -fR);
                RHS(SU(i - 1, j, k), VELU(i - 1, j, k), -fL);
                RHS(SU(i, j + 1, k), VELU(i, j + 1, k), -fT);
                RHS(SU(i, j - 1, k), VELU(i, j - 1, k), -fB);
                RHS(SU(i, j, k + 1), VELU(i, j, k + 1), -fF);
                RHS(SU(i, j, k - 1), VELU(i, j, k - 1), -fK);
                RHS(SV(i, j + 1, k), VELV(i, j + 1, k), -fT);
                RHS(SV(i - 1, j + 1, k), VELV(i - 1, j + 1, k), fT);
                RHS(SV(i, j, k), VELV(i, j, k), fB);
                RHS(SV(i - 1, j, k), VELV(i - 1, j, k), -fB);
                RHS(SW(i, j, k + 1), VELW(i, j, k + 1), -fF);
                RHS(SW(i - 1, j, k + 1), VELW(i - 1, j, k + 1), fF);
                RHS(SW(i, j, k), VELW(i, j, k), fK);
                RHS(SW(i - 1, j, k), VELW(i - 1, j, k), -fK);
                if (dg == 0.0f) rval = 0.0f;
                rows += dg != 0.0f;
            }
        }
        dgU[f] = dg;
        v.r[0][f] = (T)rval; v.x[0][f] = (T)0; v.s[0][f] = (T)0;
        babs = fmax(babs, fabs((double)rval));
    }
    if (i < I && j <= J && k < K) {  // ---- V face
        const size_t f = DIDX(i, j, k, I, J + 1);
        float dg = 0.0f, rval = 0.0f;
        if (d_row_range(1, i, j, k, I, J, K) && SV(i, j, k) == ST_FLUID) {
            const float vol = volV[f];
            if (vol > 0.0f || VEW(i + 1, j, k) > 0.0f || VEW(i, j, k) > 0.0f || VC(i, j, k) > 0.0f ||
                VC(i, j - 1, k) > 0.0f || VEU(i, j, k + 1) > 0.0f || VEU(i, j, k) > 0.0f) {
                const float fR = FEW(i + 1, j, k), fL = FEW(i, j, k), fT = FC(i, j, k), fB = FC(i, j - 1, k),
                            fF = FEU(i, j, k + 1), fK = FEU(i, j, k);
                dg = vol + fR + fL + fT + fB + fF + fK;
                rval = vol * VELV(i, j, k);
                RHS(SV(i + 1, j, k), VELV(i + 1, j, k), -fR);
                RHS(SV(i - 1, j, k), VELV(i - 1, j, k), -fL);
                RHS(SV(i, j + 1, k), VELV(i, j + 1, k), -fT);
                RHS(SV(i, j - 1, k), VELV(i, j - 1, k), -fB);
                RHS(SV(i, j, k + 1), VELV(i, j, k + 1), -fF);
                RHS(SV(i, j, k - 1), VELV(i, j, k - 1), -fK);
                RHS(SU(i + 1, j, k), VELU(i + 1, j, k), -fR);
                RHS(SU(i + 1, j - 1, k), VELU(i + 1, j - 1, k), fR);
                RHS(SU(i, j, k), VELU(i, j, k), fL);
                RHS(SU(i, j - 1, k), VELU(i, j - 1, k), -fL);
                RHS(SW(i, j, k + 1), VELW(i, j, k + 1), -fF);
                RHS(SW(i, j - 1, k + 1), VELW(i, j - 1, k + 1), fF);
                RHS(SW(i, j, k), VELW(i, j, k), fK);
                RHS(SW(i, j - 1, k), VELW(i, j - 1, k), -fK);
                if (dg == 0.0f) rval = 0.0f;
                rows += dg != 0.0f;
            }
        }
        dgV[f] = dg;
        v.r[1][f] = (T)rval; v.x[1][f] = (T)0; v.s[1][f] = (T)0;
        babs = fmax(babs, fabs((double)rval));
    }
    if (i < I && j < J && k <= K) {  // ---- W face
        const size_t f = DIDX(i, j, k, I, J);
        float dg = 0.0f, rval = 0.0f;
        if (d_row_range(2, i, j, k, I, J, K) && SW(i, j, k) == ST_FLUID) {
            const float vol = volW[f];
            if (vol > 0.0f || VEV(i + 1, j, k) > 0.0f || VEV(i, j, k) > 0.0f || VEU(i, j + 1, k) > 0.0f ||
                VEU(i, j, k) > 0.0f || VC(i, j, k) > 0.0f || VC(i, j, k - 1) > 0.0f) {
                const float fR = FEV(i + 1, j, k), fL = FEV(i, j, k), fT = FEU(i, j + 1, k), fB = FEU(i, j, k),
                            fF = FC(i, j, k), fK = FC(i, j, k - 1);
                dg = vol + fR + fL + fT + fB + fF + fK;
                rval = vol * VELW(i, j, k);
                RHS(SW(i + 1, j, k), VELW(i + 1, j, k), -fR);
                RHS(SW(i - 1, j, k), VELW(i - 1, j, k), -fL);
                RHS(SW(i, j + 1, k), VELW(i, j + 1, k), -fT);
                RHS(SW(i, j - 1, k), VELW(i, j - 1, k), -fB);
                RHS(SW(i, j, k + 1), VELW(i, j, k + 1), -fF);
                RHS(SW(i, j, k - 1), VELW(i, j, k - 1), -fK);
                RHS(SU(i + 1, j, k), VELU(i + 1, j, k), -fR);
                RHS(SU(i + 1, j, k - 1), VELU(i + 1, j, k - 1), fR);
                RHS(SU(i, j, k), VELU(i, j, k), fL);
                RHS(SU(i, j, k - 1), VELU(i, j, k - 1), -fL);
                RHS(SV(i, j + 1, k), VELV(i, j + 1, k), -fT);
                RHS(SV(i, j + 1, k - 1), VELV(i, j + 1, k - 1), fT);
                RHS(SV(i, j, k), VELV(i, j, k), fB);
                RHS(SV(i, j, k - 1), VELV(i, j, k - 1), -fB);
                if (dg == 0.0f) rval = 0.0f;
                rows += dg != 0.0f;
            }
        }
        dgW[f] = dg;
        v.r[2][f] = (T)rval; v.x[2][f] = (T)0; v.s[2][f] = (T)0;
        babs = fmax(babs, fabs((double)rval));
    }
    const double bm = block_max_256(babs, lds);
    const double nr = block_sum_256((double)rows, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        if (bm > 0.0) atomic_max_nonneg(bmax, bm);
        if (nr > 0.0) atomicAdd(nrows, (int)nr);
    }
}

// K9 SpMV: z = A s over the active tiles, fused s.z.  Row = index whose diagonal is non-zero; x is 0 on
// every other face so neighbour values need no masks (coupling signs: SURVEY.md A.6b).
template <typename T>
__global__ __launch_bounds__(256) void k_visc_spmv(const int *__restrict__ tiles, int ntiles, TileGrid tg,
                                                   const float *__restrict__ dgU, const float *__restrict__ dgV,
                                                   const float *__restrict__ dgW, const float *__restrict__ fC,
                                                   const float *__restrict__ fEU, const float *__restrict__ fEV,
                                                   const float *__restrict__ fEW, const T *__restrict__ xu,
                                                   const T *__restrict__ xv, const T *__restrict__ xw,
                                                   T *__restrict__ yu, T *__restrict__ yv, T *__restrict__ yw, int I,
                                                   int J, int K, double *__restrict__ dA, const int *__restrict__ conv) {
    if (conv && *conv >= 0) return;
    __shared__ double lds[4];
    const int slot = d_tile_slot(blockIdx.x, ntiles);
    double acc = 0.0;
    if (slot < ntiles) {
        int i, j, k0;
        d_tile_coords(tiles[slot], tg, i, j, k0);
        const int kend = min(k0 + TZ, K + 1);
        for (int k = k0; k < kend; k++) {
            if (i <= I && j < J && k < K) {
                const size_t f = DIDX(i, j, k, I + 1, J);
                const float dg = dgU[f];
                T y = (T)0;
                if (dg != 0.0f) {
                    const T fR = (T)FC(i, j, k), fL = (T)FC(i - 1, j, k), fT = (T)FEW(i, j + 1, k), fB = (T)FEW(i, j, k),
                            fF = (T)FEV(i, j, k + 1), fK = (T)FEV(i, j, k);
                    const T xc = XU(i, j, k);
                    y = (T)dg * xc - fR * XU(i + 1, j, k) - fL * XU(i - 1, j, k) - fT * XU(i, j + 1, k) -
                        fB * XU(i, j - 1, k) - fF * XU(i, j, k + 1) - fK * XU(i, j, k - 1);
                    y += fT * (XV(i - 1, j + 1, k) - XV(i, j + 1, k)) + fB * (XV(i, j, k) - XV(i - 1, j, k));
                    y += fF * (XW(i - 1, j, k + 1) - XW(i, j, k + 1)) + fK * (XW(i, j, k) - XW(i - 1, j, k));
                    acc += (double)xc * (double)y;
                }
                yu[f] = y;
            }
            if (i < I && j <= J && k < K) {
                const size_t f = DIDX(i, j, k, I, J + 1);
                const float dg = dgV[f];
                T y = (T)0;
                if (dg != 0.0f) {
                    const T fR = (T)FEW(i + 1, j, k), fL = (T)FEW(i, j, k), fT = (T)FC(i, j, k), fB = (T)FC(i, j - 1, k),
                            fF = (T)FEU(i, j, k + 1), fK = (T)FEU(i, j, k);
                    const T xc = XV(i, j, k);
                    y = (T)dg * xc - fR * XV(i + 1, j, k) - fL * XV(i - 1, j, k) - fT * XV(i, j + 1, k) -
                        fB * XV(i, j - 1, k) - fF * XV(i, j, k + 1) - fK * XV(i, j, k - 1);
                    y += fR * (XU(i + 1, j - 1, k) - XU(i + 1, j, k)) + fL * (XU(i, j, k) - XU(i, j - 1, k));
                    y += fF * (XW(i, j - 1, k + 1) - XW(i, j, k + 1)) + fK * (XW(i, j, k) - XW(i, j - 1, k));
                    acc += (double)xc * (double)y;
                }
                yv[f] = y;
            }
            if (i < I && j < J && k <= K) {
                const size_t f = DIDX(i, j, k, I, J);
                const float dg = dgW[f];
                T y = (T)0;
                if (dg != 0.0f) {
                    const T fR = (T)FEV(i + 1, j, k), fL = (T)FEV(i, j, k), fT = (T)FEU(i, j + 1, k), fB = (T)FEU(i, j, k),
                            fF = (T)FC(i, j, k), fK = (T)FC(i, j, k - 1);
                    const T xc = XW(i, j, k);
                    y = (T)dg * xc - fR * XW(i + 1, j, k) - fL * XW(i - 1, j, k) - fT * XW(i, j + 1, k) -
                        fB * XW(i, j - 1, k) - fF * XW(i, j, k + 1) - fK * XW(i, j, k - 1);
                    y += fR * (XU(i + 1, j, k - 1) - XU(i + 1, j, k)) + fL * (XU(i, j, k) - XU(i, j, k - 1));
                    y += fT * (XV(i, j + 1, k - 1) - XV(i, j + 1, k)) + fB * (XV(i, j, k) - XV(i, j, k - 1));
                    acc += (double)xc * (double)y;
                }
                yw[f] = y;
            }
        }
    }
    const double tot = block_sum_256(acc, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && tot != 0.0 && dA) atomicAdd(dA, tot);
}

template <typename T>
__global__ void k_vec_to_f32(const T *__restrict__ a, float *__restrict__ o, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) o[t] = (float)a[t];
}

// ------------------------------------------------------------------------------------------------
template <typename T>
static void launch_visc_spmv(flipv_context *c, double *dA, const int *conv) {
    const Dims &d = c->d;
    const int nb = ((c->nActiveV + 7) / 8) * 8;
    if (c->prm.kernel_timing) fv_ev_begin(c, 1, (double)c->nActiveV * TX * TY * TZ);
    hipLaunchKernelGGL(k_visc_spmv<T>, dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListV, c->nActiveV, c->tg,
                       c->vDiagU, c->vDiagV, c->vDiagW, c->fC, c->fEU, c->fEV, c->fEW, (const T *)c->vS[0],
                       (const T *)c->vS[1], (const T *)c->vS[2], (T *)c->vZ[0], (T *)c->vZ[1], (T *)c->vZ[2], d.I, d.J,
                       d.K, dA, conv);
    if (c->prm.kernel_timing) fv_ev_end(c);
}

template <typename T>
static int viscosity_solve_t(flipv_context *c, float dt, flipv_solve_info *info) {
    const Dims &d = c->d;
    const int I = d.I, J = d.J, K = d.K;
    flipv_solve_info li;
    memset(&li, 0, sizeof(li));
    li.total_tiles = c->tg.count();
    if (!c->viscosity_nonzero) {  // fluidsimulation.cpp:171-184
        li.status = 3;
        if (info) *info = li;
        return FLIPV_OK;
    }
    const int cap = c->prm.viscosity_max_iterations;
    int rc = fv_scal_reserve(c, cap);
    if (rc) return rc;
    const size_t nscal = (size_t)3 * (cap + 2) + 16;
    HIPCHK(c, hipMemsetAsync(c->d_scal, 0, nscal * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0xff, 4 * sizeof(int), c->stream));  // conv = -1
    HIPCHK(c, hipMemsetAsync(c->d_flags + 2, 0, sizeof(int), c->stream));     // row counter
    PcgScal sc;
    sc.sigma = c->d_scal;
    sc.dA = c->d_scal + (cap + 2);
    sc.rmax = c->d_scal + 2 * (cap + 2);
    double *bmax = c->d_scal + 3 * (cap + 2);
    sc.conv = c->d_flags;
    sc.tol_inclusive = 1;

    // face states
    hipLaunchKernelGGL(k_solid_center, GRID3(I, J, K), 0, c->stream, c->solid, c->scp, I, J, K);
    uint8_t *st[3] = {c->stU, c->stV, c->stW};
    for (int dir = 0; dir < 3; dir++)
        hipLaunchKernelGGL(k_face_states, GRID3(I + (dir == 0), J + (dir == 1), K + (dir == 2)), 0, c->stream, dir, c->scp,
                           st[dir], I, J, K);
    // band mask + the seven volume lattices (viscositysolver.cpp:135-178)
    hipLaunchKernelGGL(k_valid_init, GRID3(I + 1, J + 1, K + 1), 0, c->stream, c->phi, c->validCells, I, J, K);
    hipLaunchKernelGGL(k_valid_dilate, GRID3(I + 1, J + 1, K + 1), 0, c->stream, c->validCells, c->validTmp, I + 1, J + 1, K + 1);
    hipLaunchKernelGGL(k_valid_dilate, GRID3(I + 1, J + 1, K + 1), 0, c->stream, c->validTmp, c->validCells, I + 1, J + 1, K + 1);
    const float h = (float)(0.5 * c->dx);
    struct { float *v; int w, hh, dd; float cx, cy, cz; } lat[7] = {
        {c->volC, I, J, K, h, h, h},         {c->volU, I + 1, J, K, 0, h, h},     {c->volV, I, J + 1, K, h, 0, h},
        {c->volW, I, J, K + 1, h, h, 0},     {c->volEU, I, J + 1, K + 1, h, 0, 0}, {c->volEV, I + 1, J, K + 1, 0, h, 0},
        {c->volEW, I + 1, J + 1, K, 0, 0, h}};
    for (int q = 0; q < 7; q++)
        hipLaunchKernelGGL(k_volume_lattice, GRID3(lat[q].w, lat[q].hh, lat[q].dd), 0, c->stream, c->phi, c->validCells,
                           lat[q].v, lat[q].w, lat[q].hh, lat[q].dd, lat[q].cx, lat[q].cy, lat[q].cz, I, J, K, c->dx);
    const float invdx = 1.0f / c->dx;
    const float factor = dt * invdx * invdx;  // viscositysolver.cpp:379-380
    hipLaunchKernelGGL(k_visc_factors, GRID3(I + 1, J + 1, K + 1), 0, c->stream, c->visc, c->volC, c->volEU, c->volEV,
                       c->volEW, c->fC, c->fEU, c->fEV, c->fEW, I, J, K, factor);
    PcgVecs<T> v;
    for (int q = 0; q < 3; q++) { v.x[q] = (T *)c->vX[q]; v.r[q] = (T *)c->vR[q]; v.z[q] = (T *)c->vZ[q]; v.s[q] = (T *)c->vS[q]; }
    hipLaunchKernelGGL(k_visc_setup<T>, GRID3(I + 1, J + 1, K + 1), 0, c->stream, c->U, c->V, c->W, c->stU, c->stV, c->stW,
                       c->volU, c->volV, c->volW, c->volC, c->volEU, c->volEV, c->volEW, c->fC, c->fEU, c->fEV, c->fEW,
                       c->vDiagU, c->vDiagV, c->vDiagW, v, bmax, c->d_flags + 2, I, J, K);
    HIPCHK(c, hipMemcpyAsync(c->h_scal, bmax, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_flags + 2, c->d_flags + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PcgComps cp;
    memset(&cp, 0, sizeof(cp));
    cp.n = 3;
    cp.w[0] = I + 1; cp.h[0] = J; cp.d[0] = K; cp.diag[0] = c->vDiagU;
    cp.w[1] = I; cp.h[1] = J + 1; cp.d[1] = K; cp.diag[1] = c->vDiagV;
    cp.w[2] = I; cp.h[2] = J; cp.d[2] = K + 1; cp.diag[2] = c->vDiagW;
    rc = fv_build_tiles(c, cp, c->tileListV, &c->nActiveV);
    if (rc) return rc;
    const double bnorm = c->h_scal[0];
    li.rhs_norm = bnorm;
    li.rows = c->h_flags[2];
    li.active_tiles = c->nActiveV;
    c->viscosityReady = 1;
    c->viscosityPrec = std::is_same<T, float>::value ? 0 : 1;

    int conv = -1, iters = 0;
    double res = bnorm;
    bool success = false;
    if (bnorm == 0.0 || c->nActiveV == 0) {  // pcgsolver.h:254-258: zero rhs -> zero solution, success
        success = true;
    } else {
        sc.tol = c->prm.viscosity_tolerance * bnorm;
        const int nb = ((c->nActiveV + 7) / 8) * 8;
        const dim3 blk(64, 4, 1);
        hipLaunchKernelGGL(k_pcg_init<T>, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tg, cp, v, sc);
        const int every = c->prm.check_every > 0 ? c->prm.check_every : 8;
        int it = 0;
        while (it < cap && conv < 0) {
            const int stop = (it + every < cap) ? it + every : cap;
            for (; it < stop; it++) {
                launch_visc_spmv<T>(c, sc.dA + it, sc.conv);
                hipLaunchKernelGGL(k_pcg_update<T>, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tg, cp, v, sc, it);
                hipLaunchKernelGGL(k_pcg_dir<T>, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tg, cp, v, sc, it);
            }
            HIPCHK(c, hipMemcpyAsync(c->h_flags, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            conv = c->h_flags[0];
        }
        const int last = conv >= 0 ? conv : cap - 1;
        HIPCHK(c, hipMemcpyAsync(c->h_scal, sc.rmax + last, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        res = c->h_scal[0];
        iters = conv >= 0 ? conv + 1 : cap;
        success = conv >= 0;
    }
    li.iterations = iters;
    li.residual = res;
    // acceptance rule of viscositysolver.cpp:676-689
    const bool accepted = success || (iters == cap && res < c->prm.viscosity_accept_tolerance);
    li.status = success ? (iters == 0 ? 3 : 0) : (accepted ? 1 : 2);
    if (accepted) {  // _applySolutionToVelocityField (viscositysolver.cpp:692-727): x is 0 off the rows
        hipLaunchKernelGGL(k_vec_to_f32<T>, dim3(2048), dim3(256), 0, c->stream, (const T *)v.x[0], c->U, d.nu());
        hipLaunchKernelGGL(k_vec_to_f32<T>, dim3(2048), dim3(256), 0, c->stream, (const T *)v.x[1], c->V, d.nv());
        hipLaunchKernelGGL(k_vec_to_f32<T>, dim3(2048), dim3(256), 0, c->stream, (const T *)v.x[2], c->W, d.nw());
    }
    HIPCHK(c, hipGetLastError());
    if (c->prm.kernel_timing) fv_ev_collect(c);
    if (info) *info = li;
    return success ? FLIPV_OK : (accepted ? FLIPV_WARN_NOT_CONVERGED : FLIPV_WARN_SOLVE_FAILED);
}

int fv_viscosity_solve(flipv_context *c, float dt, flipv_solve_info *info) {
    if (c->prm.precision == FLIPV_PRECISION_FP64) return viscosity_solve_t<double>(c, dt, info);
    return viscosity_solve_t<float>(c, dt, info);
}

int fv_bench_viscosity_spmv(flipv_context *c, int reps, double *ms, double *cells) {
    if (!c->viscosityReady || c->nActiveV <= 0) { c->err = "flipv_bench_spmv: run flipv_viscosity_solve first"; return FLIPV_ERR_INVALID; }
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    const int saved = c->prm.kernel_timing;
    c->prm.kernel_timing = 0;
    for (int w = 0; w < 3; w++) {
        if (c->viscosityPrec) launch_visc_spmv<double>(c, nullptr, nullptr); else launch_visc_spmv<float>(c, nullptr, nullptr);
    }
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; r++) {
        if (c->viscosityPrec) launch_visc_spmv<double>(c, nullptr, nullptr); else launch_visc_spmv<float>(c, nullptr, nullptr);
    }
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipEventSynchronize(b));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, a, b));
    c->prm.kernel_timing = saved;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms = (double)t / reps;
    *cells = (double)c->nActiveV * TX * TY * TZ;
    return FLIPV_OK;
}
