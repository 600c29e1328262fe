// k_viscosity.hip -- variational (Batty-Bridson) viscosity solve, matrix-free.
// Reference: ViscositySolver::applyViscosityToVelocityField (viscositysolver.cpp:41-727) on top of the
// generic CSR PCGSolver<double> (pcgsolver/pcgsolver.h).
//
// The reference assembles a double CSR matrix with <= 15 non-zeros per row (~180 B per row) and runs
// MIC(0)-PCG on it.  Here the operator is never assembled: every coefficient of a row is one of six
// "factors" f = dt/dx^2 * nu * volume that live on four lattices (cell centres and the three edge
// families), so the coupled 15-point SpMV over the U, V and W rows of one index (i,j,k) reads
//   3 own volumes + 4 factor arrays + 3 x (+ 3 fp64 r for the fused beta dots) and writes 3 q
//   = 52 B (+24 B) per swept index in fp32, against ~540 B for the assembled form.
// Unknown faces are exactly the faces with a non-zero diagonal; x is kept 0 everywhere else, which reproduces
// the reference's silent drop of couplings to faces without a matrix row (sparsematrix.h:86-88).
#include "flipv_internal.h"
#include "pcg_common.h"
#include "flipv_comm.h"

enum { ST_FLUID = 1, ST_SOLID = 2 };

// ------------------------------------------------------------------ face states
// viscositysolver.cpp:80-133
__global__ void k_solid_center(Lay L, const float *__restrict__ solid, float *__restrict__ scp) {
    IJK_OR_RETURN(L);
    if (i < L.I && j < L.J && k < L.K) scp[c] = d_solid_center(solid, L, c);
}

__global__ void k_face_states(Lay L, const float *__restrict__ scp, uint8_t *__restrict__ stU,
                              uint8_t *__restrict__ stV, uint8_t *__restrict__ stW) {
    IJK_OR_RETURN(L);
    uint8_t *st[3] = {stU, stV, stW};
    const long back[3] = {1, L.sy, L.sz};
#pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        int w, h, d;
        lat_dims(L, LAT_U + dir, w, h, d);
        if (i >= w || j >= h || k >= d) continue;
        const int n = dir == 0 ? L.I : (dir == 1 ? L.J : L.K);
        const int cd = dir == 0 ? i : (dir == 1 ? j : k);
        bool solid = cd == 0 || cd == n;
        if (!solid) solid = scp[c - back[dir]] + scp[c] <= 0.0f;
        st[dir][c] = solid ? ST_SOLID : ST_FLUID;
    }
}

// ------------------------------------------------------------------ band mask
// viscositysolver.cpp:138-168: phi<0 cells on an (I+1,J+1,K+1) mask, then two 6-neighbour dilations
__global__ void k_valid_init(Lay L, const float *__restrict__ phi, uint8_t *__restrict__ m) {
    IJK_OR_RETURN(L);
    uint8_t v = 0;
    if (i < L.I && j < L.J && k < L.K) v = phi[c] < 0.0f;
    m[c] = v;
}
__global__ void k_valid_dilate(Lay L, const uint8_t *__restrict__ a, uint8_t *__restrict__ b) {
    IJK_OR_RETURN(L);
    uint8_t v = 0;
    if (i <= L.I && j <= L.J && k <= L.K) {
        v = a[c];
        if (i > 0) v |= a[c - 1];
        if (i < L.I) v |= a[c + 1];
        if (j > 0) v |= a[c - L.sy];
        if (j < L.J) v |= a[c + L.sy];
        if (k > 0) v |= a[c - L.sz];
        if (k < L.K) v |= a[c + L.sz];
    }
    b[c] = v;
}

// ------------------------------------------------------------------ control volumes
// liquid phi sampled like ParticleLevelSet::trilinearInterpolate (particlelevelset.cpp:88-92 ->
// interpolation.cpp:68-108): float position, fp64 weights, out-of-range corners = 0
__device__ __forceinline__ float d_liquid_phi_at(float px, float py, float pz, double dx, double invdx, float hdx,
                                                 const float *__restrict__ phi, const Lay &L) {
    px -= hdx; py -= hdx; pz -= hdx;
    const int gi = (int)floor((double)px * invdx), gj = (int)floor((double)py * invdx), gk = (int)floor((double)pz * invdx);
    const float gx = (float)(gi * dx), gy = (float)(gj * dx), gz = (float)(gk * dx);
    const double ix = (px - gx) * invdx, iy = (py - gy) * invdx, iz = (pz - gz) * invdx;
    // corner order of interpolation.cpp:54-66: 000,100,010,001,101,011,110,111; the two i-neighbours of a (j, k) row in one 8-byte gather
    float f[8];
    d_corner_pair(phi, L, gi, gj, gk, L.I, L.J, L.K, f[0], f[1]);
    d_corner_pair(phi, L, gi, gj + 1, gk, L.I, L.J, L.K, f[2], f[6]);
    d_corner_pair(phi, L, gi, gj, gk + 1, L.I, L.J, L.K, f[3], f[4]);
    d_corner_pair(phi, L, gi, gj + 1, gk + 1, L.I, L.J, L.K, f[5], f[7]);
    double p[8];
#pragma unroll
    for (int q = 0; q < 8; q++) p[q] = (double)f[q];
    return (float)(p[0] * (1 - ix) * (1 - iy) * (1 - iz) + p[1] * ix * (1 - iy) * (1 - iz) + p[2] * (1 - ix) * iy * (1 - iz) +
                   p[3] * (1 - ix) * (1 - iy) * iz + p[4] * ix * (1 - iy) * iz + p[5] * (1 - ix) * iy * iz +
                   p[6] * ix * iy * (1 - iz) + p[7] * ix * iy * iz);
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

// _estimateVolumeFractions (viscositysolver.cpp:180-270) for lattice `lat` whose sample centre is
// centerStart + cellCentre(i,j,k)
struct VolLattices { float *vol[7]; int lat[7]; float cs[7][3]; };
// Pass 1, all seven lattices in one sweep.  Every corner sample of the seven cubes of index (i,j,k) interpolates liquid
// phi of cells i-1..i+2 (x j-1..j+2 x k-1..k+2) with weights in [0,1]: if all 64 are negative every corner is (all
// eight negative -> 1, viscositysolver.cpp:254-259), if none is no corner is (-> 0).  Only indices at the liquid surface
// need the sampling path; they are appended to a list so that pass 2 runs it with full waves (a surface crosses
// nearly every 64-wide row of the band, which made one fused kernel pay the sampling path for every band wave).
__global__ void k_volume_classify(Lay L, VolLattices Q, const float *__restrict__ phi, const uint8_t *__restrict__ valid,
                                  const uint8_t *__restrict__ prevband, int full, unsigned *__restrict__ list,
                                  unsigned *__restrict__ nlist) {
    IJK_OF_THREAD(L);
    const bool inside = i < L.ie && j < L.je;
    const size_t c = inside ? gidx(L, i, j, k) : 0;
    float out = 0.0f;
    bool sample = false;
    if (inside && valid[c]) {
        sample = true;
        if (i >= 1 && j >= 1 && k >= 1 && i + 2 < L.I && j + 2 < L.J && k + 2 < L.K) {
            int nn = 0;
            for (int dk = -1; dk <= 2; dk++)
                for (int dj = -1; dj <= 2; dj++) {
                    const FloatQuad row = *reinterpret_cast<const FloatQuad *>(phi + gidx(L, i - 1, j + dj, k + dk));   // one 16-byte gather (4-byte aligned)
                    nn += (row.a < 0.0f) + (row.b < 0.0f) + (row.c < 0.0f) + (row.d < 0.0f);
                }
            if (nn == 64) { out = 1.0f; sample = false; }
            else if (nn == 0) { out = 0.0f; sample = false; }
        }
    }
    // the volumes are zero off the band and stay zero between solves: store only where the band is or was
    if (inside && (full || valid[c] || prevband[c])) {
#pragma unroll
        for (int m = 0; m < 7; m++) {
            int w, h, d;
            lat_dims(L, Q.lat[m], w, h, d);
            if (i < w && j < h && k < d) Q.vol[m][c] = out;  // surface indices are overwritten by pass 2
        }
    }
    // wave-aggregated append
    const unsigned long long m = __ballot(sample);
    if (m) {
        const int lane = threadIdx.x & 63;
        const int lead = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == lead) base = atomicAdd(nlist, (unsigned)__popcll(m));
        base = __shfl(base, lead, 64);
        if (sample) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)c;   // position inside the allocated box
    }
}

// Pass 2: _estimateVolumeFractions (viscositysolver.cpp:180-270) for the listed indices, one thread per (index, lattice)
__global__ void k_volume_sample(Lay L, VolLattices Q, const float *__restrict__ phi, const unsigned *__restrict__ list,
                                const unsigned *__restrict__ nlist, float dxf) {
    const unsigned n = *nlist;
    const double dx = (double)dxf, invdx = 1.0 / dx, hw = 0.5 * dx;
    const float hdx = 0.5f * dxf;          // viscositysolver.cpp:188
    const float hoff = (float)(0.5 * dx);  // particlelevelset.cpp:89
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < (size_t)n * 7; t += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(t / n);  // lattice-major: a wave works on one lattice
        const unsigned id = list[t - (size_t)m * n];
        const int i = L.ox + (int)(id % (unsigned)L.PX), j = L.oy + (int)((id / (unsigned)L.PX) % (unsigned)L.PY), k = L.oz + (int)(id / ((unsigned)L.PX * (unsigned)L.PY));
        int w, h, d;
        lat_dims(L, Q.lat[m], w, h, d);
        if (i >= w || j >= h || k >= d) continue;
        const float cx = Q.cs[m][0] + (float)(i * dx + hw), cy = Q.cs[m][1] + (float)(j * dx + hw), cz = Q.cs[m][2] + (float)(k * dx + hw);
        float p[8];
        int neg = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {  // q = 4*oi + 2*oj + ok
            const float sx = cx + ((q & 4) ? hdx : -hdx), sy = cy + ((q & 2) ? hdx : -hdx), sz = cz + ((q & 1) ? hdx : -hdx);
            p[q] = d_liquid_phi_at(sx, sy, sz, dx, invdx, hoff, phi, L);
            neg += p[q] < 0.0f;
        }
        float out;
        if (neg == 8) out = 1.0f;
        else if (neg == 0) out = 0.0f;
        else out = d_cube_fraction(p[0], p[4], p[2], p[6], p[1], p[5], p[3], p[7]);  // 000,100,010,110,001,101,011,111
        Q.vol[m][gidx(L, i, j, k)] = out;
    }
}

// ------------------------------------------------------------------ factors
// f = dt/dx^2 * nu * volume on the four coefficient lattices (viscositysolver.cpp:394-427 and the V/W
// analogues :492-525, :590-623).  nu is node-sampled; the edge lattices use the 4-node mean.
__global__ void k_visc_factors(Lay L, const float *__restrict__ nu, const float *__restrict__ volC,
                               const float *__restrict__ volEU, const float *__restrict__ volEV,
                               const float *__restrict__ volEW, float *__restrict__ fC, float *__restrict__ fEU,
                               float *__restrict__ fEV, float *__restrict__ fEW, float factor, const uint8_t *__restrict__ band,
                               const uint8_t *__restrict__ prevband, int full) {
    IJK_OR_RETURN(L);
    if (!full && !band[c] && !prevband[c]) return;  // a factor is a volume of the same index times viscosity: zero off the band, and it stays zero
    const long sy = L.sy, sz = L.sz;
    const int I = L.I, J = L.J, K = L.K;
    if (i > I || j > J || k > K) return;
    if (i < I && j < J && k < K) fC[c] = 2 * factor * nu[c] * volC[c];
    if (i < I) {  // edgeU (I,J+1,K+1): nodes (i, j-1..j, k-1..k)
        float f = 0.0f;
        if (j >= 1 && k >= 1) f = factor * (0.25f * (nu[c - sy] + nu[c - sy - sz] + nu[c] + nu[c - sz])) * volEU[c];
        fEU[c] = f;
    }
    if (j < J) {  // edgeV (I+1,J,K+1): nodes (i-1..i, j, k-1..k)
        float f = 0.0f;
        if (i >= 1 && k >= 1) f = factor * (0.25f * (nu[c - 1] + nu[c - 1 - sz] + nu[c] + nu[c - sz])) * volEV[c];
        fEV[c] = f;
    }
    if (k < K) {  // edgeW (I+1,J+1,K): nodes (i-1..i, j-1..j, k)
        float f = 0.0f;
        if (i >= 1 && j >= 1) f = factor * (0.25f * (nu[c - 1] + nu[c - 1 - sy] + nu[c] + nu[c - sy])) * volEW[c];
        fEW[c] = f;
    }
}

// ------------------------------------------------------------------ K8: diagonal + rhs + row selection
// row eligibility: the reference loops 1 <= i < I, 1 <= j < J, 1 <= k < K for all three components
// (viscositysolver.cpp:284-354).  A row whose stencil would leave the arrays (j = J-1 or k = K-1 for U,
// etc.) makes the reference throw std::out_of_range; with any closed solid boundary those faces are SOLID
// and never rows.  They are excluded here.
__device__ __forceinline__ bool d_row_range(int dir, int i, int j, int k, const Lay &L) {
    if (i < 1 || j < 1 || k < 1) return false;
    if (dir == 0) return i <= L.I - 1 && j <= L.J - 2 && k <= L.K - 2;
    if (dir == 1) return i <= L.I - 2 && j <= L.J - 1 && k <= L.K - 2;
    return i <= L.I - 2 && j <= L.J - 2 && k <= L.K - 1;
}

// rhs follows viscositysolver.cpp:448-465 (and :546-563, :644-659): own volume * velocity minus the
// couplings to SOLID-state neighbours, accumulated in fp32 in the reference's order.
#define RHS(st, vel, coef) do { if ((st) == ST_SOLID) rval -= (coef) * (vel); } while (0)
// The reference stores its matrix in float and forms a row's diagonal as the float sum vol + fR + fL + fT + fB + fF + fK
// (viscositysolver.cpp:394-446), so ITS matrix is the exact one plus a rounding defect on the diagonal, up to ~3 ulp of a diagonal
// that is nu dt/dx^2 ~ 10^3-10^4 times the volume term.  On a rigid motion the stress terms cancel and only the volume term is left
// of a row, so that defect is a relative change of up to ~1e-3 of what the row does to such a field, and of 1.5e-4 / 2e-4 in the
// converged velocities of the 256^3 bunny scene (measured against the reference run to 1e-8: every variant of the exact operator --
// either preconditioner, fp32 or fp64 vectors -- agrees with every other to 2e-6 and differs from the reference by that much).  The
// difference-form SpMV applies the exact operator vol*u - div(tau); storing vol + defect as the row's "own volume" makes it apply the
// reference's: the defect is fl(diagonal) minus the exact sum of the same seven floats (exact in fp64).  Values stay >= -0.02; the
// "no row" marker is -1.  Opt-in (FLIPV_REF_DIAG=1, see viscosity_solve_t): the reference's operator is visibly worse conditioned than
// the exact one (its own MIC(0) PCG needs 7 689 iterations for 1e-6 and 42 223 for 1e-8 at 256^3).
__device__ __forceinline__ float d_ref_volume(float vol, float fR, float fL, float fT, float fB, float fF, float fK, float dgf) {
    const double exact = (double)vol + (double)fR + (double)fL + (double)fT + (double)fB + (double)fF + (double)fK;
    return (float)((double)vol + ((double)dgf - exact));
}
template <typename T>
__global__ void k_visc_setup(Lay L, const float *__restrict__ U, const float *__restrict__ V,
                             const float *__restrict__ W, const uint8_t *__restrict__ SU,
                             const uint8_t *__restrict__ SV, const uint8_t *__restrict__ SW,
                             const float *__restrict__ volU, const float *__restrict__ volV,
                             const float *__restrict__ volW, const float *__restrict__ VC,
                             const float *__restrict__ VEU, const float *__restrict__ VEV,
                             const float *__restrict__ VEW, const float *__restrict__ fC,
                             const float *__restrict__ fEU, const float *__restrict__ fEV,
                             const float *__restrict__ fEW, float *__restrict__ dgU, float *__restrict__ dgV,
                             float *__restrict__ dgW, float *__restrict__ vmU, float *__restrict__ vmV,
                             float *__restrict__ vmW, float *__restrict__ vrU, float *__restrict__ vrV, float *__restrict__ vrW,
                             uint8_t *__restrict__ rowmask, const uint8_t *__restrict__ band,
                             int full, PcgSys<T, 3> v, double *__restrict__ bmax, int *__restrict__ nrows, int refdiag) {   // v.swz: layout of diag, vm, r, x
    __shared__ double lds[4];
    IJK_OF_THREAD(L);
    double babs = 0.0;
    int rows = 0;
    if (i < L.ie && j < L.je) {
        const size_t c = gidx(L, i, j, k);
        const long sy = L.sy, sz = L.sz;
        float dg[3] = {0.0f, 0.0f, 0.0f}, rv[3] = {0.0f, 0.0f, 0.0f}, vm[3] = {-1.0f, -1.0f, -1.0f}, vr[3] = {-1.0f, -1.0f, -1.0f};
        // every control volume is zero off the band mask (k_volume_lattice), and a row needs a non-zero volume at its own
        // index or at an index one step down/up an axis: no band there, no row here
        const bool near = band[c] || band[c - 1] || band[c + 1] || band[c - sy] || band[c + sy] || band[c - sz] || band[c + sz];
        const uint8_t prev = rowmask[c];
        if (!near && !prev && !full) goto done;  // no row now, none in the previous solve: every array already holds its off-row value
        if (!near) goto store;
        if (d_row_range(0, i, j, k, L) && SU[c] == ST_FLUID) {  // ---- U face (viscositysolver.cpp:374-470)
            const float vol = volU[c];
            if (vol > 0.0f || VC[c] > 0.0f || VC[c - 1] > 0.0f || VEW[c + sy] > 0.0f || VEW[c] > 0.0f || VEV[c + sz] > 0.0f ||
                VEV[c] > 0.0f) {
                const float fR = fC[c], fL = fC[c - 1], fT = fEW[c + sy], fB = fEW[c], fF = fEV[c + sz], fK = fEV[c];
                float rval = vol * U[c];
                RHS(SU[c + 1], U[c + 1], -fR);
                RHS(SU[c - 1], U[c - 1], -fL);
                RHS(SU[c + sy], U[c + sy], -fT);
                RHS(SU[c - sy], U[c - sy], -fB);
                RHS(SU[c + sz], U[c + sz], -fF);
                RHS(SU[c - sz], U[c - sz], -fK);
                RHS(SV[c + sy], V[c + sy], -fT);
                RHS(SV[c - 1 + sy], V[c - 1 + sy], fT);
                RHS(SV[c], V[c], fB);
                RHS(SV[c - 1], V[c - 1], -fB);
                RHS(SW[c + sz], W[c + sz], -fF);
                RHS(SW[c - 1 + sz], W[c - 1 + sz], fF);
                RHS(SW[c], W[c], fK);
                RHS(SW[c - 1], W[c - 1], -fK);
                dg[0] = vol + fR + fL + fT + fB + fF + fK;
                rv[0] = dg[0] != 0.0f ? rval : 0.0f;
                if (dg[0] != 0.0f) { vm[0] = vol; vr[0] = refdiag ? d_ref_volume(vol, fR, fL, fT, fB, fF, fK, dg[0]) : vol; }
            }
        }
        if (d_row_range(1, i, j, k, L) && SV[c] == ST_FLUID) {  // ---- V face (viscositysolver.cpp:472-568)
            const float vol = volV[c];
            if (vol > 0.0f || VEW[c + 1] > 0.0f || VEW[c] > 0.0f || VC[c] > 0.0f || VC[c - sy] > 0.0f || VEU[c + sz] > 0.0f ||
                VEU[c] > 0.0f) {
                const float fR = fEW[c + 1], fL = fEW[c], fT = fC[c], fB = fC[c - sy], fF = fEU[c + sz], fK = fEU[c];
                float rval = vol * V[c];
                RHS(SV[c + 1], V[c + 1], -fR);
                RHS(SV[c - 1], V[c - 1], -fL);
                RHS(SV[c + sy], V[c + sy], -fT);
                RHS(SV[c - sy], V[c - sy], -fB);
                RHS(SV[c + sz], V[c + sz], -fF);
                RHS(SV[c - sz], V[c - sz], -fK);
                RHS(SU[c + 1], U[c + 1], -fR);
                RHS(SU[c + 1 - sy], U[c + 1 - sy], fR);
                RHS(SU[c], U[c], fL);
                RHS(SU[c - sy], U[c - sy], -fL);
                RHS(SW[c + sz], W[c + sz], -fF);
                RHS(SW[c - sy + sz], W[c - sy + sz], fF);
                RHS(SW[c], W[c], fK);
                RHS(SW[c - sy], W[c - sy], -fK);
                dg[1] = vol + fR + fL + fT + fB + fF + fK;
                rv[1] = dg[1] != 0.0f ? rval : 0.0f;
                if (dg[1] != 0.0f) { vm[1] = vol; vr[1] = refdiag ? d_ref_volume(vol, fR, fL, fT, fB, fF, fK, dg[1]) : vol; }
            }
        }
        if (d_row_range(2, i, j, k, L) && SW[c] == ST_FLUID) {  // ---- W face (viscositysolver.cpp:570-664)
            const float vol = volW[c];
            if (vol > 0.0f || VEV[c + 1] > 0.0f || VEV[c] > 0.0f || VEU[c + sy] > 0.0f || VEU[c] > 0.0f || VC[c] > 0.0f ||
                VC[c - sz] > 0.0f) {
                const float fR = fEV[c + 1], fL = fEV[c], fT = fEU[c + sy], fB = fEU[c], fF = fC[c], fK = fC[c - sz];
                float rval = vol * W[c];
                RHS(SW[c + 1], W[c + 1], -fR);
                RHS(SW[c - 1], W[c - 1], -fL);
                RHS(SW[c + sy], W[c + sy], -fT);
                RHS(SW[c - sy], W[c - sy], -fB);
                RHS(SW[c + sz], W[c + sz], -fF);
                RHS(SW[c - sz], W[c - sz], -fK);
                RHS(SU[c + 1], U[c + 1], -fR);
                RHS(SU[c + 1 - sz], U[c + 1 - sz], fR);
                RHS(SU[c], U[c], fL);
                RHS(SU[c - sz], U[c - sz], -fL);
                RHS(SV[c + sy], V[c + sy], -fT);
                RHS(SV[c + sy - sz], V[c + sy - sz], fT);
                RHS(SV[c], V[c], fB);
                RHS(SV[c - sz], V[c - sz], -fB);
                dg[2] = vol + fR + fL + fT + fB + fF + fK;
                rv[2] = dg[2] != 0.0f ? rval : 0.0f;
                if (dg[2] != 0.0f) { vm[2] = vol; vr[2] = refdiag ? d_ref_volume(vol, fR, fL, fT, fB, fF, fK, dg[2]) : vol; }
            }
        }
    store:
        {
            const uint8_t now = (uint8_t)((dg[0] != 0.0f) | ((dg[1] != 0.0f) << 1) | ((dg[2] != 0.0f) << 2));
            if (now || prev || full) {  // off-row values (diag 0, volume -1, x = s = 0) persist between solves where nothing was a row
                const size_t cs = v.swz ? sidx(L, i, j, k) : c;
                dgU[cs] = dg[0]; dgV[cs] = dg[1]; dgW[cs] = dg[2];
                vmU[cs] = vm[0]; vmV[cs] = vm[1]; vmW[cs] = vm[2];
                vrU[cs] = vr[0]; vrV[cs] = vr[1]; vrW[cs] = vr[2];
                rowmask[c] = now;
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    v.r[m][cs] = (RT<T>)rv[m]; v.x[m][cs] = (T)0; v.s[m][c] = (T)0;
                    babs = fmax(babs, fabs((double)rv[m]));
                    rows += dg[m] != 0.0f;
                }
            }
        }
    done:;
    }
    const double bm = block_max_256(babs, lds);
    const double nr = block_sum_256((double)rows, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        if (bm > 0.0) atomic_max_nonneg(bmax, bm);
        if (nr > 0.0) atomicAdd(nrows, (int)nr);
    }
}

// ---- the SpMV kernel (and the shared PCG kernels it is paired with) depend on the tile geometry: once per geometry
namespace g16 {
constexpr int ROWL = 16;
#include "pcg_geo.inc"
#include "k_viscosity_geo.inc"
}  // namespace g16
namespace g64 {
constexpr int ROWL = 64;
#include "pcg_geo.inc"
#include "k_viscosity_geo.inc"
}  // namespace g64

template <typename T>
static __global__ void k_vec_to_f32(const T *__restrict__ a, float *__restrict__ o, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) o[t] = (float)a[t];
}

// x -> velocity grid over a launch box (plain layout)
template <typename T>
static __global__ void k_box_to_f32(Lay L, const T *__restrict__ a, float *__restrict__ o) {
    IJK_OR_RETURN(L);
    o[c] = (float)a[c];
}

// x in the swizzled plane layout -> velocity grid
template <typename T>
static __global__ void k_unswizzle_to_f32(Lay L, const T *__restrict__ a, float *__restrict__ o) {
    IJK_OF_THREAD(L);
    if (i < L.ie && j < L.je) o[gidx(L, i, j, k)] = (float)a[sidx(L, i, j, k)];
}

// ------------------------------------------------------------------------------------------------
template <typename T>
static PcgSys<T, 3> visc_sys(flipv_context *c) {
    PcgSys<T, 3> v;
    v.swz = c->vSwz;
    v.mask = c->vRowMask;
    v.mlist = c->mlistV;
    v.diag[0] = c->vDiagU; v.diag[1] = c->vDiagV; v.diag[2] = c->vDiagW;
    for (int m = 0; m < 3; m++) { v.x[m] = (T *)c->vX[m]; v.r[m] = (RT<T> *)c->vR[m]; v.q[m] = (T *)c->vZ[m]; v.s[m] = (T *)c->vS[m]; }
    return v;
}

int fv_viscosity_pcg_mg(flipv_context *c, const PcgScal &sc, int cap, void (*spmv)(flipv_context *, const PcgScal &, int), int *conv_out);
int fv_vmg_prepare(flipv_context *c);

template <typename T, int NV>
static void launch_visc_spmv(flipv_context *c, const PcgScal &sc, int it, int first, int count, const PcgSys<T, 3> *sys = nullptr) {
    // one resident round: the 4-wide kernel holds 189 VGPRs = 2 waves per SIMD = 2 blocks per CU = 512 blocks (measured over
    // 512..1024 at 256^3: 37.2 ms per solve at 512, 38.1 at 1024, 40.8-43.6 in between); flipv_params.viscosity_spmv_grid_cap overrides
    int nb = pcg_grid(c, count);
    const int cap = c->prm.viscosity_spmv_grid_cap > 0 ? c->prm.viscosity_spmv_grid_cap : (NV == 4 ? 512 : 1024);
    if (nb > cap) nb = cap;
    const bool timed = c->prm.kernel_timing && (it & 7) == 0 && first == 0;  // HIP events around every 8th launch
    if (timed) fv_ev_begin(c, 1, (double)count * (256 * NV));
    PcgSys<T, 3> vv = sys ? *sys : visc_sys<T>(c);
    if (vv.mlist) vv.mlist += (size_t)first * 256;   // the mask words are in list order
    // which operator: the reference's (own volumes + the rounding defect of its float diagonal, d_ref_volume) under the diagonal
    // preconditioner; the exact one under the multigrid, whose fp32 recursion bottoms out at a relative residual of 2e-5 against the
    // former in the stiff start of the 256^3 scene (the defect changes near-rigid modes of small liquid clusters by O(1) relative to
    // what the hierarchy, built from the exact rows, expects)
    const float *const vo[3] = {c->vOperatorExact ? c->vmU : c->vrU, c->vOperatorExact ? c->vmV : c->vrV, c->vOperatorExact ? c->vmW : c->vrW};
#define VSPMV(N_, P_, R_) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_visc_spmv<T, N_, P_, R_>), dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListV + first, count, c->tgV, c->L, \
                           vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, vv, sc, it))
    const bool rdot = sc.conv ? !sc.noB : c->prm.beta_from_residual != 0;   // benchmark launches (no scalars): the variant the solve would run
    if (NV == 4 && c->nRunsV > 0 && first == 0 && count == c->nActiveV) {   // k-marching over the run list (the whole system)
        int nbm = pcg_grid(c, c->nRunsV);
        if (nbm > cap) nbm = cap;
#define VMARCH(P_, R_) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_visc_spmv_march<T, P_, R_>), dim3(nbm), dim3(64, 4, 1), 0, c->stream, (const Run *)c->runsV, c->nRunsV, \
                           (const unsigned *)(c->vPred ? c->rmaskV : nullptr), c->tgV, c->L, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, vv, sc, it))
        if (c->vPred) { if (rdot) VMARCH(true, true); else VMARCH(true, false); }
        else { if (rdot) VMARCH(false, true); else VMARCH(false, false); }
#undef VMARCH
        if (timed) fv_ev_end(c);
        return;
    }
    if (NV == 4 && c->vPred) { if (rdot) VSPMV(4, true, true); else VSPMV(4, true, false); }
    else { if (rdot) VSPMV(NV, NV == 2, true); else VSPMV(NV, NV == 2, false); }
#undef VSPMV
    if (timed) fv_ev_end(c);
}

// flipv_params.viscosity_preconditioner = AUTO: the diagonal or the multigrid V-cycle, whichever the previous solve says will be
// cheaper.  Costs in units of one diagonal-preconditioned iteration (SpMV + update, 41 us at 256^3): a multigrid iteration ~7.5 (5
// fine SpMV-class launches + the coarse levels, 300 us), its set-up ~35 (1.3 ms: Galerkin gathers, lists, graph capture, the
// cycle before the first iteration); one multigrid iteration does the work of ~15 diagonal ones (10-30 measured on the bench scene,
// DESIGN.md 3).  The diagonal solve stops at the cap whether converged or not
// (the reference's budget); the multigrid is only worth starting when it is predicted to CONVERGE for less than that:
//   after a diagonal solve that converged in n iterations        -> multigrid next if 35 + 7.5 n/15 < n         (n > ~80)
//   after a diagonal solve stopped at the cap with residual rho  -> n is extrapolated, n = cap ln(tol)/ln(rho); multigrid if 35 + 7.5 n/15 < cap
//   after a multigrid solve of m iterations                      -> stay while 35 + 7.5 m < min(15 m, cap)      (5 <= m <= 88 at the stock cap)
// with 10 % hysteresis.  Decisions depend on iteration counts only, never on wall-clock times, so a run is reproducible.
// On the bench scene the first dozen substeps (the bunny at rest, nu dt/dx^2 = 3 300: 120-250 multigrid iterations) stay with the
// capped diagonal solve like the reference; once the liquid moves (dt shrinks, 15-60 iterations) the multigrid takes over and
// every solve converges: 25-30 ms per substep against 31-36.  FLIPV_VISC_AUTO=0: AUTO = always the diagonal.
static bool fv_visc_auto_pick(const flipv_context *c) {
    static const bool off = getenv("FLIPV_VISC_AUTO") && atoi(getenv("FLIPV_VISC_AUTO")) == 0;
    if (off || c->vLastPrec == 0) return false;
    const double cap = (double)c->prm.viscosity_max_iterations, tol = c->prm.viscosity_tolerance > 0 ? c->prm.viscosity_tolerance : 1e-6;
    const double MG_ITER = 7.5, MG_SETUP = 35.0, RATIO = 15.0;
    if (c->vLastPrec == 1) {
        double n = (double)c->vLastIts;
        if (!c->vLastConverged) {
            const double rho = c->vLastRelRes;
            if (!(rho > 0.0) || rho >= 1.0) return false;
            n = n * log(tol) / log(rho);
        }
        const double costD = n < cap ? n : cap, costM = MG_SETUP + MG_ITER * n / RATIO;
        return costM < 0.9 * costD;
    }
    if (!c->vLastConverged) return false;   // a multigrid solve that stalled or ran into the cap: back to the diagonal
    const double m = (double)c->vLastIts;
    const double costM = MG_SETUP + MG_ITER * m, costD = RATIO * m < cap ? RATIO * m : cap;
    return costM < 1.1 * costD;
}

template <typename T>
static int viscosity_solve_t(flipv_context *c, float dt, flipv_solve_info *info) {
    const Lay &L = c->L;
    flipv_solve_info li;
    memset(&li, 0, sizeof(li));
    if (!c->viscosity_nonzero) {  // fluidsimulation.cpp:171-184
        li.status = 3;
        if (info) *info = li;
        return FLIPV_OK;
    }
    const int cap = c->prm.viscosity_max_iterations;
    int rc = fv_scal_reserve(c, cap);
    if (rc) return rc;
    const size_t nscal = (size_t)5 * (cap + 2) * NSLOT + 16;
    HIPCHK(c, hipMemsetAsync(c->d_scal, 0, nscal * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0xff, sizeof(int), c->stream));   // conv = -1
    HIPCHK(c, hipMemsetAsync(c->d_flags + 2, 0, sizeof(int), c->stream));  // row counter
    PcgScal sc;
    double *bmax;
    fv_scal_views(c, cap, &sc, &bmax);
    sc.tol_inclusive = 1;
    sc.tol = 0.0;

    // Everything up to the factors is evaluated redundantly on the halo planes a neighbour-owned row would need
    // (inputs: phi with a 4-plane halo, the replicated solid SDF), so the setup needs no exchange of its own.
    // face states
    if (c->faceStateVersion != c->solidVersion) {  // functions of the solid SDF only: everywhere
        const Lay F1 = fv_range(c, 1), F2 = fv_range(c, 2);
        hipLaunchKernelGGL(k_solid_center, GRID3(F2), 0, c->stream, F2, c->solid, c->scp);
        hipLaunchKernelGGL(k_face_states, GRID3(F1), 0, c->stream, F1, c->scp, c->stU, c->stV, c->stW);
        c->faceStateVersion = c->solidVersion;
    }
    const Lay R0 = fv_range_liquid(c, 0, 4), R1 = fv_range_liquid(c, 1, 4), R2 = fv_range_liquid(c, 2, 4), R3 = fv_range_liquid(c, 3, 4);
    // band mask + the seven volume lattices (viscositysolver.cpp:135-178)
    hipLaunchKernelGGL(k_valid_init, GRID3(R3), 0, c->stream, R3, c->phi, c->validCells);
    hipLaunchKernelGGL(k_valid_dilate, GRID3(R2), 0, c->stream, R2, c->validCells, c->validTmp);
    hipLaunchKernelGGL(k_valid_dilate, GRID3(R1), 0, c->stream, R1, c->validTmp, c->validCells);
    const float h = (float)(0.5 * c->dx);
    const int fullVol = c->bandPrevValid ? 0 : 1;  // volumes and factors are stored only where the band is or was in the previous solve
    {
        VolLattices Q;
        float *vols[7] = {c->volC, c->volU, c->volV, c->volW, c->volEU, c->volEV, c->volEW};
        const int lats[7] = {LAT_CELL, LAT_U, LAT_V, LAT_W, LAT_EU, LAT_EV, LAT_EW};
        const float cs[7][3] = {{h, h, h}, {0, h, h}, {h, 0, h}, {h, h, 0}, {h, 0, 0}, {0, h, 0}, {0, 0, h}};  // viscositysolver.cpp:171-177
        for (int q = 0; q < 7; q++) { Q.vol[q] = vols[q]; Q.lat[q] = lats[q]; Q.cs[q][0] = cs[q][0]; Q.cs[q][1] = cs[q][1]; Q.cs[q][2] = cs[q][2]; }
        if (!c->surfList) {
            hipError_t e = hipMalloc((void **)&c->surfList, (L.n + 64) * sizeof(unsigned));
            if (e != hipSuccess) { c->err = std::string("hipMalloc(surface list): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        }
        unsigned *nlist = c->surfList + L.n;
        HIPCHK(c, hipMemsetAsync(nlist, 0, sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(k_volume_classify, GRID3(R1), 0, c->stream, R1, Q, c->phi, c->validCells, c->bandPrev, fullVol, c->surfList, nlist);
        hipLaunchKernelGGL(k_volume_sample, dim3(4096), dim3(256), 0, c->stream, c->L, Q, c->phi, c->surfList, nlist, c->dx);
    }
    const float invdx = 1.0f / c->dx;
    const float factor = dt * invdx * invdx;  // viscositysolver.cpp:379-380
    hipLaunchKernelGGL(k_visc_factors, GRID3(R1), 0, c->stream, R1, c->visc, c->volC, c->volEU, c->volEV, c->volEW, c->fC,
                       c->fEU, c->fEV, c->fEW, factor, c->validCells, c->bandPrev, fullVol);
    {
        const size_t off = plane_off(L, R1.kb), cnt = (size_t)(R1.ke - R1.kb) * L.sz;
        HIPCHK(c, hipMemcpyAsync(c->bandPrev + off, c->validCells + off, cnt, hipMemcpyDeviceToDevice, c->stream));
        c->bandPrevValid = 1;
    }
    // Layout of x, r, q and the diagonal -- the arrays that are only ever read at a lane's own indices: swizzled 8 x 4
    // patches (sidx) go with the 16-lane tile geometry; s, which the SpMV reads with its halo (and slabs exchange), stays
    // plain, and so do the multigrid's sweep vectors (za, zb, t0: each is the next sweep's input).  The geometry is only known once the tiles are built,
    // so the setup kernel runs in the layout of the previous solve's geometry and is repeated on the rare solve where
    // the geometry changes.
    // the preconditioner of this solve (the multigrid needs fp32 vectors over a whole, single-rank index space)
    const bool mgPossible = std::is_same<T, float>::value && !c->comm && !c->isBlock && c->prm.viscosity_lane_width != 2 && !c->vNoMultigridOnce;
    const bool mgPlanned = mgPossible && (c->prm.viscosity_preconditioner == FLIPV_PRECOND_MULTIGRID ||
                                          (c->prm.viscosity_preconditioner == FLIPV_PRECOND_AUTO && fv_visc_auto_pick(c)));
    const bool swzOk = c->allowSwz;   // (also under the multigrid: its own kernels address diag / x / r / q / own volumes through sidx, its sweep vectors stay plain)
    if (mgPossible && c->prm.viscosity_preconditioner == FLIPV_PRECOND_AUTO && !c->vmgState && !(getenv("FLIPV_VISC_AUTO") && atoi(getenv("FLIPV_VISC_AUTO")) == 0)) {
        const int prc = fv_vmg_prepare(c);   // AUTO may pick the multigrid later in the run: allocate its hierarchy now, not in that substep
        if (prc) return prc;
    }
    const int precNow = std::is_same<T, float>::value ? 0 : 1;
    // flipv_params.reference_diagonal = 1 (or FLIPV_REF_DIAG=1): the diagonally preconditioned solve applies the reference's operator INCLUDING the rounding of its float
    // diagonal (d_ref_volume) -- bit-faithful parity at sizes where that rounding shows (256^3: 7e-6 instead of 1.45e-4 against the
    // reference's converged answer) at the price of the reference's conditioning: fp32 solves then sit closer to their attainable
    // accuracy (tight tolerances stall more often).  Default: the exact operator.  Read per solve (tests toggle it).
    const char *refEnv = getenv("FLIPV_REF_DIAG");
    const int refDiag = refEnv ? (atoi(refEnv) != 0) : (c->prm.reference_diagonal != 0);   // flipv_params.reference_diagonal; the environment overrides
    auto run_setup = [&](int swz) -> int {
        // the setup kernel only stores where a row is or was; the first solve, a change of vector precision (the buffers
        // are shared), of the layout or of the slab make it store everywhere
        const int full = (c->viscStateValid && c->viscStatePrec == precNow && c->vSwz == swz) ? 0 : 1;
        c->viscStateValid = 1; c->viscStatePrec = precNow; c->vSwz = swz;
        PcgSys<T, 3> vs = visc_sys<T>(c);
        HIPCHK(c, hipMemsetAsync(bmax, 0, sizeof(double), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_flags + 2, 0, sizeof(int), c->stream));
        // rows of the owned planes only (the SpMV reads diag / own volume at its own index, factors and s at +-1 plane); a change
        // of layout or precision rewrites every entry, not only those near the liquid
        const Lay RS = full ? fv_range(c, 0) : R0;
        hipLaunchKernelGGL(k_visc_setup<T>, GRID3(RS), 0, c->stream, RS, c->U, c->V, c->W, c->stU, c->stV, c->stW, c->volU, c->volV,
                           c->volW, c->volC, c->volEU, c->volEV, c->volEW, c->fC, c->fEU, c->fEV, c->fEW, c->vDiagU, c->vDiagV,
                           c->vDiagW, c->vmU, c->vmV, c->vmW, c->vrU, c->vrV, c->vrW, c->vRowMask, c->validCells, full, vs, bmax, c->d_flags + 2, refDiag);
        HIPCHK(c, hipMemcpyAsync(c->h_scal, bmax, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_flags + 2, c->d_flags + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return FLIPV_OK;
    };
    if ((rc = run_setup(swzOk && (c->forceRowl ? c->forceRowl : c->tgV.rowl) == 16 ? 1 : 0))) return rc;
    // Lane width of the solver kernels: 4 consecutive i per lane (16-byte accesses).  With per-lane load predication the
    // narrow variant (2 per lane, twice the waves) no longer wins on sparse liquids (256^3 bunny: 40.7 vs 42.7 ms per
    // solve); it stays selectable for measurements.  Sparse liquids (row fill <= 0.35) use the predicated SpMV.
    HIPCHK(c, hipStreamSynchronize(c->stream));  // h_flags[2] = row count
    const double fill = (double)c->h_flags[2] / (3.0 * (double)(L.ohi[0] - L.olo[0]) * (double)(L.ohi[1] - L.olo[1]) * (double)(L.ohi[2] - L.olo[2]));
    c->vwV = 4;
    if (c->prm.viscosity_lane_width == 2 || c->prm.viscosity_lane_width == 4) c->vwV = c->prm.viscosity_lane_width;  // measurement switch: forced lane width
    c->vPred = fill <= 0.35;
    rc = fv_build_tiles(c, &c->tgV, c->vwV, 3, c->vDiagU, c->vDiagV, c->vDiagW, c->vRowMask, c->tileListV, &c->nActiveV, &c->nIntV, c->h_flags + 2, 3, &c->mlistV, &c->mlistCapV);
    if (rc) return rc;
    if ((rc = fv_build_runs(c, c->tgV, c->vwV, c->nActiveV, !c->vPred, c->vRowMask, &c->runsV, &c->runCapV, &c->nRunsV, &c->runLenV, &c->rmaskV, &c->rmaskCapV))) return rc;
    if (c->vSwz != (swzOk && c->tgV.rowl == 16 ? 1 : 0)) {  // the geometry changed: the vectors go into the other layout
        if ((rc = run_setup(1 - c->vSwz))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    PcgSys<T, 3> v = visc_sys<T>(c);
    if (c->comm) {
        float bn = (float)c->h_scal[0];
        if ((rc = fv_allreduce_max_f32(c, &bn))) return rc;
        c->h_scal[0] = (double)bn;
    }
    const double bnorm = c->h_scal[0];
    li.rhs_norm = bnorm;
    li.rows = c->h_flags[2];
    li.active_tiles = c->nActiveV;
    li.total_tiles = c->tgV.count();
    c->viscosityReady = 1;
    c->viscosityPrec = std::is_same<T, float>::value ? 0 : 1;

    int conv = -1, iters = 0;
    double res = bnorm;
    bool success = false, stalled = false, ranMg = false;
    int anyActive = c->nActiveV;
    if (c->comm) { float f = (float)anyActive; if ((rc = fv_allreduce_max_f32(c, &f))) return rc; anyActive = (int)f; }
    if (bnorm == 0.0 || anyActive == 0) {  // pcgsolver.h:254-258: zero rhs -> zero solution, success
        success = true;
    } else {
        sc.tol = c->prm.viscosity_tolerance * bnorm;
        int nb = pcg_grid(c, c->nActiveV);
        if (c->prm.viscosity_update_grid_cap > 0) { nb = ((c->nActiveV + 7) / 8) * 8; if (nb > c->prm.viscosity_update_grid_cap) nb = c->prm.viscosity_update_grid_cap; if (nb < 8) nb = 8; }  // measurement switch: grid cap of init/update
        const dim3 blk(64, 4, 1);
        const HaloArray sh[3] = {{c->vS[0], sizeof(T)}, {c->vS[1], sizeof(T)}, {c->vS[2], sizeof(T)}};
        const bool useMg = mgPlanned && c->vwV == 4;
        li.preconditioner = useMg ? 1 : 0;
        ranMg = useMg;
        c->vOperatorExact = useMg ? 1 : 0;
        if (useMg) {
            if ((rc = fv_viscosity_pcg_mg(c, sc, cap, [](flipv_context *cc, const PcgScal &s2, int it) { launch_visc_spmv<float, 4>(cc, s2, it, 0, cc->nActiveV); },
                                          &conv)))
                return rc;
        } else {
        if (c->vwV == 4)
            GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_pcg_init<T, 3, 4>), dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, L, v, sc));
        else
            GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_pcg_init<T, 3, 2>), dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, L, v, sc));
        if ((rc = fv_allreduce_scalars(c, sc.sig(0), NSLOT))) return rc;
        auto spmv = [&](int first, int count, int it) {
            if (c->vwV == 4) launch_visc_spmv<T, 4>(c, sc, it, first, count); else launch_visc_spmv<T, 2>(c, sc, it, first, count);
        };
        auto update = [&](int it) {
            if (c->vwV == 4)
                GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_pcg_update<T, 3, 4>), dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, L, v, sc, it));
            else
                GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_pcg_update<T, 3, 2>), dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, L, v, sc, it));
        };
        if ((rc = pcg_run(c, sc, cap, sh, 3, c->nIntV, c->nActiveV, spmv, update, &conv, FV_GE_VISCOSITY))) return rc;
        }
        const int last = conv >= 0 ? conv : cap - 1;
        hipLaunchKernelGGL(k_pcg_residual, dim3(1), dim3(64), 0, c->stream, sc, last, bmax);
        HIPCHK(c, hipMemcpyAsync(c->h_scal, bmax, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        res = c->h_scal[0];
        iters = conv >= 0 ? conv + 1 : cap;
        success = conv >= 0;
        if (success) {   // the stall guard (PcgScal) stops the loop through the same flag: that is not convergence
            int st = 0;
            HIPCHK(c, hipMemcpy(&st, sc.stalled, sizeof(int), hipMemcpyDeviceToHost));
            if (st) { success = false; stalled = true; }
        }
    }
    // (A multigrid-preconditioned fp32 solve that STALLS two orders of magnitude or more below the right-hand side -- seen in the stiff
    // start of the 256^3 scene, where the recursion bottoms out at a relative residual of 2e-5 against the reference's float-rounded
    // operator -- keeps its iterate: it is far closer to the solution than 700 diagonal iterations get, and is reported as "not
    // converged" like any accepted iterate.)
    if (ranMg && !success && !(stalled && res < 1e-4 * bnorm)) {
        // The multigrid-preconditioned solve did not reach the tolerance (never seen with a hierarchy assembled for this very system;
        // a stale one -- FLIPV_VMG_KEEP > 1 after a change of dt -- over-corrects and breaks PCG down).  Its iterate is not used:
        // the solve is repeated from scratch with the diagonal, whose capped iterate is what the reference's acceptance rule is about.
        if (getenv("FLIPV_VMG_DEBUG")) fprintf(stderr, "multigrid-preconditioned solve failed: %d iterations, residual %.3g (rhs %.3g), stalled %d; repeating with the diagonal\n", iters, res, bnorm, (int)stalled);
        c->viscSolves++;
        c->vLastPrec = 2; c->vLastIts = iters; c->vLastConverged = 0; c->vLastRelRes = bnorm > 0.0 ? res / bnorm : 0.0;
        c->vNoMultigridOnce = 1;
        const int rc2 = viscosity_solve_t<T>(c, dt, info);
        c->vNoMultigridOnce = 0;
        return rc2;
    }
    li.iterations = iters;
    li.residual = res;
    c->viscSolves++;
    c->vLastPrec = (bnorm == 0.0 || anyActive == 0) ? c->vLastPrec : (li.preconditioner ? 2 : 1);   // (a trivial solve says nothing)
    if (!(bnorm == 0.0 || anyActive == 0)) { c->vLastIts = iters; c->vLastConverged = success ? 1 : 0; c->vLastRelRes = bnorm > 0.0 ? res / bnorm : 0.0; }
    // acceptance rule of viscositysolver.cpp:676-689
    // (a stalled solve is treated like one that ran into the cap: its iterate is used if the residual passes the acceptance bound)
    const bool accepted = success || ((iters == cap || stalled) && res < c->prm.viscosity_accept_tolerance);
    li.status = success ? (iters == 0 ? 3 : 0) : (accepted ? 1 : 2);
    if (accepted) {  // _applySolutionToVelocityField (viscositysolver.cpp:692-727): x is 0 off the rows
        const size_t off = plane_off(L, R0.kb), cnt = (size_t)(R0.ke - R0.kb) * L.sz;
        float *uvw[3] = {c->U, c->V, c->W};
        for (int m = 0; m < 3; m++) {
            if (c->vSwz) hipLaunchKernelGGL(k_unswizzle_to_f32<T>, GRID3(R0), 0, c->stream, R0, (const T *)v.x[m], uvw[m]);
            else if (c->pgrid[0] > 1 || c->pgrid[1] > 1) hipLaunchKernelGGL(k_box_to_f32<T>, GRID3(R0), 0, c->stream, R0, (const T *)v.x[m], uvw[m]);   // only what the rank owns: its i / j halo holds the neighbours' velocities
            else hipLaunchKernelGGL(k_vec_to_f32<T>, dim3(2048), dim3(256), 0, c->stream, (const T *)v.x[m] + off, uvw[m] + off, cnt);
        }
        const HaloArray uv[3] = {{c->U, 4}, {c->V, 4}, {c->W, 4}};
        if ((rc = fv_halo_copy(c, uv, 3, 1))) return rc;  // the pressure rhs at plane k1-1 reads W(k1)
    }
    HIPCHK(c, hipGetLastError());
    if (c->prm.kernel_timing) fv_ev_collect(c);
    if (info) *info = li;
    return success ? FLIPV_OK : (accepted ? FLIPV_WARN_NOT_CONVERGED : FLIPV_WARN_SOLVE_FAILED);
}

// One fine-level sweep of the multigrid preconditioner (k_viscosity_mg.hip) with the solver's tile SpMV kernel: out = in + omega
// (r - A in)/d (epi 1; epi 3 also adds (r, out) into sig(it + sig_shift)) or out = r - A in (epi 2).  fp32 vectors in the plain
// layout, zero off the rows; `out` must not alias `in`.  it_arg < 0: the device-side iteration counter (hipGraph replay).
void fv_visc_sweep_f32(flipv_context *c, float *const in[3], float *const out[3], int epi, const PcgScal &sc, int it_arg, float omega, int sig_shift) {
    PcgSys<float, 3> v = visc_sys<float>(c);
    for (int m = 0; m < 3; m++) { v.s[m] = in[m]; v.q[m] = out[m]; }
    int nb = pcg_grid(c, c->nActiveV);
    const int cap = c->prm.viscosity_spmv_grid_cap > 0 ? c->prm.viscosity_spmv_grid_cap : 512;
    if (nb > cap) nb = cap;
#define VSWEEP(P_, E_) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_visc_spmv<float, 4, P_, true, E_>), dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, \
                           c->vmU, c->vmV, c->vmW, c->fC, c->fEU, c->fEV, c->fEW, v, sc, it_arg, omega, sig_shift))
    if (c->vPred) { if (epi == 1) VSWEEP(true, EPI_JACOBI); else if (epi == 2) VSWEEP(true, EPI_RESIDUAL); else VSWEEP(true, EPI_JACOBI_DOT); }
    else { if (epi == 1) VSWEEP(false, EPI_JACOBI); else if (epi == 2) VSWEEP(false, EPI_RESIDUAL); else VSWEEP(false, EPI_JACOBI_DOT); }
#undef VSWEEP
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
    PcgScal sc;
    memset(&sc, 0, sizeof(sc));
    for (int w = 0; w < 3; w++) {
        if (c->vwV == 4) { if (c->viscosityPrec) launch_visc_spmv<double, 4>(c, sc, 0, 0, c->nActiveV); else launch_visc_spmv<float, 4>(c, sc, 0, 0, c->nActiveV); }
        else { if (c->viscosityPrec) launch_visc_spmv<double, 2>(c, sc, 0, 0, c->nActiveV); else launch_visc_spmv<float, 2>(c, sc, 0, 0, c->nActiveV); }
    }
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; r++) {
        if (c->vwV == 4) { if (c->viscosityPrec) launch_visc_spmv<double, 4>(c, sc, 0, 0, c->nActiveV); else launch_visc_spmv<float, 4>(c, sc, 0, 0, c->nActiveV); }
        else { if (c->viscosityPrec) launch_visc_spmv<double, 2>(c, sc, 0, 0, c->nActiveV); else launch_visc_spmv<float, 2>(c, sc, 0, 0, c->nActiveV); }
    }
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipEventSynchronize(b));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, a, b));
    c->prm.kernel_timing = saved;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms = (double)t / reps;
    *cells = (double)c->nActiveV * (256 * c->vwV);
    return FLIPV_OK;
}
