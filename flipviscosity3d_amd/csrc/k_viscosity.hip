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
#include <algorithm>
#include <vector>
#include "pcg_common.h"
#include "flipv_comm.h"
#include "visc_rows.h"
#include "brick.h"

enum { ST_FLUID = 1, ST_SOLID = 2, ST_ELIM = 3 };   // (ST_ELIM: a fluid face taken out of ONE solve's system, k_visc_singular_find)

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

// Pass 2: _estimateVolumeFractions (viscositysolver.cpp:180-270) for the listed indices, one thread per (index, lattice).
// The reference MEMOISES the liquid phi at the corner nodes of a lattice's cubes (nodalPhi / isNodalSet, :184-252): a node's value is the one its
// FIRST visitor computed -- the first band cube in the k-j-i scan that has the node as a corner -- at the position THAT cube derives for it,
// centre(i', j', k') + (+-hdx, +-hdx, +-hdx) in float.  The eight cubes around a node derive positions that differ in the last bit, phi there by ~1e-8,
// a control volume by up to 7e-6 -- and, where a corner's phi is within that of zero, the cube's classification (all corners >= 0 -> volume exactly 0):
// a face whose seven volumes are all zero is no row (:284-354).  On the rod + sheet scene at nu dt/dx^2 = 1.3e5 (tests/golden/honey96_nu1422) evaluating
// every corner from the cube's own centre, as this kernel did until round 4, gave a handful of rows the reference does not have -- slivers of
// volume 1e-20 that keep their free-fall velocity, 100 % off -- and 3e-4 in the end-of-substep velocities whatever the solver did.  So the corner's
// position is the first visitor's, found from the band mask of the 27 cubes around the index: volumes, and with them the row set, are the
// reference's bit for bit.
__global__ void k_volume_sample(Lay L, VolLattices Q, const float *__restrict__ phi, const uint8_t *__restrict__ valid, const unsigned *__restrict__ list,
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
        // band cubes of this lattice among the 27 around (i, j, k): bit (dk + 1) * 9 + (dj + 1) * 3 + (di + 1)
        unsigned vm = 0u;
#pragma unroll
        for (int dk = -1; dk <= 1; dk++)
#pragma unroll
            for (int dj = -1; dj <= 1; dj++)
#pragma unroll
                for (int di = -1; di <= 1; di++) {
                    const int ii = i + di, jj = j + dj, kk = k + dk;
                    if (ii >= 0 && jj >= 0 && kk >= 0 && ii < w && jj < h && kk < d && valid[gidx(L, ii, jj, kk)]) vm |= 1u << ((dk + 1) * 9 + (dj + 1) * 3 + (di + 1));
                }
        float p[8];
        int neg = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {  // q = 4*oi + 2*oj + ok: corner node (i + oi, j + oj, k + ok)
            const int oi = (q >> 2) & 1, oj = (q >> 1) & 1, ok = q & 1;
            // first visitor: the band cube with the smallest (k', j', i') that has the node as its corner (a', b', c'), i' = i + oi - a' ...
            int fa = oi, fb = oj, fc = ok;   // (this cube itself is always a candidate: it is in the band)
            bool found = false;
#pragma unroll
            for (int c2 = 1; c2 >= 0; c2--)
#pragma unroll
                for (int b2 = 1; b2 >= 0; b2--)
#pragma unroll
                    for (int a2 = 1; a2 >= 0; a2--) {
                        const int di = oi - a2, dj = oj - b2, dk = ok - c2;
                        if (!found && ((vm >> ((dk + 1) * 9 + (dj + 1) * 3 + (di + 1))) & 1u)) { found = true; fa = a2; fb = b2; fc = c2; }
                    }
            const int vi = i + oi - fa, vj = j + oj - fb, vk = k + ok - fc;
            const float cx = Q.cs[m][0] + (float)(vi * dx + hw), cy = Q.cs[m][1] + (float)(vj * dx + hw), cz = Q.cs[m][2] + (float)(vk * dx + hw);
            const float sx = cx + (fa ? hdx : -hdx), sy = cy + (fb ? hdx : -hdx), sz = cz + (fc ? hdx : -hdx);
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
                               const uint8_t *__restrict__ prevband, int full, int brick, Lay LB) {   // brick: the factor arrays are in the brick layout (bidx)
    IJK_OR_RETURN(L);
    if (!full && !band[c] && !prevband[c]) return;  // a factor is a volume of the same index times viscosity: zero off the band, and it stays zero
    const long sy = L.sy, sz = L.sz;
    const int I = L.I, J = L.J, K = L.K;
    if (i > I || j > J || k > K) return;
    const size_t co = brick ? bidx(LB, i, j, k) : c;
    if (i < I && j < J && k < K) fC[co] = 2 * factor * nu[c] * volC[c];
    if (i < I) {  // edgeU (I,J+1,K+1): nodes (i, j-1..j, k-1..k)
        float f = 0.0f;
        if (j >= 1 && k >= 1) f = factor * (0.25f * (nu[c - sy] + nu[c - sy - sz] + nu[c] + nu[c - sz])) * volEU[c];
        fEU[co] = f;
    }
    if (j < J) {  // edgeV (I+1,J,K+1): nodes (i-1..i, j, k-1..k)
        float f = 0.0f;
        if (i >= 1 && k >= 1) f = factor * (0.25f * (nu[c - 1] + nu[c - 1 - sz] + nu[c] + nu[c - sz])) * volEV[c];
        fEV[co] = f;
    }
    if (k < K) {  // edgeW (I+1,J+1,K): nodes (i-1..i, j-1..j, k)
        float f = 0.0f;
        if (i >= 1 && j >= 1) f = factor * (0.25f * (nu[c - 1] + nu[c - 1 - sy] + nu[c] + nu[c - sy])) * volEW[c];
        fEW[co] = f;
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
// (d_ref_volume itself: visc_rows.h)
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
                             int full, PcgSys<T, 3> v, double *__restrict__ bmax, int *__restrict__ nrows, int refdiag,
                             int brick, Lay LB, uint8_t *__restrict__ maskB, float *__restrict__ bU, float *__restrict__ bV, float *__restrict__ bW,
                             const float *__restrict__ phi) {
    // v.swz: diag, vm, r, x in the swizzled plane layout.  brick: the factor arrays are read, and EVERY output but `rowmask` (which
    // stays plain: it is this kernel's memory of where rows were) is written, in the brick layout; maskB = the brick-layout copy of the
    // row mask.  bU/bV/bW (optional) = a copy of the right-hand side in the layout of s (residual replacement recomputes r = b - A x)
    __shared__ double lds[4];
    IJK_OF_THREAD(L);
    double babs = 0.0, uabs = 0.0;
    int rows = 0;
    if (i < L.ie && j < L.je) {
        const size_t c = gidx(L, i, j, k);
        const long sy = L.sy, sz = L.sz;
        // the factor arrays: index and neighbour offsets in their own layout
        const size_t cf = brick ? bidx(LB, i, j, k) : c;
        const NbOff fo = brick ? nb_brick(LB, i, j, k) : nb_plain(L);
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
                const float fR = fC[cf], fL = fC[cf + fo.xm], fT = fEW[cf + fo.yp], fB = fEW[cf], fF = fEV[cf + fo.zp], fK = fEV[cf];
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
                if (dg[0] != 0.0f) { vm[0] = vol; vr[0] = (refdiag & 1) ? d_ref_volume(vol, fR, fL, fT, fB, fF, fK, dg[0]) : vol; }
            }
        }
        if (d_row_range(1, i, j, k, L) && SV[c] == ST_FLUID) {  // ---- V face (viscositysolver.cpp:472-568)
            const float vol = volV[c];
            if (vol > 0.0f || VEW[c + 1] > 0.0f || VEW[c] > 0.0f || VC[c] > 0.0f || VC[c - sy] > 0.0f || VEU[c + sz] > 0.0f ||
                VEU[c] > 0.0f) {
                const float fR = fEW[cf + fo.xp], fL = fEW[cf], fT = fC[cf], fB = fC[cf + fo.ym], fF = fEU[cf + fo.zp], fK = fEU[cf];
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
                if (dg[1] != 0.0f) { vm[1] = vol; vr[1] = (refdiag & 1) ? d_ref_volume(vol, fR, fL, fT, fB, fF, fK, dg[1]) : vol; }
            }
        }
        if (d_row_range(2, i, j, k, L) && SW[c] == ST_FLUID) {  // ---- W face (viscositysolver.cpp:570-664)
            const float vol = volW[c];
            if (vol > 0.0f || VEV[c + 1] > 0.0f || VEV[c] > 0.0f || VEU[c + sy] > 0.0f || VEU[c] > 0.0f || VC[c] > 0.0f ||
                VC[c - sz] > 0.0f) {
                const float fR = fEV[cf + fo.xp], fL = fEV[cf], fT = fEU[cf + fo.yp], fB = fEU[cf], fF = fC[cf], fK = fC[cf + fo.zm];
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
                if (dg[2] != 0.0f) { vm[2] = vol; vr[2] = (refdiag & 1) ? d_ref_volume(vol, fR, fL, fT, fB, fF, fK, dg[2]) : vol; }
            }
        }
    store:
        {
            // bits 0-2: component m has a row here; bits 3-5: ... and that row's velocity is one the substep USES -- its face borders a liquid cell (phi < 0), which is what
            // makes a face valid after the projection (fluidsimulation.cpp:598-688); every other row's value is overwritten by the extrapolation.  The velocity criterion
            // of the solve watches these rows only (PcgScal::step): the others include massless specks whose near-null modes CG moves by O(max|u|) for ever.
            uint8_t now = (uint8_t)((dg[0] != 0.0f) | ((dg[1] != 0.0f) << 1) | ((dg[2] != 0.0f) << 2));
            if (now) {
                const bool l0 = phi[c] < 0.0f;
                if ((now & 1) && (l0 || phi[c - 1] < 0.0f)) now |= 8;
                if ((now & 2) && (l0 || phi[c - sy] < 0.0f)) now |= 16;
                if ((now & 4) && (l0 || phi[c - sz] < 0.0f)) now |= 32;
            }
            if (now || prev || full) {  // off-row values (diag 0, volume -1, x = s = 0) persist between solves where nothing was a row
                const size_t cs = brick ? cf : (v.swz ? sidx(L, i, j, k) : c);   // own-index arrays
                const size_t cp = brick ? cf : c;                                  // s (and b), read with their halo
                dgU[cs] = dg[0]; dgV[cs] = dg[1]; dgW[cs] = dg[2];
                vmU[cs] = vm[0]; vmV[cs] = vm[1]; vmW[cs] = vm[2];
                vrU[cs] = vr[0]; vrV[cs] = vr[1]; vrW[cs] = vr[2];
                rowmask[c] = now;
                if (brick) maskB[cf] = now;
                if (bU) { bU[cp] = rv[0]; bV[cp] = rv[1]; bW[cp] = rv[2]; }
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    const float uo = m == 0 ? U[c] : (m == 1 ? V[c] : W[c]);
                    const float r0 = ((refdiag & 2) && dg[m] != 0.0f) ? rv[m] - (vr[m] - vm[m]) * uo : rv[m];   // (bit 1: the defect predictor, viscosity_solve_t: b - E u_old)
                    v.r[m][cs] = (RT<T>)r0; v.x[m][cs] = (T)0; v.s[m][cp] = (T)0;
                    babs = fmax(babs, fabs((double)rv[m]));
                    rows += dg[m] != 0.0f;
                }
                if (dg[0] != 0.0f) uabs = fmax(uabs, fabs((double)U[c]));
                if (dg[1] != 0.0f) uabs = fmax(uabs, fabs((double)V[c]));
                if (dg[2] != 0.0f) uabs = fmax(uabs, fabs((double)W[c]));
            }
        }
    done:;
    }
    const double bm = block_max_256(babs, lds);
    const double um = block_max_256(uabs, lds);
    const double nr = block_sum_256((double)rows, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        if (bm > 0.0) atomic_max_nonneg(bmax, bm);
        if (um > 0.0) atomic_max_nonneg(bmax + 1, um);   // the velocity scale of the rows (the multigrid loop's second stop criterion, PcgScal::bscale)
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

int fv_viscosity_pcg_mg(flipv_context *c, const PcgScal &sc, int cap, void (*spmv)(flipv_context *, const PcgScal &, int), int replace_period, int restart, int *conv_out);
int fv_vmg_prepare(flipv_context *c);

template <typename T, int NV>
static void launch_visc_spmv(flipv_context *c, const PcgScal &sc, int it, int first, int count, const PcgSys<T, 3> *sys = nullptr, int forceRdot = -1) {
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
#define VSPMV_A(N_, P_) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_visc_spmv<T, N_, P_, false, EPI_SPMV_A>), dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListV + first, count, c->tgV, c->L, \
                           vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, vv, sc, it))
    const bool rdot = forceRdot >= 0 ? forceRdot != 0 : (sc.conv ? !sc.noB : c->prm.beta_from_conjugacy == 0);   // benchmark launches (no scalars): the variant the solve would run
    const bool onlyA = !rdot && sc.onlyA;   // the multigrid-preconditioned loop (fv_viscosity_pcg_mg; fv_bench_viscosity_spmv mode 2)
    if (NV == 4 && c->nRunsV > 0 && first == 0 && count == c->nActiveV) {   // k-marching over the run list (the whole system)
        int nbm = pcg_grid(c, c->nRunsV);
        const int capm = c->prm.viscosity_spmv_grid_cap > 0 ? cap : 256 * FLIPV_MARCH_OCC;
        if (nbm > capm) nbm = capm;
#define VMARCH(P_, R_, S_) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_visc_spmv_march<T, P_, R_, S_>), dim3(nbm), dim3(64, 4, 1), 0, c->stream, (const Run *)c->runsV, c->nRunsV, \
                           (const unsigned *)(c->vPred ? c->rmaskV : nullptr), c->tgV, c->L, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, vv, sc, it))
        // streaming accesses on filled systems whose 52 bytes per index exceed the memory-side cache (pcg_common.h: ldvs)
        const bool stream = !c->vPred && (double)c->nActiveV * (256 * 4) * 52.0 > 256.0 * 1024 * 1024;
#define VMARCH_A(P_, S_) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_visc_spmv_march<T, P_, false, S_, true>), dim3(nbm), dim3(64, 4, 1), 0, c->stream, (const Run *)c->runsV, c->nRunsV, \
                           (const unsigned *)(c->vPred ? c->rmaskV : nullptr), c->tgV, c->L, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, vv, sc, it))
        if (onlyA) { if (c->vPred) VMARCH_A(true, false); else if (stream) VMARCH_A(false, true); else VMARCH_A(false, false); }
        else if (c->vPred) { if (rdot) VMARCH(true, true, false); else VMARCH(true, false, false); }
        else if (stream) { if (rdot) VMARCH(false, true, true); else VMARCH(false, false, true); }
        else { if (rdot) VMARCH(false, true, false); else VMARCH(false, false, false); }
#undef VMARCH
#undef VMARCH_A
        if (timed) fv_ev_end(c);
        return;
    }
    if (onlyA) { if (NV == 4 && c->vPred) VSPMV_A(4, true); else VSPMV_A(NV, NV == 2); }
    else if (NV == 4 && c->vPred) { if (rdot) VSPMV(4, true, true); else VSPMV(4, true, false); }
    else { if (rdot) VSPMV(NV, NV == 2, true); else VSPMV(NV, NV == 2, false); }
#undef VSPMV
#undef VSPMV_A
    if (timed) fv_ev_end(c);
}

// flipv_params.viscosity_preconditioner = AUTO: the multigrid V-cycle unless the diagonal is known to CONVERGE for less.
// A default run never hands back an iterate stopped at the cap where a converged one is affordable: the reference's own solve does
// (its MIC(0) PCG stops 7 000-13 000 iterations short at 256^3), but that is its defect, not a target -- what can be pinned against
// the reference is the converged answer.  Costs in units of one diagonal-preconditioned iteration (SpMV + update): a multigrid
// iteration ~7.5, its set-up ~35; one multigrid iteration does the work of ~15 diagonal ones (10-30 measured).
//   nu_max dt/dx^2 <= 64 (a-priori stiffness; whatever the history) -> diagonal: it converges in ~200 iterations of a small system there,
//                                                                  and it is the safer of the two on such systems (the V-cycle leaves
//                                                                  the near-rigid modes of tiny liquid clusters -- a few faces with
//                                                                  small control volumes, residual ~ volume x error -- to the Krylov
//                                                                  loop, whose stop test cannot see them: twobody20 fixture, 3-5 faces
//                                                                  30 % off at a converged residual)
//   no history (first solve of a context, resumed run)          -> multigrid
//   after a diagonal solve that converged in n iterations       -> diagonal again while n < 35 + 7.5 n/15 with 10 % hysteresis (n < ~75)
//   after a diagonal solve that did not converge                -> multigrid (and THAT solve is repeated with the multigrid, viscosity_solve_t)
//   after a multigrid solve of m iterations                     -> diagonal if 15 m is safely inside the cap and cheaper than 35 + 7.5 m (m < ~5)
// Decisions depend on iteration counts only, never on wall-clock times, so a run is reproducible; every path converges to the same
// tolerance, so what the history changes is the cost of a solve, not its answer beyond solver tolerance.
// nu dt/dx^2 up to which AUTO takes the diagonal without asking: 8.  (64 until round 5: the holdout sweep's draws between 10 and 64 needed 104 ... 677 diagonal
// iterations -- one of them the cap's worth -- where the multigrid takes 13 ... 35, and left 6e-3 / 2e-2 of max|u| on two of them: profiles/r5/holdout_sweep.log)
constexpr double FV_AUTO_DIAGONAL_STIFFNESS = 8.0;
static bool fv_visc_auto_pick(const flipv_context *c, float dt) {
    if ((double)c->viscosity_max_any * (double)dt / ((double)c->dx * (double)c->dx) <= FV_AUTO_DIAGONAL_STIFFNESS) return false;
    if (c->vLastPrec == 0) return true;
    const double cap = (double)c->prm.viscosity_max_iterations;
    const double MG_ITER = 7.5, MG_SETUP = 35.0, RATIO = 15.0;
    if (c->vLastPrec == 1) {
        if (!c->vLastConverged) return true;
        const double n = (double)c->vLastIts;
        return MG_SETUP + MG_ITER * n / RATIO < 0.9 * n || n > 0.8 * cap;
    }
    if (!c->vLastConverged) return false;   // a multigrid solve that stalled or ran into the cap: the diagonal (whose capped iterate is what the acceptance rule is about)
    const double m = (double)c->vLastIts;
    const double costM = MG_SETUP + MG_ITER * m, costD = RATIO * m;
    return !(costD < 0.5 * cap && 1.1 * costD < costM);
}

// zero every solver array of the viscosity system (a change between the brick layout and the plain ones: the two address the same
// buffers differently, and each relies on "zero wherever nothing was ever written")
static int visc_zero_solver_arrays(flipv_context *c) {
    const size_t g = c->L.guard, n = g + c->solverCap;
    float *f32[] = {c->fC, c->fEU, c->fEV, c->fEW, c->vDiagU, c->vDiagV, c->vDiagW, c->vmU, c->vmV, c->vmW, c->vrU, c->vrV, c->vrW, c->vB[0], c->vB[1], c->vB[2]};
    for (float *p : f32) HIPCHK(c, hipMemsetAsync(p - g, 0, n * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->vMaskB - g, 0, n, c->stream));
    for (int m = 0; m < 3; m++) {
        HIPCHK(c, hipMemsetAsync(c->vXacc[m] - g, 0, n * sizeof(double), c->stream));
        void *v[4] = {c->vX[m], c->vR[m], c->vZ[m], c->vS[m]};
        for (void *p : v) HIPCHK(c, hipMemsetAsync((double *)p - g, 0, n * sizeof(double), c->stream));
    }
    return FLIPV_OK;
}
__global__ __launch_bounds__(256) void k_brick_zero_f64(const int *__restrict__ bricks, int nb, double *__restrict__ a0, double *__restrict__ a1, double *__restrict__ a2) {
    for (int e = blockIdx.x * 4 + (int)threadIdx.y; e < nb; e += gridDim.x * 4) {
        const size_t a = ((size_t)bricks[e] << 6) + threadIdx.x;
        a0[a] = 0.0; a1[a] = 0.0; a2[a] = 0.0;
    }
}

// ------------------------------------------------------------------ the system's geometry (functions of the liquid and the solid SDF only), on stream `st`
static int visc_geometry(flipv_context *c, hipStream_t st) {
    const Lay &L = c->L;
    // face states
    if (c->faceStateVersion != c->solidVersion) {  // functions of the solid SDF only: everywhere
        const Lay F1 = fv_range(c, 1), F2 = fv_range(c, 2);
        hipLaunchKernelGGL(k_solid_center, GRID3(F2), 0, st, F2, c->solid, c->scp);
        hipLaunchKernelGGL(k_face_states, GRID3(F1), 0, st, F1, c->scp, c->stU, c->stV, c->stW);
        c->faceStateVersion = c->solidVersion;
    }
    const Lay R1 = fv_range_liquid(c, 1, 4), R2 = fv_range_liquid(c, 2, 4), R3 = fv_range_liquid(c, 3, 4), R4 = fv_range_liquid(c, 4, 4);
    // band mask + the seven volume lattices (viscositysolver.cpp:135-178).  The mask must be final one entry beyond the volumes' range R1: a
    // corner's first visitor is looked up among the 27 cubes around an index (k_volume_sample)
    hipLaunchKernelGGL(k_valid_init, GRID3(R4), 0, st, R4, c->phi, c->validCells);
    hipLaunchKernelGGL(k_valid_dilate, GRID3(R3), 0, st, R3, c->validCells, c->validTmp);
    hipLaunchKernelGGL(k_valid_dilate, GRID3(R2), 0, st, R2, c->validTmp, c->validCells);
    const float h = (float)(0.5 * c->dx);
    const int fullVol = c->bandPrevValid ? 0 : 1;  // volumes and factors are stored only where the band is or was in the previous solve
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
    HIPCHK(c, hipMemsetAsync(nlist, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_volume_classify, GRID3(R1), 0, st, R1, Q, c->phi, c->validCells, c->bandPrev, fullVol, c->surfList, nlist);
    hipLaunchKernelGGL(k_volume_sample, dim3(4096), dim3(256), 0, st, c->L, Q, c->phi, (const uint8_t *)c->validCells, c->surfList, nlist, c->dx);
    return FLIPV_OK;
}
// (Measured in round 4 and dropped: the geometry on a SIDE STREAM started right after the particle level set, beside P2G + extrapolation + body force, which
// touch none of its arrays.  The P2G phase grew from 0.75 to 0.98 ms, the viscosity phase shrank from 12.46 to 12.1-12.3: 946-960 MCells/s against 954-961.)

// ------------------------------------------------------------------ the fp64 accumulator on the PLANE layouts
// What k_bflush / k_bresidual / k_unbrick_to_f32 are to the brick layout (k_viscosity_brick.hip): the solution accumulated in fp64 beside the PCG's
// fp32 x, the residual b - A_outer xacc evaluated in fp64, so that the two-stage defect correction (viscosity_solve_t) -- and the restart of a
// stalled fp32 loop -- exist wherever the liquid is too dense for bricks.  xacc and the residual's scratch (q64) are plain-layout arrays; x, r and
// the own volumes follow the solve's layout (plain, or the 8 x 4 patches of sidx).  The fp64 SpMV itself is the solver's own kernel with T = double.
// mode: as k_bflush -- 0 xacc += x, x = 0; 1 xacc += x, x kept; 2 xacc -= x, x = 0; 3 x = 0; 4 xacc = 0 (between solves)
template <typename T>
__global__ void k_plane_flush(Lay L, const uint8_t *__restrict__ rowmask, int swz, T *__restrict__ x0, T *__restrict__ x1, T *__restrict__ x2,
                              double *__restrict__ a0, double *__restrict__ a1, double *__restrict__ a2, int mode) {
    IJK_OR_RETURN(L);
    const unsigned m = rowmask[c];
    if (!m) return;
    const size_t cs = swz ? sidx(L, i, j, k) : c;
    T *x[3] = {x0, x1, x2};
    double *a[3] = {a0, a1, a2};
#pragma unroll
    for (int q = 0; q < 3; q++) {
        if (!((m >> q) & 1u)) continue;
        if (mode == 4) { a[q][c] = 0.0; continue; }
        if (mode == 0 || mode == 1) a[q][c] += (double)x[q][cs];
        else if (mode == 2) a[q][c] -= (double)x[q][cs];
        if (mode != 1) x[q][cs] = (T)0;
    }
}
// r = b - q64 on the rows (q64 = A_outer xacc, fp64), max|r| into rmax(0)
template <typename T>
__global__ void k_plane_residual_finish(Lay L, const uint8_t *__restrict__ rowmask, int swz, const float *__restrict__ b0, const float *__restrict__ b1,
                                        const float *__restrict__ b2, const double *__restrict__ q0, const double *__restrict__ q1, const double *__restrict__ q2,
                                        RT<T> *__restrict__ r0, RT<T> *__restrict__ r1, RT<T> *__restrict__ r2, PcgScal sc) {
    __shared__ double lds[4];
    IJK_OF_THREAD(L);
    double mx = 0.0;
    if (i < L.ie && j < L.je) {
        const size_t c = gidx(L, i, j, k);
        const unsigned m = rowmask[c];
        if (m) {
            const size_t cs = swz ? sidx(L, i, j, k) : c;
            const float *b[3] = {b0, b1, b2};
            const double *q[3] = {q0, q1, q2};
            RT<T> *r[3] = {r0, r1, r2};
#pragma unroll
            for (int e = 0; e < 3; e++)
                if ((m >> e) & 1u) {
                    const RT<T> rt = (RT<T>)((double)b[e][c] - q[e][cs]);
                    r[e][cs] = rt;
                    mx = fmax(mx, fabs((double)rt));
                }
        }
    }
    const double bm = block_max_256(mx, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && bm > 0.0) {
        const unsigned bl = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int sl = sc.slot0 + (int)(bl % (unsigned)sc.nslot) + (sc.nbank > 1 ? (int)((bl / (unsigned)sc.nslot) % (unsigned)sc.nbank) * sc.bstride : 0);
        atomic_max_nonneg(sc.rmax(0) + sl, bm);
    }
}
// The same residual with the REFERENCE's rows formed one by one -- each row's six factors in its own float arithmetic, its own float-rounded diagonal
// (visc_rows.h: d_ref_row_factors) -- for a variable viscosity field: r = b - A_ref xacc in fp64, max|r| into rmax(0).  One thread per index of the launch box.
template <typename T>
__global__ void k_plane_residual_ref(Lay L, const uint8_t *__restrict__ rowmask, int swz, const float *__restrict__ nu, const float *__restrict__ vC,
                                     const float *__restrict__ vEU, const float *__restrict__ vEV, const float *__restrict__ vEW, float factor,
                                     const float *__restrict__ xmU, const float *__restrict__ xmV, const float *__restrict__ xmW,
                                     const double *__restrict__ xu, const double *__restrict__ xv, const double *__restrict__ xw,
                                     const float *__restrict__ bU, const float *__restrict__ bV, const float *__restrict__ bW,
                                     RT<T> *__restrict__ r0, RT<T> *__restrict__ r1, RT<T> *__restrict__ r2, PcgScal sc) {
    __shared__ double lds[4];
    IJK_OF_THREAD(L);
    double mx = 0.0;
    if (i < L.ie && j < L.je) {
        const size_t a = gidx(L, i, j, k);
        const unsigned m = rowmask[a];
        if (m) {
            const size_t cs = swz ? sidx(L, i, j, k) : a;
            const NbOff o = nb_plain(L);
            const double U0 = xu[a], U0l = xu[a + o.xm], U0r = xu[a + o.xp], Ujm = xu[a + o.ym], Ujp = xu[a + o.yp], Ukm = xu[a + o.zm], Ukp = xu[a + o.zp];
            const double Ujmr = xu[a + o.ym + o.xp], Ukmr = xu[a + o.zm + o.xp];
            const double V0 = xv[a], V0l = xv[a + o.xm], V0r = xv[a + o.xp], Vjm = xv[a + o.ym], Vjp = xv[a + o.yp], Vkm = xv[a + o.zm], Vkp = xv[a + o.zp];
            const double Vjpl = xv[a + o.yp + o.xm], Vjpkm = xv[a + o.yp + o.zm];
            const double W0 = xw[a], W0l = xw[a + o.xm], W0r = xw[a + o.xp], Wjm = xw[a + o.ym], Wjp = xw[a + o.yp], Wkm = xw[a + o.zm], Wkp = xw[a + o.zp];
            const double Wkpl = xw[a + o.zp + o.xm], Wjmkp = xw[a + o.ym + o.zp];
            const double RU = (m & 1u) ? (double)bU[a] : 0.0, RV = (m & 2u) ? (double)bV[a] : 0.0, RW = (m & 4u) ? (double)bW[a] : 0.0;
            const RefRowFactors F = d_ref_row_factors(nu, vC, vEU, vEV, vEW, a, L.sy, L.sz, factor);
            auto refvol = [](float vol, const float *f) {
                const float dg = vol + f[0] + f[1] + f[2] + f[3] + f[4] + f[5];
                return dg != 0.0f ? d_ref_volume(vol, f[0], f[1], f[2], f[3], f[4], f[5], dg) : -1.0f;
            };
            const float none = -1.0f, z = 0.0f;
            Vec<double, 1> yU, yV, yW, d0, d1, d2;
            double ta = 0.0, tb = 0.0, tc = 0.0;
#define V1F(x_) Vec<float, 1>{{x_}}
#define V1D(x_) Vec<double, 1>{{x_}}
#define XARGS V1D(U0), V1D(Ujm), V1D(Ujp), V1D(Ukm), V1D(Ukp), V1D(V0), V1D(Vjm), V1D(Vjp), V1D(Vkm), V1D(Vkp), V1D(W0), V1D(Wjm), V1D(Wjp), V1D(Wkm), V1D(Wkp), V1D(Vjpkm), V1D(Wjmkp), V1D(RU), V1D(RV), V1D(RW)
#define XTAIL U0l, U0r, V0l, V0r, W0l, W0r, Vjpl, Wkpl, Ujmr, Ukmr
            RT<T> *r[3] = {r0, r1, r2};
            if (m & 1u) {   // U row: right, left = C0, C0l; top, bottom = EWjp, EW0; front, back = EVkp, EV0
                const float M = refvol(xmU[cs], F.U);
                d_visc_rows<double, 1, true, EPI_RESIDUAL>(V1F(M), V1F(none), V1F(none), V1F(F.U[0]), V1F(z), V1F(z), V1F(F.U[3]), V1F(F.U[2]), V1F(F.U[5]), V1F(F.U[4]), V1F(z), V1F(z), V1F(z),
                                                           XARGS, F.U[1], z, z, XTAIL, yU, d1, d2, ta, tb, tc, 0.0);
                const RT<T> rt = (RT<T>)yU.v[0]; r[0][cs] = rt; mx = fmax(mx, fabs((double)rt));
            }
            if (m & 2u) {   // V row: right, left = EW0r, EW0; top, bottom = C0, Cjm; front, back = EUkp, EU0
                const float M = refvol(xmV[cs], F.V);
                d_visc_rows<double, 1, true, EPI_RESIDUAL>(V1F(none), V1F(M), V1F(none), V1F(F.V[2]), V1F(F.V[3]), V1F(z), V1F(F.V[1]), V1F(z), V1F(z), V1F(z), V1F(F.V[5]), V1F(z), V1F(F.V[4]),
                                                           XARGS, z, F.V[0], z, XTAIL, d0, yV, d2, ta, tb, tc, 0.0);
                const RT<T> rt = (RT<T>)yV.v[0]; r[1][cs] = rt; mx = fmax(mx, fabs((double)rt));
            }
            if (m & 4u) {   // W row: right, left = EV0r, EV0; top, bottom = EUjp, EU0; front, back = C0, Ckm
                const float M = refvol(xmW[cs], F.W);
                d_visc_rows<double, 1, true, EPI_RESIDUAL>(V1F(none), V1F(none), V1F(M), V1F(F.W[4]), V1F(z), V1F(F.W[5]), V1F(z), V1F(z), V1F(F.W[1]), V1F(z), V1F(F.W[3]), V1F(F.W[2]), V1F(z),
                                                           XARGS, z, z, F.W[0], XTAIL, d0, d1, yW, ta, tb, tc, 0.0);
                const RT<T> rt = (RT<T>)yW.v[0]; r[2][cs] = rt; mx = fmax(mx, fabs((double)rt));
            }
#undef V1F
#undef V1D
#undef XARGS
#undef XTAIL
        }
    }
    const double bm = block_max_256(mx, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && bm > 0.0) {
        const unsigned bl = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int sl = sc.slot0 + (int)(bl % (unsigned)sc.nslot) + (sc.nbank > 1 ? (int)((bl / (unsigned)sc.nslot) % (unsigned)sc.nbank) * sc.bstride : 0);
        atomic_max_nonneg(sc.rmax(0) + sl, bm);
    }
}
// x + xacc -> velocity grid over a launch box
template <typename T>
__global__ void k_plane_writeback(Lay L, int swz, const T *__restrict__ x, const double *__restrict__ acc, float *__restrict__ o) {
    IJK_OR_RETURN(L);
    o[c] = (float)(acc[c] + (double)x[swz ? sidx(L, i, j, k) : c]);
}
// the three fp64 scratch arrays of the plane residual (plain layout), allocated on first use
static int plane_q64_reserve(flipv_context *c) {
    if (c->vQ64[0]) return FLIPV_OK;
    const size_t n = c->L.n + 2 * c->L.guard;
    for (int m = 0; m < 3; m++) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, n * sizeof(double));
        if (e != hipSuccess) { c->err = std::string("hipMalloc(fp64 residual scratch): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        c->allocs.push_back(q);
        HIPCHK(c, hipMemsetAsync(q, 0, n * sizeof(double), c->stream));
        c->vQ64[m] = (double *)q + c->L.guard;
    }
    return FLIPV_OK;
}
template <typename T>
static void plane_flush(flipv_context *c, const Lay &R, int mode) {
    hipLaunchKernelGGL((k_plane_flush<T>), GRID3(R), 0, c->stream, R, (const uint8_t *)c->vRowMask, c->vSwz, (T *)c->vX[0], (T *)c->vX[1], (T *)c->vX[2], c->vXacc[0], c->vXacc[1], c->vXacc[2], mode);
}
// xacc += x (flushMode, as fv_brick_refine), r = b - A_outer xacc in fp64, max|r| into rmax(0) of a cleared scalar block
template <typename T>
static int fv_plane_refine(flipv_context *c, const Lay &R, const PcgScal &sc, size_t scalBytes, bool outerExact, int flushMode) {
    int rc = plane_q64_reserve(c);
    if (rc) return rc;
    plane_flush<T>(c, R, flushMode);
    { const FillJob z[2] = {{sc.base, scalBytes, 0}, {sc.base + sc.bstride, sc.nbank > 1 ? (size_t)(sc.nbank - 1) * sc.bstride * sizeof(double) : 0, 0}};
      if ((rc = fv_fill_list(c, z, 2))) return rc; }
    if (c->comm) {
        const HaloArray xa[3] = {{c->vXacc[0], sizeof(double)}, {c->vXacc[1], sizeof(double)}, {c->vXacc[2], sizeof(double)}};
        if ((rc = fv_halo_copy(c, xa, 3, 1))) return rc;
    }
    if (!outerExact && c->vPerRowFactors) {   // a variable viscosity field: the reference's rows one by one (k_plane_residual_ref)
        hipLaunchKernelGGL((k_plane_residual_ref<T>), GRID3(R), 0, c->stream, R, (const uint8_t *)c->vRowMask, c->vSwz, (const float *)c->visc, (const float *)c->volC, (const float *)c->volEU,
                           (const float *)c->volEV, (const float *)c->volEW, c->vFactorNow, (const float *)c->vmU, (const float *)c->vmV, (const float *)c->vmW,
                           (const double *)c->vXacc[0], (const double *)c->vXacc[1], (const double *)c->vXacc[2], (const float *)c->vB[0], (const float *)c->vB[1], (const float *)c->vB[2],
                           (RT<T> *)c->vR[0], (RT<T> *)c->vR[1], (RT<T> *)c->vR[2], sc);
        return fv_allreduce_scalars(c, sc.rmax(0), NSLOT);
    }
    // q64 = A_outer xacc with the solver's own SpMV kernel in fp64 (no scalars: an unconditional launch without dot products)
    PcgSys<double, 3> v;
    v.swz = c->vSwz; v.mask = c->vRowMask; v.mlist = c->mlistV;
    v.diag[0] = c->vDiagU; v.diag[1] = c->vDiagV; v.diag[2] = c->vDiagW;
    for (int m = 0; m < 3; m++) { v.x[m] = nullptr; v.r[m] = nullptr; v.q[m] = c->vQ64[m]; v.s[m] = c->vXacc[m]; }
    PcgScal none;
    memset(&none, 0, sizeof(none));
    const int keepOp = c->vOperatorExact, keepTiming = c->prm.kernel_timing;
    c->vOperatorExact = outerExact ? 1 : 0;
    c->prm.kernel_timing = 0;
    if (c->vwV == 4) launch_visc_spmv<double, 4>(c, none, 0, 0, c->nActiveV, &v, 0); else launch_visc_spmv<double, 2>(c, none, 0, 0, c->nActiveV, &v, 0);
    c->vOperatorExact = keepOp;
    c->prm.kernel_timing = keepTiming;
    hipLaunchKernelGGL((k_plane_residual_finish<T>), GRID3(R), 0, c->stream, R, (const uint8_t *)c->vRowMask, c->vSwz, (const float *)c->vB[0], (const float *)c->vB[1], (const float *)c->vB[2],
                       (const double *)c->vQ64[0], (const double *)c->vQ64[1], (const double *)c->vQ64[2], (RT<T> *)c->vR[0], (RT<T> *)c->vR[1], (RT<T> *)c->vR[2], sc);
    return fv_allreduce_scalars(c, sc.rmax(0), NSLOT);
}

// ------------------------------------------------------------------ massless clusters (flipv_params.viscosity_massless_polish)
// A row without own volume whose diagonal is one stress term -- the factor of ONE edge of the control-volume lattice, >= 0.99 of the row's diagonal -- shares that term
// with up to three other rows around the edge; where two or more of them are such rows, the system pins their common stress down and leaves their split to couplings
// 1e-5 of it: a mode of Jacobi-scaled eigenvalue ~1e-6 that fp32 CG neither sees in its residual nor moves (holdout draws 20 and 30: ONE such face the substep uses was
// 2.2e-4 / 3.8e-4 from the reference's converged answer, and the projection and the extrapolation carried it to 8 / 180 faces; profiles/r5/holdout_misses_20_30.log).  The
// reference's MIC(0)-PCG resolves it given enough iterations.  Here: after the solve, every such cluster (<= 4 rows) is solved exactly in fp64 with everything around it
// held -- the reference's rows (viscositysolver.cpp:394-446, 491-568, 589-664) with the reference's float factors (d_ref_row_factors) and float diagonal.
// entry t (0 .. 14) of the reference's row of component `comp` with the six factors F = (right, left, top, bottom, front, back) and float diagonal `diag`:
// which component, at which offset from the row's own index, with which coefficient
__device__ __forceinline__ void d_polish_entry(int comp, int t, const float F[6], float diag, long sy, long sz, int &ec, long &eo, double &coef) {
    const long own[7] = {0, 1, -1, sy, -sy, sz, -sz};
    if (t < 7) { ec = comp; eo = own[t]; coef = t == 0 ? (double)diag : -(double)F[t - 1]; return; }
    const int q = t - 7;   // the eight cross terms: two per stress term the row shares with another component
    // per component: (other component, offset, factor slot, sign) x 8  -- viscositysolver.cpp:437-446, 544-553, 642-651
    if (comp == 0) {
        const int oc[8] = {1, 1, 1, 1, 2, 2, 2, 2}; const long of[8] = {sy, sy - 1, 0, -1, sz, sz - 1, 0, -1}; const int sl[8] = {2, 2, 3, 3, 4, 4, 5, 5}; const int sg[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
        ec = oc[q]; eo = of[q]; coef = sg[q] * (double)F[sl[q]];
    } else if (comp == 1) {
        const int oc[8] = {0, 0, 0, 0, 2, 2, 2, 2}; const long of[8] = {1, 1 - sy, 0, -sy, sz, sz - sy, 0, -sy}; const int sl[8] = {0, 0, 1, 1, 4, 4, 5, 5}; const int sg[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
        ec = oc[q]; eo = of[q]; coef = sg[q] * (double)F[sl[q]];
    } else {
        const int oc[8] = {0, 0, 0, 0, 1, 1, 1, 1}; const long of[8] = {1, 1 - sz, 0, -sz, sy, sy - sz, 0, -sz}; const int sl[8] = {0, 0, 1, 1, 2, 2, 3, 3}; const int sg[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
        ec = oc[q]; eo = of[q]; coef = sg[q] * (double)F[sl[q]];
    }
}
// pass 1, every index of the liquid's range: the edges with a PARTIAL volume (a row dominated by an edge of full volume would need every other volume around it to
// vanish: not at a full edge) and at least two massless rows around them, appended to a list (list[0] = their number)
constexpr int FV_POLISH_CAP = 1 << 16;
__global__ void k_visc_massless_find(Lay L, const float *__restrict__ volEU, const float *__restrict__ volEV, const float *__restrict__ volEW, const float *__restrict__ volU,
                                     const float *__restrict__ volV, const float *__restrict__ volW, const uint8_t *__restrict__ rowmask, unsigned long long *__restrict__ list) {
    IJK_OR_RETURN(L);
    const long sy = L.sy, sz = L.sz;
    if (rowmask[c] == 0) return;   // (each of the three edges at c has a row AT c among its four)
    if (i < 2 || j < 2 || k < 2 || i > L.I - 2 || j > L.J - 2 || k > L.K - 2) return;
    const float ev[3] = {volEV[c], volEW[c], volEU[c]};
    const float *const vol[3] = {volU, volV, volW};
    const int rc_[3][4] = {{0, 0, 2, 2}, {0, 0, 1, 1}, {1, 1, 2, 2}};
    const long ro[3][4] = {{0, -sz, 0, -1}, {0, -sy, 0, -1}, {0, -sz, 0, -sy}};
    for (int fam = 0; fam < 3; fam++) {
        if (!(ev[fam] > 0.0f && ev[fam] < 1.0f)) continue;
        int cand = 0, used = 0;   // massless rows around the edge; those among them whose velocity the substep uses (mask bits 3-5: the others are overwritten by the extrapolation)
        for (int q = 0; q < 4; q++) {
            const size_t p = c + ro[fam][q];
            const unsigned mk = rowmask[p];
            const int is = ((mk >> rc_[fam][q]) & 1u) && vol[rc_[fam][q]][p] == 0.0f;
            cand += is; used += is && ((mk >> (3 + rc_[fam][q])) & 1u);
        }
        if (cand < 2 || used < 1) continue;
        const unsigned long long at = atomicAdd(list, 1ull);
        if (at < (unsigned long long)FV_POLISH_CAP) list[1 + at] = ((unsigned long long)c << 2) | (unsigned long long)fam;
    }
}
// pass 2, one thread per listed edge (no large per-thread arrays: a kernel with kilobytes of scratch per lane pays ~100 us of scratch set-up per dispatch)
// MARK: before the solve -- the rows of every cluster the kernel will replace afterwards lose their "watched" bit (mask bits 3-5, plain and brick copy): CG moves exactly
// these rows by 1e-5 ... 1e-3 max|u| per iteration for ever, which is most of what held the velocity criterion's patience on a liquid lying on the wall; nothing is solved.
template <bool MARK>
__global__ __launch_bounds__(64) void k_visc_massless_polish(Lay L, const float *__restrict__ nu, const float *__restrict__ volC, const float *__restrict__ volEU, const float *__restrict__ volEV,
                                       const float *__restrict__ volEW, const float *__restrict__ volU, const float *__restrict__ volV, const float *__restrict__ volW,
                                       uint8_t *__restrict__ rowmask, float *__restrict__ U, float *__restrict__ V, float *__restrict__ W, float factor, int inner,
                                       const unsigned long long *__restrict__ list, int *__restrict__ count, uint8_t *__restrict__ maskB, Lay LB,
                                       unsigned long long *__restrict__ outRow, float *__restrict__ outVal) {
    // (solve pass: the clusters' values go to outRow / outVal -- four slots per listed edge, row = ~0 where unused -- and k_visc_massless_write stores them afterwards: every
    // cluster is solved against the SAME velocities, whichever threads run first.  ADVICE r5: written in place, a cluster read neighbours another thread was replacing.)
    const long sy = L.sy, sz = L.sz;
    const float *const vol[3] = {volU, volV, volW};
    float *const X[3] = {U, V, W};
    unsigned long long nlist = list[0];
    if (nlist > (unsigned long long)FV_POLISH_CAP) nlist = FV_POLISH_CAP;
    // the three edge families: their four rows as (component, offset from the edge's index, slot of the edge among the row's six factors)
    //   edgeV (tau_xz): U(c) back, U(c - sz) front, W(c) left, W(c - 1) right;  edgeW (tau_xy): U(c) bottom, U(c - sy) top, V(c) left, V(c - 1) right;
    //   edgeU (tau_yz): V(c) back, V(c - sz) front, W(c) bottom, W(c - sy) top
    const int rc_[3][4] = {{0, 0, 2, 2}, {0, 0, 1, 1}, {1, 1, 2, 2}};
    const long ro[3][4] = {{0, -sz, 0, -1}, {0, -sy, 0, -1}, {0, -sz, 0, -sy}};
    const int rs[3][4] = {{5, 4, 1, 0}, {3, 2, 1, 0}, {5, 4, 3, 2}};
    for (unsigned long long item = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; item < nlist; item += (unsigned long long)gridDim.x * blockDim.x) {
        const size_t c = (size_t)(list[1 + item] >> 2);
        const int fam = (int)(list[1 + item] & 3ull);
        const int k = (int)(c / (size_t)sz) + L.oz, j = (int)((c % (size_t)sz) / (size_t)sy) + L.oy, i = (int)(c % (size_t)sy) + L.ox;   // (the inverse of gidx: for the block-context test)
        int mc[4];
        size_t mp[4];
        float mF[4][6], md[4];
        int n = 0;
        if (!MARK) for (int r = 0; r < 4; r++) outRow[4 * item + r] = ~0ull;
        for (int q = 0; q < 4; q++) {
            const int comp = rc_[fam][q];
            const size_t p = c + ro[fam][q];
            if (!((rowmask[p] >> comp) & 1) || vol[comp][p] != 0.0f) continue;
            if (inner) {   // (a block context: rows whose stencil stays inside what the rank holds current values of)
                const int pi = i + (ro[fam][q] == -1 ? -1 : 0), pj = j + (ro[fam][q] == -sy ? -1 : 0), pk = k + (ro[fam][q] == -sz ? -1 : 0);
                // (only towards sides that HAVE a neighbouring rank: at a wall of the domain the owned box ends where the lattice ends -- ADVICE r5: the test used to drop
                // the rows along physical walls too, so that block runs polished another set of rows than the single domain)
                if ((pi - 1 < L.olo[0] && L.olo[0] > 0) || (pi + 1 >= L.ohi[0] && L.ohi[0] < L.I) || (pj - 1 < L.olo[1] && L.olo[1] > 0) || (pj + 1 >= L.ohi[1] && L.ohi[1] < L.J) ||
                    (pk - 1 < L.olo[2] && L.olo[2] > 0) || (pk + 1 >= L.ohi[2] && L.ohi[2] < L.K)) continue;
            }
            const RefRowFactors F = d_ref_row_factors(nu, volC, volEU, volEV, volEW, p, sy, sz, factor);
            const float *f = comp == 0 ? F.U : (comp == 1 ? F.V : F.W);
            const float diag = 0.0f + f[0] + f[1] + f[2] + f[3] + f[4] + f[5];   // (own volume 0; the reference's float sum, in its order)
            if (!(diag > 0.0f) || f[rs[fam][q]] < 0.99f * diag) continue;
            mc[n] = comp; mp[n] = p; md[n] = diag;
            for (int t = 0; t < 6; t++) mF[n][t] = f[t];
            n++;
        }
        if (n < 2) continue;
        if (MARK) {
            for (int r = 0; r < n; r++) {   // clear bit 3 + component of the row's mask byte: an atomic AND on the word that holds it (two clusters may meet in one byte)
                const unsigned clear = ~((8u << mc[r]) << (8u * (unsigned)(mp[r] & 3)));
                atomicAnd(reinterpret_cast<unsigned *>(rowmask + (mp[r] & ~(size_t)3)), clear);
                if (maskB) {
                    const int pk = (int)(mp[r] / (size_t)sz) + L.oz, pj = (int)((mp[r] % (size_t)sz) / (size_t)sy) + L.oy, pi = (int)(mp[r] % (size_t)sy) + L.ox;
                    const size_t b = bidx(LB, pi, pj, pk);
                    atomicAnd(reinterpret_cast<unsigned *>(maskB + (b & ~(size_t)3)), ~((8u << mc[r]) << (8u * (unsigned)(b & 3))));
                }
            }
            continue;
        }
        double A[4][5];
        for (int r = 0; r < n; r++) {
            for (int m = 0; m <= n; m++) A[r][m] = 0.0;
            for (int t = 0; t < 15; t++) {
                int ec; long eo; double coef;
                d_polish_entry(mc[r], t, mF[r], md[r], sy, sz, ec, eo, coef);
                if (coef == 0.0) continue;
                const size_t at = mp[r] + eo;
                int hit = -1;
                for (int m = 0; m < n; m++) if (mc[m] == ec && mp[m] == at) hit = m;
                if (hit >= 0) A[r][hit] += coef; else A[r][n] -= coef * (double)X[ec][at];   // (massless rows: the right-hand side is what the neighbours contribute)
            }
        }
        // Gaussian elimination with partial pivoting; a singular cluster (nothing pins the split at all) is left as it is
        bool ok = true;
        double amax = 0.0;
        for (int r = 0; r < n; r++) for (int m = 0; m < n; m++) amax = fmax(amax, fabs(A[r][m]));
        for (int col = 0; col < n && ok; col++) {
            int piv = col;
            for (int r = col + 1; r < n; r++) if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
            if (!(fabs(A[piv][col]) > 1.0e-11 * amax)) { ok = false; break; }
            if (piv != col) for (int m = 0; m <= n; m++) { const double t = A[col][m]; A[col][m] = A[piv][m]; A[piv][m] = t; }
            for (int r = col + 1; r < n; r++) {
                const double f_ = A[r][col] / A[col][col];
                for (int m = col; m <= n; m++) A[r][m] -= f_ * A[col][m];
            }
        }
        if (!ok) continue;
        double x[4];
        for (int r = n - 1; r >= 0; r--) {
            double t = A[r][n];
            for (int m = r + 1; m < n; m++) t -= A[r][m] * x[m];
            x[r] = t / A[r][r];
        }
        for (int r = 0; r < n; r++) { outRow[4 * item + r] = ((unsigned long long)mc[r] << 62) | (unsigned long long)mp[r]; outVal[4 * item + r] = (float)x[r]; }
        if (count) atomicAdd(count, n);
    }
}
// ... and the stores (a row that sits in two clusters takes the value of whichever is stored last: both were solved against the same snapshot)
__global__ __launch_bounds__(256) void k_visc_massless_write(const unsigned long long *__restrict__ list, const unsigned long long *__restrict__ outRow, const float *__restrict__ outVal,
                                                             float *__restrict__ U, float *__restrict__ V, float *__restrict__ W) {
    unsigned long long n = list[0];
    if (n > (unsigned long long)FV_POLISH_CAP) n = FV_POLISH_CAP;
    float *const X[3] = {U, V, W};
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < 4 * n; t += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long r = outRow[t];
        if (r != ~0ull) X[(int)(r >> 62)][(size_t)(r & ((1ull << 62) - 1))] = outVal[t];
    }
}

// ------------------------------------------------------------------ singular clusters: rows that are THE SAME equation (flipv_params.viscosity_massless_polish >= 0)
// A row without own volume whose six factors are zero but ONE states a single condition -- the stress on that edge (or in that cell centre) vanishes -- and every other
// such row around the same edge (up to four faces, two components) or cell centre (two faces of one component) states the same condition again: the reference's matrix is
// exactly singular there (holdout draw 9 of round 5: two rows of diagonal 4e-10, the rest of the system solved to 2e-7 by anything).  What the reference delivers is decided
// by its MIC(0) factorisation (pcgsolver.h:62-178): the pivot of the LATER of two identical rows cancels to zero, is replaced by the row's diagonal (min_diagonal_ratio
// 0.25), its couplings cancel likewise, and the preconditioned residual -- and so every search direction and the iterate -- is ZERO on that row: the first such row in the
// reference's row order (U < V < W, then k, j, i: viscositysolver.cpp:284-354) carries the condition, the later ones stay at 0.  tests/research/jump_proto.py shows it on
// the dumped system.  A Krylov loop in fp32 drifts along the null vector instead (the residual of draw 9 stalled at 1e-3 max|rhs|, velocities 0.2 ... 0.9 max|u| off).
// Here: BEFORE the set-up kernel, every later row of such a cluster is listed and its face state set to ST_ELIM for the duration of k_visc_setup -- no row, no
// right-hand-side term, velocity 0, exactly what the reference's iterate holds there.
constexpr int FV_ELIM_CAP = 4096;
constexpr int FV_FLOAT_CAP = 8192;   // candidates of floating sets (massless rows not grounded at once)
struct FvMem { signed char comp, di, dj, dk, slot; };
// the other faces that carry the same factor, per (component, slot): up to three of (component, di, dj, dk, slot there); comp -1 = none
//   U: 0 centre(c) -> U(+x) slot 1 | 1 centre(c - x) -> U(-x) 0 | 2 edgeW(c + y): U(+y) 3, V(+y) 1, V(-x + y) 0 | 3 edgeW(c): U(-y) 2, V(c) 1, V(-x) 0
//      4 edgeV(c + z): U(+z) 5, W(+z) 1, W(-x + z) 0 | 5 edgeV(c): U(-z) 4, W(c) 1, W(-x) 0
//   V: 0 edgeW(c + x): V(+x) 1, U(+x) 3, U(+x - y) 2 | 1 edgeW(c): V(-x) 0, U(c) 3, U(-y) 2 | 2 centre(c) -> V(+y) 3 | 3 centre(c - y) -> V(-y) 2
//      4 edgeU(c + z): V(+z) 5, W(+z) 3, W(-y + z) 2 | 5 edgeU(c): V(-z) 4, W(c) 3, W(-y) 2
//   W: 0 edgeV(c + x): W(+x) 1, U(+x) 5, U(+x - z) 4 | 1 edgeV(c): W(-x) 0, U(c) 5, U(-z) 4 | 2 edgeU(c + y): W(+y) 3, V(+y) 5, V(+y - z) 4 | 3 edgeU(c): W(-y) 2, V(c) 5, V(-z) 4
//      4 centre(c) -> W(+z) 5 | 5 centre(c - z) -> W(-z) 4
// (constexpr: every use below has compile-time indices -- fully unrolled loops over a template parameter --, so that nothing is indexed at run time: a kernel with dynamically
// indexed per-thread arrays goes through scratch memory, and a grid-sized launch with scratch pays ~200 us at 256^3; the run-time copy FV_MEM is for the one-workgroup kernel)
static constexpr FvMem FV_MEMC[3][6][3] = {
    {{{0, 1, 0, 0, 1}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}, {{0, -1, 0, 0, 0}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}},
     {{0, 0, 1, 0, 3}, {1, 0, 1, 0, 1}, {1, -1, 1, 0, 0}}, {{0, 0, -1, 0, 2}, {1, 0, 0, 0, 1}, {1, -1, 0, 0, 0}},
     {{0, 0, 0, 1, 5}, {2, 0, 0, 1, 1}, {2, -1, 0, 1, 0}}, {{0, 0, 0, -1, 4}, {2, 0, 0, 0, 1}, {2, -1, 0, 0, 0}}},
    {{{1, 1, 0, 0, 1}, {0, 1, 0, 0, 3}, {0, 1, -1, 0, 2}}, {{1, -1, 0, 0, 0}, {0, 0, 0, 0, 3}, {0, 0, -1, 0, 2}},
     {{1, 0, 1, 0, 3}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}, {{1, 0, -1, 0, 2}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}},
     {{1, 0, 0, 1, 5}, {2, 0, 0, 1, 3}, {2, 0, -1, 1, 2}}, {{1, 0, 0, -1, 4}, {2, 0, 0, 0, 3}, {2, 0, -1, 0, 2}}},
    {{{2, 1, 0, 0, 1}, {0, 1, 0, 0, 5}, {0, 1, 0, -1, 4}}, {{2, -1, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, -1, 4}},
     {{2, 0, 1, 0, 3}, {1, 0, 1, 0, 5}, {1, 0, 1, -1, 4}}, {{2, 0, -1, 0, 2}, {1, 0, 0, 0, 5}, {1, 0, 0, -1, 4}},
     {{2, 0, 0, 1, 5}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}, {{2, 0, 0, -1, 4}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}}};
__device__ const FvMem FV_MEM[3][6][3] = {
    {{{0, 1, 0, 0, 1}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}, {{0, -1, 0, 0, 0}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}},
     {{0, 0, 1, 0, 3}, {1, 0, 1, 0, 1}, {1, -1, 1, 0, 0}}, {{0, 0, -1, 0, 2}, {1, 0, 0, 0, 1}, {1, -1, 0, 0, 0}},
     {{0, 0, 0, 1, 5}, {2, 0, 0, 1, 1}, {2, -1, 0, 1, 0}}, {{0, 0, 0, -1, 4}, {2, 0, 0, 0, 1}, {2, -1, 0, 0, 0}}},
    {{{1, 1, 0, 0, 1}, {0, 1, 0, 0, 3}, {0, 1, -1, 0, 2}}, {{1, -1, 0, 0, 0}, {0, 0, 0, 0, 3}, {0, 0, -1, 0, 2}},
     {{1, 0, 1, 0, 3}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}, {{1, 0, -1, 0, 2}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}},
     {{1, 0, 0, 1, 5}, {2, 0, 0, 1, 3}, {2, 0, -1, 1, 2}}, {{1, 0, 0, -1, 4}, {2, 0, 0, 0, 3}, {2, 0, -1, 0, 2}}},
    {{{2, 1, 0, 0, 1}, {0, 1, 0, 0, 5}, {0, 1, 0, -1, 4}}, {{2, -1, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, -1, 4}},
     {{2, 0, 1, 0, 3}, {1, 0, 1, 0, 5}, {1, 0, 1, -1, 4}}, {{2, 0, -1, 0, 2}, {1, 0, 0, 0, 5}, {1, 0, 0, -1, 4}},
     {{2, 0, 0, 1, 5}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}, {{2, 0, 0, -1, 4}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 0}}}};
// the six control volumes behind the factors of face (COMP, p), order: right, left, top, bottom, front, back
template <int COMP>
__device__ __forceinline__ void d_vol6(size_t p, long sy, long sz, const float *__restrict__ vC, const float *__restrict__ vEU, const float *__restrict__ vEV, const float *__restrict__ vEW,
                                       float &a0, float &a1, float &a2, float &a3, float &a4, float &a5) {
    if (COMP == 0) { a0 = vC[p]; a1 = vC[p - 1]; a2 = vEW[p + sy]; a3 = vEW[p]; a4 = vEV[p + sz]; a5 = vEV[p]; }
    else if (COMP == 1) { a0 = vEW[p + 1]; a1 = vEW[p]; a2 = vC[p]; a3 = vC[p - sy]; a4 = vEU[p + sz]; a5 = vEU[p]; }
    else { a0 = vEV[p + 1]; a1 = vEV[p]; a2 = vEU[p + sy]; a3 = vEU[p]; a4 = vC[p]; a5 = vC[p - sz]; }
}
// which of the six factors of face (COMP, p) are non-zero, as a bit mask.  A factor is a viscosity x one of the six volumes: with ONE viscosity the volumes decide, the factors
// themselves are only formed for a viscosity field (which may vanish on an edge).
template <int COMP>
__device__ __forceinline__ unsigned d_factor_mask_t(size_t p, const Lay &L, const float *__restrict__ nu, const float *__restrict__ vC, const float *__restrict__ vEU,
                                                    const float *__restrict__ vEV, const float *__restrict__ vEW, float factor, int field) {
    const long sy = L.sy, sz = L.sz;
    float a0, a1, a2, a3, a4, a5;
    if (!field) d_vol6<COMP>(p, sy, sz, vC, vEU, vEV, vEW, a0, a1, a2, a3, a4, a5);
    else {
        const RefRowFactors F = d_ref_row_factors(nu, vC, vEU, vEV, vEW, p, sy, sz, factor);
        if (COMP == 0) { a0 = F.U[0]; a1 = F.U[1]; a2 = F.U[2]; a3 = F.U[3]; a4 = F.U[4]; a5 = F.U[5]; }
        else if (COMP == 1) { a0 = F.V[0]; a1 = F.V[1]; a2 = F.V[2]; a3 = F.V[3]; a4 = F.V[4]; a5 = F.V[5]; }
        else { a0 = F.W[0]; a1 = F.W[1]; a2 = F.W[2]; a3 = F.W[3]; a4 = F.W[4]; a5 = F.W[5]; }
    }
    return (a0 != 0.0f ? 1u : 0u) | (a1 != 0.0f ? 2u : 0u) | (a2 != 0.0f ? 4u : 0u) | (a3 != 0.0f ? 8u : 0u) | (a4 != 0.0f ? 16u : 0u) | (a5 != 0.0f ? 32u : 0u);
}
__device__ __forceinline__ unsigned d_factor_mask(int comp, size_t p, const Lay &L, const float *__restrict__ nu, const float *__restrict__ vC, const float *__restrict__ vEU,
                                                  const float *__restrict__ vEV, const float *__restrict__ vEW, float factor, int field) {
    return comp == 0 ? d_factor_mask_t<0>(p, L, nu, vC, vEU, vEV, vEW, factor, field) : (comp == 1 ? d_factor_mask_t<1>(p, L, nu, vC, vEU, vEV, vEW, factor, field)
                                                                                                   : d_factor_mask_t<2>(p, L, nu, vC, vEU, vEV, vEW, factor, field));
}
__device__ __forceinline__ bool d_is_row_face_p(const uint8_t *__restrict__ stc, int comp, size_t p, int i, int j, int k, const Lay &L) { return d_row_range(comp, i, j, k, L) && stc[p] == ST_FLUID; }
__device__ __forceinline__ bool d_is_row_face(int comp, size_t p, int i, int j, int k, const Lay &L, const uint8_t *const st[3]) { return d_row_range(comp, i, j, k, L) && st[comp][p] == ST_FLUID; }
// FLOATING rows.  A row without own volume is tied to the rest of the system through the rows it shares a stress term with; a connected set of such rows none of which shares a
// term with a row that HAS own volume, or with a solid face, is a system of its own -- singular in the exact operator (any constant solves it), zero right-hand side, and in the
// reference's operator held at exactly 0 by the rounding defect of a diagonal (or, where that vanishes, by PCG's zero start): holdout draw 35 of round 6's sweep, three W faces in a
// column above a speck of liquid smaller than a control volume.  The multigrid's prolongation leaks a value into such rows that nothing in the iteration can take out again (1e-1 of
// max|u| on 159 faces there, status 0).  "Grounded at once": the row shares a term with a solid face or with a row that has own volume.
__device__ __forceinline__ bool d_grounded_at_once(int comp, size_t p, int i, int j, int k, unsigned fm, const Lay &L, const uint8_t *const st[3], const float *const vol[3]) {
    const long sy = L.sy, sz = L.sz;
    for (int t = 0; t < 6; t++) {
        if (!((fm >> t) & 1u)) continue;
        for (int m = 0; m < 3; m++) {
            const FvMem e = FV_MEM[comp][t][m];
            if (e.comp < 0) continue;
            const size_t q = p + e.di + e.dj * sy + e.dk * sz;
            if (st[e.comp][q] == ST_SOLID) return true;
            if (d_is_row_face(e.comp, q, i + e.di, j + e.dj, k + e.dk, L, st) && vol[e.comp][q] > 0.0f) return true;
        }
    }
    return false;
}
// the same with everything known at compile time (the grid-sized kernel)
template <int COMP>
__device__ __forceinline__ bool d_grounded_at_once_t(size_t p, int i, int j, int k, unsigned fm, const Lay &L, const uint8_t *__restrict__ stU, const uint8_t *__restrict__ stV,
                                                     const uint8_t *__restrict__ stW, const float *__restrict__ volU, const float *__restrict__ volV, const float *__restrict__ volW) {
    const long sy = L.sy, sz = L.sz;
    bool g = false;
#pragma unroll
    for (int t = 0; t < 6; t++) {
#pragma unroll
        for (int m = 0; m < 3; m++) {
            constexpr int ec = 0;
            (void)ec;
            const int mc = FV_MEMC[COMP][t][m].comp;
            if (mc < 0) continue;
            if (!((fm >> t) & 1u) || g) continue;
            const uint8_t *stc = mc == 0 ? stU : (mc == 1 ? stV : stW);
            const float *vc = mc == 0 ? volU : (mc == 1 ? volV : volW);
            const size_t q = p + FV_MEMC[COMP][t][m].di + FV_MEMC[COMP][t][m].dj * sy + FV_MEMC[COMP][t][m].dk * sz;
            const uint8_t sq = stc[q];
            if (sq == ST_SOLID) g = true;
            else if (sq == ST_FLUID && d_row_range(mc, i + FV_MEMC[COMP][t][m].di, j + FV_MEMC[COMP][t][m].dj, k + FV_MEMC[COMP][t][m].dk, L) && vc[q] > 0.0f) g = true;
        }
    }
    return g;
}
// is face (MC, p) a row without own volume whose ONLY non-zero factor sits in slot SLOT?  (compile-time component and slot)
template <int MC, int SLOT>
__device__ __forceinline__ bool d_single_factor_at(size_t p, int i, int j, int k, const Lay &L, const uint8_t *__restrict__ stc, const float *__restrict__ volc, const float *__restrict__ nu,
                                                   const float *__restrict__ vC, const float *__restrict__ vEU, const float *__restrict__ vEV, const float *__restrict__ vEW, float factor, int field) {
    if (!d_row_range(MC, i, j, k, L) || stc[p] != ST_FLUID || volc[p] != 0.0f) return false;
    return d_factor_mask_t<MC>(p, L, nu, vC, vEU, vEV, vEW, factor, field) == (1u << SLOT);
}
// one component of k_visc_singular_find
template <int COMP>
__device__ __forceinline__ void d_singular_find_comp(const Lay &L, size_t c, int i, int j, int k, const uint8_t *__restrict__ stU, const uint8_t *__restrict__ stV, const uint8_t *__restrict__ stW,
                                                     const float *__restrict__ volU, const float *__restrict__ volV, const float *__restrict__ volW, const float *__restrict__ nu,
                                                     const float *__restrict__ vC, const float *__restrict__ vEU, const float *__restrict__ vEV, const float *__restrict__ vEW, float factor, int field,
                                                     unsigned long long *__restrict__ list, unsigned long long *__restrict__ flist) {
    const uint8_t *stc = COMP == 0 ? stU : (COMP == 1 ? stV : stW);
    const float *volc = COMP == 0 ? volU : (COMP == 1 ? volV : volW);
    if (!d_row_range(COMP, i, j, k, L) || stc[c] != ST_FLUID || volc[c] != 0.0f) return;     // not a massless fluid face
    const unsigned fm = d_factor_mask_t<COMP>(c, L, nu, vC, vEU, vEV, vEW, factor, field);
    if (!fm) return;                                                                            // no volume around it: no row
    const long sy = L.sy, sz = L.sz;
    if (!d_grounded_at_once_t<COMP>(c, i, j, k, fm, L, stU, stV, stW, volU, volV, volW)) {      // a candidate for a floating set (k_visc_floating)
        // (one atomic per wave: the second layer of a liquid's fringe is full of such rows -- thousands at 256^3 --, and that many atomics on one address took 250 us)
        const unsigned long long act = __ballot(1);
        const int lane = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63;
        const int leader = __ffsll((long long)act) - 1, rank = __popcll(act & ((1ull << lane) - 1ull));
        unsigned long long base = 0ull;
        if (lane == leader) base = atomicAdd(flist, (unsigned long long)__popcll(act));
        base = __shfl(base, leader, 64);
        const unsigned long long at = base + (unsigned long long)rank;
        if (at < (unsigned long long)FV_FLOAT_CAP) flist[1 + at] = ((unsigned long long)c << 2) | (unsigned long long)COMP;
    }
    if (fm & (fm - 1u)) return;                                                                 // more than one factor: not a row that can repeat another
    bool later = false;   // is there such a row of the same cluster BEFORE this one in the reference's row order?
#pragma unroll
    for (int t = 0; t < 6; t++) {
        if (fm != (1u << t)) continue;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            constexpr int dummy = 0;
            (void)dummy;
            if (FV_MEMC[COMP][t][m].comp < 0) continue;
            // (compile-time member: component, offset and slot fold to constants)
#define FV_MEMBER_CHECK(MC_, SL_)                                                                                                                              \
            if (FV_MEMC[COMP][t][m].comp == MC_ && FV_MEMC[COMP][t][m].slot == SL_ && !later) {                                                                 \
                const size_t p = c + FV_MEMC[COMP][t][m].di + FV_MEMC[COMP][t][m].dj * sy + FV_MEMC[COMP][t][m].dk * sz;                                        \
                if (d_single_factor_at<MC_, SL_>(p, i + FV_MEMC[COMP][t][m].di, j + FV_MEMC[COMP][t][m].dj, k + FV_MEMC[COMP][t][m].dk, L,                     \
                                                 MC_ == 0 ? stU : (MC_ == 1 ? stV : stW), MC_ == 0 ? volU : (MC_ == 1 ? volV : volW), nu, vC, vEU, vEV, vEW, factor, field)) \
                    later = MC_ < COMP || (MC_ == COMP && p < c);                                                                                               \
            }
            FV_MEMBER_CHECK(0, 0) FV_MEMBER_CHECK(0, 1) FV_MEMBER_CHECK(0, 2) FV_MEMBER_CHECK(0, 3) FV_MEMBER_CHECK(0, 4) FV_MEMBER_CHECK(0, 5)
            FV_MEMBER_CHECK(1, 0) FV_MEMBER_CHECK(1, 1) FV_MEMBER_CHECK(1, 2) FV_MEMBER_CHECK(1, 3) FV_MEMBER_CHECK(1, 4) FV_MEMBER_CHECK(1, 5)
            FV_MEMBER_CHECK(2, 0) FV_MEMBER_CHECK(2, 1) FV_MEMBER_CHECK(2, 2) FV_MEMBER_CHECK(2, 3) FV_MEMBER_CHECK(2, 4) FV_MEMBER_CHECK(2, 5)
#undef FV_MEMBER_CHECK
        }
    }
    if (!later) return;
    const unsigned long long at = atomicAdd(list, 1ull);
    if (at < (unsigned long long)FV_ELIM_CAP) list[1 + at] = ((unsigned long long)c << 2) | (unsigned long long)COMP;
}
__global__ void k_visc_singular_find(Lay L, const uint8_t *__restrict__ stU, const uint8_t *__restrict__ stV, const uint8_t *__restrict__ stW, const float *__restrict__ volU,
                                     const float *__restrict__ volV, const float *__restrict__ volW, const float *__restrict__ nu, const float *__restrict__ vC,
                                     const float *__restrict__ vEU, const float *__restrict__ vEV, const float *__restrict__ vEW, const uint8_t *__restrict__ band, float factor, int field,
                                     unsigned long long *__restrict__ list, unsigned long long *__restrict__ flist) {
    IJK_OR_RETURN(L);
    const long sy = L.sy, sz = L.sz;
    if (!(band[c] || band[c - 1] || band[c + 1] || band[c - sy] || band[c + sy] || band[c - sz] || band[c + sz])) return;   // (no volume near: no row, k_visc_setup)
    d_singular_find_comp<0>(L, c, i, j, k, stU, stV, stW, volU, volV, volW, nu, vC, vEU, vEV, vEW, factor, field, list, flist);
    d_singular_find_comp<1>(L, c, i, j, k, stU, stV, stW, volU, volV, volW, nu, vC, vEU, vEV, vEW, factor, field, list, flist);
    d_singular_find_comp<2>(L, c, i, j, k, stU, stV, stW, volU, volV, volW, nu, vC, vEU, vEV, vEW, factor, field, list, flist);
}
// One workgroup over the candidates (massless rows that share no term with mass or a wall): a candidate becomes grounded once a row it shares a term with is -- a massless row
// grounded at once (recomputed here), or a candidate grounded in an earlier round (marks: one byte per index, bits 0-2, cleared again at the end) --; rounds until nothing changes.
// What stays ungrounded floats: appended to the elimination list (state ST_ELIM, velocity 0 for this solve).
__global__ __launch_bounds__(1024) void k_visc_floating(Lay L, const uint8_t *__restrict__ stU, const uint8_t *__restrict__ stV, const uint8_t *__restrict__ stW, const float *__restrict__ volU,
                                                        const float *__restrict__ volV, const float *__restrict__ volW, const float *__restrict__ nu, const float *__restrict__ vC,
                                                        const float *__restrict__ vEU, const float *__restrict__ vEV, const float *__restrict__ vEW, float factor, int field,
                                                        const unsigned long long *__restrict__ flist, uint8_t *__restrict__ mark, unsigned long long *__restrict__ elist) {
    __shared__ int changed;
    const uint8_t *const st[3] = {stU, stV, stW};
    const float *const vol[3] = {volU, volV, volW};
    const long sy = L.sy, sz = L.sz;
    unsigned long long n = flist[0];
    if (n > (unsigned long long)FV_FLOAT_CAP) n = FV_FLOAT_CAP;
    if (n == 0) return;
    for (int round = 0; round < 256; round++) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (unsigned long long t = threadIdx.x; t < n; t += blockDim.x) {
            const size_t c = (size_t)(flist[1 + t] >> 2);
            const int comp = (int)(flist[1 + t] & 3ull);
            if ((mark[c] >> comp) & 1u) continue;
            const int k = (int)(c / (size_t)sz) + L.oz, j = (int)((c % (size_t)sz) / (size_t)sy) + L.oy, i = (int)(c % (size_t)sy) + L.ox;
            const unsigned fm = d_factor_mask(comp, c, L, nu, vC, vEU, vEV, vEW, factor, field);
            // (a block context sees the rows of its box: a set that reaches a cut face may be grounded beyond it -- candidates within three entries of a cut count as grounded, on every
            // rank that sees them alike)
            bool g = (i - 3 < L.olo[0] && L.olo[0] > 0) || (i + 3 >= L.ohi[0] && L.ohi[0] < L.I) || (j - 3 < L.olo[1] && L.olo[1] > 0) || (j + 3 >= L.ohi[1] && L.ohi[1] < L.J) ||
                     (k - 3 < L.olo[2] && L.olo[2] > 0) || (k + 3 >= L.ohi[2] && L.ohi[2] < L.K);
            for (int s6 = 0; s6 < 6 && !g; s6++) {
                if (!((fm >> s6) & 1u)) continue;
                for (int m = 0; m < 3 && !g; m++) {
                    const FvMem e = FV_MEM[comp][s6][m];
                    if (e.comp < 0) continue;
                    const size_t q = c + e.di + e.dj * sy + e.dk * sz;
                    const int qi = i + e.di, qj = j + e.dj, qk = k + e.dk;
                    if (!d_is_row_face(e.comp, q, qi, qj, qk, L, st)) continue;
                    if ((mark[q] >> e.comp) & 1u) { g = true; break; }
                    if (round == 0) {   // (a neighbour that was grounded at once is not on the list: its state is recomputed, in the first round only -- it cannot change)
                        const unsigned fq = d_factor_mask(e.comp, q, L, nu, vC, vEU, vEV, vEW, factor, field);
                        if (fq && d_grounded_at_once(e.comp, q, qi, qj, qk, fq, L, st, vol)) g = true;
                    }
                }
            }
            if (g) { atomicOr(reinterpret_cast<unsigned *>(mark + (c & ~(size_t)3)), (1u << comp) << (8u * (unsigned)(c & 3))); changed = 1; }
        }
        __syncthreads();
        const int ch = changed;
        __syncthreads();
        if (!ch) break;
    }
    for (unsigned long long t = threadIdx.x; t < n; t += blockDim.x) {
        const size_t c = (size_t)(flist[1 + t] >> 2);
        const int comp = (int)(flist[1 + t] & 3ull);
        if (!((mark[c] >> comp) & 1u)) {
            const unsigned long long at = atomicAdd(elist, 1ull);
            if (at < (unsigned long long)FV_ELIM_CAP) elist[1 + at] = ((unsigned long long)c << 2) | (unsigned long long)comp;
        }
    }
    __syncthreads();
    for (unsigned long long t = threadIdx.x; t < n; t += blockDim.x) {   // the marks go back to zero: the array is only ever cleared where it was written
        const size_t c = (size_t)(flist[1 + t] >> 2);
        atomicAnd(reinterpret_cast<unsigned *>(mark + (c & ~(size_t)3)), ~(0xffu << (8u * (unsigned)(c & 3))));
    }
}
// the listed faces: state ST_ELIM and velocity 0 before k_visc_setup (APPLY), ST_FLUID again after it (the states are kept while the solid SDF is unchanged)
template <bool APPLY>
__global__ void k_visc_singular_apply(const unsigned long long *__restrict__ list, uint8_t *__restrict__ stU, uint8_t *__restrict__ stV, uint8_t *__restrict__ stW,
                                      float *__restrict__ U, float *__restrict__ V, float *__restrict__ W) {
    unsigned long long n = list[0];
    if (n > (unsigned long long)FV_ELIM_CAP) n = FV_ELIM_CAP;
    for (unsigned long long t = threadIdx.x; t < n; t += blockDim.x) {
        const size_t c = (size_t)(list[1 + t] >> 2);
        const int comp = (int)(list[1 + t] & 3ull);
        uint8_t *st = comp == 0 ? stU : (comp == 1 ? stV : stW);
        st[c] = APPLY ? ST_ELIM : ST_FLUID;
        if (APPLY) { float *X = comp == 0 ? U : (comp == 1 ? V : W); X[c] = 0.0f; }
    }
}

// ------------------------------------------------------------------ strongly coupled pairs of rows (the multigrid loops' additive correction, k_viscosity_mg.hip: k_vmg_pairs)
// Two rows whose coupling is >= 0.7 of the geometric mean of their diagonals -- a row without (or almost without) own volume hanging on ONE stress term, and the row
// that shares that term -- carry a mode of Jacobi-scaled eigenvalue 1 - |coupling|: 2e-5 ... 1e-3 on a viscosity FIELD with a jump (holdout draws 9, 11 of round 5:
// ~10 such modes, each on 2-3 rows, and CG -- diagonal or multigrid, fp32 or fp64 -- sits on a residual plateau until it has resolved every one of them: 295
// Jacobi-PCG iterations against 74 with the pairs' 2 x 2 blocks added to the preconditioner, tests/research/jump_proto.py), and the same modes are what the velocity
// criterion waits for in ordinary scenes (31 pairs on the 64^3 bunny at rest).  The geometric hierarchy does not see them: they are features of single faces.
// Here: listed once per solve; every V-cycle adds each pair's weak mode, v v^T r / lambda (symmetric positive semi-definite; a row may sit in two pairs).
constexpr float FV_PAIR_THETA = 0.7f;
__global__ void k_visc_pairs_find(Lay L, const float *__restrict__ nu, const float *__restrict__ vC, const float *__restrict__ vEU, const float *__restrict__ vEV,
                                  const float *__restrict__ vEW, const float *__restrict__ volU, const float *__restrict__ volV, const float *__restrict__ volW,
                                  const uint8_t *__restrict__ rowmask, float factor, int brick, int swz, Lay LB, unsigned *__restrict__ count, VPair *__restrict__ list, float w0, float w1, float lam_floor) {
    IJK_OR_RETURN(L);
    const unsigned mk = rowmask[c];
    if (!(mk & 7u)) return;
    if (i < L.olo[0] || i >= L.ohi[0] || j < L.olo[1] || j >= L.ohi[1] || k < L.olo[2] || k >= L.ohi[2]) return;   // (a block context: pairs of two OWNED rows)
    const long sy = L.sy, sz = L.sz;
    const float *const vol[3] = {volU, volV, volW};
    const RefRowFactors F = d_ref_row_factors(nu, vC, vEU, vEV, vEW, c, sy, sz, factor);
    for (int comp = 0; comp < 3; comp++) {
        if (!((mk >> comp) & 1u)) continue;
        const float *f = comp == 0 ? F.U : (comp == 1 ? F.V : F.W);
        const double dp = (double)vol[comp][c] + (double)f[0] + (double)f[1] + (double)f[2] + (double)f[3] + (double)f[4] + (double)f[5];
        float fm = 0.0f;
#pragma unroll
        for (int t = 0; t < 6; t++) fm = fmaxf(fm, f[t]);
        if (!((double)fm >= (double)(FV_PAIR_THETA * FV_PAIR_THETA) * dp) || !(dp > 0.0)) continue;   // (|a| >= theta sqrt(dp dq) and dq >= |a| need |a| >= theta^2 dp)
        for (int t = 1; t < 15; t++) {
            int ec; long eo; double coef;
            d_polish_entry(comp, t, f, (float)dp, sy, sz, ec, eo, coef);
            if (!(fabs(coef) >= (double)(FV_PAIR_THETA * FV_PAIR_THETA) * dp)) continue;
            const size_t cq = c + eo;
            if (ec < comp || (ec == comp && cq <= c)) continue;            // (each pair once: from its first row in (component, index) order)
            if (!((rowmask[cq] >> ec) & 1u)) continue;
            const int qk = (int)(cq / (size_t)sz) + L.oz, qj = (int)((cq % (size_t)sz) / (size_t)sy) + L.oy, qi = (int)(cq % (size_t)sy) + L.ox;
            if (qi < L.olo[0] || qi >= L.ohi[0] || qj < L.olo[1] || qj >= L.ohi[1] || qk < L.olo[2] || qk >= L.ohi[2]) continue;
            const RefRowFactors G = d_ref_row_factors(nu, vC, vEU, vEV, vEW, cq, sy, sz, factor);
            const float *g = ec == 0 ? G.U : (ec == 1 ? G.V : G.W);
            const double dq = (double)vol[ec][cq] + (double)g[0] + (double)g[1] + (double)g[2] + (double)g[3] + (double)g[4] + (double)g[5];
            if (!(coef * coef >= (double)(FV_PAIR_THETA * FV_PAIR_THETA) * dp * dq)) continue;
            // The block's WEAK mode only: in Jacobi-scaled variables the block is [[1, s], [s, 1]], s = coef / sqrt(dp dq), with the eigenpair 1 - |s|, (1, -sign s) / sqrt 2; the
            // correction is v v^T / lambda for it.  (The whole inverse B^-1 also counts the block's strong mode, which the V-cycle already resolves, a second time: the
            // preconditioned spectrum then reaches 2 on every such pair and every solve takes 25-40 % more iterations -- measured, profiles/r6/pairs_diag.log.)
            const double gm = sqrt(dp * dq);
            double lam = (gm - fabs(coef)) / gm;
            if (!(lam >= (double)lam_floor)) lam = (double)lam_floor;       // (an almost singular block: the correction along its near-null vector is capped at 1 / (floor x diagonal))
            const double sg = coef > 0.0 ? -1.0 : 1.0;
            // ... and only what the cycle's own sweeps leave of it: the V(2,2) smoother alone contracts a mode of eigenvalue lambda by p = ((1 - w0 lambda)(1 - w1 lambda))^2
            // (the coarse levels do nothing for a mode that lives on two faces), so the cycle already applies (1 - p) / lambda; the pair adds p / lambda.  With the plain
            // 1 / lambda every pair of moderate weakness (lambda 0.03 ... 0.3) became an eigenvalue of its own between 1 and 2 above the preconditioned spectrum, one more CG
            // iteration each: 81 instead of 65 iterations on the 64^3 bunny at nu = 200 (profiles/r6/pairs_diag_rank1.log).
            // Is the weak mode of the BLOCK a weak mode of the SYSTEM?  v = D^-1/2 (1, sg) / sqrt 2 has the Rayleigh quotient lambda whatever the rows' other couplings are, but
            // S v (S = D^-1/2 A D^-1/2) also has entries on the rows' OTHER neighbours; where their norm o exceeds lambda the pair is no eigenvector of anything -- the sweeps
            // spread it to the neighbours and the cycle resolves it (the ordinary scenes' pairs: lambda 4e-4 ... 0.3 with o = 2e-2 ... 0.5), and adding its projector puts one more
            // eigenvalue between 1 and 2 on top of the preconditioned spectrum: one more CG iteration per pair (64^3 bunny, 40 pairs: 137 -> 195 iterations in the scipy model
            // of tests/research/jump_proto.py, 65 -> 85 on the device).  On a viscosity field with a jump o is 0.1 ... 1 lambda: those pairs stay (draw 11: 906 -> 390, draw 9:
            // 112 -> 59 iterations in the model).  Rule: o <= lambda.  The two rows share neighbours (the other faces of the same edge), whose entries CANCEL in S v: both rows' entries are merged.
            {
                const double vp = 1.0 / sqrt(2.0 * dp), vq = sg / sqrt(2.0 * dq);
                double o2 = 0.0;
                for (int side = 0; side < 2; side++) {
                    const int ca = side ? ec : comp, cb = side ? comp : ec;           // the row whose entries are walked, the other row of the pair
                    const size_t ia = side ? cq : c, ib = side ? c : cq;
                    const float *fa = side ? g : f, *fb = side ? f : g;
                    const double da = side ? dq : dp, db = side ? dp : dq, va = side ? vq : vp, vb = side ? vp : vq;
                    for (int u = 1; u < 15; u++) {
                        int jc; long jo; double cj;
                        d_polish_entry(ca, u, fa, (float)da, sy, sz, jc, jo, cj);
                        if (cj == 0.0) continue;
                        const size_t ij = ia + jo;
                        if (jc == cb && ij == ib) continue;                            // (the pair's own coupling)
                        double w = cj * va;
                        bool shared = false;
                        for (int u2 = 1; u2 < 15; u2++) {
                            int kc; long ko; double ck;
                            d_polish_entry(cb, u2, fb, (float)db, sy, sz, kc, ko, ck);
                            if (ck != 0.0 && kc == jc && ib + ko == ij) { w += ck * vb; shared = true; }
                        }
                        if (shared && side) continue;                                  // (counted from the first row's side)
                        if (!((rowmask[ij] >> jc) & 1u)) continue;                     // (no row there: a solid or empty face, nothing for S v to land on)
                        const RefRowFactors H = d_ref_row_factors(nu, vC, vEU, vEV, vEW, ij, sy, sz, factor);
                        const float *h = jc == 0 ? H.U : (jc == 1 ? H.V : H.W);
                        const double dj = (double)vol[jc][ij] + (double)h[0] + (double)h[1] + (double)h[2] + (double)h[3] + (double)h[4] + (double)h[5];
                        if (dj > 0.0) o2 += w * w / dj;
                    }
                }
                if (!(o2 <= lam * lam)) continue;
            }
            const double pl = (1.0 - (double)w0 * lam) * (1.0 - (double)w1 * lam);
            const double gain = pl * pl / lam;
            const unsigned at = atomicAdd(count, 1u);
            if (at >= (unsigned)FV_PAIR_CAP) continue;
            VPair P;
            P.ir0 = (unsigned)(brick ? bidx(LB, i, j, k) : (swz ? sidx(L, i, j, k) : c));
            P.ir1 = (unsigned)(brick ? bidx(LB, qi, qj, qk) : (swz ? sidx(L, qi, qj, qk) : cq));
            P.iz0 = (unsigned)(brick ? bidx(LB, i, j, k) : c);
            P.iz1 = (unsigned)(brick ? bidx(LB, qi, qj, qk) : cq);
            P.comps = (unsigned)comp | ((unsigned)ec << 2);
            P.i00 = (float)(0.5 * gain / dp); P.i01 = (float)(0.5 * gain * sg / gm); P.i11 = (float)(0.5 * gain / dq);
            list[at] = P;
        }
    }
}

// What every rank of a communicator must decide alike from (the preconditioner, the stiffness rule, the vector type of an FP64 solve): the viscosity field's
// facts over ALL ranks, one small all-gather at the start of every solve -- before anything that chooses a sequence of collectives.
static int visc_gather_field_facts(flipv_context *c) {
    if (c->comm && c->comm->nranks > 1) {   // a rank whose box holds no viscous node must still take part in every collective of the solve
        const double mine[3] = {c->viscosity_nonzero ? 1.0 : 0.0, (double)c->viscosity_max, (double)c->viscosity_min};   // (AUTO's stiffness rule must come out the same on every rank)
        double all[3 * NSLOT];
        const int rcv = fv_allgather_f64(c, mine, 3, all);
        if (rcv) return rcv;
        double nz = 0.0, vm = all[1], vlo = all[2];
        for (int r = 0; r < c->comm->nranks; r++) { nz = fmax(nz, all[3 * r]); vm = fmax(vm, all[3 * r + 1]); vlo = fmin(vlo, all[3 * r + 2]); }
        c->viscosity_nonzero_any = nz > 0.0;
        c->viscosity_max_any = (float)vm;
        c->vPerRowFactors = vlo != vm ? 1 : 0;
        c->vZeroRegion = (vm > 0.0 && vlo * 1.0e4 < vm) ? 1 : 0;
    } else {
        c->viscosity_nonzero_any = c->viscosity_nonzero; c->viscosity_max_any = c->viscosity_max;
        c->vPerRowFactors = c->viscosity_min != c->viscosity_max ? 1 : 0;
        c->vZeroRegion = (c->viscosity_max > 0.0f && (double)c->viscosity_min * 1.0e4 < (double)c->viscosity_max) ? 1 : 0;   // (a contrast beyond 1e4 -- zero included --: two correction stages, below)
    }
    return FLIPV_OK;
}

// What one viscosity solve has decided before its set-up kernel runs (visc_plan), shared by the steps of viscosity_solve_t.
struct ViscPlan {
    int cap = 0;                       // iteration cap of the whole solve
    PcgScal sc;                        // the loops' scalars (views into ctx->d_scal), stop flag, guards
    double *bmax = nullptr;            // device: max|rhs|, max|u| over the rows
    bool fieldSolve = false;           // the viscosity is a FIELD: pairs' weak modes in the preconditioner, the wider stall guard
    Lay R0, R1;                        // the liquid's range with margins 0 / 1
    int fullVol = 0;
    float factor = 0.0f;               // dt / dx^2 (viscositysolver.cpp:379-380)
    int forcedLayout = 0;
    bool brickOk = false, swzOk = false, mgPossible = false, mgPlanned = false;
    int precNow = 0, refDiag = 1;
    double stiffSolve = 0.0;           // nu_max dt / dx^2
    bool stage1Early = false, predict = false;
    double bnormAll = 0.0, umaxAll = 0.0, rowsAll = 0.0, fillLocal = 0.0;   // visc_layout_and_lists: max|rhs|, max|u|, rows over all ranks; the rank's row fill
};
// PLAN: scalars, what follows the viscosity field, the geometry of the system (face states, band, control volumes), which preconditioner, which operator, whether stage 1 stops early.
template <typename T>
static int visc_plan(flipv_context *c, float dt, ViscPlan &P) {
    P.cap = c->prm.viscosity_max_iterations;
    const int cap = P.cap;
    int rc = fv_scal_reserve(c, cap);
    if (rc) return rc;
    PcgScal &sc = P.sc;
    if ((rc = fv_pcg_reset(c, cap, false, &sc, &P.bmax, c->d_flags + 2))) return rc;   // (scalars, conv = -1, the row counter, the guard, the counters: one launch)
    sc.tol_inclusive = 1;
    sc.tol = 0.0;
    // A viscosity FIELD (flipv_set_viscosity with values that differ; decided alike on every rank: visc_gather_field_facts) is where a few isolated small eigenvalues sit under
    // the multigrid-preconditioned spectrum -- pairs of faces hanging on one stress term across a jump of the viscosity -- and where CG's max|r| rebounds by 20-50 x while it resolves
    // them one plateau at a time (holdout draws 9, 11 of round 5).  Two things follow the field: the stall guard's factor (16 x reads those rebounds as a blow-up and
    // ended every correction stage of draw 9 after 10-40 iterations) and the pairs' weak modes in the preconditioner (k_visc_pairs_find).  With ONE viscosity both stay as they were:
    // the same pairs exist there (51 at 256^3 during the bunny's fall), resolving them costs 7 + 3 iterations per solve (58 against 48 in bench.py's window) and buys nothing the
    // velocity criterion and the cluster solve after the loop do not already deliver (profiles/r6/bench_ab.log).
    P.fieldSolve = c->vPerRowFactors != 0;
    const bool fieldSolve = P.fieldSolve;
    sc.stall_ratio = c->prm.stall_guard_ratio > 0.0f ? (double)c->prm.stall_guard_ratio : (fieldSolve ? FV_STALL_RATIO_FIELD : FV_STALL_RATIO);

    // Everything up to the factors is evaluated redundantly on the halo planes a neighbour-owned row would need
    // (inputs: phi with a 4-plane halo, the replicated solid SDF), so the setup needs no exchange of its own.
    // The GEOMETRY of the system -- face states, band mask, the seven control-volume lattices: functions of the liquid SDF and the solid SDF alone (visc_geometry).
    P.R0 = fv_range_liquid(c, 0, 4); P.R1 = fv_range_liquid(c, 1, 4);
    P.fullVol = c->bandPrevValid ? 0 : 1;  // volumes and factors are stored only where the band is or was in the previous solve
    if ((rc = visc_geometry(c, c->stream))) return rc;
    const float invdx = 1.0f / c->dx;
    P.factor = dt * invdx * invdx;  // viscositysolver.cpp:379-380
    c->vFactorNow = P.factor;
    // ---- which layout the solver's arrays take (flipv_internal.h: VLAYOUT_*).  Bricks on sparse liquids of a single-domain context, the
    // plain planes (with the swizzled own-index arrays under the 16-lane tile geometry) otherwise.  How sparse the liquid is is only known
    // after the set-up kernel has counted the rows, so the set-up runs in the previous solve's layout and is repeated on the rare solve
    // where the choice changes.
    P.forcedLayout = c->prm.viscosity_layout;
    const int forcedLayout = P.forcedLayout;   // 0 auto, 1 plain, 2 plain / swizzled, 3 brick
    P.brickOk = c->prm.viscosity_lane_width != 2 && forcedLayout != 1 && forcedLayout != 2;   // (block contexts too: the halo exchange addresses either layout, flipv_comm.h: HaloArray::lay)
    P.swzOk = forcedLayout != 1;   // (also under the multigrid: its own kernels address diag / x / r / q / own volumes through sidx, its sweep vectors stay plain)
    // the preconditioner of this solve (the multigrid needs fp32 vectors over a whole, single-rank index space)
    P.mgPossible = std::is_same<T, float>::value && c->prm.viscosity_lane_width != 2 && !c->vNoMultigridOnce;   // (block contexts too: a rank-local hierarchy, k_viscosity_mg.hip)
    const bool mgPossible = P.mgPossible;
    P.mgPlanned = mgPossible && (c->prm.viscosity_preconditioner == FLIPV_PRECOND_MULTIGRID || c->vForceMultigridOnce || c->vMixed64 ||
                                          (c->prm.viscosity_preconditioner == FLIPV_PRECOND_AUTO && fv_visc_auto_pick(c, dt)));
    if (mgPossible && c->prm.viscosity_preconditioner != FLIPV_PRECOND_DIAGONAL && !c->vmgState) {
        const int prc = fv_vmg_prepare(c);   // allocate the hierarchy now, whichever solve first uses it
        if (prc) return prc;
    }
    // sweeps on the LDS-resident coarsest level: 16, 32 on stiff systems whose last solve needed more than 60 iterations; 8 where stage 1 stops early (below).  A power of two selects the
    // Chebyshev weights, any other count plain damped Jacobi.  (The scans: HISTORY.md, "Round 6: notes moved out of viscosity_solve_t", A.)
    {
        const double stiff = (double)c->viscosity_max_any * (double)dt / ((double)c->dx * (double)c->dx);
        c->vmgSweeps = c->prm.viscosity_mg_coarsest_sweeps > 0 ? (c->prm.viscosity_mg_coarsest_sweeps + 1) / 2 * 2
                       : ((stiff > 1000.0 && (c->vLastPrec != 2 || c->vLastIts > 60)) ? 32 : 16);   // (a solve whose stage 1 stops at 1e-4 takes 8: below)
        // The packed coarse rows round an entry to 11 bits; the mass term is 1/stiff of the entries.  Measured on the 256^3 bunny: identical iteration
        // counts up to nu dt/dx^2 = 131 072 (512^3, nu = 50), but at 327 680 (256^3, nu = 500) 3-4 of 20 solves end unconverged where the fp32 rows
        // lose 2: beyond 2e5 the cycle reads the fp32 grids.
        c->vmgPackedRows = c->prm.viscosity_mg_packed_rows ? (c->prm.viscosity_mg_packed_rows > 0 ? 1 : 0) : (stiff <= 2.0e5 ? 1 : 0);
    }
    P.precNow = std::is_same<T, float>::value ? 0 : 1;
    // flipv_params.exact_viscosity_operator = 0 (default): the solve applies the reference's operator INCLUDING the rounding of its float
    // diagonal (d_ref_volume) -- at 256^3 the reference's converged answer is 7e-6 from this operator's and 1.5e-4 from the exact one's.
    P.refDiag = c->prm.exact_viscosity_operator ? 0 : 1;
    // DEFECT PREDICTOR: stage 1 solves A x = b - E u_old (u_old = the row's incoming velocity) -- one step of the fixed point x <- A^-1 (b - E x) started from u_old instead of 0 --, only
    // where stage 1 stops early; the correction stage stays (skipping it was measured and misses the bar at nu dt/dx^2 = 1.2e5).  HISTORY.md, same section, B.
    P.stiffSolve = (double)c->viscosity_max_any * (double)dt / ((double)c->dx * (double)c->dx);
    const double stiffSolve = P.stiffSolve;
    P.stage1Early = c->prm.viscosity_stage1_factor != 1.0f && stiffSolve <= (c->prm.viscosity_two_stage_max_stiffness > 0.0f ? (double)c->prm.viscosity_two_stage_max_stiffness : 1.0e6);
    const int refDiag = P.refDiag;
    const bool mgPlanned = P.mgPlanned, stage1Early = P.stage1Early;
    P.predict = refDiag && mgPlanned && std::is_same<T, float>::value && c->prm.viscosity_lane_width != 2 && c->prm.viscosity_defect_predictor >= 0 &&
                         stage1Early;
    (void)fieldSolve; (void)forcedLayout;
    return FLIPV_OK;
}
// SET-UP in one layout: factors, the rows taken out of the system (k_visc_singular_find / k_visc_floating), diagonal + right-hand side + row mask (k_visc_setup), the massless
// clusters and -- for a viscosity field under the multigrid -- the strongly coupled pairs listed; max|rhs|, max|u| and the counts read back.  Repeated on the rare solve whose layout changes.
template <typename T>
static int visc_run_setup(flipv_context *c, const ViscPlan &P, int layout, bool first) {
    const Lay &L = c->L;
    const Lay &R0 = P.R0, &R1 = P.R1;
    const int fullVol = P.fullVol, precNow = P.precNow, refDiag = P.refDiag;
    const float factor = P.factor;
    const bool fieldSolve = P.fieldSolve, mgPlanned = P.mgPlanned, predict = P.predict;
    double *const bmax = P.bmax;
        const bool brick = layout == VLAYOUT_BRICK;
        if (c->viscStateValid && (c->vLayout == VLAYOUT_BRICK) != brick) {
            int zr = visc_zero_solver_arrays(c);
            if (zr) return zr;
            c->viscStateValid = 0;
            c->facValid = 0;
        }
        // the setup kernel only stores where a row is or was; the first solve, a change of vector precision (the buffers
        // are shared), of the layout or of the slab make it store everywhere
        const int full = (c->viscStateValid && c->viscStatePrec == precNow && c->vLayout == layout) ? 0 : 1;
        const int facFull = (fullVol || !c->facValid || !first) ? 1 : 0;
        c->viscStateValid = 1; c->viscStatePrec = precNow; c->vLayout = layout; c->vSwz = layout == VLAYOUT_SWZ; c->facValid = 1;
        hipLaunchKernelGGL(k_visc_factors, GRID3(R1), 0, c->stream, R1, c->visc, c->volC, c->volEU, c->volEV, c->volEW, c->fC,
                           c->fEU, c->fEV, c->fEW, factor, c->validCells, c->bandPrev, facFull, brick ? 1 : 0, c->LB);
        if (first) {
            const size_t off = plane_off(L, R1.kb), cnt = (size_t)(R1.ke - R1.kb) * L.sz;
            HIPCHK(c, hipMemcpyAsync(c->bandPrev + off, c->validCells + off, cnt, hipMemcpyDeviceToDevice, c->stream));
            c->bandPrevValid = 1;
        }
        PcgSys<T, 3> vs = visc_sys<T>(c);
        const bool elim = !c->prm.exact_viscosity_operator && c->prm.viscosity_massless_polish >= 0;   // rows that repeat another row's equation: out of this solve's system (k_visc_singular_find)
        if (elim) {
            if (!c->elimList) HIPCHK(c, hipMalloc((void **)&c->elimList, (size_t)(FV_ELIM_CAP + 1) * sizeof(unsigned long long)));
            if (!c->floatList) {
                HIPCHK(c, hipMalloc((void **)&c->floatList, (size_t)(FV_FLOAT_CAP + 1) * sizeof(unsigned long long)));
                HIPCHK(c, hipMalloc((void **)&c->groundMark, c->L.n + 2 * c->L.guard + 64));
                HIPCHK(c, hipMemsetAsync(c->groundMark, 0, c->L.n + 2 * c->L.guard + 64, c->stream));
                c->groundMark += c->L.guard;   // (indexed like every plain array; cleared by the kernel that marks)
            }
            HIPCHK(c, hipMemsetAsync(c->elimList, 0, sizeof(unsigned long long), c->stream));
            HIPCHK(c, hipMemsetAsync(c->floatList, 0, sizeof(unsigned long long), c->stream));
            const Lay RE = R0;   // (rows only exist in the liquid's range)
            hipLaunchKernelGGL(k_visc_singular_find, GRID3(RE), 0, c->stream, RE, (const uint8_t *)c->stU, (const uint8_t *)c->stV, (const uint8_t *)c->stW, (const float *)c->volU,
                               (const float *)c->volV, (const float *)c->volW, (const float *)c->visc, (const float *)c->volC, (const float *)c->volEU, (const float *)c->volEV,
                               (const float *)c->volEW, (const uint8_t *)c->validCells, factor, fieldSolve ? 1 : 0, c->elimList, c->floatList);
            hipLaunchKernelGGL(k_visc_floating, dim3(1), dim3(1024), 0, c->stream, c->L, (const uint8_t *)c->stU, (const uint8_t *)c->stV, (const uint8_t *)c->stW, (const float *)c->volU,
                               (const float *)c->volV, (const float *)c->volW, (const float *)c->visc, (const float *)c->volC, (const float *)c->volEU, (const float *)c->volEV,
                               (const float *)c->volEW, factor, fieldSolve ? 1 : 0, (const unsigned long long *)c->floatList, c->groundMark, c->elimList);
            hipLaunchKernelGGL(k_visc_singular_apply<true>, dim3(1), dim3(256), 0, c->stream, (const unsigned long long *)c->elimList, c->stU, c->stV, c->stW, c->U, c->V, c->W);
        }
        { const FillJob z[2] = {{bmax, 2 * sizeof(double), 0}, {c->d_flags + 2, sizeof(int), 0}}; const int rcz = fv_fill_list(c, z, 2); if (rcz) return rcz; }
        // rows of the owned planes only (the SpMV reads diag / own volume at its own index, factors and s at +-1 plane); a change
        // of layout or precision rewrites every entry, not only those near the liquid
        const Lay RS = full ? fv_range(c, 0) : R0;
        hipLaunchKernelGGL(k_visc_setup<T>, GRID3(RS), 0, c->stream, RS, c->U, c->V, c->W, c->stU, c->stV, c->stW, c->volU, c->volV,
                           c->volW, c->volC, c->volEU, c->volEV, c->volEW, c->fC, c->fEU, c->fEV, c->fEW, c->vDiagU, c->vDiagV,
                           c->vDiagW, c->vmU, c->vmV, c->vmW, c->vrU, c->vrV, c->vrW, c->vRowMask, c->validCells, full, vs, bmax, c->d_flags + 2, (refDiag ? 1 : 0) | (predict ? 2 : 0),
                           brick ? 1 : 0, c->LB, c->vMaskB, c->vB[0], c->vB[1], c->vB[2], c->phi);   // (the right-hand side's copy in the layout of s: the fp64 residual of either layout reads it)
        if (elim) hipLaunchKernelGGL(k_visc_singular_apply<false>, dim3(1), dim3(256), 0, c->stream, (const unsigned long long *)c->elimList, c->stU, c->stV, c->stW, c->U, c->V, c->W);
        if (!c->prm.exact_viscosity_operator && c->prm.viscosity_massless_polish >= 0) {   // the massless clusters: listed now and taken out of the velocity criterion's sight, solved after the solve
            if (!c->polishList) {   // the list of edges, and the solve pass's rows and values (four per edge)
                HIPCHK(c, hipMalloc((void **)&c->polishList, (size_t)(FV_POLISH_CAP + 1) * sizeof(unsigned long long)));
                HIPCHK(c, hipMalloc((void **)&c->polishRow, (size_t)FV_POLISH_CAP * 4 * sizeof(unsigned long long)));
                HIPCHK(c, hipMalloc((void **)&c->polishVal, (size_t)FV_POLISH_CAP * 4 * sizeof(float)));
            }
            HIPCHK(c, hipMemsetAsync(c->polishList, 0, sizeof(unsigned long long), c->stream));
            hipLaunchKernelGGL(k_visc_massless_find, GRID3(RS), 0, c->stream, RS, (const float *)c->volEU, (const float *)c->volEV, (const float *)c->volEW, (const float *)c->volU,
                               (const float *)c->volV, (const float *)c->volW, (const uint8_t *)c->vRowMask, c->polishList);
            hipLaunchKernelGGL(k_visc_massless_polish<true>, dim3(64), dim3(64), 0, c->stream, c->L, (const float *)c->visc, (const float *)c->volC, (const float *)c->volEU, (const float *)c->volEV,
                               (const float *)c->volEW, (const float *)c->volU, (const float *)c->volV, (const float *)c->volW, c->vRowMask, c->U, c->V, c->W,
                               factor, c->comm ? 1 : 0, (const unsigned long long *)c->polishList, (int *)nullptr, brick ? c->vMaskB : (uint8_t *)nullptr, c->LB,
                               (unsigned long long *)nullptr, (float *)nullptr);
        }
        const bool pairs = mgPlanned && (c->prm.viscosity_pair_correction > 0 || (c->prm.viscosity_pair_correction == 0 && fieldSolve));   // strongly coupled pairs of rows: listed for the multigrid loops' additive correction (k_visc_pairs_find)
        c->h_flags[10] = 0;
        if (pairs) {
            if (!c->pairList) HIPCHK(c, hipMalloc((void **)&c->pairList, 16 + (size_t)FV_PAIR_CAP * sizeof(VPair)));
            HIPCHK(c, hipMemsetAsync(c->pairList, 0, 16, c->stream));
            hipLaunchKernelGGL(k_visc_pairs_find, GRID3(R0), 0, c->stream, R0, (const float *)c->visc, (const float *)c->volC, (const float *)c->volEU, (const float *)c->volEV, (const float *)c->volEW,
                               (const float *)c->volU, (const float *)c->volV, (const float *)c->volW, (const uint8_t *)c->vRowMask, factor, brick ? 1 : 0, c->vSwz ? 1 : 0, c->LB,
                               (unsigned *)c->pairList, (VPair *)((char *)c->pairList + 16), c->prm.viscosity_mg_omega_first > 0.0f ? c->prm.viscosity_mg_omega_first : 1.317f,
                               c->prm.viscosity_mg_omega_second > 0.0f ? c->prm.viscosity_mg_omega_second : 0.382f,   // (the cycle's Chebyshev pair: k_viscosity_mg.hip)
                               c->prm.viscosity_pair_lambda_floor > 0.0f ? c->prm.viscosity_pair_lambda_floor : 1.0e-5f);
        }
        {
            ReadJob jobs[5] = {FV_JOB(c->h_scal, bmax, 2 * sizeof(double)), FV_JOB(c->h_flags + 2, c->d_flags + 2, sizeof(int))};   // max|rhs|, max|u| over the rows; the row count
            int nj = 2;
            c->h_flags[11] = 0;
            c->nPolishEdges = 0;
            if (!c->prm.exact_viscosity_operator && c->prm.viscosity_massless_polish >= 0) jobs[nj++] = FV_JOB(&c->nPolishEdges, c->polishList, sizeof(int));   // (the low word of the list's counter)
            if (pairs) jobs[nj++] = FV_JOB(c->h_flags + 10, c->pairList, sizeof(int));
            if (elim) jobs[nj++] = FV_JOB(c->h_flags + 11, c->elimList, sizeof(int));   // (the low word of the list's counter)
            const int rcr = fv_read_small(c, jobs, nj);
            if (rcr) return rcr;
        }
        FV_SYNC(c);  // h_flags[2] = row count
        c->nPairs = pairs ? (c->h_flags[10] < FV_PAIR_CAP ? c->h_flags[10] : FV_PAIR_CAP) : 0;
        c->nElim = elim ? (c->h_flags[11] < FV_ELIM_CAP ? c->h_flags[11] : FV_ELIM_CAP) : 0;
        if (c->prm.verbose && pairs) {
            fprintf(stderr, "viscosity solve %ld: %d strongly coupled pairs of rows in the multigrid's additive correction%s\n", c->viscSolves, c->h_flags[10],
                    c->h_flags[10] > FV_PAIR_CAP ? " -- MORE THAN THE LIST HOLDS" : "");
            if (c->prm.verbose > 1 && c->nPairs > 0) {   // how many rows sit in more than one pair, and the weakest / strongest gains
                std::vector<VPair> hp((size_t)c->nPairs);
                HIPCHK(c, hipMemcpy(hp.data(), (const char *)c->pairList + 16, hp.size() * sizeof(VPair), hipMemcpyDeviceToHost));
                std::vector<unsigned long long> keys;
                float gmin = 1e30f, gmax = 0.0f;
                for (const VPair &P : hp) {
                    keys.push_back(((unsigned long long)(P.comps & 3u) << 40) | P.iz0); keys.push_back(((unsigned long long)((P.comps >> 2) & 3u) << 40) | P.iz1);
                    gmin = fminf(gmin, P.i00); gmax = fmaxf(gmax, P.i00);
                }
                std::sort(keys.begin(), keys.end());
                size_t shared = 0;
                for (size_t t = 1; t < keys.size(); t++) shared += keys[t] == keys[t - 1];
                fprintf(stderr, "   %zu rows in pairs, %zu of them in more than one pair; i00 between %.3g and %.3g\n", keys.size() - shared, shared, gmin, gmax);
            }
        }
        if (c->prm.verbose && c->nPolishEdges > FV_POLISH_CAP)
            fprintf(stderr, "viscosity solve %ld: %d edges with massless clusters around them -- MORE THAN THE LIST HOLDS (%d): the rest are neither solved apart nor taken out of the velocity criterion's sight\n",
                    c->viscSolves, c->nPolishEdges, FV_POLISH_CAP);
        if (c->prm.verbose && c->h_flags[11] > 0)
            fprintf(stderr, "viscosity solve %ld: %d rows repeat another row's equation (a singular cluster): held at 0 like the reference's iterate%s\n", c->viscSolves, c->h_flags[11],
                    c->h_flags[11] > FV_ELIM_CAP ? " -- MORE THAN THE LIST HOLDS, the rest stay rows" : "");
    return FLIPV_OK;
}
// LAYOUT AND LISTS: the set-up in the previous solve's layout, the facts every rank must decide alike from (ONE all-gather), the layout this solve takes -- bricks on sparse
// liquids, the plane layouts otherwise (set-up repeated where the choice changes) --, the brick list or the tile / run lists.
template <typename T>
static int visc_layout_and_lists(flipv_context *c, ViscPlan &P) {
    const Lay &L = c->L;
    const Lay &R0 = P.R0;
    const bool brickOk = P.brickOk, swzOk = P.swzOk;
    const int forcedLayout = P.forcedLayout;
    int rc;
    auto run_setup = [&](int layout, bool first) -> int { return visc_run_setup<T>(c, P, layout, first); };
    const int rowlNow = c->prm.tile_rows == 16 || c->prm.tile_rows == 64 ? c->prm.tile_rows : c->tgV.rowl;
    int layoutTry = c->vLayout;
    if (!c->viscStateValid) layoutTry = brickOk ? VLAYOUT_BRICK : VLAYOUT_PLAIN;   // first solve: the reference's scenes are sparse
    if (forcedLayout == 3 && brickOk) layoutTry = VLAYOUT_BRICK;
    if (layoutTry == VLAYOUT_BRICK && !brickOk) layoutTry = VLAYOUT_PLAIN;
    if (layoutTry != VLAYOUT_BRICK) layoutTry = (swzOk && rowlNow == 16) ? VLAYOUT_SWZ : VLAYOUT_PLAIN;
    if ((rc = run_setup(layoutTry, true))) return rc;
    // Lane width of the tile kernels: 4 consecutive i per lane (16-byte accesses); 2 stays selectable for measurements.
    // Sparse liquids (row fill <= 0.35): bricks where possible, the load-predicated tile SpMV otherwise.
    const double ownVol = (double)(L.ohi[0] - L.olo[0]) * (double)(L.ohi[1] - L.olo[1]) * (double)(L.ohi[2] - L.olo[2]);
    const double fillLocal = (double)c->h_flags[2] / (3.0 * ownVol);
    // Several ranks: ONE exchange carries what every rank must decide alike from -- max|rhs| (the tolerance), the row count and the owned volume
    // (the layout: a rank that took planes where another took bricks would run a different sequence of collectives; and "no rows anywhere").
    double fill = fillLocal, rowsAll = (double)c->h_flags[2];
    if (c->comm) {
        const double mine[4] = {c->h_scal[0], (double)c->h_flags[2], ownVol, c->h_scal[1]};
        double all[4 * NSLOT];
        if ((rc = fv_allgather_f64(c, mine, 4, all))) return rc;
        double bn = 0.0, rows = 0.0, vol = 0.0, um = 0.0;
        for (int r = 0; r < c->comm->nranks; r++) { bn = fmax(bn, all[4 * r]); rows += all[4 * r + 1]; vol += all[4 * r + 2]; um = fmax(um, all[4 * r + 3]); }
        c->h_scal[0] = bn;
        c->h_scal[1] = um;
        rowsAll = rows;
        fill = rows / (3.0 * vol);
    }
    const double bnormAll = c->h_scal[0];
    const double umaxAll = c->h_scal[1];   // max|u| over the rows, all ranks: the scale of the velocity criterion (PcgScal::vel_tol)
    c->vRowsAll = rowsAll;
    c->vwV = 4;
    if (c->prm.viscosity_lane_width == 2 || c->prm.viscosity_lane_width == 4) c->vwV = c->prm.viscosity_lane_width;  // measurement switch: forced lane width
    c->vPred = fillLocal <= 0.35;   // (the plane kernels' load predication is the rank's own business)
    const bool wantBrick = brickOk && (forcedLayout == 3 || fill <= (c->vLayout == VLAYOUT_BRICK ? 0.40 : 0.30));   // (hysteresis around 0.35)
    if (wantBrick != (c->vLayout == VLAYOUT_BRICK)) {
        if ((rc = run_setup(wantBrick ? VLAYOUT_BRICK : ((swzOk && rowlNow == 16) ? VLAYOUT_SWZ : VLAYOUT_PLAIN), false))) return rc;
    }
    const bool brick = c->vLayout == VLAYOUT_BRICK;
    if (brick) {
        if ((rc = fv_build_bricks(c, R0))) return rc;
        c->nActiveV = c->nBricks; c->nRunsV = 0;   // (c->nIntV: fv_build_bricks -- under a communicator the bricks interior to the rank's box, listed first; the SpMV and the fine sweeps over them run beside the halo exchange)
    } else {
        rc = fv_build_tiles(c, &c->tgV, c->vwV, 3, c->vDiagU, c->vDiagV, c->vDiagW, c->vRowMask, c->tileListV, &c->nActiveV, &c->nIntV, c->h_flags + 2, 3, &c->mlistV, &c->mlistCapV, 0.90, &c->geoMemoV);
        if (rc) return rc;
        if ((rc = fv_build_runs(c, c->tgV, c->vwV, c->nActiveV, !c->vPred, c->vRowMask, &c->runsV, &c->runCapV, &c->nRunsV, &c->runLenV, &c->rmaskV, &c->rmaskCapV))) return rc;
        if (c->vSwz != (swzOk && c->tgV.rowl == 16 ? 1 : 0)) {  // the geometry changed: the vectors go into the other layout
            if ((rc = run_setup(c->vSwz ? VLAYOUT_PLAIN : VLAYOUT_SWZ, false))) return rc;
        }
    }
    P.bnormAll = bnormAll; P.umaxAll = umaxAll; P.rowsAll = rowsAll; P.fillLocal = fillLocal;
    return FLIPV_OK;
}

// APPLY (_applySolutionToVelocityField, viscositysolver.cpp:692-727): x (+ the fp64 accumulator) into U, V, W -- 0 off the rows --, the massless clusters solved apart
// (k_visc_massless_polish), the velocities' halo.
template <typename T>
static int visc_apply_solution(flipv_context *c, const Lay &R0, bool brick, bool useAcc, int refinements, bool nontrivial, const PcgSys<T, 3> &v) {
    const Lay &L = c->L;
    int rc;
    const size_t off = plane_off(L, R0.kb), cnt = (size_t)(R0.ke - R0.kb) * L.sz;
    float *uvw[3] = {c->U, c->V, c->W};
    for (int m = 0; m < 3; m++) {
        if (brick) fv_brick_writeback<T>(c, R0, m, useAcc, uvw[m]);   // x (+ the fp64 accumulator refinements / replacements flushed it into)
        else if (useAcc && refinements > 0) hipLaunchKernelGGL(k_plane_writeback<T>, GRID3(R0), 0, c->stream, R0, c->vSwz, (const T *)v.x[m], (const double *)c->vXacc[m], uvw[m]);
        else if (c->vSwz) hipLaunchKernelGGL(k_unswizzle_to_f32<T>, GRID3(R0), 0, c->stream, R0, (const T *)v.x[m], uvw[m]);
        else if (c->pgrid[0] > 1 || c->pgrid[1] > 1) hipLaunchKernelGGL(k_box_to_f32<T>, GRID3(R0), 0, c->stream, R0, (const T *)v.x[m], uvw[m]);   // only what the rank owns: its i / j halo holds the neighbours' velocities
        else hipLaunchKernelGGL(k_vec_to_f32<T>, dim3(2048), dim3(256), 0, c->stream, (const T *)v.x[m] + off, uvw[m] + off, cnt);
    }
    if (c->prm.verbose) HIPCHK(c, hipMemsetAsync(c->d_flags + 13, 0, sizeof(int), c->stream));   // (the count the verbose line below prints; the word is fv_build_runs' otherwise)
    if (nontrivial && !c->prm.exact_viscosity_operator && c->prm.viscosity_massless_polish >= 0)   // (the clusters the iteration leaves where fp32 cannot see them: k_visc_massless_polish)
        hipLaunchKernelGGL(k_visc_massless_polish<false>, dim3(64), dim3(64), 0, c->stream, c->L, (const float *)c->visc, (const float *)c->volC, (const float *)c->volEU, (const float *)c->volEV,
                           (const float *)c->volEW, (const float *)c->volU, (const float *)c->volV, (const float *)c->volW, c->vRowMask, c->U, c->V, c->W,
                           c->vFactorNow, c->comm ? 1 : 0, (const unsigned long long *)c->polishList, c->prm.verbose ? c->d_flags + 13 : (int *)nullptr, (uint8_t *)nullptr, c->LB,   // (the list k_visc_massless_find made before the solve)
                           c->polishRow, c->polishVal);
    if (nontrivial && !c->prm.exact_viscosity_operator && c->prm.viscosity_massless_polish >= 0)
        hipLaunchKernelGGL(k_visc_massless_write, dim3(64), dim3(256), 0, c->stream, (const unsigned long long *)c->polishList, (const unsigned long long *)c->polishRow, (const float *)c->polishVal, c->U, c->V, c->W);
    if (c->prm.verbose && nontrivial && !c->prm.exact_viscosity_operator && c->prm.viscosity_massless_polish >= 0) {
        int np = 0;
        HIPCHK(c, hipMemcpy(&np, c->d_flags + 13, sizeof(int), hipMemcpyDeviceToHost));   // (d_flags[13]: the run builder's second word, rewritten by every fv_build_runs)
        fprintf(stderr, "viscosity solve %ld: %d rows of massless clusters solved apart\n", c->viscSolves, np);
    }
    const HaloArray uv[3] = {{c->U, 4}, {c->V, 4}, {c->W, 4}};
    if ((rc = fv_halo_copy(c, uv, 3, 1))) return rc;  // the pressure rhs at plane k1-1 reads W(k1)
    return FLIPV_OK;
}

template <typename T>
static int viscosity_solve_t(flipv_context *c, float dt, flipv_solve_info *info) {
    const Lay &L = c->L;
    flipv_solve_info li;
    memset(&li, 0, sizeof(li));
    c->commBytesSetup = c->commBytesIter = 0.0;
    c->exchIter = c->allrIter = 0;
    // (vPerRowFactors: a VARIABLE viscosity field -- the fp64 residual of the two-stage solve then forms the reference's rows with their own factors,
    // visc_rows.h: d_ref_row_factors; k_bresidual on bricks, k_plane_residual_ref on the plane layouts)
    if (!c->viscosity_nonzero_any) {  // fluidsimulation.cpp:171-184
        li.status = 3;
        if (info) *info = li;
        return FLIPV_OK;
    }
    ViscPlan P;
    int rc = visc_plan<T>(c, dt, P);
    if (rc) return rc;
    const int cap = P.cap;
    PcgScal &sc = P.sc;
    double *const bmax = P.bmax;
    const bool fieldSolve = P.fieldSolve, mgPossible = P.mgPossible, mgPlanned = P.mgPlanned;
    const Lay R0 = P.R0;
    const int refDiag = P.refDiag;
    const double stiffSolve = P.stiffSolve;
    if ((rc = visc_layout_and_lists<T>(c, P))) return rc;
    const double bnormAll = P.bnormAll, umaxAll = P.umaxAll, rowsAll = P.rowsAll;
    const bool brick = c->vLayout == VLAYOUT_BRICK;
    PcgSys<T, 3> v = visc_sys<T>(c);
    const double bnorm = bnormAll;   // (a repeated set-up recomputes the rank's own maximum: the merged one stands)
    li.rhs_norm = bnorm;
    li.rows = c->h_flags[2];
    li.eliminated_rows = c->nElim;
    li.massless_cluster_edges = c->nPolishEdges;
    li.active_tiles = c->nActiveV;
    li.total_tiles = brick ? (int)(c->LB.n / 64) : c->tgV.count();
    li.layout = c->vLayout;
    c->viscosityReady = 1;
    c->viscosityPrec = std::is_same<T, float>::value ? 0 : 1;

    int conv = -1, iters = 0, refinements = 0, corrIters = 0, corrStatus = 0;   // (flipv_solve_info::correction_iterations / correction_status)
    double res = bnorm;
    bool success = false, stalled = false, ranMg = false;
    const bool defectLimited = false;
    double defectRes = 0.0, mainRes = 0.0;   // max|b - A_ref x| after the defect-correction stage; the exact-operator loop's own final residual
    double velStep = 0.0;
    int anyActive = c->nActiveV;
    int replacePeriod = 0;   // (periodic residual replacement inside a loop -- k_viscosity_brick.hip: fv_brick_replace -- was an opt-in study until version 4 of the ABI: it restarts CG with a stale direction again and again, HISTORY.md; the kernels remain as the refinement steps' building blocks)
    // DEFECT CORRECTION (the default operator under the multigrid, fp32 bricks): the Krylov loop only ever sees the exact, positive definite A; the solve for A_ref = A + E is the outer
    // iteration  r = b - A_ref x (fp64, x in the fp64 accumulator) ; solve A dx = r with the fp32 multigrid-PCG ; x += dx.  The same flush / recompute / restart step rescues any fp32
    // solve the stall guard stops.  The rule: include/flipv.h ("THE DEFAULT VISCOSITY SOLVE"); why: DESIGN.md 4.2; the measurements: HISTORY.md, same section, C.
    const bool canRefine = std::is_same<T, float>::value && (brick || c->vwV == 4);   // bricks: k_viscosity_brick.hip; planes: fv_plane_refine above
    const bool staged = canRefine && (refDiag || c->vMixed64);   // (mixed fp64 mode: refinement towards whichever operator the solve is for)
    const bool useAcc = canRefine;
    if (c->comm) anyActive = rowsAll > 0.0 ? 1 : 0;   // (rows somewhere = an active tile / brick somewhere)
    const bool nontrivial = !(bnorm == 0.0 || anyActive == 0);
    if (!nontrivial) {  // pcgsolver.h:254-258: zero rhs -> zero solution, success
        success = true;
        replacePeriod = 0;
    } else {
        // THE SCALE OF THE RESIDUAL TEST.  The reference stops at max|r| <= 1e-6 max|rhs| (pcgsolver.h:259-272).  While the liquid is clear of the walls a row's right-hand side is
        // its own volume x velocity and max|rhs| ~ max|u|; once it touches a wall the rows next to solid faces carry nu dt/dx^2 x (the solid faces' stored velocities) and
        // max|rhs| jumps by that factor (256^3 bunny: 2 900 against max|u| = 1.4), so that the same test lets the bulk of the liquid -- rows whose near-rigid motions have
        // residual = volume x error -- be wrong by 3e-3 of a volume x velocity: the reference's own iterate is then 1e-4 ... 1e-3 max|u| off the solution of its system on
        // tens of thousands of faces (round 4's rule at 256^3, 35 substeps in: 9.4e-4 on 53 000 faces; profiles/r5/eta_scan_256.log).  So the norm every tolerance of
        // this solve is a share of is capped at viscosity_mass_scale x max|u| (100: with viscosity_tolerance = 1e-6 the final target is never above 1e-4 of one full
        // control volume moving at max|u|; 30 and 10 cost 10 % / 30 % more iterations at 256^3 for the same 3e-6 ... 5e-6, profiles/r5/eta_scan_256.log).  flipv_solve_info.rhs_norm stays max|rhs|.
        const double massScale = c->prm.viscosity_mass_scale > 0.0f ? (double)c->prm.viscosity_mass_scale : (c->prm.viscosity_mass_scale < 0.0f ? 0.0 : 100.0);
        // (... and never below a floor that grows with the stiffness -- what an fp32 correction stage reaches scales with |A| ~ S --: max(0.03, min(0.3, 1e-5 S, 1e4 / S)) x max|rhs|;
        // DESIGN.md 4.3, HISTORY.md same section, D)
        const double massFloor = c->prm.viscosity_mass_floor > 0.0f ? (double)c->prm.viscosity_mass_floor : fmax(3.0e-2, fmin(0.3, fmin(1.0e-5 * stiffSolve, 1.0e4 / fmax(stiffSolve, 1.0))));   // (1e-6 x floor x S, the bound on the bulk's relative error, stays <= 1e-2: holdout draw 7, S = 2.1e5, is 1.0e-4 at 0.3 and 3e-5 at 0.03)
        const double bnormEff = (massScale > 0.0 && umaxAll > 0.0 && !c->vMixed64) ? fmin(bnorm, fmax(massScale * umaxAll, massFloor * bnorm)) : bnorm;
        const double tolFinal = c->prm.viscosity_tolerance * bnormEff;
        double resStart = bnorm;
        int nb = pcg_grid(c, c->nActiveV);
        if (c->prm.viscosity_update_grid_cap > 0) { nb = ((c->nActiveV + 7) / 8) * 8; if (nb > c->prm.viscosity_update_grid_cap) nb = c->prm.viscosity_update_grid_cap; if (nb < 8) nb = 8; }  // measurement switch: grid cap of init/update
        const dim3 blk(64, 4, 1);
        const int hl = brick ? 1 : 0;
        const HaloArray sh[3] = {{c->vS[0], sizeof(T), hl}, {c->vS[1], sizeof(T), hl}, {c->vS[2], sizeof(T), hl}};
        const bool useMg = mgPlanned && c->vwV == 4;
        li.preconditioner = useMg ? 1 : 0;
        ranMg = useMg;
        // the operator the PCG loop applies (and the multigrid hierarchy is built from): under the multigrid always the exact one -- with the defect
        // correction towards the reference's around it where that exists (fp32 bricks), without it on the plane layouts (block contexts)
        c->vOperatorExact = (refDiag && !useMg) ? 0 : 1;
        if (useAcc && brick && c->nBricks > 0) hipLaunchKernelGGL(k_brick_zero_f64, dim3(cdiv(c->nBricks, 4) < 2048 ? cdiv(c->nBricks, 4) : 2048), dim3(64, 4, 1), 0, c->stream, (const int *)c->brickList, c->nBricks, c->vXacc[0], c->vXacc[1], c->vXacc[2]);
        if (useAcc && !brick) plane_flush<T>(c, R0, 4);
        int itersDone = 0, corrections = 0;
        bool correctionDue = false, extraStage = false;
        double lastTarget = 0.0;
        double resBeforeStage = 0.0;
        const bool innerDiffers = staged && useMg;   // the Krylov loop runs on the exact operator, the solve is for the reference's
        // The constants of the two-stage rule: include/flipv.h ("THE DEFAULT VISCOSITY SOLVE") states them once; the scans they come from -- stage 1's factor, the
        // correction stage's share by stiffness, one stage against two -- are in HISTORY.md ("Round 3-4: the scans behind the two-stage rule") and profiles/r4/.
        const double stiffNow = (double)c->viscosity_max_any * (double)dt / ((double)c->dx * (double)c->dx);
        const double f1user = c->prm.viscosity_stage1_factor >= 1.0f ? (double)c->prm.viscosity_stage1_factor : 0.0;
        const double gate = c->prm.viscosity_two_stage_max_stiffness > 0.0f ? (double)c->prm.viscosity_two_stage_max_stiffness : 1.0e6;
        const int cap2 = c->prm.viscosity_stage2_max_iterations > 0 ? c->prm.viscosity_stage2_max_iterations : 200;   // (round 3: 48.  A first stage takes 7-30 iterations, a second one at nu dt/dx^2 = 1.3e5 100-150)
        // (A field whose CONTRAST exceeds 1e4 -- nearly or exactly inviscid on part of the nodes, viscous elsewhere -- takes two stages (holdout sweep: 1e-4 | 200 needs them as
        // 0 | 200 does, smooth 1 ... 1 000 does not): the inviscid faces are pure mass rows that pin the viscous body along
        // the interface like a wall with a prescribed velocity, and what one stage leaves there is 3.3e-4 in the velocities at nu = 0 | 200, 64^3 -- 4e-6 with two;
        // 0.5 | 200 is 1.5e-5 with one.  tests/test_gpu_stiff_regime.py)
        const int rounds = c->vMixed64 ? 8 : (c->prm.viscosity_stage2_rounds > 0 ? c->prm.viscosity_stage2_rounds : (c->vZeroRegion ? 2 : 1));   // (vMixed64: refinement to the fp64 tolerance, fv_viscosity_solve)
        const bool early = innerDiffers && stiffNow <= gate && f1user != 1.0;   // stage 1 stops short of the final tolerance
        const double f2 = c->vMixed64 ? 1e-2 : (c->prm.viscosity_stage2_factor > 0.0f ? (double)c->prm.viscosity_stage2_factor : (stiffNow > 2.0e4 ? 1e-3 : (early ? 1e-2 : 2e-2)));
        const double tolMain = !early ? tolFinal : (f1user > 0.0 ? f1user : (stiffNow > 1000.0 ? 3000.0 : 300.0)) * tolFinal;
        if (tolMain > tolFinal && c->prm.viscosity_mg_coarsest_sweeps <= 0) c->vmgSweeps = 8;   // (the rule above is for one loop to the final tolerance)
        const size_t scalBytes = (size_t)FV_NSC * (cap + 2) * NSLOT * sizeof(double);
        // recompute r = b - A_outer (xacc + x) in fp64 (x flushed into xacc), fetch max|r|, hand the loop a fresh set of scalars
        auto recompute_residual = [&](int flushMode) -> int {   // flushMode 1: x is kept beside the accumulator until the caller has looked at the residual (fv_brick_flush_settle)
            { const int rcr = brick ? fv_brick_refine<T>(c, sc, scalBytes, !refDiag, flushMode) : fv_plane_refine<T>(c, R0, sc, scalBytes, !refDiag, flushMode); if (rcr) return rcr; }   // (several ranks: with the accumulator's halo copy and the all-reduce of max|r|)
            refinements++;
            hipLaunchKernelGGL(k_pcg_residual, dim3(1), dim3(64), 0, c->stream, sc, 0, bmax);
            FV_READ(c, c->h_scal, bmax, sizeof(double));
            const int keepIncl = sc.tol_inclusive;
            double *dummy;
            { const int rcc = fv_pcg_reset(c, cap, true, &sc, &dummy, nullptr); if (rcc) return rcc; }
            sc.stall_ratio = c->prm.stall_guard_ratio > 0.0f ? (double)c->prm.stall_guard_ratio : (fieldSolve ? FV_STALL_RATIO_FIELD : FV_STALL_RATIO);   // (the reset restores the views' defaults)
            sc.tol_inclusive = keepIncl;
            FV_SYNC(c);
            res = resStart = c->h_scal[0];
            return FLIPV_OK;
        };
        // (A warm start -- xacc = the incoming velocity, the loop solving for the correction only -- was tried and dropped: the incoming field
        // is rough on the rows (ghost-band and extrapolated faces next to P2G faces), so max|b - A u_old| came out 5 000 x max|b| at 256^3.)
        void (*mgspmv)(flipv_context *, const PcgScal &, int) = brick
            ? +[](flipv_context *cc, const PcgScal &s2, int it) { fv_brick_spmv<float>(cc, s2, it, false); }
            : +[](flipv_context *cc, const PcgScal &s2, int it) { launch_visc_spmv<float, 4>(cc, s2, it, 0, cc->nActiveV); };
        while (!success) {
        const bool correction = correctionDue;   // this round is a bounded defect-correction stage (see above)
        const int capNow = (correction && cap - itersDone > cap2) ? cap2 : cap - itersDone;
        sc.cap = capNow;
        // (the first correction stage starts from stage 1's remainder PLUS the defect: its target is the share of what lies beyond the remainder)
        // (... and no stage is asked for more than three orders of magnitude below stage 1's tolerance: where the defect is small its share collapses to the final
        // tolerance, and an fp32 loop restarted from an fp64 residual does not take its right-hand side down by 3 000 on the hard systems -- honey 256^3 at
        // nu = 200, nu dt/dx^2 = 1.3e5: 7 of 600 substeps ended at 1.4e-6 ... 2.7e-6 max|rhs| after 170-270 iterations, restart included.  With stage 1 at 300 x
        // the tolerance this floor is below the final tolerance, i.e. void; viscosity_stage2_rounds = 2 goes further.  Not in the mixed fp64 mode.)
        const double stageFloor = c->vMixed64 ? tolFinal : fmax(tolFinal, 1e-3 * tolMain);
        sc.tol = correction ? ((extraStage && lastTarget > 0.0) ? lastTarget   // (the restart of a stage that ended short: towards the target that stage had)
                                                                : fmax(stageFloor, f2 * fmax(resStart - ((corrections == 1 && tolMain > tolFinal) ? tolMain : 0.0), 0.0))) : tolMain;
        if (correction) lastTarget = sc.tol;   // (stage 1 at 1e-6, scanned at 256^3: 2e-2 -> 3.7e-5 / 4.0e-5 from the reference's converged velocities, 5e-2 -> 9.1e-5 / 1.15e-4, 1e-1 -> 1.3e-4)
        sc.stall_below = refinements > 0 ? fmin(100.0 * sc.tol, 0.05 * resStart) : 0.0;
        // THE VELOCITY CRITERION of the solve's last loop (flipv_params.viscosity_velocity_tolerance; pcg_common.h: PcgScal::vel_tol): the loop that delivers the
        // result -- a correction stage, its restart, or the one loop of a solve without stages -- is converged when max|r| passes its target AND its last
        // `viscosity_velocity_window` iterations together moved no velocity by more than that share of max|u|.  Stages in between stop on the residual alone.
        {
            const bool lastLoop = !innerDiffers || correction;   // (every correction stage: the solve ends after whichever of them leaves the fp64 residual below the tolerance)
            // (1e-5 beyond S = 5e4: the stiffer the system the less one iteration moves of what is still missing -- 256^3 bunny at nu = 200, S = 1.3e5, 25 substeps in: 3e-5 leaves
            // 1.25e-4 on 6 500 faces against the tightened solve, 1e-5 leaves 8e-6 for 16 % more iterations; profiles/r5/eta_scan_256_nu200.log)
            const double eta = c->prm.viscosity_velocity_tolerance > 0.0f ? (double)c->prm.viscosity_velocity_tolerance : (c->prm.viscosity_velocity_tolerance < 0.0f ? 0.0 : (stiffSolve > 5.0e4 ? 1.0e-5 : 3.0e-5));
            sc.vel_tol = (lastLoop && !c->vMixed64) ? eta * umaxAll : 0.0;
            sc.vel_window = c->prm.viscosity_velocity_window > 0 ? c->prm.viscosity_velocity_window : 4;
            sc.vel_stall = c->prm.viscosity_velocity_stall_ratio > 0.0f ? (double)c->prm.viscosity_velocity_stall_ratio : 0.0;   // (off by default: flipv.h)
            // (how long the criterion may hold a loop whose residual has passed: 48 iterations, PcgScal::vel_patience.  A longer patience for the cheap diagonal loop was scanned on the
            // three low-viscosity misses of round 6's second holdout sweep and on the 256^3 bunny at nu = 1e-3: 72 / 100 / 140 / 200 move the misses up and down without order -- 7.4e-4,
            // 1.8e-4, 1.8e-4, 5.7e-4, 1.7e-5 on one draw -- and cost 6 / 16 / 22 / 30 % of that scene's substep: not a lever.  profiles/r6/patience_scan.log; flipv_debug_params.velocity_patience)
        }
        // (Stop test of every stage: the reference's own, max|r| <= tol (pcgsolver.h:259-272); two further norms were tried and dropped: HISTORY.md, same section, E.)
        conv = -1;
        if (useMg) {
            if ((rc = fv_viscosity_pcg_mg(c, sc, capNow, mgspmv, replacePeriod, itersDone > 0 ? 1 : 0, &conv))) return rc;
        } else if (brick) {
            fv_brick_init<T>(c, sc);
            if ((rc = fv_allreduce_scalars(c, sc.sig(0), NSLOT))) return rc;
            auto spmv = [&](int first, int count, int it) { fv_brick_spmv<T>(c, sc, it, !sc.noB, first, count); };
            auto update = [&](int it) { fv_brick_update<T>(c, sc, it); };
            auto post = [&](int it) { fv_brick_replace<T>(c, sc, it, replacePeriod, 1, nullptr, 0.0f); };
            if ((rc = pcg_run(c, sc, capNow, sh, 3, c->nIntV, c->nActiveV, spmv, update, &conv, FV_GE_VISCOSITY, post, replacePeriod))) return rc;
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
        if ((rc = pcg_run(c, sc, capNow, sh, 3, c->nIntV, c->nActiveV, spmv, update, &conv, FV_GE_VISCOSITY))) return rc;
        }
        const int last = conv >= 0 ? conv : capNow - 1;
        hipLaunchKernelGGL(k_pcg_residual, dim3(1), dim3(64), 0, c->stream, sc, last, bmax, 1);
        FV_READ(c, c->h_scal, bmax, 2 * sizeof(double));
        FV_SYNC(c);
        res = c->h_scal[0];
        if (sc.vel_tol > 0.0) velStep = umaxAll > 0.0 ? c->h_scal[1] / umaxAll : 0.0;   // (the delivering loop's: flipv_solve_info.velocity_step)
        const bool velUnmet = sc.vel_tol > 0.0 && c->h_scal[1] > sc.vel_tol;   // the loop ended (stalled, out of budget) while its last iterations were still moving velocities
        const int itersNow = conv >= 0 ? conv + 1 : capNow;
        success = conv >= 0;
        stalled = false;
        if (success) {   // the stall guard (PcgScal) stops the loop through the same flag: that is not convergence
            int st = 0;
            HIPCHK(c, hipMemcpy(&st, sc.stalled, sizeof(int), hipMemcpyDeviceToHost));
            if (st) { success = false; stalled = true; }
        }
        if (c->prm.verbose >= 2) {   // the residual history: max|r| per iteration
            const size_t stride = fv_scal_stride(cap);
            HIPCHK(c, hipMemcpy(c->h_scal, c->d_scal, FV_SCAL_BANKS * stride * sizeof(double), hipMemcpyDeviceToHost));
            fprintf(stderr, "  residual history (relative to max|rhs| = %.3g):", bnorm);
            for (int it = 0; it < itersNow && it < capNow; it++) {
                double m = 0.0;
                for (int bk = 0; bk < sc.nbank; bk++)
                    for (int q = 0; q < NSLOT; q++) m = fmax(m, c->h_scal[(size_t)bk * stride + (size_t)it * FV_NSC * NSLOT + 4 * NSLOT + q]);
                fprintf(stderr, " %d:%.2e", it, m / bnorm);
            }
            fprintf(stderr, "\n");
            if (sc.vel_tol > 0.0) {   // what each iteration moved (max|alpha p| over the rows the substep uses), relative to max|u|
                fprintf(stderr, "  step history (relative to max|u| = %.3g; criterion: %d iterations together <= %.1e):", umaxAll, sc.vel_window, umaxAll > 0.0 ? sc.vel_tol / umaxAll : 0.0);
                for (int it = 0; it < itersNow && it < capNow; it++) {
                    double m = 0.0;
                    for (int bk = 0; bk < sc.nbank; bk++)
                        for (int q = 0; q < NSLOT; q++) m = fmax(m, c->h_scal[(size_t)bk * stride + (size_t)it * FV_NSC * NSLOT + 5 * NSLOT + q]);
                    fprintf(stderr, " %d:%.1e", it, umaxAll > 0.0 ? m / umaxAll : 0.0);
                }
                fprintf(stderr, "\n");
            }
            // the energy the iterations added to the iterate: |x_k+1|_A^2 - |x_k|_A^2 = alpha_k sigma_k = sigma_k^2 / (p_k, A p_k)   (Hestenes-Stiefel)
            fprintf(stderr, "  energy terms alpha sigma:");
            for (int it = 0; it < itersNow && it < capNow; it++) {
                double sg = 0.0, pq = 0.0;
                for (int bk = 0; bk < sc.nbank; bk++)
                    for (int q = 0; q < NSLOT; q++) {
                        sg += c->h_scal[(size_t)bk * stride + (size_t)it * FV_NSC * NSLOT + q];
                        pq += c->h_scal[(size_t)bk * stride + (size_t)it * FV_NSC * NSLOT + NSLOT + q];
                    }
                fprintf(stderr, " %d:%.3e", it, pq != 0.0 ? sg * sg / pq : 0.0);
            }
            fprintf(stderr, "\n");
        }
        itersDone += itersNow;
        if (!correction) resBeforeStage = res;                   // the main loop's own (recurrence) residual
        if (correction) {   // a correction stage's result is kept whether or not it reached its target -- unless it RAISED the fp64 residual (below) --, but the solve says so
            corrIters += itersNow;
            if (corrStatus != 3) corrStatus = success ? 1 : 2;   // (the LAST stage's outcome; a stage taken back stays on record)
            success = true; stalled = false;
        }
        if (success && !innerDiffers) break;
        if (!canRefine) break;
        if (!success && !(stalled && refinements < 8 && itersDone < cap)) break;   // (cap reached: nothing to continue with)
        // converged on the exact operator (the solve is for the reference's), or stalled: what is the residual really?
        const bool wasConverged = success;
        if (c->prm.verbose) fprintf(stderr, "viscosity solve %ld: %s after %d iterations at %.3g (tolerance %.3g); recomputing the residual in fp64\n", c->viscSolves,
                                    stalled ? "stalled" : (correction ? (corrStatus == 2 ? "correction stage ended short of its target" : "correction stage done") : "converged"), itersDone, res, sc.tol);
        const double defectBefore = resStart;   // (a correction stage: the fp64 residual it started from)
        if ((rc = recompute_residual(correction ? 1 : 0))) return rc;
        bool tookBack = false;
        if (correction) {   // the tentative flush: confirmed, or taken back when the stage made things worse (an fp32 loop that broke down)
            tookBack = res > defectBefore;
            if (brick) fv_brick_flush_settle<T>(c, sc, tookBack); else plane_flush<T>(c, R0, tookBack ? 2 : 3);
            if (tookBack) {
                if (c->prm.verbose) fprintf(stderr, "viscosity solve %ld: the correction stage raised the fp64 residual (%.3g -> %.3g): taken back\n", c->viscSolves, defectBefore, res);
                corrStatus = 3; res = defectBefore;
            }
        }
        stalled = false;
        success = false;
        // (a loop the stall guard stopped while the velocity criterion was unmet -- CG's residual jumps when it finally resolves a light, nearly detached part -- is restarted
        // from the fp64 residual even where that residual already passes: the criterion is what the loop was still running for)
        const bool restartForVelocity = !innerDiffers && velUnmet && !wasConverged && refinements < 8 && itersDone < cap;
        if (res <= tolFinal && !restartForVelocity) { success = true; if (innerDiffers) { defectRes = res; if (corrections >= 1) res = mainRes; } break; }
        if (innerDiffers && wasConverged) {
            // The defect E x the exact-operator loop left behind: ONE bounded correction stage (two by the field's contrast), accepted as it comes; a stage that ENDED SHORT is restarted once from
            // the residual just recomputed; the solve's status and residual are the exact-operator loop's, `defect_residual` reports max|b - A_ref x| at the end.  (HISTORY.md, same section, F.)
            const bool again = !c->vMixed64 && corrStatus == 2 && corrections == rounds && !extraStage && !tookBack && itersDone < cap;
            if (again) extraStage = true;
            if ((corrections >= rounds && !again) || itersDone >= cap || tookBack) {
                success = true; defectRes = res; res = mainRes;
                if (c->vMixed64 && defectRes > tolFinal && corrStatus <= 1) corrStatus = 2;   // (the fp64 residual is this mode's criterion)
                break;
            }
            corrections++;
            correctionDue = true;
            if (corrections == 1) mainRes = resBeforeStage;
        } else correctionDue = false;
        if (itersDone >= cap) break;
        }
        iters = itersDone;
    }
    li.refinements = refinements;
    li.defect_residual = defectRes;
    li.comm_bytes_setup = c->commBytesSetup; li.comm_bytes_per_iteration = c->commBytesIter;
    li.halo_exchanges_per_iteration = c->exchIter; li.allreduces_per_iteration = c->allrIter;
    li.correction_iterations = corrIters;
    li.correction_status = corrStatus;
    li.velocity_step = velStep;
    if (c->prm.verbose && nontrivial)
        fprintf(stderr, "viscosity solve %ld: %s, %s layout, %d iterations, residual %.3g (rhs %.3g), %s\n", c->viscSolves, ranMg ? "multigrid" : "diagonal",
                brick ? "brick" : (c->vSwz ? "swizzled" : "plain"), iters, res, bnorm, success ? "converged" : (stalled ? "stalled" : "cap"));
    // (A multigrid-preconditioned solve that stalls or runs into the cap two orders of magnitude or more below the right-hand side keeps its
    // iterate: it is closer to the solution than a capped diagonal solve gets on such systems -- 1e-2 at best; 256^3 bunny at nu = 500: the
    // multigrid iterate at 1.2e-4 max|rhs| was replaced by a diagonal one at 4.8 max|rhs| while the bound was 1e-4 --, and is reported as "not
    // converged" like any accepted iterate.)
    if (ranMg && !success && !defectLimited && !(res < 1e-2 * bnorm)) {
        // The multigrid-preconditioned solve did not reach the tolerance.  Its iterate is not used: the solve is repeated from scratch
        // with the diagonal, whose capped iterate is what the reference's acceptance rule is about.
        c->viscSolves++;
        c->vLastPrec = 2; c->vLastIts = iters; c->vLastConverged = 0; c->vLastRelRes = bnorm > 0.0 ? res / bnorm : 0.0;
        c->vNoMultigridOnce = 1;
        const int rc2 = viscosity_solve_t<T>(c, dt, info);
        c->vNoMultigridOnce = 0;
        return rc2;
    }
    if (!ranMg && !success && nontrivial && mgPossible && c->prm.viscosity_preconditioner == FLIPV_PRECOND_AUTO && !c->vForceMultigridOnce && !c->vNoMultigridOnce) {
        // AUTO took the diagonal on the strength of the previous solve and it did not converge inside the cap: the answer of a default
        // run is a converged one wherever one is affordable, so this solve is repeated with the multigrid (U, V, W are still the inputs:
        // nothing has been written back).
        c->viscSolves++;
        c->vLastPrec = 1; c->vLastIts = iters; c->vLastConverged = 0; c->vLastRelRes = bnorm > 0.0 ? res / bnorm : 0.0;
        c->vForceMultigridOnce = 1;
        const int rc2 = viscosity_solve_t<T>(c, dt, info);
        c->vForceMultigridOnce = 0;
        return rc2;
    }
    li.iterations = iters;
    li.residual = res;
    c->viscSolves++;
    c->vLastPrec = !nontrivial ? c->vLastPrec : (li.preconditioner ? 2 : 1);   // (a trivial solve says nothing)
    if (nontrivial) { c->vLastIts = iters; c->vLastConverged = (success || defectLimited) ? 1 : 0; c->vLastRelRes = bnorm > 0.0 ? res / bnorm : 0.0; }
    // acceptance rule of viscositysolver.cpp:676-689
    // (a stalled solve is treated like one that ran into the cap: its iterate is used if the residual passes the acceptance bound)
    const bool accepted = success || defectLimited || ((iters == cap || stalled) && res < c->prm.viscosity_accept_tolerance);
    const bool complete = success && corrStatus <= 1;   // (a correction stage that ran out of budget, stalled or was taken back: the result is applied, the status says "not converged")
    li.status = success ? (iters == 0 ? 3 : (complete ? 0 : 1)) : (accepted ? 1 : 2);
    if (accepted && (rc = visc_apply_solution<T>(c, R0, brick, useAcc, refinements, nontrivial, v))) return rc;
    // the accumulator is zero between solves (its halo reads rely on it)
    if (useAcc && nontrivial && !brick && refinements > 0) plane_flush<T>(c, R0, 4);
    if (useAcc && nontrivial && brick && c->nBricks > 0) hipLaunchKernelGGL(k_brick_zero_f64, dim3(cdiv(c->nBricks, 4) < 2048 ? cdiv(c->nBricks, 4) : 2048), dim3(64, 4, 1), 0, c->stream, (const int *)c->brickList, c->nBricks, c->vXacc[0], c->vXacc[1], c->vXacc[2]);
    HIPCHK(c, hipGetLastError());
    if (c->prm.kernel_timing) fv_ev_collect(c);
    if (info) *info = li;
    return complete ? FLIPV_OK : (accepted ? FLIPV_WARN_NOT_CONVERGED : FLIPV_WARN_SOLVE_FAILED);
}

// One fine-level sweep of the multigrid preconditioner (k_viscosity_mg.hip) with the solver's tile SpMV kernel: out = in + omega
// (r - A in)/d (epi 1; epi 3 also adds (r, out) into sig(it + sig_shift)) or out = r - A in (epi 2).  fp32 vectors in the plain
// layout, zero off the rows; `out` must not alias `in`.  it_arg < 0: the device-side iteration counter (hipGraph replay).
void fv_visc_sweep_f32(flipv_context *c, float *const in[3], float *const out[3], int epi, const PcgScal &sc, int it_arg, float omega, int sig_shift) {
    if (c->vLayout == VLAYOUT_BRICK) { fv_brick_sweep_f32(c, in, out, epi, sc, it_arg, omega, sig_shift); return; }
    PcgSys<float, 3> v = visc_sys<float>(c);
    for (int m = 0; m < 3; m++) { v.s[m] = in[m]; v.q[m] = out[m]; }
    int nb = pcg_grid(c, c->nActiveV);
    const int cap = c->prm.viscosity_spmv_grid_cap > 0 ? c->prm.viscosity_spmv_grid_cap : 512;
    if (nb > cap) nb = cap;
    const float *const vo[3] = {c->vOperatorExact ? c->vmU : c->vrU, c->vOperatorExact ? c->vmV : c->vrV, c->vOperatorExact ? c->vmW : c->vrW};
#define VSWEEP(P_, E_) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL((k_visc_spmv<float, 4, P_, true, E_>), dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, \
                           vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, v, sc, it_arg, omega, sig_shift))
    if (c->vPred) { if (epi == 1) VSWEEP(true, EPI_JACOBI); else if (epi == 2) VSWEEP(true, EPI_RESIDUAL); else VSWEEP(true, EPI_JACOBI_DOT); }
    else { if (epi == 1) VSWEEP(false, EPI_JACOBI); else if (epi == 2) VSWEEP(false, EPI_RESIDUAL); else VSWEEP(false, EPI_JACOBI_DOT); }
#undef VSWEEP
}

int fv_viscosity_solve(flipv_context *c, float dt, flipv_solve_info *info) {
    c->vMixed64 = 0;
    { const int rcf = visc_gather_field_facts(c); if (rcf) return rcf; }
    if (c->prm.precision == FLIPV_PRECISION_FP64) {
        // flipv_params.precision = FP64 = the reference's own vector type (pcgsolver.h:241-295 with T = double).  Under the diagonal preconditioner the PCG's
        // vectors ARE fp64 (viscosity_solve_t<double>).  Under the multigrid (whose V-cycle is fp32) the fp64 answer is reached by MIXED-PRECISION ITERATIVE
        // REFINEMENT, which is what the two-stage solve already is: the solution is accumulated and the residual b - A_ref x evaluated in fp64, the Krylov
        // loops between them run in fp32 -- here repeated (up to 8 correction stages, each to 1 % of what is left) until the FP64 residual meets
        // viscosity_tolerance x max|rhs|, the reference's own criterion; status 1 if it does not inside the iteration cap.  (Until round 4 fp64 vectors
        // took the diagonal whatever was asked for: at 256^3 that is an iterate stopped at the cap.)
        const double stiff = (double)c->viscosity_max_any * (double)dt / ((double)c->dx * (double)c->dx);   // (over all ranks: a rank whose own box is inviscid must take the same path)
        const bool wantsMg = c->prm.viscosity_preconditioner == FLIPV_PRECOND_MULTIGRID || (c->prm.viscosity_preconditioner == FLIPV_PRECOND_AUTO && stiff > FV_AUTO_DIAGONAL_STIFFNESS);
        if (!wantsMg || c->prm.viscosity_lane_width == 2) return viscosity_solve_t<double>(c, dt, info);
        c->vMixed64 = 1;
        const int rc = viscosity_solve_t<float>(c, dt, info);
        c->vMixed64 = 0;
        return rc;
    }
    return viscosity_solve_t<float>(c, dt, info);
}

int fv_bench_viscosity_spmv(flipv_context *c, int reps, double *ms, double *cells, int mgLoop) {   // mgLoop: the launch of the multigrid-preconditioned loop whatever the last solve ran (q = A p and p.q alone: EPI_SPMV_A)
    if (!c->viscosityReady || c->nActiveV <= 0) { c->err = "flipv_bench_spmv: run flipv_viscosity_solve first"; return FLIPV_ERR_INVALID; }
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    const int saved = c->prm.kernel_timing;
    c->prm.kernel_timing = 0;
    PcgScal sc;
    memset(&sc, 0, sizeof(sc));
    if (c->vLayout == VLAYOUT_BRICK) {   // the brick kernel, in the variant the last solve's loop launched: with the fused (r, q) dots in the diagonal loop
        // (unless flipv_params.beta_from_conjugacy), q = A p and p.q alone in the multigrid loop
        const bool rdot = c->prm.beta_from_conjugacy == 0 && c->vLastPrec != 2 && !mgLoop;
        sc.onlyA = (mgLoop || c->vLastPrec == 2) ? 1 : 0;
        for (int w = 0; w < 3; w++) { if (c->viscosityPrec) fv_brick_spmv<double>(c, sc, 0, rdot); else fv_brick_spmv<float>(c, sc, 0, rdot); }
        HIPCHK(c, hipEventRecord(a, c->stream));
        for (int r = 0; r < reps; r++) { if (c->viscosityPrec) fv_brick_spmv<double>(c, sc, 0, rdot); else fv_brick_spmv<float>(c, sc, 0, rdot); }
        HIPCHK(c, hipEventRecord(b, c->stream));
        HIPCHK(c, hipEventSynchronize(b));
        float t = 0;
        HIPCHK(c, hipEventElapsedTime(&t, a, b));
        c->prm.kernel_timing = saved;
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        *ms = (double)t / reps;
        *cells = (double)c->nBricks * 64;
        return FLIPV_OK;
    }
    const int fr = mgLoop ? 0 : -1;
    sc.onlyA = mgLoop ? 1 : 0;
    for (int w = 0; w < 3; w++) {
        if (c->vwV == 4) { if (c->viscosityPrec) launch_visc_spmv<double, 4>(c, sc, 0, 0, c->nActiveV, nullptr, fr); else launch_visc_spmv<float, 4>(c, sc, 0, 0, c->nActiveV, nullptr, fr); }
        else { if (c->viscosityPrec) launch_visc_spmv<double, 2>(c, sc, 0, 0, c->nActiveV, nullptr, fr); else launch_visc_spmv<float, 2>(c, sc, 0, 0, c->nActiveV, nullptr, fr); }
    }
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; r++) {
        if (c->vwV == 4) { if (c->viscosityPrec) launch_visc_spmv<double, 4>(c, sc, 0, 0, c->nActiveV, nullptr, fr); else launch_visc_spmv<float, 4>(c, sc, 0, 0, c->nActiveV, nullptr, fr); }
        else { if (c->viscosityPrec) launch_visc_spmv<double, 2>(c, sc, 0, 0, c->nActiveV, nullptr, fr); else launch_visc_spmv<float, 2>(c, sc, 0, 0, c->nActiveV, nullptr, fr); }
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
