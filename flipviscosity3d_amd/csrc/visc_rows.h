// visc_rows.h -- the U, V and W rows of the variational viscosity operator at one lane's NV consecutive indices, in DIFFERENCE form
// (viscositysolver.cpp:394-446 and the V / W analogues; signs: SURVEY.md A.6b).  Shared by the tile kernels (k_viscosity_geo.inc:
// NV = 2 or 4, i-neighbours by lane shift) and the brick kernels (k_viscosity_brick.hip: NV = 1, every neighbour a load).
#pragma once
#include "pcg_common.h"

// own volume + the rounding defect of the reference's float diagonal: fl(vol + fR + ... + fK) minus the exact sum of the same seven floats (k_viscosity.hip: k_visc_setup)
__device__ __forceinline__ float d_ref_volume(float vol, float fR, float fL, float fT, float fB, float fF, float fK, float dgf) {
    const double exact = (double)vol + (double)fR + (double)fL + (double)fT + (double)fB + (double)fF + (double)fK;
    return (float)((double)vol + ((double)dgf - exact));
}

#define LSH(a, al, e) ((e) > 0 ? (a).v[(e) > 0 ? (e)-1 : 0] : (al))
#define RSH(a, ar, e) ((e) < NV - 1 ? (a).v[(e) < NV - 1 ? (e) + 1 : NV - 1] : (ar))
// The U, V and W rows of a lane's NV indices (difference form, see above) and this lane's share of the three dot products.
// Inputs: own volumes M*, the factor vectors at the index and its j/k neighbours, x at the index and its j/k neighbours,
// the residual (RDOT), and the i-neighbour scalars across the lane boundary (suffix l / r).
// EPI (the multigrid preconditioner's fine-level sweeps, k_viscosity_mg.hip; x = the sweep's input vector, R = the residual r):
//   0  y = A x and the three dot products (the PCG SpMV)
//   1  y = x + omega (r - A x)/d            one damped-Jacobi sweep
//   2  y = r - A x                          the residual that is restricted to the first coarse level
//   3  as 1, and ta = this lane's share of (r, y)
//   4  y = A x and ta = (x, y) ALONE: the multigrid-preconditioned loop's SpMV (its update needs p.q only -- no diagonal, no reciprocal, no (q, q/d): 591 -> 474 instructions
//      per step of the marching kernel, which is issue-bound on filled boxes)
constexpr int EPI_SPMV = 0, EPI_JACOBI = 1, EPI_RESIDUAL = 2, EPI_JACOBI_DOT = 3, EPI_SPMV_A = 4;
template <typename T, int NV, bool RDOT, int EPI = EPI_SPMV>
__device__ __forceinline__ void d_visc_rows(
    const Vec<float, NV> &MU, const Vec<float, NV> &MV, const Vec<float, NV> &MW, const Vec<float, NV> &C0, const Vec<float, NV> &Cjm,
    const Vec<float, NV> &Ckm, const Vec<float, NV> &EW0, const Vec<float, NV> &EWjp, const Vec<float, NV> &EV0, const Vec<float, NV> &EVkp,
    const Vec<float, NV> &EU0, const Vec<float, NV> &EUjp, const Vec<float, NV> &EUkp, const Vec<T, NV> &U0, const Vec<T, NV> &Ujm,
    const Vec<T, NV> &Ujp, const Vec<T, NV> &Ukm, const Vec<T, NV> &Ukp, const Vec<T, NV> &V0, const Vec<T, NV> &Vjm, const Vec<T, NV> &Vjp,
    const Vec<T, NV> &Vkm, const Vec<T, NV> &Vkp, const Vec<T, NV> &W0, const Vec<T, NV> &Wjm, const Vec<T, NV> &Wjp, const Vec<T, NV> &Wkm,
    const Vec<T, NV> &Wkp, const Vec<T, NV> &Vjpkm, const Vec<T, NV> &Wjmkp, const Vec<RT<T>, NV> &RU, const Vec<RT<T>, NV> &RV,
    const Vec<RT<T>, NV> &RW, float C0l, float EW0r, float EV0r, T U0l, T U0r, T V0l, T V0r, T W0l, T W0r, T Vjpl, T Wkpl, T Ujmr, T Ukmr,
    Vec<T, NV> &yU, Vec<T, NV> &yV, Vec<T, NV> &yW, T &ta, T &tb, T &tc, T omega = (T)0) {
#pragma clang fp contract(fast)
    // the three dot products: this lane's <= 3 NV rows are summed in the vector precision, then folded into the fp64
    // accumulators once per tile (12 fp32 FMAs instead of ~100 fp64 operations per tile and lane; every sum across
    // lanes, tiles and blocks stays fp64)
#pragma unroll
    for (int e = 0; e < NV; e++) {
        const T uc = U0.v[e], vc = V0.v[e], wc = W0.v[e];
        const T ur = RSH(U0, U0r, e), ul = LSH(U0, U0l, e);
        const T vr = RSH(V0, V0r, e), vl = LSH(V0, V0l, e);
        const T wr = RSH(W0, W0r, e), wl = LSH(W0, W0l, e);
        {   // U row
            const float fR = C0.v[e], fL = LSH(C0, C0l, e), fT = EWjp.v[e], fB = EW0.v[e], fF = EVkp.v[e], fK = EV0.v[e];
            T y = (T)0;
            if (MU.v[e] > -0.5f) {
                const T txx = (T)fR * (ur - uc) - (T)fL * (uc - ul);
                const T txy = (T)fT * ((Ujp.v[e] - uc) + (Vjp.v[e] - LSH(Vjp, Vjpl, e))) - (T)fB * ((uc - Ujm.v[e]) + (vc - vl));
                const T txz = (T)fF * ((Ukp.v[e] - uc) + (Wkp.v[e] - LSH(Wkp, Wkpl, e))) - (T)fK * ((uc - Ukm.v[e]) + (wc - wl));
                y = (T)MU.v[e] * uc - txx - txy - txz;
                const float dg = MU.v[e] + fR + fL + fT + fB + fF + fK;  // same order as k_visc_setup
                if (EPI == EPI_SPMV) {
                    const T yi = y * d_recip<T>(dg);
                    ta += uc * y; if (RDOT) tb += (T)RU.v[e] * yi; tc += y * yi;
                } else if (EPI == EPI_SPMV_A) {
                    ta += uc * y;
                } else if (EPI == EPI_RESIDUAL) {
                    y = (T)RU.v[e] - y;
                } else {
                    y = uc + omega * ((T)RU.v[e] - y) * d_recip<T>(dg);
                    if (EPI == EPI_JACOBI_DOT) ta += (T)RU.v[e] * y;
                }
            }
            yU.v[e] = y;
        }
        {   // V row
            const float fR = RSH(EW0, EW0r, e), fL = EW0.v[e], fT = C0.v[e], fB = Cjm.v[e], fF = EUkp.v[e], fK = EU0.v[e];
            T y = (T)0;
            if (MV.v[e] > -0.5f) {
                const T tyy = (T)fT * (Vjp.v[e] - vc) - (T)fB * (vc - Vjm.v[e]);
                const T txy = (T)fR * ((vr - vc) + (ur - RSH(Ujm, Ujmr, e))) - (T)fL * ((vc - vl) + (uc - Ujm.v[e]));
                const T tyz = (T)fF * ((Vkp.v[e] - vc) + (Wkp.v[e] - Wjmkp.v[e])) - (T)fK * ((vc - Vkm.v[e]) + (wc - Wjm.v[e]));
                y = (T)MV.v[e] * vc - tyy - txy - tyz;
                const float dg = MV.v[e] + fR + fL + fT + fB + fF + fK;
                if (EPI == EPI_SPMV) {
                    const T yi = y * d_recip<T>(dg);
                    ta += vc * y; if (RDOT) tb += (T)RV.v[e] * yi; tc += y * yi;
                } else if (EPI == EPI_SPMV_A) {
                    ta += vc * y;
                } else if (EPI == EPI_RESIDUAL) {
                    y = (T)RV.v[e] - y;
                } else {
                    y = vc + omega * ((T)RV.v[e] - y) * d_recip<T>(dg);
                    if (EPI == EPI_JACOBI_DOT) ta += (T)RV.v[e] * y;
                }
            }
            yV.v[e] = y;
        }
        {   // W row
            const float fR = RSH(EV0, EV0r, e), fL = EV0.v[e], fT = EUjp.v[e], fB = EU0.v[e], fF = C0.v[e], fK = Ckm.v[e];
            T y = (T)0;
            if (MW.v[e] > -0.5f) {
                const T tzz = (T)fF * (Wkp.v[e] - wc) - (T)fK * (wc - Wkm.v[e]);
                const T txz = (T)fR * ((wr - wc) + (ur - RSH(Ukm, Ukmr, e))) - (T)fL * ((wc - wl) + (uc - Ukm.v[e]));
                const T tyz = (T)fT * ((Wjp.v[e] - wc) + (Vjp.v[e] - Vjpkm.v[e])) - (T)fB * ((wc - Wjm.v[e]) + (vc - Vkm.v[e]));
                y = (T)MW.v[e] * wc - tzz - txz - tyz;
                const float dg = MW.v[e] + fR + fL + fT + fB + fF + fK;
                if (EPI == EPI_SPMV) {
                    const T yi = y * d_recip<T>(dg);
                    ta += wc * y; if (RDOT) tb += (T)RW.v[e] * yi; tc += y * yi;
                } else if (EPI == EPI_SPMV_A) {
                    ta += wc * y;
                } else if (EPI == EPI_RESIDUAL) {
                    y = (T)RW.v[e] - y;
                } else {
                    y = wc + omega * ((T)RW.v[e] - y) * d_recip<T>(dg);
                    if (EPI == EPI_JACOBI_DOT) ta += (T)RW.v[e] * y;
                }
            }
            yW.v[e] = y;
        }
    }
}

// The six factors of the U, V and W row at plain index c AS THE REFERENCE FORMS THEM (viscositysolver.cpp:394-427, 491-524, 589-622): every row averages the
// four viscosity values around an edge in its own order -- (a + b + c + d) in float is not symmetric -- so with a VARIABLE viscosity field the two rows that
// share an edge carry factors one ulp apart, and the reference's matrix is not the symmetric one the stored factors (k_visc_factors: one value per edge) give.
// One ulp of a factor is ~6e-8 nu dt/dx^2 of what a row does to a near-rigid motion -- like the float-rounded diagonal, a defect of A_ref against the operator
// the Krylov loop applies, and corrected the same way: the fp64 residual b - A_ref x uses THESE factors (order: right, left, top, bottom, front, back).
struct RefRowFactors { float U[6], V[6], W[6]; };
__device__ __forceinline__ RefRowFactors d_ref_row_factors(const float *__restrict__ nu, const float *__restrict__ vC, const float *__restrict__ vEU,
                                                           const float *__restrict__ vEV, const float *__restrict__ vEW, size_t c, long sy, long sz, float factor) {
    RefRowFactors F;
    const float f2 = 2 * factor, n0 = nu[c];
    const float nim = nu[c - 1], nip = nu[c + 1], njm = nu[c - sy], njp = nu[c + sy], nkm = nu[c - sz], nkp = nu[c + sz];
    float v;
    // U row at (i, j, k)
    F.U[0] = (f2 * n0) * vC[c];
    F.U[1] = (f2 * nim) * vC[c - 1];
    v = 0.25f * (((nu[c - 1 + sy] + nim) + njp) + n0);             F.U[2] = (factor * v) * vEW[c + sy];
    v = 0.25f * (((nim + nu[c - 1 - sy]) + n0) + njm);             F.U[3] = (factor * v) * vEW[c];
    v = 0.25f * (((nu[c - 1 + sz] + nim) + nkp) + n0);             F.U[4] = (factor * v) * vEV[c + sz];
    v = 0.25f * (((nim + nu[c - 1 - sz]) + n0) + nkm);             F.U[5] = (factor * v) * vEV[c];
    // V row
    v = 0.25f * (((njm + nu[c + 1 - sy]) + n0) + nip);             F.V[0] = (factor * v) * vEW[c + 1];
    v = 0.25f * (((njm + nu[c - 1 - sy]) + n0) + nim);             F.V[1] = (factor * v) * vEW[c];
    F.V[2] = (f2 * n0) * vC[c];
    F.V[3] = (f2 * njm) * vC[c - sy];
    v = 0.25f * (((njm + nu[c - sy + sz]) + n0) + nkp);            F.V[4] = (factor * v) * vEU[c + sz];
    v = 0.25f * (((njm + nu[c - sy - sz]) + n0) + nkm);            F.V[5] = (factor * v) * vEU[c];
    // W row
    v = 0.25f * (((n0 + nkm) + nip) + nu[c + 1 - sz]);             F.W[0] = (factor * v) * vEV[c + 1];
    v = 0.25f * (((n0 + nkm) + nim) + nu[c - 1 - sz]);             F.W[1] = (factor * v) * vEV[c];
    v = 0.25f * (((n0 + nkm) + njp) + nu[c + sy - sz]);            F.W[2] = (factor * v) * vEU[c + sy];
    v = 0.25f * (((n0 + nkm) + njm) + nu[c - sy - sz]);            F.W[3] = (factor * v) * vEU[c];
    F.W[4] = (f2 * n0) * vC[c];
    F.W[5] = (f2 * nkm) * vC[c - sz];
    return F;
}
