// pcg_common.h -- tile-based, device-resident preconditioned conjugate gradients shared by the
// pressure (1 component, cell-centred) and viscosity (3 components, face-centred) solves.
//
// Algorithm = the reference's PCG loops (pressuresolver.cpp:521-567, pcgsolver.h:241-295) with the
// sequential MIC(0) triangular solves replaced by a diagonal preconditioner M = diag(A), which is the
// only part of the reference algorithm that does not parallelise.  Everything stays on the
// device: alpha/beta are formed inside the kernels from fp64 dot products accumulated with one fp64
// atomic per block, indexed by iteration so nothing has to be reset inside the loop, and a device flag
// stops all later launches once max|r| passes the tolerance, so the host only polls every few iterations.
//
//   init   : z = r/diag ; s = z ; sigma[0] = z.r
//   spmv   : z = A s                    ; dA[it]    = s.z          (kernel supplied by the caller)
//   update : alpha = sigma[it]/dA[it]   ; x += alpha s ; r -= alpha z ; rmax[it] = max|r| ;
//            sigma[it+1] = (r/diag).r
//   dir    : if rmax[it] passes tol -> conv = it, stop ; beta = sigma[it+1]/sigma[it] ; s = r/diag + beta s
#pragma once
#include "flipv_internal.h"

struct PcgComps {   // up to 3 components, each its own Array3d-layout grid
    int n;
    int w[3], h[3], d[3];
    const float *diag[3];
};

template <typename T>
struct PcgVecs {
    T *x[3], *r[3], *z[3], *s[3];
};

struct PcgScal {    // views into ctx->d_scal
    double *sigma;  // [cap+1]
    double *dA;     // [cap]
    double *rmax;   // [cap]
    int *conv;      // converged-at iteration, -1 while running
    double tol;
    int tol_inclusive;  // 1: res <= tol (pcgsolver.h:270), 0: res < tol (pressuresolver.cpp:548)
};

__device__ __forceinline__ void d_tile_coords(int tile, const TileGrid &tg, int &i, int &j, int &k0) {
    const int tx = tile % tg.ntx;
    const int t2 = tile / tg.ntx;
    const int ty = t2 % tg.nty;
    const int tz = t2 / tg.nty;
    i = tx * TX + threadIdx.x;
    j = ty * TY + threadIdx.y;
    k0 = tz * TZ;
}

template <typename T>
__global__ __launch_bounds__(256) void k_pcg_init(const int *__restrict__ tiles, int ntiles, TileGrid tg, PcgComps cp,
                                                  PcgVecs<T> v, PcgScal sc) {
    __shared__ double lds[4];
    const int slot = d_tile_slot(blockIdx.x, ntiles);
    double acc = 0.0;
    if (slot < ntiles) {
        int i, j, k0;
        d_tile_coords(tiles[slot], tg, i, j, k0);
        for (int c = 0; c < cp.n; c++) {
            if (i >= cp.w[c] || j >= cp.h[c]) continue;
            for (int kk = 0; kk < TZ; kk++) {
                const int k = k0 + kk;
                if (k >= cp.d[c]) break;
                const size_t f = DIDX(i, j, k, cp.w[c], cp.h[c]);
                const float dg = cp.diag[c][f];
                const T r = v.r[c][f];
                const T z = dg != 0.0f ? r / (T)dg : (T)0;
                v.s[c][f] = z;
                acc += (double)z * (double)r;
            }
        }
    }
    const double tot = block_sum_256(acc, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && tot != 0.0) atomicAdd(&sc.sigma[0], tot);
}

template <typename T>
__global__ __launch_bounds__(256) void k_pcg_update(const int *__restrict__ tiles, int ntiles, TileGrid tg, PcgComps cp,
                                                    PcgVecs<T> v, PcgScal sc, int it) {
    if (*sc.conv >= 0) return;
    __shared__ double lds[4];
    const double dA = sc.dA[it];
    const double alpha_d = dA != 0.0 ? sc.sigma[it] / dA : 0.0;
    const T alpha = (T)alpha_d;
    const int slot = d_tile_slot(blockIdx.x, ntiles);
    double acc = 0.0, mx = 0.0;
    if (slot < ntiles) {
        int i, j, k0;
        d_tile_coords(tiles[slot], tg, i, j, k0);
        for (int c = 0; c < cp.n; c++) {
            if (i >= cp.w[c] || j >= cp.h[c]) continue;
            for (int kk = 0; kk < TZ; kk++) {
                const int k = k0 + kk;
                if (k >= cp.d[c]) break;
                const size_t f = DIDX(i, j, k, cp.w[c], cp.h[c]);
                const float dg = cp.diag[c][f];
                if (dg == 0.0f) continue;  // not an unknown: x, r, s stay 0
                const T s = v.s[c][f], z = v.z[c][f];
                v.x[c][f] += alpha * s;
                const T r = v.r[c][f] - alpha * z;
                v.r[c][f] = r;
                mx = fmax(mx, fabs((double)r));
                acc += ((double)r / (double)dg) * (double)r;
            }
        }
    }
    const double tot = block_sum_256(acc, lds);
    const double bm = block_max_256(mx, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        if (tot != 0.0) atomicAdd(&sc.sigma[it + 1], tot);
        if (bm > 0.0) atomic_max_nonneg(&sc.rmax[it], bm);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_pcg_dir(const int *__restrict__ tiles, int ntiles, TileGrid tg, PcgComps cp,
                                                 PcgVecs<T> v, PcgScal sc, int it) {
    if (*sc.conv >= 0) return;
    const double res = sc.rmax[it];
    const bool done = sc.tol_inclusive ? (res <= sc.tol) : (res < sc.tol);
    if (done) {
        // every block takes the same decision from the same completed value; one of them records it
        if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) *sc.conv = it;
        return;
    }
    const double sg = sc.sigma[it];
    const T beta = (T)(sg != 0.0 ? sc.sigma[it + 1] / sg : 0.0);
    const int slot = d_tile_slot(blockIdx.x, ntiles);
    if (slot >= ntiles) return;
    int i, j, k0;
    d_tile_coords(tiles[slot], tg, i, j, k0);
    for (int c = 0; c < cp.n; c++) {
        if (i >= cp.w[c] || j >= cp.h[c]) continue;
        for (int kk = 0; kk < TZ; kk++) {
            const int k = k0 + kk;
            if (k >= cp.d[c]) break;
            const size_t f = DIDX(i, j, k, cp.w[c], cp.h[c]);
            const float dg = cp.diag[c][f];
            if (dg == 0.0f) continue;
            v.s[c][f] = v.r[c][f] / (T)dg + beta * v.s[c][f];
        }
    }
}

// Race note for k_pcg_dir: block 0 writes *conv while other blocks of the SAME launch may still read it at
// their top.  They read either -1 (and then take the same `done` branch from rmax[it]) or `it` (return) --
// both leave s untouched, so the outcome is identical.

// ---- host-side driver --------------------------------------------------------------------------
struct PcgHost {
    flipv_context *c;
    PcgComps cp;
    PcgScal sc;
    int cap;
};

int fv_scal_reserve(flipv_context *c, int cap);             // makes d_scal hold 3*(cap+2) doubles
int fv_build_tiles(flipv_context *c, const PcgComps &cp, int *list, int *nActive);
