// flipv_internal.h -- shared host/device declarations of libflipv (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flipv.h"

// ---------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------
struct Dims {
    int I, J, K;
    __host__ __device__ size_t nu() const { return (size_t)(I + 1) * J * K; }
    __host__ __device__ size_t nv() const { return (size_t)I * (J + 1) * K; }
    __host__ __device__ size_t nw() const { return (size_t)I * J * (K + 1); }
    __host__ __device__ size_t nc() const { return (size_t)I * J * K; }
    __host__ __device__ size_t nn() const { return (size_t)(I + 1) * (J + 1) * (K + 1); }
};

// Solver tiles: TX x TY x TZ indices of the (I+1,J+1,K+1) index space; one 256-thread block per tile,
// one wave per x-row of 64 consecutive i (coalesced 256-byte lines), TZ planes marched per thread.
constexpr int TX = 64, TY = 4, TZ = 4;

struct TileGrid {
    int ntx, nty, ntz;
    __host__ __device__ int count() const { return ntx * nty * ntz; }
};

// ---------------------------------------------------------------------------------------------
// device memory helpers
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct flipv_context {
    Dims d;
    float dx;
    int device;
    hipStream_t stream;
    flipv_params prm;
    float gravity[3];
    std::string err;
    std::vector<void *> allocs;

    // persistent grids (Array3d layout)
    float *U, *V, *W, *sU, *sV, *sW, *wU, *wV, *wW, *phi, *solid, *visc, *pressure;
    uint8_t *vU, *vV, *vW;
    // particles
    float *particles;  // AoS 6 floats
    size_t np, pcap;
    // P2G accumulators (value, weight) per component
    float *accU, *accV, *accW, *wgtU, *wgtV, *wgtW;
    // extrapolation stamps
    uint8_t *stampU, *stampV, *stampW;
    // scalars
    double *d_scal;   // device scalar scratch (see solver)
    double *h_scal;   // pinned host mirror
    int *d_flags;     // device int scratch
    int *h_flags;     // pinned host mirror
    int viscosity_nonzero;  // cached host-side: any viscosity node > 0

    // solver tiles
    TileGrid tg;
    int *tileListP, *tileListV;  // active tile ids of the pressure / viscosity systems
    int *tileFlag;
    int nActiveP, nActiveV;
    size_t scalCap;
    float *scp;                  // solid phi at cell centres (viscosity face states)

    // pressure system (dense cell arrays; zero outside pressure cells)
    float *pDiag, *pPi, *pPj, *pPk;
    void *pX, *pR, *pZ, *pS;  // vectors (float or double per precision), nc elements
    // viscosity system
    float *volC, *volU, *volV, *volW, *volEU, *volEV, *volEW;  // control volumes (kept for parity reads)
    float *fC, *fEU, *fEV, *fEW;                               // factor arrays
    float *vDiagU, *vDiagV, *vDiagW;
    uint8_t *rowU, *rowV, *rowW, *stU, *stV, *stW;
    void *vX[3], *vR[3], *vZ[3], *vS[3];
    uint8_t *validCells;  // (I+1,J+1,K+1) dilation mask
    uint8_t *validTmp;

    // kernel timing
    flipv_kernel_stats kstats;
    std::vector<hipEvent_t> evPool;
    size_t evUsed;
    struct EvSpan { size_t a, b; int which; double cells; };
    std::vector<EvSpan> evSpans;
    hipEvent_t phaseEv[FLIPV_PHASE_COUNT + 1];

    // last solve (for flipv_bench_spmv)
    float lastDt;
    int pressureReady, viscosityReady;
    int pressurePrec, viscosityPrec;
};

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return FLIPV_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// device helpers shared by all kernels
// ---------------------------------------------------------------------------------------------
#define DIDX(i, j, k, w, h) ((size_t)(i) + (size_t)(w) * ((size_t)(j) + (size_t)(h) * (size_t)(k)))

__device__ __forceinline__ bool d_in_range(int i, int j, int k, int w, int h, int d) {
    return i >= 0 && j >= 0 && k >= 0 && i < w && j < h && k < d;
}

// LevelsetUtils::fractionInside, 2-point (reference levelsetutils.cpp:15-27)
__device__ __forceinline__ float d_frac2(float l, float r) {
    if (l < 0 && r < 0) return 1.0f;
    if (l < 0 && r >= 0) return l / (l - r);
    if (l >= 0 && r < 0) return r / (r - l);
    return 0.0f;
}

// LevelsetUtils::fractionInside, 4-point marching squares (reference levelsetutils.cpp:38-119).
// The reference rotates a cyclic list; here the rotation is resolved to a start index s so that
// l0 = c[s], l1 = c[s+1], ... (cyclic list order bl, br, tr, tl).
__device__ __forceinline__ float d_frac4(float bl, float br, float tl, float tr) {
    float c[4] = {bl, br, tr, tl};
    int n = (bl < 0) + (br < 0) + (tl < 0) + (tr < 0);
    if (n == 4) return 1.0f;
    if (n == 0) return 0.0f;
    int s = 0;
    if (n == 3) {
        while (c[s & 3] < 0) s++;
    } else if (n == 1) {
        while (c[s & 3] >= 0) s++;
    } else {
        while (c[s & 3] >= 0 || !(c[(s + 1) & 3] < 0 || c[(s + 2) & 3] < 0)) s++;
    }
    float l0 = c[s & 3], l1 = c[(s + 1) & 3], l2 = c[(s + 2) & 3], l3 = c[(s + 3) & 3];
    if (n == 3) {
        float s0 = 1.0f - d_frac2(l0, l3), s1 = 1.0f - d_frac2(l0, l1);
        return 1.0f - 0.5f * s0 * s1;
    }
    if (n == 1) {
        return 0.5f * d_frac2(l0, l3) * d_frac2(l0, l1);
    }
    if (l1 < 0) return 0.5f * (d_frac2(l0, l3) + d_frac2(l1, l2));
    float mid = 0.25f * (l0 + l1 + l2 + l3);
    if (mid < 0) {
        float area = 0.5f * (1.0f - d_frac2(l0, l3)) * (1.0f - d_frac2(l2, l3));
        area += 0.5f * (1.0f - d_frac2(l0, l1)) * (1.0f - d_frac2(l2, l1));
        return 1.0f - area;
    }
    float area = 0.5f * d_frac2(l0, l1) * d_frac2(l0, l3);
    area += 0.5f * d_frac2(l2, l1) * d_frac2(l2, l3);
    return area;
}

// MeshLevelSet::getDistanceAtCellCenter (reference meshlevelset.cpp:66-76); same summation order
__device__ __forceinline__ float d_solid_center(const float *__restrict__ s, int i, int j, int k, int I, int J) {
    const int w = I + 1, h = J + 1;
    return 0.125f * (s[DIDX(i, j, k, w, h)] + s[DIDX(i + 1, j, k, w, h)] + s[DIDX(i, j + 1, k, w, h)] +
                     s[DIDX(i + 1, j + 1, k, w, h)] + s[DIDX(i, j, k + 1, w, h)] + s[DIDX(i + 1, j, k + 1, w, h)] +
                     s[DIDX(i, j + 1, k + 1, w, h)] + s[DIDX(i + 1, j + 1, k + 1, w, h)]);
}

// Grid3d::isFaceBorderingValueU/V/W on the predicate phi<0 (reference grid3d.h:496-530)
__device__ __forceinline__ bool d_face_borders_fluid(int dir, int i, int j, int k, int I, int J, int K,
                                                     const float *__restrict__ phi) {
    const int n = dir == 0 ? I : (dir == 1 ? J : K);
    const int c = dir == 0 ? i : (dir == 1 ? j : k);
    const int di = dir == 0, dj = dir == 1, dk = dir == 2;
    if (c == n) return phi[DIDX(i - di, j - dj, k - dk, I, J)] < 0.0f;
    if (c > 0) return phi[DIDX(i, j, k, I, J)] < 0.0f || phi[DIDX(i - di, j - dj, k - dk, I, J)] < 0.0f;
    return phi[DIDX(i, j, k, I, J)] < 0.0f;
}

// wave64 reductions (no LDS): butterfly over the 64 lanes
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// block (256 threads = 4 waves) reduction to thread 0 via one LDS slot per wave
__device__ __forceinline__ double block_sum_256(double v, double *lds4) {
    v = wave_sum(v);
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) lds4[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) r = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max_256(double v, double *lds4) {
    v = wave_max(v);
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) lds4[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) r = fmax(fmax(lds4[0], lds4[1]), fmax(lds4[2], lds4[3]));
    __syncthreads();
    return r;
}

// atomic max of a non-negative double (bit pattern is monotone for x >= 0)
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
    atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// XCD-aware tile fetch: blocks b, b+8, b+16... run on the same XCD (block b -> XCD b%8); give each XCD a
// contiguous chunk of the (spatially ordered) tile list so that j/k-neighbour tiles share an L2.
__device__ __forceinline__ int d_tile_slot(int b, int n) {
    const int per = (n + 7) >> 3;
    const int slot = (b & 7) * per + (b >> 3);
    return slot;  // may be >= n: caller checks
}

// ---------------------------------------------------------------------------------------------
// cross-file entry points (host side)
// ---------------------------------------------------------------------------------------------
int fv_particle_sdf(flipv_context *c);
int fv_p2g(flipv_context *c);
int fv_extrapolate(flipv_context *c);
int fv_body_force(flipv_context *c, float dt);
int fv_compute_weights(flipv_context *c);
int fv_apply_pressure(flipv_context *c, float dt);
int fv_constrain(flipv_context *c);
int fv_cfl(flipv_context *c, float *dt_out);
int fv_update_particle_velocities(flipv_context *c);
int fv_advect_particles(flipv_context *c, float dt);
int fv_pressure_solve(flipv_context *c, float dt, flipv_solve_info *info);
int fv_viscosity_solve(flipv_context *c, float dt, flipv_solve_info *info);
int fv_bench_spmv(flipv_context *c, int which, int reps, double *ms, double *cells);

// event-pool helpers for kernel timing
void fv_ev_begin(flipv_context *c, int which, double cells);
void fv_ev_end(flipv_context *c);
void fv_ev_collect(flipv_context *c);
