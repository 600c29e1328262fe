// k_particles.hip -- particle <-> grid kernels (gfx950): liquid SDF scatter-min (K1), P2G scatter-add
// (K3), G2P + FLIP/PIC update + RK2 advection + solid push-out (K15).
#include "flipv_internal.h"
#include "flipv_comm.h"

// Grid3d::positionToGridIndex(vec3, double dx): float coordinate promoted, times 1/dx, floor
// (reference grid3d.h:60-65).  Done in fp64 exactly like the reference so that a particle lands in the
// same cell on both sides.
__device__ __forceinline__ int d_pos_index(float p, double invdx) { return (int)floor((double)p * invdx); }

// atomic min on a float through integer punning: order-free, so K1 is bit-reproducible
__device__ __forceinline__ void atomic_min_f32(float *addr, float v) {
    if (v >= 0.0f)
        atomicMin((int *)addr, __float_as_int(v));
    else
        atomicMax((unsigned *)addr, __float_as_uint(v));
}

// ------------------------------------------------------------------ K1: liquid SDF from particles
// reference particlelevelset.cpp:98-125
__global__ void k_sdf_scatter(Lay L, const float *__restrict__ aos6, size_t n, float *__restrict__ phi, float dx,
                              float radius) {
    const int I = L.I, J = L.J, K = L.K;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double dxd = (double)dx, invdx = 1.0 / dxd, hw = 0.5 * dxd;
    const float px = aos6[6 * p], py = aos6[6 * p + 1], pz = aos6[6 * p + 2];
    const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
    const int i0 = max(0, gi - 1), j0 = max(0, gj - 1), k0 = max(0, gk - 1);
    const int i1 = min(gi + 1, I - 1), j1 = min(gj + 1, J - 1), k1 = min(gk + 1, K - 1);
    for (int k = k0; k <= k1; k++) {
        const float vz = (float)(k * dxd + hw) - pz;
        for (int j = j0; j <= j1; j++) {
            const float vy = (float)(j * dxd + hw) - py;
            for (int i = i0; i <= i1; i++) {
                const float vx = (float)(i * dxd + hw) - px;
                const float dist = sqrtf(vx * vx + vy * vy + vz * vz) - radius;
                float *a = &phi[gidx(L, i, j, k)];
                if (dist < *a) atomic_min_f32(a, dist);  // plain pre-test only skips useless atomics
            }
        }
    }
}

// ------------------------------------------------------------------ K3: particle -> grid
// reference fluidsimulation.cpp:364-420 (all three components in one pass over the particles).
// v1: one thread per particle, global fp32 atomics (hardware global_atomic_add_f32).
__global__ void k_p2g_scatter(Lay L, const float *__restrict__ aos6, size_t n, float *__restrict__ accU,
                              float *__restrict__ wgtU, float *__restrict__ accV, float *__restrict__ wgtV,
                              float *__restrict__ accW, float *__restrict__ wgtW, float dx) {
    const int I = L.I, J = L.J, K = L.K;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double dxd = (double)dx, invdx = 1.0 / dxd;
    const float hdx = (float)(0.5 * dx);
    const float r = dx, rsq = r * r;
    const float coef1 = (4.0f / 9.0f) * (1.0f / (r * r * r * r * r * r));
    const float coef2 = (17.0f / 9.0f) * (1.0f / (r * r * r * r));
    const float coef3 = (22.0f / 9.0f) * (1.0f / (r * r));
    const float P[3] = {aos6[6 * p], aos6[6 * p + 1], aos6[6 * p + 2]};
    const float Vv[3] = {aos6[6 * p + 3], aos6[6 * p + 4], aos6[6 * p + 5]};
    float *acc[3] = {accU, accV, accW};
    float *wgt[3] = {wgtU, wgtV, wgtW};
#pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        const int w = I + (dir == 0), h = J + (dir == 1), d = K + (dir == 2);
        const float px = P[0] - (dir == 0 ? 0.0f : hdx);
        const float py = P[1] - (dir == 1 ? 0.0f : hdx);
        const float pz = P[2] - (dir == 2 ? 0.0f : hdx);
        const float vel = Vv[dir];
        const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
        const int i0 = max(gi - 1, 0), j0 = max(gj - 1, 0), k0 = max(gk - 1, 0);
        const int i1 = min(gi + 1, w - 1), j1 = min(gj + 1, h - 1), k1 = min(gk + 1, d - 1);
        for (int k = k0; k <= k1; k++) {
            const float vz = (float)(k * dxd) - pz;
            for (int j = j0; j <= j1; j++) {
                const float vy = (float)(j * dxd) - py;
                for (int i = i0; i <= i1; i++) {
                    const float vx = (float)(i * dxd) - px;
                    const float q = vx * vx + vy * vy + vz * vz;
                    if (q < rsq) {
                        const float weight = 1.0f - coef1 * q * q * q + coef2 * q * q - coef3 * q;
                        const size_t f = gidx(L, i, j, k);
                        atomicAdd(&acc[dir][f], weight * vel);
                        atomicAdd(&wgt[dir][f], weight);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ particle bins
// Particles are grouped (indices only, the caller's particle order is never changed) by tile of BIN_T^3 cells of the
// owned slab: count per tile -> exclusive scan + list of non-empty tiles (one block) -> fill.  The scatter kernels
// below then give one workgroup a tile, accumulate its particles in LDS (ds_add_f32 / ds_min) and touch global memory
// once per tile node instead of up to 162 times per particle.
constexpr int BIN_T = 8;
struct BinGrid { int nbx, nby, nbz, c0[3], c1[3]; };  // c0/c1: the box of cells the rank owns, which the tiles cover

__device__ __forceinline__ int d_bin_of(const BinGrid &B, const Lay &L, float px, float py, float pz, double invdx) {
    int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
    gi = min(max(gi, B.c0[0]), B.c1[0] - 1); gj = min(max(gj, B.c0[1]), B.c1[1] - 1); gk = min(max(gk, B.c0[2]), B.c1[2] - 1);
    return (gi - B.c0[0]) / BIN_T + B.nbx * ((gj - B.c0[1]) / BIN_T + B.nby * ((gk - B.c0[2]) / BIN_T));
}

// Particles that are neighbours in the caller's order are mostly neighbours in space: the lanes of a wave that hit the
// same tile are merged into ONE atomic (4096 particles per tile would otherwise serialise on one L2 address).
// Returns this lane's rank among the wave's lanes with the same tile, the group size and whether it is the group's leader.
__device__ __forceinline__ void d_wave_group(bool valid, int t, int &rank, int &size, int &leader) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    rank = 0; size = 0; leader = -1;
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int tl = __shfl(t, lead, 64);
        const unsigned long long m = __ballot(valid && t == tl);
        if (valid && t == tl) {
            rank = __popcll(m & ((1ull << lane) - 1ull));
            size = __popcll(m);
            leader = lead;
        }
        todo &= ~m;
    }
}

__global__ void k_bin_count(BinGrid B, Lay L, const float *__restrict__ aos6, size_t n, int *__restrict__ cnt, float dx) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < n;
    const int t = valid ? d_bin_of(B, L, aos6[6 * p], aos6[6 * p + 1], aos6[6 * p + 2], 1.0 / (double)dx) : -1;
    int rank, size, leader;
    d_wave_group(valid, t, rank, size, leader);
    if (valid && rank == 0) atomicAdd(&cnt[t], size);
}

// bounding box of the non-empty tiles: bb = {min x, y, z, max x, y, z} in tile coordinates
__global__ void k_bin_bbox(BinGrid B, const int *__restrict__ cnt, int *__restrict__ bb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B.nbx * B.nby * B.nbz || cnt[t] <= 0) return;
    const int x = t % B.nbx, y = (t / B.nbx) % B.nby, z = t / (B.nbx * B.nby);
    atomicMin(bb + 0, x); atomicMin(bb + 1, y); atomicMin(bb + 2, z);
    atomicMax(bb + 3, x); atomicMax(bb + 4, y); atomicMax(bb + 5, z);
}

// exclusive scan of the tile counts + ordered list of the non-empty tiles, one block of 1024 threads
__global__ __launch_bounds__(1024) void k_bin_scan(const int *__restrict__ cnt, int ntiles, int *__restrict__ off,
                                                   int *__restrict__ cur, int *__restrict__ list, int *__restrict__ nlist) {
    __shared__ int wsum[16], wne[16];
    __shared__ int base, nbase;
    if (threadIdx.x == 0) { base = 0; nbase = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int start = 0; start < ntiles; start += 1024) {
        const int t = start + threadIdx.x;
        const int v = t < ntiles ? cnt[t] : 0;
        int incl = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        const unsigned long long m = __ballot(v > 0);
        if (lane == 63) wsum[wv] = incl;
        if (lane == 0) wne[wv] = __popcll(m);
        __syncthreads();
        int woff = 0, noff = 0, total = 0, ntot = 0;
        for (int q = 0; q < 16; q++) {
            if (q < wv) { woff += wsum[q]; noff += wne[q]; }
            total += wsum[q]; ntot += wne[q];
        }
        if (t < ntiles) {
            const int o = base + woff + incl - v;
            off[t] = o;
            cur[t] = o;
            if (v > 0) list[nbase + noff + __popcll(m & ((1ull << lane) - 1ull))] = t;
        }
        __syncthreads();
        if (threadIdx.x == 0) { base += total; nbase += ntot; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *nlist = nbase;
}

__global__ void k_bin_fill(BinGrid B, Lay L, const float *__restrict__ aos6, size_t n, int *__restrict__ cur,
                           unsigned *__restrict__ idx, float dx) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < n;
    const int t = valid ? d_bin_of(B, L, aos6[6 * p], aos6[6 * p + 1], aos6[6 * p + 2], 1.0 / (double)dx) : -1;
    int rank, size, leader;
    d_wave_group(valid, t, rank, size, leader);
    int base = 0;
    if (valid && rank == 0) base = atomicAdd(&cur[t], size);
    base = __shfl(base, leader < 0 ? 0 : leader, 64);
    if (valid) idx[base + rank] = (unsigned)p;
}

// K1 on bins: LDS min over the tile's particles, then one global atomic-min per touched cell.  min is order-free:
// bit-identical to k_sdf_scatter.
constexpr int SDF_R = BIN_T + 2;  // a particle reaches its cell +-1
__global__ __launch_bounds__(256) void k_sdf_tiles(BinGrid B, Lay L, const float *__restrict__ aos6,
                                                   const unsigned *__restrict__ idx, const int *__restrict__ off,
                                                   const int *__restrict__ cnt, const int *__restrict__ list,
                                                   const int *__restrict__ nlist, float *__restrict__ phi, float dx,
                                                   float radius, float maxd) {
    __shared__ float sh[SDF_R * SDF_R * SDF_R];
    const int I = L.I, J = L.J, K = L.K;
    const double dxd = (double)dx, invdx = 1.0 / dxd, hw = 0.5 * dxd;
    const int nl = *nlist;
    for (int tt = blockIdx.x; tt < nl; tt += gridDim.x) {
        const int tile = list[tt];
        const int tx = tile % B.nbx, ty = (tile / B.nbx) % B.nby, tz = tile / (B.nbx * B.nby);
        const int bi = B.c0[0] + tx * BIN_T - 1, bj = B.c0[1] + ty * BIN_T - 1, bk = B.c0[2] + tz * BIN_T - 1;
        for (int e = threadIdx.x; e < SDF_R * SDF_R * SDF_R; e += 256) sh[e] = maxd;
        __syncthreads();
        const int start = off[tile], n = cnt[tile];
        for (int q = threadIdx.x; q < n; q += 256) {
            const size_t p = idx[start + q];
            const float px = aos6[6 * p], py = aos6[6 * p + 1], pz = aos6[6 * p + 2];
            const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
            const int i0 = max(0, gi - 1), j0 = max(0, gj - 1), k0 = max(0, gk - 1);
            const int i1 = min(gi + 1, I - 1), j1 = min(gj + 1, J - 1), k1 = min(gk + 1, K - 1);
            for (int k = k0; k <= k1; k++) {
                const float vz = (float)(k * dxd + hw) - pz;
                for (int j = j0; j <= j1; j++) {
                    const float vy = (float)(j * dxd + hw) - py;
                    for (int i = i0; i <= i1; i++) {
                        const float vx = (float)(i * dxd + hw) - px;
                        const float dist = sqrtf(vx * vx + vy * vy + vz * vz) - radius;
                        const int li = i - bi, lj = j - bj, lk = k - bk;
                        if ((unsigned)li < (unsigned)SDF_R && (unsigned)lj < (unsigned)SDF_R && (unsigned)lk < (unsigned)SDF_R) {
                            float *a = &sh[li + SDF_R * (lj + SDF_R * lk)];
                            if (dist < *a) {
                                if (dist >= 0.0f) atomicMin((int *)a, __float_as_int(dist));
                                else atomicMax((unsigned *)a, __float_as_uint(dist));
                            }
                        } else {  // particle outside its clamped tile (never for particles inside the domain)
                            float *a = &phi[gidx(L, i, j, k)];
                            if (dist < *a) atomic_min_f32(a, dist);
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < SDF_R * SDF_R * SDF_R; e += 256) {
            const float v = sh[e];
            if (v < maxd) {
                const int li = e % SDF_R, lj = (e / SDF_R) % SDF_R, lk = e / (SDF_R * SDF_R);
                float *a = &phi[gidx(L, bi + li, bj + lj, bk + lk)];
                if (v < *a) atomic_min_f32(a, v);
            }
        }
        __syncthreads();
    }
}

// K3 on bins ("LDS-binned atomic P2G"): value and weight accumulators of the three components for the tile's reach
// (cells -2..+1 around the tile: the half-cell shift of the transverse axes moves the stencil centre down by one) live in
// LDS; the particles of the tile are accumulated with ds_add_f32, then every touched node is added to the global
// accumulators once.
// one component of one particle with global atomics: the un-binned scatter's inner loops (fluidsimulation.cpp:384-417),
// used by the tile kernel for the (never expected) particle whose stencil leaves its tile's LDS box
__device__ __noinline__ void d_p2g_global(Lay L, int dir, float px, float py, float pz, float vel, double dxd, double invdx,
                                          float *__restrict__ acc, float *__restrict__ wgt, float rsq, float coef1, float coef2,
                                          float coef3) {
    const int w = L.I + (dir == 0), h = L.J + (dir == 1), d = L.K + (dir == 2);
    const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
    const int i0 = max(gi - 1, 0), j0 = max(gj - 1, 0), k0 = max(gk - 1, 0);
    const int i1 = min(gi + 1, w - 1), j1 = min(gj + 1, h - 1), k1 = min(gk + 1, d - 1);
    for (int k = k0; k <= k1; k++) {
        const float vz = (float)(k * dxd) - pz;
        for (int j = j0; j <= j1; j++) {
            const float vy = (float)(j * dxd) - py;
            for (int i = i0; i <= i1; i++) {
                const float vx = (float)(i * dxd) - px;
                const float q = vx * vx + vy * vy + vz * vz;
                if (q < rsq) {
                    const float weight = 1.0f - coef1 * q * q * q + coef2 * q * q - coef3 * q;
                    const size_t f = gidx(L, i, j, k);
                    atomicAdd(&acc[f], weight * vel);
                    atomicAdd(&wgt[f], weight);
                }
            }
        }
    }
}

#ifndef FLIPV_P2G_STRIDE
#define FLIPV_P2G_STRIDE 8
#endif
#ifndef FLIPV_P2G_NATIVE_ADD
#define FLIPV_P2G_NATIVE_ADD 0   // 1: ds_add_f32 (A/B)
#endif
constexpr int P2G_STRIDE = FLIPV_P2G_STRIDE;   // 8 particles per cell at seeding time
constexpr int P2G_R = BIN_T + 3;
constexpr int P2G_RN = P2G_R * P2G_R * P2G_R;
// float add on an LDS word through a compare-and-swap loop.  ds_add_f32 retires about one LANE per three cycles on gfx950 whatever the
// addresses are (tools/micro/lds_atomic_rate.hip: 0.33 lane-atomics per CU-cycle against 13 for ds_add_u32); ds_cmpst_rtn_b32 runs at
// the integer rate, so the loop wins as long as few lanes of a wave meet on one word (4.3 per cycle without collisions, 3.0 on random
// addresses, break-even at 8 lanes per word).  Same semantics: a float sum in arbitrary order.  A bound on the rounds with the native add as
// the fall-back for lanes that keep losing was tried and is slower on the bench scene (4 rounds: 390 us, 16 rounds: 422 us, this loop: 292 us);
// a wave whose lanes pile onto one node (hundreds of particles in one cell) pays for it with a loop as long as the pile.
__device__ __forceinline__ void d_lds_add_f32(float *addr, float v) {
#if FLIPV_P2G_NATIVE_ADD
    atomicAdd(addr, v);
#else
    unsigned *w = (unsigned *)addr;
    unsigned old = *w, assumed;
    do {
        assumed = old;
        old = atomicCAS(w, assumed, __float_as_uint(__uint_as_float(assumed) + v));
    } while (old != assumed);
#endif
}
// the LDS adds of one particle and component over the stencil offsets T0..2 per axis (T0 = 0: the whole 3 x 3 x 3 stencil; 1: the
// corners of the particle's own cell)
template <int T0>
__device__ __forceinline__ void d_p2g_lds(float *sv, float *sw, int lbase, float vel, const float (&ox2)[3], const float (&oy2)[3], const float (&oz2)[3],
                                          const bool (&vi)[3], const bool (&vj)[3], const bool (&vk)[3], float rsq, float coef1, float coef2, float coef3) {
#pragma unroll
    for (int tk = T0; tk < 3; tk++) {
        if (!vk[tk]) continue;
#pragma unroll
        for (int tj = T0; tj < 3; tj++) {
            if (!vj[tj] || !(oy2[tj] + oz2[tk] < rsq)) continue;
#pragma unroll
            for (int ti = T0; ti < 3; ti++) {
                const float qq = ox2[ti] + oy2[tj] + oz2[tk];
                if (vi[ti] && qq < rsq) {
                    const float weight = 1.0f - coef1 * qq * qq * qq + coef2 * qq * qq - coef3 * qq;
                    const int l = lbase + ti + P2G_R * (tj + P2G_R * tk);
                    d_lds_add_f32(&sv[l], weight * vel);
                    d_lds_add_f32(&sw[l], weight);
                }
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_p2g_tiles(BinGrid B, Lay L, const float *__restrict__ aos6,
                                                   const unsigned *__restrict__ idx, const int *__restrict__ off,
                                                   const int *__restrict__ cnt, const int *__restrict__ list,
                                                   const int *__restrict__ nlist, float *__restrict__ accU,
                                                   float *__restrict__ wgtU, float *__restrict__ accV, float *__restrict__ wgtV,
                                                   float *__restrict__ accW, float *__restrict__ wgtW, float dx) {
    __shared__ float sh[6 * P2G_RN];  // [dir][value|weight][node]
    const int I = L.I, J = L.J, K = L.K;
    const double dxd = (double)dx, invdx = 1.0 / dxd;
    const float hdx = (float)(0.5 * dx);
    const float r = dx, rsq = r * r;
    const float coef1 = (4.0f / 9.0f) * (1.0f / (r * r * r * r * r * r));
    const float coef2 = (17.0f / 9.0f) * (1.0f / (r * r * r * r));
    const float coef3 = (22.0f / 9.0f) * (1.0f / (r * r));
    float *gacc[6] = {accU, wgtU, accV, wgtV, accW, wgtW};
    const int nl = *nlist;
    for (int tt = blockIdx.x; tt < nl; tt += gridDim.x) {
        const int tile = list[tt];
        const int tx = tile % B.nbx, ty = (tile / B.nbx) % B.nby, tz = tile / (B.nbx * B.nby);
        const int bi = B.c0[0] + tx * BIN_T - 2, bj = B.c0[1] + ty * BIN_T - 2, bk = B.c0[2] + tz * BIN_T - 2;
        for (int e = threadIdx.x; e < 6 * P2G_RN; e += 256) sh[e] = 0.0f;
        __syncthreads();
        const int start = off[tile], n = cnt[tile];
        // Lane -> particle: consecutive entries of a bin are mostly particles of the same cell (the caller's order is cell order at
        // seeding time and stays close to it), i.e. the lanes of a wave would add to the same few LDS words and the adds
        // serialise.  With P2G_STRIDE > 1 consecutive lanes take entries P2G_STRIDE apart (entry = (q mod rows) * stride + q div rows).
        const int rows = (n + P2G_STRIDE - 1) / P2G_STRIDE;
        for (int q0 = threadIdx.x; q0 < rows * P2G_STRIDE; q0 += 256) {
            const int q = (q0 % rows) * P2G_STRIDE + q0 / rows;
            if (q >= n) continue;
            const size_t p = idx[start + q];
            const float P[3] = {aos6[6 * p], aos6[6 * p + 1], aos6[6 * p + 2]};
            const float Vv[3] = {aos6[6 * p + 3], aos6[6 * p + 4], aos6[6 * p + 5]};
#pragma unroll
            for (int dir = 0; dir < 3; dir++) {
                const int w = I + (dir == 0), h = J + (dir == 1), d = K + (dir == 2);
                const float px = P[0] - (dir == 0 ? 0.0f : hdx);
                const float py = P[1] - (dir == 1 ? 0.0f : hdx);
                const float pz = P[2] - (dir == 2 ? 0.0f : hdx);
                const float vel = Vv[dir];
                const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
                const int i0 = max(gi - 1, 0), j0 = max(gj - 1, 0), k0 = max(gk - 1, 0);
                const int i1 = min(gi + 1, w - 1), j1 = min(gj + 1, h - 1), k1 = min(gk + 1, d - 1);
                float *sv = sh + (2 * dir) * P2G_RN, *sw = sv + P2G_RN;
                // per-axis offsets once (the fp64 index->position products of the reference), then 27 fp32 combinations;
                // q = (ox^2 + oy^2) + oz^2 in the reference's order, rows of the stencil that cannot reach the kernel radius
                // are skipped as a whole (squares are non-negative: ox^2 + oy^2 >= oy^2)
                float ox2[3], oy2[3], oz2[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const float ax = (float)((gi - 1 + t) * dxd) - px, ay = (float)((gj - 1 + t) * dxd) - py, az = (float)((gk - 1 + t) * dxd) - pz;
                    ox2[t] = ax * ax; oy2[t] = ay * ay; oz2[t] = az * az;
                }
                const int lbase = (gi - 1 - bi) + P2G_R * ((gj - 1 - bj) + P2G_R * (gk - 1 - bk));
                const bool inbox = (unsigned)(gi - 1 - bi) <= (unsigned)(P2G_R - 3) && (unsigned)(gj - 1 - bj) <= (unsigned)(P2G_R - 3) &&
                                   (unsigned)(gk - 1 - bk) <= (unsigned)(P2G_R - 3);  // the whole 3x3x3 stencil lies in the tile's LDS box
                if (!inbox) {  // particle outside its clamped tile (never for particles inside the domain): global atomics
                    d_p2g_global(L, dir, px, py, pz, vel, dxd, invdx, gacc[2 * dir], gacc[2 * dir + 1], rsq, coef1, coef2, coef3);
                    continue;
                }
                // per-axis validity of the three stencil offsets (the stencil is clamped to the lattice, :395-401)
                bool vi[3], vj[3], vk[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    vi[t] = gi - 1 + t >= i0 && gi - 1 + t <= i1;
                    vj[t] = gj - 1 + t >= j0 && gj - 1 + t <= j1;
                    vk[t] = gk - 1 + t >= k0 && gk - 1 + t <= k1 && oz2[t] < rsq;
                }
                // Only the 8 corners of the cell that holds the (shifted) particle can lie inside the kernel radius dx: the lower stencil
                // plane of an axis is a whole cell width away unless rounding put the particle a hair beyond a cell boundary, and
                // q >= ox^2 (sums of non-negative floats are monotone), so "no lower plane is closer than dx" is an exact test.  The
                // usual case then issues 16 LDS adds per component instead of 54 -- the kernel is bound by LDS atomic issue
                // (SQ_WAIT_INST_LDS 76 % of its wave cycles) -- and the rare case walks all 27 nodes as before: same sums either way.
#ifdef FLIPV_P2G_TEST_NO_FULL_STENCIL   // (build switch: the corner path alone; tests/test_gpu_parity.py::test_p2g_with_particles_on_and_next_to_cell_boundaries passes with it too -- the rounding case is that rare)
                if (false) d_p2g_lds<0>(sv, sw, lbase, vel, ox2, oy2, oz2, vi, vj, vk, rsq, coef1, coef2, coef3);
                else
#endif
                if (ox2[0] < rsq || oy2[0] < rsq || oz2[0] < rsq) d_p2g_lds<0>(sv, sw, lbase, vel, ox2, oy2, oz2, vi, vj, vk, rsq, coef1, coef2, coef3);
                else d_p2g_lds<1>(sv, sw, lbase, vel, ox2, oy2, oz2, vi, vj, vk, rsq, coef1, coef2, coef3);
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < P2G_RN; e += 256) {
            const int li = e % P2G_R, lj = (e / P2G_R) % P2G_R, lk = e / (P2G_R * P2G_R);
            size_t f = 0;
            bool have = false;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float wv = sh[(2 * a + 1) * P2G_RN + e], vv = sh[(2 * a) * P2G_RN + e];
                if (wv != 0.0f || vv != 0.0f) {  // an untouched node holds exactly 0
                    if (!have) { f = gidx(L, bi + li, bj + lj, bk + lk); have = true; }
                    atomicAdd(&gacc[2 * a][f], vv);
                    atomicAdd(&gacc[2 * a + 1][f], wv);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ K15: grid -> particle
// MACVelocityField::_interpolateLinearU/V/W (reference macvelocityfield.cpp:455-546): fp64 position,
// cell origin and weights; out-of-range corners contribute 0; corner order of interpolation.cpp:54-66.
__device__ __forceinline__ double d_mac_lerp(int dir, double x, double y, double z, double dx, const Lay &L,
                                             const float *__restrict__ g) {
    const int w = L.I + (dir == 0), h = L.J + (dir == 1), d = L.K + (dir == 2);
    if (dir != 0) x -= 0.5 * dx;
    if (dir != 1) y -= 0.5 * dx;
    if (dir != 2) z -= 0.5 * dx;
    const double invdx = 1.0 / dx;
    const int i = (int)floor(x * invdx), j = (int)floor(y * invdx), k = (int)floor(z * invdx);
    const double ix = (x - (double)i * dx) * invdx, iy = (y - (double)j * dx) * invdx, iz = (z - (double)k * dx) * invdx;
    float f0, f1, f2, f3, f4, f5, f6, f7;
    d_corner_pair(g, L, i, j, k, w, h, d, f0, f1);
    d_corner_pair(g, L, i, j + 1, k, w, h, d, f2, f6);
    d_corner_pair(g, L, i, j, k + 1, w, h, d, f3, f4);
    d_corner_pair(g, L, i, j + 1, k + 1, w, h, d, f5, f7);
    const double p0 = f0, p1 = f1, p2 = f2, p3 = f3, p4 = f4, p5 = f5, p6 = f6, p7 = f7;
    return p0 * (1 - ix) * (1 - iy) * (1 - iz) + p1 * ix * (1 - iy) * (1 - iz) + p2 * (1 - ix) * iy * (1 - iz) +
           p3 * (1 - ix) * (1 - iy) * iz + p4 * ix * (1 - iy) * iz + p5 * (1 - ix) * iy * iz + p6 * ix * iy * (1 - iz) +
           p7 * ix * iy * iz;
}

// evaluateVelocityAtPositionLinear (reference macvelocityfield.cpp:564-578)
__device__ __forceinline__ void d_mac_velocity(float px, float py, float pz, double dx, const Lay &L,
                                               const float *__restrict__ U, const float *__restrict__ V,
                                               const float *__restrict__ W, float out[3]) {
    const double x = px, y = py, z = pz;
    if (!(x >= 0 && y >= 0 && z >= 0 && x < dx * L.I && y < dx * L.J && z < dx * L.K)) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    out[0] = (float)d_mac_lerp(0, x, y, z, dx, L, U);
    out[1] = (float)d_mac_lerp(1, x, y, z, dx, L, V);
    out[2] = (float)d_mac_lerp(2, x, y, z, dx, L, W);
}

// _updateFluidParticleVelocities (reference fluidsimulation.cpp:341-352)
__device__ __forceinline__ void d_update_velocity(float *q, double dx, const Lay &L, const float *U,
                                                  const float *V, const float *W, const float *sU, const float *sV,
                                                  const float *sW, float ratio) {
    float vn[3], vo[3];
    d_mac_velocity(q[0], q[1], q[2], dx, L, U, V, W, vn);
    d_mac_velocity(q[0], q[1], q[2], dx, L, sU, sV, sW, vo);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float pic = vn[c];
        const float flip = q[3 + c] + vn[c] - vo[c];
        q[3 + c] = ratio * pic + (1.0f - ratio) * flip;
    }
}

__global__ void k_update_velocities(Lay L, float *__restrict__ aos6, size_t n, const float *__restrict__ U,
                                    const float *__restrict__ V, const float *__restrict__ W,
                                    const float *__restrict__ sU, const float *__restrict__ sV,
                                    const float *__restrict__ sW, float dx, float ratio) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float q[6];
#pragma unroll
    for (int c = 0; c < 6; c++) q[c] = aos6[6 * p + c];
    d_update_velocity(q, (double)dx, L, U, V, W, sU, sV, sW, ratio);
#pragma unroll
    for (int c = 3; c < 6; c++) aos6[6 * p + c] = q[c];
}

// scalar-field trilinear value + gradient at a point of the solid node grid
// (reference interpolation.cpp:68-108 and :122-184): position differences in fp32, weights in fp64.
__device__ __forceinline__ float d_solid_value_grad(float px, float py, float pz, double dx, const Lay &L,
                                                    const float *__restrict__ g, float grad[3]) {
    const int w = L.I + 1, h = L.J + 1, d = L.K + 1;
    const double invdx = 1.0 / dx;
    const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
    const float gx = (float)(gi * dx), gy = (float)(gj * dx), gz = (float)(gk * dx);
    const double ix = (px - gx) * invdx, iy = (py - gy) * invdx, iz = (pz - gz) * invdx;
    float v000, v100, v010, v001, v101, v011, v110, v111;
    d_corner_pair(g, L, gi, gj, gk, w, h, d, v000, v100);
    d_corner_pair(g, L, gi, gj + 1, gk, w, h, d, v010, v110);
    d_corner_pair(g, L, gi, gj, gk + 1, w, h, d, v001, v101);
    d_corner_pair(g, L, gi, gj + 1, gk + 1, w, h, d, v011, v111);
    const double val = (double)v000 * (1 - ix) * (1 - iy) * (1 - iz) + (double)v100 * ix * (1 - iy) * (1 - iz) +
                       (double)v010 * (1 - ix) * iy * (1 - iz) + (double)v001 * (1 - ix) * (1 - iy) * iz +
                       (double)v101 * ix * (1 - iy) * iz + (double)v011 * (1 - ix) * iy * iz +
                       (double)v110 * ix * iy * (1 - iz) + (double)v111 * ix * iy * iz;
    // gradient: differences in fp32, bilinear blend in fp64 (interpolation.cpp:161-178)
    {
        const float a00 = v100 - v000, a10 = v110 - v010, a01 = v101 - v001, a11 = v111 - v011;
        const double l1 = (1 - iy) * a00 + iy * a10, l2 = (1 - iy) * a01 + iy * a11;
        grad[0] = (float)((1 - iz) * l1 + iz * l2);
    }
    {
        const float a00 = v010 - v000, a10 = v110 - v100, a01 = v011 - v001, a11 = v111 - v101;
        const double l1 = (1 - ix) * a00 + ix * a10, l2 = (1 - ix) * a01 + ix * a11;
        grad[1] = (float)((1 - iz) * l1 + iz * l2);
    }
    {
        const float a00 = v001 - v000, a10 = v101 - v100, a01 = v011 - v010, a11 = v111 - v110;
        const double l1 = (1 - ix) * a00 + ix * a10, l2 = (1 - ix) * a01 + ix * a11;
        grad[2] = (float)((1 - iy) * l1 + iy * l2);
    }
    return (float)val;
}

struct ClampBox {  // AABB boundary(0,0,0,I*dx,J*dx,K*dx).expand(-2*dx-1e-4)  (fluidsimulation.cpp:319-320)
    float bx, by, bz;
    double bw, bh, bd;
};

// _advectFluidParticles (reference fluidsimulation.cpp:315-339)
__global__ void k_advect_particles(Lay L, float *__restrict__ aos6, size_t n, const float *__restrict__ U,
                                   const float *__restrict__ V, const float *__restrict__ W,
                                   const float *__restrict__ sU, const float *__restrict__ sV,
                                   const float *__restrict__ sW, const float *__restrict__ solid, float dxf, float dt,
                                   float ratio, ClampBox box) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double dx = (double)dxf;
    float q[6];
#pragma unroll
    for (int c = 0; c < 6; c++) q[c] = aos6[6 * p + c];
    d_update_velocity(q, dx, L, U, V, W, sU, sV, sW, ratio);
    // _traceRK2 (fluidsimulation.cpp:535-541)
    float v[3];
    d_mac_velocity(q[0], q[1], q[2], dx, L, U, V, W, v);
    const float hs = 0.5f * dt;
    d_mac_velocity(q[0] + hs * v[0], q[1] + hs * v[1], q[2] + hs * v[2], dx, L, U, V, W, v);
    float x = q[0] + dt * v[0], y = q[1] + dt * v[1], z = q[2] + dt * v[2];
    // solid push-out (fluidsimulation.cpp:326-333)
    float g[3];
    const float phi_val = d_solid_value_grad(x, y, z, dx, L, solid, g);
    if (phi_val < 0.0f) {
        const float lsq = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
        if (lsq > 0.0f) {
            const float inv = 1.0f / sqrtf(lsq);
            g[0] *= inv; g[1] *= inv; g[2] *= inv;
        }
        x -= phi_val * g[0]; y -= phi_val * g[1]; z -= phi_val * g[2];
    }
    // AABB clamp (aabb.cpp:126-129, 213-234)
    const bool inside = x >= box.bx && y >= box.by && z >= box.bz && (double)x < box.bx + box.bw &&
                        (double)y < box.by + box.bh && (double)z < box.bz + box.bd;
    if (!inside) {
        const float mx = box.bx + (float)box.bw, my = box.by + (float)box.bh, mz = box.bz + (float)box.bd;
        const float eps = (float)1e-6;
        x = fminf(fmaxf(x, box.bx), mx - eps);
        y = fminf(fmaxf(y, box.by), my - eps);
        z = fminf(fmaxf(z, box.bz), mz - eps);
    }
    q[0] = x; q[1] = y; q[2] = z;
#pragma unroll
    for (int c = 0; c < 6; c++) aos6[6 * p + c] = q[c];
}

// =================================================================== host launchers
int fv_sdf_finish(flipv_context *c);
int fv_p2g_finalize(flipv_context *c);

static BinGrid bin_grid(const flipv_context *c) {
    BinGrid B;
    for (int a = 0; a < 3; a++) { B.c0[a] = c->cell0[a]; B.c1[a] = c->cell1[a]; }
    B.nbx = (B.c1[0] - B.c0[0] + BIN_T - 1) / BIN_T;
    B.nby = (B.c1[1] - B.c0[1] + BIN_T - 1) / BIN_T;
    B.nbz = (B.c1[2] - B.c0[2] + BIN_T - 1) / BIN_T;
    return B;
}

// (re)build the particle bins if the particles moved since the last build
int fv_bin_particles(flipv_context *c) {
    if (c->binsValid || !c->np) return FLIPV_OK;
    const BinGrid B = bin_grid(c);
    const int nt = B.nbx * B.nby * B.nbz;
    if (nt > c->binTilesCap) {
        if (c->binCnt) (void)hipFree(c->binCnt);
        c->binCnt = nullptr; c->binTilesCap = 0;
        HIPCHK(c, hipMalloc((void **)&c->binCnt, ((size_t)4 * nt + 16) * sizeof(int)));
        c->binOff = c->binCnt + nt; c->binCur = c->binOff + nt; c->binList = c->binCur + nt; c->binNList = c->binList + nt;
        c->binTilesCap = nt;
    }
    if (c->np > c->binIdxCap) {
        if (c->binIdx) (void)hipFree(c->binIdx);
        c->binIdx = nullptr; c->binIdxCap = 0;
        const size_t cap = c->pcap > c->np ? c->pcap : c->np;
        HIPCHK(c, hipMalloc((void **)&c->binIdx, cap * sizeof(unsigned)));
        c->binIdxCap = cap;
    }
    c->nbx = B.nbx; c->nby = B.nby; c->nbz = B.nbz;
    HIPCHK(c, hipMemsetAsync(c->binCnt, 0, (size_t)nt * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_bin_count, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, B, c->L, c->particles, c->np, c->binCnt, c->dx);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, c->stream, c->binCnt, nt, c->binOff, c->binCur, c->binList, c->binNList);
    hipLaunchKernelGGL(k_bin_fill, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, B, c->L, c->particles, c->np, c->binCur, c->binIdx,
                       c->dx);
    HIPCHK(c, hipGetLastError());
    c->binsValid = 1;
    return FLIPV_OK;
}

static unsigned bin_blocks(const flipv_context *c) {
    const int nt = c->nbx * c->nby * c->nbz;
    return (unsigned)(nt < 4096 ? (nt > 0 ? nt : 1) : 4096);
}

int fv_particle_sdf(flipv_context *c) {
    const float maxd = 3.0f * (float)(double)c->dx;
    // where the liquid is (flipv_context::liqValid): the box of the non-empty particle bins + LIQ_MARGIN, known before anything
    // of this substep is swept
    c->liqPrevValid = c->liqValid;
    for (int a = 0; a < 3; a++) { c->liqPrevLo[a] = c->liqLo[a]; c->liqPrevHi[a] = c->liqHi[a]; }
    c->liqValid = 0;
    // Several ranks: a neighbour's particles scatter into this rank's boundary cells, which its own bins know nothing of -- so the box is the UNION of
    // the ranks' boxes (one all-gather of 7 values per substep; every rank then clips it to what it allocates, fv_range_liquid): whatever lies outside it
    // is trivial on every rank, halo entries included.
    const bool multi = c->comm && c->comm->nranks > 1;
    if (c->inSubstep && (c->np || multi) && !c->prm.unbinned_scatter) {
        double mine[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // [valid, lo x y z, hi x y z]
        if (c->np) {
            int rcb = fv_bin_particles(c);
            if (rcb) return rcb;
            const BinGrid B = bin_grid(c);
            int *bb = c->d_flags + 8;   // 6 ints: min x, y, z, max x, y, z of the non-empty bins (d_flags[8..13]; [12], [13] are rewritten by the run builder later)
            { const FillJob z[2] = {{bb, 3 * sizeof(int), 0x7f}, {bb + 3, 3 * sizeof(int), 0xff}};   // minima start at 0x7f7f7f7f, maxima at -1
              const int rcz = fv_fill_list(c, z, 2); if (rcz) return rcz; }
            const int nt = B.nbx * B.nby * B.nbz;
            hipLaunchKernelGGL(k_bin_bbox, dim3(cdiv(nt, 256)), dim3(256), 0, c->stream, B, c->binCnt, bb);
            int h[6];
            FV_READ(c, h, bb, sizeof(h));
            FV_SYNC(c);
            if (h[3] >= 0) {
                mine[0] = 1.0;
                for (int a = 0; a < 3; a++) {
                    mine[1 + a] = B.c0[a] + h[a] * BIN_T - LIQ_MARGIN;
                    mine[4 + a] = B.c0[a] + (h[3 + a] + 1) * BIN_T + LIQ_MARGIN;
                }
            }
        }
        double all[7 * NSLOT];
        const int nr = multi ? c->comm->nranks : 1;
        if (multi) { const int rcg = fv_allgather_f64(c, mine, 7, all); if (rcg) return rcg; }
        else for (int q = 0; q < 7; q++) all[q] = mine[q];
        for (int r = 0; r < nr; r++) {
            if (all[7 * r] == 0.0) continue;
            for (int a = 0; a < 3; a++) {
                const int lo = (int)all[7 * r + 1 + a], hi = (int)all[7 * r + 4 + a];
                if (!c->liqValid || lo < c->liqLo[a]) c->liqLo[a] = lo;
                if (!c->liqValid || hi > c->liqHi[a]) c->liqHi[a] = hi;
            }
            c->liqValid = 1;
        }
    }
    fv_fill_cells_liquid(c, c->phi, maxd, 1, 0);  // _getMaxDistance (particlelevelset.cpp:94-96)
    if (c->np) {
        // _particleRadius (fluidsimulation.cpp:36)
        const float radius = (float)(c->dx * 1.01 * sqrt(3.0) / 2.0);
        if (c->prm.unbinned_scatter) {  // un-binned scatter (kept for A/B measurements)
            hipLaunchKernelGGL(k_sdf_scatter, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                               c->phi, c->dx, radius);
        } else {
            int rcb = fv_bin_particles(c);
            if (rcb) return rcb;
            hipLaunchKernelGGL(k_sdf_tiles, dim3(bin_blocks(c)), dim3(256), 0, c->stream, bin_grid(c), c->L, c->particles, c->binIdx,
                               c->binOff, c->binCnt, c->binList, c->binNList, c->phi, c->dx, radius, maxd);
        }
    }
    // contributions to the neighbours' boundary planes (a particle reaches one cell beyond its own)
    float *parr[1] = {c->phi};
    int rc = fv_halo_reduce(c, parr, 1, 1, 1, HALO_MIN_F32);
    if (rc) return rc;
    fv_sdf_finish(c);
    HIPCHK(c, hipGetLastError());
    const HaloArray ph[1] = {{c->phi, 4}};
    return fv_halo_copy(c, ph, 1, 4);  // P2G masks, volumes (trilinear +-1.5 cells, 2 dilation layers) read phi across the cut
}

int fv_p2g(flipv_context *c) {
    const Lay R = fv_range_liquid(c, 2, 1);  // planes this rank's particles can reach (whole allocated planes are cleared)
    const size_t off = plane_off(c->L, R.kb), bytes = (size_t)(R.ke - R.kb) * c->L.sz * 4;
    float *acc[6] = {c->accU, c->accV, c->accW, c->wgtU, c->wgtV, c->wgtW};
    { FillJob z[6]; for (int q = 0; q < 6; q++) z[q] = {acc[q] + off, bytes, 0}; const int rcz = fv_fill_list(c, z, 6); if (rcz) return rcz; }
    if (c->np) {
        if (c->prm.unbinned_scatter) {  // un-binned scatter (kept for A/B measurements)
            hipLaunchKernelGGL(k_p2g_scatter, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                               c->accU, c->wgtU, c->accV, c->wgtV, c->accW, c->wgtW, c->dx);
        } else {
            int rcb = fv_bin_particles(c);
            if (rcb) return rcb;
            hipLaunchKernelGGL(k_p2g_tiles, dim3(bin_blocks(c)), dim3(256), 0, c->stream, bin_grid(c), c->L, c->particles, c->binIdx,
                               c->binOff, c->binCnt, c->binList, c->binNList, c->accU, c->wgtU, c->accV, c->wgtV, c->accW, c->wgtW,
                               c->dx);
        }
    }
    int rc = fv_halo_reduce(c, acc, 6, 2, 2, HALO_ADD_F32);
    if (rc) return rc;
    fv_p2g_finalize(c);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_update_particle_velocities(flipv_context *c) {
    if (c->np)
        hipLaunchKernelGGL(k_update_velocities, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                           c->U, c->V, c->W, c->sU, c->sV, c->sW, c->dx, c->prm.pic_ratio);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_advect_particles(flipv_context *c, float dt) {
    const Lay &d = c->L;
    ClampBox b;
    const float dxf = c->dx;
    double bw = (double)(d.I * dxf), bh = (double)(d.J * dxf), bd = (double)(d.K * dxf);
    const double ev = -2 * dxf - 1e-4, eh = 0.5 * ev;  // AABB::expand (aabb.cpp:118-124)
    b.bx = b.by = b.bz = 0.0f - (float)eh;
    b.bw = bw + ev; b.bh = bh + ev; b.bd = bd + ev;
    // a particle samples the fields up to CFL (5) cells + the trilinear support away from its cell
    const HaloArray vel[6] = {{c->U, 4}, {c->V, 4}, {c->W, 4}, {c->sU, 4}, {c->sV, 4}, {c->sW, 4}};
    int rc = fv_halo_copy(c, vel, 6, (int)ceilf(c->prm.cfl_number) + 3);
    if (rc) return rc;
    if (c->np)
        hipLaunchKernelGGL(k_advect_particles, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                           c->U, c->V, c->W, c->sU, c->sV, c->sW, c->solid, c->dx, dt, c->prm.pic_ratio, b);
    HIPCHK(c, hipGetLastError());
    c->binsValid = 0;
    return fv_migrate_particles(c);
}
