// spmv7_variants.hip -- stand-alone micro-benchmark of the 7-point pressure SpMV q = A s on a FILLED box (every cell an unknown), k-marching
// variants, to decide the shape of k_pressure_spmv_march on boxes beyond the memory-side cache.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/spmv7_variants.hip -o tools/micro/spmv7_variants && tools/micro/spmv7_variants 512
// Layout = the library's: index (i, j, k) at i + PX (j + PY k), PX = roundup8(N + 1), PY = roundup4(N + 1), one guard plane each side.
// Arrays: diag, pi (coefficient towards i+1), pj, pk, s, q.  Algorithmic bytes: 24 per cell (5 reads + 1 write).
// A work unit = a column of tiles (256 i x R j) marched over `runlen` planes; block b takes unit d_slot(b): every XCD a contiguous eighth of the list.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
struct Geo { int N, PX, PY, ntx, nty, nkc, runlen; long sy, sz; };
__device__ __forceinline__ int d_slot(int b, int n) { const int per = (n + 7) >> 3; return (b & 7) * per + (b >> 3); }
template <bool NT> __device__ __forceinline__ v4 ld(const float *p) { return NT ? __builtin_nontemporal_load((const v4 *)p) : *(const v4 *)p; }
template <bool NT> __device__ __forceinline__ void st(float *p, v4 v) { if (NT) __builtin_nontemporal_store(v, (v4 *)p); else *(v4 *)p = v; }

// RPT rows per thread (adjacent in j): tile = 256 i x (4 RPT) j, block (64, 4).  j-neighbours inside a thread's rows come from registers;
// across threads / tiles from memory (L1 / L2 / the memory-side cache).  NTC: nontemporal coefficient loads and q stores; s always cached.
template <int RPT, bool NTC>
__global__ __launch_bounds__(256) void k_march(Geo g, const float *__restrict__ diag, const float *__restrict__ pi, const float *__restrict__ pj,
                                               const float *__restrict__ pk, const float *__restrict__ s, float *__restrict__ q, int nunits) {
    const int lane = threadIdx.x, ty = threadIdx.y;
    for (int b = blockIdx.x; b < ((nunits + 7) & ~7); b += gridDim.x) {
        const int u = d_slot(b, nunits);
        if (u >= nunits) continue;
        const int tx = u % g.ntx, r = u / g.ntx, tyy = r % g.nty, kc = r / g.nty;
        const int i0 = tx * 256 + lane * 4, j0 = tyy * (4 * RPT) + ty * RPT, k0 = kc * g.runlen;
        if (i0 >= g.N) continue;   // (whole wave-rows drop out together only when 256 does not divide N: lanes stay converged per row)
        const int klen = min(g.runlen, g.N - k0);
        bool rowok[RPT];
        size_t c[RPT];
#pragma unroll
        for (int t = 0; t < RPT; t++) { rowok[t] = j0 + t < g.N; c[t] = (size_t)(i0 + 8) + (size_t)g.PX * ((size_t)(j0 + t + 4) + (size_t)g.PY * (size_t)(k0 + 1)); }
        const bool first = lane == 0, last = lane == 63 || i0 + 4 >= g.N;
        v4 skm[RPT], sc[RPT], ckm[RPT];
#pragma unroll
        for (int t = 0; t < RPT; t++) { skm[t] = ld<false>(s + c[t] - g.sz); sc[t] = ld<false>(s + c[t]); ckm[t] = ld<NTC>(pk + c[t] - g.sz); }
        for (int e0 = 0; e0 < klen; e0++) {
            v4 dg[RPT], ci[RPT], cj[RPT], ck[RPT], skp[RPT];
#pragma unroll
            for (int t = 0; t < RPT; t++) {
                dg[t] = ld<NTC>(diag + c[t]); ci[t] = ld<NTC>(pi + c[t]); cj[t] = ld<false>(pj + c[t]); ck[t] = ld<NTC>(pk + c[t]); skp[t] = ld<false>(s + c[t] + g.sz);
            }
            const v4 sjm0 = ld<false>(s + c[0] - g.sy), cjm0 = ld<false>(pj + c[0] - g.sy), sjpL = ld<false>(s + c[RPT - 1] + g.sy);
            float esl[RPT], esr[RPT], ecl[RPT];
#pragma unroll
            for (int t = 0; t < RPT; t++) { esl[t] = first ? s[c[t] - 1] : 0.0f; ecl[t] = first ? pi[c[t] - 1] : 0.0f; esr[t] = last ? s[c[t] + 4] : 0.0f; }
#pragma unroll
            for (int t = 0; t < RPT; t++) {
                float sl = __shfl_up(sc[t].w, 1, 64), sr = __shfl_down(sc[t].x, 1, 64), cil = __shfl_up(ci[t].w, 1, 64);
                if (first) { sl = esl[t]; cil = ecl[t]; }
                if (last) sr = esr[t];
                const v4 sjm = t == 0 ? sjm0 : sc[t > 0 ? t - 1 : 0], cjm = t == 0 ? cjm0 : cj[t > 0 ? t - 1 : 0], sjp = t == RPT - 1 ? sjpL : sc[t < RPT - 1 ? t + 1 : 0];
                v4 y;
                y.x = sl * cil + sc[t].y * ci[t].x + sjm.x * cjm.x + sjp.x * cj[t].x + skm[t].x * ckm[t].x + skp[t].x * ck[t].x + sc[t].x * dg[t].x;
                y.y = sc[t].x * ci[t].x + sc[t].z * ci[t].y + sjm.y * cjm.y + sjp.y * cj[t].y + skm[t].y * ckm[t].y + skp[t].y * ck[t].y + sc[t].y * dg[t].y;
                y.z = sc[t].y * ci[t].y + sc[t].w * ci[t].z + sjm.z * cjm.z + sjp.z * cj[t].z + skm[t].z * ckm[t].z + skp[t].z * ck[t].z + sc[t].z * dg[t].z;
                y.w = sc[t].z * ci[t].z + sr * ci[t].w + sjm.w * cjm.w + sjp.w * cj[t].w + skm[t].w * ckm[t].w + skp[t].w * ck[t].w + sc[t].w * dg[t].w;
                if (rowok[t]) st<NTC>(q + c[t], y);
            }
#pragma unroll
            for (int t = 0; t < RPT; t++) { skm[t] = sc[t]; sc[t] = skp[t]; ckm[t] = ck[t]; c[t] += g.sz; }
        }
    }
}

// tile-at-a-time reference (no marching, every neighbour a load): the numerics check and the "what the caches do alone" baseline
__global__ __launch_bounds__(256) void k_plain(Geo g, const float *__restrict__ diag, const float *__restrict__ pi, const float *__restrict__ pj, const float *__restrict__ pk,
                                               const float *__restrict__ s, float *__restrict__ q) {
    const int i = (blockIdx.x * 64 + threadIdx.x) * 4, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= g.N || j >= g.N) return;
    const size_t c = (size_t)(i + 8) + (size_t)g.PX * ((size_t)(j + 4) + (size_t)g.PY * (size_t)(k + 1));
    for (int e = 0; e < 4; e++) {
        const size_t a = c + e;
        q[a] = s[a - 1] * pi[a - 1] + s[a + 1] * pi[a] + s[a - g.sy] * pj[a - g.sy] + s[a + g.sy] * pj[a] + s[a - g.sz] * pk[a - g.sz] + s[a + g.sz] * pk[a] + s[a] * diag[a];
    }
}


// tile-at-a-time, vectorised: one thread = 4 consecutive i of RPT adjacent rows of ONE plane, i-neighbours by lane shift, every j / k neighbour a 16-byte load
// (caches: the k-1 / k / k+1 planes of s and pk are 3-4 MB at 512^2 -- L2 / the memory-side cache hold them while the launch sweeps the box in ADDRESS order).
// ORDER 0: blocks sweep (tx, ty, k) x fastest, i.e. in address order over the whole chip; 1: every XCD sweeps its own contiguous eighth of the planes.
template <int RPT, bool NTC, int ORDER>
__global__ __launch_bounds__(256) void k_tile(Geo g, const float *__restrict__ diag, const float *__restrict__ pi, const float *__restrict__ pj,
                                              const float *__restrict__ pk, const float *__restrict__ s, float *__restrict__ q, int nunits) {
    const int lane = threadIdx.x, ty = threadIdx.y;
    const int b = blockIdx.x;
    const int u = ORDER == 0 ? b : d_slot(b, nunits);
    if (u >= nunits) return;
    const int tx = u % g.ntx, r = u / g.ntx, tyy = r % g.nty, k = r / g.nty;
    const int i0 = tx * 256 + lane * 4, j0 = tyy * (4 * RPT) + ty * RPT;
    if (i0 >= g.N) return;
    const bool first = lane == 0, last = lane == 63 || i0 + 4 >= g.N;
    size_t c[RPT];
    v4 dg[RPT], ci[RPT], cj[RPT], ck[RPT], ckm[RPT], sc[RPT], skm[RPT], skp[RPT];
#pragma unroll
    for (int t = 0; t < RPT; t++) {
        c[t] = (size_t)(i0 + 8) + (size_t)g.PX * ((size_t)(j0 + t + 4) + (size_t)g.PY * (size_t)(k + 1));
        dg[t] = ld<NTC>(diag + c[t]); ci[t] = ld<NTC>(pi + c[t]); cj[t] = ld<false>(pj + c[t]); ck[t] = ld<false>(pk + c[t]); ckm[t] = ld<false>(pk + c[t] - g.sz);
        sc[t] = ld<false>(s + c[t]); skm[t] = ld<false>(s + c[t] - g.sz); skp[t] = ld<false>(s + c[t] + g.sz);
    }
    const v4 sjm0 = ld<false>(s + c[0] - g.sy), cjm0 = ld<false>(pj + c[0] - g.sy), sjpL = ld<false>(s + c[RPT - 1] + g.sy);
    float esl[RPT], esr[RPT], ecl[RPT];
#pragma unroll
    for (int t = 0; t < RPT; t++) { esl[t] = first ? s[c[t] - 1] : 0.0f; ecl[t] = first ? pi[c[t] - 1] : 0.0f; esr[t] = last ? s[c[t] + 4] : 0.0f; }
#pragma unroll
    for (int t = 0; t < RPT; t++) {
        float sl = __shfl_up(sc[t].w, 1, 64), sr = __shfl_down(sc[t].x, 1, 64), cil = __shfl_up(ci[t].w, 1, 64);
        if (first) { sl = esl[t]; cil = ecl[t]; }
        if (last) sr = esr[t];
        const v4 sjm = t == 0 ? sjm0 : sc[t > 0 ? t - 1 : 0], cjm = t == 0 ? cjm0 : cj[t > 0 ? t - 1 : 0], sjp = t == RPT - 1 ? sjpL : sc[t < RPT - 1 ? t + 1 : 0];
        v4 y;
        y.x = sl * cil + sc[t].y * ci[t].x + sjm.x * cjm.x + sjp.x * cj[t].x + skm[t].x * ckm[t].x + skp[t].x * ck[t].x + sc[t].x * dg[t].x;
        y.y = sc[t].x * ci[t].x + sc[t].z * ci[t].y + sjm.y * cjm.y + sjp.y * cj[t].y + skm[t].y * ckm[t].y + skp[t].y * ck[t].y + sc[t].y * dg[t].y;
        y.z = sc[t].y * ci[t].y + sc[t].w * ci[t].z + sjm.z * cjm.z + sjp.z * cj[t].z + skm[t].z * ckm[t].z + skp[t].z * ck[t].z + sc[t].z * dg[t].z;
        y.w = sc[t].z * ci[t].z + sr * ci[t].w + sjm.w * cjm.w + sjp.w * cj[t].w + skm[t].w * ckm[t].w + skp[t].w * ck[t].w + sc[t].w * dg[t].w;
        if (j0 + t < g.N) st<NTC>(q + c[t], y);
    }
}
// the ceiling: 5 reads + 1 write of the same byte mix with no stencil at all (16 bytes per lane, 4 independent chunks per lane, nontemporal)
__global__ __launch_bounds__(256) void k_mix(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n) {
    for (size_t base = (size_t)blockIdx.x * 1024; base < n; base += (size_t)gridDim.x * 1024) {
        v4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const size_t t = base + (size_t)u * 256 + threadIdx.x; v[u] = v4{0, 0, 0, 0}; if (t < n) for (int m = 0; m < 5; m++) v[u] += __builtin_nontemporal_load(a + (size_t)m * n + t); }
#pragma unroll
        for (int u = 0; u < 4; u++) { const size_t t = base + (size_t)u * 256 + threadIdx.x; if (t < n) __builtin_nontemporal_store(v[u], b + t); }
    }
}
template <int RPT, bool NTC, int ORDER>
static double run_tile(const char *name, Geo g, float **d, const std::vector<float> &ref, std::vector<float> &out, size_t n) {
    g.nty = (g.N + 4 * RPT - 1) / (4 * RPT);
    const int nunits = g.ntx * g.nty * g.N;
    const int grid = ORDER == 0 ? nunits : ((nunits + 7) / 8) * 8;
    CK(hipMemset(d[5], 0, n * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_tile<RPT, NTC, ORDER>), dim3(grid), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5], nunits);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_tile<RPT, NTC, ORDER>), dim3(grid), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5], nunits);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(out.data(), d[5], n * sizeof(float), hipMemcpyDeviceToHost));
    double maxd = 0; for (size_t a = 0; a < n; a += 97) maxd = std::max(maxd, (double)fabsf(out[a] - ref[a]));
    const double us = ms * 1000.0 / reps, tb = 24.0 * (double)g.N * g.N * g.N / (us * 1e-6) / 1e12;
    printf("%-28s rows/thread %d nt %d order %d blocks %7d: %8.1f us  %.2f TB/s algorithmic = %.3f of 8 TB/s   (max diff to plain %.1e)\n", name, RPT, (int)NTC, ORDER, grid, us, tb, tb / 8.0, maxd);
    fflush(stdout);
    return us;
}

// persistent form of k_tile (what a library kernel with fused dot products needs: one atomic per block, not per tile): G blocks, block b takes units b, b + G, ...
// so that at any moment the resident blocks work on neighbouring units -- the sweep stays in address order.  DOT: accumulate s.q per block and add it to 128 slots.
template <int RPT, bool NTC, bool DOT>
__global__ __launch_bounds__(256) void k_tile_persistent(Geo g, const float *__restrict__ diag, const float *__restrict__ pi, const float *__restrict__ pj,
                                                         const float *__restrict__ pk, const float *__restrict__ s, float *__restrict__ q, int nunits, double *__restrict__ slots) {
    const int lane = threadIdx.x, ty = threadIdx.y;
    double acc = 0.0;
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        const int tx = u % g.ntx, r = u / g.ntx, tyy = r % g.nty, k = r / g.nty;
        const int i0 = tx * 256 + lane * 4, j0 = tyy * (4 * RPT) + ty * RPT;
        if (i0 >= g.N) continue;
        const bool first = lane == 0, last = lane == 63 || i0 + 4 >= g.N;
        size_t c[RPT];
        v4 dg[RPT], ci[RPT], cj[RPT], ck[RPT], ckm[RPT], sc[RPT], skm[RPT], skp[RPT];
#pragma unroll
        for (int t = 0; t < RPT; t++) {
            c[t] = (size_t)(i0 + 8) + (size_t)g.PX * ((size_t)(j0 + t + 4) + (size_t)g.PY * (size_t)(k + 1));
            dg[t] = ld<NTC>(diag + c[t]); ci[t] = ld<NTC>(pi + c[t]); cj[t] = ld<false>(pj + c[t]); ck[t] = ld<false>(pk + c[t]); ckm[t] = ld<false>(pk + c[t] - g.sz);
            sc[t] = ld<false>(s + c[t]); skm[t] = ld<false>(s + c[t] - g.sz); skp[t] = ld<false>(s + c[t] + g.sz);
        }
        const v4 sjm0 = ld<false>(s + c[0] - g.sy), cjm0 = ld<false>(pj + c[0] - g.sy), sjpL = ld<false>(s + c[RPT - 1] + g.sy);
        float esl[RPT], esr[RPT], ecl[RPT];
#pragma unroll
        for (int t = 0; t < RPT; t++) { esl[t] = first ? s[c[t] - 1] : 0.0f; ecl[t] = first ? pi[c[t] - 1] : 0.0f; esr[t] = last ? s[c[t] + 4] : 0.0f; }
        float ta = 0.0f;
#pragma unroll
        for (int t = 0; t < RPT; t++) {
            float sl = __shfl_up(sc[t].w, 1, 64), sr = __shfl_down(sc[t].x, 1, 64), cil = __shfl_up(ci[t].w, 1, 64);
            if (first) { sl = esl[t]; cil = ecl[t]; }
            if (last) sr = esr[t];
            const v4 sjm = t == 0 ? sjm0 : sc[t > 0 ? t - 1 : 0], cjm = t == 0 ? cjm0 : cj[t > 0 ? t - 1 : 0], sjp = t == RPT - 1 ? sjpL : sc[t < RPT - 1 ? t + 1 : 0];
            v4 y;
            y.x = sl * cil + sc[t].y * ci[t].x + sjm.x * cjm.x + sjp.x * cj[t].x + skm[t].x * ckm[t].x + skp[t].x * ck[t].x + sc[t].x * dg[t].x;
            y.y = sc[t].x * ci[t].x + sc[t].z * ci[t].y + sjm.y * cjm.y + sjp.y * cj[t].y + skm[t].y * ckm[t].y + skp[t].y * ck[t].y + sc[t].y * dg[t].y;
            y.z = sc[t].y * ci[t].y + sc[t].w * ci[t].z + sjm.z * cjm.z + sjp.z * cj[t].z + skm[t].z * ckm[t].z + skp[t].z * ck[t].z + sc[t].z * dg[t].z;
            y.w = sc[t].z * ci[t].z + sr * ci[t].w + sjm.w * cjm.w + sjp.w * cj[t].w + skm[t].w * ckm[t].w + skp[t].w * ck[t].w + sc[t].w * dg[t].w;
            if (j0 + t < g.N) { st<NTC>(q + c[t], y); if (DOT) ta += sc[t].x * y.x + sc[t].y * y.y + sc[t].z * y.z + sc[t].w * y.w; }
        }
        acc += (double)ta;
    }
    if (DOT) {
        __shared__ double red[4];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        const int tid = threadIdx.y * 64 + threadIdx.x;
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) atomicAdd(slots + (blockIdx.x & 127), red[0] + red[1] + red[2] + red[3]);
    }
}
template <int RPT, bool NTC, bool DOT>
static double run_persistent(const char *name, Geo g, int G, float **d, const std::vector<float> &ref, std::vector<float> &out, size_t n, double *slots) {
    g.nty = (g.N + 4 * RPT - 1) / (4 * RPT);
    const int nunits = g.ntx * g.nty * g.N;
    CK(hipMemset(d[5], 0, n * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_tile_persistent<RPT, NTC, DOT>), dim3(G), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5], nunits, slots);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_tile_persistent<RPT, NTC, DOT>), dim3(G), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5], nunits, slots);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(out.data(), d[5], n * sizeof(float), hipMemcpyDeviceToHost));
    double maxd = 0; for (size_t a = 0; a < n; a += 97) maxd = std::max(maxd, (double)fabsf(out[a] - ref[a]));
    const double us = ms * 1000.0 / reps, tb = 24.0 * (double)g.N * g.N * g.N / (us * 1e-6) / 1e12;
    printf("%-28s rows/thread %d nt %d dot %d blocks %7d: %8.1f us  %.2f TB/s algorithmic = %.3f of 8 TB/s   (max diff to plain %.1e)\n", name, RPT, (int)NTC, (int)DOT, G, us, tb, tb / 8.0, maxd);
    fflush(stdout);
    return us;
}

template <int RPT, bool NTC>
static double run_variant(const char *name, Geo g, int runlen, int blocks, float **d, const std::vector<float> &ref, std::vector<float> &out, size_t n) {
    g.runlen = runlen; g.nty = (g.N + 4 * RPT - 1) / (4 * RPT); g.nkc = (g.N + runlen - 1) / runlen;
    const int nunits = g.ntx * g.nty * g.nkc;
    const int grid = ((std::min(blocks, nunits) + 7) / 8) * 8;
    CK(hipMemset(d[5], 0, n * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_march<RPT, NTC>), dim3(grid), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5], nunits);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_march<RPT, NTC>), dim3(grid), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5], nunits);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(out.data(), d[5], n * sizeof(float), hipMemcpyDeviceToHost));
    double maxd = 0; for (size_t a = 0; a < n; a += 97) maxd = std::max(maxd, (double)fabsf(out[a] - ref[a]));
    const double us = ms * 1000.0 / reps, tb = 24.0 * (double)g.N * g.N * g.N / (us * 1e-6) / 1e12;
    printf("%-28s rows/thread %d nt %d runlen %3d blocks %4d units %5d: %8.1f us  %.2f TB/s algorithmic = %.3f of 8 TB/s   (max diff to plain %.1e)\n", name, RPT, (int)NTC, runlen, grid, nunits, us, tb, tb / 8.0, maxd);
    fflush(stdout);
    return us;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 512;
    const int pad = argc > 2 ? atoi(argv[2]) : 8;   // (the library pads a row by 8 - N % 8 ... entries and a plane by 4 rows: pad 0 = its strides at N = 512)
    Geo g; g.N = N; g.PX = ((N + 1 + 7) / 8) * 8 + pad; g.PY = ((N + 1 + 3) / 4) * 4 + pad; g.sy = g.PX; g.sz = (long)g.PX * g.PY; g.ntx = (N + 255) / 256; g.nty = 0; g.nkc = 0; g.runlen = 64;
    const size_t n = (size_t)g.sz * (N + 3);
    float *d[6];
    std::vector<float> h(n);
    for (int a = 0; a < 6; a++) {
        CK(hipMalloc(&d[a], n * sizeof(float)));
        // coefficients and s zero outside the cells (as in the library: no branches on the lattice's edge)
        std::fill(h.begin(), h.end(), 0.0f);
        if (a < 5)
            for (int k = 0; k < N; k++) for (int j = 0; j < N; j++) { size_t c = (size_t)8 + (size_t)g.PX * ((size_t)(j + 4) + (size_t)g.PY * (size_t)(k + 1)); for (int i = 0; i < N; i++) h[c + i] = (a == 0 ? 6.0f : -1.0f) * (1.0f + 0.001f * ((i * 7 + j * 13 + k * 29 + a * 3) % 17)); }
        if (a == 1) for (int k = 0; k < N; k++) for (int j = 0; j < N; j++) h[(size_t)8 + (N - 1) + (size_t)g.PX * ((size_t)(j + 4) + (size_t)g.PY * (size_t)(k + 1))] = 0.0f;   // no coupling past the last cell
        if (a == 2) for (int k = 0; k < N; k++) for (int i = 0; i < N; i++) h[(size_t)8 + i + (size_t)g.PX * ((size_t)(N - 1 + 4) + (size_t)g.PY * (size_t)(k + 1))] = 0.0f;
        if (a == 3) for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) h[(size_t)8 + i + (size_t)g.PX * ((size_t)(j + 4) + (size_t)g.PY * (size_t)(N - 1 + 1))] = 0.0f;
        CK(hipMemcpy(d[a], h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    std::vector<float> ref(n), out(n);
    hipLaunchKernelGGL(k_plain, dim3((N + 255) / 256, (N + 3) / 4, N), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5]);
    CK(hipDeviceSynchronize());
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_plain, dim3((N + 255) / 256, (N + 3) / 4, N), dim3(64, 4), 0, 0, g, d[0], d[1], d[2], d[3], d[4], d[5]);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain (no marching, scalar accesses)  %8.1f us  %.2f TB/s\n", ms * 200.0, 24.0 * (double)N * N * N / (ms * 200e-6) / 1e12);
    }
    CK(hipMemcpy(ref.data(), d[5], n * sizeof(float), hipMemcpyDeviceToHost));
    {   // the stencil-free ceiling of the same byte mix
        const size_t nn = (size_t)N * N * N / 4;
        v4 *a, *b; CK(hipMalloc(&a, 5 * nn * 16)); CK(hipMalloc(&b, nn * 16)); CK(hipMemset(a, 0, 5 * nn * 16));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int grid : {2048, 4096, 8192}) {
            hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, a, b, nn);
            CK(hipEventRecord(e0));
            for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, a, b, nn);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("stencil-free 5 reads + 1 write, %d blocks: %8.1f us  %.2f TB/s\n", grid, ms * 100.0, 24.0 * (double)N * N * N / (ms * 100e-6) / 1e12);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    double *slots; CK(hipMalloc(&slots, 128 * sizeof(double))); CK(hipMemset(slots, 0, 128 * sizeof(double)));
    run_tile<2, true, 0>("tile 256x8 nt", g, d, ref, out, n);
    for (int G : {1024, 1280, 2048, 2560, 4096, 8192}) {
        run_persistent<2, true, false>("persistent 256x8 nt", g, G, d, ref, out, n, slots);
        run_persistent<2, true, true>("persistent 256x8 nt dot", g, G, d, ref, out, n, slots);
    }
    {   // one block per unit WITH the fused dot product and its atomic (the persistent kernel launched with as many blocks as units)
        Geo g2 = g; g2.nty = (g.N + 7) / 8;
        run_persistent<2, true, true>("one block per unit, dot", g, g2.ntx * g2.nty * g.N, d, ref, out, n, slots);
        run_persistent<2, true, false>("one block per unit", g, g2.ntx * g2.nty * g.N, d, ref, out, n, slots);
    }
    run_persistent<1, true, true>("persistent 256x4 nt dot", g, 2048, d, ref, out, n, slots);
    run_persistent<4, true, true>("persistent 256x16 nt dot", g, 1280, d, ref, out, n, slots);
    run_persistent<2, false, true>("persistent 256x8 dot", g, 2048, d, ref, out, n, slots);
    return 0;
}
