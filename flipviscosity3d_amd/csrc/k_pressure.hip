// k_pressure.hip -- variational pressure projection solve (reference pressuresolver.cpp:166-567).
//
// Data layout: the reference compacts the pressure cells into a list + dense int key map
// (pressuresolver.cpp:196-225) and stores 4 float coefficients per cell (pressuresolver.h:103-110).
// Here the system lives on the dense (I,J,K) cell grid in Array3d order so a wave reads 64 consecutive
// cells = one 256-byte line per stream: four coefficient arrays diag/plusi/plusj/plusk that are ZERO
// outside the pressure cells (and towards non-pressure neighbours), so the 7-point SpMV needs no index map
// and no masks.  Only tiles that contain pressure cells are swept (tile list, pcg_common.h).
//
// Algorithmic traffic of the SpMV: 4 coefficient reads + 1 read of s + 1 write of z = 24 B per swept cell
// in fp32 (32 B with fp64 vectors); the three "minus" coefficients and the six neighbour values of s are
// re-used from L1/L2.
#include "flipv_internal.h"
#include "pcg_common.h"

#define GRID3(w, h, d) dim3(cdiv((w), 64), cdiv((h), 4), (unsigned)(d)), dim3(64, 4, 1)

__device__ __forceinline__ bool d_is_pcell(const float *__restrict__ phi, int i, int j, int k, int I, int J, int K) {
    // interior cells with phi < 0 (pressuresolver.cpp:206-216)
    return i >= 1 && j >= 1 && k >= 1 && i <= I - 2 && j <= J - 2 && k <= K - 2 && phi[DIDX(i, j, k, I, J)] < 0.0f;
}

// K11: coefficients (pressuresolver.cpp:248-322) and right-hand side (pressuresolver.cpp:227-246)
template <typename T>
__global__ void k_pressure_setup(const float *__restrict__ phi, const float *__restrict__ U,
                                 const float *__restrict__ V, const float *__restrict__ W,
                                 const float *__restrict__ wU, const float *__restrict__ wV,
                                 const float *__restrict__ wW, float *__restrict__ diag, float *__restrict__ pi,
                                 float *__restrict__ pj, float *__restrict__ pk, T *__restrict__ r, T *__restrict__ x,
                                 T *__restrict__ s, double *__restrict__ bmax, int I, int J, int K, float dxf, float dtf,
                                 float minfrac) {
    __shared__ double lds[4];
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    double babs = 0.0;
    if (i < I && j < J) {
        const size_t c = DIDX(i, j, k, I, J);
        float dg = 0.0f, ci = 0.0f, cj = 0.0f, ck = 0.0f;
        double b = 0.0;
        if (d_is_pcell(phi, i, j, k, I, J, K)) {
            const double dx = (double)dxf, dt = (double)dtf;
            const float scale = (float)(dt / (dx * dx));
            const float pc = phi[c];
            const size_t uR = DIDX(i + 1, j, k, I + 1, J), uL = DIDX(i, j, k, I + 1, J);
            const size_t vT = DIDX(i, j + 1, k, I, J + 1), vB = DIDX(i, j, k, I, J + 1);
            const size_t wF = DIDX(i, j, k + 1, I, J), wN = DIDX(i, j, k, I, J);
            float term, pn;
            // right
            term = wU[uR] * scale; pn = phi[DIDX(i + 1, j, k, I, J)];
            if (pn < 0) { dg += term; if (i + 1 <= I - 2) ci = -term; }
            else dg += term / fmaxf(d_frac2(pc, pn), minfrac);
            // left
            term = wU[uL] * scale; pn = phi[DIDX(i - 1, j, k, I, J)];
            if (pn < 0) dg += term; else dg += term / fmaxf(d_frac2(pn, pc), minfrac);
            // top
            term = wV[vT] * scale; pn = phi[DIDX(i, j + 1, k, I, J)];
            if (pn < 0) { dg += term; if (j + 1 <= J - 2) cj = -term; }
            else dg += term / fmaxf(d_frac2(pc, pn), minfrac);
            // bottom
            term = wV[vB] * scale; pn = phi[DIDX(i, j - 1, k, I, J)];
            if (pn < 0) dg += term; else dg += term / fmaxf(d_frac2(pn, pc), minfrac);
            // far
            term = wW[wF] * scale; pn = phi[DIDX(i, j, k + 1, I, J)];
            if (pn < 0) { dg += term; if (k + 1 <= K - 2) ck = -term; }
            else dg += term / fmaxf(d_frac2(pc, pn), minfrac);
            // near
            term = wW[wN] * scale; pn = phi[DIDX(i, j, k - 1, I, J)];
            if (pn < 0) dg += term; else dg += term / fmaxf(d_frac2(pn, pc), minfrac);
            // negative divergence: float products accumulated in fp64 (pressuresolver.cpp:236-243)
            b -= (double)(wU[uR] * U[uR]);
            b += (double)(wU[uL] * U[uL]);
            b -= (double)(wV[vT] * V[vT]);
            b += (double)(wV[vB] * V[vB]);
            b -= (double)(wW[wF] * W[wF]);
            b += (double)(wW[wN] * W[wN]);
            b /= dx;
            if (dg == 0.0f) b = 0.0;  // a cell with no open face has an all-zero row; keep it out of the system
        }
        diag[c] = dg; pi[c] = ci; pj[c] = cj; pk[c] = ck;
        r[c] = (T)b;
        x[c] = (T)0;
        s[c] = (T)0;
        babs = fabs(b);
    }
    const double bm = block_max_256(babs, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && bm > 0.0) atomic_max_nonneg(bmax, bm);
}

// K12: z = A s with fused s.z  (pressuresolver.cpp:464-499; same term order -i,+i,-j,+j,-k,+k, diagonal)
template <typename T>
__global__ __launch_bounds__(256) void k_pressure_spmv(const int *__restrict__ tiles, int ntiles, TileGrid tg,
                                                       const float *__restrict__ diag, const float *__restrict__ pi,
                                                       const float *__restrict__ pj, const float *__restrict__ pk,
                                                       const T *__restrict__ s, T *__restrict__ z, int I, int J, int K,
                                                       double *__restrict__ dA, const int *__restrict__ conv) {
    if (conv && *conv >= 0) return;
    __shared__ double lds[4];
    const int slot = d_tile_slot(blockIdx.x, ntiles);
    double acc = 0.0;
    if (slot < ntiles) {
        int i, j, k0;
        d_tile_coords(tiles[slot], tg, i, j, k0);
        if (i < I && j < J) {
            const size_t sy = (size_t)I, sz = (size_t)I * J;
            const int kend = min(k0 + TZ, K);
            for (int k = k0; k < kend; k++) {
                const size_t c = DIDX(i, j, k, I, J);
                const float dg = diag[c];
                T y = (T)0;
                if (dg != 0.0f) {  // unknowns are interior cells: all six neighbours exist
                    const T sc = s[c];
                    y = s[c - 1] * (T)pi[c - 1];
                    y += s[c + 1] * (T)pi[c];
                    y += s[c - sy] * (T)pj[c - sy];
                    y += s[c + sy] * (T)pj[c];
                    y += s[c - sz] * (T)pk[c - sz];
                    y += s[c + sz] * (T)pk[c];
                    y += sc * (T)dg;
                    acc += (double)sc * (double)y;
                }
                z[c] = y;
            }
        }
    }
    const double tot = block_sum_256(acc, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && tot != 0.0 && dA) atomicAdd(dA, tot);
}

__global__ void k_f64_to_f32(const double *__restrict__ a, float *__restrict__ o, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) o[t] = (float)a[t];
}

// ---- tile activity ----
// flag[t] = 1 if tile t holds at least one unknown (diag != 0) of any component
__global__ __launch_bounds__(256) void k_tile_flags(TileGrid tg, PcgComps cp, int *__restrict__ flag) {
    const int tile = blockIdx.x;
    int i, j, k0;
    d_tile_coords(tile, tg, i, j, k0);
    int any = 0;
    for (int c = 0; c < cp.n; c++) {
        if (i >= cp.w[c] || j >= cp.h[c]) continue;
        for (int kk = 0; kk < TZ; kk++) {
            const int k = k0 + kk;
            if (k >= cp.d[c]) break;
            any |= cp.diag[c][DIDX(i, j, k, cp.w[c], cp.h[c])] != 0.0f;
        }
    }
    const int r = __syncthreads_or(any);
    if (threadIdx.x == 0 && threadIdx.y == 0) flag[tile] = r;
}

// ordered compaction of the flagged tiles by one block (tile counts are 1e4..1e5)
__global__ __launch_bounds__(1024) void k_tile_compact(const int *__restrict__ flag, int ntiles, int *__restrict__ list,
                                                       int *__restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int start = 0; start < ntiles; start += 1024) {
        const int t = start + threadIdx.x;
        const int f = (t < ntiles) ? (flag[t] != 0) : 0;
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wv; q++) woff += wsum[q];
        int total = 0;
        for (int q = 0; q < 16; q++) total += wsum[q];
        if (f) list[base + woff + before] = t;
        __syncthreads();
        if (threadIdx.x == 0) base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

// ------------------------------------------------------------------------------------------------
int fv_scal_reserve(flipv_context *c, int cap) {
    const size_t need = (size_t)3 * (cap + 2) + 16;
    if (c->d_scal && c->h_scal && c->scalCap >= need) return FLIPV_OK;
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    HIPCHK(c, hipMalloc((void **)&c->d_scal, need * sizeof(double)));
    HIPCHK(c, hipHostMalloc((void **)&c->h_scal, need * sizeof(double)));
    c->scalCap = need;
    return FLIPV_OK;
}

int fv_build_tiles(flipv_context *c, const PcgComps &cp, int *list, int *nActive) {
    const int nt = c->tg.count();
    hipLaunchKernelGGL(k_tile_flags, dim3(nt), dim3(64, 4, 1), 0, c->stream, c->tg, cp, c->tileFlag);
    hipLaunchKernelGGL(k_tile_compact, dim3(1), dim3(1024), 0, c->stream, c->tileFlag, nt, list, c->d_flags + 1);
    HIPCHK(c, hipMemcpyAsync(c->h_flags + 1, c->d_flags + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *nActive = c->h_flags[1];
    return FLIPV_OK;
}

template <typename T>
static void launch_pressure_spmv(flipv_context *c, double *dA, const int *conv) {
    const Dims &d = c->d;
    const int nb = ((c->nActiveP + 7) / 8) * 8;
    if (c->prm.kernel_timing) fv_ev_begin(c, 0, (double)c->nActiveP * TX * TY * TZ);
    hipLaunchKernelGGL(k_pressure_spmv<T>, dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListP, c->nActiveP, c->tg,
                       c->pDiag, c->pPi, c->pPj, c->pPk, (const T *)c->pS, (T *)c->pZ, d.I, d.J, d.K, dA, conv);
    if (c->prm.kernel_timing) fv_ev_end(c);
}

template <typename T>
static int pressure_solve_t(flipv_context *c, float dt, flipv_solve_info *info) {
    const Dims &d = c->d;
    flipv_solve_info li;
    memset(&li, 0, sizeof(li));
    li.total_tiles = c->tg.count();
    const int cap = c->prm.pressure_max_iterations;
    int rc = fv_scal_reserve(c, cap);
    if (rc) return rc;
    const size_t nscal = (size_t)3 * (cap + 2) + 16;
    HIPCHK(c, hipMemsetAsync(c->d_scal, 0, nscal * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0xff, 4 * sizeof(int), c->stream));  // conv = -1
    PcgScal sc;
    sc.sigma = c->d_scal;
    sc.dA = c->d_scal + (cap + 2);
    sc.rmax = c->d_scal + 2 * (cap + 2);
    double *bmax = c->d_scal + 3 * (cap + 2);
    sc.conv = c->d_flags;
    sc.tol_inclusive = 0;

    // with fp32 vectors x IS the pressure grid
    T *x = std::is_same<T, float>::value ? (T *)c->pressure : (T *)c->pX;
    hipLaunchKernelGGL(k_pressure_setup<T>, GRID3(d.I, d.J, d.K), 0, c->stream, c->phi, c->U, c->V, c->W, c->wU, c->wV,
                       c->wW, c->pDiag, c->pPi, c->pPj, c->pPk, (T *)c->pR, x, (T *)c->pS, bmax, d.I, d.J, d.K, c->dx, dt,
                       c->prm.min_frac);
    HIPCHK(c, hipMemcpyAsync(c->h_scal, bmax, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PcgComps cp;
    memset(&cp, 0, sizeof(cp));
    cp.n = 1; cp.w[0] = d.I; cp.h[0] = d.J; cp.d[0] = d.K; cp.diag[0] = c->pDiag;
    rc = fv_build_tiles(c, cp, c->tileListP, &c->nActiveP);  // synchronises: h_scal[0] = max|b|
    if (rc) return rc;
    const double bnorm = c->h_scal[0];
    li.rhs_norm = bnorm;
    li.active_tiles = c->nActiveP;
    c->pressureReady = 1;
    c->pressurePrec = std::is_same<T, float>::value ? 0 : 1;
    c->lastDt = dt;
    // early out (pressuresolver.cpp:173-175): pressure grid is zero
    if (!(bnorm >= c->prm.pressure_tolerance) || c->nActiveP == 0) {
        li.status = 3;
        li.residual = bnorm;
        if (!std::is_same<T, float>::value)
            hipLaunchKernelGGL(k_f64_to_f32, dim3(2048), dim3(256), 0, c->stream, (const double *)x, c->pressure, d.nc());
        if (info) *info = li;
        return FLIPV_OK;
    }
    sc.tol = fmax(c->prm.pressure_tolerance, c->prm.pressure_rel_tolerance * bnorm);

    PcgVecs<T> v;
    memset(&v, 0, sizeof(v));
    v.x[0] = x; v.r[0] = (T *)c->pR; v.z[0] = (T *)c->pZ; v.s[0] = (T *)c->pS;
    const int nb = ((c->nActiveP + 7) / 8) * 8;
    const dim3 blk(64, 4, 1);
    hipLaunchKernelGGL(k_pcg_init<T>, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tg, cp, v, sc);
    const int every = c->prm.check_every > 0 ? c->prm.check_every : 8;
    int it = 0, conv = -1;
    while (it < cap && conv < 0) {
        const int stop = (it + every < cap) ? it + every : cap;
        for (; it < stop; it++) {
            launch_pressure_spmv<T>(c, sc.dA + it, sc.conv);
            hipLaunchKernelGGL(k_pcg_update<T>, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tg, cp, v, sc, it);
            hipLaunchKernelGGL(k_pcg_dir<T>, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tg, cp, v, sc, it);
        }
        HIPCHK(c, hipMemcpyAsync(c->h_flags, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        conv = c->h_flags[0];
    }
    const int last = conv >= 0 ? conv : cap - 1;
    HIPCHK(c, hipMemcpyAsync(c->h_scal, sc.rmax + last, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (!std::is_same<T, float>::value)
        hipLaunchKernelGGL(k_f64_to_f32, dim3(2048), dim3(256), 0, c->stream, (const double *)x, c->pressure, d.nc());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    li.iterations = conv >= 0 ? conv + 1 : cap;
    li.residual = c->h_scal[0];
    li.status = conv >= 0 ? 0 : 1;
    if (c->prm.kernel_timing) fv_ev_collect(c);
    if (info) *info = li;
    return conv >= 0 ? FLIPV_OK : FLIPV_WARN_NOT_CONVERGED;
}

int fv_pressure_solve(flipv_context *c, float dt, flipv_solve_info *info) {
    if (c->prm.precision == FLIPV_PRECISION_FP64) return pressure_solve_t<double>(c, dt, info);
    return pressure_solve_t<float>(c, dt, info);
}

int fv_bench_pressure_spmv(flipv_context *c, int reps, double *ms, double *cells) {
    if (!c->pressureReady || c->nActiveP <= 0) { c->err = "flipv_bench_spmv: run flipv_pressure_solve first"; return FLIPV_ERR_INVALID; }
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    const int saved = c->prm.kernel_timing;
    c->prm.kernel_timing = 0;
    for (int w = 0; w < 3; w++) {
        if (c->pressurePrec) launch_pressure_spmv<double>(c, nullptr, nullptr); else launch_pressure_spmv<float>(c, nullptr, nullptr);
    }
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; r++) {
        if (c->pressurePrec) launch_pressure_spmv<double>(c, nullptr, nullptr); else launch_pressure_spmv<float>(c, nullptr, nullptr);
    }
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipEventSynchronize(b));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, a, b));
    c->prm.kernel_timing = saved;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms = (double)t / reps;
    *cells = (double)c->nActiveP * TX * TY * TZ;
    return FLIPV_OK;
}
