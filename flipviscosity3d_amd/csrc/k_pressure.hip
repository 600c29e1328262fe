// k_pressure.hip -- variational pressure projection solve (reference pressuresolver.cpp:166-567).
//
// Data layout: the reference compacts the pressure cells into a list + dense int key map
// (pressuresolver.cpp:196-225) and stores 4 float coefficients per cell (pressuresolver.h:103-110).
// Here the system lives on the dense cell lattice of the shared index space: four coefficient arrays
// diag/plusi/plusj/plusk that are ZERO outside the pressure cells (and towards non-pressure neighbours), so the
// 7-point SpMV needs no index map, no masks and no bounds checks (guard zones + zero coefficients).
// Only tiles that contain pressure cells are swept (tile list, pcg_common.h).
//
// Algorithmic traffic of the SpMV: 4 coefficient reads + s read, q written = 24 B per swept cell in fp32
// (SURVEY.md 8d), plus 8 B for the fp64 residual that feeds the fused beta dot products.
#include "flipv_internal.h"
#include "pcg_common.h"
#include "flipv_comm.h"

__device__ __forceinline__ bool d_is_pcell(const float *__restrict__ phi, const Lay &L, int i, int j, int k) {
    // interior cells with phi < 0 (pressuresolver.cpp:206-216)
    return i >= 1 && j >= 1 && k >= 1 && i <= L.I - 2 && j <= L.J - 2 && k <= L.K - 2 && phi[gidx(L, i, j, k)] < 0.0f;
}

// K11: coefficients (pressuresolver.cpp:248-322) and right-hand side (pressuresolver.cpp:227-246)
template <typename T>
__global__ void k_pressure_setup(Lay L, const float *__restrict__ phi, const float *__restrict__ U,
                                 const float *__restrict__ V, const float *__restrict__ W,
                                 const float *__restrict__ wU, const float *__restrict__ wV,
                                 const float *__restrict__ wW, float *__restrict__ diag, float *__restrict__ pi,
                                 float *__restrict__ pj, float *__restrict__ pk, RT<T> *__restrict__ r, T *__restrict__ x,
                                 T *__restrict__ s, uint8_t *__restrict__ cellmask, double *__restrict__ bmax,
                                 int *__restrict__ ncells, float dxf, float dtf, float minfrac) {
    __shared__ double lds[4];
    IJK_OF_THREAD(L);
    double babs = 0.0;
    int cells = 0;
    if (i < L.ie && j < L.je) {
        const size_t c = gidx(L, i, j, k);
        float dg = 0.0f, ci = 0.0f, cj = 0.0f, ck = 0.0f;
        double b = 0.0;
        if (i < L.I && j < L.J && k < L.K && d_is_pcell(phi, L, i, j, k)) {
            const double dx = (double)dxf, dt = (double)dtf;
            const float scale = (float)(dt / (dx * dx));
            const long sy = L.sy, sz = L.sz;
            const float pc = phi[c];
            float term, pn;
            // right
            term = wU[c + 1] * scale; pn = phi[c + 1];
            if (pn < 0) { dg += term; if (i + 1 <= L.I - 2) ci = -term; }
            else dg += term / fmaxf(d_frac2(pc, pn), minfrac);
            // left
            term = wU[c] * scale; pn = phi[c - 1];
            if (pn < 0) dg += term; else dg += term / fmaxf(d_frac2(pn, pc), minfrac);
            // top
            term = wV[c + sy] * scale; pn = phi[c + sy];
            if (pn < 0) { dg += term; if (j + 1 <= L.J - 2) cj = -term; }
            else dg += term / fmaxf(d_frac2(pc, pn), minfrac);
            // bottom
            term = wV[c] * scale; pn = phi[c - sy];
            if (pn < 0) dg += term; else dg += term / fmaxf(d_frac2(pn, pc), minfrac);
            // far
            term = wW[c + sz] * scale; pn = phi[c + sz];
            if (pn < 0) { dg += term; if (k + 1 <= L.K - 2) ck = -term; }
            else dg += term / fmaxf(d_frac2(pc, pn), minfrac);
            // near
            term = wW[c] * scale; pn = phi[c - sz];
            if (pn < 0) dg += term; else dg += term / fmaxf(d_frac2(pn, pc), minfrac);
            // negative divergence: float products accumulated in fp64 (pressuresolver.cpp:236-243)
            b -= (double)(wU[c + 1] * U[c + 1]);
            b += (double)(wU[c] * U[c]);
            b -= (double)(wV[c + sy] * V[c + sy]);
            b += (double)(wV[c] * V[c]);
            b -= (double)(wW[c + sz] * W[c + sz]);
            b += (double)(wW[c] * W[c]);
            b /= dx;
            if (dg == 0.0f) b = 0.0;  // a cell with no open face has an all-zero row; keep it out of the system
        }
        diag[c] = dg; pi[c] = ci; pj[c] = cj; pk[c] = ck;
        cellmask[c] = (uint8_t)(dg != 0.0f);
        cells = dg != 0.0f && d_owned(L, i, j, k);
        r[c] = (RT<T>)b;
        x[c] = (T)0;
        s[c] = (T)0;
        babs = d_owned(L, i, j, k) ? fabs(b) : 0.0;
    }
    const double bm = block_max_256(babs, lds);
    const double nc = block_sum_256((double)cells, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        if (bm > 0.0) atomic_max_nonneg(bmax, bm);
        if (nc > 0.0) atomicAdd(ncells, (int)nc);
    }
}

template <typename T>
static __global__ void k_copy_to_f32(const T *__restrict__ a, float *__restrict__ o, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) o[t] = (float)a[t];
}

// ---- tile activity ----
// Tiles are enumerated (and therefore listed, and therefore scheduled) column by column: a column is JCH tile rows
// (16 grid rows) wide in j and runs through all owned k-planes.  Consecutive blocks of an XCD then work on consecutive
// planes of the same narrow column, so the k+-1 planes every stencil row re-reads are still in that XCD's 4 MiB L2
// (a whole k-plane of the ~19 arrays the viscosity SpMV touches is ~5 MB and would not be).
#ifndef FLIPV_JCH
#define FLIPV_JCH 4
#endif
constexpr int JCH = FLIPV_JCH;
__device__ __forceinline__ int d_virtual_tile(int v, const TileGrid &tg, int k0, int nk) {
    const int tx = v % tg.ntx;
    int r = v / tg.ntx;
    const int tyi = r % JCH;
    r /= JCH;
    const int kk = r % nk, chunk = r / nk;
    const int ty = chunk * JCH + tyi;
    if (ty >= tg.nty) return -1;
    return tx + tg.ntx * (ty + tg.nty * (k0 + kk));   // k0 = 0: tile planes count from the first owned plane (TileGrid::oz)
}

// ---- the kernels that depend on the tile geometry, once per geometry (pcg_geo.inc)
namespace g16 {
constexpr int ROWL = 16;
#include "pcg_geo.inc"
#include "k_pressure_geo.inc"
}  // namespace g16
namespace g64 {
constexpr int ROWL = 64;
#include "pcg_geo.inc"
#include "k_pressure_geo.inc"
}  // namespace g64


// K12 on FILLED boxes beyond the memory-side cache: the box swept in ADDRESS order, tile by tile, two adjacent rows per lane.
// The k-marching kernel above keeps 1 024 blocks walking 1 024 different columns of the box: 6 000 concurrent streams of 1 KB pieces a plane apart, which HBM serves at
// 4.4-4.5 TB/s however the march is shaped (tools/micro/spmv7_variants.hip, 512^3: 256 x 4 / x 8 / x 16 tiles, runs of 32-128 planes, with or without nontemporal
// accesses: 0.55-0.62 of the 8 TB/s peak), while a stencil-free stream of the same 5 reads + 1 write reaches 5.6-5.85 (0.70-0.73).  What the march saves -- the k +- 1
// planes of s and pk fetched once -- the caches give for free when ALL blocks work on the same few planes at the same time: three planes of s are 3 MB at 512^2.
// So: units (tx, pair of tile rows, k) enumerated x fastest = in address order, one block per unit (launch_pressure_spmv says why), a lane owning 4 consecutive i of TWO adjacent rows (the j-neighbours between them come from registers: 19 vector loads per 8 cells
// instead of 22), coefficients and q nontemporal.  Same arithmetic, term by term, as the kernels above.  Measured: 5.3 TB/s = 0.66 of peak at 512^3 (march: 0.557).
// Single-rank contexts (a block context's tiles split into interior / cut-face lists).
template <typename T, int DOTS, int RL>
__global__ __launch_bounds__(256) void k_pressure_spmv_sweep(TileGrid tg, Lay L, int k0, int nk, const float *__restrict__ diag, const float *__restrict__ pi,
                                                             const float *__restrict__ pj, const float *__restrict__ pk, const T *__restrict__ s,
                                                             const RT<T> *__restrict__ r, T *__restrict__ q, PcgScal sc, int it_arg) {
    // RL lanes along i: 64 (a wave = one row of 256 indices: one contiguous 1 KB per access) or 32 (a wave = 2 row groups of 128 indices) where the extent fits 256-wide
    // tiles badly (384 = 1.5 x 256: a quarter of the lanes idle; launch_pressure_spmv picks).  Tile = (4 RL) i x (8 x 64 / RL) j, two adjacent rows per lane.
    constexpr int RG = 64 / RL, RPB = 8 * RG, TW = 4 * RL;
    __shared__ double lds[12];
    bool stop;
    const int it = d_iter_spmv(sc, it_arg, stop);
    if (stop) return;
    // Units and XCDs.  The hardware deals consecutive workgroups to the 8 XCDs in turn, and each XCD has its own L2: a unit's k +- 1 and j +- 1 neighbour rows are L2
    // hits only if the units that own them run on the SAME XCD.  So XCD x = blockIdx & 7 takes a contiguous slab of the plane's row pairs, for every plane, and walks
    // it tx fastest, then rows, then k: all 8 XCDs sweep the same planes at the same time (DRAM sees one front), and inside a slab every neighbour but the two border
    // rows is the XCD's own.  (Units dealt out in plain address order put plane k + 1 on another XCD whenever the units per plane are no multiple of 8 -- 195 at
    // 512^3 with the padded columns: PMC 1.7 x the bytes, 0.55 of peak; this mapping: profiles/r5.)
    const int ntx = (L.I - tg.ox + TW - 1) / TW, nrow2 = (L.J - tg.oy + RPB - 1) / RPB;   // tiles that hold cells (not the padding behind the last cell)
    const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
    const int base = nrow2 >> 3, rem = nrow2 & 7, r0s = x * base + (x < rem ? x : rem), cnt = base + (x < rem ? 1 : 0);
    const int per = ntx * cnt;
    const int lane = (int)threadIdx.x & (RL - 1), rg = (int)threadIdx.x / RL;
    const long sy = L.sy, sz = L.sz;
    double da = 0.0, db = 0.0, dc = 0.0;
    if (cnt > 0 && l < per * nk) {
        const int kz = l / per, rr = l - kz * per, ty2 = r0s + rr / ntx, tx = rr % ntx;
        const int i0 = tg.ox + tx * TW + lane * 4, j0 = tg.oy + ty2 * RPB + ((int)threadIdx.y * RG + rg) * 2, k = k0 + kz;
        // Lanes beyond the allocated box stay ACTIVE with `inbox` false -- they load and store nothing, their vectors are zero -- so that the lane shifts below never read a
        // lane the branch has masked out (a lane inside the box never needs more than zeros from them: every coefficient towards an index without a cell is zero,
        // the invariant of k_pressure_setup that this kernel and the march rely on -- diag, pi, pj, pk and s are exactly 0 wherever there is no cell).
        const bool inbox = i0 < L.ox + L.PX && j0 + 1 < L.oy + L.PY;
        {
        const bool lfirst = inbox && lane == 0 && i0 > 0, llast = inbox && lane == RL - 1 && i0 + 4 < L.I;
        const size_t c0 = inbox ? gidx(L, i0, j0, k) : 0, c1 = c0 + sy;
        const bool own0 = inbox && i0 < L.I && j0 < L.J, own1 = inbox && i0 < L.I && j0 + 1 < L.J;   // rows / lanes that hold cells; the others load nothing (every coefficient towards them is zero)
        Vec<float, 4> dg0{}, dg1{}, ci0{}, ci1{}, cj0{}, cj1{}, cjm{}, ck0{}, ck1{}, ckm0{}, ckm1{};
        Vec<T, 4> s0{}, s1{}, sjm{}, sjp{}, skm0{}, skm1{}, skp0{}, skp1{};
        Vec<RT<T>, 4> r0{}, r1{};
        if (own0) {
            dg0 = ldvs<true, 4>(diag + c0); ci0 = ldvs<true, 4>(pi + c0); cj0 = ldv<4>(pj + c0); cjm = ldv<4>(pj + c0 - sy); ck0 = ldv<4>(pk + c0); ckm0 = ldv<4>(pk + c0 - sz);
            s0 = ldv<4>(s + c0); sjm = ldv<4>(s + c0 - sy); skm0 = ldv<4>(s + c0 - sz); skp0 = ldv<4>(s + c0 + sz);
            if (DOTS == 2) r0 = ldv<4>(r + c0);
        }
        if (own1) {
            dg1 = ldvs<true, 4>(diag + c1); ci1 = ldvs<true, 4>(pi + c1); cj1 = ldv<4>(pj + c1); ck1 = ldv<4>(pk + c1); ckm1 = ldv<4>(pk + c1 - sz);
            s1 = ldv<4>(s + c1); sjp = ldv<4>(s + c1 + sy); skm1 = ldv<4>(s + c1 - sz); skp1 = ldv<4>(s + c1 + sz);
            if (DOTS == 2) r1 = ldv<4>(r + c1);
        }
        T esl0 = (T)0, esl1 = (T)0, esr0 = (T)0, esr1 = (T)0;
        float ecl0 = 0.0f, ecl1 = 0.0f;
        if (lfirst && own0) { esl0 = s[c0 - 1]; ecl0 = pi[c0 - 1]; }
        if (lfirst && own1) { esl1 = s[c1 - 1]; ecl1 = pi[c1 - 1]; }
        if (llast && own0) esr0 = s[c0 + 4];
        if (llast && own1) esr1 = s[c1 + 4];
        T ta = (T)0, tb = (T)0, tc = (T)0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const Vec<float, 4> &dg = t ? dg1 : dg0, &ci = t ? ci1 : ci0, &cj = t ? cj1 : cj0, &cjmm = t ? cj0 : cjm, &ck = t ? ck1 : ck0, &ckm = t ? ckm1 : ckm0;
            const Vec<T, 4> &sc4 = t ? s1 : s0, &sm = t ? s0 : sjm, &sp = t ? sjp : s1, &skm = t ? skm1 : skm0, &skp = t ? skp1 : skp0;
            const Vec<RT<T>, 4> &r4 = t ? r1 : r0;
            // (32-lane rows shift across the whole wave like 64-lane ones: the first / last lane of a row takes its neighbour from memory either way)
            T sl = RL != 16 ? g64::wave_up1(sc4.v[3]) : g16::wave_up1(sc4.v[3]), sr = RL != 16 ? g64::wave_down1(sc4.v[0]) : g16::wave_down1(sc4.v[0]);
            float cil = RL != 16 ? g64::wave_up1(ci.v[3]) : g16::wave_up1(ci.v[3]);
            if (lane == 0) { sl = t ? esl1 : esl0; cil = t ? ecl1 : ecl0; }
            if (lane == RL - 1) sr = t ? esr1 : esr0;
            const bool own = t ? own1 : own0;
            Vec<T, 4> y;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const T smi = e > 0 ? sc4.v[e - 1] : sl;
                const T spi = e < 3 ? sc4.v[e + 1] : sr;
                const float cim = e > 0 ? ci.v[e - 1] : cil;
                T acc = smi * (T)cim;
                acc += spi * (T)ci.v[e];
                acc += sm.v[e] * (T)cjmm.v[e];
                acc += sp.v[e] * (T)cj.v[e];
                acc += skm.v[e] * (T)ckm.v[e];
                acc += skp.v[e] * (T)ck.v[e];
                acc += sc4.v[e] * (T)dg.v[e];
                y.v[e] = acc;
                if (own && dg.v[e] != 0.0f) {
                    ta += sc4.v[e] * acc;
                    if (DOTS >= 1) {
                        const T yi = acc * d_recip<T>(dg.v[e]);
                        if (DOTS == 2) tb += (T)r4.v[e] * yi;
                        tc += acc * yi;
                    }
                }
            }
            if (own) stvs<true, 4>(q + (t ? c1 : c0), y);
        }
        da += (double)ta; db += (double)tb; dc += (double)tc;
        }
    }
    block_sum3_256(da, db, dc, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && sc.conv) {
        const int sl = sc.my_slot();
        if (da != 0.0) atomicAdd(sc.a(it) + sl, da);
        if (db != 0.0) atomicAdd(sc.b(it) + sl, db);
        if (dc != 0.0) atomicAdd(sc.c(it) + sl, dc);
    }
}

// ---- run candidates (pcg_geo.inc explains runs)

// flag per (column, k-chunk): first active plane and the length up to the last active one (holes are walked)
__global__ __launch_bounds__(256) void k_run_flags(TileGrid tg, const int *__restrict__ vflag, int k0, int nk, int jch, int runlen,
                                                   int nchunk, Run *__restrict__ cand) {
    const int r = blockIdx.x * 256 + threadIdx.x;   // candidate index: tx + ntx*(ty + nty*chunk)
    const int ncol = tg.ntx * tg.nty;
    if (r >= ncol * nchunk) return;
    const int col = r % ncol, ch = r / ncol;
    const int tx = col % tg.ntx, ty = col / tg.ntx;
    const int tyi = ty % jch, ychunk = ty / jch;
    int first = -1, last = -1;
    for (int q = 0; q < runlen; q++) {
        const int kk = ch * runlen + q;
        if (kk >= nk) break;
        const int v = tx + tg.ntx * (tyi + jch * (kk + nk * ychunk));   // d_virtual_tile's enumeration (k_pressure.hip)
        if (vflag[v]) { if (first < 0) first = kk; last = kk; }
    }
    Run o;
    o.tile = first >= 0 ? tx + tg.ntx * (ty + tg.nty * (k0 + first)) : -1;
    o.len = first >= 0 ? last - first + 1 : 0;
    cand[r] = o;
}

// ordered compaction of the candidates by one block; count[0] = runs, count[1] = planes walked in total
__global__ __launch_bounds__(1024) void k_run_compact(const Run *__restrict__ cand, int ncand, Run *__restrict__ runs, int *__restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base, planes;
    if (threadIdx.x == 0) { base = 0; planes = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int acc = 0;
    for (int start = 0; start < ncand; start += 1024) {
        const int t = start + threadIdx.x;
        Run r{-1, 0};
        if (t < ncand) r = cand[t];
        const int f = r.len > 0;
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int woff = 0, total = 0;
        for (int q = 0; q < 16; q++) { if (q < wv) woff += wsum[q]; total += wsum[q]; }
        if (f) { runs[base + woff + before] = r; acc += r.len; }
        __syncthreads();
        if (threadIdx.x == 0) base += total;
        __syncthreads();
    }
    if (acc) atomicAdd(&planes, acc);
    __syncthreads();
    if (threadIdx.x == 0) { count[0] = base; count[1] = planes; }
}


// ordered compaction of the flagged tiles by one block (tile counts are 1e4..1e5)
// mode 0: every flagged tile; 1: only INTERIOR tiles -- tiles none of whose indices sits next to a cut face of the rank's box
// (cut[2 a] / cut[2 a + 1]: the lower / upper face of axis a has a neighbouring rank) --; 2: only the tiles at cut faces, appended
// after *prev entries (multi-rank: the SpMV of the interior tiles overlaps the halo exchange)
struct CutMask { int cut[6]; };
__global__ __launch_bounds__(1024) void k_tile_compact(const int *__restrict__ flag, int ntiles, int *__restrict__ list,
                                                       int *__restrict__ count, TileGrid tg, int k0, int nk, int mode, CutMask cm,
                                                       const int *__restrict__ prev, int tw, int th, int wdom, int hdom,
                                                       int *__restrict__ lanesIn) {
    // lanesIn: indices of the listed tiles that lie inside the lattices' extent (wdom x hdom), in units of 4 -- the
    // denominator of the tile fill that picks the geometry (a tile that hangs over the edge of the domain is not "empty")
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ int inside;
    int acc = 0;
    if (threadIdx.x == 0) { base = prev ? *prev : 0; inside = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int start = 0; start < ntiles; start += 1024) {
        const int t = start + threadIdx.x;
        int f = (t < ntiles) ? (flag[t] != 0) : 0;
        if (f && mode) {
            const int tile = d_virtual_tile(t, tg, k0, nk);
            const int tx = tile % tg.ntx, ty = (tile / tg.ntx) % tg.nty, kk = tile / (tg.ntx * tg.nty);
            const bool boundary = (cm.cut[0] && tx == 0) || (cm.cut[1] && tx == tg.ntx - 1) || (cm.cut[2] && ty == 0) || (cm.cut[3] && ty == tg.nty - 1) ||
                                  (cm.cut[4] && kk == 0) || (cm.cut[5] && kk == nk - 1);
            f = (mode == 2) == boundary;
        }
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wv; q++) woff += wsum[q];
        int total = 0;
        for (int q = 0; q < 16; q++) total += wsum[q];
        if (f) {
            const int tile = d_virtual_tile(t, tg, k0, nk);
            list[base + woff + before] = tile;
            const int tx = tile % tg.ntx, ty = (tile / tg.ntx) % tg.nty;
            const int w = min(tw, wdom - (tg.ox + tx * tw)), h = min(th, hdom - (tg.oy + ty * th));
            if (w > 0 && h > 0) acc += (w * h + 3) >> 2;
        }
        __syncthreads();
        if (threadIdx.x == 0) base += total;
        __syncthreads();
    }
    if (acc) atomicAdd(&inside, acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        *count = base;
        *lanesIn = (mode == 2 ? *lanesIn : 0) + inside;
    }
}

// ------------------------------------------------------------------------------------------------
int fv_scal_reserve(flipv_context *c, int cap) {
    const size_t need = FV_SCAL_BANKS * fv_scal_stride(cap);
    if (c->d_scal && c->h_scal && c->scalCap >= need) return FLIPV_OK;
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    c->d_scal = nullptr; c->h_scal = nullptr; c->scalCap = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_scal, need * sizeof(double)));
    HIPCHK(c, hipHostMalloc((void **)&c->h_scal, need * sizeof(double)));
    c->scalCap = need;
    return FLIPV_OK;
}

// the fill jobs that zero the slot blocks of every bank (and, unless keepExtra, the 16 extra doubles behind bank 0's)
static int scal_clear_jobs(flipv_context *c, int cap, bool keepExtra, FillJob *z) {
    const size_t slots = (size_t)FV_NSC * (cap + 2) * NSLOT, stride = fv_scal_stride(cap);
    if (keepExtra) {
        z[0] = {c->d_scal, slots * sizeof(double), 0};
        z[1] = {c->d_scal + stride, (FV_SCAL_BANKS - 1) * stride * sizeof(double), 0};
        return 2;
    }
    z[0] = {c->d_scal, FV_SCAL_BANKS * stride * sizeof(double), 0};
    return 1;
}
int fv_scal_clear(flipv_context *c, int cap, bool keepExtra) {
    FillJob z[2];
    const int n = scal_clear_jobs(c, cap, keepExtra, z);
    return fv_fill_list(c, z, n);
}

static void scal_views_host(flipv_context *c, int cap, PcgScal *sc, double **extra);
// reset the stall guard (enqueued before any kernel of the solve): best = 0x7f7f... = 1.4e306, "nothing seen yet"
static int guard_jobs(const PcgScal *sc, FillJob *z) {
    z[0] = {sc->best, sizeof(double), 0x7f}; z[1] = {sc->stalled, sizeof(int), 0}; z[2] = {sc->bestIt, 2 * sizeof(int), 0};   // (bestIt and passIt: adjacent ints)
    return 3;
}
void fv_scal_views(flipv_context *c, int cap, PcgScal *sc, double **extra) {
    scal_views_host(c, cap, sc, extra);
    FillJob z[3];
    (void)fv_fill_list(c, z, guard_jobs(sc, z));
}
// What a PCG loop's start needs, as ONE launch: the scalars cleared (fv_scal_clear), the stop flag at -1, the stall guard and the device-side iteration
// counters reset, optionally one more int zeroed (the set-up kernel's row counter); *sc, *extra as fv_scal_views leaves them
int fv_pcg_reset(flipv_context *c, int cap, bool keepExtra, PcgScal *sc, double **extra, int *alsoZero) {
    scal_views_host(c, cap, sc, extra);
    FillJob z[8];
    int n = scal_clear_jobs(c, cap, keepExtra, z);
    z[n++] = {c->d_flags, sizeof(int), 0xff};   // conv = -1
    if (alsoZero) z[n++] = {alsoZero, sizeof(int), 0};
    n += guard_jobs(sc, z + n);
    z[n++] = {sc->itA, 2 * sizeof(int), 0};      // itA, itB
    return fv_fill_list(c, z, n);
}
static void scal_views_host(flipv_context *c, int cap, PcgScal *sc, double **extra) {
    const size_t n = ((size_t)cap + 2) * NSLOT;
    sc->base = c->d_scal;
    sc->conv = c->d_flags;
    const int nr = c->comm ? c->comm->nranks : 1, rk = c->comm ? c->comm->rank : 0;
    sc->nslot = NSLOT / nr > 0 ? NSLOT / nr : 1;   // disjoint slot ranges per rank (nranks <= NSLOT)
    sc->slot0 = rk * sc->nslot;
    sc->nbank = c->comm ? 1 : FV_SCAL_BANKS;   // (a communicator's all-reduce sums bank 0 only)
    sc->bstride = (int)fv_scal_stride(cap);
    sc->cap = cap;
    sc->noB = c->prm.beta_from_conjugacy ? 1 : 0;
    sc->onlyA = 0;
    sc->itA = c->d_flags + 4;
    sc->itB = c->d_flags + 5;
    sc->best = c->d_scal_small + 49;
    sc->stall_below = 0.0;
    sc->stall_ratio = c->prm.stall_guard_ratio > 0.0f ? (double)c->prm.stall_guard_ratio : FV_STALL_RATIO;   // (the viscosity solve of a viscosity FIELD takes FV_STALL_RATIO_FIELD: k_viscosity.hip)
    sc->stalled = c->d_flags + 11;
    sc->bestIt = c->d_flags + 14;
    sc->vel_tol = 0.0; sc->vel_stall = 0.0; sc->vel_window = 0; sc->vel_patience = c->prm.velocity_patience > 0 ? c->prm.velocity_patience : 48; sc->passIt = c->d_flags + 15;   // (the velocity criterion: set by the viscosity solve for its last loop)
    *extra = c->d_scal + FV_NSC * n;
}

static int build_tiles_once(flipv_context *c, const TileGrid &tg, int vw, int nc, const float *d0, const float *d1, const float *d2,
                            const uint8_t *mask, int *list, int *nActive, int *nInterior) {
    const int nk = tg.ntz, nchunks = (tg.nty + JCH - 1) / JCH;
    const int nt = tg.ntx * JCH * nk * nchunks;  // virtual tiles of the owned planes (column-major enumeration)
    GEO_RUN(tg.rowl, hipLaunchKernelGGL(k_tile_flags, dim3(nt), dim3(64, 4, 1), 0, c->stream, tg, c->L, vw, nc, d0, d1, d2, mask, c->tileFlag, 0, nk));
    // The interior / boundary split of the list: the SpMV over the tiles that touch no cut face of the rank's box runs while the
    // halo of the search direction is exchanged.  (With cuts along i a 64- or 256-wide tile column is a large share of the box.)
    const bool split = c->comm != nullptr && c->comm->nranks > 1;
    CutMask cm;
    for (int a = 0; a < 3; a++) { cm.cut[2 * a] = c->comm && c->pcoord[a] > 0; cm.cut[2 * a + 1] = c->comm && c->pcoord[a] < c->pgrid[a] - 1; }
    if (!split) {
        hipLaunchKernelGGL(k_tile_compact, dim3(1), dim3(1024), 0, c->stream, c->tileFlag, nt, list, c->d_flags + 1, tg, 0, nk, 0, cm,
                           (const int *)nullptr, tg.rowl * vw, geo_ty(tg.rowl), c->L.I + 1, c->L.J + 1, c->d_flags + 7);
        FV_READ_JOBS(c, FV_JOB(c->h_flags + 7, c->d_flags + 7, sizeof(int)), FV_JOB(c->h_flags + 1, c->d_flags + 1, sizeof(int)));
        FV_SYNC(c);
        *nActive = *nInterior = c->h_flags[1];
        return FLIPV_OK;
    }
    // list = [interior tiles | tiles at the cut faces]
    hipLaunchKernelGGL(k_tile_compact, dim3(1), dim3(1024), 0, c->stream, c->tileFlag, nt, list, c->d_flags + 6, tg, 0, nk, 1, cm,
                       (const int *)nullptr, tg.rowl * vw, geo_ty(tg.rowl), c->L.I + 1, c->L.J + 1, c->d_flags + 7);
    hipLaunchKernelGGL(k_tile_compact, dim3(1), dim3(1024), 0, c->stream, c->tileFlag, nt, list, c->d_flags + 1, tg, 0, nk, 2, cm,
                       (const int *)(c->d_flags + 6), tg.rowl * vw, geo_ty(tg.rowl), c->L.I + 1, c->L.J + 1, c->d_flags + 7);
    FV_READ_JOBS(c, FV_JOB(c->h_flags + 7, c->d_flags + 7, sizeof(int)), FV_JOB(c->h_flags + 1, c->d_flags + 1, sizeof(int)), FV_JOB(c->h_flags + 6, c->d_flags + 6, sizeof(int)));
    FV_SYNC(c);
    *nActive = c->h_flags[1];
    *nInterior = c->h_flags[6];
    return FLIPV_OK;
}

// the lanes' mask words of the listed tiles, in list order: the PCG kernels then fetch ids and masks side by side
static int gather_masks(flipv_context *c, const TileGrid &tg, int vw, const uint8_t *mask, const int *list, int nActive, unsigned **mlist,
                        size_t *cap) {
    if (!mask || !mlist || nActive <= 0) return FLIPV_OK;
    if ((size_t)nActive > *cap) {
        if (*mlist) { FV_SYNC(c); (void)hipFree(*mlist); *mlist = nullptr; *cap = 0; }
        const size_t want = (size_t)nActive + (size_t)nActive / 4 + 64;
        hipError_t e = hipMalloc((void **)mlist, want * 256 * sizeof(unsigned));
        if (e != hipSuccess) { c->err = std::string("hipMalloc(mask list): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        *cap = want;
    }
    const int nb = nActive < 4096 ? nActive : 4096;
    if (vw == 4) GEO_RUN(tg.rowl, hipLaunchKernelGGL(k_mask_gather<4>, dim3(nb), dim3(64, 4, 1), 0, c->stream, list, nActive, tg, c->L, mask, *mlist));
    else GEO_RUN(tg.rowl, hipLaunchKernelGGL(k_mask_gather<2>, dim3(nb), dim3(64, 4, 1), 0, c->stream, list, nActive, tg, c->L, mask, *mlist));
    return FLIPV_OK;
}

int fv_build_tiles(flipv_context *c, TileGrid *tg, int vw, int nc, const float *d0, const float *d1, const float *d2,
                   const uint8_t *mask, int *list, int *nActive, int *nInterior, const int *hostCount, int perIndex, unsigned **mlist, size_t *mlistCap, double minLanes, int *memo) {
    const int forceRowl = (c->prm.tile_rows == 16 || c->prm.tile_rows == 64) ? c->prm.tile_rows : 0;
    int rowl = forceRowl ? forceRowl : tg->rowl;
    *tg = make_tile_grid(c->L, rowl, vw);
    int rc = build_tiles_once(c, *tg, vw, nc, d0, d1, d2, mask, list, nActive, nInterior);
    if (rc) return rc;
    // how full the tiles are: indices with unknowns / indices of the listed tiles inside the lattices' extent
    auto fill_now = [&]() { return c->h_flags[7] > 0 ? ((double)*hostCount / perIndex) / (4.0 * (double)c->h_flags[7]) : 1.0; };
    // ... and how much of the listed tiles lies inside the extent at all: a 256-wide row on a 385-wide lattice idles a quarter of its lanes
    // (filled boxes, profiles/r4/dense_geometry_scan.log: the viscosity SpMV at 320^3 / 384^3 / 448^3 -- 0.63 / 0.75 / 0.875 of the wide rows' lanes inside --
    // runs at 0.36 / 0.48 / 0.51 of the HBM peak in 256-wide rows and at 0.57 / 0.53 / 0.55 in 64-wide ones, the pressure SpMV at 0.46 / 0.52 / 0.52 and
    // 0.60 / 0.49 / 0.49; at 512^3 wide rows win both.  minLanes = the share below which the caller's kernel is better off in narrow rows.)
    auto lanes_now = [&]() { return *nActive > 0 ? 4.0 * (double)c->h_flags[7] / ((double)*nActive * 256.0 * vw) : 1.0; };
    c->tileFill = fill_now();
    if (!forceRowl && *nActive > 0) {
        // *memo remembers a geometry tried and turned down, so that a steady scene does not pay for the trial every solve: n > 0 = narrow rows were no better
        // when the wide list had n tiles, n < 0 = wide rows idled too many lanes when the narrow list had -n
        auto same = [&](int then) { return then > 0 && abs(*nActive - then) <= then / 10; };
        auto rebuild = [&](int r) { *tg = make_tile_grid(c->L, r, vw); return build_tiles_once(c, *tg, vw, nc, d0, d1, d2, mask, list, nActive, nInterior); };
        const double fill = c->tileFill, lanes = lanes_now();
        if (rowl == 64 && fill < 0.45) { if ((rc = rebuild(16))) return rc; *memo = 0; }
        else if (rowl == 64 && lanes < minLanes && !same(*memo)) {
            const int n64 = *nActive;
            if ((rc = rebuild(16))) return rc;
            if (lanes_now() < 1.1 * lanes) { if ((rc = rebuild(64))) return rc; *memo = n64; } else *memo = 0;
        } else if (rowl == 16 && fill > 0.65 && !same(-*memo)) {
            const int n16 = *nActive;
            if ((rc = rebuild(64))) return rc;
            if (lanes_now() < minLanes && lanes >= 1.1 * lanes_now()) { if ((rc = rebuild(16))) return rc; *memo = -n16; } else *memo = 0;
        }
        c->tileFill = fill_now();
    }
    return gather_masks(c, *tg, vw, mask, list, *nActive, mlist, mlistCap);
}

// dots: 0 = only a = s.q (multigrid-preconditioned loop, benchmark launches), 1 = a and c, 2 = a, b and c (reads r)
int fv_build_runs(flipv_context *c, const TileGrid &tg, int vw, int nActive, bool dense, const uint8_t *mask, Run **runs, size_t *runCap, int *nruns,
                  int *runLen, unsigned **rmask, size_t *rmaskCap) {
    *nruns = 0;
    if (c->comm || vw != 4 || nActive <= 0 || c->prm.spmv_run_length < 0) return FLIPV_OK;
    // Measured (MI355X, 256^3): on a filled box the marching kernels are 10-20 % faster than the tile-at-a-time ones and the
    // longer the run the better (8: 207 us, 32: 199 us per viscosity SpMV; tiles: 226-244 us); on the bunny scene (4 % of the
    // box liquid, 1 600 tiles) they lose, the more the longer the runs (tiles 17.8 us; runs of 2-4: 20.5-21 us, 16: 50 us):
    // there a launch is bound by how many blocks have work, and a run serialises its tiles.
    int runlen = c->prm.spmv_run_length;
    if (runlen == 0) { if (!dense) return FLIPV_OK; runlen = RUNLEN_MAX; }
    if (runlen < 2) runlen = 2;
    if (runlen > RUNLEN_MAX) runlen = RUNLEN_MAX;
    const int nk = tg.ntz;
    const int nchunk = (nk + runlen - 1) / runlen;
    const size_t ncand = (size_t)tg.ntx * tg.nty * nchunk;
    auto grow = [&](void **p, size_t *cap, size_t want, size_t elem) -> int {
        if (want <= *cap) return FLIPV_OK;
        if (*p) { FV_SYNC(c); (void)hipFree(*p); *p = nullptr; *cap = 0; }
        const size_t n = want + want / 4 + 64;
        hipError_t e = hipMalloc(p, n * elem);
        if (e != hipSuccess) { c->err = std::string("hipMalloc(run list): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        *cap = n;
        return FLIPV_OK;
    };
    int rc;
    if ((rc = grow((void **)&c->runCand, &c->runCandCap, ncand, sizeof(Run)))) return rc;
    if ((rc = grow((void **)runs, runCap, ncand, sizeof(Run)))) return rc;
    hipLaunchKernelGGL(k_run_flags, dim3(cdiv(ncand, 256)), dim3(256), 0, c->stream, tg, (const int *)c->tileFlag, 0, nk, JCH, runlen, nchunk, c->runCand);
    hipLaunchKernelGGL(k_run_compact, dim3(1), dim3(1024), 0, c->stream, (const Run *)c->runCand, (int)ncand, *runs, c->d_flags + 12);
    FV_READ(c, c->h_flags + 12, c->d_flags + 12, 2 * sizeof(int));
    FV_SYNC(c);
    const int n = c->h_flags[12];
    if (n <= 0) return FLIPV_OK;
    if (mask) {
        if ((rc = grow((void **)rmask, rmaskCap, (size_t)n * 256, sizeof(unsigned)))) return rc;
        const int nb = n < 4096 ? n : 4096;
        GEO_RUN(tg.rowl, hipLaunchKernelGGL(k_run_masks<4>, dim3(nb), dim3(64, 4, 1), 0, c->stream, (const Run *)*runs, n, tg, c->L, mask, *rmask));
    }
    *nruns = n;
    *runLen = runlen;
    return FLIPV_OK;
}

template <typename T>
static void launch_pressure_spmv(flipv_context *c, const PcgScal &sc, int it, int first, int count, int dots) {
    const int nb = pcg_grid(c, count);
    const bool timed = c->prm.kernel_timing && (it & 7) == 0 && first == 0;  // HIP events around every 8th launch
    if (timed) fv_ev_begin(c, 0, (double)count * (256 * VW_P));
#define PSPMV(D) GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL((k_pressure_spmv<T, D>), dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListP + first, count, c->tgP, c->L, \
                       c->pDiag, c->pPi, c->pPj, c->pPk, c->pMask, c->mlistP ? c->mlistP + (size_t)first * 256 : (const unsigned *)nullptr, (const T *)c->pS, (const RT<T> *)c->pR, (T *)c->pZ, sc, it))
#define PMARCH(D, S) GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL((k_pressure_spmv_march<T, D, S>), dim3(nbm), dim3(64, 4, 1), 0, c->stream, (const Run *)c->runsP, c->nRunsP, (const unsigned *)c->rmaskP, c->tgP, c->L, \
                       c->pDiag, c->pPi, c->pPj, c->pPk, (const T *)c->pS, (const RT<T> *)c->pR, (T *)c->pZ, sc, it))
    // streaming accesses once the system's 25 bytes per cell exceed the memory-side cache (ldvs, pcg_common.h)
    const bool stream = (double)c->nActiveP * (256 * VW_P) * 25.0 > 256.0 * 1024 * 1024;
    // filled boxes: the sweep in address order (k_pressure_spmv_sweep; spmv_run_length = -2 forces it on any single-rank box: tests)
    const bool sweep = !c->comm && std::is_same<T, float>::value && first == 0 && count == c->nActiveP &&
                       (c->prm.spmv_run_length == -2 || (c->prm.spmv_run_length == 0 && c->nRunsP > 0 && c->tileFillP > 0.65));
    if (sweep || (c->nRunsP > 0 && first == 0 && count == c->nActiveP)) {   // the whole system: sweep, or k-marching over the run list
        const int nbm = pcg_grid(c, c->nRunsP > 0 ? c->nRunsP : 8);
        if (sweep) {
            const Lay R = fv_range_liquid(c, 1, 5);
            const int kb = R.kb > c->tgP.oz ? R.kb : c->tgP.oz, ke0 = R.ke < c->tgP.oz + c->tgP.ntz ? R.ke : c->tgP.oz + c->tgP.ntz, ke = ke0 < c->L.K ? ke0 : c->L.K;   // planes that hold cells
            // ONE BLOCK PER UNIT: the hardware starts a block the moment another retires, so the loads of a fresh unit overlap the stores of its predecessors; a resident
            // grid walking units b, b + G, ... serialises each wave's load -> compute -> store chain (512^3: 0.51-0.59 of peak with G = 1 024 ... 8 192 against 0.66-0.70;
            // profiles/r5/spmv7_variants_512.log).  The price is one fp64 atomic per block and scalar -- 65 536 fire-and-forget atomics onto 128 slot words at 512^3,
            // hidden in the 600 us of the launch.  8 x (the largest XCD slab's units): the kernel's own XCD mapping.
            // lanes of a row: 64 (tiles of 256 i) unless those fit the extent badly and 128-wide ones better -- 384 = 1.5 x 256 leaves a quarter of the lanes idle:
            // 0.55 of peak in 64-lane rows, 0.62 in 32-lane rows; at 448 and 512 the 64-lane rows win (0.64 / 0.67 against 0.52 / 0.62), 16-lane rows lose everywhere
            // (profiles/r5/pressure_sweep_probe.log).  tile_rows = 64 pins the wide rows.
            const int ext = c->L.I - c->tgP.ox;
            const double use64 = (double)ext / (((ext + 255) / 256) * 256.0), use32 = (double)ext / (((ext + 127) / 128) * 128.0);
            const int rl = (c->prm.tile_rows != 64 && use64 < 0.8 && use32 > 1.1 * use64) ? 32 : 64;
            const int tw = 4 * rl, rpb = 8 * (64 / rl);
            const int ntx = (ext + tw - 1) / tw, nrow2 = (c->L.J - c->tgP.oy + rpb - 1) / rpb;
            const int g = ke > kb ? 8 * ntx * ((nrow2 + 7) / 8) * (ke - kb) : 8;
#define PSWEEP(D, R) hipLaunchKernelGGL((k_pressure_spmv_sweep<T, D, R>), dim3(g), dim3(64, 4, 1), 0, c->stream, c->tgP, c->L, kb, ke > kb ? ke - kb : 0, c->pDiag, c->pPi, c->pPj, c->pPk, \
                                        (const T *)c->pS, (const RT<T> *)c->pR, (T *)c->pZ, sc, it)
            if (rl == 64) { if (dots == 2) PSWEEP(2, 64); else if (dots == 1) PSWEEP(1, 64); else PSWEEP(0, 64); }
            else { if (dots == 2) PSWEEP(2, 32); else if (dots == 1) PSWEEP(1, 32); else PSWEEP(0, 32); }
#undef PSWEEP
        } else if (stream) { if (dots == 2) PMARCH(2, true); else if (dots == 1) PMARCH(1, true); else PMARCH(0, true); }
        else if (dots == 2) PMARCH(2, false); else if (dots == 1) PMARCH(1, false); else PMARCH(0, false);
    } else if (dots == 2) PSPMV(2); else if (dots == 1) PSPMV(1); else PSPMV(0);
#undef PSPMV
#undef PMARCH
    if (timed) fv_ev_end(c);
}

int fv_pressure_pcg_mg(flipv_context *c, const PcgScal &sc, int cap, void (*spmv)(flipv_context *, const PcgScal &, int), int *conv_out);

template <typename T>
static int pressure_solve_t(flipv_context *c, float dt, flipv_solve_info *info) {
    const Lay &L = c->L;
    flipv_solve_info li;
    memset(&li, 0, sizeof(li));
    c->commBytesSetup = c->commBytesIter = 0.0;
    c->exchIter = c->allrIter = 0;
    const int cap = c->prm.pressure_max_iterations;
    int rc = fv_scal_reserve(c, cap);
    if (rc) return rc;
    PcgScal sc;
    double *bmax;
    if ((rc = fv_pcg_reset(c, cap, false, &sc, &bmax, c->d_flags + 2))) return rc;   // (scalars, conv = -1, the pressure-cell counter, the guard, the counters: one launch)
    sc.tol_inclusive = 0;
    sc.tol = 0.0;

    // with fp32 vectors x IS the pressure grid
    constexpr bool f32 = std::is_same<T, float>::value;
    T *x = f32 ? (T *)c->pressure : (T *)c->pX;
    const Lay R1 = fv_range_liquid(c, 1, 5);  // one halo plane: s is zeroed there, the coefficients towards it are the neighbour's business
    hipLaunchKernelGGL(k_pressure_setup<T>, GRID3(R1), 0, c->stream, R1, c->phi, c->U, c->V, c->W, c->wU, c->wV, c->wW,
                       c->pDiag, c->pPi, c->pPj, c->pPk, (RT<T> *)c->pR, x, (T *)c->pS, c->pMask, bmax, c->d_flags + 2, c->dx, dt, c->prm.min_frac);
    FV_READ_JOBS(c, FV_JOB(c->h_flags + 2, c->d_flags + 2, sizeof(int)), FV_JOB(c->h_scal, bmax, sizeof(double)));
    rc = fv_build_tiles(c, &c->tgP, VW_P, 1, c->pDiag, nullptr, nullptr, c->pMask, c->tileListP, &c->nActiveP, &c->nIntP, c->h_flags + 2, 1, &c->mlistP, &c->mlistCapP, 0.70, &c->geoMemoP);  // synchronises: h_scal[0] = max|b|
    if (rc) return rc;
    c->tileFillP = c->tileFill;
    if ((rc = fv_build_runs(c, c->tgP, VW_P, c->nActiveP, c->tgP.rowl == 64 || c->tileFill > 0.65, c->pMask, &c->runsP, &c->runCapP, &c->nRunsP, &c->runLenP, &c->rmaskP, &c->rmaskCapP))) return rc;
    {
        float bn = (float)c->h_scal[0];   // global max|b| (fp32 is enough for a tolerance scale)
        double bd = c->h_scal[0];
        if (c->comm) { rc = fv_allreduce_max_f32(c, &bn); if (rc) return rc; bd = (double)bn; }
        c->h_scal[0] = bd;
    }
    const double bnorm = c->h_scal[0];
    li.rhs_norm = bnorm;
    li.active_tiles = c->nActiveP;
    li.total_tiles = c->tgP.count();
    li.rows = c->h_flags[2];  // pressure cells of this rank
    c->pressureReady = 1;
    c->pressurePrec = f32 ? 0 : 1;
    c->lastDt = dt;
    // early out (pressuresolver.cpp:173-175): pressure grid is zero
    int anyActive = c->nActiveP;
    if (c->comm) { float f = (float)anyActive; rc = fv_allreduce_max_f32(c, &f); if (rc) return rc; anyActive = (int)f; }
    if (!(bnorm >= c->prm.pressure_tolerance) || anyActive == 0) {
        li.status = 3;
        li.residual = bnorm;
        if (!f32) hipLaunchKernelGGL(k_copy_to_f32<T>, dim3(2048), dim3(256), 0, c->stream, (const T *)x, c->pressure, L.n);
        if (info) *info = li;
        return FLIPV_OK;
    }
    sc.tol = fmax(c->prm.pressure_tolerance, c->prm.pressure_rel_tolerance * bnorm);

    int conv = -1;
    // preconditioner: aggregation multigrid with fp32 vectors (k_pressure_mg.hip; rank-local in a slab-decomposed run), the
    // diagonal otherwise (fp64 vectors, tiny grids, or flipv_params.pressure_preconditioner = DIAGONAL)
    const bool useMg = f32 && c->prm.pressure_preconditioner != FLIPV_PRECOND_DIAGONAL && (L.I > 16 || L.J > 16 || L.K > 16);  // at least two levels below the tile-list level
    li.preconditioner = useMg ? 1 : 0;
    if (useMg) {
        if ((rc = fv_pressure_pcg_mg(c, sc, cap, [](flipv_context *cc, const PcgScal &s, int it) { launch_pressure_spmv<float>(cc, s, it, 0, cc->nActiveP, 0); },
                                     &conv)))
            return rc;
    } else {
        PcgSys<T, 1> v;
        v.swz = 0;
        v.mask = c->pMask;
        v.mlist = c->mlistP;
        v.diag[0] = c->pDiag; v.x[0] = x; v.r[0] = (RT<T> *)c->pR; v.q[0] = (T *)c->pZ; v.s[0] = (T *)c->pS;
        const int nb = pcg_grid(c, c->nActiveP);
        const dim3 blk(64, 4, 1);
        const HaloArray sh[1] = {{c->pS, sizeof(T)}};
        GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL((k_pcg_init<T, 1, VW_P>), dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, L, v, sc));
        if ((rc = fv_allreduce_scalars(c, sc.sig(0), NSLOT))) return rc;
        auto spmv = [&](int first, int count, int it) { launch_pressure_spmv<T>(c, sc, it, first, count, sc.noB ? 1 : 2); };
        auto update = [&](int it) {
            GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL((k_pcg_update<T, 1, VW_P>), dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, L, v, sc, it));
        };
        if ((rc = pcg_run(c, sc, cap, sh, 1, c->nIntP, c->nActiveP, spmv, update, &conv, FV_GE_PRESSURE))) return rc;
    }
    const int last = conv >= 0 ? conv : cap - 1;
    hipLaunchKernelGGL(k_pcg_residual, dim3(1), dim3(64), 0, c->stream, sc, last, bmax);
    FV_READ(c, c->h_scal, bmax, sizeof(double));
    if (!f32) hipLaunchKernelGGL(k_copy_to_f32<T>, dim3(2048), dim3(256), 0, c->stream, (const T *)x, c->pressure, L.n);
    FV_SYNC(c);
    li.iterations = conv >= 0 ? conv + 1 : cap;
    li.residual = c->h_scal[0];
    li.status = conv >= 0 ? 0 : 1;
    if (conv >= 0 && !useMg) {   // the diagonal loop's stall guard stops through the same flag: that is not convergence
        int st = 0;
        HIPCHK(c, hipMemcpy(&st, sc.stalled, sizeof(int), hipMemcpyDeviceToHost));
        if (st) { li.status = 1; conv = -1; }
    }
    if (c->prm.kernel_timing) fv_ev_collect(c);
    li.comm_bytes_setup = c->commBytesSetup; li.comm_bytes_per_iteration = c->commBytesIter;
    li.halo_exchanges_per_iteration = c->exchIter; li.allreduces_per_iteration = c->allrIter;
    if (info) *info = li;
    {
        const HaloArray ph[1] = {{c->pressure, 4}};  // the gradient at plane k0 reads p(k0-1)
        if ((rc = fv_halo_copy(c, ph, 1, 1))) return rc;
    }
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
    PcgScal sc;
    memset(&sc, 0, sizeof(sc));  // no scalars, no stop flag: pure kernel launches
    // the variant the solve runs: the multigrid-preconditioned loop needs s.q only, the diagonal one also (q/d).q [and (r/d).q]
    const bool mgLoop = !c->pressurePrec && c->prm.pressure_preconditioner != FLIPV_PRECOND_DIAGONAL && (c->L.I > 16 || c->L.J > 16 || c->L.K > 16);
    const int bdots = mgLoop ? 0 : (c->prm.beta_from_conjugacy ? 1 : 2);
    for (int w = 0; w < 3; w++) {
        if (c->pressurePrec) launch_pressure_spmv<double>(c, sc, 0, 0, c->nActiveP, bdots); else launch_pressure_spmv<float>(c, sc, 0, 0, c->nActiveP, bdots);
    }
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; r++) {
        if (c->pressurePrec) launch_pressure_spmv<double>(c, sc, 0, 0, c->nActiveP, bdots); else launch_pressure_spmv<float>(c, sc, 0, 0, c->nActiveP, bdots);
    }
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipEventSynchronize(b));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, a, b));
    c->prm.kernel_timing = saved;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms = (double)t / reps;
    *cells = (double)c->nActiveP * (256 * VW_P);
    return FLIPV_OK;
}
