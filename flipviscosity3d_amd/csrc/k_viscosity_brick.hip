// k_viscosity_brick.hip -- the viscosity PCG on the BRICK layout (flipv_internal.h: bidx): one wave per brick of 8 x 4 x 2 indices, one
// lane per index.
//
// Why a second set of kernels.  On the reference's scenes the liquid is a compact body filling a few per cent of the box (256^3 bunny:
// 4 %).  The tile kernels (k_viscosity_geo.inc: 64 x 16 x 1 tiles, a lane owns 4 consecutive i of a plain k-plane) move 1.9x the
// algorithmic bytes there -- 128-byte lines that are 32 x 1 sticks are half used at both ends of every i-run of the liquid -- and a
// block walks its 3-4 tiles one dependent round trip after the other at 2 waves per SIMD (189 VGPRs): 20 us per SpMV of which the bytes
// explain 8.  Here every array the solve touches is stored in bricks, so that
//   * a 128-byte line is a 4 x 4 x 2 block of space (1.1x over-fetch on that scene instead of 1.5x),
//   * a wave's own-index access is one contiguous 256-byte run whatever the array,
//   * the 15-point coupled stencil is addressed with six per-lane constants (the layout is separable: NbOff), every neighbour a plain
//     load that hits the lines the wave's own brick and its six face neighbours occupy,
//   * a lane holds one index: ~70 VGPRs, 6-7 waves per SIMD -- the latency of a brick's loads is hidden by the other bricks in
//     flight, not by a deeper pipeline inside one wave.
// Same arithmetic as the tile kernels: the rows come from d_visc_rows (visc_rows.h) with NV = 1.
// Used when the liquid is sparse (row fill <= 0.35), on single-domain contexts; filled boxes keep the k-marching tile kernels, which
// stream whole planes, block contexts (multi-GPU) keep the plain layout their halo exchange packs from.
#include "flipv_internal.h"
#include "pcg_common.h"
#include "visc_rows.h"
#include "brick.h"

// ------------------------------------------------------------------ active bricks
// flag per brick of the box's brick range and the number of flagged bricks per chunk of 1024: one workgroup per chunk, one thread per brick (its 64
// mask bytes as four 16-byte loads; consecutive codes of a brick row are consecutive in memory).  (One wave per brick with an atomic per flagged brick on
// the few dozen chunk counters took 110 us at 256^3.)
// Cut: which bricks are INTERIOR to a rank's owned box -- none of their 8 x 4 x 2 indices lies in the first / last owned entry along an axis that has a neighbour rank on that
// side, so that no row of theirs reads a halo entry.  want: 0 every active brick, 1 the interior ones, 2 the others (the bricks along the cut faces).
struct BrickCut { int lo[3], hi[3]; };   // an index p is "inner" along axis a iff lo[a] <= p < hi[a]  (lo = olo + 1 with a lower neighbour, else -inf; hi = ohi - 1 with an upper one)
__global__ __launch_bounds__(1024) void k_brick_flags(BrickBox R, Lay LB, const uint8_t *__restrict__ maskB, int *__restrict__ flag, int *__restrict__ chunkCount, int n, BrickCut cut, int want) {
    __shared__ int wcount[16];
    const int code = (int)blockIdx.x * 1024 + (int)threadIdx.x;
    bool any = false;
    if (code < n) {
        const int brick = d_brick_of_code(R, LB, code);
        const uint4 *m = reinterpret_cast<const uint4 *>(maskB + ((size_t)brick << 6));
        const uint4 a = m[0], b = m[1], c = m[2], d = m[3];
        any = ((a.x | a.y | a.z | a.w) | (b.x | b.y | b.z | b.w) | (c.x | c.y | c.z | c.w) | (d.x | d.y | d.z | d.w)) != 0u;
        if (any && want) {
            int i0, j0, k0;
            d_brick_ijk(LB, brick, 0, i0, j0, k0);
            const bool inner = i0 >= cut.lo[0] && i0 + 8 <= cut.hi[0] && j0 >= cut.lo[1] && j0 + 4 <= cut.hi[1] && k0 >= cut.lo[2] && k0 + 2 <= cut.hi[2];
            any = inner == (want == 1);
        }
        flag[code] = any ? 1 : 0;
    }
    const unsigned long long bal = __ballot(any);
    if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) t += wcount[w];
        chunkCount[blockIdx.x] = t;
    }
}
// exclusive scan of the chunk counts by one workgroup; total into *count
__global__ __launch_bounds__(1024) void k_brick_scan(int *__restrict__ chunkCount, int nchunks, int *__restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int start = 0; start < nchunks; start += 1024) {
        const int t = start + (int)threadIdx.x;
        const int v = t < nchunks ? chunkCount[t] : 0;
        int inc = v;   // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int woff = 0, total = 0;
        for (int q = 0; q < 16; q++) { if (q < wv) woff += wsum[q]; total += wsum[q]; }
        if (t < nchunks) chunkCount[t] = base + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}
// ordered scatter: chunk `blockIdx.x` writes its flagged bricks behind its offset
__global__ __launch_bounds__(1024) void k_brick_scatter(BrickBox R, Lay LB, const int *__restrict__ flag, const int *__restrict__ chunkOff, int n, int *__restrict__ list) {
    __shared__ int wsum[16];
    const int code = (int)blockIdx.x * 1024 + (int)threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f = code < n ? flag[code] != 0 : 0;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wv; q++) woff += wsum[q];
    if (f) list[chunkOff[blockIdx.x] + woff + __popcll(m & ((1ull << lane) - 1ull))] = d_brick_of_code(R, LB, code);
}

int fv_build_bricks(flipv_context *c, const Lay &box) {
    const Lay &LB = c->LB;
    const BrickBox R = brick_box(box);
    const int n = R.nb[0] * R.nb[1] * R.nb[2];
    const int nchunks = (n + 1023) / 1024;
    int *chunk = c->brickFlag + c->brickCap - nchunks - 1;   // the tail of the flag array: the box's bricks never fill it (padding bricks are never in a box)
    // Several ranks: the list is [interior bricks | bricks along the cut faces] -- the SpMV and the multigrid's fine-level sweeps run over the first part while the halo of
    // their input travels on the communication stream, over the second part once it has arrived (fv_viscosity_pcg_mg, vmg_vcycle; c->nIntV = the split).
    const bool split = c->comm && c->comm->nranks > 1 && !c->prm.no_comm_overlap;
    BrickCut cut;
    for (int a = 0; a < 3; a++) {
        const bool lower = split && c->pcoord[a] > 0, upper = split && c->pcoord[a] + 1 < c->pgrid[a];
        cut.lo[a] = lower ? c->L.olo[a] + 1 : -(1 << 30);
        cut.hi[a] = upper ? c->L.ohi[a] - 1 : (1 << 30);
    }
    int total = 0;
    c->nIntV = 0;
    for (int pass = split ? 1 : 0; pass <= (split ? 2 : 0); pass++) {
        hipLaunchKernelGGL(k_brick_flags, dim3(nchunks), dim3(1024), 0, c->stream, R, LB, (const uint8_t *)c->vMaskB, c->brickFlag, chunk, n, cut, pass);
        hipLaunchKernelGGL(k_brick_scan, dim3(1), dim3(1024), 0, c->stream, chunk, nchunks, c->d_flags + 1);
        hipLaunchKernelGGL(k_brick_scatter, dim3(nchunks), dim3(1024), 0, c->stream, R, LB, (const int *)c->brickFlag, (const int *)chunk, n, c->brickList + total);
        FV_READ(c, c->h_flags + 1, c->d_flags + 1, sizeof(int));
        FV_SYNC(c);
        total += c->h_flags[1];
        if (pass == 1) c->nIntV = total;
    }
    c->nBricks = total;
    if (!split) c->nIntV = c->comm ? 0 : total;
    return FLIPV_OK;
}

// ------------------------------------------------------------------ K9 SpMV / multigrid fine-level sweeps
// q = A s (EPI_SPMV, with the fused dot products) or one of the multigrid preconditioner's fine-level sweeps (EPI_*: v.s = input,
// v.q = output, v.r = the residual), exactly like k_visc_spmv (k_viscosity_geo.inc).  vm* = the rows' own volumes (exact operator) or
// own volumes + the reference's diagonal defect (d_ref_volume); a lane has rows where its mask byte says so.
template <typename T, bool RDOT, int EPI>
__global__ __launch_bounds__(256) void k_bvisc_spmv(const int *__restrict__ bricks, int nb, const float *__restrict__ vmU, const float *__restrict__ vmV,
                                                    const float *__restrict__ vmW, const float *__restrict__ fC, const float *__restrict__ fEU,
                                                    const float *__restrict__ fEV, const float *__restrict__ fEW, BrickSys<T> v, PcgScal sc, int it_arg,
                                                    float omega, int sig_shift) {
#pragma clang fp contract(fast)
    __shared__ double lds[12];
    BrickWalk w;
    w.begin(bricks, nb, v.mask);
    bool stop;
    const int it = d_iter_spmv(sc, it_arg, stop);
    if (stop) return;
    const NbOff o = d_lane_off(v.sby, v.sbz);
    const T *__restrict__ xu = v.s[0], *__restrict__ xv = v.s[1], *__restrict__ xw = v.s[2];
    double da = 0.0, db = 0.0, dc = 0.0;
    const bool faceL = ((int)threadIdx.x & 35) == 0, faceR = ((int)threadIdx.x & 35) == 35;   // the brick's two x faces: x = 0 (lower half, li = 0), x = 7 (upper half, li = 3)
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, v.mask);
        if (!__any(m != 0u)) continue;   // (wave-uniform: the lane shuffles below need every lane of a brick that holds rows)
        // Every load first.  The ten values an x-neighbour may need come from EVERY lane of the brick (the same wave instructions and cache lines as before) ...
        const float C0 = fC[a], EW0 = fEW[a], EV0 = fEV[a];
        const T U0 = xu[a], V0 = xv[a], W0 = xw[a];
        const T Ujm = xu[a + o.ym], Ukm = xu[a + o.zm], Vjp = xv[a + o.yp], Wkp = xw[a + o.zp];
        // ... what lies across an x face of the brick from the 8 + 8 lanes of the two faces, a left and a right quantity per instruction ...
        float e1 = 0.0f, e2 = 0.0f;
        T e3 = (T)0, e4 = (T)0, e5 = (T)0, e6 = (T)0, e7 = (T)0;
        if (faceL || faceR) {
            const int ox = faceL ? o.xm : o.xp;
            e1 = (faceL ? fC : fEW)[a + ox];                                   // C0l | EW0r
            e2 = fEV[a + ox];                                                  //     | EV0r
            e3 = xu[a + ox]; e4 = xv[a + ox]; e5 = xw[a + ox];                 // U0l V0l W0l | U0r V0r W0r
            e6 = faceL ? xv[a + o.yp + ox] : xu[a + o.ym + ox];                // Vjpl | Ujmr
            e7 = faceL ? xw[a + o.zp + ox] : xu[a + o.zm + ox];                // Wkpl | Ukmr
        }
        // ... and the other 23 from the lanes that hold rows only (no initialisers: every read below sits under the same condition, so nothing has to be defined for the
        // other lanes -- zero-initialised values cost 60 moves per lane, loading whole bricks costs 10 % more traffic: 1.36 x against 1.49 x the algorithmic bytes).
        const bool rows = m != 0u;
        float Cjm, Ckm, EWjp, EVkp, EU0, EUjp, EUkp, MU, MV, MW;
        T Ujp, Ukp, Vjm, Vkm, Vkp, Vjpkm, Wjm, Wjp, Wkm, Wjmkp;
        RT<T> RU = (RT<T>)0, RV = (RT<T>)0, RW = (RT<T>)0;
        if (rows) {
            Cjm = fC[a + o.ym]; Ckm = fC[a + o.zm];
            EWjp = fEW[a + o.yp]; EVkp = fEV[a + o.zp];
            EU0 = fEU[a]; EUjp = fEU[a + o.yp]; EUkp = fEU[a + o.zp];
            Ujp = xu[a + o.yp]; Ukp = xu[a + o.zp];
            Vjm = xv[a + o.ym]; Vkm = xv[a + o.zm]; Vkp = xv[a + o.zp]; Vjpkm = xv[a + o.yp + o.zm];
            Wjm = xw[a + o.ym]; Wjp = xw[a + o.yp]; Wkm = xw[a + o.zm]; Wjmkp = xw[a + o.ym + o.zp];
            MU = (m & 1u) ? vmU[a] : -1.0f; MV = (m & 2u) ? vmV[a] : -1.0f; MW = (m & 4u) ? vmW[a] : -1.0f;
            if (RDOT) { RU = v.r[0][a]; RV = v.r[1][a]; RW = v.r[2][a]; }
        }
        // x - 1 / x + 1 by lane shuffles inside the brick (brick.h: d_xm, d_xp); every lane takes part
        const float C0l = d_xm(C0, e1), EW0r = d_xp(EW0, e1), EV0r = d_xp(EV0, e2);
        T U0l, U0r, V0l, V0r, W0l, W0r;
        d_xnb(U0, e3, e3, U0l, U0r); d_xnb(V0, e4, e4, V0l, V0r); d_xnb(W0, e5, e5, W0l, W0r);
        const T Vjpl = d_xm(Vjp, e6), Ujmr = d_xp(Ujm, e6), Wkpl = d_xm(Wkp, e7), Ukmr = d_xp(Ukm, e7);
        if (rows) {
            Vec<T, 1> yU, yV, yW;
            T ta = (T)0, tb = (T)0, tc = (T)0;
#define V1F(x_) Vec<float, 1>{{x_}}
#define V1T(x_) Vec<T, 1>{{x_}}
#define V1R(x_) Vec<RT<T>, 1>{{x_}}
            d_visc_rows<T, 1, RDOT, EPI>(V1F(MU), V1F(MV), V1F(MW), V1F(C0), V1F(Cjm), V1F(Ckm), V1F(EW0), V1F(EWjp), V1F(EV0), V1F(EVkp), V1F(EU0), V1F(EUjp),
                                         V1F(EUkp), V1T(U0), V1T(Ujm), V1T(Ujp), V1T(Ukm), V1T(Ukp), V1T(V0), V1T(Vjm), V1T(Vjp), V1T(Vkm), V1T(Vkp), V1T(W0),
                                         V1T(Wjm), V1T(Wjp), V1T(Wkm), V1T(Wkp), V1T(Vjpkm), V1T(Wjmkp), V1R(RU), V1R(RV), V1R(RW), C0l, EW0r, EV0r, U0l, U0r, V0l,
                                         V0r, W0l, W0r, Vjpl, Wkpl, Ujmr, Ukmr, yU, yV, yW, ta, tb, tc, (T)omega);
#undef V1F
#undef V1T
#undef V1R
            da += (double)ta; dc += (double)tc;
            db += (double)tb;
            if (m & 1u) v.q[0][a] = yU.v[0];
            if (m & 2u) v.q[1][a] = yV.v[0];
            if (m & 4u) v.q[2][a] = yW.v[0];
        }
    }
    if (EPI == EPI_JACOBI || EPI == EPI_RESIDUAL) return;
    if (EPI == EPI_JACOBI_DOT || EPI == EPI_SPMV_A) {   // (r, z) into sig(it + sig_shift) | p.q into a(it)
        const double rz = block_sum_256(da, lds);
        if (threadIdx.x == 0 && threadIdx.y == 0 && sc.conv) {
            const int sl = sc.my_slot();
            if (rz != 0.0) atomicAdd((EPI == EPI_SPMV_A ? sc.a(it) : sc.sig(it + sig_shift)) + sl, rz);
        }
        return;
    }
    block_sum3_256(da, db, dc, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && sc.conv) {
        const int sl = sc.my_slot();
        if (da != 0.0) atomicAdd(sc.a(it) + sl, da);
        if (db != 0.0) atomicAdd(sc.b(it) + sl, db);
        if (dc != 0.0) atomicAdd(sc.c(it) + sl, dc);
    }
}

// ------------------------------------------------------------------ z = r/d, s = z, sigma(0) = (r, z)
template <typename T>
__global__ __launch_bounds__(256) void k_bpcg_init(const int *__restrict__ bricks, int nb, BrickSys<T> v, PcgScal sc) {
    __shared__ double lds[4];
    BrickWalkV w;
    w.begin(bricks, nb, v.mask);
    double acc = 0.0;
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, v.mask);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!BrickWalkV::any(m, c)) continue;
            const Vec<float, 4> d = ldv<4>(v.diag[c] + a);
            const Vec<RT<T>, 4> r = ldv<4>(v.r[c] + a);
            Vec<T, 4> s;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double zd = (BrickWalkV::row(m, c, e) && d.v[e] != 0.0f) ? (double)r.v[e] / (double)d.v[e] : 0.0;
                s.v[e] = (T)zd;
                acc += zd * (double)r.v[e];
            }
            stv(v.s[c] + a, s);
        }
    }
    const double tot = block_sum_256(acc, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && tot != 0.0) atomicAdd(sc.sig(0) + sc.my_slot(), tot);
}

// ------------------------------------------------------------------ K2: x += alpha s ; r -= alpha q ; s = r/d + beta s ; sigma' ; rmax
// (the scalar prologue, the stop test and the stall guard are k_pcg_update's, pcg_geo.inc)
template <typename T>
__global__ __launch_bounds__(256) void k_bpcg_update(const int *__restrict__ bricks, int nb, BrickSys<T> v, PcgScal sc, int it_arg) {
    BrickWalkV w;   // 16-byte accesses: a lane owns 4 consecutive entries of a brick (brick.h)
    w.begin(bricks, nb, v.mask);
    // The 15 vectors of a lane are requested as soon as its mask word is known -- the first group's BEFORE the scalar prologue (stop flag,
    // iteration counter, the 160 partial sums and a barrier, none of which the data depends on), the next group's before the current
    // one's stores (which may alias for all the compiler knows): the kernel is a chain of dependent round trips, not a stream.
    struct Data { Vec<float, 4> d[3]; Vec<T, 4> x[3], s[3], q[3]; Vec<RT<T>, 4> r[3]; };
    auto fetch = [&](size_t a, unsigned m) {
        Data D;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const bool on = BrickWalkV::any(m, c);
            D.d[c] = on ? ldv<4>(v.diag[c] + a) : Vec<float, 4>{};
            D.x[c] = on ? ldv<4>(v.x[c] + a) : Vec<T, 4>{}; D.s[c] = on ? ldv<4>(v.s[c] + a) : Vec<T, 4>{}; D.q[c] = on ? ldv<4>(v.q[c] + a) : Vec<T, 4>{};
            D.r[c] = on ? ldv<4>(v.r[c] + a) : Vec<RT<T>, 4>{};
        }
        return D;
    };
    Data cur = fetch(w.a, w.m);
    __shared__ double lds[8];
    int it;
    double alpha_d, beta_d;
    if (!d_update_scalars(sc, it_arg, lds, it, alpha_d, beta_d)) return;
    const T alpha = (T)alpha_d;
    double acc = 0.0;
    float mxf = 0.0f, mxs = 0.0f;   // max|r|, max|alpha s| (PcgScal::step)
    double mxd = 0.0;
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, v.mask);
        const Data nxt = fetch(w.a, w.m);
        Data D = cur;
        cur = nxt;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!BrickWalkV::any(m, c)) continue;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float d = D.d[c].v[e];
                if (!BrickWalkV::row(m, c, e) || d == 0.0f) continue;
                const RT<T> rn_t = (RT<T>)((double)D.r[c].v[e] - alpha_d * (double)D.q[c].v[e]);
                const double rn = (double)rn_t;
                const double zn = sizeof(RT<T>) == 4 ? (double)((float)rn_t / d) : rn / (double)d;
                D.x[c].v[e] = D.x[c].v[e] + alpha * D.s[c].v[e];
                if ((m >> (8 * e + 3 + c)) & 1u) mxs = fmaxf(mxs, fabsf((float)(alpha * D.s[c].v[e])));   // (rows whose velocity the substep uses: k_visc_setup)
                D.r[c].v[e] = rn_t;
                D.s[c].v[e] = (T)(zn + beta_d * (double)D.s[c].v[e]);
                if (sizeof(RT<T>) == 4) mxf = fmaxf(mxf, fabsf((float)rn_t)); else mxd = fmax(mxd, fabs(rn));
                acc += zn * rn;
            }
            stv(v.x[c] + a, D.x[c]);
            stv(v.r[c] + a, D.r[c]);
            stv(v.s[c] + a, D.s[c]);
        }
    }
    __shared__ double red[12];
    double tot = wave_sum(acc), bm = wave_max(fmax((double)mxf, mxd)), bs = wave_max((double)mxs);
    {
        const int tid = d_tid256();
        if ((tid & 63) == 0) { red[tid >> 6] = tot; red[4 + (tid >> 6)] = bm; red[8 + (tid >> 6)] = bs; }
        __syncthreads();
        if (tid == 0) {
            tot = red[0] + red[1] + red[2] + red[3];
            bm = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
            bs = fmax(fmax(red[8], red[9]), fmax(red[10], red[11]));
        }
    }
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int sl = sc.my_slot();
        if (tot != 0.0) atomicAdd(sc.sig(it + 1) + sl, tot);
        if (bm > 0.0) atomic_max_nonneg(sc.rmax(it) + sl, bm);
        if (bs > 0.0) atomic_max_nonneg(sc.step(it) + sl, bs);
        if (it_arg < 0 && blockIdx.x == 0) *sc.itA = it + 1;
    }
}

// (One kernel per Jacobi-PCG iteration -- the update of iteration k-1 recomputed on load at the 27 (position, component) pairs of iteration
// k's SpMV, r / s / q double-buffered: 136 algorithmic bytes per unit instead of 160, one launch instead of two -- was built and measured:
// identical iterates, 93 us per iteration against 42 at 256^3 (65.7 against 29.1 ms per capped solve), because the stencil loads
// quadruple (r, q, d next to s at every position: 127 dword loads per lane through L1/L2).  profiles/r3/fused_iteration_ab.log; the code
// is in the commit that added that file.)

// ------------------------------------------------------------------ residual replacement (group-wise update, van der Vorst & Ye)
// An fp32 PCG recurrence r -= alpha q drifts away from the true residual b - A x once |r| has dropped a few orders below |b| (here:
// nu dt/dx^2 ~ 3e3, diagonal / own volume ~ 1e3-1e4), and an fp32 x cannot even represent the solution to a true residual of 1e-6 |b|.
// Every `period` iterations the x accumulated so far is flushed into an fp64 accumulator (xacc += x; x = 0) and r is REPLACED by
// b - A xacc evaluated in fp64; the search direction is kept, sigma' = (r/d, r) and max|r| are recomputed from the replaced residual.
// Cost: two launches per `period` iterations.  The solution of the solve is xacc (after a last flush).
//   k_bflush:     xacc += x, x = 0 ; clears the scalars the residual kernel re-accumulates
//   k_bresidual:  r = b - A xacc (fp64 arithmetic, fp32 coefficients) ; optional z = omega r/d (the multigrid loop's first pre-sweep) ;
//                 rmax(it) [and sigma(it + 1) = (r/d, r) when `withSigma`: the diagonal loop]
// when: the launches act if force, or if iteration (it + 1) is a multiple of period, it = it_arg or the device-side counter *sc.itB
__device__ __forceinline__ bool d_replace_now(const PcgScal &sc, int it_arg, int period, int force, int &it) {
    it = it_arg >= 0 ? it_arg : *sc.itB;
    if (force) return true;
    if (*sc.conv >= 0 || it >= sc.cap) return false;
    return period > 0 && ((it + 1) % period) == 0;
}
template <typename T>
__global__ __launch_bounds__(256) void k_bflush(const int *__restrict__ bricks, int nb, BrickSys<T> v, double *__restrict__ xaU, double *__restrict__ xaV,
                                                double *__restrict__ xaW, PcgScal sc, int it_arg, int period, int force, int withSigma, int mode) {
    // mode 0: xacc += x, x = 0 (the flush); 1: xacc += x, x KEPT (a tentative flush: the caller decides from the recomputed residual);
    // 2: xacc -= x, x = 0 (the tentative flush taken back); 3: x = 0 (... or confirmed)
    BrickWalk w;
    w.begin(bricks, nb, v.mask);
    int it;
    if (!d_replace_now(sc, it_arg, period, force, it)) return;
    if (!force && blockIdx.x == 0 && threadIdx.y == 0 && threadIdx.x < NSLOT) {   // the replaced residual's scalars are accumulated afresh
        for (int b = 0; b < sc.nbank; b++) {
            sc.rmax(it)[threadIdx.x + (size_t)b * sc.bstride] = 0.0;
            if (withSigma) sc.sig(it + 1)[threadIdx.x + (size_t)b * sc.bstride] = 0.0;
        }

    }
    double *xa[3] = {xaU, xaV, xaW};
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, v.mask);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!((m >> c) & 1u)) continue;
            if (mode == 0 || mode == 1) xa[c][a] += (double)v.x[c][a];
            else if (mode == 2) xa[c][a] -= (double)v.x[c][a];
            if (mode != 1) v.x[c][a] = (T)0;
        }
    }
}
// (RefRowFactors / d_ref_row_factors: visc_rows.h)
struct RefRowInputs {   // perRow != 0: the plain-layout arrays the factors above are formed from, and the rows' exact own volumes (brick layout)
    const float *nu, *vC, *vEU, *vEV, *vEW, *xmU, *xmV, *xmW;
    Lay L, LB;
    float factor;
    int perRow;
};
template <typename T>
__global__ __launch_bounds__(256) void k_bresidual(const int *__restrict__ bricks, int nb, const float *__restrict__ vmU, const float *__restrict__ vmV,
                                                   const float *__restrict__ vmW, const float *__restrict__ fC, const float *__restrict__ fEU,
                                                   const float *__restrict__ fEV, const float *__restrict__ fEW, BrickSys<T> v,
                                                   const double *__restrict__ xu, const double *__restrict__ xv, const double *__restrict__ xw,
                                                   const float *__restrict__ bU, const float *__restrict__ bV, const float *__restrict__ bW,
                                                   float *__restrict__ zU, float *__restrict__ zV, float *__restrict__ zW, float omega,
                                                   PcgScal sc, int it_arg, int period, int withSigma, int force, RefRowInputs ref) {
    __shared__ double red[8];
    BrickWalk w;
    w.begin(bricks, nb, v.mask);
    int it;
    if (!d_replace_now(sc, it_arg, period, force, it)) return;
    const NbOff o = d_lane_off(v.sby, v.sbz);
    double acc = 0.0, mx = 0.0, mz = 0.0;
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, v.mask);
        if (m == 0u) continue;
        const float C0 = fC[a], C0l = fC[a + o.xm], Cjm = fC[a + o.ym], Ckm = fC[a + o.zm];
        const float EW0 = fEW[a], EW0r = fEW[a + o.xp], EWjp = fEW[a + o.yp];
        const float EV0 = fEV[a], EV0r = fEV[a + o.xp], EVkp = fEV[a + o.zp];
        const float EU0 = fEU[a], EUjp = fEU[a + o.yp], EUkp = fEU[a + o.zp];
        const double U0 = xu[a], U0l = xu[a + o.xm], U0r = xu[a + o.xp], Ujm = xu[a + o.ym], Ujp = xu[a + o.yp], Ukm = xu[a + o.zm], Ukp = xu[a + o.zp];
        const double Ujmr = xu[a + o.ym + o.xp], Ukmr = xu[a + o.zm + o.xp];
        const double V0 = xv[a], V0l = xv[a + o.xm], V0r = xv[a + o.xp], Vjm = xv[a + o.ym], Vjp = xv[a + o.yp], Vkm = xv[a + o.zm], Vkp = xv[a + o.zp];
        const double Vjpl = xv[a + o.yp + o.xm], Vjpkm = xv[a + o.yp + o.zm];
        const double W0 = xw[a], W0l = xw[a + o.xm], W0r = xw[a + o.xp], Wjm = xw[a + o.ym], Wjp = xw[a + o.yp], Wkm = xw[a + o.zm], Wkp = xw[a + o.zp];
        const double Wkpl = xw[a + o.zp + o.xm], Wjmkp = xw[a + o.ym + o.zp];
        const float MU = (m & 1u) ? vmU[a] : -1.0f, MV = (m & 2u) ? vmV[a] : -1.0f, MW = (m & 4u) ? vmW[a] : -1.0f;
        const double RU = (m & 1u) ? (double)bU[a] : 0.0, RV = (m & 2u) ? (double)bV[a] : 0.0, RW = (m & 4u) ? (double)bW[a] : 0.0;
        Vec<double, 1> yU, yV, yW;
        double ta = 0.0, tb = 0.0, tc = 0.0;
#define V1F(x_) Vec<float, 1>{{x_}}
#define V1D(x_) Vec<double, 1>{{x_}}
        d_visc_rows<double, 1, true, EPI_RESIDUAL>(V1F(MU), V1F(MV), V1F(MW), V1F(C0), V1F(Cjm), V1F(Ckm), V1F(EW0), V1F(EWjp), V1F(EV0), V1F(EVkp), V1F(EU0),
                                                   V1F(EUjp), V1F(EUkp), V1D(U0), V1D(Ujm), V1D(Ujp), V1D(Ukm), V1D(Ukp), V1D(V0), V1D(Vjm), V1D(Vjp), V1D(Vkm),
                                                   V1D(Vkp), V1D(W0), V1D(Wjm), V1D(Wjp), V1D(Wkm), V1D(Wkp), V1D(Vjpkm), V1D(Wjmkp), V1D(RU), V1D(RV), V1D(RW),
                                                   C0l, EW0r, EV0r, U0l, U0r, V0l, V0r, W0l, W0r, Vjpl, Wkpl, Ujmr, Ukmr, yU, yV, yW, ta, tb, tc, 0.0);
        if (ref.perRow) {   // the reference's rows, each with its own six factors and its own float-rounded diagonal (three passes: one component's row each)
            int pi, pj, pk;
            d_brick_ijk(ref.LB, (int)(a >> 6), (int)threadIdx.x, pi, pj, pk);
            const RefRowFactors F = d_ref_row_factors(ref.nu, ref.vC, ref.vEU, ref.vEV, ref.vEW, gidx(ref.L, pi, pj, pk), ref.L.sy, ref.L.sz, ref.factor);
            auto refvol = [](float vol, const float *f) {
                const float dg = vol + f[0] + f[1] + f[2] + f[3] + f[4] + f[5];
                return dg != 0.0f ? d_ref_volume(vol, f[0], f[1], f[2], f[3], f[4], f[5], dg) : -1.0f;
            };
            const float none = -1.0f;
            Vec<double, 1> d0, d1, d2;
            if (m & 1u) {
                const float M = refvol(ref.xmU[a], F.U);
                d_visc_rows<double, 1, true, EPI_RESIDUAL>(V1F(M), V1F(none), V1F(none), V1F(F.U[0]), V1F(Cjm), V1F(Ckm), V1F(F.U[3]), V1F(F.U[2]), V1F(F.U[5]), V1F(F.U[4]), V1F(EU0),
                                                           V1F(EUjp), V1F(EUkp), V1D(U0), V1D(Ujm), V1D(Ujp), V1D(Ukm), V1D(Ukp), V1D(V0), V1D(Vjm), V1D(Vjp), V1D(Vkm),
                                                           V1D(Vkp), V1D(W0), V1D(Wjm), V1D(Wjp), V1D(Wkm), V1D(Wkp), V1D(Vjpkm), V1D(Wjmkp), V1D(RU), V1D(RV), V1D(RW),
                                                           F.U[1], EW0r, EV0r, U0l, U0r, V0l, V0r, W0l, W0r, Vjpl, Wkpl, Ujmr, Ukmr, yU, d1, d2, ta, tb, tc, 0.0);
            }
            if (m & 2u) {
                const float M = refvol(ref.xmV[a], F.V);
                d_visc_rows<double, 1, true, EPI_RESIDUAL>(V1F(none), V1F(M), V1F(none), V1F(F.V[2]), V1F(F.V[3]), V1F(Ckm), V1F(F.V[1]), V1F(EWjp), V1F(EV0), V1F(EVkp), V1F(F.V[5]),
                                                           V1F(EUjp), V1F(F.V[4]), V1D(U0), V1D(Ujm), V1D(Ujp), V1D(Ukm), V1D(Ukp), V1D(V0), V1D(Vjm), V1D(Vjp), V1D(Vkm),
                                                           V1D(Vkp), V1D(W0), V1D(Wjm), V1D(Wjp), V1D(Wkm), V1D(Wkp), V1D(Vjpkm), V1D(Wjmkp), V1D(RU), V1D(RV), V1D(RW),
                                                           C0l, F.V[0], EV0r, U0l, U0r, V0l, V0r, W0l, W0r, Vjpl, Wkpl, Ujmr, Ukmr, d0, yV, d2, ta, tb, tc, 0.0);
            }
            if (m & 4u) {
                const float M = refvol(ref.xmW[a], F.W);
                d_visc_rows<double, 1, true, EPI_RESIDUAL>(V1F(none), V1F(none), V1F(M), V1F(F.W[4]), V1F(Cjm), V1F(F.W[5]), V1F(EW0), V1F(EWjp), V1F(F.W[1]), V1F(EVkp), V1F(F.W[3]),
                                                           V1F(F.W[2]), V1F(EUkp), V1D(U0), V1D(Ujm), V1D(Ujp), V1D(Ukm), V1D(Ukp), V1D(V0), V1D(Vjm), V1D(Vjp), V1D(Vkm),
                                                           V1D(Vkp), V1D(W0), V1D(Wjm), V1D(Wjp), V1D(Wkm), V1D(Wkp), V1D(Vjpkm), V1D(Wjmkp), V1D(RU), V1D(RV), V1D(RW),
                                                           C0l, EW0r, F.W[0], U0l, U0r, V0l, V0r, W0l, W0r, Vjpl, Wkpl, Ujmr, Ukmr, d0, d1, yW, ta, tb, tc, 0.0);
            }
        }
#undef V1F
#undef V1D
        const double rr[3] = {yU.v[0], yV.v[0], yW.v[0]};
        float *zz[3] = {zU, zV, zW};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!((m >> c) & 1u)) continue;
            const RT<T> rt = (RT<T>)rr[c];
            v.r[c][a] = rt;
            const float d = v.diag[c][a];
            const double z = d != 0.0f ? (double)rt / (double)d : 0.0;
            if (zU) zz[c][a] = omega * (float)z;
            acc += z * (double)rt;
            mx = fmax(mx, fabs((double)rt));
            mz = fmax(mz, fabs(z));
        }
    }
    double tot = wave_sum(acc), bm = wave_max(mx);
    {
        const int tid = d_tid256();
        if ((tid & 63) == 0) { red[tid >> 6] = tot; red[4 + (tid >> 6)] = bm; }
        __syncthreads();
        if (tid == 0) {
            tot = red[0] + red[1] + red[2] + red[3];
            bm = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
        }
    }
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int sl = sc.my_slot();
        if (withSigma && tot != 0.0) atomicAdd(sc.sig(it + 1) + sl, tot);
        if (bm > 0.0) atomic_max_nonneg(sc.rmax(it) + sl, bm);

    }
}

// ------------------------------------------------------------------ solution -> velocity grid (plain layout), over a launch box
template <typename T>
__global__ void k_unbrick_to_f32(Lay L, Lay LB, const T *__restrict__ a, const double *__restrict__ acc, float *__restrict__ o) {
    IJK_OF_THREAD(L);
    if (i >= L.ie || j >= L.je) return;
    const size_t b = bidx(LB, i, j, k);
    o[gidx(L, i, j, k)] = acc ? (float)(acc[b] + (double)a[b]) : (float)a[b];
}

// ------------------------------------------------------------------ host side
template <typename T>
static BrickSys<T> brick_sys(flipv_context *c) {
    BrickSys<T> v;
    v.mask = c->vMaskB;
    v.sby = (int)c->LB.sy * 64; v.sbz = (int)c->LB.sz * 64;
    v.diag[0] = c->vDiagU; v.diag[1] = c->vDiagV; v.diag[2] = c->vDiagW;
    for (int m = 0; m < 3; m++) { v.x[m] = (T *)c->vX[m]; v.r[m] = (RT<T> *)c->vR[m]; v.q[m] = (T *)c->vZ[m]; v.s[m] = (T *)c->vS[m]; }
    return v;
}
int fv_brick_grid(const flipv_context *c, int nbricks, int cap) {
    int nb = (((nbricks + 3) / 4 + 7) / 8) * 8;
    if (c->prm.grid_cap > 0 && cap > ((c->prm.grid_cap + 7) / 8) * 8) cap = ((c->prm.grid_cap + 7) / 8) * 8;   // test hook: every block walks many bricks
    if (nb > cap) nb = cap;
    return nb < 8 ? 8 : nb;
}
// 1280 blocks = 5 per CU = one resident round at 93-100 VGPRs (1536 measured slower, 2048 the same over a substep: profiles/r5/brick_spmv_shuffles.log)
static int spmv_grid(const flipv_context *c) { return fv_brick_grid(c, c->nBricks, c->prm.viscosity_spmv_grid_cap > 0 ? c->prm.viscosity_spmv_grid_cap : 1280); }
static int update_grid(const flipv_context *c) { return fv_brick_grid(c, c->nBricks, c->prm.viscosity_update_grid_cap > 0 ? c->prm.viscosity_update_grid_cap : 2048); }
// (the kernels that walk 16 bricks per block step: BrickWalkV)
static int updatev_grid(const flipv_context *c) {
    int cap = c->prm.viscosity_update_grid_cap > 0 ? c->prm.viscosity_update_grid_cap : 2048;
    if (c->prm.grid_cap > 0 && cap > ((c->prm.grid_cap + 7) / 8) * 8) cap = ((c->prm.grid_cap + 7) / 8) * 8;
    return fv_brickv_grid(c->nBricks, cap);
}

template <typename T>
void fv_brick_spmv(flipv_context *c, const PcgScal &sc, int it, bool rdot, int first, int count) {
    if (count < 0) { first = 0; count = c->nBricks; }
    const bool timed = c->prm.kernel_timing && (it & 7) == 0 && first == 0;
    if (timed) fv_ev_begin(c, 1, (double)count * 64);
    const float *const vo[3] = {c->vOperatorExact ? c->vmU : c->vrU, c->vOperatorExact ? c->vmV : c->vrV, c->vOperatorExact ? c->vmW : c->vrW};
    const BrickSys<T> v = brick_sys<T>(c);
    const dim3 g(fv_brick_grid(c, count, c->prm.viscosity_spmv_grid_cap > 0 ? c->prm.viscosity_spmv_grid_cap : 1280)), b(64, 4, 1);
    const int *list = (const int *)c->brickList + first;
    if (rdot) hipLaunchKernelGGL((k_bvisc_spmv<T, true, EPI_SPMV>), g, b, 0, c->stream, list, count, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, v, sc, it, 0.0f, 0);
    else if (sc.onlyA) hipLaunchKernelGGL((k_bvisc_spmv<T, false, EPI_SPMV_A>), g, b, 0, c->stream, list, count, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, v, sc, it, 0.0f, 0);
    else hipLaunchKernelGGL((k_bvisc_spmv<T, false, EPI_SPMV>), g, b, 0, c->stream, list, count, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, v, sc, it, 0.0f, 0);
    if (timed) fv_ev_end(c);
}
template void fv_brick_spmv<float>(flipv_context *, const PcgScal &, int, bool, int, int);
template void fv_brick_spmv<double>(flipv_context *, const PcgScal &, int, bool, int, int);

template <typename T>
void fv_brick_init(flipv_context *c, const PcgScal &sc) {
    hipLaunchKernelGGL((k_bpcg_init<T>), dim3(updatev_grid(c)), dim3(64, 4, 1), 0, c->stream, (const int *)c->brickList, c->nBricks, brick_sys<T>(c), sc);
}
template void fv_brick_init<float>(flipv_context *, const PcgScal &);
template void fv_brick_init<double>(flipv_context *, const PcgScal &);

template <typename T>
void fv_brick_update(flipv_context *c, const PcgScal &sc, int it) {
    hipLaunchKernelGGL((k_bpcg_update<T>), dim3(updatev_grid(c)), dim3(64, 4, 1), 0, c->stream, (const int *)c->brickList, c->nBricks, brick_sys<T>(c), sc, it);
}
template void fv_brick_update<float>(flipv_context *, const PcgScal &, int);
template void fv_brick_update<double>(flipv_context *, const PcgScal &, int);

// the multigrid's fine-level sweeps (fp32): out = in + omega (r - A in)/d (epi 1; 3 also adds (r, out) into sig(it + sig_shift)), out = r - A in (epi 2)
void fv_brick_sweep_f32(flipv_context *c, float *const in[3], float *const out[3], int epi, const PcgScal &sc, int it_arg, float omega, int sig_shift, int first, int count) {
    if (count < 0) { first = 0; count = c->nBricks; }
    if (count == 0) return;
    BrickSys<float> v = brick_sys<float>(c);
    for (int m = 0; m < 3; m++) { v.s[m] = in[m]; v.q[m] = out[m]; }
    const float *const vo[3] = {c->vOperatorExact ? c->vmU : c->vrU, c->vOperatorExact ? c->vmV : c->vrV, c->vOperatorExact ? c->vmW : c->vrW};
    const dim3 g(fv_brick_grid(c, count, c->prm.viscosity_spmv_grid_cap > 0 ? c->prm.viscosity_spmv_grid_cap : 1280)), b(64, 4, 1);
    const int *list = (const int *)c->brickList + first;
#define BSWEEP(E_) hipLaunchKernelGGL((k_bvisc_spmv<float, true, E_>), g, b, 0, c->stream, list, count, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, v, sc, it_arg, omega, sig_shift)
    if (epi == 1) BSWEEP(EPI_JACOBI); else if (epi == 2) BSWEEP(EPI_RESIDUAL); else BSWEEP(EPI_JACOBI_DOT);
#undef BSWEEP
}

// residual replacement (see above); z (optional): the multigrid loop's first pre-sweep vector, rewritten as omega r/d
template <typename T>
void fv_brick_replace(flipv_context *c, const PcgScal &sc, int it_arg, int period, int withSigma, float *const z[3], float omega) {
    const BrickSys<T> v = brick_sys<T>(c);
    const float *const vo[3] = {c->vOperatorExact ? c->vmU : c->vrU, c->vOperatorExact ? c->vmV : c->vrV, c->vOperatorExact ? c->vmW : c->vrW};
    const dim3 b(64, 4, 1);
    hipLaunchKernelGGL((k_bflush<T>), dim3(update_grid(c)), b, 0, c->stream, (const int *)c->brickList, c->nBricks, v, c->vXacc[0], c->vXacc[1], c->vXacc[2], sc, it_arg, period, 0, withSigma, 0);
    hipLaunchKernelGGL((k_bresidual<T>), dim3(spmv_grid(c)), b, 0, c->stream, (const int *)c->brickList, c->nBricks, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, v,
                       (const double *)c->vXacc[0], (const double *)c->vXacc[1], (const double *)c->vXacc[2], (const float *)c->vB[0], (const float *)c->vB[1],
                       (const float *)c->vB[2], z ? z[0] : nullptr, z ? z[1] : nullptr, z ? z[2] : nullptr, omega, sc, it_arg, period, withSigma, 0, RefRowInputs{});
}
template void fv_brick_replace<float>(flipv_context *, const PcgScal &, int, int, int, float *const[3], float);
template void fv_brick_replace<double>(flipv_context *, const PcgScal &, int, int, int, float *const[3], float);
// Iterative refinement after a stalled fp32 solve: xacc += x, x = 0, r = b - A xacc evaluated in fp64 -- the PCG loop is then run again
// on the correction equation A dx = r (whatever the stop flag says: the caller resets the solve's scalars afterwards)
// scalBytes: the solve's scalar block is cleared in between, so that rmax(0) afterwards is max|r| of the recomputed residual
// outerExact: which operator the recomputed residual belongs to -- the exact one, or the reference's float-rounded one (the operator the
// SOLVE is for; the PCG loop in between may run on the exact operator, see viscosity_solve_t)
template <typename T>
int fv_brick_refine(flipv_context *c, const PcgScal &sc, size_t scalBytes, bool outerExact, int flushMode) {
    const BrickSys<T> v = brick_sys<T>(c);
    const float *const vo[3] = {outerExact ? c->vmU : c->vrU, outerExact ? c->vmV : c->vrV, outerExact ? c->vmW : c->vrW};
    const dim3 b(64, 4, 1);
    RefRowInputs ref{};
    if (!outerExact && c->vPerRowFactors) {   // a variable viscosity field: the reference's rows with their own factors (d_ref_row_factors)
        ref.nu = c->visc; ref.vC = c->volC; ref.vEU = c->volEU; ref.vEV = c->volEV; ref.vEW = c->volEW;
        ref.xmU = c->vmU; ref.xmV = c->vmV; ref.xmW = c->vmW;
        ref.L = c->L; ref.LB = c->LB; ref.factor = c->vFactorNow; ref.perRow = 1;
    }
    hipLaunchKernelGGL((k_bflush<T>), dim3(update_grid(c)), b, 0, c->stream, (const int *)c->brickList, c->nBricks, v, c->vXacc[0], c->vXacc[1], c->vXacc[2], sc, 0, 0, 1, 0, flushMode);
    {   // bank 0's slot blocks (the extra scalars behind them stay), then the other banks
        const FillJob z[2] = {{sc.base, scalBytes, 0}, {sc.base + sc.bstride, sc.nbank > 1 ? (size_t)(sc.nbank - 1) * sc.bstride * sizeof(double) : 0, 0}};
        const int rcz = fv_fill_list(c, z, 2);
        if (rcz) return rcz;
    }
    if (c->comm) {   // the residual's stencil reads the neighbours' accumulated solution across the cuts
        const HaloArray xa[3] = {{c->vXacc[0], sizeof(double), 1}, {c->vXacc[1], sizeof(double), 1}, {c->vXacc[2], sizeof(double), 1}};
        const int rc = fv_halo_copy(c, xa, 3, 1);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((k_bresidual<T>), dim3(spmv_grid(c)), b, 0, c->stream, (const int *)c->brickList, c->nBricks, vo[0], vo[1], vo[2], c->fC, c->fEU, c->fEV, c->fEW, v,
                       (const double *)c->vXacc[0], (const double *)c->vXacc[1], (const double *)c->vXacc[2], (const float *)c->vB[0], (const float *)c->vB[1],
                       (const float *)c->vB[2], (float *)nullptr, (float *)nullptr, (float *)nullptr, 0.0f, sc, 0, 0, 0, 1, ref);
    return fv_allreduce_scalars(c, sc.rmax(0), NSLOT);   // (ranks write disjoint slots: the sum merges their maxima)
}
template int fv_brick_refine<float>(flipv_context *, const PcgScal &, size_t, bool, int);
template int fv_brick_refine<double>(flipv_context *, const PcgScal &, size_t, bool, int);
// a tentative flush (fv_brick_refine with flushMode 1) confirmed (x = 0) or taken back (xacc -= x, x = 0)
template <typename T>
void fv_brick_flush_settle(flipv_context *c, const PcgScal &sc, bool takeBack) {
    hipLaunchKernelGGL((k_bflush<T>), dim3(update_grid(c)), dim3(64, 4, 1), 0, c->stream, (const int *)c->brickList, c->nBricks, brick_sys<T>(c), c->vXacc[0], c->vXacc[1], c->vXacc[2], sc, 0, 0, 1, 0, takeBack ? 2 : 3);
}
template void fv_brick_flush_settle<float>(flipv_context *, const PcgScal &, bool);
template void fv_brick_flush_settle<double>(flipv_context *, const PcgScal &, bool);

// x (+ xacc) -> velocity grid over the launch box R
template <typename T>
void fv_brick_writeback(flipv_context *c, const Lay &R, int m, bool withAcc, float *dst) {
    hipLaunchKernelGGL((k_unbrick_to_f32<T>), GRID3(R), 0, c->stream, R, c->LB, (const T *)c->vX[m], withAcc ? (const double *)c->vXacc[m] : (const double *)nullptr, dst);
}
template void fv_brick_writeback<float>(flipv_context *, const Lay &, int, bool, float *);
template void fv_brick_writeback<double>(flipv_context *, const Lay &, int, bool, float *);
