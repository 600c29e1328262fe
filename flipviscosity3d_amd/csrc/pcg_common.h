// pcg_common.h -- tile-based, device-resident preconditioned conjugate gradients shared by the
// pressure (1 component, cell-centred) and viscosity (3 components, face-centred) solves.
//
// Algorithm = the reference's PCG loops (pressuresolver.cpp:521-567, pcgsolver.h:241-295) with the
// sequential MIC(0) triangular solves replaced by the diagonal preconditioner M = diag(A) -- the only part of
// the reference algorithm that does not parallelise.  One iteration is TWO kernels:
//
//   spmv   (K1): q = A s ;  a = s.q ;  b = (r/d).q ;  c = (q/d).q            (kernel supplied by the solver)
//   update (K2): alpha = sigma/a ; x += alpha s ; r -= alpha q ;
//                beta = (sigma - 2 alpha b + alpha^2 c)/sigma ;  s = r/d + beta s ;
//                sigma' = (r/d).r ; rmax = max|r|
//
// (r',z') = (r - alpha q, (r - alpha q)/d) expands to sigma - 2 alpha b + alpha^2 c, so beta is known as soon as the
// SpMV's three dot products are, and the classic third kernel (s = z + beta s after a second reduction) folds into
// the update.  The next iteration's alpha uses sigma' recomputed from the actual vectors, so nothing drifts.
// Everything stays on the device: the scalars are fp64 sums accumulated with one atomic per block into slots
// indexed by iteration (nothing is reset inside the loop); K2 stops the solve by setting a device flag once
// rmax of the previous iteration passes the tolerance, after which every launch returns immediately, so the
// host only polls every few iterations.
//
// Work decomposition: tiles of (ROWL N) x TY x 1 indices of the shared index space in one of two geometries (pcg_geo.inc:
// 16-lane rows = 64 x 16 tiles for sparse liquids, 64-lane rows = 256 x 4 tiles for full ones, picked per solve by
// fv_build_tiles); a lane owns N consecutive i (N = 4, or 2) and moves them with one 16- or 8-byte access per array;
// i-neighbours come from the adjacent lane (DPP row / wave shift), j/k-neighbours from aligned loads of the adjacent rows.
// This header holds everything that does not depend on the geometry; kernels live in the *_geo.inc files.
#pragma once
#include "flipv_internal.h"
#include "flipv_comm.h"

// N consecutive i of one lane (N = 4: one dwordx4 access for fp32; N = 2: dwordx2)
template <typename T, int N> struct Vec { T v[N]; };

template <int N> __device__ __forceinline__ Vec<float, N> ldv(const float *__restrict__ p);
template <> __device__ __forceinline__ Vec<float, 4> ldv<4>(const float *__restrict__ p) {
    const float4 q = *reinterpret_cast<const float4 *>(p);
    Vec<float, 4> r;
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
    return r;
}
template <> __device__ __forceinline__ Vec<float, 2> ldv<2>(const float *__restrict__ p) {
    const float2 q = *reinterpret_cast<const float2 *>(p);
    Vec<float, 2> r;
    r.v[0] = q.x; r.v[1] = q.y;
    return r;
}
template <int N> __device__ __forceinline__ Vec<double, N> ldv(const double *__restrict__ p);
template <> __device__ __forceinline__ Vec<double, 4> ldv<4>(const double *__restrict__ p) {
    const double2 a = *reinterpret_cast<const double2 *>(p);
    const double2 b = *reinterpret_cast<const double2 *>(p + 2);
    Vec<double, 4> r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    return r;
}
template <> __device__ __forceinline__ Vec<double, 2> ldv<2>(const double *__restrict__ p) {
    const double2 a = *reinterpret_cast<const double2 *>(p);
    Vec<double, 2> r;
    r.v[0] = a.x; r.v[1] = a.y;
    return r;
}
__device__ __forceinline__ void stv(float *__restrict__ p, const Vec<float, 4> &r) {
    *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
}
__device__ __forceinline__ void stv(float *__restrict__ p, const Vec<float, 2> &r) {
    *reinterpret_cast<float2 *>(p) = make_float2(r.v[0], r.v[1]);
}
__device__ __forceinline__ void stv(double *__restrict__ p, const Vec<double, 4> &r) {
    *reinterpret_cast<double2 *>(p) = make_double2(r.v[0], r.v[1]);
    *reinterpret_cast<double2 *>(p + 2) = make_double2(r.v[2], r.v[3]);
}
__device__ __forceinline__ void stv(double *__restrict__ p, const Vec<double, 2> &r) {
    *reinterpret_cast<double2 *>(p) = make_double2(r.v[0], r.v[1]);
}

// Streaming variants: the nontemporal hint of the load / store instruction, for arrays a kernel touches exactly once when the system
// is larger than the 256 MB memory-side cache (measured, filled 256^3 box: pressure SpMV 82 -> 66 us; on systems that fit the cache the
// hint costs 1-2 %, tools/micro/stream_mix.hip and DESIGN.md section 5)
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
template <bool STREAM, int N, typename S> __device__ __forceinline__ Vec<S, N> ldvs(const S *__restrict__ p) {
    if constexpr (sizeof(S) == 4 && STREAM) {
        Vec<S, N> r;
        if constexpr (N == 4) { const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p)); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
        else { const nt_f2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f2 *>(p)); r.v[0] = t.x; r.v[1] = t.y; }
        return r;
    } else {
        return ldv<N>(p);
    }
}
template <bool STREAM, int N, typename S> __device__ __forceinline__ void stvs(S *__restrict__ p, const Vec<S, N> &r) {
    if constexpr (sizeof(S) == 4 && STREAM) {
        if constexpr (N == 4) { nt_f4 t; t.x = r.v[0]; t.y = r.v[1]; t.z = r.v[2]; t.w = r.v[3]; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4 *>(p)); }
        else { nt_f2 t; t.x = r.v[0]; t.y = r.v[1]; __builtin_nontemporal_store(t, reinterpret_cast<nt_f2 *>(p)); }
    } else {
        stv(p, r);
    }
}

// 1/d for the preconditioned dot products: hardware reciprocal (1 ulp) with fp32 vectors, a division with fp64 vectors
template <typename T> __device__ __forceinline__ T d_recip(float d);
template <> __device__ __forceinline__ float d_recip<float>(float d) { return __builtin_amdgcn_rcpf(d); }
template <> __device__ __forceinline__ double d_recip<double>(float d) { return 1.0 / (double)d; }


// Every reduced scalar of iteration `it` is spread over NSLOT partial sums (slot = blockIdx & (NSLOT-1)) so that the
// one-atomic-per-block accumulation does not serialise on a single L2 address (thousands of blocks per launch);
// the consuming kernel folds the NSLOT partials with one wave.
// (NSLOT itself lives in flipv_internal.h: it also bounds the number of ranks of a communicator)
// Layout of ctx->d_scal: block `it` = [sig | a | b | c | rmax | step] x NSLOT doubles (FV_NSC of them).  rmax[it-1], step[it-1], sig[it] (all
// left by update it-1) and a,b,c[it] (left by SpMV it) are contiguous: ONE all-reduce per iteration, between SpMV and update.
// step[it] = max over the rows of |alpha s| of iteration it: by how much the iteration moved any one unknown (velocity units) -- what the
// velocity criterion of a solve's last loop tests (PcgScal::vel_tol).
// In a multi-rank run rank r accumulates only into slots [slot0, slot0 + nslot): the slots are disjoint between
// ranks, so a SUM all-reduce merges sums and maxima alike.
constexpr int FV_NSC = 6;   // scalars per iteration block
// The stall guard's factor: a loop stops once max|r| exceeds this x the smallest value it has reached.  What the guard is FOR -- an fp32 recurrence that has left the true
// residual behind and blows up by 1e8 over thousands of iterations (pcg_common.h: PcgScal::best) -- 16 catches early and cheaply.  But CG's max-norm residual legitimately rebounds
// by more than that on a system with a few isolated small eigenvalues -- a viscosity FIELD with a jump: 14 x in an fp64 Jacobi-PCG model of holdout draw 9, 20-50 x on the device --,
// and there the guard ended every correction stage after 10-40 iterations (the draw's 0.2 ... 0.9 max|u| of round 5): such solves take 1 000 (a blow-up is still caught; stagnation
// has its own guard, VMG_NO_PROGRESS).
constexpr double FV_STALL_RATIO = 16.0;
constexpr double FV_STALL_RATIO_FIELD = 1000.0;
struct PcgScal {
    double *base;
    int *conv;      // converged-at iteration, -1 while running (nullptr: benchmark launch, no scalars)
    double tol;
    int tol_inclusive;  // 1: res <= tol (pcgsolver.h:270), 0: res < tol (pressuresolver.cpp:548)
    int slot0, nslot;
    // Banks: NSLOT slots still put ~40 blocks of a 1 280-block launch on the SAME 8-byte address, and same-address atomics serialise in L2 -- the
    // tail of every kernel with a reduction (measured, brick SpMV at 256^3: 18.0 us with its three dot products, 13.9 without, 13.9 with the blocks
    // spread over four copies of the slot area).  Block b accumulates into bank (b / nslot) % nbank = the same slot layout `bstride` doubles
    // further on; consumers add (or max) the banks while they load the slots.  One bank under a communicator (the all-reduce sums bank 0).
    int nbank;
    int bstride;
    int cap;        // iteration cap
    int noB;        // 1: the SpMV did not form b = (r/d).q; the update uses b = a (conjugacy of successive directions)
    int onlyA;      // 1: the loop's update reads a = p.q alone (the multigrid-preconditioned loop): its SpMV launches the EPI_SPMV_A variant (visc_rows.h)
    // Stall guard.  An fp32 solve whose attainable residual sits right at the tolerance can miss it by a hair, stagnate and --
    // thousands of iterations later, the recurrence residual having drifted from the true one -- blow up (seen on a thin-sheet
    // scene with the cap lifted: relative residual 1.4e-6 against a tolerance of 1e-6 around iteration 1 100, 1e+2 at 3 000).
    // *best = smallest max|r| so far; once that is within 100 x the tolerance and the current residual exceeds 16 x *best the
    // update kernel stops the solve (sets *stalled and the stop flag): the iterate of that moment is returned as "not converged".
    double *best;   // nullptr: no guard
    int *stalled;
    int *bestIt;    // iteration at which *best last improved by 10 % (the multigrid loop's no-progress guard, k_viscosity_mg.hip: d_vmg_stop_test); nullptr: none
    double stall_ratio;   // the guard's factor (FV_STALL_RATIO; flipv_debug_params.stall_guard_ratio)
    double stall_below;   // the guard arms once *best <= this; 0 = 100 x tol.  (A loop restarted close to its tolerance -- iterative refinement --
                          // sets it to a fraction of the restart's residual: the first iterations of CG overshoot it in the max norm.)
    // device-side iteration counters for hipGraph replay (kernels launched with it_arg = -1): the SpMV reads itA and
    // publishes it in itB, the update reads itB and stores itB+1 in itA -- a kernel never reads a counter that is
    // written inside the same launch, so late-starting blocks cannot see a half-advanced iteration.
    int *itA, *itB;
    // Velocity criterion (0 = off): a loop that carries it is only "converged" once max|r| passes `tol` AND the last `vel_window` iterations together
    // moved no unknown by more than vel_tol (the sum of their step[] entries).  A residual that passes the reference's max|r| <= 1e-6 max|rhs| does not
    // bound the velocity error where the liquid holds light, weakly attached parts -- films and specks whose control volumes sum to a few per cent of a
    // cell: their near-rigid modes have residual = mass x error -- and there CG is still moving velocities by 1e-4 of their maximum per iteration when
    // the residual test passes (64^3 bunny resting on the wall at nu = 200: 2.7e-4 from the converged reference at 2.4e-6 max|rhs|; profiles/r5).
    double vel_tol;
    double vel_stall;   // > 0: the loop also ends once its residual has passed and the window's movement is >= vel_stall x the previous window's, within 10 x vel_tol (d_steps_small)
    int vel_window;
    int vel_patience;   // the criterion holds a loop whose residual has passed for at most this many further iterations (then the loop ends as converged; flipv_solve_info.velocity_step says what was left)
    int *passIt;        // device: 1 + the iteration at which the residual test first passed while the velocity criterion did not (0: not yet); the int behind bestIt
    __host__ __device__ double *sig(int it) const { return base + (size_t)it * FV_NSC * NSLOT; }          // (r,z) entering iteration it
    __host__ __device__ double *a(int it) const { return base + (size_t)it * FV_NSC * NSLOT + NSLOT; }
    __host__ __device__ double *b(int it) const { return base + (size_t)it * FV_NSC * NSLOT + 2 * NSLOT; }
    __host__ __device__ double *c(int it) const { return base + (size_t)it * FV_NSC * NSLOT + 3 * NSLOT; }
    __host__ __device__ double *rmax(int it) const { return base + (size_t)it * FV_NSC * NSLOT + 4 * NSLOT; }  // max|r| after iteration it
    __host__ __device__ double *step(int it) const { return base + (size_t)it * FV_NSC * NSLOT + 5 * NSLOT; }  // max|alpha s| of iteration it
    __device__ int my_slot() const {   // offset of this block's partial inside a slot block (bank included)
        const unsigned b = blockIdx.x;
        return slot0 + (int)(b % (unsigned)nslot) + (nbank > 1 ? (int)((b / (unsigned)nslot) % (unsigned)nbank) * bstride : 0);
    }
    // slot i of a slot block with the banks folded in
    // (nbank is 1 or FV_SCAL_BANKS = 4: the four loads are issued together -- these sit in the prologue of every block of the consumer)
    __device__ __forceinline__ double slot_sum(const double *p, int i) const {
        if (nbank <= 1) return p[i];
        const double v0 = p[i], v1 = p[i + (size_t)bstride], v2 = p[i + 2 * (size_t)bstride], v3 = p[i + 3 * (size_t)bstride];
        return (v0 + v1) + (v2 + v3);
    }
    __device__ __forceinline__ double slot_max(const double *p, int i) const {
        if (nbank <= 1) return p[i];
        const double v0 = p[i], v1 = p[i + (size_t)bstride], v2 = p[i + 2 * (size_t)bstride], v3 = p[i + 3 * (size_t)bstride];
        return fmax(fmax(v0, v1), fmax(v2, v3));
    }
};

__device__ __forceinline__ int d_tid256() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }

// fold the NSLOT partials of up to four scalars (sum) -- executed by every thread of the block, one barrier
__device__ __forceinline__ void d_fold_sums(const PcgScal &sc, const double *p0, const double *p1, const double *p2, const double *p3,
                                            double out[4], double *lds8) {
    const int tid = d_tid256();
    if (tid < 64) {
        const int sl = tid & (NSLOT - 1);
        const double *p = (tid < NSLOT) ? p0 : p1;
        double v0 = p ? sc.slot_sum(p, sl) : 0.0;
        const double *pp = (tid < NSLOT) ? p2 : p3;
        double v1 = pp ? sc.slot_sum(pp, sl) : 0.0;
        // lanes 0..31 hold p0/p2 partials, lanes 32..63 hold p1/p3 partials: reduce the two halves separately
#pragma unroll
        for (int off = NSLOT / 2; off > 0; off >>= 1) {
            v0 += __shfl_down(v0, off, NSLOT);
            v1 += __shfl_down(v1, off, NSLOT);
        }
        if (tid == 0) { lds8[0] = v0; lds8[2] = v1; }
        if (tid == NSLOT) { lds8[1] = v0; lds8[3] = v1; }
    }
    __syncthreads();
    out[0] = lds8[0]; out[1] = lds8[1]; out[2] = lds8[2]; out[3] = lds8[3];
    __syncthreads();
}
__device__ __forceinline__ double d_fold_max(const PcgScal &sc, const double *p, double *lds8) {
    const int tid = d_tid256();
    if (tid < NSLOT) {
        double v = sc.slot_max(p, tid);
#pragma unroll
        for (int off = NSLOT / 2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, NSLOT));
        if (tid == 0) lds8[0] = v;
    }
    __syncthreads();
    const double r = lds8[0];
    __syncthreads();
    return r;
}

// block-wide sum of three values with a single barrier pair; result valid in thread 0
__device__ __forceinline__ void block_sum3_256(double &a, double &b, double &c, double *lds12) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int tid = d_tid256();
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) { lds12[wv] = a; lds12[4 + wv] = b; lds12[8 + wv] = c; }
    __syncthreads();
    if (tid == 0) {
        a = lds12[0] + lds12[1] + lds12[2] + lds12[3];
        b = lds12[4] + lds12[5] + lds12[6] + lds12[7];
        c = lds12[8] + lds12[9] + lds12[10] + lds12[11];
    }
    __syncthreads();
}

// x, s (search direction), q = A s and the residual r are stored in T (fp32 by default, fp64 with
// flipv_params.precision = 1).  -DFLIPV_R64=1 keeps r in fp64 even for T = float (measured: no gain in attainable
// accuracy once the SpMV is evaluated in difference form, 9 % slower).
#ifndef FLIPV_MARCH_OCC
#define FLIPV_MARCH_OCC 2   // blocks per CU of the k-marching viscosity SpMV (k_viscosity_geo.inc)
#endif
#ifndef FLIPV_R64
#define FLIPV_R64 0
#endif
#if FLIPV_R64
template <typename T> using RT = double;
#else
template <typename T> using RT = T;
#endif

template <typename T, int NC>
struct PcgSys {
    const float *diag[NC];
    T *x[NC], *q[NC], *s[NC];
    RT<T> *r[NC];
    int swz;              // 1: diag, x, q, r are stored in the swizzled plane layout (sidx, flipv_internal.h); s and the mask never are
    const unsigned *mlist;  // optional: the mask words of the listed tiles in list order (fv_build_tiles); must match the list the kernel is given
    const uint8_t *mask;  // optional (nullptr: none): non-zero where any component has an unknown at the index; a lane whose
                          // N indices are all zero skips every load (sparse liquids: most lanes of an active tile)
};

// the N mask bytes of a lane as one integer (N = 2: 2-byte load, N = 4: 4-byte load; the index is a multiple of N)
template <int N> __device__ __forceinline__ unsigned ld_mask(const uint8_t *__restrict__ p);
template <> __device__ __forceinline__ unsigned ld_mask<2>(const uint8_t *__restrict__ p) { return *reinterpret_cast<const unsigned short *>(p); }
template <> __device__ __forceinline__ unsigned ld_mask<4>(const uint8_t *__restrict__ p) { return *reinterpret_cast<const unsigned *>(p); }

// Blocks loop over tiles with a grid stride (grids are capped at MAX_PCG_BLOCKS so a launch never issues more than
// that many scalar atomics); `b` is the virtual block index b = blockIdx.x + n*gridDim.x.
constexpr int MAX_PCG_BLOCKS = 1024;  // 4 blocks per CU: one resident round at the SpMV kernels' occupancy (measured best of 128..2048 at 256^3)

// Tile look-ahead.  The kernels are latency-bound on the reference's scenes (a wave spends most of its life parked on
// dependent loads: tile id -> mask -> data), so a block fetches the ids of its next TBATCH tiles with independent loads,
// then their masks with independent loads, and only then walks them: 2 + TBATCH dependent round trips per TBATCH
// tiles instead of 3 per tile.
constexpr int TBATCH = 4;
struct TileBatch { int id[TBATCH]; };  // -1: no tile
__device__ __forceinline__ TileBatch d_fetch_tiles(int base, int nvb, const int *__restrict__ tiles, int ntiles) {
    TileBatch B;
#pragma unroll
    for (int t = 0; t < TBATCH; t++) {
        const int b = base + t * (int)gridDim.x;
        const int slot = d_tile_slot(b, ntiles);
        B.id[t] = (b < nvb && slot < ntiles) ? tiles[slot] : -1;
    }
    return B;
}
__device__ __forceinline__ int d_pick(const int v[TBATCH], int t) { return t == 0 ? v[0] : (t == 1 ? v[1] : (t == 2 ? v[2] : v[3])); }
__device__ __forceinline__ unsigned d_pick(const unsigned v[TBATCH], int t) { return t == 0 ? v[0] : (t == 1 ? v[1] : (t == 2 ? v[2] : v[3])); }

// iteration argument of the multigrid-PCG vector kernels meaning "read the device-side counter" (hipGraph replay); the
// SpMV / update kernels use -1 for the same purpose, where -1 already means "before the first iteration" to the former
constexpr int IT_DEVICE = -2;

// k-marching work unit of the SpMV kernels (pcg_geo.inc): `len` tiles of one column, consecutive in k, starting at `tile`
constexpr int RUNLEN_MAX = 64;
struct Run { int tile, len; };

__device__ __forceinline__ bool d_pass(const PcgScal &sc, double res) { return sc.tol_inclusive ? (res <= sc.tol) : (res < sc.tol); }
// The velocity criterion (PcgScal::vel_tol): true when the iterations it_last - vel_window + 1 .. it_last together moved no unknown by more than vel_tol.
// EVERY thread of the block must call it (one barrier pair); lds8[7] is used.  Blocks of at least 64 threads.
__device__ __forceinline__ bool d_steps_small(const PcgScal &sc, int it_last, double *lds8) {
    if (!(sc.vel_tol > 0.0)) return true;
    const int tid = d_tid256();
    const int passed = sc.passIt ? *sc.passIt - 1 : -1;   // (stored + 1: zero = not yet)   // (written by an EARLIER launch, or in this one by block 0 with this very it_last: every block decides alike either way)
    if (tid < 64) {
        double sum = 0.0, before = 0.0;   // this window's movement; the window before it (only where the stall exit is on and the loop is old enough)
        const int nw = sc.vel_stall > 0.0 && it_last + 1 >= 2 * sc.vel_window ? 2 * sc.vel_window : sc.vel_window;
        for (int w = 0; w < nw; w++) {
            const int j = it_last - w;
            if (j < 0) break;
            double v = tid < NSLOT ? sc.slot_max(sc.step(j), tid) : 0.0;
#pragma unroll
            for (int off = NSLOT / 2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, NSLOT));
            if (w < sc.vel_window) sum += v; else before += v;
        }
        if (tid == 0) { lds8[7] = sum; lds8[6] = nw > sc.vel_window ? before : -1.0; }
    }
    __syncthreads();
    const double s = lds8[7], sb = lds8[6];
    __syncthreads();
    if (s <= sc.vel_tol) return true;
    // The movement has stopped shrinking (this window >= vel_stall x the window before it) within 10 x the tolerance: what still moves is a part the system does not
    // determine at this precision -- massless fringe rows, a speck -- and CG goes on moving it by 1e-5 ... 1e-3 max|u| per iteration while max|r| falls by three more
    // orders (256^3 bunny on the wall, 40 % of the substeps: profiles/r5/step_history_256.log).  More iterations add nothing there but a random walk of those rows.
    if (sc.vel_stall > 0.0 && sb >= 0.0 && s <= 10.0 * sc.vel_tol && s >= sc.vel_stall * sb) return true;
    // A light speck whose velocity the system barely determines keeps CG moving it for ever: the criterion holds the loop for vel_patience iterations past
    // the residual test, no longer.
    if (passed >= 0 && it_last - passed >= sc.vel_patience) return true;
    if (passed < 0 && sc.passIt && blockIdx.x == 0 && tid == 0) *sc.passIt = it_last + 1;
    return false;
}

// K1 prologue shared by both SpMV kernels: returns true if the launch must do nothing
// (the stop flag and the iteration counter are fetched together, not one after the other)
__device__ __forceinline__ int d_iter_spmv(const PcgScal &sc, int it_arg, bool &stop) {
    stop = false;
    if (!sc.conv) return it_arg < 0 ? 0 : it_arg;  // benchmark launches
    const int conv = *sc.conv;
    int it = it_arg;
    if (it_arg < 0) {
        it = *sc.itA;
        if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) *sc.itB = it;
    }
    stop = conv >= 0 || it >= sc.cap;
    return it;
}


// after a chunk of iterations: record convergence of the chunk's last iteration (K1 of the next iteration would)
static __global__ void k_pcg_check(PcgScal sc, int it_last_arg) {  // <<<1, 64>>>
    __shared__ double lds[8];
    const int it_last = it_last_arg >= 0 ? it_last_arg : *sc.itA - 1;
    if (it_last < 0) return;
    const double res = d_fold_max(sc, sc.rmax(it_last), lds);
    const bool ok = d_pass(sc, res) && d_steps_small(sc, it_last, lds);
    if (threadIdx.x == 0 && *sc.conv < 0 && ok) *sc.conv = it_last;
}
// final residual of iteration `it` into out[0]
// (out[1], where asked for: what the iterations it - vel_window + 1 .. it moved, the sum of their step[] entries -- the velocity criterion's quantity)
static __global__ void k_pcg_residual(PcgScal sc, int it, double *out, int withSteps = 0) {  // <<<1, 64>>>
    __shared__ double lds[8];
    const double res = d_fold_max(sc, sc.rmax(it), lds);
    if (threadIdx.x == 0) out[0] = res;
    if (withSteps) {
        PcgScal t = sc;
        t.vel_tol = 1.0;
        t.passIt = nullptr;
        if (t.vel_window < 1) t.vel_window = 4;
        (void)d_steps_small(t, it, lds);
        if (threadIdx.x == 0) out[1] = lds[7];
    }
}

// Run a statement with the kernels of one geometry in scope:  GEO_RUN(tg.rowl, hipLaunchKernelGGL((k_pcg_update<T, 3, 4>), ...));
#define GEO_RUN(rowl, ...)                                                  \
    do {                                                                    \
        if ((rowl) == 16) { using namespace g16; __VA_ARGS__; }             \
        else { using namespace g64; __VA_ARGS__; }                          \
    } while (0)

// ---- host-side helpers (k_pressure.hip) ----
int fv_scal_reserve(flipv_context *c, int cap);  // d_scal holds FV_SCAL_BANKS banks of FV_NSC*(cap+2)*NSLOT+16 doubles (rounded up to 512)
int fv_scal_clear(flipv_context *c, int cap, bool keepExtra);   // zero the slot blocks of every bank (and, unless keepExtra, the 16 extra doubles behind bank 0's)
constexpr int FV_SCAL_BANKS = 4;   // (PcgScal::slot_sum / slot_max spell the four banks out)
static inline size_t fv_scal_stride(int cap) { return (((size_t)FV_NSC * (cap + 2) * NSLOT + 16) + 511) / 512 * 512; }
// never 0: a rank without unknowns still runs the (empty) kernels so that the stop logic is identical on every rank
static inline int pcg_grid(const flipv_context *c, int ntiles) {
    int cap = c->prm.grid_cap > 0 ? ((c->prm.grid_cap + 7) / 8) * 8 : MAX_PCG_BLOCKS;  // test hook: small grids make every block walk many tiles
    if (cap > MAX_PCG_BLOCKS) cap = MAX_PCG_BLOCKS;
    const int nb = ((ntiles + 7) / 8) * 8;
    return nb < 8 ? 8 : (nb < cap ? nb : cap);
}
// Builds the tile list and picks the geometry: starts from *tg's (the previous solve's), and switches when the tiles come out
// less than 45 % full with 64-lane rows / more than 65 % full with 16-lane rows (full = share of the listed tiles' indices
// inside the lattices' extent that carry unknowns).  `hostCount` (read after the internal
// synchronisation) = unknowns of this rank, `perIndex` unknowns per index (1 pressure, 3 viscosity).
int fv_build_tiles(flipv_context *c, TileGrid *tg, int vw, int nc, const float *d0, const float *d1, const float *d2,
                   const uint8_t *mask, int *list, int *nActive, int *nInterior, const int *hostCount, int perIndex, unsigned **mlist, size_t *mlistCap, double minLanes, int *memo);
void fv_scal_views(flipv_context *c, int cap, PcgScal *sc, double **extra);
int fv_pcg_reset(flipv_context *c, int cap, bool keepExtra, PcgScal *sc, double **extra, int *alsoZero);   // clear + views + stop flag + counters: one launch
// Runs of the tile list fv_build_tiles just built (call right after it: uses its tile flags).  *nruns = 0 when the k-marching
// kernels are not to be used (multi-rank runs, lane width 2, flipv_params.spmv_run_length = -1).
// `dense`: the liquid fills the listed tiles (what flipv_params.spmv_run_length = 0 decides by: k-marching pays where the
// launch is bound by bytes, not where it is bound by the number of blocks that have work).
int fv_build_runs(flipv_context *c, const TileGrid &tg, int vw, int nActive, bool dense, const uint8_t *mask, Run **runs, size_t *runCap, int *nruns,
                  int *runLen, unsigned **rmask, size_t *rmaskCap);

// An executable for the freshly captured graph g: the slot's cached executable updated in place when the topology still matches,
// a newly instantiated one otherwise (which then takes the slot).  The caller launches *ge and destroys g, never *ge.
static int fv_graph_exec(flipv_context *c, int slot, hipGraph_t g, hipGraphExec_t *ge) {
    hipGraphExec_t &cached = c->geCache[slot];
    if (cached) {
        hipGraphNode_t bad = nullptr;
        hipGraphExecUpdateResult res;
        if (hipGraphExecUpdate(cached, g, &bad, &res) == hipSuccess) { *ge = cached; return FLIPV_OK; }
        (void)hipGetLastError();
    }
    if (cached) { (void)hipGraphExecDestroy(cached); cached = nullptr; }
    hipError_t e = hipGraphInstantiate(&cached, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { cached = nullptr; c->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return FLIPV_ERR_HIP; }
    *ge = cached;
    return FLIPV_OK;
}

// The iteration loop shared by both solves.
//   spmv(first, count, it)  enqueues K1 over list entries [first, first+count) on c->stream
//   update(it)              enqueues K2 over the whole list
// it = -1 selects the device-side iteration counters (hipGraph replay).
//
// Single GPU without per-launch event timing: `every` iterations + the convergence check + the flag read-back are
// captured once into a hipGraph and replayed.
//
// Multi-rank (one slab per rank): per iteration
//   communication stream:  halo exchange of s (starts when update it-1 has finished)
//   c->stream:             K1 over the tiles of the interior planes  |  wait for the halo  |  K1 over the tiles of the
//                          two boundary planes  |  ONE all-reduce [rmax step(it-1) sig(it) a b c(it)]  |  K2
// so the exchange hides behind the interior SpMV and the only exposed communication is one 1.3 KB all-reduce.
//   post(it)                (optional, period `postPeriod` > 0) enqueues work after the update of every iteration whose number + 1 is a
//                           multiple of postPeriod -- residual replacement (k_viscosity_brick.hip).  The kernels re-test that condition
//                           themselves from the iteration number; the host only has to launch them wherever it can hold.
struct PcgNoPost { void operator()(int) const {} };
template <class Spmv, class Update, class Post = PcgNoPost>
static int pcg_run(flipv_context *c, const PcgScal &sc, int cap, const HaloArray *sh, int nsh, int nInt, int nAct, Spmv spmv,
                   Update update, int *conv_out, int geSlot, Post post = Post(), int postPeriod = 0) {
    // poll interval: an iteration after the stop costs two empty launches (~6 us) on one GPU but a halo exchange and an
    // all-reduce in a multi-rank run; a poll costs a read-back and a host wake-up (~14 us)
    const int every = c->prm.check_every > 0 ? c->prm.check_every : (c->comm ? 8 : 32);
    int conv = -1, rc;
    // post: a replayed chunk (it = -1) carries the launch at every position -- the kernels re-test the absolute iteration number themselves and the chunk
    // does not know where it sits in the solve --, the kernel-by-kernel loop launches it where it is due
    auto post_due = [&](int it) { return postPeriod > 0 && (it < 0 || ((it + 1) % postPeriod) == 0); };
    auto launch_iter = [&](int it, int e) -> int {
        int r;
        (void)e;
        if (!c->comm) { spmv(0, nAct, it); update(it); if (post_due(it)) post(it); return FLIPV_OK; }
        const long ex0 = c->nExchanges, ar0 = c->nAllReduces;
        if ((r = fv_halo_copy_begin(c, sh, nsh, 1))) return r;
        if (nInt > 0) spmv(0, nInt, it);
        if ((r = fv_halo_wait(c))) return r;
        if (nAct > nInt) spmv(nInt, nAct - nInt, it);
        r = it == 0 ? fv_allreduce_scalars(c, sc.a(0), 3 * NSLOT) : fv_allreduce_scalars(c, sc.rmax(it - 1), 6 * NSLOT);   // [rmax step](it - 1) [sig a b c](it)
        if (r) return r;
        update(it);
        c->exchIter = (int)(c->nExchanges - ex0); c->allrIter = (int)(c->nAllReduces - ar0);
        return FLIPV_OK;
    };
    const bool graph = !c->comm && !c->prm.kernel_timing && !c->prm.no_graph_replay;
    if (graph) {
        HIPCHK(c, hipMemsetAsync(sc.itA, 0, 2 * sizeof(int), c->stream));
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        rc = FLIPV_OK;
        for (int e = 0; e < every && rc == FLIPV_OK; e++) rc = launch_iter(-1, e);
        const int e1 = fv_read_capture(c, c->d_flags, 1);   // the stop flag, published to the host at the end of every replay
        hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        if (rc != FLIPV_OK || e1 != FLIPV_OK || e2 != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            c->err = "pcg_run: stream capture failed";
            return rc != FLIPV_OK ? rc : FLIPV_ERR_HIP;
        }
        if ((rc = fv_graph_exec(c, geSlot, g, &ge))) { (void)hipGraphDestroy(g); return rc; }
        // Replays are pipelined: replay k+1 is enqueued before the host waits for replay k's stop flag, so the GPU does not idle
        // through the host's wake-up and launch between replays (22 of them in a capped solve: ~0.4 ms).  The flag only ever goes
        // from -1 to an iteration number, so whichever replay's copy the host reads is valid; a solve that stops early leaves one
        // replay of launches that return at once behind it.
        {
            int pending = 0;   // the publication number of the replay before the one just enqueued (0: none)
            bool bad = false;
            if ((rc = fv_read_wait(c))) { (void)hipGraphDestroy(g); return rc; }
            for (int done = 0; done < cap && conv < 0; done += every) {
                bad = bad || hipGraphLaunch(ge, c->stream) != hipSuccess;
                fv_read_replayed(c);
                if (pending) { bad = bad || fv_read_wait_seq(c, pending) != FLIPV_OK; conv = __atomic_load_n(c->h_pub + FV_PUB_REPLAY, __ATOMIC_ACQUIRE); }
                if (bad) break;
                pending = c->pubSeq;
            }
            bad = bad || fv_read_wait(c) != FLIPV_OK;
            if (bad) { (void)hipGraphDestroy(g); c->err = "hipGraphLaunch failed"; return FLIPV_ERR_HIP; }
            conv = c->h_pub[FV_PUB_REPLAY];
        }
        (void)hipGraphDestroy(g);
    } else {
        // Chunks of `every` iterations, each followed by a read-back of the stop flag into its own pinned slot.  The next
        // chunk is enqueued BEFORE the host waits for the previous chunk's flag, so the GPU never drains while the host
        // polls; if that flag says "converged" the chunk already in flight returns launch by launch at once.
        // (h_flags[8], h_flags[9] are the two slots.)  Every rank reads the same all-reduced values, so all ranks leave the
        // loop after the same chunk.
        int it = 0;
        {
            hipEvent_t ev[2] = {c->evPoll[0], c->evPoll[1]};
            int pending = -1;  // slot whose read-back has been enqueued but not yet waited for
            int slot = 0;
            while (it < cap && conv < 0) {
                const int stop = (it + every < cap) ? it + every : cap;
                for (; it < stop; it++)
                    if ((rc = launch_iter(it, it % every))) return rc;
                // (the stop is recorded by the update kernel of the NEXT iteration -- in a multi-rank run after the all-reduce
                // that merges the partial maxima -- so a solve that converges on a chunk's last iteration is seen one poll later)
                HIPCHK(c, hipMemcpyAsync(c->h_flags + 8 + slot, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipEventRecord(ev[slot], c->stream));
                if (pending >= 0) {
                    HIPCHK(c, hipEventSynchronize(ev[pending]));
                    conv = c->h_flags[8 + pending];
                }
                pending = slot;
                slot ^= 1;
            }
            if (conv < 0 && pending >= 0) {
                HIPCHK(c, hipEventSynchronize(ev[pending]));
                conv = c->h_flags[8 + pending];
            }
        }
    }
    {
        if (conv < 0) {  // cap reached: the last iteration's residual has not been merged or tested yet
            // ([rmax step](cap - 1) are adjacent: k_pcg_check and the later k_pcg_residual read the step block as well, and every rank must see the same one)
            if (c->comm && (rc = fv_allreduce_scalars(c, sc.rmax(cap - 1), 2 * NSLOT))) return rc;
            hipLaunchKernelGGL(k_pcg_check, dim3(1), dim3(64), 0, c->stream, sc, cap - 1);
            FV_READ(c, c->h_flags, c->d_flags, sizeof(int));
            FV_SYNC(c);
            conv = c->h_flags[0];
        }
    }
    *conv_out = conv;
    return FLIPV_OK;
}
