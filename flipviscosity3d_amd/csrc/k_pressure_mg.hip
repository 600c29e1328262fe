// k_pressure_mg.hip -- aggregation-multigrid preconditioner for the pressure PCG (single GPU, fp32 vectors).
//
// The reference preconditions its pressure CG with MIC(0) (pressuresolver.cpp:337-462): two sequential triangular
// sweeps, 52 iterations at 256^3.  The diagonal preconditioner of pcg_common.h parallelises but needs 303.  This file
// gives the PCG a V-cycle instead:
//   hierarchy   2x2x2 aggregates, piecewise-constant transfer P, Galerkin coarse operators P^T A P.  A is a 7-point
//               M-matrix stored as diag + three "plus" couplings per cell (pressuresolver.h:103-108); with this P the
//               coarse operator is again a 7-point M-matrix in the same storage: diag_c = sum of the aggregate's
//               diagonals + twice its internal couplings, plus_c = sum of the couplings that cross the aggregate's face.
//   cycle       V(1,1) with damped Jacobi (omega 0.9), zero initial guess, coarse correction scaled by 1.8 (plain
//               aggregation under-corrects), 16 Jacobi sweeps on the coarsest level: a symmetric positive definite
//               operator, as CG needs.
//   measured    (scipy prototype on the oracle's 256^3 bunny matrix) 13 iterations against 76 with the diagonal for a
//               random right-hand side, 13 against 96 (128^3) for a smooth one.
// Everything is matrix-free on the levels' own dense index spaces (same layout rules as the fine grid); level 0 walks the
// solver's tile list.  Coarse levels are launch-bound (a few microseconds per kernel).
// Block-decomposed runs, default: the single domain's V-cycle (MgState::global: level 0 sweeps after a halo copy of x0, ONE global coarse
// hierarchy whose first operator and right-hand side are summed over the ranks).  With flipv_params.multigrid_rank_local:
// the V-cycle is RANK-LOCAL (block-Jacobi multigrid): every rank builds the hierarchy of its own
// box of cells, with the couplings across the box's cut faces dropped, and cycles it without any exchange; the CG around it
// uses the true operator (halo exchange of p, two scalar all-reduces per iteration).  A block-diagonal SPD preconditioner:
// same fixed point, a few more iterations than the global cycle, no halo traffic per level and sweep.
// A level's Lay: I, J, K = the GLOBAL extent of the level (cells), olo/ohi = the box of cells of the rank on that level, and
// the arrays of a coarse level are allocated for just that box (ox, oy, oz = olo; one spare column / row / plane at the end,
// so that the +-1 neighbours of a box cell are zero entries of the same array, as on the fine grid).
#include "flipv_comm.h"
#include "pcg_common.h"

namespace {

// smoothing weight and over-correction of the coarse-grid term; in __constant__ memory so that flipv_params.pressure_mg_omega /
// pressure_mg_overcorrection can override them for parameter scans
__constant__ float MG_OMEGA = 0.9f;   // scan at 256^3 (omega, over -> iterations): (0.8,1.0) 38, (0.8,1.5) 25, (0.8,1.8) 24, (0.9,1.5) 22,
__constant__ float MG_OVER = 1.8f;    // (0.9,1.8) 20, (1.0,1.5) 84
__constant__ int MG_COARSEST_SWEEPS = 8;    // even (the global-memory variant ping-pongs and must end in t); flipv_params.pressure_mg_coarsest_sweeps overrides it for scans.
                                            // The count does not move the iterations (2 ... 128 sweeps: 19.35 on the 256^3 bunny, 15.6-15.8 on the 512x256x256 sheet, 16.15 at 128^3)

struct CutBox { int lo[3], hi[3]; };   // a box of cells, half-open
struct MgLevel {
    Lay L;
    float *diag, *pi, *pj, *pk;  // operator (level 0: the context's arrays)
    float *b, *x, *t;            // right-hand side, pre-smoothed iterate, residual / result (level 0: b = PCG residual)
};

static Lay coarse_lay(const Lay &F) {
    Lay C;
    C.I = (F.I + 1) / 2; C.J = (F.J + 1) / 2; C.K = (F.K + 1) / 2;
    for (int a = 0; a < 3; a++) {      // the aggregates that hold the rank's cells
        C.olo[a] = F.olo[a] >> 1;
        C.ohi[a] = ((F.ohi[a] - 1) >> 1) + 1;
    }
    C.ox = C.olo[0]; C.oy = C.olo[1]; C.oz = C.olo[2];
    C.PX = ((C.ohi[0] - C.olo[0] + 1 + 3) / 4) * 4; C.PY = C.ohi[1] - C.olo[1] + 1; C.PZ = C.ohi[2] - C.olo[2] + 1;
    C.sy = C.PX; C.sz = (long)C.PX * C.PY;
    C.n = (size_t)C.sz * C.PZ;
    C.guard = (((size_t)C.sz + (size_t)C.sy + 8) + 63) / 64 * 64;
    C.ib = C.olo[0]; C.ie = C.ohi[0]; C.jb = C.olo[1]; C.je = C.ohi[1]; C.kb = C.olo[2]; C.ke = C.ohi[2];
    return C;
}
__host__ __device__ __forceinline__ bool d_in_cells(const Lay &L, int i, int j, int k) {   // a cell of the rank on this level
    return i >= L.olo[0] && i < L.ohi[0] && j >= L.olo[1] && j < L.ohi[1] && k >= L.olo[2] && k < L.ohi[2];
}

// ---- one cell of the pre-smoothing + residual: x = omega b/d from a zero guess, t = b - A x
__device__ __forceinline__ float d_jac0(const float *__restrict__ b, const float *__restrict__ d, size_t c) {
    const float dd = d[c];
    return dd != 0.0f ? MG_OMEGA * b[c] / dd : 0.0f;
}
__device__ __forceinline__ void d_mg_pre_cell(const Lay &L, size_t c, const float *__restrict__ d, const float *__restrict__ pi,
                                              const float *__restrict__ pj, const float *__restrict__ pk, const float *__restrict__ b,
                                              float *__restrict__ x, float *__restrict__ t) {
    const float dd = d[c];
    if (dd == 0.0f) { x[c] = 0.0f; t[c] = 0.0f; return; }
    const long sy = L.sy, sz = L.sz;
    const float xc = MG_OMEGA * b[c] / dd;
    float ax = dd * xc;
    ax += pi[c] * d_jac0(b, d, c + 1) + pi[c - 1] * d_jac0(b, d, c - 1);
    ax += pj[c] * d_jac0(b, d, c + sy) + pj[c - sy] * d_jac0(b, d, c - sy);
    ax += pk[c] * d_jac0(b, d, c + sz) + pk[c - sz] * d_jac0(b, d, c - sz);
    x[c] = xc;
    t[c] = b[c] - ax;
}

// ---- one cell of the coarse correction + post-smoothing: y = x + over * xc[parent], out = y + omega (b - A y)/d
__device__ __forceinline__ float d_mg_y(const Lay &L, const Lay &C, const float *__restrict__ x, const float *__restrict__ xc, int i,
                                        int j, int k) {
    if (!d_in_cells(L, i, j, k)) return 0.0f;
    return x[gidx(L, i, j, k)] + MG_OVER * xc[gidx(C, i >> 1, j >> 1, k >> 1)];
}
__device__ __forceinline__ float d_mg_up_cell(const Lay &L, const Lay &C, int i, int j, int k, const float *__restrict__ d,
                                              const float *__restrict__ pi, const float *__restrict__ pj, const float *__restrict__ pk,
                                              const float *__restrict__ b, const float *__restrict__ x, const float *__restrict__ xc) {
    const size_t c = gidx(L, i, j, k);
    const float dd = d[c];
    if (dd == 0.0f) return 0.0f;
    const long sy = L.sy, sz = L.sz;
    const float y = d_mg_y(L, C, x, xc, i, j, k);
    float ay = dd * y;
    // a coupling is non-zero only between two cells of the level, so the parents read below exist
    const float ci = pi[c], cim = pi[c - 1], cj = pj[c], cjm = pj[c - sy], ck = pk[c], ckm = pk[c - sz];
    if (ci != 0.0f) ay += ci * d_mg_y(L, C, x, xc, i + 1, j, k);
    if (cim != 0.0f) ay += cim * d_mg_y(L, C, x, xc, i - 1, j, k);
    if (cj != 0.0f) ay += cj * d_mg_y(L, C, x, xc, i, j + 1, k);
    if (cjm != 0.0f) ay += cjm * d_mg_y(L, C, x, xc, i, j - 1, k);
    if (ck != 0.0f) ay += ck * d_mg_y(L, C, x, xc, i, j, k + 1);
    if (ckm != 0.0f) ay += ckm * d_mg_y(L, C, x, xc, i, j, k - 1);
    return y + MG_OMEGA * (b[c] - ay) / dd;
}

// ---- Galerkin coarsening: one thread per coarse cell.  hi: couplings towards cells at or beyond it are dropped -- the rank's box of cells (rank-local
// hierarchy) or the domain (global hierarchy: the rank's children carry their couplings across the cuts into the sums over the ranks)
__global__ void k_mg_coarsen(Lay F, Lay C, const float *__restrict__ df, const float *__restrict__ pif, const float *__restrict__ pjf,
                             const float *__restrict__ pkf, float *__restrict__ dc, float *__restrict__ pic, float *__restrict__ pjc,
                             float *__restrict__ pkc, int hi0, int hi1, int hi2) {
    const int I = C.ib + blockIdx.x * 64 + threadIdx.x, J = C.jb + blockIdx.y * 4 + threadIdx.y, K = blockIdx.z + C.kb;
    if (I >= C.ie || J >= C.je) return;
    const size_t cc = gidx(C, I, J, K);
    float ds = 0.0f, si = 0.0f, sj = 0.0f, sk = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int a = q & 1, b = (q >> 1) & 1, e = q >> 2;
        const int i = 2 * I + a, j = 2 * J + b, k = 2 * K + e;
        if (!d_in_cells(F, i, j, k)) continue;   // only the rank's own cells
        const size_t c = gidx(F, i, j, k);
        ds += df[c];
        // coupling to the +1 neighbour: inside the aggregate / across its face; dropped where the neighbour is not the rank's
        if (i + 1 < hi0) { if (a == 0) ds += 2.0f * pif[c]; else si += pif[c]; }
        if (j + 1 < hi1) { if (b == 0) ds += 2.0f * pjf[c]; else sj += pjf[c]; }
        if (k + 1 < hi2) { if (e == 0) ds += 2.0f * pkf[c]; else sk += pkf[c]; }
    }
    dc[cc] = ds; pic[cc] = si; pjc[cc] = sj; pkc[cc] = sk;
}

// ---- coarse levels: dense sweeps over the level's index space
__global__ void k_mg_pre(Lay L, const float *__restrict__ d, const float *__restrict__ pi, const float *__restrict__ pj,
                         const float *__restrict__ pk, const float *__restrict__ b, float *__restrict__ x, float *__restrict__ t) {
    const int i = L.ib + blockIdx.x * 64 + threadIdx.x, j = L.jb + blockIdx.y * 4 + threadIdx.y, k = blockIdx.z + L.kb;
    if (i >= L.ie || j >= L.je) return;
    d_mg_pre_cell(L, gidx(L, i, j, k), d, pi, pj, pk, b, x, t);
}
__global__ void k_mg_restrict(Lay F, Lay C, const float *__restrict__ tf, float *__restrict__ bc) {
    const int I = C.ib + blockIdx.x * 64 + threadIdx.x, J = C.jb + blockIdx.y * 4 + threadIdx.y, K = blockIdx.z + C.kb;
    if (I >= C.ie || J >= C.je) return;
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int i = 2 * I + (q & 1), j = 2 * J + ((q >> 1) & 1), k = 2 * K + (q >> 2);
        // (children outside the box the finer level's sweeps cover hold another solve's residual)
        if (d_in_cells(F, i, j, k) && i >= F.ib && i < F.ie && j >= F.jb && j < F.je && k >= F.kb && k < F.ke) s += tf[gidx(F, i, j, k)];
    }
    bc[gidx(C, I, J, K)] = s;
}
__global__ void k_mg_up(Lay L, Lay C, const float *__restrict__ d, const float *__restrict__ pi, const float *__restrict__ pj,
                        const float *__restrict__ pk, const float *__restrict__ b, const float *__restrict__ x,
                        const float *__restrict__ xc, float *__restrict__ out) {
    const int i = L.ib + blockIdx.x * 64 + threadIdx.x, j = L.jb + blockIdx.y * 4 + threadIdx.y, k = blockIdx.z + L.kb;
    if (i >= L.ie || j >= L.je) return;
    out[gidx(L, i, j, k)] = d_mg_up_cell(L, C, i, j, k, d, pi, pj, pk, b, x, xc);
}
// The tail of the hierarchy (every level of at most MG_TAIL_CELLS cells, i.e. 16^3 and coarser; measured: with 32^3 included the single workgroup is slower than the launches it saves) in ONE
// workgroup: down-sweeps, coarsest-level Jacobi, up-sweeps, separated by workgroup barriers instead of kernel
// boundaries (each of those levels is a few thousand cells: a launch costs more than its work).
constexpr int MG_MAX_TAIL = 6;
constexpr long MG_TAIL_CELLS = 18 * 18 * 18;
struct MgTail { int n; MgLevel lev[MG_MAX_TAIL]; };
__device__ __forceinline__ void d_cell_of(const Lay &L, int q, int &i, int &j, int &k) {
    const int w = L.ohi[0] - L.olo[0], h = L.ohi[1] - L.olo[1];
    i = L.olo[0] + q % w; j = L.olo[1] + (q / w) % h; k = L.olo[2] + q / (w * h);
}
__device__ __forceinline__ int d_ncells(const Lay &L) { return (L.ohi[0] - L.olo[0]) * (L.ohi[1] - L.olo[1]) * (L.ohi[2] - L.olo[2]); }
__global__ __launch_bounds__(1024) void k_mg_tail(MgTail T) {
    for (int l = 0; l + 1 < T.n; l++) {  // down
        const MgLevel &F = T.lev[l];
        const MgLevel &C = T.lev[l + 1];
        const int nf = d_ncells(F.L), nc = d_ncells(C.L);
        for (int q = threadIdx.x; q < nf; q += blockDim.x) {
            int i, j, k;
            d_cell_of(F.L, q, i, j, k);
            d_mg_pre_cell(F.L, gidx(F.L, i, j, k), F.diag, F.pi, F.pj, F.pk, F.b, F.x, F.t);
        }
        __syncthreads();
        for (int q = threadIdx.x; q < nc; q += blockDim.x) {
            int I, J, K;
            d_cell_of(C.L, q, I, J, K);
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int i = 2 * I + (e & 1), j = 2 * J + ((e >> 1) & 1), k = 2 * K + (e >> 2);
                if (d_in_cells(F.L, i, j, k)) s += F.t[gidx(F.L, i, j, k)];
            }
            C.b[gidx(C.L, I, J, K)] = s;
        }
        __syncthreads();
    }
    {   // coarsest level: Jacobi sweeps from a zero guess, result in t
        const MgLevel &B = T.lev[T.n - 1];
        const Lay &L = B.L;
        const int n = d_ncells(L);
        if (n <= 1024 && (L.ohi[0] - L.olo[0] + 2) * (L.ohi[1] - L.olo[1] + 2) * (L.ohi[2] - L.olo[2] + 2) <= 1000) {
            // at most one cell per thread: its row of the operator lives in registers, the iterate in LDS (with a zero rim),
            // so a sweep is an LDS exchange instead of a round trip through L2 (17 sweeps: 25 of the kernel's 33 us before)
            __shared__ float sx[2][1000];
            const int q = threadIdx.x;
            const int W = L.ohi[0] - L.olo[0] + 2, H = L.ohi[1] - L.olo[1] + 2;
            for (int e = q; e < 2000; e += blockDim.x) (&sx[0][0])[e] = 0.0f;
            float dd = 0.0f, ci = 0.0f, cim = 0.0f, cj = 0.0f, cjm = 0.0f, ck = 0.0f, ckm = 0.0f, bb = 0.0f;
            int li = 0;
            size_t c = 0;
            if (q < n) {
                int i, j, k;
                d_cell_of(L, q, i, j, k);
                c = gidx(L, i, j, k);
                li = (i - L.olo[0] + 1) + W * ((j - L.olo[1] + 1) + H * (k - L.olo[2] + 1));
                dd = B.diag[c]; bb = B.b[c];
                ci = B.pi[c]; cim = B.pi[c - 1]; cj = B.pj[c]; cjm = B.pj[c - L.sy]; ck = B.pk[c]; ckm = B.pk[c - L.sz];
            }
            __syncthreads();
            if (q < n) sx[0][li] = dd != 0.0f ? MG_OMEGA * bb / dd : 0.0f;
            __syncthreads();
            int cur = 0;
            for (int s = 0; s < MG_COARSEST_SWEEPS + 1; s++) {
                if (q < n) {
                    const float *x = sx[cur];
                    float v = 0.0f;
                    if (dd != 0.0f) {
                        const float ax = dd * x[li] + ci * x[li + 1] + cim * x[li - 1] + cj * x[li + W] + cjm * x[li - W] + ck * x[li + W * H] +
                                         ckm * x[li - W * H];
                        v = x[li] + MG_OMEGA * (bb - ax) / dd;
                    }
                    sx[cur ^ 1][li] = v;
                }
                __syncthreads();
                cur ^= 1;
            }
            if (q < n) B.t[c] = sx[cur][li];
            __syncthreads();
        } else {
        float *cur = B.x, *nxt = B.t;
        for (int q = threadIdx.x; q < n; q += blockDim.x) {
            int i, j, k;
            d_cell_of(L, q, i, j, k);
            const size_t c = gidx(L, i, j, k);
            cur[c] = B.diag[c] != 0.0f ? MG_OMEGA * B.b[c] / B.diag[c] : 0.0f;
        }
        __syncthreads();
        for (int s = 0; s < MG_COARSEST_SWEEPS + 1; s++) {  // odd count: the last sweep writes t
            for (int q = threadIdx.x; q < n; q += blockDim.x) {
                int i, j, k;
                d_cell_of(L, q, i, j, k);
                const size_t c = gidx(L, i, j, k);
                const float dd = B.diag[c];
                float v = 0.0f;
                if (dd != 0.0f) {
                    const float ax = dd * cur[c] + B.pi[c] * cur[c + 1] + B.pi[c - 1] * cur[c - 1] + B.pj[c] * cur[c + L.sy] +
                                     B.pj[c - L.sy] * cur[c - L.sy] + B.pk[c] * cur[c + L.sz] + B.pk[c - L.sz] * cur[c - L.sz];
                    v = cur[c] + MG_OMEGA * (B.b[c] - ax) / dd;
                }
                nxt[c] = v;
            }
            __syncthreads();
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        }
    }
    for (int l = T.n - 2; l >= 0; l--) {  // up
        const MgLevel &F = T.lev[l];
        const MgLevel &C = T.lev[l + 1];
        const int nf = d_ncells(F.L);
        for (int q = threadIdx.x; q < nf; q += blockDim.x) {
            int i, j, k;
            d_cell_of(F.L, q, i, j, k);
            F.t[gidx(F.L, i, j, k)] = d_mg_up_cell(F.L, C.L, i, j, k, F.diag, F.pi, F.pj, F.pk, F.b, F.x, C.t);
        }
        __syncthreads();
    }
}

// ---- global hierarchy (block contexts): the box of a level's cells with an operator row, and box-shaped packing for the all-reduces
__global__ __launch_bounds__(256) void k_mg_bbox(Lay L, const float *__restrict__ d, int *__restrict__ box) {
    const int i = L.ib + blockIdx.x * 64 + threadIdx.x, j = L.jb + blockIdx.y * 4 + threadIdx.y, k = blockIdx.z + L.kb;
    if (i >= L.ie || j >= L.je || d[gidx(L, i, j, k)] == 0.0f) return;
    atomicMin(box + 0, i); atomicMin(box + 1, j); atomicMin(box + 2, k);
    atomicMax(box + 3, i + 1); atomicMax(box + 4, j + 1); atomicMax(box + 5, k + 1);
}
struct Ptr4 { float *p[4]; };
// unpack = 0: arrays -> dense buffer [blockIdx.y][cells of the box], 1: buffer -> arrays
__global__ __launch_bounds__(256) void k_mg_box_pack(Lay L, CutBox B, Ptr4 arr, float *__restrict__ buf, int unpack) {
    const int w = B.hi[0] - B.lo[0], h = B.hi[1] - B.lo[1], d = B.hi[2] - B.lo[2];
    const size_t n = (size_t)w * h * d, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float *g = arr.p[blockIdx.y] + gidx(L, B.lo[0] + (int)(t % w), B.lo[1] + (int)((t / w) % h), B.lo[2] + (int)(t / ((size_t)w * h)));
    float *q = buf + (size_t)blockIdx.y * n + t;
    if (unpack) *g = *q; else *q = *g;
}

// ---- the level-0 kernels walk the solver's tile list: once per tile geometry (pcg_geo.inc)
namespace g16 {
constexpr int ROWL = 16;
#include "pcg_geo.inc"
#include "k_pressure_mg_geo.inc"
}  // namespace g16
namespace g64 {
constexpr int ROWL = 64;
#include "pcg_geo.inc"
#include "k_pressure_mg_geo.inc"
}  // namespace g64

struct MgState {
    std::vector<MgLevel> lev;
    std::vector<Lay> range;   // per level: the level's Lay with the launch box (ib..ke) of this solve's sweeps -- the cells within reach of the
                              // liquid (fv_range_liquid, halved level by level); the whole level in multi-rank runs and outside a substep
    int tailFirst = 0;  // first level handled by k_mg_tail
    float omega = 0.9f, over = 1.8f; int sweeps = 8;   // what the __constant__ scan parameters currently hold (flipv_params.pressure_mg_*)
    std::vector<void *> allocs;
    int I = 0, J = 0, K = 0;
    // Block contexts (flipv_comm.h).  global: levels 1.. are the GLOBAL hierarchy, held and cycled redundantly by every rank -- level 1's operator is the
    // sum over the ranks of their Galerkin contributions (one all-reduce per solve), its right-hand side the sum of their restricted residuals
    // (one all-reduce per iteration), and level 0's sweeps read the neighbours' x0 (one halo copy per iteration): the single domain's V-cycle.
    // Otherwise (flipv_params.multigrid_rank_local) every rank cycles the hierarchy of its own box of cells, couplings across the cuts dropped.
    bool global = false;
    float *bacc = nullptr;       // where k_mg_down0 accumulates the rank's share of level 1's right-hand side (global: a separate array; else lev[1].b)
    float *stage = nullptr;      // dense staging buffer of the all-reduces
    size_t stageCap = 0;
    int *d_bbox = nullptr;       // 6 ints
    double *d_gbox = nullptr;    // 6 doubles per rank
    CutBox gbox;                 // this solve's box of level-1 cells with a row, over all ranks
    int rc = 0;                  // a communication error inside the V-cycle
    // global hierarchy: what follows the right-hand-side all-reduce of a V-cycle is kernels only (~12 small launches): captured once per solve and
    // replayed every iteration (the executable lives in flipv_context::geCache)
    hipGraphExec_t midExec = nullptr;
    bool midReady = false;
    ~MgState() { for (void *p : allocs) (void)hipFree(p); if (stage) (void)hipFree(stage); }
};

static int mg_alloc(flipv_context *c, MgState *s, const Lay &L, float **p) {
    const size_t tot = L.n + 2 * L.guard;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, tot * sizeof(float));
    if (e != hipSuccess) { c->err = std::string("hipMalloc(multigrid level): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
    s->allocs.push_back(q);
    HIPCHK(c, hipMemsetAsync(q, 0, tot * sizeof(float), c->stream));
    *p = (float *)q + L.guard;
    return FLIPV_OK;
}

#define MGGRID(Lv) dim3(cdiv((Lv).ie - (Lv).ib, 64), cdiv((Lv).je - (Lv).jb, 4), (unsigned)((Lv).ke - (Lv).kb)), dim3(64, 4, 1)

}  // namespace

void fv_mg_free(flipv_context *c) {
    delete (MgState *)c->mgState;
    c->mgState = nullptr;
}

// sum over the ranks of `narr` arrays of a level inside s->gbox, through the dense staging buffer (`into`: where the sums go)
static int mg_allreduce_box(flipv_context *c, MgState *s, const Lay &L, const Ptr4 &arr, const Ptr4 &into, int narr) {
    const CutBox &B = s->gbox;
    const size_t n = (size_t)(B.hi[0] - B.lo[0]) * (B.hi[1] - B.lo[1]) * (B.hi[2] - B.lo[2]), tot = n * (size_t)narr;
    if (tot > s->stageCap) {
        FV_SYNC(c);
        if (s->stage) (void)hipFree(s->stage);
        s->stage = nullptr; s->stageCap = 0;
        hipError_t e = hipMalloc((void **)&s->stage, tot * sizeof(float));
        if (e != hipSuccess) { c->err = std::string("hipMalloc(pressure multigrid staging): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        s->stageCap = tot;
    }
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)narr);
    hipLaunchKernelGGL(k_mg_box_pack, grid, dim3(256), 0, c->stream, L, B, arr, s->stage, 0);
    const int rc = fv_allreduce_f32(c, s->stage, tot);
    if (rc) return rc;
    if (narr > 1) c->commBytesSetup += (double)tot * sizeof(float); else c->commBytesIter = (double)tot * sizeof(float);
    if (into.p[0]) hipLaunchKernelGGL(k_mg_box_pack, grid, dim3(256), 0, c->stream, L, B, into, s->stage, 1);   // (no target: the caller unpacks, mg_box_take)
    return FLIPV_OK;
}
static void mg_box_take(flipv_context *c, MgState *s, const Lay &L, const Ptr4 &into, int narr) {
    const CutBox &B = s->gbox;
    const size_t n = (size_t)(B.hi[0] - B.lo[0]) * (B.hi[1] - B.lo[1]) * (B.hi[2] - B.lo[2]);
    hipLaunchKernelGGL(k_mg_box_pack, dim3((unsigned)((n + 255) / 256), (unsigned)narr), dim3(256), 0, c->stream, L, B, into, s->stage, 1);
}

// The level structure (allocated once per context) and this substep's coarse operators.
static int mg_setup(flipv_context *c, MgState **out) {
    MgState *s = (MgState *)c->mgState;
    const bool wantGlobal = c->comm && !c->prm.multigrid_rank_local;
    if (s && s->global != wantGlobal) {   // (a communicator attached, or the switch flipped, after the first solve)
        FV_SYNC(c);
        fv_mg_free(c);
        s = nullptr;
    }
    if (!s) {
        s = new MgState();
        c->mgState = s;
        s->global = wantGlobal;
        MgLevel l0;
        l0.L = c->L;
        for (int a = 0; a < 3; a++) { l0.L.olo[a] = c->cell0[a]; l0.L.ohi[a] = c->cell1[a]; }   // the rank's CELLS (all of them on a single GPU)
        l0.L.ib = l0.L.olo[0]; l0.L.ie = l0.L.ohi[0]; l0.L.jb = l0.L.olo[1]; l0.L.je = l0.L.ohi[1]; l0.L.kb = l0.L.olo[2]; l0.L.ke = l0.L.ohi[2];
        l0.diag = c->pDiag; l0.pi = c->pPi; l0.pj = c->pPj; l0.pk = c->pPk;
        l0.b = (float *)c->pR;
        int rc;
        if ((rc = mg_alloc(c, s, c->L, &l0.x)) || (rc = mg_alloc(c, s, c->L, &l0.t))) return rc;
        s->lev.push_back(l0);
        HIPCHK(c, hipMalloc((void **)&s->d_bbox, 8 * sizeof(int)));
        s->allocs.push_back(s->d_bbox);
        HIPCHK(c, hipMalloc((void **)&s->d_gbox, 6 * 32 * sizeof(double)));
        s->allocs.push_back(s->d_gbox);
        Lay whole = l0.L;   // global hierarchy: the coarse levels are those of the whole domain's cells
        whole.olo[0] = whole.olo[1] = whole.olo[2] = 0; whole.ohi[0] = c->L.I; whole.ohi[1] = c->L.J; whole.ohi[2] = c->L.K;
        while (true) {
            const Lay &F = (s->global && s->lev.size() == 1) ? whole : s->lev.back().L;
            const int e3[3] = {F.ohi[0] - F.olo[0], F.ohi[1] - F.olo[1], F.ohi[2] - F.olo[2]};   // the rank's box decides the depth of ITS hierarchy
            const int m = e3[0] > e3[1] ? (e3[0] > e3[2] ? e3[0] : e3[2]) : (e3[1] > e3[2] ? e3[1] : e3[2]);
            if (m <= 8 || s->lev.size() >= 8) break;
            MgLevel l;
            l.L = coarse_lay(F);
            float **arr[7] = {&l.diag, &l.pi, &l.pj, &l.pk, &l.b, &l.x, &l.t};
            for (auto a : arr) if ((rc = mg_alloc(c, s, l.L, a))) return rc;
            s->lev.push_back(l);
        }
        if (s->lev.size() > 1) {
            s->bacc = s->lev[1].b;
            if (s->global && (rc = mg_alloc(c, s, s->lev[1].L, &s->bacc))) return rc;
        }
        // the tail: the coarsest levels that are small enough for one workgroup (at least the last one, at most MG_MAX_TAIL);
        // level 0 always runs on the tile list
        s->tailFirst = (int)s->lev.size() - 1;
        while (s->tailFirst > 1 && (int)s->lev.size() - (s->tailFirst - 1) <= MG_MAX_TAIL &&
               (long)(s->lev[s->tailFirst - 1].L.ohi[0] - s->lev[s->tailFirst - 1].L.olo[0]) * (s->lev[s->tailFirst - 1].L.ohi[1] - s->lev[s->tailFirst - 1].L.olo[1]) *
                       (s->lev[s->tailFirst - 1].L.ohi[2] - s->lev[s->tailFirst - 1].L.olo[2]) <= MG_TAIL_CELLS)
            s->tailFirst--;
        if (s->lev.size() == 1) s->tailFirst = 0;
    }
    {   // flipv_params.pressure_mg_*: scan parameters in __constant__ memory (per device, i.e. shared by the contexts of one device)
        const float om = c->prm.pressure_mg_omega > 0.0f ? c->prm.pressure_mg_omega : 0.9f, ov = c->prm.pressure_mg_overcorrection > 0.0f ? c->prm.pressure_mg_overcorrection : 1.8f;
        const int sw = c->prm.pressure_mg_coarsest_sweeps > 0 ? (c->prm.pressure_mg_coarsest_sweeps + 1) / 2 * 2 : 8;
        if (om != s->omega || ov != s->over || sw != s->sweeps) {
            FV_SYNC(c);
            HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(MG_OMEGA), &om, sizeof(float)));
            HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(MG_OVER), &ov, sizeof(float)));
            HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(MG_COARSEST_SWEEPS), &sw, sizeof(int)));
            s->omega = om; s->over = ov; s->sweeps = sw;
        }
    }
    for (size_t l = 0; l + 1 < s->lev.size(); l++) {
        const MgLevel &F = s->lev[l];
        const MgLevel &C = s->lev[l + 1];
        const bool cross = s->global && l == 0;   // the rank's children with their couplings across the cuts
        hipLaunchKernelGGL(k_mg_coarsen, MGGRID(C.L), 0, c->stream, F.L, C.L, F.diag, F.pi, F.pj, F.pk, C.diag, C.pi, C.pj, C.pk,
                           cross ? c->L.I : F.L.ohi[0], cross ? c->L.J : F.L.ohi[1], cross ? c->L.K : F.L.ohi[2]);   // (the WHOLE level: coefficients are fresh everywhere)
        if (cross) {   // level 1's operator = the sum over the ranks, inside the union of their boxes of rows
            int rc;
            const int big = 0x7fffffff;
            const int init[6] = {big, big, big, 0, 0, 0};
            HIPCHK(c, hipMemcpyAsync(s->d_bbox, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_mg_bbox, MGGRID(C.L), 0, c->stream, C.L, C.diag, s->d_bbox);
            int hb[6];
            FV_READ(c, hb, s->d_bbox, sizeof(hb));
            FV_SYNC(c);
            const int nr = c->comm->nranks, me = c->comm->rank;
            std::vector<double> hbx((size_t)6 * nr, 0.0);
            const bool any = hb[3] > hb[0];
            for (int a = 0; a < 3; a++) { hbx[(size_t)6 * me + a] = any ? hb[a] : big; hbx[(size_t)6 * me + 3 + a] = any ? hb[3 + a] : 0; }
            HIPCHK(c, hipMemcpyAsync(s->d_gbox, hbx.data(), hbx.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            if ((rc = fv_allreduce_scalars(c, s->d_gbox, hbx.size()))) return rc;
            FV_READ(c, hbx.data(), s->d_gbox, hbx.size() * sizeof(double));
            FV_SYNC(c);
            CutBox g = {{big, big, big}, {0, 0, 0}};
            for (int r = 0; r < nr; r++)
                for (int a = 0; a < 3; a++) {
                    const int lo = (int)hbx[(size_t)6 * r + a], hi = (int)hbx[(size_t)6 * r + 3 + a];
                    if (lo < g.lo[a]) g.lo[a] = lo;
                    if (hi > g.hi[a]) g.hi[a] = hi;
                }
            for (int a = 0; a < 3; a++) if (g.hi[a] <= g.lo[a]) { g.lo[a] = 0; g.hi[a] = 1; }   // no pressure cell anywhere
            s->gbox = g;
            const Ptr4 op = {{C.diag, C.pi, C.pj, C.pk}};
            if ((rc = mg_allreduce_box(c, s, C.L, op, op, 4))) return rc;
        }
    }
    // Where this solve's sweeps run: the cells within reach of the liquid, halved level by level (the whole level when the liquid's
    // box is not known: multi-rank runs, operators called outside a substep).  Outside the box the vectors of a level keep whatever
    // an earlier solve left there; nothing reads it: a coupling is non-zero only between two pressure cells (both inside the box),
    // and the restriction skips children outside the finer level's box.
    {
        const Lay R = fv_range_liquid(c, 1, 5);
        s->range.resize(s->lev.size());
        int lo[3] = {R.ib, R.jb, R.kb}, hi[3] = {R.ie, R.je, R.ke};
        for (size_t l = 0; l < s->lev.size(); l++) {
            Lay Lr = s->lev[l].L;
            int *b[3] = {&Lr.ib, &Lr.jb, &Lr.kb}, *e[3] = {&Lr.ie, &Lr.je, &Lr.ke};
            for (int a = 0; a < 3; a++) {
                if (s->global && l == 1) { lo[a] = s->gbox.lo[a]; hi[a] = s->gbox.hi[a]; }   // (the global levels: the box of level 1's rows over all ranks, halved level by level)
                else if (l > 0) { lo[a] = lo[a] >> 1; hi[a] = ((hi[a] - 1) >> 1) + 1; }
                const int l0 = lo[a] > Lr.olo[a] ? lo[a] : Lr.olo[a], h0 = hi[a] < Lr.ohi[a] ? hi[a] : Lr.ohi[a];
                *b[a] = l0; *e[a] = h0 > l0 ? h0 : l0 + 1;
            }
            s->range[l] = Lr;
        }
    }
    HIPCHK(c, hipGetLastError());
    s->midReady = false;   // (this solve's tile list, launch boxes and staging buffer: the V-cycle's captured segment is cut anew)
    *out = s;
    return FLIPV_OK;
}

// z = M^-1 r into level 0's t, (r, z) accumulated into sig(it_next)
static void mg_vcycle(flipv_context *c, MgState *s, const PcgScal &sc, int it_next) {   // it_next = IT_DEVICE: device-side counter + 1
    const int nl = (int)s->lev.size();
    const int nb = pcg_grid(c, c->nActiveP);
    const int t0 = s->tailFirst;  // levels [t0, nl) run inside k_mg_tail
    CutBox cut;   // level 0: couplings across these faces are dropped (the rank's cells; the whole domain under the global hierarchy)
    for (int a = 0; a < 3; a++) { cut.lo[a] = s->global ? 0 : s->lev[0].L.olo[a]; cut.hi[a] = s->global ? (a == 0 ? c->L.I : (a == 1 ? c->L.J : c->L.K)) : s->lev[0].L.ohi[a]; }
    const bool sums = s->global && t0 > 0;   // level 1's right-hand side is summed over the ranks
    if (t0 > 0) {  // level 0 -> right-hand side of level 1: x0 is in F.x already (k_mgp_xr); the residual goes straight into the coarse right-hand side
        const MgLevel &F = s->lev[0];
        const MgLevel &C = s->lev[1];
        // (bacc: zero on entry to the solve, then k_mg_up0 clears what k_mg_down0 filled)
        if (s->global) {   // the neighbours' x0 on the halo entries: level 0's sweeps are the single domain's
            const HaloArray hx[1] = {{F.x, sizeof(float)}};
            if ((s->rc = fv_halo_copy(c, hx, 1, 1))) return;
        }
        GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mg_down0, dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListP, c->nActiveP, c->tgP, F.L, C.L, cut, F.diag, F.pi,
                           F.pj, F.pk, c->pMask, F.x, F.b, s->bacc));
        if (sums) {   // level 1's right-hand side = the sum of the ranks' shares (bacc stays the rank's own: k_mg_up0 clears exactly what it filled)
            const Ptr4 from = {{s->bacc, nullptr, nullptr, nullptr}}, none = {{nullptr, nullptr, nullptr, nullptr}};
            if ((s->rc = mg_allreduce_box(c, s, C.L, from, none, 1))) return;
        }
    }
    // everything after the all-reduce: kernels only
    auto rest = [&]() {
        if (sums) { const Ptr4 to = {{s->lev[1].b, nullptr, nullptr, nullptr}}; mg_box_take(c, s, s->lev[1].L, to, 1); }
        for (int l = 1; l < t0; l++) {  // down: level l -> right-hand side of level l+1
            const MgLevel &F = s->lev[l];
            const MgLevel &C = s->lev[l + 1];
            // (a fused sweep, one thread per coarse cell walking its eight children, measured 23 us against 14 for the pair)
            const Lay &Fr = s->range[l];
            const Lay &Cr = l + 1 < t0 ? s->range[l + 1] : C.L;   // the first level of the tail is swept whole: its right-hand side is written everywhere
            hipLaunchKernelGGL(k_mg_pre, MGGRID(Fr), 0, c->stream, Fr, F.diag, F.pi, F.pj, F.pk, F.b, F.x, F.t);
            hipLaunchKernelGGL(k_mg_restrict, MGGRID(Cr), 0, c->stream, Fr, Cr, F.t, C.b);
        }
        MgTail T;
        T.n = nl - t0;
        for (int l = t0; l < nl; l++) T.lev[l - t0] = s->lev[l];
        hipLaunchKernelGGL(k_mg_tail, dim3(1), dim3(1024), 0, c->stream, T);
        for (int l = t0 - 1; l >= 0; l--) {
            const MgLevel &F = s->lev[l];
            const MgLevel &C = s->lev[l + 1];
            if (l == 0)
                GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mg_up0, dim3(nb), dim3(64, 4, 1), 0, c->stream, c->tileListP, c->nActiveP, c->tgP, F.L, C.L, cut, F.diag, F.pi,
                                   F.pj, F.pk, c->pMask, F.x, F.b, C.t, F.t, s->bacc, sc, it_next));
            else
                hipLaunchKernelGGL(k_mg_up, MGGRID(s->range[l]), 0, c->stream, s->range[l], C.L, F.diag, F.pi, F.pj, F.pk, F.b, F.x, C.t, F.t);
        }
    };
    // replayed only where the iteration number is the device-side counter (the loop under a communicator passes IT_DEVICE from its first iteration on)
    const bool replay = sums && it_next == IT_DEVICE && !c->prm.kernel_timing && !c->prm.no_graph_replay;
    if (replay && !s->midReady) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            rest();
            const hipError_t e2 = hipStreamEndCapture(c->stream, &g);
            s->midReady = e2 == hipSuccess && g && fv_graph_exec(c, FV_GE_PRESSURE_MID, g, &s->midExec) == FLIPV_OK;
            if (!s->midReady) { s->midExec = nullptr; (void)hipGetLastError(); }
            if (g) (void)hipGraphDestroy(g);
        } else (void)hipGetLastError();
    }
    if (replay && s->midReady) { if (hipGraphLaunch(s->midExec, c->stream) != hipSuccess) { s->rc = FLIPV_ERR_HIP; c->err = "pressure multigrid: hipGraphLaunch of the coarse segment failed"; } }
    else rest();
}

// PCG with the V-cycle as preconditioner.  On entry the setup kernel has left r = b, x = 0 and the tile list; the
// scalars' slot blocks are zero.  spmv(it) must compute q = A p with a(it) = p.q (k_pressure_spmv does).
int fv_pressure_pcg_mg(flipv_context *c, const PcgScal &sc, int cap, void (*spmv)(flipv_context *, const PcgScal &, int), int *conv_out) {
    MgState *s = nullptr;
    int rc = mg_setup(c, &s);
    if (rc) return rc;
    const int nb = pcg_grid(c, c->nActiveP);
    const dim3 blk(64, 4, 1);
    float *x = c->pressure, *r = (float *)c->pR, *q = (float *)c->pZ, *p = (float *)c->pS, *z = s->lev[0].t;
    float *x0 = s->lev[0].x;
    const HaloArray ph[1] = {{p, sizeof(float)}};
    if (s->tailFirst > 0) HIPCHK(c, hipMemsetAsync(s->bacc, 0, s->lev[1].L.n * sizeof(float), c->stream));   // once per solve; every cycle leaves it cleared (k_mg_up0)
    GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mgp_xr, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, c->L, c->pDiag, x, r, p, q, x0, sc, -1));
    mg_vcycle(c, s, sc, 0);
    if (s->rc) return s->rc;
    if ((rc = fv_allreduce_scalars(c, sc.sig(0), NSLOT))) return rc;
    GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mgp_p, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, c->L, c->pDiag, z, p, sc, -1));
    // an iteration after the stop is a full V-cycle (plus, multi-rank, three exchanges): poll often
    const int every = c->prm.check_every > 0 ? c->prm.check_every : 4;
    int conv = -1, it = 0;
    // One GPU: `every` iterations (16 kernels each: most of them a few microseconds of work on a coarse level) + the read-back
    // of the stop flag are captured once into a hipGraph and replayed -- the loop was bound by launch overhead.
    const bool graph = !c->comm && !c->prm.kernel_timing && !c->prm.no_graph_replay;
    if (graph) {
        HIPCHK(c, hipMemsetAsync(sc.itA, 0, 2 * sizeof(int), c->stream));
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        for (int e = 0; e < every; e++) {
            spmv(c, sc, -1);
            GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mgp_xr, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, c->L, c->pDiag, x, r, p, q, x0, sc, IT_DEVICE));
            mg_vcycle(c, s, sc, IT_DEVICE);
            GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mgp_p, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, c->L, c->pDiag, z, p, sc, IT_DEVICE));
        }
        const int e1 = fv_read_capture(c, c->d_flags, 1);   // the stop flag, published to the host at the end of every replay
        hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        if (e1 != FLIPV_OK || e2 != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); c->err = "multigrid PCG: stream capture failed"; return FLIPV_ERR_HIP; }
        if ((rc = fv_graph_exec(c, FV_GE_PRESSURE_MG, g, &ge))) { (void)hipGraphDestroy(g); return rc; }
        if ((rc = fv_read_wait(c))) { (void)hipGraphDestroy(g); return rc; }
        for (; it < cap && conv < 0; it += every) {
            hipError_t el = hipGraphLaunch(ge, c->stream);
            fv_read_replayed(c);
            if (el != hipSuccess || fv_read_wait(c) != FLIPV_OK) { (void)hipGraphDestroy(g); c->err = "hipGraphLaunch failed"; return FLIPV_ERR_HIP; }
            conv = c->h_pub[FV_PUB_REPLAY];
        }
        (void)hipGraphDestroy(g);
        // (cap reached: the stop test of the last iteration ran inside k_mgp_p already, and the last replay published its flag)
        HIPCHK(c, hipGetLastError());
        *conv_out = conv;
        return FLIPV_OK;
    }
    // Under a communicator the kernels read the DEVICE-side iteration counter too (the host's `it` only addresses the slot blocks its all-reduces sum:
    // the two agree as long as the solve runs, and nothing is looked at once it has stopped), so that the V-cycle's kernel-only segment can be a
    // replayed graph (mg_vcycle).  Per-launch event timing keeps the explicit iteration numbers.
    const bool devIt = c->comm && !c->prm.kernel_timing && !c->prm.no_graph_replay;
    if (devIt) HIPCHK(c, hipMemsetAsync(sc.itA, 0, 2 * sizeof(int), c->stream));
    while (it < cap && conv < 0) {
        const int stop = it + every < cap ? it + every : cap;
        for (; it < stop; it++) {
            const long ex0 = c->nExchanges, ar0 = c->nAllReduces;
            if ((rc = fv_halo_copy(c, ph, 1, 1))) return rc;                                   // p on the neighbours' boundary planes
            spmv(c, sc, devIt ? -1 : it);
            if (c->comm && (rc = fv_allreduce_scalars(c, sc.a(it), NSLOT))) return rc;          // p.q
            GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mgp_xr, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, c->L, c->pDiag, x, r, p, q, x0, sc, devIt ? IT_DEVICE : it));
            mg_vcycle(c, s, sc, devIt ? IT_DEVICE : it + 1);
            if (s->rc) return s->rc;
            if (c->comm && (rc = fv_allreduce_scalars(c, sc.rmax(it), 3 * NSLOT))) return rc;   // max|r| of this iteration, (the unused step slot,) (r,z) of the next
            GEO_RUN(c->tgP.rowl, hipLaunchKernelGGL(k_mgp_p, dim3(nb), blk, 0, c->stream, c->tileListP, c->nActiveP, c->tgP, c->L, c->pDiag, z, p, sc, devIt ? IT_DEVICE : it));
            c->exchIter = (int)(c->nExchanges - ex0); c->allrIter = (int)(c->nAllReduces - ar0);
        }
        if ((rc = fv_read_now(c, c->h_flags, c->d_flags, 1))) return rc;
        conv = c->h_flags[0];
    }
    HIPCHK(c, hipGetLastError());
    *conv_out = conv;
    return FLIPV_OK;
}
