// k_viscosity_mg.hip -- Galerkin multigrid preconditioner for the variational viscosity PCG (fp32 vectors; one GPU or block contexts).
//
// The viscosity system (viscositysolver.cpp:276-664) couples the three face-velocity components through the shear
// stresses; its rigid modes make piecewise-constant coarse spaces useless and only Galerkin coarse operators survive the
// near-empty control volumes at the free surface.  What works (scipy prototype on the oracle's matrices,
// tests/research/vmg_proto.py: 65 PCG iterations at 64^3, 78 at 128^3, 139-151 at 256^3 with V(2,2), against 440 / 1490 / 2071 with the
// diagonal; what limits it is in DESIGN.md 8.1):
//   transfer P   per component, on the MAC lattices: linear along the face normal (a fine face on a coarse face plane
//                takes that coarse face, one between two planes the mean of both), piecewise constant across
//   coarse A     P^T A P.  With this P the coarse operator of every level has the SAME 23-entry pattern per row: 15
//                same-component neighbours (normal offset -1/0/+1 times {centre, +-1 along either transverse axis}) and 4
//                + 4 cross-component neighbours -- the fine 15-point pattern plus the normal-times-transverse diagonals.
//                Storage: one coefficient grid per (component, slot) on the level's dense index space ("dense slots").
//                The cycle's kernels read a packed copy: 22 fp16 off-diagonal entries under a power-of-two scale + an fp32 diagonal (d_row_dot).
//   level 0      matrix-free: the SpMV kernel of the solve's layout -- k_bvisc_spmv<.., EPI> on bricks (k_viscosity_brick.hip), k_visc_spmv<.., EPI> on
//                planes (k_viscosity.hip) -- with the Jacobi update / the residual as its epilogue: a sweep is ONE launch that reads the iterate with
//                its halo and writes the next one
//   assembly     Galerkin products as GATHERS, one thread per coarse row: level 1 from the matrix-free fine rows (k_vmg_rap_gather_fine: a coarse row
//                collects w_child A(child, q) w_parent from its <= 12 children; parities and slots are compile-time constants, the 23
//                accumulators registers), level l+1 from level l (k_vmg_rap_gather).  Every entry of a level's box is written, nothing is zeroed
//   cycle        V(2,2), zero initial guess; the two Jacobi sweeps of a pair use the Chebyshev weights (1.317, 0.382) of [3/8, 3] (lambda_max(D^-1 A) ~ 3;
//                VMG_W_DEFAULT); the hierarchy stops at 16^3, which is solved in LDS by 16 or 32 Chebyshev-weighted sweeps on [lambda_hi / 100, lambda_hi]
//                (VMG_CHEB_KAPPA; lambda_hi from the level's Gershgorin bound) -- what 64 / 256 plain sweeps did (deeper levels bought nothing in the
//                prototype, tests/research/vmg_proto.py)
//   where        every coarse sweep covers only the strips (64 consecutive i of a (j, k) row) that hold rows, inside the box of the
//                level's index space that the listed tiles reach; levels whose box holds <= VMG_TAIL_POS positions run inside
//                ONE single-workgroup launch (k_vmg_coarsest when that is just the LDS-resident coarsest level, k_vmg_tail otherwise), the larger ones
//                as 5 launches per level
//   who          flipv_params.viscosity_preconditioner: MULTIGRID always; AUTO (default) on every system with nu dt/dx^2 > 64 unless the previous
//                solve shows the diagonal to converge for less (k_viscosity.hip: fv_visc_auto_pick)
//   ranks        block contexts (flipv_comm.h): the same preconditioner -- fine-level sweeps after a halo copy of their input, ONE global coarse hierarchy
//                (operator summed over the ranks per solve, first coarse right-hand side per iteration as the level's listed bricks, VmgState::globalFrom) cycled redundantly by every rank; or
//                rank-local block-Jacobi (flipv_params.multigrid_rank_local)
//   loop         PCG around it; `check_every` iterations are captured into a hipGraph once per solve and replayed
//                (device-side iteration counters, as in pcg_common.h)
#include "flipv_internal.h"
#include "pcg_common.h"
#include "brick.h"
#include "flipv_comm.h"

#include <chrono>
#include <vector>

void fv_visc_sweep_f32(flipv_context *c, float *const in[3], float *const out[3], int epi, const PcgScal &sc, int it_arg, float omega, int sig_shift);  // k_viscosity.hip (tile kernels or, in the brick layout, k_viscosity_brick.hip)

namespace {

constexpr int VS = 23;             // slots per row
constexpr int VH = 11;             // 32-bit words of a row's packed off-diagonal entries (two fp16 per word; see d_row_dot)
constexpr float VMG_OMEGA = 0.6f;           // the coarsest level's sweeps
// damping of the first / second sweep of the V(2,2) smoother on every level above the coarsest (VLevelDev::w; flipv_params.viscosity_mg_omega_*)
constexpr float VMG_W_DEFAULT[2] = {1.317f, 0.382f};   // the roots of the degree-2 Chebyshev polynomial on [3/8, 3]: |p| <= 0.43 there, < 1 up to lambda = 3.38
                                                       // (0.6, 0.6: < 1 up to 3.33); bunny 256^3, 12 substeps: 1476 -> 1335 iterations
// The LDS-resident coarsest level is swept with Chebyshev weights: sweep k of m uses omega_k = 1/lambda_k, lambda_k the roots of the degree-m
// Chebyshev polynomial on [lambda_hi / VMG_CHEB_KAPPA, lambda_hi] (Richardson's form of the semi-iteration: the same polynomial, no third
// vector), taken in the Lebedev-Finogenov order that keeps the partial products bounded.  lambda_hi = min(Gershgorin bound of the level's
// D^-1 A, VMG_LAMBDA_MAX -- what the fixed weights of the other levels assume anyway).  A polynomial in D^-1 A: the cycle stays symmetric.
// Measured (bunny 256^3, 12 substeps from rest): 64 Jacobi sweeps 1476 iterations at 341 us, 256 sweeps 1363 at 532 us
// (a sweep is ~1 us: 69 LDS reads per thread, bound by one CU's LDS bandwidth).
// (the table cos(pi (2 j_k + 1) / (2 m)), k = 0..m-1, travels in the coarsest level's descriptor: VLevelDev::cheb)
constexpr float VMG_LAMBDA_MAX = 3.3f;
#ifndef VMG_KAPPA
#define VMG_KAPPA 100.0f
#endif
constexpr float VMG_CHEB_KAPPA = VMG_KAPPA;
constexpr int VMG_COARSEST_SWEEPS = 16;   // even: the sweeps ping-pong between x and y and must end in x
constexpr int VMG_MIN_DIM = 16;           // no level below this many cells along the longest axis
constexpr int VMG_TAIL_POS = 640;         // levels with at most this many index positions in their box go into the single-workgroup tail
constexpr int VMG_TAIL_MAX = 3;           // ... at most this many levels
constexpr int VMG_MAX_LEVELS = 15;
constexpr int VMG_NO_PROGRESS = 64;         // iterations without a 10 % gain on max|r| after which the multigrid loop gives up (d_vmg_stop_test)

// slot tables: neighbour component and offset of slot s of a row of component c; inverse look-up by (c, c', offset)
struct SlotTables {
    signed char comp[3][VS];
    signed char off[3][VS][3];
    signed char lut[3][3][27];     // [c][c'][(dz+1)*9 + (dy+1)*3 + (dx+1)] -> slot or -1
    signed char diag[3];           // slot of (c, 0,0,0)
};
__constant__ SlotTables ST;

static void build_slot_tables(SlotTables *T) {
    memset(T->lut, -1, sizeof(T->lut));
    for (int c = 0; c < 3; c++) {
        const int n = c, t1 = (c + 1) % 3, t2 = (c + 2) % 3;
        int s = 0;
        auto put = [&](int c2, int d0, int d1, int d2) {
            T->comp[c][s] = (signed char)c2;
            T->off[c][s][0] = (signed char)d0; T->off[c][s][1] = (signed char)d1; T->off[c][s][2] = (signed char)d2;
            T->lut[c][c2][(d2 + 1) * 9 + (d1 + 1) * 3 + (d0 + 1)] = (signed char)s;
            if (c2 == c && d0 == 0 && d1 == 0 && d2 == 0) T->diag[c] = (signed char)s;
            s++;
        };
        const int tr[5][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}};
        for (int dn = -1; dn <= 1; dn++)
            for (int q = 0; q < 5; q++) {
                int d[3] = {0, 0, 0};
                d[n] = dn; d[t1] = tr[q][0]; d[t2] = tr[q][1];
                put(c, d[0], d[1], d[2]);
            }
        for (int a : {t1, t2})  // cross component a: -1/0 along the row's normal, 0/+1 along the column's normal
            for (int dn = -1; dn <= 0; dn++)
                for (int da = 0; da <= 1; da++) {
                    int d[3] = {0, 0, 0};
                    d[n] = dn; d[a] = da;
                    put(a, d[0], d[1], d[2]);
                }
    }
}

// The same tables as compile-time functions: inside fully unrolled loops the neighbour component and offset of a slot fold to
// constants, so a row's 23 neighbour addresses are immediate offsets and no pointer array is indexed at run time (which would
// push the kernel arguments into scratch memory).
constexpr int slot_comp(int c, int s) { return s < 15 ? c : (s < 19 ? (c + 1) % 3 : (c + 2) % 3); }
constexpr int slot_off(int c, int s, int a) {   // offset of slot s of a row of component c along axis a
    const int n = c, t1 = (c + 1) % 3, t2 = (c + 2) % 3;
    if (s < 15) {
        const int dn = s / 5 - 1, q = s % 5;
        const int tr0 = q == 1 ? -1 : (q == 2 ? 1 : 0), tr1 = q == 3 ? -1 : (q == 4 ? 1 : 0);
        return a == n ? dn : (a == t1 ? tr0 : tr1);
    }
    const int x = s < 19 ? t1 : t2, idx = s < 19 ? s - 15 : s - 19;
    const int dn = idx / 2 - 1, da = idx % 2;
    (void)t2;
    return a == n ? dn : (a == x ? da : 0);
}
constexpr int slot_diag(int c) { (void)c; return 5; }   // (c, 0, 0, 0): dn = 0, q = 0

struct Box3 { int lo[3], hi[3]; };   // index box of a level, half-open
struct Vec3p { float *p[3]; };
struct VLevel {          // a coarse level (>= 1)
    Lay L;
    float *coef[3][VS];
    unsigned *ch[3][VH];     // the off-diagonal entries again, packed for the cycle's kernels (k_vmg_pack)
    float *dd[3];            // ... and the diagonal that goes with them
    float *x[3], *y[3], *b[3], *t[3];
    Box3 box;
    int *strips = nullptr, *stripFlag = nullptr;   // capacity: the strips of the level's whole index space
    int nstrips = 0;
    size_t per = 0;          // floats between consecutive grids of the level (coef[m][q + 1] - coef[m][q], b[m + 1] - b[m])
};
struct VLevelDev {       // what kernels need of a coarse level
    Lay L;
    float *coef[3][VS];
    const unsigned *ch[3][VH];
    const float *dd[3];
    int packed;           // 1: the cycle reads ch / dd; 0: the fp32 grids (flipv_context::vmgPackedRows)
    Vec3p x, y, b, t;
    Box3 box;
    // "strips" (the name is from the first version: 64 consecutive i of a row): the BRICKS (8 x 4 x 2 indices, cidx) of the box that hold
    // at least one row of the operator, numbered inside the box's brick range; the sweeps of a level visit only these
    const int *strips;
    int nstrips;
    // the coarsest level: per component the rows (as box positions), at most 1024 each, for the LDS-resident sweeps; rowcnt[3] = 1
    // when box and rows fit
    const int *rowlist;   // [3][1024]
    const int *rowcnt;    // [4]
    float w[2];           // damping of the first / second sweep of a smoothing pair (every level of a solve carries the same pair)
    float cheb[64];       // the coarsest level's Chebyshev table (see VMG_CHEB_KAPPA)
};
struct FineOp {          // the matrix-free level 0 (k_viscosity.hip's arrays)
    int swz;             // 1: vm is stored in the swizzled plane layout (sidx); the factors and the mask never are
    int brick;           // 1: everything (vm, factors, mask) is stored in the brick layout (bidx over LB)
    Lay LB;
    const float *vm[3];
    const float *fC, *fE[3];
    const uint8_t *mask;
};

// Coarse levels are stored in BRICKS of 8 x 4 x 2 indices (64 entries = two 128-byte lines): the liquid fills 10-30 % of a level's
// bounding box, and with plain rows a wave's 64 consecutive i of a (j, k) row carried rows in a third of its lanes while every
// array was fetched in whole lines (level 1 of the 256^3 bunny: 80 MB per sweep for 25 MB of operator).  A wave now owns one brick
// -- a compact 8 x 4 x 2 piece of space, far fuller wherever the liquid is at all -- and its own-index accesses are 256 contiguous
// bytes.  One brick of padding on every side: indices -1 and PX..PX+7 are addressable, nothing needs a guard zone.
// For a coarse level Lay::sy / Lay::sz hold the BRICK strides (bricks per row, bricks per plane), not index strides.
// (cidx itself: flipv_internal.h)
static Lay coarse_lay(const Lay &F) {
    Lay C;
    C.I = (F.I + 1) / 2; C.J = (F.J + 1) / 2; C.K = (F.K + 1) / 2;
    C.PX = C.I + 2; C.PY = C.J + 2; C.PZ = C.K + 2;                         // indices 0..I / J / K, one more for the stencils' reach
    const long nbx = (C.PX + 8 + 7) / 8 + 1, nby = (C.PY + 4 + 3) / 4 + 1, nbz = (C.PZ + 2 + 1) / 2 + 1;   // with the padding bricks
    C.sy = nbx; C.sz = nbx * nby;
    C.n = (size_t)(nbx * nby * nbz) * 64;
    C.guard = 64;
    C.ox = C.oy = C.oz = 0;   // single-domain hierarchy: every level's box is the level's whole index space
    C.ib = 0; C.ie = C.PX; C.jb = 0; C.je = C.PY; C.kb = 0; C.ke = C.PZ;
    C.olo[0] = C.olo[1] = C.olo[2] = 0; C.ohi[0] = C.PX; C.ohi[1] = C.PY; C.ohi[2] = C.PZ;
    return C;
}

// lattice extents of component c on a level: one more face than cells along the normal
__device__ __forceinline__ bool d_in_lattice(const Lay &L, int c, const int p[3]) {
    const int ext[3] = {L.I + (c == 0), L.J + (c == 1), L.K + (c == 2)};
    return p[0] >= 0 && p[1] >= 0 && p[2] >= 0 && p[0] < ext[0] && p[1] < ext[1] && p[2] < ext[2];
}
// parents of fine dof (c, p) in the next coarser lattice: 1 or 2, weights in w
__device__ __forceinline__ int d_parents(int c, const int p[3], int P[2][3], float w[2]) {
#pragma unroll
    for (int a = 0; a < 3; a++) { P[0][a] = p[a] >> 1; P[1][a] = p[a] >> 1; }
    if (p[c] & 1) {
        P[0][c] = (p[c] - 1) >> 1; P[1][c] = (p[c] + 1) >> 1;
        w[0] = 0.5f; w[1] = 0.5f;
        return 2;
    }
    w[0] = 1.0f; w[1] = 0.0f;
    return 1;
}
// slot of column (c2, I + (dx, dy, dz)) in a row of component c, or -1: the inverse of slot_comp / slot_off in arithmetic (a table in
// __constant__ memory indexed with computed offsets is a dependent load per scattered contribution: 60 per fine row)
constexpr __host__ __device__ __forceinline__ int d_slot_of(int c, int c2, int dx, int dy, int dz) {
    const int t1 = c == 2 ? 0 : c + 1, t2 = c == 0 ? 2 : c - 1;           // (c + 1) % 3, (c + 2) % 3
    const int dn = c == 0 ? dx : (c == 1 ? dy : dz);
    const int e1 = t1 == 0 ? dx : (t1 == 1 ? dy : dz), e2 = t2 == 0 ? dx : (t2 == 1 ? dy : dz);
    if (dn < -1 || dn > 1) return -1;
    if (c2 == c) {
        if (e1 != 0 && e2 != 0) return -1;
        const int q = e2 == 0 ? (e1 == 0 ? 0 : (e1 == -1 ? 1 : (e1 == 1 ? 2 : -1))) : (e2 == -1 ? 3 : (e2 == 1 ? 4 : -1));
        return q < 0 ? -1 : (dn + 1) * 5 + q;
    }
    const bool first = c2 == t1;
    const int da = first ? e1 : e2, other = first ? e2 : e1;
    if (other != 0 || dn > 0 || da < 0 || da > 1) return -1;
    return 15 + (first ? 0 : 4) + (dn + 1) * 2 + da;
}
// thread -> index of a level's box (64 x 4 x 1 threads per block)
#define BOX_IJK_OR_RETURN(B)                                                                                   \
    const int i = (B).lo[0] + blockIdx.x * 64 + threadIdx.x, j = (B).lo[1] + blockIdx.y * 4 + threadIdx.y,     \
              k = (B).lo[2] + blockIdx.z;                                                                      \
    if (i >= (B).hi[0] || j >= (B).hi[1]) return

// ---- level l -> level l+1 as a GATHER (no atomics, no divergence): coarse row (C, I) collects, from its <= 12 children p = 2 I + delta
// (delta along the normal -1/0/+1 with weights 1/2, 1, 1/2; 0/1 across) and each child's 23 stored entries, w_child * A(p, q) * w_parent
// into the slot of q's parent(s).  Everything but the values is known at compile time once the loops are unrolled -- the parity of q
// along its own normal decides between one parent and two, the parent's offset from I picks the slot -- so the 23 accumulators are
// registers.  (The scatter with atomics this replaced took 0.26-0.36 ms per level against 20-40 us.)
constexpr int floor_half(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }
template <int C, int DN, int A_, int B_, int T>
__device__ __forceinline__ void d_rap_gather_term(float v, float (&acc)[VS]) {
    constexpr int t1 = (C + 1) % 3, t2 = (C + 2) % 3;
    constexpr int c2 = slot_comp(C, T);
    // q - 2 I per axis
    constexpr int e0 = (C == 0 ? DN : (t1 == 0 ? A_ : B_)) + slot_off(C, T, 0);
    constexpr int e1 = (C == 1 ? DN : (t1 == 1 ? A_ : B_)) + slot_off(C, T, 1);
    constexpr int e2 = (C == 2 ? DN : (t1 == 2 ? A_ : B_)) + slot_off(C, T, 2);
    (void)t2;
    constexpr int en = c2 == 0 ? e0 : (c2 == 1 ? e1 : e2);          // along the column's own normal
    constexpr bool two = (en & 1) != 0;
    constexpr float wc = DN == 0 ? 1.0f : 0.5f;
    // first (or only) parent
    constexpr int p0 = c2 == 0 && two ? floor_half(e0 - 1) : floor_half(e0);
    constexpr int p1 = c2 == 1 && two ? floor_half(e1 - 1) : floor_half(e1);
    constexpr int p2 = c2 == 2 && two ? floor_half(e2 - 1) : floor_half(e2);
    constexpr int s0 = d_slot_of(C, c2, p0, p1, p2);
    static_assert(s0 >= 0 && s0 < VS, "the 23-slot pattern is closed under this coarsening");
    acc[s0] += (two ? 0.5f * wc : wc) * v;
    if constexpr (two) {
        constexpr int s1 = d_slot_of(C, c2, p0 + (c2 == 0), p1 + (c2 == 1), p2 + (c2 == 2));
        static_assert(s1 >= 0 && s1 < VS, "the 23-slot pattern is closed under this coarsening");
        acc[s1] += 0.5f * wc * v;
    }
}
template <int C, int DN, int A_, int B_, int T>
struct RapGatherSlots {
    static __device__ __forceinline__ void run(const VLevelDev &F, size_t ci, float (&acc)[VS]) {
        d_rap_gather_term<C, DN, A_, B_, T>(F.coef[C][T][ci], acc);
        RapGatherSlots<C, DN, A_, B_, T + 1>::run(F, ci, acc);
    }
};
template <int C, int DN, int A_, int B_>
struct RapGatherSlots<C, DN, A_, B_, VS> {
    static __device__ __forceinline__ void run(const VLevelDev &, size_t, float (&)[VS]) {}
};
template <int C, int DN, int A_, int B_>
__device__ __forceinline__ void d_rap_gather_child(const VLevelDev &F, const int I[3], float (&acc)[VS]) {
    constexpr int t1 = (C + 1) % 3, t2 = (C + 2) % 3;
    int p[3];
    p[C] = 2 * I[C] + DN; p[t1] = 2 * I[t1] + A_; p[t2] = 2 * I[t2] + B_;
    // only children inside the finer level's box: what lies outside it was not written by this solve (it may hold another solve's rows)
#ifndef FLIPV_VMG_TEST_NO_BOXCHECK   // (build switch for checking that tests/test_gpu_parity.py::test_multigrid_hierarchy_carries_nothing_over... has teeth)
    if (p[0] < F.box.lo[0] || p[0] >= F.box.hi[0] || p[1] < F.box.lo[1] || p[1] >= F.box.hi[1] || p[2] < F.box.lo[2] || p[2] >= F.box.hi[2]) return;
#endif
    const size_t ci = cidx(F.L, p[0], p[1], p[2]);
    if (F.coef[C][slot_diag(C)][ci] == 0.0f) return;   // no row: all its slots are 0
    RapGatherSlots<C, DN, A_, B_, 0>::run(F, ci, acc);
}
// ---- level 0 -> level 1 the same way: the child's entries come from the matrix-free operator (viscositysolver.cpp:394-465 and the
// V / W analogues).  In the
// 23-slot numbering a fine row of component C has: the diagonal (slot 5); -fP / -fM towards its same-component neighbours along each
// axis a (fP, fM = the factor on the + / - side: the cell-centre factor along the normal, an edge factor across); and, for each
// transverse axis a, four entries of component a: -fP at p + e_a, +fP at p + e_a - e_C, +fM at p, -fM at p - e_C.  An entry exists
// only where the neighbour is a row (mask bit).
struct FineChild {   // everything a child row needs, loaded up front
    float vm, fP[3], fM[3];
    unsigned m, mP[3], mM[3], mPc[3];   // row masks at p, p + e_a, p - e_a, p + e_a - e_C
};
template <int C, int T>
__device__ __forceinline__ float d_fine_entry(const FineChild &f) {
    constexpr int t1 = (C + 1) % 3, t2 = (C + 2) % 3;
    if constexpr (T < 15) {
        constexpr int dn = T / 5 - 1, q = T % 5;
        if constexpr (dn == 0 && q == 0) return f.vm + ((f.fP[0] + f.fM[0]) + (f.fP[1] + f.fM[1]) + (f.fP[2] + f.fM[2]));
        else if constexpr (dn != 0 && q == 0) return dn > 0 ? (((f.mP[C] >> C) & 1u) ? -f.fP[C] : 0.0f) : (((f.mM[C] >> C) & 1u) ? -f.fM[C] : 0.0f);
        else if constexpr (dn == 0) {
            constexpr int a = (q == 1 || q == 2) ? t1 : t2;
            constexpr bool plus = q == 2 || q == 4;
            return plus ? (((f.mP[a] >> C) & 1u) ? -f.fP[a] : 0.0f) : (((f.mM[a] >> C) & 1u) ? -f.fM[a] : 0.0f);
        } else return 0.0f;
    } else {
        constexpr int a = T < 19 ? t1 : t2, idx = T < 19 ? T - 15 : T - 19;
        constexpr int dn = idx / 2 - 1, da = idx % 2;
        if constexpr (dn == 0 && da == 1) return ((f.mP[a] >> a) & 1u) ? -f.fP[a] : 0.0f;
        else if constexpr (dn == -1 && da == 1) return ((f.mPc[a] >> a) & 1u) ? f.fP[a] : 0.0f;
        else if constexpr (dn == 0 && da == 0) return ((f.m >> a) & 1u) ? f.fM[a] : 0.0f;
        else return ((f.mM[C] >> a) & 1u) ? -f.fM[a] : 0.0f;
    }
}
template <int C, int DN, int A_, int B_, int T>
struct RapGatherFineSlots {
    static __device__ __forceinline__ void run(const FineChild &f, float (&acc)[VS]) {
        constexpr bool structural_zero = T < 15 && (T / 5 - 1) != 0 && (T % 5) != 0;   // a fine row has no normal-times-transverse neighbour
        if constexpr (!structural_zero) d_rap_gather_term<C, DN, A_, B_, T>(d_fine_entry<C, T>(f), acc);
        RapGatherFineSlots<C, DN, A_, B_, T + 1>::run(f, acc);
    }
};
template <int C, int DN, int A_, int B_>
struct RapGatherFineSlots<C, DN, A_, B_, VS> {
    static __device__ __forceinline__ void run(const FineChild &, float (&)[VS]) {}
};
template <int C, int DN, int A_, int B_>
__device__ __forceinline__ void d_rap_gather_child_fine(const FineOp &A, const Lay &L, const int I[3], float (&acc)[VS]) {
    constexpr int t1 = (C + 1) % 3, t2 = (C + 2) % 3;
    int p[3];
    p[C] = 2 * I[C] + DN; p[t1] = 2 * I[t1] + A_; p[t2] = 2 * I[t2] + B_;
    if (!d_in_lattice(L, C, p) || !d_owned(L, p[0], p[1], p[2])) return;   // rows are the rank's own (block contexts: what lies beyond is a neighbour's, or not allocated at all)
    const size_t ci = A.brick ? bidx(A.LB, p[0], p[1], p[2]) : gidx(L, p[0], p[1], p[2]);
    FineChild f;
    f.m = A.mask[ci];
    if (!((f.m >> C) & 1u)) return;
    const NbOff o = A.brick ? nb_brick(A.LB, p[0], p[1], p[2]) : nb_plain(L);
    const long stp[3] = {o.xp, o.yp, o.zp}, stm[3] = {o.xm, o.ym, o.zm};   // offsets to the +- neighbours along each axis
    f.vm = A.vm[C][A.swz ? sidx(L, p[0], p[1], p[2]) : ci];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (a == C) { f.fP[a] = A.fC[ci]; f.fM[a] = A.fC[ci + stm[C]]; }
        else { f.fP[a] = A.fE[3 - C - a][ci + stp[a]]; f.fM[a] = A.fE[3 - C - a][ci]; }
        f.mP[a] = A.mask[ci + stp[a]]; f.mM[a] = A.mask[ci + stm[a]];
        f.mPc[a] = a == C ? 0u : A.mask[ci + stp[a] + stm[C]];
    }
    RapGatherFineSlots<C, DN, A_, B_, 0>::run(f, acc);
}
template <int C>
__device__ __forceinline__ void d_rap_gather_row_fine(const FineOp &A, const Lay &L, const VLevelDev &Cl, int i, int j, int k) {
    const int I[3] = {i, j, k};
    float acc[VS];
#pragma unroll
    for (int s = 0; s < VS; s++) acc[s] = 0.0f;
    d_rap_gather_child_fine<C, -1, 0, 0>(A, L, I, acc); d_rap_gather_child_fine<C, -1, 1, 0>(A, L, I, acc); d_rap_gather_child_fine<C, -1, 0, 1>(A, L, I, acc); d_rap_gather_child_fine<C, -1, 1, 1>(A, L, I, acc);
    d_rap_gather_child_fine<C, 0, 0, 0>(A, L, I, acc); d_rap_gather_child_fine<C, 0, 1, 0>(A, L, I, acc); d_rap_gather_child_fine<C, 0, 0, 1>(A, L, I, acc); d_rap_gather_child_fine<C, 0, 1, 1>(A, L, I, acc);
    d_rap_gather_child_fine<C, 1, 0, 0>(A, L, I, acc); d_rap_gather_child_fine<C, 1, 1, 0>(A, L, I, acc); d_rap_gather_child_fine<C, 1, 0, 1>(A, L, I, acc); d_rap_gather_child_fine<C, 1, 1, 1>(A, L, I, acc);
    const size_t co = cidx(Cl.L, i, j, k);
#pragma unroll
    for (int s = 0; s < VS; s++) Cl.coef[C][s][co] = acc[s];
}
__global__ __launch_bounds__(256) void k_vmg_rap_gather_fine(FineOp A, Lay L, VLevelDev Cl) {   // grid: level 1's box, blockIdx.z = 3 * plane + component
    const int c = (int)blockIdx.z % 3;
    const int i = Cl.box.lo[0] + blockIdx.x * 64 + threadIdx.x, j = Cl.box.lo[1] + blockIdx.y * 4 + threadIdx.y, k = Cl.box.lo[2] + (int)blockIdx.z / 3;
    if (i >= Cl.box.hi[0] || j >= Cl.box.hi[1]) return;
    if (c == 0) d_rap_gather_row_fine<0>(A, L, Cl, i, j, k);
    else if (c == 1) d_rap_gather_row_fine<1>(A, L, Cl, i, j, k);
    else d_rap_gather_row_fine<2>(A, L, Cl, i, j, k);
}

template <int C>
__device__ __forceinline__ void d_rap_gather_row(const VLevelDev &F, const VLevelDev &Cl, int i, int j, int k) {
    const int I[3] = {i, j, k};
    float acc[VS];
#pragma unroll
    for (int s = 0; s < VS; s++) acc[s] = 0.0f;
    d_rap_gather_child<C, -1, 0, 0>(F, I, acc); d_rap_gather_child<C, -1, 1, 0>(F, I, acc); d_rap_gather_child<C, -1, 0, 1>(F, I, acc); d_rap_gather_child<C, -1, 1, 1>(F, I, acc);
    d_rap_gather_child<C, 0, 0, 0>(F, I, acc); d_rap_gather_child<C, 0, 1, 0>(F, I, acc); d_rap_gather_child<C, 0, 0, 1>(F, I, acc); d_rap_gather_child<C, 0, 1, 1>(F, I, acc);
    d_rap_gather_child<C, 1, 0, 0>(F, I, acc); d_rap_gather_child<C, 1, 1, 0>(F, I, acc); d_rap_gather_child<C, 1, 0, 1>(F, I, acc); d_rap_gather_child<C, 1, 1, 1>(F, I, acc);
    const size_t co = cidx(Cl.L, i, j, k);
#pragma unroll
    for (int s = 0; s < VS; s++) Cl.coef[C][s][co] = acc[s];
}
__global__ __launch_bounds__(256) void k_vmg_rap_gather(VLevelDev F, VLevelDev Cl) {   // grid: the COARSE level's box, blockIdx.z = 3 * plane + component
    const int c = (int)blockIdx.z % 3;
    const int i = Cl.box.lo[0] + blockIdx.x * 64 + threadIdx.x, j = Cl.box.lo[1] + blockIdx.y * 4 + threadIdx.y, k = Cl.box.lo[2] + (int)blockIdx.z / 3;
    if (i >= Cl.box.hi[0] || j >= Cl.box.hi[1]) return;
    if (c == 0) d_rap_gather_row<0>(F, Cl, i, j, k);
    else if (c == 1) d_rap_gather_row<1>(F, Cl, i, j, k);
    else d_rap_gather_row<2>(F, Cl, i, j, k);
}

// ---- coarse levels: y = A x for one dof of component C at (i, j, k)
// What the cycle's kernels read of a row is its fp32 diagonal and its 22 off-diagonal entries as fp16, two per 32-bit word, scaled by the power
// of two below the diagonal (2^e <= d < 2^(e+1)): 11 + 1 loads and 48 bytes per row instead of 23 loads and 92 bytes -- the sweeps of level 1 are
// bound by exactly that stream (2 040 bricks x 3 components x 23 grids = 36 MB per launch at 256^3, five launches per iteration).  The scale is a
// power of two, so an entry is ROUNDED TO 11 BITS and nothing else: a_ij and a_ji round alike and the level's operator stays as symmetric as the
// Galerkin sums left it (entries below 6e-5 of the diagonal fall into fp16's subnormals: an absolute error of 3e-8 d).  The diagonal that goes with
// the rounded entries (VLevelDev::dd, fp32) takes up what the rounding changed in the row's SAME-COMPONENT entries, d' = d + sum (a - a'): the
// operator is a mass term plus a viscous term that annihilates a translation of a component, the mass term is 1e-3 .. 1e-5 of the entries when the
// system is stiff, and a rounding of 5e-4 per entry without that correction leaves the translation of a small liquid cluster with a NEGATIVE energy
// on the coarse levels.  The fp32 grids stay what the set-up works on (Galerkin gathers, the all-reduce over the ranks, the LDS-resident level's rows).
constexpr int off_slot(int n) { return n < 5 ? n : n + 1; }   // the n-th off-diagonal slot (slot_diag = 5)
__device__ __forceinline__ float d_row_scale(float d) { return __uint_as_float(__float_as_uint(d) & 0x7f800000u); }
__device__ __forceinline__ float d_half_bits(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
template <int C>
__device__ __forceinline__ float d_row_dot(const VLevelDev &A, size_t ci, float d, const unsigned (&u)[VH], const float (&xv)[VS]) {
    float s = 0.0f;
#pragma unroll
    for (int n = 0; n < 2 * VH; n++) s += d_half_bits((n & 1) ? (u[n >> 1] >> 16) : (u[n >> 1] & 0xffffu)) * xv[off_slot(n)];
    return d_row_scale(d) * s + d * xv[slot_diag(C)];
}
template <int C, bool PK>
__device__ __forceinline__ float d_apply(const VLevelDev &A, const Vec3p &x, size_t ci, int i, int j, int k, float d) {
    // all coefficients, then all neighbour values, then the sum: independent loads instead of dependent
    // load-test-load chains (empty slots hold 0 and the padding bricks make every neighbour address valid)
    float xv[VS];
    if (!PK) {
        float v[VS];
#pragma unroll
        for (int q = 0; q < VS; q++) v[q] = A.coef[C][q][ci];
#pragma unroll
        for (int q = 0; q < VS; q++) xv[q] = x.p[slot_comp(C, q)][cidx(A.L, i + slot_off(C, q, 0), j + slot_off(C, q, 1), k + slot_off(C, q, 2))];
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < VS; q++) s += v[q] * xv[q];
        return s;
    }
    unsigned u[VH];
#pragma unroll
    for (int w = 0; w < VH; w++) u[w] = A.ch[C][w][ci];
#pragma unroll
    for (int q = 0; q < VS; q++) xv[q] = x.p[slot_comp(C, q)][cidx(A.L, i + slot_off(C, q, 0), j + slot_off(C, q, 1), k + slot_off(C, q, 2))];
    return d_row_dot<C>(A, ci, d, u, xv);
}

// The six steps of a coarse level's share of the V-cycle, for the three dofs of index (i, j, k).  Vectors of a level are zero
// wherever there is no row (diagonal slot 0) and outside the level's box.
enum VmgOp { OP_RESTRICT = 0,   // b = P^T (finer level's t) ; x = omega b/d          (first pre-sweep from a zero guess)
       OP_PRE2 = 1,       // y = x + omega (b - A x)/d
       OP_RESID = 2,      // t = b - A y
       OP_PROLONG = 3,    // y += P (coarser level's x)
       OP_POST1 = 4,      // t = y + omega (b - A y)/d
       OP_POST2 = 5,      // x = t + omega (b - A t)/d
       OP_SWEEP_XY = 6,   // y = x + omega (b - A x)/d     (coarsest level)
       OP_SWEEP_YX = 7,   // x = y + omega (b - A y)/d
       OP_FIRST = 9,      // x = omega b/d from a right-hand side that is already there (the level whose b is summed over the ranks)
       OP_PROPOST = 8 };  // OP_PROLONG and OP_POST1 in one launch: t = y' + omega (b - A y')/d with y' = y + P (coarser level's x) formed at the 23 stencil positions; y itself is not updated
// (P x)(M, p) for ANY index p of a level (rows or not): along the component's normal an even index has one parent, an odd one the mean of two --
// written as the mean of parents (p[M] >> 1) and ((p[M] + 1) >> 1), which coincide for even p[M]; across, the parent is p >> 1
__device__ __forceinline__ float d_prolong_at(int M, const Lay &Cn, const Vec3p &cx, int p0, int p1, int p2) {   // (M folds to a constant in the unrolled callers)
    const int q0 = p0 >> 1, q1 = p1 >> 1, q2 = p2 >> 1;
    const int r0 = M == 0 ? (p0 + 1) >> 1 : q0, r1 = M == 1 ? (p1 + 1) >> 1 : q1, r2 = M == 2 ? (p2 + 1) >> 1 : q2;
    return 0.5f * (cx.p[M][cidx(Cn, q0, q1, q2)] + cx.p[M][cidx(Cn, r0, r1, r2)]);
}
// (P^T t)(C, P): the <= 12 fine children of coarse dof P of component C.  FINE0: the finer level is level 0 (plain rows, gidx);
// otherwise a coarse level (bricks, cidx)
// FINE0: what the finer level is: 0 a coarse level (bricks, cidx), 1 level 0 in plain rows (gidx), 2 level 0 in the brick layout (bidx; F = the brick Lay)
template <int C, int FINE0>
__device__ __forceinline__ float d_restrict(const Lay &F, const Vec3p &ft, const int P[3]) {
    constexpr int t1 = (C + 1) % 3, t2 = (C + 2) % 3;
    float s = 0.0f;
#pragma unroll
    for (int dn = -1; dn <= 1; dn++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                int q[3];
                q[C] = 2 * P[C] + dn; q[t1] = 2 * P[t1] + a; q[t2] = 2 * P[t2] + b;
                if (d_in_lattice(F, C, q) && d_owned(F, q[0], q[1], q[2]))   // (level 0 of a block context, either layout, and a distributed coarse level: only the rank's own indices are ITS rows; the halo holds the neighbours' residuals.  A global level's Lay owns everything)
                    s += (dn == 0 ? 1.0f : 0.5f) * ft.p[C][FINE0 == 1 ? gidx(F, q[0], q[1], q[2]) : (FINE0 == 2 ? bidx(F, q[0], q[1], q[2]) : cidx(F, q[0], q[1], q[2]))];
            }
    return s;
}
// PK: the level's rows are read in the packed form (ch, dd); otherwise from the fp32 grids (the tail's levels always; every level when the system is too stiff for 11 bits)
template <int OP, int C, int FINE0, bool PK>
__device__ __forceinline__ void d_vmg_step(const VLevelDev &A, const Lay &F, const Vec3p &ft, const Lay &Cn, const Vec3p &cx, int i, int j, int k) {
    const size_t ci = cidx(A.L, i, j, k);
    const int P[3] = {i, j, k};
    const float d = PK ? A.dd[C][ci] : A.coef[C][slot_diag(C)][ci];
    if (OP == OP_RESTRICT) {
        const float s = d != 0.0f ? d_restrict<C, FINE0>(F, ft, P) : 0.0f;   // (a coarse dof with a fine child that is a row has a diagonal)
        A.b.p[C][ci] = s;
        A.x.p[C][ci] = d != 0.0f ? A.w[0] * s / d : 0.0f;
        return;
    }
    if (OP == OP_FIRST) { A.x.p[C][ci] = d != 0.0f ? A.w[0] * A.b.p[C][ci] / d : 0.0f; return; }
    if (d == 0.0f) return;   // no row: every vector stays 0 here
    if (OP == OP_PROLONG) {
        int Q[2][3];
        float w[2];
        const int n = d_parents(C, P, Q, w);
        float s = w[0] * cx.p[C][cidx(Cn, Q[0][0], Q[0][1], Q[0][2])];
        if (n == 2) s += w[1] * cx.p[C][cidx(Cn, Q[1][0], Q[1][1], Q[1][2])];
        A.y.p[C][ci] += s;
        return;
    }
    if (OP == OP_PROPOST) {
        // Entries towards indices without a row are exactly 0 on every level (an entry exists only where the neighbour is a row), so the
        // prolongated values formed at such neighbours drop out, as the zeros of y did in the two-launch form.
        unsigned u[VH];
        float xv[VS], v[VS];
        if (PK) {
#pragma unroll
            for (int w = 0; w < VH; w++) u[w] = A.ch[C][w][ci];
        } else {
#pragma unroll
            for (int q = 0; q < VS; q++) v[q] = A.coef[C][q][ci];
        }
#pragma unroll
        for (int q = 0; q < VS; q++) {
            const int pi = i + slot_off(C, q, 0), pj = j + slot_off(C, q, 1), pk = k + slot_off(C, q, 2);
            xv[q] = A.y.p[slot_comp(C, q)][cidx(A.L, pi, pj, pk)] + d_prolong_at(slot_comp(C, q), Cn, cx, pi, pj, pk);
        }
        float ax = 0.0f;
        if (PK) ax = d_row_dot<C>(A, ci, d, u, xv);
        else {
#pragma unroll
            for (int q = 0; q < VS; q++) ax += v[q] * xv[q];
        }
        A.t.p[C][ci] = xv[slot_diag(C)] + A.w[0] * (A.b.p[C][ci] - ax) / d;
        return;
    }
    const Vec3p &in = (OP == OP_PRE2 || OP == OP_SWEEP_XY) ? A.x : (OP == OP_POST2 ? A.t : A.y);
    const Vec3p &out = (OP == OP_PRE2 || OP == OP_SWEEP_XY) ? A.y : ((OP == OP_RESID || OP == OP_POST1) ? A.t : A.x);
    const float ax = d_apply<C, PK>(A, in, ci, i, j, k, d);
    const float bb = A.b.p[C][ci];
    // first sweep of a pair: OP_POST1 (OP_RESTRICT / OP_PROPOST above); second: OP_PRE2, OP_POST2; the coarsest level's sweeps: VMG_OMEGA
    const float w = (OP == OP_SWEEP_XY || OP == OP_SWEEP_YX) ? VMG_OMEGA : A.w[OP == OP_POST1 ? 0 : 1];
    out.p[C][ci] = OP == OP_RESID ? bb - ax : in.p[C][ci] + w * (bb - ax) / d;
}
template <int OP, int FINE0, bool PK>
__device__ __forceinline__ void d_vmg_step_c(int c, const VLevelDev &A, const Lay &F, const Vec3p &ft, const Lay &Cn, const Vec3p &cx, int i, int j, int k) {
    if (c == 0) d_vmg_step<OP, 0, FINE0, PK>(A, F, ft, Cn, cx, i, j, k);
    else if (c == 1) d_vmg_step<OP, 1, FINE0, PK>(A, F, ft, Cn, cx, i, j, k);
    else d_vmg_step<OP, 2, FINE0, PK>(A, F, ft, Cn, cx, i, j, k);
}
// One step of one level: a wave per listed BRICK (8 x 4 x 2 indices) and component.  The list holds the bricks of the level's box
// that carry at least one row, numbered inside the box's brick range (bx fastest); positions of an edge brick that lie outside the
// box are skipped (they hold other solves' entries).  The level descriptors live in device memory (VmgState::d_lev, uploaded per
// solve): lev[l] is this level, lev[l - 1] the finer one (l = 0: the fine level's lattice and residual come as arguments), lev[l + 1]
// the coarser one.  conv: nothing to do once the solve has stopped.
struct BrickRange { int b0[3], nb[3]; };   // first brick and brick counts of a box, per axis (padded brick coordinates)
__device__ __forceinline__ BrickRange d_brick_range(const Box3 &B) {
    BrickRange R;
    R.b0[0] = (B.lo[0] + 8) >> 3; R.nb[0] = ((B.hi[0] - 1 + 8) >> 3) - R.b0[0] + 1;
    R.b0[1] = (B.lo[1] + 4) >> 2; R.nb[1] = ((B.hi[1] - 1 + 4) >> 2) - R.b0[1] + 1;
    R.b0[2] = (B.lo[2] + 2) >> 1; R.nb[2] = ((B.hi[2] - 1 + 2) >> 1) - R.b0[2] + 1;
    return R;
}
// this lane's index inside brick `code` of the box's brick range; false if it lies outside the box
__device__ __forceinline__ bool d_brick_lane(const Box3 &B, int code, int lane, int &i, int &j, int &k) {
    const BrickRange R = d_brick_range(B);
    const int bx = code % R.nb[0], r = code / R.nb[0], by = r % R.nb[1], bz = r / R.nb[1];
    i = ((R.b0[0] + bx) << 3) - 8 + (lane & 7);
    j = ((R.b0[1] + by) << 2) - 4 + ((lane >> 3) & 3);
    k = ((R.b0[2] + bz) << 1) - 2 + (lane >> 5);
    return i >= B.lo[0] && i < B.hi[0] && j >= B.lo[1] && j < B.hi[1] && k >= B.lo[2] && k < B.hi[2];
}
template <int OP, bool PK>
__global__ __launch_bounds__(256) void k_vmg_step(const VLevelDev *__restrict__ lev, int l, Lay F0, Vec3p ft0, const int *__restrict__ conv, int fineBrick) {   // fineBrick: level 0 (F0, ft0) is in the brick layout
    if (*conv >= 0) return;
    const VLevelDev &A = lev[l];
    const int sidx = (int)blockIdx.x * 4 + (int)threadIdx.y;   // a wave per brick and component
    if (sidx >= A.nstrips) return;
    const int c = (int)blockIdx.y;
    int i, j, k;
    if (!d_brick_lane(A.box, A.strips[sidx], (int)threadIdx.x, i, j, k)) return;
    if (OP == OP_RESTRICT) {
        if (l == 0) { if (fineBrick) d_vmg_step_c<OP, 2, PK>(c, A, F0, ft0, A.L, ft0, i, j, k); else d_vmg_step_c<OP, 1, PK>(c, A, F0, ft0, A.L, ft0, i, j, k); }
        else d_vmg_step_c<OP, 0, PK>(c, A, lev[l - 1].L, lev[l - 1].t, A.L, ft0, i, j, k);
    } else if (OP == OP_PROLONG || OP == OP_PROPOST) d_vmg_step_c<OP, 0, PK>(c, A, A.L, ft0, lev[l + 1].L, lev[l + 1].x, i, j, k);
    else d_vmg_step_c<OP, 0, PK>(c, A, A.L, ft0, A.L, ft0, i, j, k);
}
// b of level l over its whole box from the finer level's residual, without the first sweep: the level whose right-hand side is summed
// over the ranks (VmgState::globalFrom).  Grid: (ceil(w / 64), ceil(h / 4), 3 depth), block (64, 4)
// (the result goes straight into the dense staging buffer of the all-reduce: [component][position of the box, x fastest])
template <int C, int FINE0>
__device__ __forceinline__ float d_restrict_only(const VLevelDev &A, const Lay &F, const Vec3p &ft, int i, int j, int k) {
    const size_t ci = cidx(A.L, i, j, k);
    const int P[3] = {i, j, k};
    return A.coef[C][slot_diag(C)][ci] != 0.0f ? d_restrict<C, FINE0>(F, ft, P) : 0.0f;
}
// box-shaped copy between `narr` grids of a level (array a at base + a * per, cidx addressing) and a dense buffer [narr][positions of the box]:
// what the all-reduces of the global hierarchy move (unpack = 0: grids -> buffer, 1: buffer -> grids)
__global__ __launch_bounds__(256) void k_vmg_box_pack(Lay L, Box3 B, float *__restrict__ base, size_t per, float *__restrict__ buf, int unpack) {
    const int w = B.hi[0] - B.lo[0], h = B.hi[1] - B.lo[1], d = B.hi[2] - B.lo[2];
    const size_t n = (size_t)w * h * d, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int i = B.lo[0] + (int)(t % w), j = B.lo[1] + (int)((t / w) % h), k = B.lo[2] + (int)(t / ((size_t)w * h));
    float *g = base + (size_t)blockIdx.y * per + cidx(L, i, j, k), *q = buf + (size_t)blockIdx.y * n + t;
    if (unpack) *g = *q; else *q = *g;
}
__global__ __launch_bounds__(256) void k_vmg_restrict_box(const VLevelDev *__restrict__ lev, int l, Lay F0, Vec3p ft0, const int *__restrict__ conv, int fineBrick, float *__restrict__ buf) {
    if (*conv >= 0) return;
    const VLevelDev &A = lev[l];
    const int w = A.box.hi[0] - A.box.lo[0], h = A.box.hi[1] - A.box.lo[1], dpt = A.box.hi[2] - A.box.lo[2];
    const int di = (int)(blockIdx.x * 64 + threadIdx.x), dj = (int)(blockIdx.y * 4 + threadIdx.y);
    const int c = (int)blockIdx.z % 3, dk = (int)blockIdx.z / 3;
    if (di >= w || dj >= h || dk >= dpt) return;
    const int i = A.box.lo[0] + di, j = A.box.lo[1] + dj, k = A.box.lo[2] + dk;
    float v;
#define RONLY(C_) (l > 0 ? d_restrict_only<C_, 0>(A, lev[l - 1].L, lev[l - 1].t, i, j, k) : (fineBrick ? d_restrict_only<C_, 2>(A, F0, ft0, i, j, k) : d_restrict_only<C_, 1>(A, F0, ft0, i, j, k)))
    if (c == 0) v = RONLY(0); else if (c == 1) v = RONLY(1); else v = RONLY(2);
#undef RONLY
    buf[(size_t)c * ((size_t)w * h * dpt) + (size_t)di + (size_t)w * ((size_t)dj + (size_t)h * dk)] = v;
}
// b of a DISTRIBUTED level l over the box `R` (its owned box widened by one entry): the rank's share of P^T t -- its own finer rows only (d_restrict's
// ownership test) --, written straight into the level's b; the shares of the entries beyond the owned box then travel to their owners (fv_halo_level, add)
__global__ __launch_bounds__(256) void k_vmg_restrict_partial(const VLevelDev *__restrict__ lev, int l, Lay F0, Vec3p ft0, const int *__restrict__ conv, int fineBrick, Box3 R) {
    if (*conv >= 0) return;
    const VLevelDev &A = lev[l];
    const int i = R.lo[0] + (int)(blockIdx.x * 64 + threadIdx.x), j = R.lo[1] + (int)(blockIdx.y * 4 + threadIdx.y);
    const int c = (int)blockIdx.z % 3, k = R.lo[2] + (int)blockIdx.z / 3;
    if (i >= R.hi[0] || j >= R.hi[1] || k >= R.hi[2]) return;
    float v;
#define RONLY(C_) (l > 0 ? d_restrict_only<C_, 0>(A, lev[l - 1].L, lev[l - 1].t, i, j, k) : (fineBrick ? d_restrict_only<C_, 2>(A, F0, ft0, i, j, k) : d_restrict_only<C_, 1>(A, F0, ft0, i, j, k)))
    if (c == 0) v = RONLY(0); else if (c == 1) v = RONLY(1); else v = RONLY(2);
#undef RONLY
    A.b.p[c][cidx(A.L, i, j, k)] = v;
}
// the summed right-hand side back into the level's grids, with the first sweep x = omega b/d from the zero guess (`first`; the tail kernel does its own)
__global__ __launch_bounds__(256) void k_vmg_unpack_rhs(const VLevelDev *__restrict__ lev, int l, const float *__restrict__ buf, const int *__restrict__ conv, int first) {
    if (*conv >= 0) return;
    const VLevelDev &A = lev[l];
    const int w = A.box.hi[0] - A.box.lo[0], h = A.box.hi[1] - A.box.lo[1], dpt = A.box.hi[2] - A.box.lo[2];
    const size_t n = (size_t)w * h * dpt, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int c = (int)blockIdx.y;
    const size_t ci = cidx(A.L, A.box.lo[0] + (int)(t % w), A.box.lo[1] + (int)((t / w) % h), A.box.lo[2] + (int)(t / ((size_t)w * h)));
    const float b = buf[(size_t)c * n + t];
    A.b.p[c][ci] = b;
    if (first) { const float d = A.dd[c][ci]; A.x.p[c][ci] = d != 0.0f ? A.w[0] * b / d : 0.0f; }
}
// The same two steps over the level's LISTED bricks only (a wave per listed brick and component, like k_vmg_step): the staging buffer is
// [component][listed brick][64] -- 1.5 MB instead of the union box's 4 MB at 256^3 on level 1.  Every rank lists the same bricks in the same
// order: the list is cut from the level's summed coefficient grids, which the all-reduce leaves identical on all ranks.
__global__ __launch_bounds__(256) void k_vmg_restrict_list(const VLevelDev *__restrict__ lev, int l, Lay F0, Vec3p ft0, const int *__restrict__ conv, int fineBrick, float *__restrict__ buf) {
    if (*conv >= 0) return;
    const VLevelDev &A = lev[l];
    const int sidx = (int)blockIdx.x * 4 + (int)threadIdx.y;
    if (sidx >= A.nstrips) return;
    const int c = (int)blockIdx.y;
    int i, j, k;
    float v = 0.0f;
    if (d_brick_lane(A.box, A.strips[sidx], (int)threadIdx.x, i, j, k)) {
#define RONLY(C_) (l > 0 ? d_restrict_only<C_, 0>(A, lev[l - 1].L, lev[l - 1].t, i, j, k) : (fineBrick ? d_restrict_only<C_, 2>(A, F0, ft0, i, j, k) : d_restrict_only<C_, 1>(A, F0, ft0, i, j, k)))
        if (c == 0) v = RONLY(0); else if (c == 1) v = RONLY(1); else v = RONLY(2);
#undef RONLY
    }
    buf[((size_t)c * (size_t)A.nstrips + (size_t)sidx) * 64 + threadIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_vmg_unpack_list(const VLevelDev *__restrict__ lev, int l, const float *__restrict__ buf, const int *__restrict__ conv, int first) {
    if (*conv >= 0) return;
    const VLevelDev &A = lev[l];
    const int sidx = (int)blockIdx.x * 4 + (int)threadIdx.y;
    if (sidx >= A.nstrips) return;
    const int c = (int)blockIdx.y;
    int i, j, k;
    if (!d_brick_lane(A.box, A.strips[sidx], (int)threadIdx.x, i, j, k)) return;
    const size_t ci = cidx(A.L, i, j, k);
    const float b = buf[((size_t)c * (size_t)A.nstrips + (size_t)sidx) * 64 + threadIdx.x];
    A.b.p[c][ci] = b;
    if (first) { const float d = A.dd[c][ci]; A.x.p[c][ci] = d != 0.0f ? A.w[0] * b / d : 0.0f; }
}
// bricks of a level's box that hold rows: flags (one wave per brick), then an ordered compaction by one workgroup
__global__ __launch_bounds__(256) void k_vmg_strip_flags(VLevelDev A, int *__restrict__ flag, int nbricks) {
    const int sidx = (int)blockIdx.x * 4 + (int)threadIdx.y;
    if (sidx >= nbricks) return;
    int i, j, k;
    bool any = false;
    if (d_brick_lane(A.box, sidx, (int)threadIdx.x, i, j, k)) {
        const size_t ci = cidx(A.L, i, j, k);
        any = A.coef[0][slot_diag(0)][ci] != 0.0f || A.coef[1][slot_diag(1)][ci] != 0.0f || A.coef[2][slot_diag(2)][ci] != 0.0f;
    }
    const unsigned long long m = __ballot(any);
    if (threadIdx.x == 0) flag[sidx] = m != 0ull;
}
// the packed copy of a level's rows (d_row_dot): a wave per brick of the level's box and component
__global__ __launch_bounds__(256) void k_vmg_pack(VLevelDev A, int nbricks) {
    const int sidx = (int)blockIdx.x * 4 + (int)threadIdx.y;
    if (sidx >= nbricks) return;
    const int c = (int)blockIdx.y;
    int i, j, k;
    if (!d_brick_lane(A.box, sidx, (int)threadIdx.x, i, j, k)) return;
    const size_t ci = cidx(A.L, i, j, k);
    const float d = A.coef[c][slot_diag(c)][ci];
    float *dd = const_cast<float *>(A.dd[c]);
    if (d == 0.0f) { dd[ci] = 0.0f; return; }   // (inside the box every position is rewritten: an earlier solve's row may have been here)
    if (!A.packed) { dd[ci] = d; return; }
    float a[2 * VH];
#pragma unroll
    for (int n = 0; n < 2 * VH; n++) a[n] = A.coef[c][off_slot(n)][ci];
    // rounding an entry to fp16 under a power-of-two scale is rounding it to 11 bits whatever the scale (outside the subnormals): the diagonal's
    // correction can be formed under the scale of d and the entries packed under the scale of d'
    auto scale_inv = [](float v) { const unsigned eb = __float_as_uint(v) & 0x7f800000u; return eb != 0u && eb < 0x7f000000u ? __uint_as_float(0x7f000000u - eb) : 0.0f; };   // 2^-e
    auto round11 = [](float v) { return (float)(_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f); };
    const float inv0 = scale_inv(d), sc0 = d_row_scale(d);
    float corr = 0.0f;
#pragma unroll
    for (int n = 0; n < 14; n++) corr += a[n] - sc0 * round11(a[n] * inv0);   // off-diagonal slots 0..14 without 5: the row's own component
    float dn = d + corr;
    if (!(dn > 0.5f * d)) dn = d;
    dd[ci] = dn;
    const float inv = scale_inv(dn);
#pragma unroll
    for (int w = 0; w < VH; w++) {
        const _Float16 h0 = (_Float16)fminf(fmaxf(a[2 * w] * inv, -65504.0f), 65504.0f), h1 = (_Float16)fminf(fmaxf(a[2 * w + 1] * inv, -65504.0f), 65504.0f);
        const_cast<unsigned *>(A.ch[c][w])[ci] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    }
}
__global__ __launch_bounds__(1024) void k_vmg_strip_compact(const int *__restrict__ flag, int n, int *__restrict__ list, int *__restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int start = 0; start < n; start += 1024) {
        const int t = start + (int)threadIdx.x;
        const int f = t < n ? flag[t] != 0 : 0;
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int woff = 0, total = 0;
        for (int q = 0; q < 16; q++) { if (q < wv) woff += wsum[q]; total += wsum[q]; }
        if (f) list[base + woff + before] = t;
        __syncthreads();
        if (threadIdx.x == 0) base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}
// the coarsest level's rows per component (box positions), for the LDS-resident sweeps of k_vmg_tail: out = [3][1024], cnt[0..2]
// the row counts, cnt[3] = 1 if the box (with its rim) and every component's rows fit
constexpr int VMG_LDS_POS = 4096;     // positions of the box including its rim
__global__ __launch_bounds__(1024) void k_vmg_coarsest_rows(VLevelDev A, int *__restrict__ out, int *__restrict__ cnt) {
    __shared__ int n[3];
    if (threadIdx.x < 3) n[threadIdx.x] = 0;
    __syncthreads();
    const int w = A.box.hi[0] - A.box.lo[0], h = A.box.hi[1] - A.box.lo[1], dz = A.box.hi[2] - A.box.lo[2];
    const int npos = w * h * dz;
    const bool fits = (w + 2) * (h + 2) * (dz + 2) <= VMG_LDS_POS;
    if (fits)
        for (int base = 0; base < 3 * npos; base += 1024) {   // in order within a component (a wave's rows stay neighbours)
            const int q = base + (int)threadIdx.x;
            int c = 0, r = 0;
            bool row = false;
            if (q < 3 * npos) {
                c = q / npos; r = q - c * npos;
                row = A.coef[c][slot_diag(c)][cidx(A.L, A.box.lo[0] + r % w, A.box.lo[1] + (r / w) % h, A.box.lo[2] + r / (w * h))] != 0.0f;
            }
            if (row) { const int idx = atomicAdd(&n[c], 1); if (idx < 1024) out[c * 1024 + idx] = r; }
        }
    __syncthreads();
    if (threadIdx.x < 3) cnt[threadIdx.x] = n[threadIdx.x];
    if (threadIdx.x == 0) cnt[3] = fits && n[0] <= 1024 && n[1] <= 1024 && n[2] <= 1024;
}

// the coarsest levels in one workgroup: levels lev[first..n), lev[first] restricts from lev[first - 1] (first = 0: from (F0, ft0))
template <int OP, int FINE0 = 0>
__device__ __forceinline__ void d_tail_step(const VLevelDev &A, const Lay &F, const Vec3p &ft, const Lay &Cn, const Vec3p &cx) {
    const int w = A.box.hi[0] - A.box.lo[0], h = A.box.hi[1] - A.box.lo[1], dz = A.box.hi[2] - A.box.lo[2];
    const int n = w * h * dz;
    for (int q = threadIdx.x; q < 3 * n; q += blockDim.x) {
        const int c = q / n, r = q - c * n;
        const int i = A.box.lo[0] + r % w, j = A.box.lo[1] + (r / w) % h, k = A.box.lo[2] + r / (w * h);
        d_vmg_step_c<OP, FINE0, false>(c, A, F, ft, Cn, cx, i, j, k);
    }
    __syncthreads();
}
// The coarsest level with the iterate in LDS and the operator rows in registers: a thread owns at most one row per component
// (the rows of each component are compacted into a list first), the three components' iterates sit in LDS boxes with a rim of
// zeros, so a sweep is 23 LDS reads per row and one barrier instead of a round trip through L2 per row (16 sweeps at 16^3: 200 us
// of the cycle's 520 before).  Falls back to the global-memory sweeps when the box or a component's row count does not fit.
struct CoarseRow { float cf[VS]; float invd, b, gersh; int li; size_t ci; bool has; };
template <int C>
__device__ __forceinline__ void d_coarsest_load(const VLevelDev &A, const Lay &F, const Vec3p &ft, int fine0, const int *rowlist, int row, bool active, int nrows, int W, int H, CoarseRow &R, float *xs0, int NP, int given) {   // row: this thread's index into the component's row list (if active); fine0: d_restrict's FINE0; given: b is in A.b already
    R.has = active && row < nrows;
    R.gersh = 0.0f;
    if (!R.has) return;
    const int w = A.box.hi[0] - A.box.lo[0], h = A.box.hi[1] - A.box.lo[1];
    const int r = rowlist[row];
    const int di = r % w, dj = (r / w) % h, dk = r / (w * h);
    const int P[3] = {A.box.lo[0] + di, A.box.lo[1] + dj, A.box.lo[2] + dk};
    R.ci = cidx(A.L, P[0], P[1], P[2]);
    R.li = (di + 1) + W * ((dj + 1) + H * (dk + 1));
#pragma unroll
    for (int q = 0; q < VS; q++) R.cf[q] = A.coef[C][q][R.ci];
    R.invd = 1.0f / R.cf[slot_diag(C)];
    R.b = given ? A.b.p[C][R.ci] : (fine0 == 1 ? d_restrict<C, 1>(F, ft, P) : (fine0 == 2 ? d_restrict<C, 2>(F, ft, P) : d_restrict<C, 0>(F, ft, P)));
    float g = 0.0f;
#pragma unroll
    for (int q = 0; q < VS; q++) g += fabsf(R.cf[q]);
    R.gersh = g * R.invd;
}
template <int C>
__device__ __forceinline__ void d_coarsest_sweep(const CoarseRow &R, const float *cur, float *nxt, int NP, int W, int WH, float omega) {
    if (!R.has) return;
    float ax = 0.0f;
#pragma unroll
    for (int q = 0; q < VS; q++) ax += R.cf[q] * cur[slot_comp(C, q) * NP + R.li + slot_off(C, q, 0) + slot_off(C, q, 1) * W + slot_off(C, q, 2) * WH];
    nxt[C * NP + R.li] = cur[C * NP + R.li] + omega * (R.b - ax) * R.invd;
}

// The LDS-resident coarsest level: restriction (or the given right-hand side), Chebyshev / damped-Jacobi sweeps, write-back of x.  xs: 2 x 3 x VMG_LDS_POS
// floats of LDS, zero on entry inside the box's rim.  A function of its own so that k_vmg_coarsest -- the whole tail whenever the tail is just this
// level, which is the usual case -- is compiled without the other levels' code paths around it (inside the generic tail the register allocator
// spilled two of a row's values and reloaded them in every sweep: 60.9 -> 53.2 us).  Measured and worse, all through spills or branches at the
// 128-VGPR budget of a 1 024-thread block: the 23 LDS values of a sweep loaded before the sum (55.9 us), rows dealt out over all threads with the
// component chosen at run time per row (130 us) or per wave with two rows per lane (77.9 us).
__device__ __forceinline__ void d_coarsest_solve(const VLevelDev *__restrict__ lev, int first, int n, int sweeps, int cheb, const Lay &F0, const Vec3p &ft0, int fineBrick, int bGiven,
                                                 float *xs, int W, int H, int NP, const int *const rowlist[3], const int cnt[3]) {
        const VLevelDev &A = lev[n - 1];
        const Lay &F = n - 1 == 0 ? F0 : lev[n - 2].L;
        const Vec3p &ft = n - 1 == 0 ? ft0 : lev[n - 2].t;
        CoarseRow RU, RV, RW;
        const int fine0 = n - 1 == 0 ? (fineBrick ? 2 : 1) : 0;
        const int given = bGiven && first == n - 1;
        // Who sweeps what.  With at most 512 rows per component (a 16^3 level of the bench scene has ~400) the lower eight waves take the U and V rows
        // and the upper eight the W rows: 14 busy waves with two rows or one per lane instead of 7 with three -- the sweeps are bound by LDS latency
        // on the critical path of the busiest wave.  The code is the same either way: a wave without rows of a component skips that block (execz).
        const bool split = cnt[0] <= 512 && cnt[1] <= 512 && cnt[2] <= 512;
        const int row = split ? ((int)threadIdx.x & 511) : (int)threadIdx.x;
        const bool lower = threadIdx.x < 512;
        d_coarsest_load<0>(A, F, ft, fine0, rowlist[0], row, !split || lower, cnt[0], W, H, RU, xs, NP, given);
        d_coarsest_load<1>(A, F, ft, fine0, rowlist[1], row, !split || lower, cnt[1], W, H, RV, xs, NP, given);
        d_coarsest_load<2>(A, F, ft, fine0, rowlist[2], row, !split || !lower, cnt[2], W, H, RW, xs, NP, given);
        // weights: Chebyshev on [hi / kappa, hi] when `cheb`, the fixed damping otherwise; the first one turns the zero guess into x = omega b/d
        float wlane = VMG_OMEGA;    // lane k of every wave holds sweep k's weight; a sweep reads its weight with v_readlane (a load from the level
                                    // descriptor per sweep sits on the critical path: +0.75 us per sweep; a broadcast LDS read: +0.3 us)
        if (cheb) {
            __shared__ float gmax[16];
            float g = fmaxf(RU.gersh, fmaxf(RV.gersh, RW.gersh));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) g = fmaxf(g, __shfl_xor(g, off, 64));
            if ((threadIdx.x & 63) == 0) gmax[threadIdx.x >> 6] = g;
            __syncthreads();
            g = gmax[0];
#pragma unroll
            for (int w = 1; w < 16; w++) g = fmaxf(g, gmax[w]);
            const float hi = fminf(g, VMG_LAMBDA_MAX), lo = hi / VMG_CHEB_KAPPA;
            wlane = 1.0f / (0.5f * (hi + lo) + 0.5f * (hi - lo) * A.cheb[threadIdx.x & 63]);
        }
        auto weight = [&](int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wlane), k)); };
        {
            const float w0 = weight(0);
            if (RU.has) xs[0 * NP + RU.li] = w0 * RU.b * RU.invd;
            if (RV.has) xs[1 * NP + RV.li] = w0 * RV.b * RV.invd;
            if (RW.has) xs[2 * NP + RW.li] = w0 * RW.b * RW.invd;
        }
        __syncthreads();
        float *cur = xs, *nxt = xs + 3 * VMG_LDS_POS;
        for (int s = 1; s <= sweeps; s++) {
            const float w = weight(s);
            d_coarsest_sweep<0>(RU, cur, nxt, NP, W, W * H, w);
            d_coarsest_sweep<1>(RV, cur, nxt, NP, W, W * H, w);
            d_coarsest_sweep<2>(RW, cur, nxt, NP, W, W * H, w);
            __syncthreads();
            float *t = cur; cur = nxt; nxt = t;
        }
        if (RU.has) A.x.p[0][RU.ci] = cur[0 * NP + RU.li];
        if (RV.has) A.x.p[1][RV.ci] = cur[1 * NP + RV.li];
        if (RW.has) A.x.p[2][RW.ci] = cur[2 * NP + RW.li];
        __syncthreads();
}
// sweeps: Jacobi sweeps on the coarsest level after the one that turns the zero guess into omega b/d; cheb: they use the sweeps + 1 Chebyshev weights of lev[n - 1].cheb
// bGiven: the right-hand side of level `first` is in its b already (summed over the ranks by the caller)
__global__ __launch_bounds__(1024) void k_vmg_tail(const VLevelDev *__restrict__ lev, int first, int n, int sweeps, int cheb, Lay F0, Vec3p ft0, const int *__restrict__ conv, int fineBrick, int bGiven) {
    if (*conv >= 0) return;
    // the coarsest level lives in LDS when its box and rows fit (k_vmg_coarsest_rows decided that for this solve)
    __shared__ float xs[2 * 3 * VMG_LDS_POS];
    int W = 0, H = 0, NP = 0;
    const int *rowlist[3] = {lev[n - 1].rowlist, lev[n - 1].rowlist + 1024, lev[n - 1].rowlist + 2048};
    int cnt[3] = {0, 0, 0};
    const bool coarsest_in_lds = lev[n - 1].rowcnt[3] != 0;
    if (coarsest_in_lds) {
        const VLevelDev &A = lev[n - 1];
        W = A.box.hi[0] - A.box.lo[0] + 2; H = A.box.hi[1] - A.box.lo[1] + 2; NP = W * H * (A.box.hi[2] - A.box.lo[2] + 2);
        cnt[0] = A.rowcnt[0]; cnt[1] = A.rowcnt[1]; cnt[2] = A.rowcnt[2];
        for (int e = threadIdx.x; e < 3 * NP; e += blockDim.x) { xs[e] = 0.0f; xs[3 * VMG_LDS_POS + e] = 0.0f; }
        __syncthreads();
    }
    for (int l = first; l < n; l++) {   // down
        const VLevelDev &A = lev[l];
        if (l == n - 1 && coarsest_in_lds) break;   // restricted straight into registers below
        if (l == first && bGiven) d_tail_step<OP_FIRST>(A, A.L, ft0, A.L, ft0);
        else if (l == 0) { if (fineBrick) d_tail_step<OP_RESTRICT, 2>(A, F0, ft0, A.L, ft0); else d_tail_step<OP_RESTRICT, 1>(A, F0, ft0, A.L, ft0); }
        else d_tail_step<OP_RESTRICT>(A, lev[l - 1].L, lev[l - 1].t, A.L, ft0);
        if (l + 1 < n) {
            d_tail_step<OP_PRE2>(A, A.L, ft0, A.L, ft0);
            d_tail_step<OP_RESID>(A, A.L, ft0, A.L, ft0);
        }
    }
    if (!coarsest_in_lds) {   // coarsest level through global memory (its restriction was done by the loop above)
        const VLevelDev &A = lev[n - 1];
        for (int s = 0; s < sweeps; s += 2) {
            d_tail_step<OP_SWEEP_XY>(A, A.L, ft0, A.L, ft0);
            d_tail_step<OP_SWEEP_YX>(A, A.L, ft0, A.L, ft0);
        }
    } else {
        d_coarsest_solve(lev, first, n, sweeps, cheb, F0, ft0, fineBrick, bGiven, xs, W, H, NP, rowlist, cnt);
    }
    for (int l = n - 2; l >= first; l--) {   // up
        const VLevelDev &A = lev[l];
        d_tail_step<OP_PROLONG>(A, A.L, ft0, lev[l + 1].L, lev[l + 1].x);
        d_tail_step<OP_POST1>(A, A.L, ft0, A.L, ft0);
        d_tail_step<OP_POST2>(A, A.L, ft0, A.L, ft0);
    }
}

// the tail when it is just the LDS-resident coarsest level (first == n - 1 and the level fits: k_vmg_coarsest_rows)
__global__ __launch_bounds__(1024) void k_vmg_coarsest(const VLevelDev *__restrict__ lev, int n, int sweeps, int cheb, Lay F0, Vec3p ft0, const int *__restrict__ conv, int fineBrick, int bGiven) {
    if (*conv >= 0) return;
    __shared__ float xs[2 * 3 * VMG_LDS_POS];
    const VLevelDev &A = lev[n - 1];
    const int *rowlist[3] = {A.rowlist, A.rowlist + 1024, A.rowlist + 2048};
    const int W = A.box.hi[0] - A.box.lo[0] + 2, H = A.box.hi[1] - A.box.lo[1] + 2, NP = W * H * (A.box.hi[2] - A.box.lo[2] + 2);
    const int cnt[3] = {A.rowcnt[0], A.rowcnt[1], A.rowcnt[2]};
    for (int e = threadIdx.x; e < 3 * NP; e += blockDim.x) { xs[e] = 0.0f; xs[3 * VMG_LDS_POS + e] = 0.0f; }
    __syncthreads();
    d_coarsest_solve(lev, n - 1, n, sweeps, cheb, F0, ft0, fineBrick, bGiven, xs, W, H, NP, rowlist, cnt);
}

// ---- what the x / r kernels of either layout publish and the p kernels test: max|r| into rmax(it)
__device__ __forceinline__ void d_vmg_publish_max(const PcgScal &sc, int it, float mx, float mxs, double *lds) {   // mxs: max|alpha p| (PcgScal::step)
    if (it < 0) return;
    const double bm = block_max_256((double)mx, lds);
    const double bs = block_max_256((double)mxs, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        if (bm > 0.0) atomic_max_nonneg(sc.rmax(it) + sc.my_slot(), bm);
        if (bs > 0.0) atomic_max_nonneg(sc.step(it) + sc.my_slot(), bs);
    }
}
// stop test and stall guard of iteration it >= 0; true: the launch returns (every thread of the block calls this)
__device__ __forceinline__ bool d_vmg_stop_test(const PcgScal &sc, int it, double *lds) {
    const double res = d_fold_max(sc, sc.rmax(it), lds);
    const bool first = blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0;
    if (d_pass(sc, res) && d_steps_small(sc, it, lds)) {   // (with the velocity criterion where the loop carries one: PcgScal::vel_tol)
        if (first) *sc.conv = it;
        return true;
    }
    if (sc.best) {   // stall guard, as in k_pcg_update (pcg_common.h: PcgScal::best)
        const double bestNow = *sc.best;
        if (bestNow <= (sc.stall_below > 0.0 ? sc.stall_below : 100.0 * sc.tol) && res > sc.stall_ratio * bestNow) {
            if (first) { *sc.stalled = 1; *sc.conv = it; }
            return true;
        }
        // No progress: max|r| has not come down by 10 % for VMG_NO_PROGRESS iterations -- an fp32 recurrence that has lost its conjugacy hovers a
        // decade above the tolerance for hundreds of iterations (512 x 256 x 256 sheet, one solve in ~500: 700 iterations at 1.7e-5 max|rhs| where the
        // same system converges in ~100 on other runs).  The solve stops as "stalled"; where it can (fp32 bricks) the caller restarts it from the
        // fp64 residual.  Every block decides alike whichever of the first block's updates it sees: bestIt is written before best, and a block
        // that still sees the old best sees progress through `res` itself.
        if (sc.bestIt) {
            const int bi = *sc.bestIt;
            if (res >= 0.9 * bestNow && it - bi > VMG_NO_PROGRESS) {
                if (first) { *sc.stalled = 1; *sc.conv = it; }
                return true;
            }
            if (first && res < 0.9 * bestNow) { *sc.bestIt = it; __threadfence(); }
        }
        if (first && res < bestNow) *sc.best = res;
    }
    return false;
}

// ---- level 0 in the brick layout (k_viscosity_brick.hip): the vector kernels of the PCG loop, one wave per listed brick, one lane per index
__device__ __forceinline__ int d_bvmg_iter(const PcgScal &sc, int it_arg) { return it_arg == IT_DEVICE ? *sc.itB : it_arg; }
// bounding box of the listed bricks' indices
__global__ __launch_bounds__(256) void k_bvmg_bbox(const int *__restrict__ bricks, int nb, Lay LB, int *__restrict__ box) {
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
    for (int e = blockIdx.x * 256 + (int)(threadIdx.y * 64 + threadIdx.x); e < nb; e += gridDim.x * 256) {
        int i, j, k;
        d_brick_ijk(LB, bricks[e], 0, i, j, k);
        const int p0[3] = {i, j, k}, p1[3] = {i + 8, j + 4, k + 2};
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], p0[a]); hi[a] = max(hi[a], p1[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = -(int)wave_max((double)(-lo[a])); hi[a] = (int)wave_max((double)hi[a]);
        if ((threadIdx.x & 63) == 0) { atomicMin(box + a, lo[a]); atomicMax(box + 3 + a, hi[a]); }
    }
}
// zero `narr` brick-layout arrays (array a at base + a * per) on the listed bricks
__global__ __launch_bounds__(256) void k_bvmg_zero(const int *__restrict__ bricks, int nb, float *__restrict__ base, size_t per, int narr) {
    for (int e = (int)blockIdx.x * 4 + (int)threadIdx.y; e < nb; e += (int)gridDim.x * 4) {
        const size_t a = ((size_t)bricks[e] << 6) + threadIdx.x;
        for (int q = 0; q < narr; q++) base[(size_t)q * per + a] = 0.0f;
    }
}
// x += alpha p ; r -= alpha q ; rmax(it) ; z = omega r/d      (k_vpcg_xr of k_viscosity_mg_geo.inc)
__global__ __launch_bounds__(256) void k_bvpcg_xr(const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask, Vec3p d, Vec3p x, Vec3p r, Vec3p p, Vec3p q,
                                                  Vec3p z, float omega, PcgScal sc, int it_arg) {
    BrickWalkV w;
    w.begin(bricks, nb, mask);
    if (*sc.conv >= 0) return;
    const int it = d_bvmg_iter(sc, it_arg);
    if (it >= sc.cap) return;
    __shared__ double lds[8];
    double alpha_d = 0.0;
    if (it >= 0) {
        double f[4];
        d_fold_sums(sc, sc.sig(it), sc.a(it), nullptr, nullptr, f, lds);
        alpha_d = f[1] != 0.0 ? f[0] / f[1] : 0.0;
    }
    const float alpha = (float)alpha_d;
    float mx = 0.0f, mxs = 0.0f;
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, mask);
        if (m == 0u) continue;
        // every load of the lane first (one round trip per group of bricks, not one per component)
        Vec<float, 4> dd[3], rr[3], xx[3], pp[3], qq[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const bool on = BrickWalkV::any(m, c);
            dd[c] = on ? ldv<4>(d.p[c] + a) : Vec<float, 4>{};
            rr[c] = on ? ldv<4>(r.p[c] + a) : Vec<float, 4>{};
            xx[c] = (on && it >= 0) ? ldv<4>(x.p[c] + a) : Vec<float, 4>{};
            pp[c] = (on && it >= 0) ? ldv<4>(p.p[c] + a) : Vec<float, 4>{};
            qq[c] = (on && it >= 0) ? ldv<4>(q.p[c] + a) : Vec<float, 4>{};
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!BrickWalkV::any(m, c)) continue;
            Vec<float, 4> zz;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool row = BrickWalkV::row(m, c, e) && dd[c].v[e] != 0.0f;
                if (it >= 0 && row) {
                    xx[c].v[e] += alpha * pp[c].v[e];
                    if ((m >> (8 * e + 3 + c)) & 1u) mxs = fmaxf(mxs, fabsf(alpha * pp[c].v[e]));   // (rows whose velocity the substep uses: k_visc_setup)
                    rr[c].v[e] = (float)((double)rr[c].v[e] - alpha_d * (double)qq[c].v[e]);
                    mx = fmaxf(mx, fabsf(rr[c].v[e]));
                }
                zz.v[e] = row ? omega * (rr[c].v[e] / dd[c].v[e]) : 0.0f;
            }
            if (it >= 0) { stv(x.p[c] + a, xx[c]); stv(r.p[c] + a, rr[c]); }
            stv(z.p[c] + a, zz);
        }
    }
    d_vmg_publish_max(sc, it, mx, mxs, lds);
}
// z += P xc on the rows (the coarse correction of level 1)
__global__ __launch_bounds__(256) void k_bvmg_prolong_fine(const int *__restrict__ bricks, int nb, Lay LB, Lay C, const uint8_t *__restrict__ mask, Vec3p z, Vec3p xc,
                                                           PcgScal sc, int it_arg) {
    BrickWalk w;
    w.begin(bricks, nb, mask);
    if (*sc.conv >= 0 || d_bvmg_iter(sc, it_arg) >= sc.cap) return;
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, mask);
        if (m == 0u) continue;
        int p[3];
        d_brick_ijk(LB, (int)(a >> 6), (int)threadIdx.x, p[0], p[1], p[2]);
        float add[3], zz[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {   // the loads of all three components first
            add[c] = 0.0f; zz[c] = 0.0f;
            if (!((m >> c) & 1u)) continue;
            int Q[2][3];
            float wt[2];
            const int n = d_parents(c, p, Q, wt);
            zz[c] = z.p[c][a];
            add[c] = wt[0] * xc.p[c][cidx(C, Q[0][0], Q[0][1], Q[0][2])];
            if (n == 2) add[c] += wt[1] * xc.p[c][cidx(C, Q[1][0], Q[1][1], Q[1][2])];
        }
#pragma unroll
        for (int c = 0; c < 3; c++)
            if ((m >> c) & 1u) z.p[c][a] = zz[c] + add[c];
    }
}
// stop test on rmax(it) ; beta = sig(it+1)/sig(it) ; p = z + beta p          (it = -1: p = z)
__global__ __launch_bounds__(256) void k_bvpcg_p(const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask, Vec3p z, Vec3p p, PcgScal sc, int it_arg) {
    BrickWalkV w;
    w.begin(bricks, nb, mask);
    if (*sc.conv >= 0) return;
    const int it = d_bvmg_iter(sc, it_arg);
    if (it >= sc.cap) return;
    __shared__ double lds[8];
    float beta = 0.0f;
    if (it >= 0) {
        if (d_vmg_stop_test(sc, it, lds)) return;
        double f[4];
        d_fold_sums(sc, sc.sig(it + 1), sc.sig(it), nullptr, nullptr, f, lds);
        beta = f[1] != 0.0 ? (float)(f[0] / f[1]) : 0.0f;
    }
    while (w.valid()) {
        const size_t a = w.a;
        const unsigned m = w.m;
        w.next(bricks, nb, mask);
        if (m == 0u) continue;
        Vec<float, 4> zz[3], pp[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const bool on = BrickWalkV::any(m, c);
            zz[c] = on ? ldv<4>(z.p[c] + a) : Vec<float, 4>{};
            pp[c] = (on && it >= 0) ? ldv<4>(p.p[c] + a) : Vec<float, 4>{};
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!BrickWalkV::any(m, c)) continue;
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (BrickWalkV::row(m, c, e)) pp[c].v[e] = zz[c].v[e] + beta * pp[c].v[e];
            stv(p.p[c] + a, pp[c]);
        }
    }
    if (it_arg == IT_DEVICE && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) *sc.itA = it + 1;
}

// ---- the kernels that walk the solver's tile list: once per tile geometry (pcg_geo.inc)
namespace g16 {
constexpr int ROWL = 16;
#include "pcg_geo.inc"
#include "k_viscosity_mg_geo.inc"
}  // namespace g16
namespace g64 {
constexpr int ROWL = 64;
#include "pcg_geo.inc"
#include "k_viscosity_mg_geo.inc"
}  // namespace g64

struct VmgState {
    std::vector<VLevel> lev;     // coarse levels 1..
    float *za[3] = {nullptr, nullptr, nullptr}, *zb[3] = {nullptr, nullptr, nullptr}, *t0[3] = {nullptr, nullptr, nullptr};   // level 0: the sweeps' two iterates, the residual
    std::vector<void *> allocs;
    std::vector<std::pair<void *, size_t>> vecBlocks;    // (base, bytes) of every level's vector storage, zeroed per solve
    int *d_box = nullptr;
    int *d_rowlist = nullptr, *d_rowcnt = nullptr, *d_stripCount = nullptr;   // coarsest level's rows; per level the number of listed strips
    VLevelDev *d_lev = nullptr;  // the level descriptors in device memory (this solve's boxes), h_lev their pinned staging copy
    VLevelDev *h_lev = nullptr;
    int tailFirst = 0;           // index into lev of the first level the tail kernel handles (this solve)
    bool coarsestInLds = false;  // k_vmg_coarsest_rows' verdict for this solve (the level's box and rows fit one workgroup's LDS and threads)
    // Block contexts (flipv_comm.h): levels lev[0 .. globalFrom) are the rank's own (its rows only, couplings across the cuts dropped, nothing
    // exchanged: block-Jacobi); from lev[globalFrom] on the hierarchy is the GLOBAL one, held and cycled redundantly by every rank -- its operator is
    // the sum over the ranks of their Galerkin contributions (one all-reduce per solve), its right-hand side the sum of their restricted
    // residuals (one all-reduce per iteration).  -1: no global level (single domain, or switched off)
    int globalFrom = -1;
    // DISTRIBUTED coarse levels (round 4): lev[0 .. nDist) are cycled by their owners only -- rank r owns the indices P of such a level whose fine
    // ancestor 2^(l+1) P it owns (own[l]: a tiling of the level like the blocks tile the domain); a sweep's input vector gets a 1-entry halo copy first
    // (fv_halo_level), right-hand sides and Galerkin sums that children on both sides of a cut contribute to are halo-REDUCED to the owner -- exactly what
    // happens on level 0.  The global, redundantly cycled hierarchy then starts at lev[nDist] = lev[globalFrom]: its operator (once per solve) and
    // right-hand side (once per iteration) are 1/8 per distributed level of what level 1 would all-reduce, and nobody cycles level 1 of the WHOLE
    // domain any more (at 512^3 on 8 ranks that level is as much work as a rank's share of the fine level).  nDist is 0 or 1.
    int nDist = 0;
    Box3 own[VMG_MAX_LEVELS], rbox[VMG_MAX_LEVELS];   // owned box of a distributed level; the same widened by one entry (clipped to the level): where partial sums are formed
    int rc = 0;                  // a communication error inside the V-cycle (checked by the loop around it)
    // Under a communicator the V-cycle cannot be one replayed graph (its collectives are host calls of the backend), but everything between the
    // right-hand-side all-reduce of the first global level and the halo copy before the first post-sweep is kernels only -- ~17 dependent launches,
    // most of them at the floor of a launch: that segment is captured once per solve and replayed every iteration (midExec: the cached executable,
    // owned by flipv_context::geCache; midReady: captured for this solve's hierarchy)
    hipGraphExec_t midExec = nullptr;
    bool midReady = false;
    bool listRhs = true;         // the first global level's right-hand side travels as its listed bricks (every rank holds the same list), not as the union box
    float *stage = nullptr;      // dense staging buffer of the global hierarchy's all-reduces (grown on demand)
    size_t stageCap = 0;
    double *d_gbox = nullptr;    // 6 ints per rank as doubles: the ranks' boxes on lev[globalFrom], merged by a sum all-reduce over disjoint slots
    bool ready = false;          // every allocation of vmg_alloc_state succeeded
    float w[2] = {VMG_W_DEFAULT[0], VMG_W_DEFAULT[1]};   // this solve's smoother weights (VMG_W)
    float chebTab[64] = {};      // the Chebyshev table for degree chebM (0: the coarsest level is swept with the fixed damping)
    int chebM = 0;
    int minDim = 0;              // the coarsest level's longest axis the hierarchy was allocated for (flipv_params.viscosity_mg_min_dim)
    void *fineVecs = nullptr;    // the fine level's three sweep vectors (zeroed every solve; the coarse levels' only with a new hierarchy)
    // brick layout: the sweep vectors are non-zero only on the bricks the previous solve listed, so that list (a copy) is what gets zeroed, not
    // the 9 whole arrays (705 MB at 256^3: two memsets of ~100 us per solve); -1: nothing known, zero everything
    int *prevBricks = nullptr;
    int prevCount = -1;
    size_t fineStride = 0;       // floats between consecutive fine vectors
    size_t fineVecBytes = 0;
    ~VmgState() { for (void *p : allocs) (void)hipFree(p); if (stage) (void)hipFree(stage); if (h_lev) (void)hipHostFree(h_lev); }
};

static int vmg_alloc(flipv_context *c, VmgState *s, size_t per, size_t count, float **base) {
    const size_t tot = per * count;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, tot * sizeof(float));
    if (e != hipSuccess) { c->err = std::string("hipMalloc(viscosity multigrid): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
    s->allocs.push_back(q);
    HIPCHK(c, hipMemsetAsync(q, 0, tot * sizeof(float), c->stream));
    *base = (float *)q;
    return FLIPV_OK;
}

static Vec3p v3(float *const p[3]) { Vec3p v; v.p[0] = p[0]; v.p[1] = p[1]; v.p[2] = p[2]; return v; }
static VLevelDev dev_of(const VLevel &l) {
    VLevelDev d{};
    d.L = l.L;
    for (int c = 0; c < 3; c++) for (int s = 0; s < VS; s++) d.coef[c][s] = l.coef[c][s];
    for (int c = 0; c < 3; c++) for (int w = 0; w < VH; w++) d.ch[c][w] = l.ch[c][w];
    for (int c = 0; c < 3; c++) d.dd[c] = l.dd[c];
    d.x = v3(l.x); d.y = v3(l.y); d.b = v3(l.b); d.t = v3(l.t);
    d.box = l.box;
    d.strips = l.strips; d.nstrips = l.nstrips;
    d.rowlist = nullptr; d.rowcnt = nullptr;
    return d;
}
static long box_positions(const Box3 &b) { return (long)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]); }
// sum over the ranks of `narr` grids of level A (the first at g0, consecutive ones A.per apart) inside the level's box, through the dense staging buffer
static int vmg_stage_reserve(flipv_context *c, VmgState *s, size_t tot) {
    if (tot > s->stageCap) {
        FV_SYNC(c);
        if (s->stage) (void)hipFree(s->stage);
        s->stage = nullptr; s->stageCap = 0;
        hipError_t e = hipMalloc((void **)&s->stage, tot * sizeof(float));
        if (e != hipSuccess) { c->err = std::string("hipMalloc(viscosity multigrid staging): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        s->stageCap = tot;
    }
    return FLIPV_OK;
}
static int vmg_allreduce_box(flipv_context *c, VmgState *s, const VLevel &A, float *g0, int narr) {
    const size_t n = (size_t)box_positions(A.box), tot = n * (size_t)narr;
    const int rc0 = vmg_stage_reserve(c, s, tot);
    if (rc0) return rc0;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)narr);
    hipLaunchKernelGGL(k_vmg_box_pack, grid, dim3(256), 0, c->stream, A.L, A.box, g0, A.per, s->stage, 0);
    const int rc = fv_allreduce_f32(c, s->stage, tot);
    if (rc) return rc;
    c->commBytesSetup += (double)tot * sizeof(float);
    hipLaunchKernelGGL(k_vmg_box_pack, grid, dim3(256), 0, c->stream, A.L, A.box, g0, A.per, s->stage, 1);
    return FLIPV_OK;
}

}  // namespace

void fv_vmg_free(flipv_context *c) {
    delete (VmgState *)c->vmgState;
    c->vmgState = nullptr;
}

static int vmg_min_dim(const flipv_context *c) { return c->prm.viscosity_mg_min_dim > 0 ? c->prm.viscosity_mg_min_dim : VMG_MIN_DIM; }

// the level structure and its storage (once per context; again if flipv_params.viscosity_mg_min_dim changes)
static int vmg_alloc_state(flipv_context *c) {
    VmgState *s = (VmgState *)c->vmgState;
    int rc;
    if (s && (!s->ready || s->minDim != vmg_min_dim(c))) {   // an earlier attempt ran out of memory half way, or another depth is asked for: start over
        FV_SYNC(c);
        fv_vmg_free(c);
        s = nullptr;
    }
    if (!s) {
        SlotTables T;
        build_slot_tables(&T);
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(ST), &T, sizeof(T)));
        s = new VmgState();
        c->vmgState = s;
        s->minDim = vmg_min_dim(c);
        HIPCHK(c, hipMalloc((void **)&s->d_box, 8 * sizeof(int)));
        s->allocs.push_back(s->d_box);
        HIPCHK(c, hipMalloc((void **)&s->d_rowlist, (3 * 1024 + 4 + VMG_MAX_LEVELS) * sizeof(int)));
        s->allocs.push_back(s->d_rowlist);
        s->d_rowcnt = s->d_rowlist + 3 * 1024;
        s->d_stripCount = s->d_rowcnt + 4;
        HIPCHK(c, hipMalloc((void **)&s->d_gbox, 6 * 32 * sizeof(double)));
        s->allocs.push_back(s->d_gbox);
        HIPCHK(c, hipMalloc((void **)&s->d_lev, VMG_MAX_LEVELS * sizeof(VLevelDev)));
        s->allocs.push_back(s->d_lev);
        HIPCHK(c, hipHostMalloc((void **)&s->h_lev, VMG_MAX_LEVELS * sizeof(VLevelDev)));
        float *base;
        const size_t per0 = c->L.guard + c->solverCap + c->L.guard;   // room for either layout of level 0
        if ((rc = vmg_alloc(c, s, per0, 9, &base))) return rc;
        s->fineVecs = base; s->fineVecBytes = 9 * per0 * sizeof(float);
        s->fineStride = per0;
        HIPCHK(c, hipMalloc((void **)&s->prevBricks, ((size_t)c->brickCap + 64) * sizeof(int)));
        s->allocs.push_back(s->prevBricks);
        for (int m = 0; m < 3; m++) {
            s->za[m] = base + (size_t)m * per0 + c->L.guard;
            s->zb[m] = base + (size_t)(3 + m) * per0 + c->L.guard;
            s->t0[m] = base + (size_t)(6 + m) * per0 + c->L.guard;
        }
        Lay F = c->L;
        while (true) {
            const int mx = F.I > F.J ? (F.I > F.K ? F.I : F.K) : (F.J > F.K ? F.J : F.K);
            if (mx <= s->minDim || (int)s->lev.size() >= VMG_MAX_LEVELS) break;
            VLevel l;
            l.L = coarse_lay(F);
            const size_t per = l.L.n + 2 * l.L.guard;
            float *cb, *vb, *hb;
            if ((rc = vmg_alloc(c, s, per, 3 * VS, &cb)) || (rc = vmg_alloc(c, s, per, 12, &vb)) || (rc = vmg_alloc(c, s, per, 3 * (VH + 1), &hb))) return rc;
            s->vecBlocks.push_back({vb, per * 12 * sizeof(float)});
            l.per = per;
            {
                const size_t nstr = l.L.n / 64;   // bricks of the level
                HIPCHK(c, hipMalloc((void **)&l.strips, 2 * nstr * sizeof(int)));
                s->allocs.push_back(l.strips);
                l.stripFlag = l.strips + nstr;
            }
            for (int m = 0; m < 3; m++) {
                for (int q = 0; q < VS; q++) l.coef[m][q] = cb + (size_t)(m * VS + q) * per + l.L.guard;
                for (int w = 0; w < VH; w++) l.ch[m][w] = (unsigned *)(hb + (size_t)(m * VH + w) * per + l.L.guard);
                l.dd[m] = hb + (size_t)(3 * VH + m) * per + l.L.guard;
                l.x[m] = vb + (size_t)m * per + l.L.guard;
                l.y[m] = vb + (size_t)(3 + m) * per + l.L.guard;
                l.b[m] = vb + (size_t)(6 + m) * per + l.L.guard;
                l.t[m] = vb + (size_t)(9 + m) * per + l.L.guard;
            }
            s->lev.push_back(l);
            F = l.L;
        }
        s->ready = true;
    }
    return FLIPV_OK;
}
// Allocate the hierarchy ahead of its first use (the allocation -- 1.3 GB at 256^3 -- and its memsets then do not land in a timed substep)
int fv_vmg_prepare(flipv_context *c) { return vmg_alloc_state(c); }

static bool vmg_brick(const flipv_context *c) { return c->vLayout == VLAYOUT_BRICK; }

static int vmg_setup(flipv_context *c, VmgState **out) {
    int rc = vmg_alloc_state(c);
    if (rc) return rc;
    VmgState *s = (VmgState *)c->vmgState;
    const bool brick = vmg_brick(c);
    s->midReady = false;   // (a new hierarchy: new grids and lists in the captured segment)
    // the fine level's sweep vectors must be zero wherever there is no row (the SpMV and the restriction read neighbours unmasked)
    if (brick && s->prevCount >= 0) {
        if (s->prevCount > 0) hipLaunchKernelGGL(k_bvmg_zero, dim3(cdiv(s->prevCount, 4) < 2048 ? cdiv(s->prevCount, 4) : 2048), dim3(64, 4, 1), 0, c->stream, (const int *)s->prevBricks, s->prevCount, s->za[0], s->fineStride, 9);
    } else HIPCHK(c, hipMemsetAsync(s->fineVecs, 0, s->fineVecBytes, c->stream));
    if (brick) {
        HIPCHK(c, hipMemcpyAsync(s->prevBricks, c->brickList, (size_t)c->nBricks * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
        s->prevCount = c->nBricks;
    } else s->prevCount = -1;
    // ---- a new hierarchy for every solve (a kept one over-corrects after a change of dt): the box of the rows, level by level
    {
        HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)s->d_box, 0x7fffffff, 3, c->stream));
        HIPCHK(c, hipMemsetAsync(s->d_box + 3, 0, 3 * sizeof(int), c->stream));
        if (brick) hipLaunchKernelGGL(k_bvmg_bbox, dim3(64), dim3(64, 4, 1), 0, c->stream, (const int *)c->brickList, c->nBricks, c->LB, s->d_box);
        else GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vmg_tile_bbox, dim3(64), dim3(64, 4, 1), 0, c->stream, c->tileListV, c->nActiveV, c->tgV, s->d_box));
        int hb[6];
        FV_READ(c, hb, s->d_box, sizeof(hb));
        FV_SYNC(c);
        Box3 fb;
        const int ext0[3] = {c->L.I + 1, c->L.J + 1, c->L.K + 1};
        bool noRows = false;   // (a rank of a block decomposition without liquid: its one-position stand-in box must not enter the union over the ranks)
        for (int a = 0; a < 3; a++) {   // (tiles hang over the end of the lattices and, on a block context, of the rank's own box: the rows do not)
            fb.lo[a] = hb[a] < c->L.olo[a] ? c->L.olo[a] : hb[a];
            fb.hi[a] = hb[3 + a] > ext0[a] ? ext0[a] : hb[3 + a];
            if (fb.hi[a] > c->L.ohi[a]) fb.hi[a] = c->L.ohi[a];
            if (fb.hi[a] <= fb.lo[a]) { fb.lo[a] = c->L.olo[a]; fb.hi[a] = c->L.olo[a] + 1; noRows = true; }
        }
        for (size_t l = 0; l < s->lev.size(); l++) {
            VLevel &A = s->lev[l];
            const int ext[3] = {A.L.I + 1, A.L.J + 1, A.L.K + 1};
            for (int a = 0; a < 3; a++) {   // parents of fine index p: p >> 1 and (p + 1) >> 1
                A.box.lo[a] = fb.lo[a] >> 1;
                A.box.hi[a] = (fb.hi[a] >> 1) + 1;
                if (A.box.hi[a] > ext[a]) A.box.hi[a] = ext[a];
                if (A.box.hi[a] <= A.box.lo[a]) A.box.hi[a] = A.box.lo[a] + 1;
            }
            fb = A.box;
        }
        // block contexts: the levels every rank cycles for itself -- DISTRIBUTED ones, lev[0 .. nDist), where the system is large (VmgState::nDist; the decision
        // is taken from the all-gathered row count, so every rank takes it alike) -- and from lev[nDist] on the GLOBAL hierarchy (VmgState::globalFrom): its
        // box is the union of the ranks' boxes
        const bool commMg = c->comm && !c->prm.multigrid_rank_local && !s->lev.empty();
        s->nDist = 0;
        if (commMg && c->comm->nranks > 1 && s->lev.size() >= 2) {
            const int want = c->prm.multigrid_distributed_levels;   // 0 = by size, 1 = level 1 distributed, -1 = none
            if (want > 0 || (want == 0 && c->vRowsAll > 4.5e6)) s->nDist = 1;   // (4.5e6 rows ~ a 512^3-class liquid: level 1 has then > ~3 000 bricks of rows)
        }
        s->globalFrom = commMg ? s->nDist : -1;
        for (size_t l = 0; l < s->lev.size(); l++) {   // a global level's Lay owns all of it; a distributed level's owned box: the indices whose fine ancestor the rank owns
            Lay &LL = s->lev[l].L;
            LL.olo[0] = LL.olo[1] = LL.olo[2] = 0; LL.ohi[0] = LL.PX; LL.ohi[1] = LL.PY; LL.ohi[2] = LL.PZ;
        }
        {
            int lo[3] = {c->L.olo[0], c->L.olo[1], c->L.olo[2]}, hi[3] = {c->L.ohi[0], c->L.ohi[1], c->L.ohi[2]};
            for (int l = 0; l < s->nDist; l++) {
                VLevel &A = s->lev[l];
                const int ext[3] = {A.L.I + 1, A.L.J + 1, A.L.K + 1};
                for (int a = 0; a < 3; a++) {
                    lo[a] = (lo[a] + 1) >> 1; hi[a] = (hi[a] + 1) >> 1;   // index P is the rank's iff 2 P is
                    s->own[l].lo[a] = lo[a]; s->own[l].hi[a] = hi[a] > ext[a] ? ext[a] : hi[a];
                    if (s->own[l].hi[a] < s->own[l].lo[a]) s->own[l].hi[a] = s->own[l].lo[a];
                    s->rbox[l].lo[a] = s->own[l].lo[a] > 0 ? s->own[l].lo[a] - 1 : 0;
                    s->rbox[l].hi[a] = s->own[l].hi[a] + 1 > ext[a] ? ext[a] : s->own[l].hi[a] + 1;
                    A.L.olo[a] = s->own[l].lo[a]; A.L.ohi[a] = s->own[l].hi[a];
                }
            }
        }
        if (s->globalFrom >= 0) {
            const int nr = c->comm->nranks, me = c->comm->rank;
            std::vector<double> hbx((size_t)6 * nr, 0.0);
            VLevel &G = s->lev[s->globalFrom];
            for (int a = 0; a < 3; a++) { hbx[(size_t)6 * me + a] = noRows ? 1e9 : G.box.lo[a]; hbx[(size_t)6 * me + 3 + a] = noRows ? -1e9 : G.box.hi[a]; }
            HIPCHK(c, hipMemcpyAsync(s->d_gbox, hbx.data(), hbx.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            if ((rc = fv_allreduce_scalars(c, s->d_gbox, hbx.size()))) return rc;
            FV_READ(c, hbx.data(), s->d_gbox, hbx.size() * sizeof(double));
            FV_SYNC(c);
            bool first = true;
            for (int r = 0; r < nr; r++) {
                if (hbx[(size_t)6 * r] > 1e8) continue;   // a rank without rows
                for (int a = 0; a < 3; a++) {
                    const int lo = (int)hbx[(size_t)6 * r + a], hi = (int)hbx[(size_t)6 * r + 3 + a];
                    if (first || lo < G.box.lo[a]) G.box.lo[a] = lo;
                    if (first || hi > G.box.hi[a]) G.box.hi[a] = hi;
                }
                first = false;
            }
            Box3 gb = G.box;
            for (size_t l = (size_t)s->globalFrom + 1; l < s->lev.size(); l++) {
                VLevel &A = s->lev[l];
                const int ext[3] = {A.L.I + 1, A.L.J + 1, A.L.K + 1};
                for (int a = 0; a < 3; a++) {
                    A.box.lo[a] = gb.lo[a] >> 1;
                    A.box.hi[a] = (gb.hi[a] >> 1) + 1;
                    if (A.box.hi[a] > ext[a]) A.box.hi[a] = ext[a];
                    if (A.box.hi[a] <= A.box.lo[a]) A.box.hi[a] = A.box.lo[a] + 1;
                }
                gb = A.box;
            }
        }
        for (int l = 0; l < s->nDist; l++) s->lev[l].box = s->own[l];   // (what the sweeps, the lists and the packing of a distributed level cover: the rank's own indices)
        // the tail: the coarsest levels whose boxes are small enough for one workgroup (the last level always)
        s->tailFirst = (int)s->lev.size() - 1;
        while (s->tailFirst > 0 && (int)s->lev.size() - (s->tailFirst - 1) <= VMG_TAIL_MAX && box_positions(s->lev[s->tailFirst - 1].box) <= VMG_TAIL_POS) s->tailFirst--;
        if (s->globalFrom >= 0 && s->tailFirst < s->globalFrom) s->tailFirst = s->globalFrom;   // (the rank's own levels are launches: the all-reduce sits between them and the tail)
    }
    for (auto &b : s->vecBlocks) HIPCHK(c, hipMemsetAsync(b.first, 0, b.second, c->stream));   // the coarse levels' vectors: zero off their rows
    // this solve's coarse operators, level by level, as gathers that write every entry of the level's box (rows or not): nothing has to
    // be zeroed first, and what lies outside the box is never looked at.  The rows' own volumes are those of the operator the solve
    // applies: the exact one, or the reference's float-rounded one (vr*, k_viscosity.hip: d_ref_volume) -- the defect is a diagonal term
    // and goes through the Galerkin product like the volume itself.
    if (s->globalFrom >= 0) {   // the neighbours' row masks one entry into the halo: a fine row at a cut face then carries its entries across the cut into the Galerkin sums
        const HaloArray hm[1] = {{brick ? (void *)c->vMaskB : (void *)c->vRowMask, 1, brick ? 1 : 0}};
        if ((rc = fv_halo_copy(c, hm, 1, 1))) return rc;
    }
    if (!s->lev.empty()) {
        FineOp A;
        A.swz = c->vSwz;
        A.brick = brick ? 1 : 0;
        A.LB = c->LB;
        A.vm[0] = c->vOperatorExact ? c->vmU : c->vrU; A.vm[1] = c->vOperatorExact ? c->vmV : c->vrV; A.vm[2] = c->vOperatorExact ? c->vmW : c->vrW;
        A.fC = c->fC; A.fE[0] = c->fEU; A.fE[1] = c->fEV; A.fE[2] = c->fEW;
        A.mask = brick ? c->vMaskB : c->vRowMask;
#define CGRID(B) dim3(cdiv((B).hi[0] - (B).lo[0], 64), cdiv((B).hi[1] - (B).lo[1], 4), 3u * (unsigned)((B).hi[2] - (B).lo[2])), dim3(64, 4, 1)
        // (on the first global level every rank gathers over the union box: children outside its own finer box are skipped, so what it
        // writes is its share of every row -- zero where it has none -- and the sum over the ranks is the single-domain Galerkin operator)
        if (s->nDist > 0) {   // level 1 distributed: the rank's share of the rows of its own box AND of the ring around it, the ring's shares then go to their owners
            VLevelDev D0 = dev_of(s->lev[0]);
            D0.box = s->rbox[0];
            hipLaunchKernelGGL(k_vmg_rap_gather_fine, CGRID(s->rbox[0]), 0, c->stream, A, c->L, D0);
            for (int g0 = 0; g0 < 3 * VS; g0 += 6) {
                float *arr[6];
                const int n = 3 * VS - g0 < 6 ? 3 * VS - g0 : 6;
                for (int q = 0; q < n; q++) arr[q] = s->lev[0].coef[0][0] + (size_t)(g0 + q) * s->lev[0].per;
                if ((rc = fv_halo_level(c, s->lev[0].L, s->own[0].lo, s->own[0].hi, arr, n, 1, 1))) return rc;
            }
            const long ring = box_positions(s->rbox[0]) - box_positions(s->own[0]);
            c->commBytesSetup += (double)(ring > 0 ? ring : 0) * 3 * VS * sizeof(float);   // (point to point, to the <= 26 neighbours: what leaves this rank)
        } else hipLaunchKernelGGL(k_vmg_rap_gather_fine, CGRID(s->lev[0].box), 0, c->stream, A, c->L, dev_of(s->lev[0]));
        if (s->globalFrom == 0 && (rc = vmg_allreduce_box(c, s, s->lev[0], s->lev[0].coef[0][0], 3 * VS))) return rc;
        for (size_t l = 0; l + 1 < s->lev.size(); l++) {
            hipLaunchKernelGGL(k_vmg_rap_gather, CGRID(s->lev[l + 1].box), 0, c->stream, dev_of(s->lev[l]), dev_of(s->lev[l + 1]));
            if (s->globalFrom == (int)l + 1 && (rc = vmg_allreduce_box(c, s, s->lev[l + 1], s->lev[l + 1].coef[0][0], 3 * VS))) return rc;
        }
#undef CGRID
    }
    // where the rows are on the levels that run as launches (strip lists) and on the coarsest one (row lists); then the level
    // descriptors go to the device
    if (!s->lev.empty()) {
        for (int l = 0; l < s->tailFirst; l++) {
            VLevel &A = s->lev[l];
            const int nstr = ((((A.box.hi[0] - 1 + 8) >> 3) - ((A.box.lo[0] + 8) >> 3)) + 1) * ((((A.box.hi[1] - 1 + 4) >> 2) - ((A.box.lo[1] + 4) >> 2)) + 1) *
                             ((((A.box.hi[2] - 1 + 2) >> 1) - ((A.box.lo[2] + 2) >> 1)) + 1);   // bricks of the box (d_brick_range)
            hipLaunchKernelGGL(k_vmg_strip_flags, dim3(cdiv(nstr, 4)), dim3(64, 4, 1), 0, c->stream, dev_of(A), A.stripFlag, nstr);
            hipLaunchKernelGGL(k_vmg_strip_compact, dim3(1), dim3(1024), 0, c->stream, (const int *)A.stripFlag, nstr, A.strips, s->d_stripCount + l);
        }
        for (size_t l = 0; l < s->lev.size(); l++) {   // every level's rows in the packed form the cycle reads (the LDS-resident level loads the fp32 grids)
            const VLevel &A = s->lev[l];
            const int nstr = ((((A.box.hi[0] - 1 + 8) >> 3) - ((A.box.lo[0] + 8) >> 3)) + 1) * ((((A.box.hi[1] - 1 + 4) >> 2) - ((A.box.lo[1] + 4) >> 2)) + 1) *
                             ((((A.box.hi[2] - 1 + 2) >> 1) - ((A.box.lo[2] + 2) >> 1)) + 1);
            VLevelDev Ad = dev_of(A);
            Ad.packed = c->vmgPackedRows;
            hipLaunchKernelGGL(k_vmg_pack, dim3(cdiv(nstr, 4), 3), dim3(64, 4, 1), 0, c->stream, Ad, nstr);
        }
        hipLaunchKernelGGL(k_vmg_coarsest_rows, dim3(1), dim3(1024), 0, c->stream, dev_of(s->lev.back()), s->d_rowlist, s->d_rowcnt);
        int counts[VMG_MAX_LEVELS];
        int fitsLds = 0;
        if (s->tailFirst > 0) FV_READ_JOBS(c, FV_JOB(&fitsLds, s->d_rowcnt + 3, sizeof(int)), FV_JOB(counts, s->d_stripCount, s->tailFirst * sizeof(int)));
        else FV_READ(c, &fitsLds, s->d_rowcnt + 3, sizeof(int));
        FV_SYNC(c);
        s->coarsestInLds = fitsLds != 0;
        s->w[0] = c->prm.viscosity_mg_omega_first > 0.0f ? c->prm.viscosity_mg_omega_first : VMG_W_DEFAULT[0];
        s->w[1] = c->prm.viscosity_mg_omega_second > 0.0f ? c->prm.viscosity_mg_omega_second : VMG_W_DEFAULT[1];
        {   // Chebyshev weights of the coarsest level when the sweep count is a power of two (k_vmg_tail); any other count: plain damped Jacobi
            const int m = c->vmgSweeps > 0 ? c->vmgSweeps : VMG_COARSEST_SWEEPS;
            const bool pow2 = m >= 4 && m <= 64 && (m & (m - 1)) == 0;
            if (!pow2) s->chebM = 0;
            else if (s->chebM != m) {
                std::vector<int> o(1, 0);   // Lebedev-Finogenov order of the roots: o(2n) = (j, 2n - 1 - j) for j in o(n)
                for (int n = 1; n < m; n *= 2) {
                    std::vector<int> t;
                    for (int j : o) { t.push_back(j); t.push_back(2 * n - 1 - j); }
                    o.swap(t);
                }
                for (int k = 0; k < m; k++) s->chebTab[k] = (float)cos(M_PI * (2.0 * o[k] + 1.0) / (2.0 * m));
                s->chebM = m;
            }
        }
        s->listRhs = true;
        if (s->globalFrom >= 0 && s->globalFrom < s->tailFirst) {   // the list-shaped right-hand-side exchange needs the same list on every rank
            const double mine[1] = {(double)counts[s->globalFrom]};
            double all[NSLOT];
            if ((rc = fv_allgather_f64(c, mine, 1, all))) return rc;
            for (int r = 0; r < c->comm->nranks; r++) if (all[r] != mine[0]) s->listRhs = false;   // (then: the box-shaped exchange)
        }
        for (size_t l = 0; l < s->lev.size(); l++) {
            s->lev[l].nstrips = (int)l < s->tailFirst ? counts[l] : 0;
            s->h_lev[l] = dev_of(s->lev[l]);
            s->h_lev[l].packed = c->vmgPackedRows;
            s->h_lev[l].w[0] = s->w[0]; s->h_lev[l].w[1] = s->w[1];
            memcpy(s->h_lev[l].cheb, s->chebTab, sizeof(s->chebTab));
            s->h_lev[l].rowlist = s->d_rowlist; s->h_lev[l].rowcnt = s->d_rowcnt;
        }
        HIPCHK(c, hipMemcpyAsync(s->d_lev, s->h_lev, s->lev.size() * sizeof(VLevelDev), hipMemcpyHostToDevice, c->stream));
    }
    if (c->prm.verbose) {
        (void)hipStreamSynchronize(c->stream);
        fprintf(stderr, "viscosity multigrid: %zu coarse levels, tail from %d, level 0 in the %s layout\n", s->lev.size(), s->tailFirst, brick ? "brick" : "plain");
        for (size_t l = 0; l < s->lev.size(); l++) { const Box3 &b = s->lev[l].box; fprintf(stderr, "  level %zu box [%d,%d) x [%d,%d) x [%d,%d), %d bricks\n", l + 1, b.lo[0], b.hi[0], b.lo[1], b.hi[1], b.lo[2], b.hi[2], s->lev[l].nstrips); }
        int rcnt[4] = {0, 0, 0, 0};
        (void)hipMemcpy(rcnt, s->d_rowcnt, sizeof(rcnt), hipMemcpyDeviceToHost);
        fprintf(stderr, "  coarsest level: %d / %d / %d rows per component, %s; the cycle reads %s rows\n", rcnt[0], rcnt[1], rcnt[2], rcnt[3] ? "in LDS" : "through global memory", c->vmgPackedRows ? "packed fp16" : "fp32");
    }
    HIPCHK(c, hipGetLastError());
    *out = s;
    return FLIPV_OK;
}

// The additive correction of the V-cycle on strongly coupled pairs of rows (k_viscosity.hip: k_visc_pairs_find): z += B^-1 r over every listed pair's 2 x 2 block,
// (r, z) adjusted by the same amount.  One small block; a row may sit in two pairs (a chain of three rows), hence the atomic adds.
__global__ __launch_bounds__(256) void k_vmg_pairs(const unsigned *__restrict__ count, const VPair *__restrict__ list, Vec3p r, Vec3p z, PcgScal sc, int it_arg, int sig_shift) {
    __shared__ double lds[8];
    bool stop;
    const int it = d_iter_spmv(sc, it_arg, stop);
    if (stop) return;
    unsigned n = *count;
    if (n > (unsigned)FV_PAIR_CAP) n = FV_PAIR_CAP;
    double acc = 0.0;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const VPair P = list[t];
        const int c0 = (int)(P.comps & 3u), c1 = (int)((P.comps >> 2) & 3u);
        const float r0 = r.p[c0][P.ir0], r1 = r.p[c1][P.ir1];
        const float d0 = P.i00 * r0 + P.i01 * r1, d1 = P.i01 * r0 + P.i11 * r1;
        atomicAdd(z.p[c0] + P.iz0, d0);
        atomicAdd(z.p[c1] + P.iz1, d1);
        acc += (double)r0 * (double)d0 + (double)r1 * (double)d1;
    }
    acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0 && sc.conv && acc != 0.0) atomicAdd(sc.sig(it + sig_shift) + sc.my_slot(), acc);
}

// z = M^-1 r (level 0) into zb, (r, z) into sig(it + sig_shift).  On entry za = omega r/d (k_vpcg_xr left it).
// it_arg = IT_DEVICE: the device-side iteration counter (inside the replayed graph)
static void vmg_vcycle(flipv_context *c, VmgState *s, const PcgScal &sc, int it_arg, int sig_shift) {
    const bool brick = vmg_brick(c);
    const int nb = brick ? fv_brick_grid(c, c->nBricks, 2048) : pcg_grid(c, c->nActiveV);
    const dim3 blk(64, 4, 1);
    const int it_spmv = it_arg == IT_DEVICE ? -1 : it_arg;   // the SpMV kernel's spelling of "device-side counter"
    float *dg[3] = {c->vDiagU, c->vDiagV, c->vDiagW};
    const int *conv = sc.conv;
    // with the global hierarchy the fine level is the single domain's too: every sweep reads its input with the neighbours' current values
    const int hl = brick ? 1 : 0;   // (HaloArray::lay)
    auto halo3 = [&](float *const v[3]) { if (s->globalFrom >= 0 && !s->rc) { const HaloArray h[3] = {{v[0], sizeof(float), hl}, {v[1], sizeof(float), hl}, {v[2], sizeof(float), hl}}; s->rc = fv_halo_copy(c, h, 3, 1); } };
    // A fine-level sweep under a communicator: the 1-entry halo of its input travels on the communication stream while the sweep runs over the bricks interior to the
    // rank's box; the bricks along the cut faces follow once it has arrived (brick layout: fv_build_bricks lists the interior bricks first; plane layouts: the exchange first).
    auto sweep = [&](float *const in[3], float *const out[3], int epi, float omega, int sshift) {
        const bool overlap = brick && s->globalFrom >= 0 && c->comm && c->comm->nranks > 1 && !c->prm.no_comm_overlap && c->nIntV > 0 && !s->rc;
        if (!overlap) { halo3(in); fv_visc_sweep_f32(c, in, out, epi, sc, it_spmv, omega, sshift); return; }
        const HaloArray h[3] = {{in[0], sizeof(float), 1}, {in[1], sizeof(float), 1}, {in[2], sizeof(float), 1}};
        s->rc = fv_halo_copy_begin(c, h, 3, 1);
        fv_brick_sweep_f32(c, in, out, epi, sc, it_spmv, omega, sshift, 0, c->nIntV);
        if (!s->rc) s->rc = fv_halo_wait(c);
        fv_brick_sweep_f32(c, in, out, epi, sc, it_spmv, omega, sshift, c->nIntV, c->nBricks - c->nIntV);
    };
    sweep(s->za, s->zb, 1, s->w[1], 0);                                                      // second pre-sweep: za -> zb
    if (!s->lev.empty()) {
        sweep(s->zb, s->t0, 2, 0.0f, 0);                                                    // t0 = r - A zb
        const int nl = (int)s->lev.size(), t0 = s->tailFirst;
        const Lay F0 = brick ? c->LB : c->L;
        const int fb = brick ? 1 : 0;
        const Vec3p ft0 = v3(s->t0);
#define STEP_(OP_, PK_, l_) hipLaunchKernelGGL((k_vmg_step<OP_, PK_>), dim3(cdiv(s->lev[l_].nstrips > 0 ? s->lev[l_].nstrips : 1, 4), 3), dim3(64, 4, 1), 0, c->stream, (const VLevelDev *)s->d_lev, (int)(l_), F0, ft0, conv, fb)
#define STEP(OP_, l_) do { if (c->vmgPackedRows) STEP_(OP_, true, l_); else STEP_(OP_, false, l_); } while (0)
        const int gl = s->globalFrom;
        // the first global level's right-hand side: every rank restricts its own residual over the union box, the sum over the ranks is b
        // (gl is 0 or -1: the all-reduce, if any, sits in front of level 1 = lev[0] -- in the loop below, or in front of the tail when t0 = 0)
        const bool listed = gl >= 0 && s->lev[gl].nstrips > 0 && s->listRhs;   // a level with a brick list: only the listed bricks travel
        auto global_rhs_send = [&](int l) {   // restriction into the staging buffer and its sum over the ranks
            const Box3 &B = s->lev[l].box;
            const size_t n = listed ? (size_t)s->lev[l].nstrips * 64 : (size_t)box_positions(B);
            int r2 = vmg_stage_reserve(c, s, 3 * n);
            if (r2) return r2;
            if (listed) hipLaunchKernelGGL(k_vmg_restrict_list, dim3(cdiv(s->lev[l].nstrips, 4), 3), dim3(64, 4, 1), 0, c->stream, (const VLevelDev *)s->d_lev, l, F0, ft0, conv, fb, s->stage);
            else hipLaunchKernelGGL(k_vmg_restrict_box, dim3(cdiv(B.hi[0] - B.lo[0], 64), cdiv(B.hi[1] - B.lo[1], 4), 3u * (unsigned)(B.hi[2] - B.lo[2])), dim3(64, 4, 1), 0, c->stream,
                                    (const VLevelDev *)s->d_lev, l, F0, ft0, conv, fb, s->stage);
            c->commBytesIter = (double)(3 * n) * sizeof(float);
            return fv_allreduce_f32(c, s->stage, 3 * n);
        };
        auto global_rhs_take = [&](int l, int first) {   // back into b (and x = omega b/d: the first sweep)
            if (listed) hipLaunchKernelGGL(k_vmg_unpack_list, dim3(cdiv(s->lev[l].nstrips, 4), 3), dim3(64, 4, 1), 0, c->stream, (const VLevelDev *)s->d_lev, l, (const float *)s->stage, conv, first);
            else { const size_t n = (size_t)box_positions(s->lev[l].box); hipLaunchKernelGGL(k_vmg_unpack_rhs, dim3((unsigned)((n + 255) / 256), 3), dim3(256), 0, c->stream, (const VLevelDev *)s->d_lev, l, (const float *)s->stage, conv, first); }
        };
        // A distributed level's vectors: 1-entry halo copy of one of them (0 x, 1 y, 3 t) before the sweep that reads it; its right-hand side: halo REDUCTION
        const int nd = s->nDist;
        auto lvl_copy = [&](int l, int which) {
            if (s->rc) return;
            VLevel &A = s->lev[l];
            float *const *v = which == 0 ? A.x : (which == 1 ? A.y : A.t);
            s->rc = fv_halo_level(c, A.L, s->own[l].lo, s->own[l].hi, v, 3, 1, 0);
        };
        // the levels below the all-reduce (or all of them, without one), down, tail and up again: kernels only
        auto global_part = [&](bool afterSend, int from) {
            for (int l = from; l < t0; l++) {   // down
                if (l == gl) { if (afterSend) global_rhs_take(l, 1); }
                else STEP(OP_RESTRICT, l);
                STEP(OP_PRE2, l);
                STEP(OP_RESID, l);
            }
            {
                const int sweeps = c->vmgSweeps > 0 ? c->vmgSweeps : VMG_COARSEST_SWEEPS;
                if (t0 == gl && afterSend) global_rhs_take(t0, 0);
                if (t0 == nl - 1 && s->coarsestInLds)   // the tail is just the LDS-resident coarsest level: its own kernel (no other level's code around the sweeps)
                    hipLaunchKernelGGL(k_vmg_coarsest, dim3(1), dim3(1024), 0, c->stream, (const VLevelDev *)s->d_lev, nl, s->chebM ? sweeps - 1 : sweeps, s->chebM ? 1 : 0, F0, ft0, conv, fb, t0 == gl ? 1 : 0);
                else
                    hipLaunchKernelGGL(k_vmg_tail, dim3(1), dim3(1024), 0, c->stream, (const VLevelDev *)s->d_lev, t0, nl, s->chebM ? sweeps - 1 : sweeps, s->chebM ? 1 : 0, F0, ft0, conv, fb, t0 == gl ? 1 : 0);
            }
            for (int l = t0 - 1; l >= from; l--) {   // up
                STEP(OP_PROPOST, l);
                STEP(OP_POST2, l);
            }
        };
        auto prolong_fine = [&](int itp) {
            if (brick) hipLaunchKernelGGL(k_bvmg_prolong_fine, dim3(nb), blk, 0, c->stream, (const int *)c->brickList, c->nBricks, c->LB, s->lev[0].L, (const uint8_t *)c->vMaskB, v3(s->zb), v3(s->lev[0].x), sc, itp);
            else GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vmg_prolong_fine, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, s->lev[0].L, c->vRowMask, (const unsigned *)c->mlistV,
                               c->vSwz, v3(dg), v3(s->zb), v3(s->lev[0].x), sc, itp));
        };
        if (gl < 0) { global_part(false, 0); prolong_fine(it_arg); }
        else {
            // ---- the distributed levels, down: the rank's share of the right-hand side over its box + ring, the ring's shares to their owners, then the
            // sweeps on the rank's own rows, each after a halo copy of its input
            for (int l = 0; l < nd; l++) {
                const Box3 &R = s->rbox[l];
                hipLaunchKernelGGL(k_vmg_restrict_partial, dim3(cdiv(R.hi[0] - R.lo[0], 64), cdiv(R.hi[1] - R.lo[1], 4), 3u * (unsigned)(R.hi[2] - R.lo[2])), dim3(64, 4, 1), 0, c->stream,
                                   (const VLevelDev *)s->d_lev, l, F0, ft0, conv, fb, R);
                if ((s->rc = fv_halo_level(c, s->lev[l].L, s->own[l].lo, s->own[l].hi, s->lev[l].b, 3, 1, 1))) return;
                STEP(OP_FIRST, l);
                lvl_copy(l, 0);
                STEP(OP_PRE2, l);
                lvl_copy(l, 1);
                STEP(OP_RESID, l);
                if (s->rc) return;
            }
            // ---- the global hierarchy: right-hand side summed over the ranks, then kernels only (captured once per solve, replayed)
            if ((s->rc = global_rhs_send(gl))) return;
            const bool replay = !c->prm.kernel_timing && !c->prm.no_graph_replay;
            if (replay && !s->midReady) {   // first V-cycle of this solve: capture the segment (the all-reduce it follows has been enqueued)
                hipGraph_t g = nullptr;
                if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    global_part(true, gl);
                    if (nd == 0) prolong_fine(0);   // (inside the captured segment the iteration number is not known: the kernel only tests it against the cap, which the host's loop respects anyway)
                    const hipError_t e2 = hipStreamEndCapture(c->stream, &g);
                    s->midReady = e2 == hipSuccess && g && fv_graph_exec(c, FV_GE_VISCOSITY_MID, g, &s->midExec) == FLIPV_OK;
                    if (!s->midReady) { s->midExec = nullptr; (void)hipGetLastError(); }
                    if (g) (void)hipGraphDestroy(g);
                } else (void)hipGetLastError();
            }
            if (replay && s->midReady) { if (hipGraphLaunch(s->midExec, c->stream) != hipSuccess) { s->rc = FLIPV_ERR_HIP; c->err = "viscosity multigrid: hipGraphLaunch of the coarse segment failed"; return; } }
            else { global_part(true, gl); if (nd == 0) prolong_fine(it_arg); }
            // ---- the distributed levels, up
            for (int l = nd - 1; l >= 0; l--) {
                STEP(OP_PROPOST, l);     // (reads y with the halo the residual step's copy left, and the coarser level's x: global, or halo-copied below)
                lvl_copy(l, 3);
                STEP(OP_POST2, l);
                lvl_copy(l, 0);          // x for the finer level's prolongation
                if (s->rc) return;
            }
            if (nd > 0) prolong_fine(it_arg);
        }
#undef STEP
#undef STEP_
    }
    sweep(s->zb, s->za, 1, s->w[0], 0);                                                      // post-sweeps: zb -> za -> zb
    sweep(s->za, s->zb, 3, s->w[1], sig_shift);
    if (c->nPairs > 0) {   // + the strongly coupled pairs' blocks (additive)
        float *r[3] = {(float *)c->vR[0], (float *)c->vR[1], (float *)c->vR[2]};
        hipLaunchKernelGGL(k_vmg_pairs, dim3(1), dim3(256), 0, c->stream, (const unsigned *)c->pairList, (const VPair *)((const char *)c->pairList + 16), v3(r), v3(s->zb), sc, it_spmv, sig_shift);
    }
}

// PCG with the V-cycle as preconditioner.  On entry k_visc_setup has left r = rhs, x = 0, s (= p) = 0 and the tile / brick list;
// the scalars' slot blocks are zero.  spmv(it) computes q = A p with a(it) = p.q (it = -1: the device-side counter).
// replace_period > 0 (brick layout only): residual replacement every that many iterations (k_viscosity_brick.hip), right after the
// x / r update and before the V-cycle that turns the (replaced) residual into z.
// restart: the hierarchy of this solve exists already (iterative refinement after a stall: the same system, a new right-hand side in r)
int fv_viscosity_pcg_mg(flipv_context *c, const PcgScal &sc_in, int cap, void (*spmv)(flipv_context *, const PcgScal &, int), int replace_period, int restart, int *conv_out) {
    VmgState *s = (VmgState *)c->vmgState;
    int rc = FLIPV_OK;
    if (!restart || !s) rc = vmg_setup(c, &s);
    else if (vmg_brick(c) && s->prevCount >= 0) {   // (a restart inside a solve: the same brick list)
        if (s->prevCount > 0) hipLaunchKernelGGL(k_bvmg_zero, dim3(cdiv(s->prevCount, 4) < 2048 ? cdiv(s->prevCount, 4) : 2048), dim3(64, 4, 1), 0, c->stream, (const int *)s->prevBricks, s->prevCount, s->za[0], s->fineStride, 9);
    } else HIPCHK(c, hipMemsetAsync(s->fineVecs, 0, s->fineVecBytes, c->stream));
    if (rc) return rc;
    PcgScal sc = sc_in;
    sc.noB = 1;   // this loop needs p.q only: the SpMV variant that does not read the residual ...
    sc.onlyA = 1; // ... and forms neither the diagonal nor (q, q/d) (EPI_SPMV_A, visc_rows.h)
    const bool brick = vmg_brick(c);
    const int nb = brick ? fv_brick_grid(c, c->nBricks, 2048) : pcg_grid(c, c->nActiveV);
    const dim3 blk(64, 4, 1);
    float *dg[3] = {c->vDiagU, c->vDiagV, c->vDiagW};
    float *x[3] = {(float *)c->vX[0], (float *)c->vX[1], (float *)c->vX[2]}, *r[3] = {(float *)c->vR[0], (float *)c->vR[1], (float *)c->vR[2]};
    float *p[3] = {(float *)c->vS[0], (float *)c->vS[1], (float *)c->vS[2]}, *q[3] = {(float *)c->vZ[0], (float *)c->vZ[1], (float *)c->vZ[2]};
    auto XR = [&](int it_) {
        if (brick) hipLaunchKernelGGL(k_bvpcg_xr, dim3(fv_brickv_grid(c->nBricks, c->prm.grid_cap > 0 ? ((c->prm.grid_cap + 7) / 8) * 8 : 2048)), blk, 0, c->stream, (const int *)c->brickList, c->nBricks, (const uint8_t *)c->vMaskB, v3(dg), v3(x), v3(r), v3(p), v3(q), v3(s->za), s->w[0], sc, it_);
        else GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vpcg_xr, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, c->vRowMask, (const unsigned *)c->mlistV, c->vSwz, v3(dg), v3(x), v3(r), v3(p), v3(q), v3(s->za), s->w[0], sc, it_));
    };
    auto PP = [&](int it_) {
        if (brick) hipLaunchKernelGGL(k_bvpcg_p, dim3(fv_brickv_grid(c->nBricks, c->prm.grid_cap > 0 ? ((c->prm.grid_cap + 7) / 8) * 8 : 2048)), blk, 0, c->stream, (const int *)c->brickList, c->nBricks, (const uint8_t *)c->vMaskB, v3(s->zb), v3(p), sc, it_);
        else GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vpcg_p, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, c->vRowMask, (const unsigned *)c->mlistV, c->vSwz, v3(s->zb), v3(p), sc, it_));
    };
    if (!brick) replace_period = 0;
    HIPCHK(c, hipMemsetAsync(sc.itA, 0, 2 * sizeof(int), c->stream));
    // Block contexts: the V-cycle is the single domain's (VmgState::globalFrom = 0: halo copies before the fine-level sweeps, one global coarse
    // hierarchy), or, with flipv_params.multigrid_rank_local, a cycle over the rank's OWN rows with the couplings across the cut faces dropped
    // (its sweep vectors are zero on the halo) and no exchange -- a block-diagonal, symmetric positive definite preconditioner, like the
    // pressure multigrid's.  Either way the CG around it applies the true operator (halo copy of p before the SpMV) and all-reduces its scalars.
    const int hl = brick ? 1 : 0;
    const HaloArray ph[3] = {{p[0], sizeof(float), hl}, {p[1], sizeof(float), hl}, {p[2], sizeof(float), hl}};
    XR(-1);                       // za = omega r/d
    vmg_vcycle(c, s, sc, 0, 0);   // z, sig(0)
    if (s->rc) return s->rc;
    if (c->comm && (rc = fv_allreduce_scalars(c, sc.sig(0), NSLOT))) return rc;
    PP(-1);                       // p = z
    const int every = c->prm.check_every > 0 ? c->prm.check_every : 4;   // (iterations per replayed chunk; 2 / 3 / 4 / 5 / 8 / 12 / 16 scanned on the bench scene: 382 / 381 / 372-378 / 384 / 382-391 / 438 / 421 ms of viscosity solves over 25 substeps)
    int conv = -1;
    auto iteration = [&](int it, bool replace) -> int {   // it = IT_DEVICE inside the graph
        int r2;
        const long ex0 = c->nExchanges, ar0 = c->nAllReduces;
        // p on the neighbours' halo entries -- in the brick layout beside the SpMV over the bricks interior to the rank's box (fv_build_bricks: listed first)
        if (c->comm && brick && c->comm->nranks > 1 && !c->prm.no_comm_overlap && c->nIntV > 0) {
            if ((r2 = fv_halo_copy_begin(c, ph, 3, 1))) return r2;
            fv_brick_spmv<float>(c, sc, it == IT_DEVICE ? -1 : it, false, 0, c->nIntV);
            if ((r2 = fv_halo_wait(c))) return r2;
            if (c->nBricks > c->nIntV) fv_brick_spmv<float>(c, sc, it == IT_DEVICE ? -1 : it, false, c->nIntV, c->nBricks - c->nIntV);
        } else {
            if (c->comm && (r2 = fv_halo_copy(c, ph, 3, 1))) return r2;
            spmv(c, sc, it == IT_DEVICE ? -1 : it);
        }
        if (c->comm && (r2 = fv_allreduce_scalars(c, sc.a(it), NSLOT))) return r2;                    // p.q
        XR(it);
        if (replace) fv_brick_replace<float>(c, sc, it == IT_DEVICE ? -1 : it, replace_period, 0, s->za, s->w[0]);
        vmg_vcycle(c, s, sc, it, 1);
        if (s->rc) return s->rc;
        if (c->comm && (r2 = fv_allreduce_scalars(c, sc.rmax(it), 3 * NSLOT))) return r2;             // max|r| and max|alpha p| of this iteration, (r, z) of the next
        PP(it);
        c->exchIter = (int)(c->nExchanges - ex0); c->allrIter = (int)(c->nAllReduces - ar0);
        return FLIPV_OK;
    };
    // where in a chunk of `every` iterations a replacement can fall due (the kernels decide exactly, from the iteration number)
    // (the kernels decide from the ABSOLUTE iteration number; a replayed chunk does not know where it sits in the solve -- its first position shifts with the
    // directly launched first iteration, and a period that does not divide the chunk length falls due at a different position every replay -- so a chunk
    // carries the (mostly empty) replacement launches at EVERY position; the kernel-by-kernel loop launches them where they are due)
    auto may_replace = [&](int e) { (void)e; return replace_period > 0; };
    auto due_at = [&](int it) { return replace_period > 0 && ((it + 1) % replace_period) == 0; };
    const bool graph = !c->comm && !c->prm.kernel_timing && !c->prm.no_graph_replay;
    if (graph) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        // The first iteration is launched directly: capturing the chunk and updating the graph executable takes the host ~0.35 ms, the start-up cycle
        // above keeps the device busy for ~0.25 ms -- with one more iteration queued the capture is hidden (the device sat idle ~0.1 ms per stage).
        int done0 = 0;
        if (cap > 1) { if ((rc = iteration(IT_DEVICE, may_replace(0)))) return rc; done0 = 1; }
        const auto tc0 = std::chrono::steady_clock::now();
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        for (int e = 0; e < every; e++) (void)iteration(IT_DEVICE, may_replace(e));
        const int e1 = fv_read_capture(c, c->d_flags, 1);   // the stop flag, published to the host at the end of every replay
        hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        const auto tc1 = std::chrono::steady_clock::now();
        if (e1 != FLIPV_OK || e2 != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            c->err = "viscosity multigrid: stream capture failed";
            return FLIPV_ERR_HIP;
        }
        if ((rc = fv_graph_exec(c, FV_GE_VISCOSITY_MG, g, &ge))) { (void)hipGraphDestroy(g); return rc; }
        if (c->prm.verbose) fprintf(stderr, "  multigrid loop: stream capture %.3f ms, graph executable %.3f ms (host)\n", std::chrono::duration<double, std::milli>(tc1 - tc0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc1).count());
        if ((rc = fv_read_wait(c))) { (void)hipGraphDestroy(g); return rc; }
        for (int done = done0; done < cap && conv < 0; done += every) {
            hipError_t el = hipGraphLaunch(ge, c->stream);
            fv_read_replayed(c);
            if (el != hipSuccess || fv_read_wait(c) != FLIPV_OK) { (void)hipGraphDestroy(g); c->err = "hipGraphLaunch failed"; return FLIPV_ERR_HIP; }
            conv = c->h_pub[FV_PUB_REPLAY];
        }
        (void)hipGraphDestroy(g);
    } else {
        int it = 0;
        while (it < cap && conv < 0) {
            const int stop = it + every < cap ? it + every : cap;
            for (; it < stop; it++)
                if ((rc = iteration(it, due_at(it)))) return rc;
            if ((rc = fv_read_now(c, c->h_flags, c->d_flags, 1))) return rc;
            conv = c->h_flags[0];
        }
    }
    HIPCHK(c, hipGetLastError());
    *conv_out = conv;
    return FLIPV_OK;
}
