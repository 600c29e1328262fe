// k_viscosity_mg.hip -- Galerkin multigrid preconditioner for the variational viscosity PCG (single GPU, fp32 vectors).
//
// The viscosity system (viscositysolver.cpp:276-664) couples the three face-velocity components through the shear
// stresses; its rigid modes make piecewise-constant coarse spaces useless and only Galerkin coarse operators survive the
// near-empty control volumes at the free surface.  What works (scipy prototype on the oracle's matrices:
// 64 PCG iterations at 64^3, 68 at 128^3 with V(2,2), against 440 / 1490 with the diagonal):
//   transfer P   per component, on the MAC lattices: linear along the face normal (a fine face on a coarse face plane
//                takes that coarse face, one between two planes the mean of both), piecewise constant across
//   coarse A     P^T A P.  With this P the coarse operator of every level has the SAME 23-entry pattern per row: 15
//                same-component neighbours (normal offset -1/0/+1 times {centre, +-1 along either transverse axis}) and 4
//                + 4 cross-component neighbours -- the fine 15-point pattern plus the normal-times-transverse diagonals.
//                Storage: one coefficient grid per (component, slot) on the level's dense index space ("dense slots").
//   level 0      matrix-free: the SpMV kernel of k_viscosity.hip applies A, pointwise kernels do the Jacobi updates
//   assembly     level 1 is scattered from the matrix-free fine rows with atomics (each fine row knows its <= 15 entries
//                and each end of an entry its <= 2 parents), level l+1 from level l the same way
//   cycle        V(2,2), damped Jacobi (omega 0.6: lambda_max(D^-1 A) ~ 3), zero initial guess
#include "flipv_internal.h"
#include "pcg_common.h"

#include <vector>

void fv_visc_apply_f32(flipv_context *c, float *const in[3], float *const out[3]);  // k_viscosity.hip: out = A in (fp32, no dots)

namespace {

constexpr int VS = 23;             // slots per row
constexpr float VMG_OMEGA = 0.6f;
constexpr int VMG_COARSEST_SWEEPS = 8;

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

struct VLevel {          // a coarse level (>= 1)
    Lay L;
    float *coef[3][VS];
    float *x[3], *y[3], *b[3], *t[3];
};
struct VLevelDev {       // what kernels need of a coarse level
    Lay L;
    float *coef[3][VS];
};
struct FineOp {          // the matrix-free level 0 (k_viscosity.hip's arrays)
    const float *vm[3];
    const float *fC, *fE[3];
    const uint8_t *mask;
};

__device__ unsigned g_dropped[2];
static Lay coarse_lay(const Lay &F) {
    Lay C;
    C.I = (F.I + 1) / 2; C.J = (F.J + 1) / 2; C.K = (F.K + 1) / 2;
    C.PX = ((C.I + 1 + 3) / 4) * 4; C.PY = C.J + 1; C.PZ = C.K + 1;
    C.sy = C.PX; C.sz = (long)C.PX * C.PY;
    C.n = (size_t)C.sz * C.PZ;
    C.guard = (((size_t)C.sz + (size_t)C.sy + 8) + 63) / 64 * 64;
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
// add value to the coarse entry (row (c, I), column (c2, J)); the offset J - I is always one of the 23 slots
__device__ __forceinline__ void d_coarse_add(const VLevelDev &C, int c, const int I[3], int c2, const int J[3], float v) {
    const int dx = J[0] - I[0], dy = J[1] - I[1], dz = J[2] - I[2];
    if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) { if (v != 0.0f) atomicAdd(&g_dropped[0], 1u); return; }  // cannot happen: the pattern is closed under this coarsening
    const int s = ST.lut[c][c2][(dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)];
    if (s >= 0) atomicAdd(C.coef[c][s] + gidx(C.L, I[0], I[1], I[2]), v);
    else if (v != 0.0f) atomicAdd(&g_dropped[1], 1u);
}
// scatter one entry A(a, b) = v of a finer level into the coarse operator
__device__ __forceinline__ void d_rap_entry(const VLevelDev &C, int c, const int p[3], int c2, const int q[3], float v) {
    int PI[2][3], PJ[2][3];
    float wi[2], wj[2];
    const int ni = d_parents(c, p, PI, wi), nj = d_parents(c2, q, PJ, wj);
    for (int a = 0; a < ni; a++)
        for (int b = 0; b < nj; b++) d_coarse_add(C, c, PI[a], c2, PJ[b], wi[a] * v * wj[b]);
}

// ---- level l -> level l+1 (l >= 1): every stored entry of the dense-slot operator
__global__ void k_vmg_rap(VLevelDev F, VLevelDev C) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= F.L.PX || j >= F.L.PY) return;
    const size_t ci = gidx(F.L, i, j, k);
    const int p[3] = {i, j, k};
    for (int c = 0; c < 3; c++) {
        if (F.coef[c][ST.diag[c]][ci] == 0.0f) continue;
        for (int s = 0; s < VS; s++) {
            const float v = F.coef[c][s][ci];
            if (v == 0.0f) continue;
            const int q[3] = {i + ST.off[c][s][0], j + ST.off[c][s][1], k + ST.off[c][s][2]};
            d_rap_entry(C, c, p, ST.comp[c][s], q, v);
        }
    }
}

// ---- coarse levels: y = A x for one dof
__device__ __forceinline__ float d_apply(const VLevelDev &A, float *const x[3], int c, size_t ci) {
    const long sy = A.L.sy, sz = A.L.sz;
    // all coefficients, then all neighbour values, then the sum: 46 independent loads instead of 23 dependent
    // load-test-load chains (empty slots hold 0 and the guard zones make every neighbour address valid)
    float v[VS], xv[VS];
#pragma unroll
    for (int q = 0; q < VS; q++) v[q] = A.coef[c][q][ci];
#pragma unroll
    for (int q = 0; q < VS; q++) xv[q] = x[ST.comp[c][q]][(long)ci + ST.off[c][q][0] + ST.off[c][q][1] * sy + ST.off[c][q][2] * sz];
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < VS; q++) s += v[q] * xv[q];
    return s;
}
struct Vec3p { float *p[3]; };
// mode 0: out = omega b/d (sweep from a zero guess)   1: out = x + omega (b - A x)/d   2: out = b - A x
__global__ void k_vmg_op(VLevelDev A, Vec3p x, Vec3p b, Vec3p out, int mode) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= A.L.PX || j >= A.L.PY) return;
    const size_t ci = gidx(A.L, i, j, k);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float d = A.coef[c][ST.diag[c]][ci];
        if (d == 0.0f) { if (mode != 2) out.p[c][ci] = 0.0f; else out.p[c][ci] = 0.0f; continue; }
        const float bb = b.p[c][ci];
        if (mode == 0) { out.p[c][ci] = VMG_OMEGA * bb / d; continue; }
        const float ax = d_apply(A, x.p, c, ci);
        out.p[c][ci] = mode == 1 ? x.p[c][ci] + VMG_OMEGA * (bb - ax) / d : bb - ax;
    }
}

// ---- transfers (the fine side is any level's dense index space, level 0 included)
// restriction: coarse b = P^T t
__global__ void k_vmg_restrict(Lay F, Lay C, Vec3p tf, Vec3p bc) {
    const int I = blockIdx.x * 64 + threadIdx.x, J = blockIdx.y * 4 + threadIdx.y, K = blockIdx.z;
    if (I >= C.PX || J >= C.PY) return;
    const int P[3] = {I, J, K};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float s = 0.0f;
        if (d_in_lattice(C, c, P)) {
            const int t1 = (c + 1) % 3, t2 = (c + 2) % 3;
            for (int dn = -1; dn <= 1; dn++)
                for (int a = 0; a < 2; a++)
                    for (int b = 0; b < 2; b++) {
                        int q[3];
                        q[c] = 2 * P[c] + dn; q[t1] = 2 * P[t1] + a; q[t2] = 2 * P[t2] + b;
                        if (d_in_lattice(F, c, q)) s += (dn == 0 ? 1.0f : 0.5f) * tf.p[c][gidx(F, q[0], q[1], q[2])];
                    }
        }
        bc.p[c][gidx(C, I, J, K)] = s;
    }
}
// prolongation: fine x += P xc, on dofs flagged by `fmask` (level 0: the row mask) or everywhere (coarse levels: nullptr)
__global__ void k_vmg_prolong(Lay F, Lay C, Vec3p xf, Vec3p xc, const uint8_t *__restrict__ fmask) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= F.PX || j >= F.PY) return;
    const size_t ci = gidx(F, i, j, k);
    const unsigned m = fmask ? fmask[ci] : 7u;
    if (!m) return;
    const int p[3] = {i, j, k};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (!((m >> c) & 1u) || !d_in_lattice(F, c, p)) continue;
        int P[2][3];
        float w[2];
        const int n = d_parents(c, p, P, w);
        float s = w[0] * xc.p[c][gidx(C, P[0][0], P[0][1], P[0][2])];
        if (n == 2) s += w[1] * xc.p[c][gidx(C, P[1][0], P[1][1], P[1][2])];
        xf.p[c][ci] += s;
    }
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
    float *z[3] = {nullptr, nullptr, nullptr}, *q2[3] = {nullptr, nullptr, nullptr};   // level 0: z = M^-1 r, q2 = A z / residual
    std::vector<void *> allocs;
    std::vector<std::pair<void *, size_t>> coefBlocks;   // (base, bytes) of every level's coefficient storage, zeroed per solve
    ~VmgState() { for (void *p : allocs) (void)hipFree(p); }
};

static int vmg_alloc(flipv_context *c, VmgState *s, const Lay &L, size_t count, float **base) {
    const size_t tot = (L.n + 2 * L.guard) * count;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, tot * sizeof(float));
    if (e != hipSuccess) { c->err = std::string("hipMalloc(viscosity multigrid): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
    s->allocs.push_back(q);
    HIPCHK(c, hipMemsetAsync(q, 0, tot * sizeof(float), c->stream));
    *base = (float *)q;
    return FLIPV_OK;
}

static VLevelDev dev_of(const VLevel &l) {
    VLevelDev d;
    d.L = l.L;
    for (int c = 0; c < 3; c++) for (int s = 0; s < VS; s++) d.coef[c][s] = l.coef[c][s];
    return d;
}
static Vec3p v3(float *const p[3]) { Vec3p v; v.p[0] = p[0]; v.p[1] = p[1]; v.p[2] = p[2]; return v; }

#define LGRID(Lv) dim3(cdiv((Lv).PX, 64), cdiv((Lv).PY, 4), (unsigned)(Lv).PZ), dim3(64, 4, 1)

}  // namespace

void fv_vmg_free(flipv_context *c) {
    delete (VmgState *)c->vmgState;
    c->vmgState = nullptr;
}

static int vmg_setup(flipv_context *c, VmgState **out) {
    VmgState *s = (VmgState *)c->vmgState;
    int rc;
    if (!s) {
        SlotTables T;
        build_slot_tables(&T);
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(ST), &T, sizeof(T)));
        s = new VmgState();
        c->vmgState = s;
        float *base;
        if ((rc = vmg_alloc(c, s, c->L, 6, &base))) return rc;
        const size_t per0 = c->L.n + 2 * c->L.guard;
        for (int m = 0; m < 3; m++) { s->z[m] = base + (size_t)m * per0 + c->L.guard; s->q2[m] = base + (size_t)(3 + m) * per0 + c->L.guard; }
        Lay F = c->L;
        while (true) {
            const int mx = F.I > F.J ? (F.I > F.K ? F.I : F.K) : (F.J > F.K ? F.J : F.K);
            static const int minDim = getenv("FLIPV_VMG_MINDIM") ? atoi(getenv("FLIPV_VMG_MINDIM")) : 4;
            if (mx <= minDim || s->lev.size() >= 8) break;
            VLevel l;
            l.L = coarse_lay(F);
            const size_t per = l.L.n + 2 * l.L.guard;
            float *cb, *vb;
            if ((rc = vmg_alloc(c, s, l.L, 3 * VS, &cb)) || (rc = vmg_alloc(c, s, l.L, 12, &vb))) return rc;
            s->coefBlocks.push_back({cb, per * 3 * VS * sizeof(float)});
            for (int m = 0; m < 3; m++) {
                for (int q = 0; q < VS; q++) l.coef[m][q] = cb + (size_t)(m * VS + q) * per + l.L.guard;
                l.x[m] = vb + (size_t)m * per + l.L.guard;
                l.y[m] = vb + (size_t)(3 + m) * per + l.L.guard;
                l.b[m] = vb + (size_t)(6 + m) * per + l.L.guard;
                l.t[m] = vb + (size_t)(9 + m) * per + l.L.guard;
            }
            s->lev.push_back(l);
            F = l.L;
        }
    }
    // this solve's coarse operators; z and q2 must be zero wherever there is no row (the SpMV reads neighbours unmasked)
    HIPCHK(c, hipMemsetAsync(s->z[0] - c->L.guard, 0, 6 * (c->L.n + 2 * c->L.guard) * sizeof(float), c->stream));
    for (auto &b : s->coefBlocks) HIPCHK(c, hipMemsetAsync(b.first, 0, b.second, c->stream));
    if (!s->lev.empty()) {
        FineOp A;
        A.vm[0] = c->vmU; A.vm[1] = c->vmV; A.vm[2] = c->vmW;
        A.fC = c->fC; A.fE[0] = c->fEU; A.fE[1] = c->fEV; A.fE[2] = c->fEW;
        A.mask = c->vRowMask;
        GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vmg_rap_fine, dim3(pcg_grid(c, c->nActiveV)), dim3(64, 4, 1), 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, A,
                           dev_of(s->lev[0])));
        for (size_t l = 0; l + 1 < s->lev.size(); l++)
            hipLaunchKernelGGL(k_vmg_rap, LGRID(s->lev[l].L), 0, c->stream, dev_of(s->lev[l]), dev_of(s->lev[l + 1]));
    }
    if (getenv("FLIPV_VMG_DEBUG")) { unsigned h[2]; (void)hipStreamSynchronize(c->stream); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dropped), sizeof(h)); fprintf(stderr, "vmg dropped entries: out of range %u, no slot %u\n", h[0], h[1]); }
    HIPCHK(c, hipGetLastError());
    *out = s;
    return FLIPV_OK;
}

// coarse level l (index into s->lev): x <- V-cycle(b), result in lev[l].x
static void vmg_coarse(flipv_context *c, VmgState *s, size_t l) {
    VLevel &A = s->lev[l];
    const VLevelDev Ad = dev_of(A);
    const bool last = l + 1 == s->lev.size();
    hipLaunchKernelGGL(k_vmg_op, LGRID(A.L), 0, c->stream, Ad, v3(A.x), v3(A.b), v3(A.x), 0);      // x = omega b/d
    if (last) {
        static const int sweeps = getenv("FLIPV_VMG_SWEEPS") ? atoi(getenv("FLIPV_VMG_SWEEPS")) : VMG_COARSEST_SWEEPS;
        for (int q = 0; q < sweeps; q += 2) {
            hipLaunchKernelGGL(k_vmg_op, LGRID(A.L), 0, c->stream, Ad, v3(A.x), v3(A.b), v3(A.y), 1);
            hipLaunchKernelGGL(k_vmg_op, LGRID(A.L), 0, c->stream, Ad, v3(A.y), v3(A.b), v3(A.x), 1);
        }
        return;
    }
    hipLaunchKernelGGL(k_vmg_op, LGRID(A.L), 0, c->stream, Ad, v3(A.x), v3(A.b), v3(A.y), 1);      // second pre-sweep -> y
    hipLaunchKernelGGL(k_vmg_op, LGRID(A.L), 0, c->stream, Ad, v3(A.y), v3(A.b), v3(A.t), 2);      // t = b - A y
    VLevel &C = s->lev[l + 1];
    hipLaunchKernelGGL(k_vmg_restrict, LGRID(C.L), 0, c->stream, A.L, C.L, v3(A.t), v3(C.b));
    vmg_coarse(c, s, l + 1);
    hipLaunchKernelGGL(k_vmg_prolong, LGRID(A.L), 0, c->stream, A.L, C.L, v3(A.y), v3(C.x), (const uint8_t *)nullptr);
    hipLaunchKernelGGL(k_vmg_op, LGRID(A.L), 0, c->stream, Ad, v3(A.y), v3(A.b), v3(A.t), 1);      // post-sweeps: y -> t -> x
    hipLaunchKernelGGL(k_vmg_op, LGRID(A.L), 0, c->stream, Ad, v3(A.t), v3(A.b), v3(A.x), 1);
}

// z = M^-1 r (level 0), (r, z) into sig(it_next)
static void vmg_vcycle(flipv_context *c, VmgState *s, const PcgScal &sc, int it_next) {
    const int nb = pcg_grid(c, c->nActiveV);
    const dim3 blk(64, 4, 1);
    float *dg[3] = {c->vDiagU, c->vDiagV, c->vDiagW};
    float *r[3] = {(float *)c->vR[0], (float *)c->vR[1], (float *)c->vR[2]};
    PcgScal none;
    memset(&none, 0, sizeof(none));
#define FINE(mode, scal, itn) GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vmg_fine, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, c->vRowMask, v3(dg), v3(r), v3(s->z), v3(s->q2), mode, scal, itn))
    FINE(0, none, 0);                                  // z = omega r/d
    fv_visc_apply_f32(c, s->z, s->q2); FINE(1, none, 0);   // second pre-sweep
    if (!s->lev.empty()) {
        fv_visc_apply_f32(c, s->z, s->q2); FINE(2, none, 0);   // q2 = r - A z
        VLevel &C = s->lev[0];
        hipLaunchKernelGGL(k_vmg_restrict, LGRID(C.L), 0, c->stream, c->L, C.L, v3(s->q2), v3(C.b));
        vmg_coarse(c, s, 0);
        hipLaunchKernelGGL(k_vmg_prolong, LGRID(c->L), 0, c->stream, c->L, C.L, v3(s->z), v3(C.x), (const uint8_t *)c->vRowMask);
    }
    fv_visc_apply_f32(c, s->z, s->q2); FINE(1, none, 0);   // post-sweeps
    fv_visc_apply_f32(c, s->z, s->q2); FINE(3, sc, it_next);
#undef FINE
}

// PCG with the V-cycle as preconditioner.  On entry k_visc_setup has left r = rhs, x = 0, s (= p) = 0 and the tile list;
// the scalars' slot blocks are zero.  spmv(it) computes q = A p with a(it) = p.q.
int fv_viscosity_pcg_mg(flipv_context *c, const PcgScal &sc, int cap, void (*spmv)(flipv_context *, const PcgScal &, int), int *conv_out) {
    VmgState *s = nullptr;
    int rc = vmg_setup(c, &s);
    if (rc) return rc;
    const int nb = pcg_grid(c, c->nActiveV);
    const dim3 blk(64, 4, 1);
    float *x[3] = {(float *)c->vX[0], (float *)c->vX[1], (float *)c->vX[2]}, *r[3] = {(float *)c->vR[0], (float *)c->vR[1], (float *)c->vR[2]};
    float *p[3] = {(float *)c->vS[0], (float *)c->vS[1], (float *)c->vS[2]}, *q[3] = {(float *)c->vZ[0], (float *)c->vZ[1], (float *)c->vZ[2]};
    vmg_vcycle(c, s, sc, 0);
    GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vpcg_p, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, c->vRowMask, v3(s->z), v3(p), sc, -1));
    const int every = c->prm.check_every > 0 ? c->prm.check_every : 4;
    int conv = -1, it = 0;
    while (it < cap && conv < 0) {
        const int stop = it + every < cap ? it + every : cap;
        for (; it < stop; it++) {
            spmv(c, sc, it);
            GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vpcg_xr, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, c->vRowMask, v3(x), v3(r), v3(p), v3(q), sc, it));
            vmg_vcycle(c, s, sc, it + 1);
            GEO_RUN(c->tgV.rowl, hipLaunchKernelGGL(k_vpcg_p, dim3(nb), blk, 0, c->stream, c->tileListV, c->nActiveV, c->tgV, c->L, c->vRowMask, v3(s->z), v3(p), sc, it));
        }
        HIPCHK(c, hipMemcpyAsync(c->h_flags, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        conv = c->h_flags[0];
    }
    HIPCHK(c, hipGetLastError());
    *conv_out = conv;
    return FLIPV_OK;
}
