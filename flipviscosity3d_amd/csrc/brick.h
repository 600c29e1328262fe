// brick.h -- device helpers of the kernels that walk the viscosity system in the brick layout (flipv_internal.h: bidx):
// k_viscosity_brick.hip (PCG) and k_viscosity_mg.hip (the multigrid preconditioner's fine-level vector kernels).
// One wave per brick of 8 x 4 x 2 indices, one lane per index; a block of (64, 4, 1) threads takes four consecutive list entries.
#pragma once
#include "pcg_common.h"

// the bricks an index box [ib, ie) x [jb, je) x [kb, ke) touches, in padded brick coordinates
struct BrickBox { int b0[3], nb[3]; };
static inline BrickBox brick_box(const Lay &box) {
    BrickBox R;
    R.b0[0] = (box.ib - box.ox + 8) >> 3; R.nb[0] = ((box.ie - box.ox - 1 + 8) >> 3) - R.b0[0] + 1;   // (box-local: bidx)
    R.b0[1] = (box.jb - box.oy + 4) >> 2; R.nb[1] = ((box.je - box.oy - 1 + 4) >> 2) - R.b0[1] + 1;
    R.b0[2] = (box.kb - box.oz + 2) >> 1; R.nb[2] = ((box.ke - box.oz - 1 + 2) >> 1) - R.b0[2] + 1;
    return R;
}
// linear brick id (x fastest over the whole padded brick grid) of brick `code` of the box's brick range
__device__ __forceinline__ int d_brick_of_code(const BrickBox &R, const Lay &LB, int code) {
    const int bx = code % R.nb[0], r = code / R.nb[0], by = r % R.nb[1], bz = r / R.nb[1];
    return (int)((long)(R.b0[2] + bz) * LB.sz + (long)(R.b0[1] + by) * LB.sy + (long)(R.b0[0] + bx));
}
// GLOBAL index (i, j, k) of lane `lane` of brick `brick`
__device__ __forceinline__ void d_brick_ijk(const Lay &LB, int brick, int lane, int &i, int &j, int &k) {
    const int bx = brick % (int)LB.sy, r = brick / (int)LB.sy, nby = (int)(LB.sz / LB.sy), by = r % nby, bz = r / nby;
    i = LB.ox + (bx << 3) - 8 + ((lane >> 5) << 2) + (lane & 3);
    j = LB.oy + (by << 2) - 4 + ((lane >> 2) & 3);
    k = LB.oz + (bz << 1) - 2 + ((lane >> 4) & 1);
}
// neighbour offsets of this lane (threadIdx.x = position inside the brick): nb_brick() from the lane's bits
__device__ __forceinline__ NbOff d_lane_off(int sby, int sbz) {
    const int lane = (int)threadIdx.x, li = lane & 3, lj = (lane >> 2) & 3, lk = (lane >> 4) & 1;
    NbOff o;
    o.xp = li < 3 ? 1 : 29;          o.xm = li > 0 ? -1 : -29;
    o.yp = lj < 3 ? 4 : sby - 12;    o.ym = lj > 0 ? -4 : -(sby - 12);
    o.zp = lk == 0 ? 16 : sbz - 16;  o.zm = lk == 1 ? -16 : -(sbz - 16);
    return o;
}

// x-neighbours inside a brick WITHOUT a load.  A brick is two halves of 4 x 4 x 2 (lanes 0..31: x = 0..3, lanes 32..63: x = 4..7), x fastest inside a half: the value at
// x - 1 / x + 1 sits in the lane's own quad, or -- at the seam between the halves -- in the other half's quad (v_permlane32_swap + a DPP quad permute), or in the neighbouring
// brick: only the 8 lanes of a brick face load it (`edge`).  (13 of the SpMV's 46 loads per lane are x-shifted; a load whose lanes straddle two bricks costs the vector cache
// two passes: with them the launch on the bench scene takes 12.4 us, with the shifted addresses replaced by aligned ones 10.1.)
typedef unsigned fv_v2u __attribute__((ext_vector_type(2)));
// v_permlane32_swap of a register with itself: .x = the LOWER half's values in both halves (what an upper-half lane needs from lane l - 32), .y = the UPPER half's in
// both (what a lower-half lane needs from lane l + 32) -- no select: d_xm reads the other half only from upper-half lanes, d_xp only from lower-half lanes
__device__ __forceinline__ fv_v2u d_halves_u(unsigned v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false); }
template <int CTRL> __device__ __forceinline__ unsigned d_quad_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
// both x-neighbours of a value at once: lo = the value at x - 1, hi = at x + 1 (same j, k); eL / eR = what a lane of the brick's x = 0 / x = 7 face loaded
// quad_perm [0,0,1,2] = 0x90, [3,3,3,3] = 0xFF, [1,2,3,3] = 0xF9, [0,0,0,0] = 0x00
__device__ __forceinline__ void d_xnb(float v, float eL, float eR, float &lo, float &hi) {
    const unsigned u = __float_as_uint(v);
    const fv_v2u hv = d_halves_u(u);
    const int li = (int)threadIdx.x & 3;
    const bool upper = ((int)threadIdx.x & 32) != 0;
    // (every shuffle as an unconditional statement: inside a ?: it would run under the lanes' branch, and a DPP read of an inactive lane returns nothing)
    const float ql = __uint_as_float(d_quad_u<0x90>(u)), hl = __uint_as_float(d_quad_u<0xFF>(hv.x));
    const float qh = __uint_as_float(d_quad_u<0xF9>(u)), hh = __uint_as_float(d_quad_u<0x00>(hv.y));
    lo = li ? ql : (upper ? hl : eL);
    hi = li < 3 ? qh : (upper ? eR : hh);
}
__device__ __forceinline__ void d_xnb(double v, double eL, double eR, double &lo, double &hi) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned l = (unsigned)b, h = (unsigned)(b >> 32);
    const fv_v2u hl = d_halves_u(l), hh = d_halves_u(h);
    const int li = (int)threadIdx.x & 3;
    const bool upper = ((int)threadIdx.x & 32) != 0;
    auto pack = [](unsigned a, unsigned c) { return __longlong_as_double((long long)(((unsigned long long)c << 32) | a)); };
    const double ql = pack(d_quad_u<0x90>(l), d_quad_u<0x90>(h)), sl = pack(d_quad_u<0xFF>(hl.x), d_quad_u<0xFF>(hh.x));
    const double qh = pack(d_quad_u<0xF9>(l), d_quad_u<0xF9>(h)), sh = pack(d_quad_u<0x00>(hl.y), d_quad_u<0x00>(hh.y));
    lo = li ? ql : (upper ? sl : eL);
    hi = li < 3 ? qh : (upper ? eR : sh);
}
template <typename T> __device__ __forceinline__ T d_xm(T v, T e) { T lo, hi; d_xnb(v, e, e, lo, hi); return lo; }
template <typename T> __device__ __forceinline__ T d_xp(T v, T e) { T lo, hi; d_xnb(v, e, e, lo, hi); return hi; }

// the system's arrays, all in the brick layout
template <typename T>
struct BrickSys {
    const float *diag[3];
    T *x[3], *q[3], *s[3];
    RT<T> *r[3];
    const uint8_t *mask;   // bit m: component m has a row at the index
    int sby, sbz;          // element strides between bricks along j and k
};

// A wave's walk over the brick list.  Blocks take quads of list entries (wave w of a block the quad's entry w) with a grid stride;
// d_tile_slot hands each XCD a contiguous eighth of the list (bricks are listed x fastest, so an eighth is a slab of brick planes:
// the face neighbours of a wave's brick are bricks of the same XCD's L2).  The next brick's id and this lane's mask byte are
// requested before the current brick is worked on: the id -> mask -> data chain of dependent loads is paid once per wave, not per brick.
struct BrickWalk {
    int vb, nquads, nvb;
    size_t a, an;
    unsigned m, mn;
    __device__ __forceinline__ void fetch(int v, const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask, size_t &ao, unsigned &mo) const {
        ao = 0; mo = 0u;
        if (v < nvb) {
            const int slot = d_tile_slot(v, nquads);
            const int e = slot * 4 + (int)threadIdx.y;
            if (slot < nquads && e < nb) {
                ao = ((size_t)bricks[e] << 6) + threadIdx.x;
                mo = mask[ao];
            }
        }
    }
    __device__ __forceinline__ void begin(const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask) {
        nquads = (nb + 3) >> 2;
        nvb = ((nquads + 7) >> 3) << 3;
        vb = (int)blockIdx.x;
        fetch(vb, bricks, nb, mask, a, m);
        fetch(vb + (int)gridDim.x, bricks, nb, mask, an, mn);
    }
    __device__ __forceinline__ bool valid() const { return vb < nvb; }
    __device__ __forceinline__ void next(const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask) {
        a = an; m = mn;
        vb += (int)gridDim.x;
        fetch(vb + (int)gridDim.x, bricks, nb, mask, an, mn);
    }
};

// The walk of the kernels WITHOUT neighbour accesses (x / r / p updates): the lane <-> index map is free there, so a lane takes 4 consecutive
// entries of one brick -- lane l of wave w of a block works on list entry 16 g + 4 w + (l >> 4), entries 4 (l & 15) .. + 3 of that brick -- and
// every access is 16 bytes (a wave instruction moves four whole bricks instead of one).  m: the four indices' mask bytes.
struct BrickWalkV {
    int vb, ngroups, nvb;
    size_t a, an;
    unsigned m, mn;
    __device__ __forceinline__ void fetch(int v, const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask, size_t &ao, unsigned &mo) const {
        ao = 0; mo = 0u;
        if (v < nvb) {
            const int slot = d_tile_slot(v, ngroups);
            const int e = slot * 16 + (int)threadIdx.y * 4 + ((int)threadIdx.x >> 4);
            if (slot < ngroups && e < nb) {
                ao = ((size_t)bricks[e] << 6) + (size_t)(((int)threadIdx.x & 15) << 2);
                mo = *reinterpret_cast<const unsigned *>(mask + ao);
            }
        }
    }
    __device__ __forceinline__ void begin(const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask) {
        ngroups = (nb + 15) >> 4;
        nvb = ((ngroups + 7) >> 3) << 3;
        vb = (int)blockIdx.x;
        fetch(vb, bricks, nb, mask, a, m);
        fetch(vb + (int)gridDim.x, bricks, nb, mask, an, mn);
    }
    __device__ __forceinline__ bool valid() const { return vb < nvb; }
    __device__ __forceinline__ void next(const int *__restrict__ bricks, int nb, const uint8_t *__restrict__ mask) {
        a = an; m = mn;
        vb += (int)gridDim.x;
        fetch(vb + (int)gridDim.x, bricks, nb, mask, an, mn);
    }
    // component c has a row among the lane's four indices / at index e of them
    __device__ __forceinline__ static bool any(unsigned mk, int c) { return ((mk >> c) & 0x01010101u) != 0u; }
    __device__ __forceinline__ static bool row(unsigned mk, int c, int e) { return ((mk >> (8 * e + c)) & 1u) != 0u; }
};
static inline int fv_brickv_grid(int nbricks, int cap) {
    const int g = (((nbricks + 15) / 16 + 7) / 8) * 8;
    return g < 8 ? 8 : (g < cap ? g : cap);
}

// Scalar prologue of the update kernel (K2 of pcg_common.h), shared with k_pcg_update's logic (pcg_geo.inc): folds rmax(it-1), step(it-1), sigma(it),
// a, b, c(it), runs the stop test and the stall guard, forms alpha and beta.  Returns false when the launch must do nothing.
// lds: 8 doubles.  Every thread of the block must call it.
__device__ __forceinline__ bool d_update_scalars(const PcgScal &sc, int it_arg, double *lds, int &it, double &alpha_d, double &beta_d) {
    const int conv_now = *sc.conv, itB_now = it_arg >= 0 ? it_arg : *sc.itB;   // two independent loads
    if (conv_now >= 0) return false;
    it = itB_now;
    if (it >= sc.cap) return false;
    const int tid = d_tid256();
    const int grp = tid >> 5;  // 0 rmax(it-1), 1 step(it-1), 2 sig, 3 a, 4 b, 5 c
    double v = 0.0;
    if (grp < 6 && (it > 0 || grp > 1)) v = grp <= 1 ? sc.slot_max(sc.sig(it) - 2 * NSLOT + grp * NSLOT, tid & (NSLOT - 1)) : sc.slot_sum(sc.sig(it) + (grp - 2) * NSLOT, tid & (NSLOT - 1));
#pragma unroll
    for (int off = NSLOT / 2; off > 0; off >>= 1) {
        const double o = __shfl_down(v, off, NSLOT);
        v = grp <= 1 ? fmax(v, o) : v + o;
    }
    if (grp < 6 && (tid & (NSLOT - 1)) == 0) lds[grp] = v;
    __syncthreads();
    if (it > 0 && d_pass(sc, lds[0]) && d_steps_small(sc, it - 1, lds)) {   // (with the velocity criterion where the loop carries one: PcgScal::vel_tol)
        if (blockIdx.x == 0 && tid == 0) *sc.conv = it - 1;
        return false;
    }
    if (it > 0 && sc.best) {   // stall guard (PcgScal)
        const double bestNow = *sc.best, res = lds[0];
        if (bestNow <= (sc.stall_below > 0.0 ? sc.stall_below : 100.0 * sc.tol) && res > sc.stall_ratio * bestNow) {
            if (blockIdx.x == 0 && tid == 0) { *sc.stalled = 1; *sc.conv = it - 1; }
            return false;
        }
        if (blockIdx.x == 0 && tid == 0 && res < bestNow) *sc.best = res;
    }
    const double sg = lds[2], a = lds[3];
    alpha_d = a != 0.0 ? sg / a : 0.0;
    const double bdot = sc.noB ? a : lds[4];
    double est = sg - 2.0 * alpha_d * bdot + alpha_d * alpha_d * lds[5];
    if (!(est > 0.0)) est = 0.0;
    beta_d = sg != 0.0 ? est / sg : 0.0;
    return true;
}

// ---- host entry points of k_viscosity_brick.hip
int fv_build_bricks(flipv_context *c, const Lay &box);     // c->brickList / c->nBricks from c->vMaskB inside the launch box
int fv_brick_grid(const flipv_context *c, int nbricks, int cap);
template <typename T> void fv_brick_spmv(flipv_context *c, const PcgScal &sc, int it, bool rdot, int first = 0, int count = -1);   // list entries [first, first + count); count < 0: all
template <typename T> void fv_brick_init(flipv_context *c, const PcgScal &sc);
template <typename T> void fv_brick_update(flipv_context *c, const PcgScal &sc, int it);
void fv_brick_sweep_f32(flipv_context *c, float *const in[3], float *const out[3], int epi, const PcgScal &sc, int it_arg, float omega, int sig_shift, int first = 0, int count = -1);
template <typename T> void fv_brick_replace(flipv_context *c, const PcgScal &sc, int it_arg, int period, int withSigma, float *const z[3], float omega);
template <typename T> int fv_brick_refine(flipv_context *c, const PcgScal &sc, size_t scalBytes, bool outerExact, int flushMode = 0);
template <typename T> void fv_brick_flush_settle(flipv_context *c, const PcgScal &sc, bool takeBack);
template <typename T> void fv_brick_writeback(flipv_context *c, const Lay &R, int m, bool withAcc, float *dst);
