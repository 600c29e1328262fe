// k_grid.hip -- pointwise / stencil grid kernels of the FLIP substep (gfx950).
//
// Every kernel maps one wave to 64 consecutive i of one (j,k) row of the shared padded index space
// (flipv_internal.h): block (64,4,1), grid (ceil(PX/64), ceil(PY/4), PZ), so each global access of a wave is
// a whole number of aligned 256-byte lines and no integer division is needed to recover (i,j,k).
// All of these are HBM-bound with 5..20 B per face.
#include <chrono>
#include "flipv_internal.h"
#include "flipv_comm.h"

// ------------------------------------------------------------------ Array3d <-> device layout
// linear = the reference's Array3d order for lattice `lat` (flat = i + w*(j + h*k), array3d.h:397-400)
// `linear` holds the box [b0, b1) of the lattice, box-shaped
struct IBox { int lo[3], hi[3]; };
__device__ __forceinline__ bool d_in_box(const IBox &b, int i, int j, int k) {
    return i >= b.lo[0] && i < b.hi[0] && j >= b.lo[1] && j < b.hi[1] && k >= b.lo[2] && k < b.hi[2];
}
__device__ __forceinline__ size_t d_box_index(const IBox &b, int i, int j, int k) {
    return (size_t)(i - b.lo[0]) + (size_t)(b.hi[0] - b.lo[0]) * ((size_t)(j - b.lo[1]) + (size_t)(b.hi[1] - b.lo[1]) * (size_t)(k - b.lo[2]));
}
__global__ void k_unpack(Lay L, IBox b, const float *__restrict__ linear, float *__restrict__ dstf,
                         uint8_t *__restrict__ dstb) {
    IJK_OR_RETURN(L);
    float v = 0.0f;
    if (d_in_box(b, i, j, k)) v = linear[d_box_index(b, i, j, k)];
    if (dstf) dstf[c] = v;
    else dstb[c] = v != 0.0f;
}
__global__ void k_pack(Lay L, IBox b, const float *__restrict__ srcf, const uint8_t *__restrict__ srcb,
                       float *__restrict__ linear) {
    IJK_OR_RETURN(L);
    if (d_in_box(b, i, j, k)) linear[d_box_index(b, i, j, k)] = srcf ? srcf[c] : (srcb[c] ? 1.0f : 0.0f);
}

// ------------------------------------------------------------------ K2: liquid SDF into solids
// reference particlelevelset.cpp:127-139
__global__ void k_sdf_into_solids(Lay L, float *__restrict__ phi, const float *__restrict__ solid, float dx) {
    IJK_OR_RETURN(L);
    if (i >= L.I || j >= L.J || k >= L.K) return;
    const double dxd = (double)dx;
    if ((double)phi[c] < 0.5 * dxd) {
        if (d_solid_center(solid, L, c) < 0.0f) phi[c] = -0.5f * (float)dxd;
    }
}

// fill the cells of a cell-centred array (padding stays 0)
__global__ void k_fill_cells(Lay L, float *__restrict__ p, float v) {
    IJK_OR_RETURN(L);
    if (i < L.I && j < L.J && k < L.K) p[c] = v;
}

__global__ void k_fill_f32(float *__restrict__ p, size_t n, float v) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) p[t] = v;
}

// ------------------------------------------------------------------ K3 tail + K4: normalise and select
// reference fluidsimulation.cpp:423-437 (normalise where the sum of weights >= 1e-9) and :440-498
// (keep only faces bordering a phi<0 cell; set valid).  All three components in one launch.
__global__ void k_p2g_finalize(Lay L, const float *__restrict__ accU, const float *__restrict__ wgtU,
                               const float *__restrict__ accV, const float *__restrict__ wgtV,
                               const float *__restrict__ accW, const float *__restrict__ wgtW,
                               const float *__restrict__ phi, float *__restrict__ U, float *__restrict__ V,
                               float *__restrict__ W, uint8_t *__restrict__ vU, uint8_t *__restrict__ vV,
                               uint8_t *__restrict__ vW) {
    IJK_OR_RETURN(L);
    const float *acc[3] = {accU, accV, accW};
    const float *wgt[3] = {wgtU, wgtV, wgtW};
    float *out[3] = {U, V, W};
    uint8_t *val[3] = {vU, vV, vW};
#pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        int w, h, d;
        lat_dims(L, LAT_U + dir, w, h, d);
        if (i >= w || j >= h || k >= d) continue;
        const float weight = wgt[dir][c];
        float v = 0.0f;
        uint8_t ok = 0;
        if (!((double)weight < 1e-9) && d_face_borders_fluid(dir, i, j, k, L, phi)) {
            v = acc[dir][c] / weight;
            ok = 1;
        }
        out[dir][c] = v;
        val[dir][c] = ok;
    }
}

// ------------------------------------------------------------------ K5: layered extrapolation
// reference macvelocityfield.cpp:580-687.  The reference sweeps KNOWN interior cells and pushes their
// UNKNOWN neighbours on a list, then averages; that is a Jacobi step per layer.  Here a cell's status is a
// stamp: 0 = valid input, L+1 = filled in layer L, 255 = unknown, 254 = unknown on the array border (frozen,
// :592-595).  "Known at the start of layer L" == stamp <= L, so the stamp array can be updated in place:
// a neighbour written concurrently carries 255 or L+1, both > L.
// Activity blocks of ACT_B^3 indices: a block is active if it or one of its 26 neighbours holds a valid face, i.e. if a
// cell of it can be reached by `layers` <= ACT_B extrapolation layers.  The layer kernel skips every other block
// (the liquid fills a few percent of the box; the full sweep was 14 x 216 us per substep at 256^3).
constexpr int ACT_B = 8;
struct ActGrid { int nx, ny, nz, ox, oy, oz; };   // blocks of the allocated box; (ox,oy,oz) = its first index
__device__ __forceinline__ int d_act_index(const ActGrid &A, int i, int j, int k) {
    return ((i - A.ox) / ACT_B) + A.nx * (((j - A.oy) / ACT_B) + A.ny * ((k - A.oz) / ACT_B));
}

__global__ void k_extrap_init(Lay L, ActGrid A, const uint8_t *__restrict__ vU, const uint8_t *__restrict__ vV,
                              const uint8_t *__restrict__ vW, uint8_t *__restrict__ sU, uint8_t *__restrict__ sV,
                              uint8_t *__restrict__ sW, uint8_t *__restrict__ act, uint8_t *__restrict__ unk) {
    IJK_OR_RETURN(L);
    const uint8_t *val[3] = {vU, vV, vW};
    uint8_t *st[3] = {sU, sV, sW};
    bool any = false, fillable = false;
#pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        int w, h, d;
        lat_dims(L, LAT_U + dir, w, h, d);
        if (i >= w || j >= h || k >= d) continue;
        const bool border = i == 0 || j == 0 || k == 0 || i == w - 1 || j == h - 1 || k == d - 1;
        const bool v = val[dir][c] != 0;
        any = any || v;
        fillable = fillable || (!v && !border);
        st[dir][c] = v ? 0 : (border ? 254 : 255);
    }
    if (any) act[d_act_index(A, i, j, k)] = 1;  // same value from every writer
    if (fillable) unk[d_act_index(A, i, j, k)] = 1;   // blocks without a single unknown face (the liquid's interior) have nothing to fill
}

// 3x3x3 dilation; in a multi-rank run every block within ACT_B entries of a face of the owned box that has a neighbour is
// active as well (the neighbour's valid faces can reach across the cut; its masks are only known one entry deep)
struct CutFaces { int lo[3], hi[3], has_lo[3], has_hi[3]; };   // owned box and which of its faces are interior cuts
__global__ void k_act_dilate(ActGrid A, const uint8_t *__restrict__ in, uint8_t *__restrict__ out, CutFaces F) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.nx * A.ny * A.nz) return;
    const int bx = t % A.nx, by = (t / A.nx) % A.ny, bz = t / (A.nx * A.ny);
    int any = 0;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dxx = -1; dxx <= 1; dxx++) {
                const int x = bx + dxx, y = by + dy, z = bz + dz;
                if (x >= 0 && y >= 0 && z >= 0 && x < A.nx && y < A.ny && z < A.nz) any |= in[x + A.nx * (y + A.ny * z)];
            }
    const int b3[3] = {bx, by, bz}, o3[3] = {A.ox, A.oy, A.oz};
    for (int a = 0; a < 3; a++) {
        const int lo = o3[a] + b3[a] * ACT_B, hi = lo + ACT_B - 1;  // entries of this block along the axis
        if (F.has_lo[a] && hi >= F.lo[a] - ACT_B && lo <= F.lo[a] + ACT_B - 1) any = 1;
        if (F.has_hi[a] && hi >= F.hi[a] - ACT_B && lo <= F.hi[a] + ACT_B - 1) any = 1;
    }
    out[t] = (uint8_t)any;
}

// the active blocks that hold entries of the launch box of R, compacted by one workgroup: list[0] = count, ids from list[1]
__global__ __launch_bounds__(1024) void k_act_compact(ActGrid A, const uint8_t *__restrict__ act, const uint8_t *__restrict__ unk, Lay R,
                                                      int *__restrict__ list) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int n = A.nx * A.ny * A.nz;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int t = start + (int)threadIdx.x;
        int f = 0;
        if (t < n && act[t] && unk[t]) {
            const int bx = t % A.nx, by = (t / A.nx) % A.ny, bz = t / (A.nx * A.ny);
            const int x0 = A.ox + bx * ACT_B, y0 = A.oy + by * ACT_B, z0 = A.oz + bz * ACT_B;
            f = x0 < R.ie && x0 + ACT_B > R.ib && y0 < R.je && y0 + ACT_B > R.jb && z0 < R.ke && z0 + ACT_B > R.kb;
        }
        const unsigned long long m = __ballot(f);
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int woff = 0, total = 0;
        for (int q = 0; q < 16; q++) { if (q < wv) woff += wsum[q]; total += wsum[q]; }
        if (f) list[1 + base + woff + __popcll(m & ((1ull << lane) - 1ull))] = t;
        __syncthreads();
        if (threadIdx.x == 0) base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) list[0] = base;
}

// One workgroup per active block (grid-stride over the compacted list): 8 x 8 lanes in (i, j), four planes at a time.  A
// layer only reads stamps <= layer and writes layer + 1, so the order of the cells within a layer does not matter -- and the
// block's stamps with a one-entry rim (10^3 bytes per lattice) can be staged in LDS up front: whatever a neighbouring block
// writes meanwhile is layer + 1, which reads like 255 here.  (7 byte loads per cell and lattice -> 2; the P2G + extrapolation phase 0.74 -> 0.69 ms at 256^3.)
__global__ __launch_bounds__(256) void k_extrap_layer(Lay L, ActGrid A, const int *__restrict__ list, float *__restrict__ U, float *__restrict__ V,
                               float *__restrict__ W, uint8_t *__restrict__ sU, uint8_t *__restrict__ sV, uint8_t *__restrict__ sW,
                               int layer) {
    constexpr int H = ACT_B + 2, HH = H * H, HHH = H * H * H;
    __shared__ uint8_t S[3][HHH + 8];
    float *g[3] = {U, V, W};
    uint8_t *st[3] = {sU, sV, sW};
    const long off[6] = {-1, 1, -L.sy, L.sy, -L.sz, L.sz};
    const int soff[6] = {-1, 1, -H, H, -HH, HH};
    const int count = list[0];
    const int lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x >> 6;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const int id = list[1 + b];
        const int bx = id % A.nx, by = (id / A.nx) % A.ny, bz = id / (A.nx * A.ny);
        const int i0 = A.ox + bx * ACT_B, j0 = A.oy + by * ACT_B, k0 = A.oz + bz * ACT_B;
        __syncthreads();   // (the previous block's readers are done)
        for (int t = threadIdx.x; t < HHH; t += 256) {
            const int x = t % H, y = (t / H) % H, z = t / HH;
            const int gi = i0 - 1 + x, gj = j0 - 1 + y, gk = k0 - 1 + z;
            // entries the cells of the launch box can read: the box and one entry around it (inside the allocation's guard zone)
            const bool in = gi >= L.ib - 1 && gi <= L.ie && gj >= L.jb - 1 && gj <= L.je && gk >= L.kb - 1 && gk <= L.ke;
            const size_t c = in ? gidx(L, gi, gj, gk) : 0;
#pragma unroll
            for (int dir = 0; dir < 3; dir++) S[dir][t] = in ? st[dir][c] : (uint8_t)255;
        }
        __syncthreads();
        const int i = i0 + lx, j = j0 + ly;
        if (i < L.ib || i >= L.ie || j < L.jb || j >= L.je) continue;   // (no barrier below this line inside the iteration)
#pragma unroll
        for (int p = 0; p < ACT_B; p += 4) {
            const int k = k0 + p + lz;
            if (k < L.kb || k >= L.ke) continue;
            const size_t c = gidx(L, i, j, k);
            const int sc = (lx + 1) + H * (ly + 1) + HH * (p + lz + 1);
#pragma unroll
            for (int dir = 0; dir < 3; dir++) {
                int w, h, d;
                lat_dims(L, LAT_U + dir, w, h, d);
                if (i >= w || j >= h || k >= d) continue;
                if (S[dir][sc] != 255) continue;  // only unknown, non-border cells are ever filled (so all six neighbours exist)
                // a neighbour can only trigger this cell if it is an interior cell of the array (:604-606)
                const bool nin[6] = {i - 1 >= 1, i + 1 <= w - 2, j - 1 >= 1, j + 1 <= h - 2, k - 1 >= 1, k + 1 <= d - 2};
                float sum = 0.0f;
                int cnt = 0;
                bool trigger = false;
#pragma unroll
                for (int q = 0; q < 6; q++) {  // order -i,+i,-j,+j,-k,+k (:671-676)
                    if (S[dir][sc + soff[q]] <= layer) {
                        sum += g[dir][(size_t)((long)c + off[q])];
                        cnt++;
                        trigger = trigger || nin[q];
                    }
                }
                if (trigger) {
                    g[dir][c] = sum / (float)cnt;
                    st[dir][c] = (uint8_t)(layer + 1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ K6: body force
// reference fluidsimulation.cpp:271-312
__global__ void k_body_force(Lay L, float *__restrict__ U, float *__restrict__ V, float *__restrict__ W,
                             const float *__restrict__ phi, float incU, float incV, float incW) {
    IJK_OR_RETURN(L);
    float *g[3] = {U, V, W};
    const float inc[3] = {incU, incV, incW};
#pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        int w, h, d;
        lat_dims(L, LAT_U + dir, w, h, d);
        if (i >= w || j >= h || k >= d) continue;
        if (d_face_borders_fluid(dir, i, j, k, L, phi)) g[dir][c] += inc[dir];
    }
}

// ------------------------------------------------------------------ K10: face weights
// reference fluidsimulation.cpp:549-582 with the node orders of meshlevelset.cpp:92-126
__global__ void k_weights(Lay L, const float *__restrict__ s, float *__restrict__ wU, float *__restrict__ wV,
                          float *__restrict__ wW) {
    IJK_OR_RETURN(L);
    const long sy = L.sy, sz = L.sz;
    if (i <= L.I && j < L.J && k < L.K) {
        const float f = d_frac4(s[c], s[c + sy], s[c + sz], s[c + sy + sz]);
        wU[c] = fmaxf(0.0f, fminf(1.0f - f, 1.0f));
    }
    if (i < L.I && j <= L.J && k < L.K) {
        const float f = d_frac4(s[c], s[c + sz], s[c + 1], s[c + 1 + sz]);
        wV[c] = fmaxf(0.0f, fminf(1.0f - f, 1.0f));
    }
    if (i < L.I && j < L.J && k <= L.K) {
        const float f = d_frac4(s[c], s[c + sy], s[c + 1], s[c + 1 + sy]);
        wW[c] = fmaxf(0.0f, fminf(1.0f - f, 1.0f));
    }
}

// ------------------------------------------------------------------ K13: pressure gradient
// reference fluidsimulation.cpp:598-688
__global__ void k_apply_pressure(Lay L, float *__restrict__ U, float *__restrict__ V, float *__restrict__ W,
                                 uint8_t *__restrict__ vU, uint8_t *__restrict__ vV, uint8_t *__restrict__ vW,
                                 const float *__restrict__ wU, const float *__restrict__ wV,
                                 const float *__restrict__ wW, const float *__restrict__ p,
                                 const float *__restrict__ phi, float dx, float dt, float minfrac) {
    IJK_OR_RETURN(L);
    float *g[3] = {U, V, W};
    uint8_t *val[3] = {vU, vV, vW};
    const float *wg[3] = {wU, wV, wW};
    const long back[3] = {1, L.sy, L.sz};
#pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        int w, h, d;
        lat_dims(L, LAT_U + dir, w, h, d);
        if (i >= w || j >= h || k >= d) continue;
        const int n = dir == 0 ? L.I : (dir == 1 ? L.J : L.K);
        const int cd = dir == 0 ? i : (dir == 1 ? j : k);
        float v = 0.0f;
        uint8_t ok = 0;
        if (cd >= 1 && cd < n && wg[dir][c] > 0.0f && d_face_borders_fluid(dir, i, j, k, L, phi)) {
            const size_t c0 = c - back[dir];
            const float p0 = p[c0], p1 = p[c];
            const float theta = fmaxf(d_frac2(phi[c0], phi[c]), minfrac);
            v = g[dir][c] + (-dt * (p1 - p0) / (dx * theta));
            ok = 1;
        }
        g[dir][c] = v;
        val[dir][c] = ok;
    }
}

// ------------------------------------------------------------------ K14: constrain
// reference fluidsimulation.cpp:696-729.  Padding entries have weight 0 and velocity 0 already.
__global__ void k_constrain(const float *__restrict__ wgt, float *__restrict__ vel, float *__restrict__ saved, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride)
        if (wgt[t] == 0.0f) {
            vel[t] = 0.0f;
            saved[t] = 0.0f;
        }
}

// ------------------------------------------------------------------ K16: CFL max-reduction
// reference fluidsimulation.cpp:241-269.  max is exact, so any order gives the reference's value
// (padding entries are 0 and cannot raise a max of absolute values).
__global__ void k_absmax3(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c3,
                          size_t n, unsigned *__restrict__ out_bits) {
    __shared__ double lds[4];
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.0f;
    for (; t < n; t += stride) m = fmaxf(m, fmaxf(fabsf(a[t]), fmaxf(fabsf(b[t]), fabsf(c3[t]))));
    const double r = block_max_256((double)m, lds);
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint((float)r));
}

// the same over a launch box (block-decomposed runs: only the entries the rank owns; its halo holds copies of the neighbours'
// values taken at another moment of the substep)
__global__ void k_absmax3_box(Lay L, const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c3,
                              unsigned *__restrict__ out_bits) {
    __shared__ double lds[4];
    IJK_OF_THREAD(L);
    float m = 0.0f;
    if (i < L.ie && j < L.je) {
        const size_t c = gidx(L, i, j, k);
        m = fmaxf(fabsf(a[c]), fmaxf(fabsf(b[c]), fabsf(c3[c])));
    }
    const double r = block_max_256((double)m, lds);
    if (threadIdx.x == 0 && threadIdx.y == 0 && r > 0.0) atomicMax(out_bits, __float_as_uint((float)r));
}

// =================================================================== host launchers
// Ranges: fv_range(c, h) = the planes this rank owns widened by h halo planes (the whole index space on one GPU).
static unsigned grid1d(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 2048 ? 2048 : (b ? b : 1));
}

int fv_unpack(flipv_context *c, int lat, const float *linear, float *dstf, uint8_t *dstb, const int lo[3], const int hi[3]) {
    IBox b;
    for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; }
    (void)lat;
    hipLaunchKernelGGL(k_unpack, GRID3(c->L), 0, c->stream, c->L, b, linear, dstf, dstb);   // the whole allocated box: zero outside [lo, hi)
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}
int fv_pack(flipv_context *c, int lat, const float *srcf, const uint8_t *srcb, float *linear, const int lo[3], const int hi[3]) {
    IBox b;
    for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; }
    (void)lat;
    Lay R = c->L;
    R.ib = lo[0]; R.ie = hi[0]; R.jb = lo[1]; R.je = hi[1]; R.kb = lo[2]; R.ke = hi[2];
    hipLaunchKernelGGL(k_pack, GRID3(R), 0, c->stream, R, b, srcf, srcb, linear);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_sdf_finish(flipv_context *c) {
    const Lay R = fv_range_liquid(c, 0, 0);
    hipLaunchKernelGGL(k_sdf_into_solids, GRID3(R), 0, c->stream, R, c->phi, c->solid, c->dx);
    return FLIPV_OK;
}

// ---- several memsets as one launch (flipv_internal.h: fv_fill_list)
struct FillDev { void *p[8]; unsigned long long bytes[8]; unsigned word[8]; int n; };
__global__ __launch_bounds__(256) void k_fill_list(FillDev f) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (int q = 0; q < f.n; q++) {
        uint8_t *b = (uint8_t *)f.p[q];
        const size_t n = f.bytes[q];
        const size_t head = n < 16 ? n : ((16 - ((size_t)b & 15)) & 15);   // bytes up to the first 16-byte boundary (short jobs: bytewise altogether)
        const size_t vecs = (n - head) / 16, tail = n - head - vecs * 16;
        const uint4 w = make_uint4(f.word[q], f.word[q], f.word[q], f.word[q]);
        uint4 *v = (uint4 *)(b + head);
        for (size_t i = t; i < vecs; i += nt) v[i] = w;
        if (t < head) b[t] = (uint8_t)f.word[q];
        if (t < tail) b[head + vecs * 16 + t] = (uint8_t)f.word[q];
    }
}
int fv_fill_list(flipv_context *c, const FillJob *jobs, int n, hipStream_t st) {
    for (int at = 0; at < n; at += 8) {
        FillDev f;
        size_t most = 0;
        f.n = 0;
        for (int q = at; q < n && q < at + 8; q++) {
            if (!jobs[q].bytes) continue;
            const unsigned bt = (unsigned)jobs[q].byte & 255u;
            f.p[f.n] = jobs[q].p; f.bytes[f.n] = jobs[q].bytes; f.word[f.n] = bt * 0x01010101u;
            if (jobs[q].bytes > most) most = jobs[q].bytes;
            f.n++;
        }
        if (!f.n) continue;
        size_t nb = (most / 16 + 1023) / 1024;   // ~4 16-byte stores per thread on the largest job
        if (nb < 1) nb = 1;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(k_fill_list, dim3((unsigned)nb), dim3(256), 0, st ? st : c->stream, f);
    }
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

// ---- small reads without a copy dispatch (flipv_internal.h: fv_read_small)
struct PubDev { const int *src[6]; int words[6], at[6], n; };
__global__ __launch_bounds__(64) void k_publish(PubDev j, int *__restrict__ host, int *__restrict__ seq) {
    for (int q = 0; q < j.n; q++)
        for (int w = (int)threadIdx.x; w < j.words[q]; w += 64) __hip_atomic_store(host + j.at[q] + w, j.src[q][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();   // (every lane's stores have landed before lane 0 announces them ...
    __syncthreads();          //  ... and every lane has passed its fence: the block is one wave today, the barrier is what says so -- ADVICE r5)
    if (threadIdx.x == 0) {
        const int s = (int)((unsigned)*seq + 1u);
        *seq = s;
        __hip_atomic_store(host, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int fv_read_small(flipv_context *c, const ReadJob *jobs, int n) {
    if (n < 1 || n > 6) { c->err = "fv_read_small: 1..6 jobs"; return FLIPV_ERR_INVALID; }
    int need = 0;
    for (int q = 0; q < n; q++) need += jobs[q].words;
    if (need > FV_PUB_WORDS) {   // (a gather over very many ranks: the plain copies; the caller's synchronisation point completes them)
        for (int q = 0; q < n; q++) HIPCHK(c, hipMemcpyAsync(jobs[q].host, jobs[q].dev, (size_t)jobs[q].words * 4, hipMemcpyDeviceToHost, c->stream));
        return FLIPV_OK;
    }
    if (c->pubPendingN + n > 16 || c->pubUsed + need > FV_PUB_WORDS) {
        const int rc = fv_read_wait(c);
        if (rc) return rc;
    }
    PubDev j;
    j.n = n;
    for (int q = 0; q < n; q++) {
        j.src[q] = (const int *)jobs[q].dev; j.words[q] = jobs[q].words; j.at[q] = FV_PUB_DATA + c->pubUsed;
        c->pubPending[c->pubPendingN++] = {jobs[q].host, j.at[q], jobs[q].words};
        c->pubUsed += jobs[q].words;
    }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, j, c->d_pubMap, c->d_pubSeq);
    if (hipGetLastError() != hipSuccess) {   // nothing was enqueued: the jobs are dropped and the sequence number stays where the device's is
        c->pubPendingN -= n; c->pubUsed -= need;
        c->err = "fv_read_small: the publish kernel did not launch";
        return FLIPV_ERR_HIP;
    }
    c->pubSeq = (int)((unsigned)c->pubSeq + 1u);
    return FLIPV_OK;
}
int fv_read_capture(flipv_context *c, const void *dev, int words) {
    PubDev j;
    j.n = 1; j.src[0] = (const int *)dev; j.words[0] = words; j.at[0] = FV_PUB_REPLAY;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, j, c->d_pubMap, c->d_pubSeq);
    return hipGetLastError() == hipSuccess ? FLIPV_OK : FLIPV_ERR_HIP;
}
void fv_read_replayed(flipv_context *c) { c->pubSeq = (int)((unsigned)c->pubSeq + 1u); }
int fv_read_wait_seq(flipv_context *c, int want) {
    auto arrived = [&]() { return (int)((unsigned)__atomic_load_n(c->h_pub, __ATOMIC_ACQUIRE) - (unsigned)want) >= 0; };
    if (arrived()) return FLIPV_OK;
    // spin on the mapped word; hand over to the runtime's wait if the device takes long (long kernels in front of the read) or has failed
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool ok = false;
        for (int q = 0; q < 256 && !(ok = arrived()); q++) __builtin_ia32_pause();
        if (ok) return FLIPV_OK;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2500)) {   // (a replayed chunk of four multigrid-PCG iterations takes ~1 ms at 256^3)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (want - c->pubSeq == 0 && !arrived()) { c->err = "fv_read_wait: the stream drained without the published values"; return FLIPV_ERR_HIP; }
            return FLIPV_OK;
        }
    }
}
int fv_read_wait(flipv_context *c) {
    const int rc = fv_read_wait_seq(c, c->pubSeq);
    if (rc) {
        // the pending jobs' host pointers are mostly the callers' stack locals: never copy into them after an error return; and take the sequence number the
        // device actually reached, so that a later wait does not spin on a publication that was never made
        c->pubPendingN = 0; c->pubUsed = 0;
        if (hipStreamSynchronize(c->stream) == hipSuccess) c->pubSeq = __atomic_load_n(c->h_pub, __ATOMIC_ACQUIRE);
        return rc;
    }
    for (int q = 0; q < c->pubPendingN; q++)
        if (c->pubPending[q].host) memcpy(c->pubPending[q].host, c->h_pub + c->pubPending[q].at, (size_t)c->pubPending[q].words * 4);
    c->pubPendingN = 0;
    c->pubUsed = 0;
    return FLIPV_OK;
}

int fv_sync(flipv_context *c) {
    if (c->pubPendingN) { const int rc = fv_read_wait(c); if (rc) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FLIPV_OK;
}

int fv_fill(flipv_context *c, float *p, size_t n, float v) {
    hipLaunchKernelGGL(k_fill_f32, dim3(grid1d(n)), dim3(256), 0, c->stream, p, n, v);
    return FLIPV_OK;
}
int fv_fill_cells(flipv_context *c, float *p, float v, int halo) {
    const Lay R = fv_range(c, halo);
    hipLaunchKernelGGL(k_fill_cells, GRID3(R), 0, c->stream, R, p, v);
    return FLIPV_OK;
}

int fv_fill_cells_liquid(flipv_context *c, float *p, float v, int halo, int site) {
    const Lay R = fv_range_liquid(c, halo, site);
    hipLaunchKernelGGL(k_fill_cells, GRID3(R), 0, c->stream, R, p, v);
    return FLIPV_OK;
}

int fv_p2g_finalize(flipv_context *c) {
    const Lay R = fv_range_liquid(c, 0, 1);
    hipLaunchKernelGGL(k_p2g_finalize, GRID3(R), 0, c->stream, R, c->accU, c->wgtU, c->accV, c->wgtV, c->accW, c->wgtW, c->phi,
                       c->U, c->V, c->W, c->vU, c->vV, c->vW);
    return FLIPV_OK;
}

int fv_extrapolate(flipv_context *c) {
    const int layers = c->prm.extrapolation_layers > 0 ? c->prm.extrapolation_layers : (int)ceilf(c->prm.cfl_number) + 2;
    const HaloArray in[6] = {{c->U, 4}, {c->V, 4}, {c->W, 4}, {c->vU, 1}, {c->vV, 1}, {c->vW, 1}};
    int rc = fv_halo_copy(c, in, 6, 1);
    if (rc) return rc;
    const Lay R1 = fv_range_liquid(c, 1, 2), R0 = fv_range_liquid(c, 0, 2);
    // activity blocks: flagged by the init sweep, dilated once per ACT_B layers
    ActGrid A;
    A.nx = (c->L.PX + ACT_B - 1) / ACT_B; A.ny = (c->L.PY + ACT_B - 1) / ACT_B; A.nz = (c->L.PZ + ACT_B - 1) / ACT_B;
    A.ox = c->L.ox; A.oy = c->L.oy; A.oz = c->L.oz;
    CutFaces F;
    for (int a = 0; a < 3; a++) {
        F.lo[a] = c->L.olo[a]; F.hi[a] = c->L.ohi[a];
        F.has_lo[a] = c->comm && c->pcoord[a] > 0;
        F.has_hi[a] = c->comm && c->pcoord[a] < c->pgrid[a] - 1;
    }
    const int nact = A.nx * A.ny * A.nz;
    uint8_t *actA = c->actFlags, *actB = c->actFlags + nact, *unk = c->actFlags + 2 * (size_t)nact;
    { const FillJob z[2] = {{actA, (size_t)nact, 0}, {unk, (size_t)nact, 0}}; if ((rc = fv_fill_list(c, z, 2))) return rc; }
    hipLaunchKernelGGL(k_extrap_init, GRID3(R1), 0, c->stream, R1, A, c->vU, c->vV, c->vW, c->stampU, c->stampV, c->stampW, actA, unk);
    for (int q = 0; q < (layers + ACT_B - 1) / ACT_B; q++) {
        hipLaunchKernelGGL(k_act_dilate, dim3(cdiv(nact, 256)), dim3(256), 0, c->stream, A, actA, actB, F);
        uint8_t *t = actA; actA = actB; actB = t;
    }
    const HaloArray lay[6] = {{c->U, 4}, {c->V, 4}, {c->W, 4}, {c->stampU, 1}, {c->stampV, 1}, {c->stampW, 1}};
    hipLaunchKernelGGL(k_act_compact, dim3(1), dim3(1024), 0, c->stream, A, actA, unk, R0, c->actList);
    const int ngrid = nact < 4096 ? nact : 4096;
    for (int q = 0; q < layers; q++) {
        hipLaunchKernelGGL(k_extrap_layer, dim3(ngrid), dim3(256), 0, c->stream, R0, A, c->actList, c->U, c->V, c->W, c->stampU, c->stampV,
                           c->stampW, q);
        rc = fv_halo_copy(c, lay, 6, 1);
        if (rc) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_body_force(flipv_context *c, float dt) {
    const Lay R = fv_range_liquid(c, 1, 3);
    hipLaunchKernelGGL(k_body_force, GRID3(R), 0, c->stream, R, c->U, c->V, c->W, c->phi, c->gravity[0] * dt, c->gravity[1] * dt,
                       c->gravity[2] * dt);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_compute_weights(flipv_context *c) {
    const Lay R = fv_range(c, 2);
    hipLaunchKernelGGL(k_weights, GRID3(R), 0, c->stream, R, c->solid, c->wU, c->wV, c->wW);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_apply_pressure(flipv_context *c, float dt) {
    const Lay R = fv_range_liquid(c, 0, 6);
    hipLaunchKernelGGL(k_apply_pressure, GRID3(R), 0, c->stream, R, c->U, c->V, c->W, c->vU, c->vV, c->vW, c->wU, c->wV, c->wW,
                       c->pressure, c->phi, c->dx, dt, c->prm.min_frac);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_constrain(flipv_context *c) {
    const Lay R = fv_range_liquid(c, 1, 7);   // (whole planes of its k-range)
    const size_t off = plane_off(c->L, R.kb), n = (size_t)(R.ke - R.kb) * c->L.sz;   // whole allocated planes (pointwise; harmless in the halo)
    hipLaunchKernelGGL(k_constrain, dim3(grid1d(n)), dim3(256), 0, c->stream, c->wU + off, c->U + off, c->sU + off, n);
    hipLaunchKernelGGL(k_constrain, dim3(grid1d(n)), dim3(256), 0, c->stream, c->wV + off, c->V + off, c->sV + off, n);
    hipLaunchKernelGGL(k_constrain, dim3(grid1d(n)), dim3(256), 0, c->stream, c->wW + off, c->W + off, c->sW + off, n);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_cfl(flipv_context *c, float *dt_out) {
    unsigned *bits = (unsigned *)(c->d_flags + 3);
    const Lay R = fv_range(c, 0);
    const size_t off = plane_off(c->L, R.kb), n = (size_t)(R.ke - R.kb) * c->L.sz;
    HIPCHK(c, hipMemsetAsync(bits, 0, sizeof(unsigned), c->stream));
    if (c->comm) hipLaunchKernelGGL(k_absmax3_box, GRID3(R), 0, c->stream, R, c->U, c->V, c->W, bits);
    else hipLaunchKernelGGL(k_absmax3, dim3(grid1d(n)), dim3(256), 0, c->stream, c->U + off, c->V + off, c->W + off, n, bits);
    FV_READ(c, c->h_flags + 3, bits, sizeof(unsigned));
    FV_SYNC(c);
    unsigned b = *(unsigned *)(c->h_flags + 3);
    float maxvel;
    memcpy(&maxvel, &b, 4);
    int rc = fv_allreduce_max_f32(c, &maxvel);
    if (rc) return rc;
    // (float)((_CFLConditionNumber * _dx) / maxvel): +inf on a zero field (fluidsimulation.cpp:268)
    *dt_out = (float)((c->prm.cfl_number * c->dx) / maxvel);
    return FLIPV_OK;
}
