// k_grid.hip -- pointwise / stencil grid kernels of the FLIP substep (gfx950).
//
// Every kernel maps one wave to 64 consecutive i of one (j,k) row: block (64,4,1), grid
// (ceil(w/64), ceil(h/4), d), so each global access of a wave is one or two 256-byte lines and no
// integer division is needed to recover (i,j,k).  All of these are HBM-bound with 5..20 B per face.
#include "flipv_internal.h"

#define GRID3(w, h, d) dim3(cdiv((w), 64), cdiv((h), 4), (unsigned)(d)), dim3(64, 4, 1)

// ------------------------------------------------------------------ K2: liquid SDF into solids
// reference particlelevelset.cpp:127-139
__global__ void k_sdf_into_solids(float *__restrict__ phi, const float *__restrict__ solid, int I, int J, int K,
                                  float dx) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= I || j >= J) return;
    const size_t c = DIDX(i, j, k, I, J);
    const double dxd = (double)dx;
    if ((double)phi[c] < 0.5 * dxd) {
        if (d_solid_center(solid, i, j, k, I, J) < 0.0f) phi[c] = -0.5f * (float)dxd;
    }
}

__global__ void k_fill_f32(float *__restrict__ p, size_t n, float v) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) p[t] = v;
}

// ------------------------------------------------------------------ K3 tail + K4: normalise and select
// reference fluidsimulation.cpp:423-437 (normalise where sum of weights >= 1e-9) and :440-498
// (keep only faces bordering a phi<0 cell; set valid).
__global__ void k_p2g_finalize(int dir, const float *__restrict__ acc, const float *__restrict__ wgt,
                               const float *__restrict__ phi, float *__restrict__ out, uint8_t *__restrict__ valid,
                               int I, int J, int K) {
    const int w = I + (dir == 0), h = J + (dir == 1);
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    const size_t f = DIDX(i, j, k, w, h);
    const float weight = wgt[f];
    float v = 0.0f;
    uint8_t ok = 0;
    if (!((double)weight < 1e-9) && d_face_borders_fluid(dir, i, j, k, I, J, K, phi)) {
        v = acc[f] / weight;
        ok = 1;
    }
    out[f] = v;
    valid[f] = ok;
}

// ------------------------------------------------------------------ K5: layered extrapolation
// reference macvelocityfield.cpp:580-687.  The reference sweeps KNOWN interior cells and pushes their
// UNKNOWN neighbours on a list, then averages; that is a Jacobi step per layer.  Here a cell's status is a
// stamp: 0 = valid input, L+1 = filled in layer L, 255 = unknown, 254 = unknown on the array border (frozen,
// :592-595).  "Known at the start of layer L" == stamp <= L, so the stamp array can be updated in place:
// a neighbour written concurrently carries 255 or L+1, both > L.
__global__ void k_extrap_init(const uint8_t *__restrict__ valid, uint8_t *__restrict__ stamp, int w, int h, int d) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    const size_t c = DIDX(i, j, k, w, h);
    const bool border = i == 0 || j == 0 || k == 0 || i == w - 1 || j == h - 1 || k == d - 1;
    stamp[c] = valid[c] ? 0 : (border ? 254 : 255);
}

__global__ void k_extrap_layer(float *__restrict__ g, uint8_t *__restrict__ stamp, int w, int h, int d, int L) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    const size_t c = DIDX(i, j, k, w, h);
    if (stamp[c] != 255) return;  // only unknown, non-border cells are ever filled
    const long sx = 1, sy = w, sz = (long)w * h;
    const long off[6] = {-sx, sx, -sy, sy, -sz, sz};
    // a neighbour can only trigger this cell if it is an interior cell of the array (:604-606)
    const bool nin[6] = {i - 1 >= 1, i + 1 <= w - 2, j - 1 >= 1, j + 1 <= h - 2, k - 1 >= 1, k + 1 <= d - 2};
    const bool rowin_i = (j >= 1 && j <= h - 2 && k >= 1 && k <= d - 2);
    const bool rowin_j = (i >= 1 && i <= w - 2 && k >= 1 && k <= d - 2);
    const bool rowin_k = (i >= 1 && i <= w - 2 && j >= 1 && j <= h - 2);
    const bool other[6] = {rowin_i, rowin_i, rowin_j, rowin_j, rowin_k, rowin_k};
    float sum = 0.0f;
    int count = 0;
    bool trigger = false;
#pragma unroll
    for (int q = 0; q < 6; q++) {  // order -i,+i,-j,+j,-k,+k (:671-676); cell is interior so all six exist
        const size_t nb = (size_t)((long)c + off[q]);
        if (stamp[nb] <= L) {
            sum += g[nb];
            count++;
            trigger = trigger || (nin[q] && other[q]);
        }
    }
    if (trigger) {
        g[c] = sum / (float)count;
        stamp[c] = (uint8_t)(L + 1);
    }
}

// ------------------------------------------------------------------ K6: body force
// reference fluidsimulation.cpp:271-312
__global__ void k_body_force(int dir, float *__restrict__ vel, const float *__restrict__ phi, int I, int J, int K,
                             float inc) {
    const int w = I + (dir == 0), h = J + (dir == 1);
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    if (d_face_borders_fluid(dir, i, j, k, I, J, K, phi)) vel[DIDX(i, j, k, w, h)] += inc;
}

// ------------------------------------------------------------------ K10: face weights
// reference fluidsimulation.cpp:549-582 with the node orders of meshlevelset.cpp:92-126
__global__ void k_weights(int dir, const float *__restrict__ s, float *__restrict__ wout, int I, int J, int K) {
    const int w = I + (dir == 0), h = J + (dir == 1);
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    const int nw = I + 1, nh = J + 1;
    float f;
    if (dir == 0)
        f = d_frac4(s[DIDX(i, j, k, nw, nh)], s[DIDX(i, j + 1, k, nw, nh)], s[DIDX(i, j, k + 1, nw, nh)],
                    s[DIDX(i, j + 1, k + 1, nw, nh)]);
    else if (dir == 1)
        f = d_frac4(s[DIDX(i, j, k, nw, nh)], s[DIDX(i, j, k + 1, nw, nh)], s[DIDX(i + 1, j, k, nw, nh)],
                    s[DIDX(i + 1, j, k + 1, nw, nh)]);
    else
        f = d_frac4(s[DIDX(i, j, k, nw, nh)], s[DIDX(i, j + 1, k, nw, nh)], s[DIDX(i + 1, j, k, nw, nh)],
                    s[DIDX(i + 1, j + 1, k, nw, nh)]);
    wout[DIDX(i, j, k, w, h)] = fmaxf(0.0f, fminf(1.0f - f, 1.0f));
}

// ------------------------------------------------------------------ K13: pressure gradient
// reference fluidsimulation.cpp:598-688
__global__ void k_apply_pressure(int dir, float *__restrict__ vel, uint8_t *__restrict__ valid,
                                 const float *__restrict__ wgt, const float *__restrict__ p,
                                 const float *__restrict__ phi, int I, int J, int K, float dx, float dt,
                                 float minfrac) {
    const int w = I + (dir == 0), h = J + (dir == 1);
    const int n = dir == 0 ? I : (dir == 1 ? J : K);
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= w || j >= h) return;
    const int cdir = dir == 0 ? i : (dir == 1 ? j : k);
    const size_t f = DIDX(i, j, k, w, h);
    float v = 0.0f;
    uint8_t ok = 0;
    if (cdir >= 1 && cdir < n && wgt[f] > 0.0f && d_face_borders_fluid(dir, i, j, k, I, J, K, phi)) {
        const size_t c1 = DIDX(i, j, k, I, J);
        const size_t c0 = DIDX(i - (dir == 0), j - (dir == 1), k - (dir == 2), I, J);
        const float p0 = p[c0], p1 = p[c1];
        const float theta = fmaxf(d_frac2(phi[c0], phi[c1]), minfrac);
        v = vel[f] + (-dt * (p1 - p0) / (dx * theta));
        ok = 1;
    }
    vel[f] = v;
    valid[f] = ok;
}

// ------------------------------------------------------------------ K14: constrain
// reference fluidsimulation.cpp:696-729
__global__ void k_constrain(const float *__restrict__ wgt, float *__restrict__ vel, float *__restrict__ saved,
                            size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride)
        if (wgt[t] == 0.0f) {
            vel[t] = 0.0f;
            saved[t] = 0.0f;
        }
}

// ------------------------------------------------------------------ K16: CFL max-reduction
// reference fluidsimulation.cpp:241-269.  max is exact, so any order gives the reference's value.
__global__ void k_absmax(const float *__restrict__ a, size_t n, unsigned *__restrict__ out_bits) {
    __shared__ double lds[4];
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.0f;
    for (; t < n; t += stride) m = fmaxf(m, fabsf(a[t]));
    const double r = block_max_256((double)m, lds);
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint((float)r));
}

// =================================================================== host launchers
static unsigned grid1d(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 2048 ? 2048 : (b ? b : 1));
}

int fv_sdf_finish(flipv_context *c) {
    const Dims &d = c->d;
    hipLaunchKernelGGL(k_sdf_into_solids, GRID3(d.I, d.J, d.K), 0, c->stream, c->phi, c->solid, d.I, d.J, d.K, c->dx);
    return FLIPV_OK;
}

int fv_fill(flipv_context *c, float *p, size_t n, float v) {
    hipLaunchKernelGGL(k_fill_f32, dim3(grid1d(n)), dim3(256), 0, c->stream, p, n, v);
    return FLIPV_OK;
}

int fv_p2g_finalize(flipv_context *c) {
    const Dims &d = c->d;
    float *acc[3] = {c->accU, c->accV, c->accW}, *wg[3] = {c->wgtU, c->wgtV, c->wgtW};
    float *out[3] = {c->U, c->V, c->W};
    uint8_t *val[3] = {c->vU, c->vV, c->vW};
    for (int dir = 0; dir < 3; dir++) {
        const int w = d.I + (dir == 0), h = d.J + (dir == 1), dd = d.K + (dir == 2);
        hipLaunchKernelGGL(k_p2g_finalize, GRID3(w, h, dd), 0, c->stream, dir, acc[dir], wg[dir], c->phi, out[dir],
                           val[dir], d.I, d.J, d.K);
    }
    return FLIPV_OK;
}

int fv_extrapolate(flipv_context *c) {
    const Dims &d = c->d;
    int layers = c->prm.extrapolation_layers > 0 ? c->prm.extrapolation_layers : (int)ceilf(c->prm.cfl_number) + 2;
    float *g[3] = {c->U, c->V, c->W};
    uint8_t *val[3] = {c->vU, c->vV, c->vW}, *st[3] = {c->stampU, c->stampV, c->stampW};
    for (int dir = 0; dir < 3; dir++) {
        const int w = d.I + (dir == 0), h = d.J + (dir == 1), dd = d.K + (dir == 2);
        hipLaunchKernelGGL(k_extrap_init, GRID3(w, h, dd), 0, c->stream, val[dir], st[dir], w, h, dd);
    }
    for (int L = 0; L < layers; L++)
        for (int dir = 0; dir < 3; dir++) {
            const int w = d.I + (dir == 0), h = d.J + (dir == 1), dd = d.K + (dir == 2);
            hipLaunchKernelGGL(k_extrap_layer, GRID3(w, h, dd), 0, c->stream, g[dir], st[dir], w, h, dd, L);
        }
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_body_force(flipv_context *c, float dt) {
    const Dims &d = c->d;
    float *g[3] = {c->U, c->V, c->W};
    for (int dir = 0; dir < 3; dir++) {
        const int w = d.I + (dir == 0), h = d.J + (dir == 1), dd = d.K + (dir == 2);
        const float inc = c->gravity[dir] * dt;
        hipLaunchKernelGGL(k_body_force, GRID3(w, h, dd), 0, c->stream, dir, g[dir], c->phi, d.I, d.J, d.K, inc);
    }
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_compute_weights(flipv_context *c) {
    const Dims &d = c->d;
    float *wg[3] = {c->wU, c->wV, c->wW};
    for (int dir = 0; dir < 3; dir++) {
        const int w = d.I + (dir == 0), h = d.J + (dir == 1), dd = d.K + (dir == 2);
        hipLaunchKernelGGL(k_weights, GRID3(w, h, dd), 0, c->stream, dir, c->solid, wg[dir], d.I, d.J, d.K);
    }
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_apply_pressure(flipv_context *c, float dt) {
    const Dims &d = c->d;
    float *g[3] = {c->U, c->V, c->W}, *wg[3] = {c->wU, c->wV, c->wW};
    uint8_t *val[3] = {c->vU, c->vV, c->vW};
    for (int dir = 0; dir < 3; dir++) {
        const int w = d.I + (dir == 0), h = d.J + (dir == 1), dd = d.K + (dir == 2);
        hipLaunchKernelGGL(k_apply_pressure, GRID3(w, h, dd), 0, c->stream, dir, g[dir], val[dir], wg[dir], c->pressure,
                           c->phi, d.I, d.J, d.K, c->dx, dt, c->prm.min_frac);
    }
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_constrain(flipv_context *c) {
    const Dims &d = c->d;
    hipLaunchKernelGGL(k_constrain, dim3(grid1d(d.nu())), dim3(256), 0, c->stream, c->wU, c->U, c->sU, d.nu());
    hipLaunchKernelGGL(k_constrain, dim3(grid1d(d.nv())), dim3(256), 0, c->stream, c->wV, c->V, c->sV, d.nv());
    hipLaunchKernelGGL(k_constrain, dim3(grid1d(d.nw())), dim3(256), 0, c->stream, c->wW, c->W, c->sW, d.nw());
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_cfl(flipv_context *c, float *dt_out) {
    const Dims &d = c->d;
    unsigned *bits = (unsigned *)c->d_flags;
    HIPCHK(c, hipMemsetAsync(bits, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(k_absmax, dim3(grid1d(d.nu())), dim3(256), 0, c->stream, c->U, d.nu(), bits);
    hipLaunchKernelGGL(k_absmax, dim3(grid1d(d.nv())), dim3(256), 0, c->stream, c->V, d.nv(), bits);
    hipLaunchKernelGGL(k_absmax, dim3(grid1d(d.nw())), dim3(256), 0, c->stream, c->W, d.nw(), bits);
    HIPCHK(c, hipMemcpyAsync(c->h_flags, bits, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned b = *(unsigned *)c->h_flags;
    float maxvel;
    memcpy(&maxvel, &b, 4);
    // (float)((_CFLConditionNumber * _dx) / maxvel): +inf on a zero field (fluidsimulation.cpp:268)
    *dt_out = (float)((c->prm.cfl_number * c->dx) / maxvel);
    return FLIPV_OK;
}
