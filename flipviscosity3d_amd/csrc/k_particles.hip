// k_particles.hip -- particle <-> grid kernels (gfx950): liquid SDF scatter-min (K1), P2G scatter-add
// (K3), G2P + FLIP/PIC update + RK2 advection + solid push-out (K15).
#include "flipv_internal.h"
#include "flipv_comm.h"

// Grid3d::positionToGridIndex(vec3, double dx): float coordinate promoted, times 1/dx, floor
// (reference grid3d.h:60-65).  Done in fp64 exactly like the reference so that a particle lands in the
// same cell on both sides.
__device__ __forceinline__ int d_pos_index(float p, double invdx) { return (int)floor((double)p * invdx); }

// atomic min on a float through integer punning: order-free, so K1 is bit-reproducible
__device__ __forceinline__ void atomic_min_f32(float *addr, float v) {
    if (v >= 0.0f)
        atomicMin((int *)addr, __float_as_int(v));
    else
        atomicMax((unsigned *)addr, __float_as_uint(v));
}

// ------------------------------------------------------------------ K1: liquid SDF from particles
// reference particlelevelset.cpp:98-125
__global__ void k_sdf_scatter(Lay L, const float *__restrict__ aos6, size_t n, float *__restrict__ phi, float dx,
                              float radius) {
    const int I = L.I, J = L.J, K = L.K;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double dxd = (double)dx, invdx = 1.0 / dxd, hw = 0.5 * dxd;
    const float px = aos6[6 * p], py = aos6[6 * p + 1], pz = aos6[6 * p + 2];
    const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
    const int i0 = max(0, gi - 1), j0 = max(0, gj - 1), k0 = max(0, gk - 1);
    const int i1 = min(gi + 1, I - 1), j1 = min(gj + 1, J - 1), k1 = min(gk + 1, K - 1);
    for (int k = k0; k <= k1; k++) {
        const float vz = (float)(k * dxd + hw) - pz;
        for (int j = j0; j <= j1; j++) {
            const float vy = (float)(j * dxd + hw) - py;
            for (int i = i0; i <= i1; i++) {
                const float vx = (float)(i * dxd + hw) - px;
                const float dist = sqrtf(vx * vx + vy * vy + vz * vz) - radius;
                float *a = &phi[gidx(L, i, j, k)];
                if (dist < *a) atomic_min_f32(a, dist);  // plain pre-test only skips useless atomics
            }
        }
    }
}

// ------------------------------------------------------------------ K3: particle -> grid
// reference fluidsimulation.cpp:364-420 (all three components in one pass over the particles).
// v1: one thread per particle, global fp32 atomics (hardware global_atomic_add_f32).
__global__ void k_p2g_scatter(Lay L, const float *__restrict__ aos6, size_t n, float *__restrict__ accU,
                              float *__restrict__ wgtU, float *__restrict__ accV, float *__restrict__ wgtV,
                              float *__restrict__ accW, float *__restrict__ wgtW, float dx) {
    const int I = L.I, J = L.J, K = L.K;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double dxd = (double)dx, invdx = 1.0 / dxd;
    const float hdx = (float)(0.5 * dx);
    const float r = dx, rsq = r * r;
    const float coef1 = (4.0f / 9.0f) * (1.0f / (r * r * r * r * r * r));
    const float coef2 = (17.0f / 9.0f) * (1.0f / (r * r * r * r));
    const float coef3 = (22.0f / 9.0f) * (1.0f / (r * r));
    const float P[3] = {aos6[6 * p], aos6[6 * p + 1], aos6[6 * p + 2]};
    const float Vv[3] = {aos6[6 * p + 3], aos6[6 * p + 4], aos6[6 * p + 5]};
    float *acc[3] = {accU, accV, accW};
    float *wgt[3] = {wgtU, wgtV, wgtW};
#pragma unroll
    for (int dir = 0; dir < 3; dir++) {
        const int w = I + (dir == 0), h = J + (dir == 1), d = K + (dir == 2);
        const float px = P[0] - (dir == 0 ? 0.0f : hdx);
        const float py = P[1] - (dir == 1 ? 0.0f : hdx);
        const float pz = P[2] - (dir == 2 ? 0.0f : hdx);
        const float vel = Vv[dir];
        const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
        const int i0 = max(gi - 1, 0), j0 = max(gj - 1, 0), k0 = max(gk - 1, 0);
        const int i1 = min(gi + 1, w - 1), j1 = min(gj + 1, h - 1), k1 = min(gk + 1, d - 1);
        for (int k = k0; k <= k1; k++) {
            const float vz = (float)(k * dxd) - pz;
            for (int j = j0; j <= j1; j++) {
                const float vy = (float)(j * dxd) - py;
                for (int i = i0; i <= i1; i++) {
                    const float vx = (float)(i * dxd) - px;
                    const float q = vx * vx + vy * vy + vz * vz;
                    if (q < rsq) {
                        const float weight = 1.0f - coef1 * q * q * q + coef2 * q * q - coef3 * q;
                        const size_t f = gidx(L, i, j, k);
                        atomicAdd(&acc[dir][f], weight * vel);
                        atomicAdd(&wgt[dir][f], weight);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ K15: grid -> particle
// MACVelocityField::_interpolateLinearU/V/W (reference macvelocityfield.cpp:455-546): fp64 position,
// cell origin and weights; out-of-range corners contribute 0; corner order of interpolation.cpp:54-66.
__device__ __forceinline__ double d_mac_lerp(int dir, double x, double y, double z, double dx, const Lay &L,
                                             const float *__restrict__ g) {
    const int w = L.I + (dir == 0), h = L.J + (dir == 1), d = L.K + (dir == 2);
    if (dir != 0) x -= 0.5 * dx;
    if (dir != 1) y -= 0.5 * dx;
    if (dir != 2) z -= 0.5 * dx;
    const double invdx = 1.0 / dx;
    const int i = (int)floor(x * invdx), j = (int)floor(y * invdx), k = (int)floor(z * invdx);
    const double ix = (x - (double)i * dx) * invdx, iy = (y - (double)j * dx) * invdx, iz = (z - (double)k * dx) * invdx;
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
    if (d_in_range(i, j, k, w, h, d)) p0 = g[gidx(L, i, j, k)];
    if (d_in_range(i + 1, j, k, w, h, d)) p1 = g[gidx(L, i + 1, j, k)];
    if (d_in_range(i, j + 1, k, w, h, d)) p2 = g[gidx(L, i, j + 1, k)];
    if (d_in_range(i, j, k + 1, w, h, d)) p3 = g[gidx(L, i, j, k + 1)];
    if (d_in_range(i + 1, j, k + 1, w, h, d)) p4 = g[gidx(L, i + 1, j, k + 1)];
    if (d_in_range(i, j + 1, k + 1, w, h, d)) p5 = g[gidx(L, i, j + 1, k + 1)];
    if (d_in_range(i + 1, j + 1, k, w, h, d)) p6 = g[gidx(L, i + 1, j + 1, k)];
    if (d_in_range(i + 1, j + 1, k + 1, w, h, d)) p7 = g[gidx(L, i + 1, j + 1, k + 1)];
    return p0 * (1 - ix) * (1 - iy) * (1 - iz) + p1 * ix * (1 - iy) * (1 - iz) + p2 * (1 - ix) * iy * (1 - iz) +
           p3 * (1 - ix) * (1 - iy) * iz + p4 * ix * (1 - iy) * iz + p5 * (1 - ix) * iy * iz + p6 * ix * iy * (1 - iz) +
           p7 * ix * iy * iz;
}

// evaluateVelocityAtPositionLinear (reference macvelocityfield.cpp:564-578)
__device__ __forceinline__ void d_mac_velocity(float px, float py, float pz, double dx, const Lay &L,
                                               const float *__restrict__ U, const float *__restrict__ V,
                                               const float *__restrict__ W, float out[3]) {
    const double x = px, y = py, z = pz;
    if (!(x >= 0 && y >= 0 && z >= 0 && x < dx * L.I && y < dx * L.J && z < dx * L.K)) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    out[0] = (float)d_mac_lerp(0, x, y, z, dx, L, U);
    out[1] = (float)d_mac_lerp(1, x, y, z, dx, L, V);
    out[2] = (float)d_mac_lerp(2, x, y, z, dx, L, W);
}

// _updateFluidParticleVelocities (reference fluidsimulation.cpp:341-352)
__device__ __forceinline__ void d_update_velocity(float *q, double dx, const Lay &L, const float *U,
                                                  const float *V, const float *W, const float *sU, const float *sV,
                                                  const float *sW, float ratio) {
    float vn[3], vo[3];
    d_mac_velocity(q[0], q[1], q[2], dx, L, U, V, W, vn);
    d_mac_velocity(q[0], q[1], q[2], dx, L, sU, sV, sW, vo);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float pic = vn[c];
        const float flip = q[3 + c] + vn[c] - vo[c];
        q[3 + c] = ratio * pic + (1.0f - ratio) * flip;
    }
}

__global__ void k_update_velocities(Lay L, float *__restrict__ aos6, size_t n, const float *__restrict__ U,
                                    const float *__restrict__ V, const float *__restrict__ W,
                                    const float *__restrict__ sU, const float *__restrict__ sV,
                                    const float *__restrict__ sW, float dx, float ratio) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float q[6];
#pragma unroll
    for (int c = 0; c < 6; c++) q[c] = aos6[6 * p + c];
    d_update_velocity(q, (double)dx, L, U, V, W, sU, sV, sW, ratio);
#pragma unroll
    for (int c = 3; c < 6; c++) aos6[6 * p + c] = q[c];
}

// scalar-field trilinear value + gradient at a point of the solid node grid
// (reference interpolation.cpp:68-108 and :122-184): position differences in fp32, weights in fp64.
__device__ __forceinline__ float d_solid_value_grad(float px, float py, float pz, double dx, const Lay &L,
                                                    const float *__restrict__ g, float grad[3]) {
    const int w = L.I + 1, h = L.J + 1, d = L.K + 1;
    const double invdx = 1.0 / dx;
    const int gi = d_pos_index(px, invdx), gj = d_pos_index(py, invdx), gk = d_pos_index(pz, invdx);
    const float gx = (float)(gi * dx), gy = (float)(gj * dx), gz = (float)(gk * dx);
    const double ix = (px - gx) * invdx, iy = (py - gy) * invdx, iz = (pz - gz) * invdx;
    float v000 = 0, v100 = 0, v010 = 0, v001 = 0, v101 = 0, v011 = 0, v110 = 0, v111 = 0;
    if (d_in_range(gi, gj, gk, w, h, d)) v000 = g[gidx(L, gi, gj, gk)];
    if (d_in_range(gi + 1, gj, gk, w, h, d)) v100 = g[gidx(L, gi + 1, gj, gk)];
    if (d_in_range(gi, gj + 1, gk, w, h, d)) v010 = g[gidx(L, gi, gj + 1, gk)];
    if (d_in_range(gi, gj, gk + 1, w, h, d)) v001 = g[gidx(L, gi, gj, gk + 1)];
    if (d_in_range(gi + 1, gj, gk + 1, w, h, d)) v101 = g[gidx(L, gi + 1, gj, gk + 1)];
    if (d_in_range(gi, gj + 1, gk + 1, w, h, d)) v011 = g[gidx(L, gi, gj + 1, gk + 1)];
    if (d_in_range(gi + 1, gj + 1, gk, w, h, d)) v110 = g[gidx(L, gi + 1, gj + 1, gk)];
    if (d_in_range(gi + 1, gj + 1, gk + 1, w, h, d)) v111 = g[gidx(L, gi + 1, gj + 1, gk + 1)];
    const double val = (double)v000 * (1 - ix) * (1 - iy) * (1 - iz) + (double)v100 * ix * (1 - iy) * (1 - iz) +
                       (double)v010 * (1 - ix) * iy * (1 - iz) + (double)v001 * (1 - ix) * (1 - iy) * iz +
                       (double)v101 * ix * (1 - iy) * iz + (double)v011 * (1 - ix) * iy * iz +
                       (double)v110 * ix * iy * (1 - iz) + (double)v111 * ix * iy * iz;
    // gradient: differences in fp32, bilinear blend in fp64 (interpolation.cpp:161-178)
    {
        const float a00 = v100 - v000, a10 = v110 - v010, a01 = v101 - v001, a11 = v111 - v011;
        const double l1 = (1 - iy) * a00 + iy * a10, l2 = (1 - iy) * a01 + iy * a11;
        grad[0] = (float)((1 - iz) * l1 + iz * l2);
    }
    {
        const float a00 = v010 - v000, a10 = v110 - v100, a01 = v011 - v001, a11 = v111 - v101;
        const double l1 = (1 - ix) * a00 + ix * a10, l2 = (1 - ix) * a01 + ix * a11;
        grad[1] = (float)((1 - iz) * l1 + iz * l2);
    }
    {
        const float a00 = v001 - v000, a10 = v101 - v100, a01 = v011 - v010, a11 = v111 - v110;
        const double l1 = (1 - ix) * a00 + ix * a10, l2 = (1 - ix) * a01 + ix * a11;
        grad[2] = (float)((1 - iy) * l1 + iy * l2);
    }
    return (float)val;
}

struct ClampBox {  // AABB boundary(0,0,0,I*dx,J*dx,K*dx).expand(-2*dx-1e-4)  (fluidsimulation.cpp:319-320)
    float bx, by, bz;
    double bw, bh, bd;
};

// _advectFluidParticles (reference fluidsimulation.cpp:315-339)
__global__ void k_advect_particles(Lay L, float *__restrict__ aos6, size_t n, const float *__restrict__ U,
                                   const float *__restrict__ V, const float *__restrict__ W,
                                   const float *__restrict__ sU, const float *__restrict__ sV,
                                   const float *__restrict__ sW, const float *__restrict__ solid, float dxf, float dt,
                                   float ratio, ClampBox box) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double dx = (double)dxf;
    float q[6];
#pragma unroll
    for (int c = 0; c < 6; c++) q[c] = aos6[6 * p + c];
    d_update_velocity(q, dx, L, U, V, W, sU, sV, sW, ratio);
    // _traceRK2 (fluidsimulation.cpp:535-541)
    float v[3];
    d_mac_velocity(q[0], q[1], q[2], dx, L, U, V, W, v);
    const float hs = 0.5f * dt;
    d_mac_velocity(q[0] + hs * v[0], q[1] + hs * v[1], q[2] + hs * v[2], dx, L, U, V, W, v);
    float x = q[0] + dt * v[0], y = q[1] + dt * v[1], z = q[2] + dt * v[2];
    // solid push-out (fluidsimulation.cpp:326-333)
    float g[3];
    const float phi_val = d_solid_value_grad(x, y, z, dx, L, solid, g);
    if (phi_val < 0.0f) {
        const float lsq = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
        if (lsq > 0.0f) {
            const float inv = 1.0f / sqrtf(lsq);
            g[0] *= inv; g[1] *= inv; g[2] *= inv;
        }
        x -= phi_val * g[0]; y -= phi_val * g[1]; z -= phi_val * g[2];
    }
    // AABB clamp (aabb.cpp:126-129, 213-234)
    const bool inside = x >= box.bx && y >= box.by && z >= box.bz && (double)x < box.bx + box.bw &&
                        (double)y < box.by + box.bh && (double)z < box.bz + box.bd;
    if (!inside) {
        const float mx = box.bx + (float)box.bw, my = box.by + (float)box.bh, mz = box.bz + (float)box.bd;
        const float eps = (float)1e-6;
        x = fminf(fmaxf(x, box.bx), mx - eps);
        y = fminf(fmaxf(y, box.by), my - eps);
        z = fminf(fmaxf(z, box.bz), mz - eps);
    }
    q[0] = x; q[1] = y; q[2] = z;
#pragma unroll
    for (int c = 0; c < 6; c++) aos6[6 * p + c] = q[c];
}

// =================================================================== host launchers
int fv_sdf_finish(flipv_context *c);
int fv_p2g_finalize(flipv_context *c);

int fv_particle_sdf(flipv_context *c) {
    fv_fill_cells(c, c->phi, 3.0f * (float)(double)c->dx, 1);  // _getMaxDistance (particlelevelset.cpp:94-96)
    if (c->np) {
        // _particleRadius (fluidsimulation.cpp:36)
        const float radius = (float)(c->dx * 1.01 * sqrt(3.0) / 2.0);
        hipLaunchKernelGGL(k_sdf_scatter, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                           c->phi, c->dx, radius);
    }
    // contributions to the neighbours' boundary planes (a particle reaches one cell beyond its own)
    float *parr[1] = {c->phi};
    int rc = fv_halo_reduce(c, parr, 1, 1, 1, HALO_MIN_F32);
    if (rc) return rc;
    fv_sdf_finish(c);
    HIPCHK(c, hipGetLastError());
    const HaloArray ph[1] = {{c->phi, 4}};
    return fv_halo_copy(c, ph, 1, 4);  // P2G masks, volumes (trilinear +-1.5 cells, 2 dilation layers) read phi across the cut
}

int fv_p2g(flipv_context *c) {
    const Lay R = fv_range(c, 2);  // planes this rank's particles can reach
    const size_t off = (size_t)R.kb * c->L.sz, bytes = (size_t)(R.ke - R.kb) * c->L.sz * 4;
    float *acc[6] = {c->accU, c->accV, c->accW, c->wgtU, c->wgtV, c->wgtW};
    for (int q = 0; q < 6; q++) HIPCHK(c, hipMemsetAsync(acc[q] + off, 0, bytes, c->stream));
    if (c->np)
        hipLaunchKernelGGL(k_p2g_scatter, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                           c->accU, c->wgtU, c->accV, c->wgtV, c->accW, c->wgtW, c->dx);
    int rc = fv_halo_reduce(c, acc, 6, 2, 2, HALO_ADD_F32);
    if (rc) return rc;
    fv_p2g_finalize(c);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_update_particle_velocities(flipv_context *c) {
    if (c->np)
        hipLaunchKernelGGL(k_update_velocities, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                           c->U, c->V, c->W, c->sU, c->sV, c->sW, c->dx, c->prm.pic_ratio);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_advect_particles(flipv_context *c, float dt) {
    const Lay &d = c->L;
    ClampBox b;
    const float dxf = c->dx;
    double bw = (double)(d.I * dxf), bh = (double)(d.J * dxf), bd = (double)(d.K * dxf);
    const double ev = -2 * dxf - 1e-4, eh = 0.5 * ev;  // AABB::expand (aabb.cpp:118-124)
    b.bx = b.by = b.bz = 0.0f - (float)eh;
    b.bw = bw + ev; b.bh = bh + ev; b.bd = bd + ev;
    // a particle samples the fields up to CFL (5) cells + the trilinear support away from its cell
    const HaloArray vel[6] = {{c->U, 4}, {c->V, 4}, {c->W, 4}, {c->sU, 4}, {c->sV, 4}, {c->sW, 4}};
    int rc = fv_halo_copy(c, vel, 6, (int)ceilf(c->prm.cfl_number) + 3);
    if (rc) return rc;
    if (c->np)
        hipLaunchKernelGGL(k_advect_particles, dim3(cdiv(c->np, 256)), dim3(256), 0, c->stream, c->L, c->particles, c->np,
                           c->U, c->V, c->W, c->sU, c->sV, c->sW, c->solid, c->dx, dt, c->prm.pic_ratio, b);
    HIPCHK(c, hipGetLastError());
    return fv_migrate_particles(c);
}
