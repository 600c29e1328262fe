// flipv_api.hip -- C-ABI entry points of libflipv.so (include/flipv.h): context lifetime, host<->device
// transfers in the reference's Array3d layout, and the substep sequencing of
// FluidSimulation::advance (reference fluidsimulation.cpp:135-168).
#include "flipv_internal.h"

#include <new>

static thread_local std::string g_create_error;

// ------------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(flipv_context *c, T **p, size_t n, bool zero = true) {
    void *q = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? FLIPV_ERR_OOM : FLIPV_ERR_HIP;
    }
    c->allocs.push_back(q);
    if (zero) {
        e = hipMemsetAsync(q, 0, bytes, c->stream);
        if (e != hipSuccess) { c->err = std::string("hipMemset: ") + hipGetErrorString(e); return FLIPV_ERR_HIP; }
    }
    *p = (T *)q;
    return FLIPV_OK;
}

extern "C" int flipv_default_params(flipv_params *p) {
    if (!p) return FLIPV_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    p->cfl_number = 5.0f;
    p->min_frac = 0.01f;
    p->pic_ratio = 0.05f;
    p->extrapolation_layers = 0;
    p->pressure_tolerance = 1e-9;
    p->pressure_rel_tolerance = 1e-6;
    p->pressure_max_iterations = 2000;
    p->viscosity_tolerance = 1e-6;
    p->viscosity_max_iterations = 700;
    p->viscosity_accept_tolerance = 10.0;
    p->precision = FLIPV_PRECISION_FP32;
    p->kernel_timing = 0;
    p->check_every = 8;
    return FLIPV_OK;
}

extern "C" const char *flipv_last_error(flipv_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int flipv_create_on_device(int I, int J, int K, float dx, int dev, flipv_context **out) {
    if (!out) return FLIPV_ERR_INVALID;
    *out = nullptr;
    if (I < 1 || J < 1 || K < 1 || !(dx > 0.0f)) {
        g_create_error = "flipv_create: grid dimensions must be >= 1 and dx > 0";
        return FLIPV_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "flipv_create: no HIP device visible (libflipv has no CPU path)";
        return FLIPV_ERR_NO_DEVICE;
    }
    if (dev < 0 || dev >= ndev) {
        g_create_error = "flipv_create: invalid device ordinal";
        return FLIPV_ERR_INVALID;
    }
    flipv_context *c = new (std::nothrow) flipv_context();
    if (!c) return FLIPV_ERR_OOM;
    c->d.I = I; c->d.J = J; c->d.K = K;
    c->dx = dx;
    c->device = dev;
    c->np = c->pcap = 0;
    c->particles = nullptr;
    c->d_scal = nullptr; c->h_scal = nullptr; c->scalCap = 0;
    c->evUsed = 0;
    c->pressureReady = c->viscosityReady = 0;
    c->nActiveP = c->nActiveV = 0;
    c->viscosity_nonzero = 1;
    memset(&c->kstats, 0, sizeof(c->kstats));
    flipv_default_params(&c->prm);
    c->gravity[0] = 0.0f; c->gravity[1] = -9.81f; c->gravity[2] = 0.0f;  // fluidsimulation.cpp:40
#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(e_); flipv_destroy(c); return FLIPV_ERR_HIP; } } while (0)
#define ALLOC(ptr, n) do { int rc_ = dev_alloc(c, &(ptr), (n)); if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; } } while (0)
    c->stream = nullptr;
    CHK(hipSetDevice(dev));
    CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int q = 0; q <= FLIPV_PHASE_COUNT; q++) c->phaseEv[q] = nullptr;
    for (int q = 0; q <= FLIPV_PHASE_COUNT; q++) CHK(hipEventCreate(&c->phaseEv[q]));
    const Dims &d = c->d;
    const size_t nu = d.nu(), nv = d.nv(), nw = d.nw(), nc = d.nc(), nn = d.nn();
    ALLOC(c->U, nu); ALLOC(c->V, nv); ALLOC(c->W, nw);
    ALLOC(c->sU, nu); ALLOC(c->sV, nv); ALLOC(c->sW, nw);
    ALLOC(c->wU, nu); ALLOC(c->wV, nv); ALLOC(c->wW, nw);
    ALLOC(c->vU, nu); ALLOC(c->vV, nv); ALLOC(c->vW, nw);
    ALLOC(c->phi, nc); ALLOC(c->pressure, nc); ALLOC(c->solid, nn); ALLOC(c->visc, nn);
    ALLOC(c->accU, nu); ALLOC(c->accV, nv); ALLOC(c->accW, nw);
    ALLOC(c->wgtU, nu); ALLOC(c->wgtV, nv); ALLOC(c->wgtW, nw);
    ALLOC(c->stampU, nu); ALLOC(c->stampV, nv); ALLOC(c->stampW, nw);
    ALLOC(c->d_flags, 16);
    CHK(hipHostMalloc((void **)&c->h_flags, 16 * sizeof(int)));
    memset(c->h_flags, 0, 16 * sizeof(int));
    // solver tiles over the (I+1,J+1,K+1) index space
    c->tg.ntx = (I + 1 + TX - 1) / TX; c->tg.nty = (J + 1 + TY - 1) / TY; c->tg.ntz = (K + 1 + TZ - 1) / TZ;
    ALLOC(c->tileListP, (size_t)c->tg.count() + 8);
    ALLOC(c->tileListV, (size_t)c->tg.count() + 8);
    ALLOC(c->tileFlag, (size_t)c->tg.count() + 8);
    // pressure system
    ALLOC(c->pDiag, nc); ALLOC(c->pPi, nc); ALLOC(c->pPj, nc); ALLOC(c->pPk, nc);
    double *tmp;
    ALLOC(tmp, nc); c->pX = tmp; ALLOC(tmp, nc); c->pR = tmp; ALLOC(tmp, nc); c->pZ = tmp; ALLOC(tmp, nc); c->pS = tmp;
    // viscosity system
    ALLOC(c->scp, nc);
    ALLOC(c->volC, nc); ALLOC(c->volU, nu); ALLOC(c->volV, nv); ALLOC(c->volW, nw);
    const size_t neu = (size_t)I * (J + 1) * (K + 1), nev = (size_t)(I + 1) * J * (K + 1), new_ = (size_t)(I + 1) * (J + 1) * K;
    ALLOC(c->volEU, neu); ALLOC(c->volEV, nev); ALLOC(c->volEW, new_);
    ALLOC(c->fC, nc); ALLOC(c->fEU, neu); ALLOC(c->fEV, nev); ALLOC(c->fEW, new_);
    ALLOC(c->vDiagU, nu); ALLOC(c->vDiagV, nv); ALLOC(c->vDiagW, nw);
    ALLOC(c->stU, nu); ALLOC(c->stV, nv); ALLOC(c->stW, nw);
    ALLOC(c->validCells, nn); ALLOC(c->validTmp, nn);
    const size_t nf[3] = {nu, nv, nw};
    for (int q = 0; q < 3; q++) {
        ALLOC(tmp, nf[q]); c->vX[q] = tmp; ALLOC(tmp, nf[q]); c->vR[q] = tmp;
        ALLOC(tmp, nf[q]); c->vZ[q] = tmp; ALLOC(tmp, nf[q]); c->vS[q] = tmp;
    }
    // defaults of initialize(): viscosity 1.0 at every node (fluidsimulation.cpp:39), liquid phi = 3 dx
    {
        std::vector<float> ones(nn, 1.0f);
        CHK(hipMemcpyAsync(c->visc, ones.data(), nn * 4, hipMemcpyHostToDevice, c->stream));
        std::vector<float> far(nc, 3.0f * dx);
        CHK(hipMemcpyAsync(c->phi, far.data(), nc * 4, hipMemcpyHostToDevice, c->stream));
        CHK(hipStreamSynchronize(c->stream));
    }
    CHK(hipStreamSynchronize(c->stream));
#undef CHK
#undef ALLOC
    *out = c;
    return FLIPV_OK;
}

extern "C" int flipv_create(int I, int J, int K, float dx, flipv_context **out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return flipv_create_on_device(I, J, K, dx, dev, out);
}

extern "C" int flipv_destroy(flipv_context *c) {
    if (!c) return FLIPV_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->particles) (void)hipFree(c->particles);
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    for (hipEvent_t e : c->evPool) (void)hipEventDestroy(e);
    for (int q = 0; q <= FLIPV_PHASE_COUNT; q++) if (c->phaseEv[q]) (void)hipEventDestroy(c->phaseEv[q]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FLIPV_OK;
}

extern "C" int flipv_device_name(flipv_context *c, char *buf, size_t len) {
    if (!c || !buf || !len) return FLIPV_ERR_INVALID;
    hipDeviceProp_t p;
    HIPCHK(c, hipGetDeviceProperties(&p, c->device));
    snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return FLIPV_OK;
}

extern "C" int flipv_set_params(flipv_context *c, const flipv_params *p) {
    if (!c || !p) return FLIPV_ERR_INVALID;
    if (p->pressure_max_iterations < 1 || p->viscosity_max_iterations < 1 || !(p->cfl_number > 0) ||
        (p->precision != FLIPV_PRECISION_FP32 && p->precision != FLIPV_PRECISION_FP64)) {
        c->err = "flipv_set_params: invalid parameter";
        return FLIPV_ERR_INVALID;
    }
    c->prm = *p;
    return FLIPV_OK;
}
extern "C" int flipv_get_params(flipv_context *c, flipv_params *p) {
    if (!c || !p) return FLIPV_ERR_INVALID;
    *p = c->prm;
    return FLIPV_OK;
}
extern "C" int flipv_set_gravity(flipv_context *c, float gx, float gy, float gz) {
    if (!c) return FLIPV_ERR_INVALID;
    c->gravity[0] = gx; c->gravity[1] = gy; c->gravity[2] = gz;
    return FLIPV_OK;
}

// ------------------------------------------------------------------------------------------------ grids
struct GridRef { float *f; uint8_t *m; size_t n; };
static bool grid_ref(flipv_context *c, int which, GridRef *g) {
    const Dims &d = c->d;
    g->f = nullptr; g->m = nullptr;
    switch (which) {
        case FLIPV_GRID_U: g->f = c->U; g->n = d.nu(); return true;
        case FLIPV_GRID_V: g->f = c->V; g->n = d.nv(); return true;
        case FLIPV_GRID_W: g->f = c->W; g->n = d.nw(); return true;
        case FLIPV_GRID_SAVED_U: g->f = c->sU; g->n = d.nu(); return true;
        case FLIPV_GRID_SAVED_V: g->f = c->sV; g->n = d.nv(); return true;
        case FLIPV_GRID_SAVED_W: g->f = c->sW; g->n = d.nw(); return true;
        case FLIPV_GRID_VALID_U: g->m = c->vU; g->n = d.nu(); return true;
        case FLIPV_GRID_VALID_V: g->m = c->vV; g->n = d.nv(); return true;
        case FLIPV_GRID_VALID_W: g->m = c->vW; g->n = d.nw(); return true;
        case FLIPV_GRID_LIQUID_PHI: g->f = c->phi; g->n = d.nc(); return true;
        case FLIPV_GRID_SOLID_PHI: g->f = c->solid; g->n = d.nn(); return true;
        case FLIPV_GRID_WEIGHT_U: g->f = c->wU; g->n = d.nu(); return true;
        case FLIPV_GRID_WEIGHT_V: g->f = c->wV; g->n = d.nv(); return true;
        case FLIPV_GRID_WEIGHT_W: g->f = c->wW; g->n = d.nw(); return true;
        case FLIPV_GRID_VISCOSITY: g->f = c->visc; g->n = d.nn(); return true;
        case FLIPV_GRID_PRESSURE: g->f = c->pressure; g->n = d.nc(); return true;
    }
    return false;
}

extern "C" size_t flipv_grid_elements(flipv_context *c, int which) {
    GridRef g;
    if (!c || !grid_ref(c, which, &g)) return 0;
    return g.n;
}

extern "C" int flipv_read_grid(flipv_context *c, int which, float *out) {
    GridRef g;
    if (!c || !out || !grid_ref(c, which, &g)) { if (c) c->err = "flipv_read_grid: bad grid id"; return FLIPV_ERR_INVALID; }
    if (g.f) {
        HIPCHK(c, hipMemcpyAsync(out, g.f, g.n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        std::vector<uint8_t> tmp(g.n);
        HIPCHK(c, hipMemcpyAsync(tmp.data(), g.m, g.n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t t = 0; t < g.n; t++) out[t] = tmp[t] ? 1.0f : 0.0f;
    }
    return FLIPV_OK;
}

extern "C" int flipv_write_grid(flipv_context *c, int which, const float *in) {
    GridRef g;
    if (!c || !in || !grid_ref(c, which, &g)) { if (c) c->err = "flipv_write_grid: bad grid id"; return FLIPV_ERR_INVALID; }
    if (g.f) {
        HIPCHK(c, hipMemcpyAsync(g.f, in, g.n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (which == FLIPV_GRID_VISCOSITY) {
            int nz = 0;
            for (size_t t = 0; t < g.n; t++) if (in[t] > 0.0f) { nz = 1; break; }
            c->viscosity_nonzero = nz;
        }
    } else {
        std::vector<uint8_t> tmp(g.n);
        for (size_t t = 0; t < g.n; t++) tmp[t] = in[t] != 0.0f;
        HIPCHK(c, hipMemcpyAsync(g.m, tmp.data(), g.n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FLIPV_OK;
}

extern "C" int flipv_set_solid_sdf(flipv_context *c, const float *nodes) { return flipv_write_grid(c, FLIPV_GRID_SOLID_PHI, nodes); }
extern "C" int flipv_set_viscosity(flipv_context *c, const float *nodes) {
    if (!c || !nodes) return FLIPV_ERR_INVALID;
    const size_t n = c->d.nn();
    for (size_t t = 0; t < n; t++)
        if (!(nodes[t] >= 0.0f)) { c->err = "flipv_set_viscosity: negative viscosity"; return FLIPV_ERR_INVALID; }  // fluidsimulation.cpp:119
    return flipv_write_grid(c, FLIPV_GRID_VISCOSITY, nodes);
}
extern "C" int flipv_set_viscosity_uniform(flipv_context *c, float value) {
    if (!c) return FLIPV_ERR_INVALID;
    if (!(value >= 0.0f)) { c->err = "flipv_set_viscosity_uniform: negative viscosity"; return FLIPV_ERR_INVALID; }  // fluidsimulation.cpp:100
    std::vector<float> v(c->d.nn(), value);
    return flipv_write_grid(c, FLIPV_GRID_VISCOSITY, v.data());
}

// ------------------------------------------------------------------------------------------------ particles
extern "C" int flipv_upload_particles(flipv_context *c, const float *aos6, size_t n) {
    if (!c || (n && !aos6)) return FLIPV_ERR_INVALID;
    if (n > c->pcap) {
        if (c->particles) (void)hipFree(c->particles);
        c->particles = nullptr;
        c->pcap = 0;
        const size_t cap = n + n / 8 + 1024;
        hipError_t e = hipMalloc((void **)&c->particles, cap * 6 * sizeof(float));
        if (e != hipSuccess) { c->err = std::string("hipMalloc(particles): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        c->pcap = cap;
    }
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->particles, aos6, n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->np = n;
    return FLIPV_OK;
}
extern "C" int flipv_download_particles(flipv_context *c, float *aos6, size_t capacity, size_t *n_out) {
    if (!c) return FLIPV_ERR_INVALID;
    if (n_out) *n_out = c->np;
    if (capacity < c->np) { c->err = "flipv_download_particles: buffer too small"; return FLIPV_ERR_INVALID; }
    if (c->np) {
        if (!aos6) return FLIPV_ERR_INVALID;
        HIPCHK(c, hipMemcpyAsync(aos6, c->particles, c->np * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FLIPV_OK;
}
extern "C" size_t flipv_num_particles(flipv_context *c) { return c ? c->np : 0; }

// ------------------------------------------------------------------------------------------------ operators
#define ENTER(c) do { if (!(c)) return FLIPV_ERR_INVALID; hipError_t e_ = hipSetDevice((c)->device); if (e_ != hipSuccess) { (c)->err = hipGetErrorString(e_); return FLIPV_ERR_HIP; } } while (0)
#define SYNC_RET(c, rc) do { int rc__ = (rc); hipError_t e_ = hipStreamSynchronize((c)->stream); if (e_ != hipSuccess) { (c)->err = std::string("stream sync: ") + hipGetErrorString(e_); return FLIPV_ERR_HIP; } return rc__; } while (0)

extern "C" int flipv_cfl(flipv_context *c, float *dt_out) { ENTER(c); if (!dt_out) return FLIPV_ERR_INVALID; return fv_cfl(c, dt_out); }
extern "C" int flipv_particle_sdf(flipv_context *c) { ENTER(c); SYNC_RET(c, fv_particle_sdf(c)); }
extern "C" int flipv_p2g(flipv_context *c) { ENTER(c); SYNC_RET(c, fv_p2g(c)); }
extern "C" int flipv_extrapolate(flipv_context *c) { ENTER(c); SYNC_RET(c, fv_extrapolate(c)); }

static int save_velocity(flipv_context *c) {
    const Dims &d = c->d;
    HIPCHK(c, hipMemcpyAsync(c->sU, c->U, d.nu() * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sV, c->V, d.nv() * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sW, c->W, d.nw() * 4, hipMemcpyDeviceToDevice, c->stream));
    return FLIPV_OK;
}
extern "C" int flipv_save_velocity(flipv_context *c) { ENTER(c); SYNC_RET(c, save_velocity(c)); }

static int advect_velocity_field(flipv_context *c) {  // fluidsimulation.cpp:500-519
    int rc = fv_p2g(c);
    if (rc) return rc;
    rc = fv_extrapolate(c);
    if (rc) return rc;
    return save_velocity(c);
}
extern "C" int flipv_advect_velocity_field(flipv_context *c) { ENTER(c); SYNC_RET(c, advect_velocity_field(c)); }
extern "C" int flipv_body_force(flipv_context *c, float dt) { ENTER(c); SYNC_RET(c, fv_body_force(c, dt)); }
extern "C" int flipv_viscosity_solve(flipv_context *c, float dt, flipv_solve_info *info) { ENTER(c); SYNC_RET(c, fv_viscosity_solve(c, dt, info)); }
extern "C" int flipv_compute_weights(flipv_context *c) { ENTER(c); SYNC_RET(c, fv_compute_weights(c)); }
extern "C" int flipv_pressure_solve(flipv_context *c, float dt, flipv_solve_info *info) { ENTER(c); SYNC_RET(c, fv_pressure_solve(c, dt, info)); }
extern "C" int flipv_apply_pressure(flipv_context *c, float dt) { ENTER(c); SYNC_RET(c, fv_apply_pressure(c, dt)); }
extern "C" int flipv_constrain(flipv_context *c) { ENTER(c); SYNC_RET(c, fv_constrain(c)); }
extern "C" int flipv_update_particle_velocities(flipv_context *c) { ENTER(c); SYNC_RET(c, fv_update_particle_velocities(c)); }
extern "C" int flipv_advect_particles(flipv_context *c, float dt) { ENTER(c); SYNC_RET(c, fv_advect_particles(c, dt)); }

extern "C" int flipv_read_viscosity_volume(flipv_context *c, int which, float *out) {
    ENTER(c);
    if (!out || which < 0 || which > 6) return FLIPV_ERR_INVALID;
    const int I = c->d.I, J = c->d.J, K = c->d.K;
    const float *src[7] = {c->volC, c->volU, c->volV, c->volW, c->volEU, c->volEV, c->volEW};
    const size_t n[7] = {c->d.nc(), c->d.nu(), c->d.nv(), c->d.nw(), (size_t)I * (J + 1) * (K + 1),
                         (size_t)(I + 1) * J * (K + 1), (size_t)(I + 1) * (J + 1) * K};
    HIPCHK(c, hipMemcpyAsync(out, src[which], n[which] * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FLIPV_OK;
}

// ------------------------------------------------------------------------------------------------ substep
static int substep(flipv_context *c, float dt, flipv_stats *st) {
    // phase order of the while-loop body, fluidsimulation.cpp:145-164
    int rc, warn = FLIPV_OK;
    flipv_solve_info vi, pi;
    memset(&vi, 0, sizeof(vi));
    memset(&pi, 0, sizeof(pi));
#define MARK(q) HIPCHK(c, hipEventRecord(c->phaseEv[q], c->stream))
    MARK(0);
    if ((rc = fv_particle_sdf(c)) < 0) return rc;
    MARK(1);
    if ((rc = advect_velocity_field(c)) < 0) return rc;
    MARK(2);
    if ((rc = fv_body_force(c, dt)) < 0) return rc;
    MARK(3);
    if ((rc = fv_viscosity_solve(c, dt, &vi)) < 0) return rc;
    if (rc > warn) warn = rc;
    MARK(4);
    if ((rc = fv_compute_weights(c)) < 0) return rc;
    if ((rc = fv_pressure_solve(c, dt, &pi)) < 0) return rc;
    if (rc > warn) warn = rc;
    if ((rc = fv_apply_pressure(c, dt)) < 0) return rc;
    if ((rc = fv_extrapolate(c)) < 0) return rc;
    MARK(5);
    if ((rc = fv_constrain(c)) < 0) return rc;
    MARK(6);
    if ((rc = fv_advect_particles(c, dt)) < 0) return rc;
    MARK(7);
#undef MARK
    HIPCHK(c, hipEventSynchronize(c->phaseEv[7]));
    if (st) {
        double tot = 0;
        for (int q = 0; q < FLIPV_PHASE_COUNT; q++) {
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->phaseEv[q], c->phaseEv[q + 1]));
            st->phase_ms[q] = ms;
            tot += ms;
        }
        st->total_ms = tot;
        st->dt = dt;
        st->substeps = 1;
        st->viscosity = vi;
        st->pressure = pi;
    }
    return warn;
}

extern "C" int flipv_substep(flipv_context *c, float dt, flipv_stats *st) {
    ENTER(c);
    if (!(dt > 0.0f)) { c->err = "flipv_substep: dt must be > 0"; return FLIPV_ERR_INVALID; }
    return substep(c, dt, st);
}

extern "C" int flipv_advance(flipv_context *c, float dt, flipv_stats *st) {
    ENTER(c);
    if (!(dt > 0.0f)) { c->err = "flipv_advance: dt must be > 0"; return FLIPV_ERR_INVALID; }
    // fluidsimulation.cpp:135-168
    float t = 0;
    int n = 0, warn = FLIPV_OK;
    flipv_stats acc, one;
    memset(&acc, 0, sizeof(acc));
    while (t < dt) {
        float sub;
        int rc = fv_cfl(c, &sub);
        if (rc < 0) return rc;
        if (t + sub > dt) sub = dt - t;
        rc = substep(c, sub, &one);
        if (rc < 0) return rc;
        if (rc > warn) warn = rc;
        for (int q = 0; q < FLIPV_PHASE_COUNT; q++) acc.phase_ms[q] += one.phase_ms[q];
        acc.total_ms += one.total_ms;
        acc.viscosity = one.viscosity;
        acc.pressure = one.pressure;
        acc.dt = sub;
        t += sub;
        n++;
    }
    acc.substeps = n;
    if (st) *st = acc;
    return warn;
}

// ------------------------------------------------------------------------------------------------ timing
void fv_ev_begin(flipv_context *c, int which, double cells) {
    if (c->evUsed + 2 > c->evPool.size()) {
        const size_t grow = c->evPool.size() + 512;
        while (c->evPool.size() < grow) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            c->evPool.push_back(e);
        }
    }
    flipv_context::EvSpan s;
    s.a = c->evUsed++; s.b = c->evUsed++; s.which = which; s.cells = cells;
    (void)hipEventRecord(c->evPool[s.a], c->stream);
    c->evSpans.push_back(s);
}
void fv_ev_end(flipv_context *c) {
    if (c->evSpans.empty()) return;
    (void)hipEventRecord(c->evPool[c->evSpans.back().b], c->stream);
}
void fv_ev_collect(flipv_context *c) {
    if (c->evSpans.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (const auto &s : c->evSpans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->evPool[s.a], c->evPool[s.b]) != hipSuccess) continue;
        if (s.which == 0) { c->kstats.pressure_spmv_ms += ms; c->kstats.pressure_spmv_launches++; c->kstats.pressure_spmv_cells += s.cells; }
        else { c->kstats.viscosity_spmv_ms += ms; c->kstats.viscosity_spmv_launches++; c->kstats.viscosity_spmv_cells += s.cells; }
    }
    c->evSpans.clear();
    c->evUsed = 0;
}

extern "C" int flipv_kernel_stats_reset(flipv_context *c) {
    if (!c) return FLIPV_ERR_INVALID;
    memset(&c->kstats, 0, sizeof(c->kstats));
    return FLIPV_OK;
}
extern "C" int flipv_kernel_stats_get(flipv_context *c, flipv_kernel_stats *out) {
    if (!c || !out) return FLIPV_ERR_INVALID;
    fv_ev_collect(c);
    *out = c->kstats;
    return FLIPV_OK;
}
extern "C" int flipv_synchronize(flipv_context *c) {
    ENTER(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FLIPV_OK;
}

int fv_bench_pressure_spmv(flipv_context *c, int reps, double *ms, double *cells);
int fv_bench_viscosity_spmv(flipv_context *c, int reps, double *ms, double *cells);
extern "C" int flipv_bench_spmv(flipv_context *c, int which, int reps, double *ms_out, double *cells_out) {
    ENTER(c);
    if (!ms_out || !cells_out || reps < 1) return FLIPV_ERR_INVALID;
    return which == 0 ? fv_bench_pressure_spmv(c, reps, ms_out, cells_out) : fv_bench_viscosity_spmv(c, reps, ms_out, cells_out);
}

extern "C" int flipv_bench_copy(flipv_context *c, size_t bytes, int reps, double *gbps_out) {
    ENTER(c);
    if (!gbps_out || reps < 1 || bytes < 1024) return FLIPV_ERR_INVALID;
    void *a = nullptr, *b = nullptr;
    HIPCHK(c, hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); c->err = "flipv_bench_copy: out of memory"; return FLIPV_ERR_OOM; }
    (void)hipMemsetAsync(a, 1, bytes, c->stream);
    (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps; r++) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    *gbps_out = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    return FLIPV_OK;
}
