/*
 * oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A thin C-ABI harness around the UNMODIFIED reference sources, compiled where they lie
 * under /root/reference/src by oracle/Makefile into oracle/_ref/libflipref.so.  It exposes
 * each phase of FluidSimulation::advance() (fluidsimulation.cpp:135-168) separately so the
 * CPU restatement (oracle/flip_oracle.c) and the HIP path can be compared phase by phase,
 * and it is the "reference" kind CPU baseline of bench.py.
 *
 * Nothing from the reference is copied here: the harness only #includes its headers (with
 * private members made reachable) and calls its functions.
 */
#include <vector>
#include <string>
#include <sstream>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <limits>
#include <queue>
#include <cmath>
#include <stdexcept>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <unistd.h>
#include <fcntl.h>

#define private public
#define protected public
#include "fluidsimulation.h"
#undef private
#undef protected

namespace {

struct StdoutCapture {
    // The reference reports solver iteration counts only on stdout
    // (pressuresolver.cpp:550-551, viscositysolver.cpp:677-687); capture and parse them.
    int saved = -1;
    char path[64];
    void begin() {
        fflush(stdout);
        std::cout.flush();
        strcpy(path, "/tmp/flipref_XXXXXX");
        int fd = mkstemp(path);
        saved = dup(1);
        dup2(fd, 1);
        close(fd);
    }
    std::string end() {
        fflush(stdout);
        std::cout.flush();
        dup2(saved, 1);
        close(saved);
        std::ifstream f(path);
        std::stringstream ss;
        ss << f.rdbuf();
        unlink(path);
        return ss.str();
    }
};

bool parse_after(const std::string &s, const char *key, double *out) {
    size_t p = s.rfind(key);
    if (p == std::string::npos) return false;
    *out = atof(s.c_str() + p + strlen(key));
    return true;
}

struct Ref {
    FluidSimulation sim;
    Array3d<float> pressure;
    int I, J, K;
    float dx;
    // last-solve statistics
    int visc_iters = -1, pres_iters = -1;
    double visc_err = -1, pres_err = -1;
    int visc_failed = 0, pres_failed = 0;
    // viscosity-solver overrides (<=0 keeps the stock 700 / 1e-6)
    int visc_maxiter = 0;
    double visc_tol = 0;
};

TriangleMesh make_mesh(const float *verts, int nv, const int *tris, int nt) {
    TriangleMesh m;
    m.vertices.resize(nv);
    for (int i = 0; i < nv; i++)
        m.vertices[i] = vmath::vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    m.triangles.resize(nt);
    for (int i = 0; i < nt; i++)
        m.triangles[i] = Triangle(tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]);
    return m;
}

template <class T>
void copy_out(Array3d<T> &a, float *out) {
    T *raw = a.getRawArray();
    size_t n = (size_t)a.width * a.height * a.depth;
    for (size_t i = 0; i < n; i++) out[i] = (float)raw[i];
}
template <class T>
void copy_in(Array3d<T> &a, const float *in) {
    T *raw = a.getRawArray();
    size_t n = (size_t)a.width * a.height * a.depth;
    for (size_t i = 0; i < n; i++) raw[i] = (T)in[i];
}

}  // namespace

extern "C" {

void *ref_create(int I, int J, int K, float dx) {
    Ref *r = new Ref();
    r->I = I; r->J = J; r->K = K; r->dx = dx;
    StdoutCapture cap; cap.begin();
    r->sim.initialize(I, J, K, dx);
    cap.end();
    r->pressure = Array3d<float>(I, J, K, 0.0f);
    return r;
}

void ref_destroy(void *h) { delete (Ref *)h; }

void ref_srand(unsigned seed) { srand(seed); }

void ref_add_boundary(void *h, const float *verts, int nv, const int *tris, int nt, int inverted) {
    Ref *r = (Ref *)h;
    TriangleMesh m = make_mesh(verts, nv, tris, nt);
    r->sim.addBoundary(m, inverted != 0);
}

void ref_reset_boundary(void *h) { ((Ref *)h)->sim.resetBoundary(); }

void ref_add_liquid(void *h, const float *verts, int nv, const int *tris, int nt) {
    Ref *r = (Ref *)h;
    TriangleMesh m = make_mesh(verts, nv, tris, nt);
    r->sim.addLiquid(m);
}

/* MeshLevelSet::calculateSignedDistanceField on its own (meshlevelset.cpp:138): out = (I+1)(J+1)(K+1) */
void ref_mesh_sdf(int I, int J, int K, float dx, const float *verts, int nv, const int *tris, int nt,
                  int band, float *out_phi, int *out_closest) {
    MeshLevelSet ls(I, J, K, dx);
    TriangleMesh m = make_mesh(verts, nv, tris, nt);
    ls.calculateSignedDistanceField(m, band);
    copy_out(ls._phi, out_phi);
    if (out_closest) {
        int *raw = ls._closestTriangles.getRawArray();
        size_t n = (size_t)(I + 1) * (J + 1) * (K + 1);
        memcpy(out_closest, raw, n * sizeof(int));
    }
}

void ref_set_viscosity(void *h, float v) { ((Ref *)h)->sim.setViscosity(v); }
void ref_set_viscosity_grid(void *h, const float *nodes) {
    Ref *r = (Ref *)h;
    Array3d<float> g(r->I + 1, r->J + 1, r->K + 1, 0.0f);
    copy_in(g, nodes);
    r->sim.setViscosity(g);
}
void ref_set_gravity(void *h, float gx, float gy, float gz) { ((Ref *)h)->sim.setGravity(gx, gy, gz); }
void ref_set_viscosity_solver(void *h, int maxiter, double tol) {
    ((Ref *)h)->visc_maxiter = maxiter;
    ((Ref *)h)->visc_tol = tol;
}

size_t ref_num_particles(void *h) { return ((Ref *)h)->sim.particles.size(); }
void ref_get_particles(void *h, float *aos6) {
    Ref *r = (Ref *)h;
    for (size_t i = 0; i < r->sim.particles.size(); i++) {
        const FluidParticle &p = r->sim.particles[i];
        aos6[6 * i + 0] = p.position.x; aos6[6 * i + 1] = p.position.y; aos6[6 * i + 2] = p.position.z;
        aos6[6 * i + 3] = p.velocity.x; aos6[6 * i + 4] = p.velocity.y; aos6[6 * i + 5] = p.velocity.z;
    }
}
void ref_set_particles(void *h, const float *aos6, size_t n) {
    Ref *r = (Ref *)h;
    r->sim.particles.resize(n);
    for (size_t i = 0; i < n; i++) {
        r->sim.particles[i].position = vmath::vec3(aos6[6 * i], aos6[6 * i + 1], aos6[6 * i + 2]);
        r->sim.particles[i].velocity = vmath::vec3(aos6[6 * i + 3], aos6[6 * i + 4], aos6[6 * i + 5]);
    }
}

/* grid ids are shared with include/flipv.h (flipv_grid) */
static int grid_access(Ref *r, int which, float *out, const float *in) {
    FluidSimulation &s = r->sim;
#define RW(arr) do { if (out) copy_out(arr, out); else copy_in(arr, in); return 0; } while (0)
    switch (which) {
        case 0: RW(s._MACVelocity._u);
        case 1: RW(s._MACVelocity._v);
        case 2: RW(s._MACVelocity._w);
        case 3: RW(s._savedVelocityField._u);
        case 4: RW(s._savedVelocityField._v);
        case 5: RW(s._savedVelocityField._w);
        case 6: RW(s._validVelocities.validU);
        case 7: RW(s._validVelocities.validV);
        case 8: RW(s._validVelocities.validW);
        case 9: RW(s._liquidSDF._phi);
        case 10: RW(s._solidSDF._phi);
        case 11: RW(s._weightGrid.U);
        case 12: RW(s._weightGrid.V);
        case 13: RW(s._weightGrid.W);
        case 14: RW(s._viscosity);
        case 15: RW(r->pressure);
    }
#undef RW
    return -1;
}
int ref_get_grid(void *h, int which, float *out) { return grid_access((Ref *)h, which, out, nullptr); }
int ref_set_grid(void *h, int which, const float *in) { return grid_access((Ref *)h, which, nullptr, in); }

/* ---- phases of advance(), in call order (fluidsimulation.cpp:138-167) ---- */
float ref_cfl(void *h) { return ((Ref *)h)->sim._cfl(); }
void ref_update_liquid_sdf(void *h) { ((Ref *)h)->sim._updateLiquidSDF(); }
void ref_advect_velocity_field(void *h) { ((Ref *)h)->sim._advectVelocityField(); }
void ref_add_body_force(void *h, float dt) { ((Ref *)h)->sim._addBodyForce(dt); }

/* P2G of one component only (fluidsimulation.cpp:364-438): field/isset are face-grid sized */
void ref_p2g_component(void *h, int dir, float *field, float *isset) {
    Ref *r = (Ref *)h;
    int w = r->I + (dir == 0), ht = r->J + (dir == 1), d = r->K + (dir == 2);
    Array3d<float> f(w, ht, d, 0.0f);
    Array3d<bool> s(w, ht, d, false);
    r->sim._computeVelocityScalarField(f, s, dir);
    copy_out(f, field);
    copy_out(s, isset);
}

int ref_apply_viscosity(void *h, float dt) {
    Ref *r = (Ref *)h;
    FluidSimulation &s = r->sim;
    StdoutCapture cap; cap.begin();
    bool ok = true;
    if (r->visc_maxiter <= 0 && r->visc_tol <= 0) {
        s._applyViscosity(dt);
    } else {
        // same body as FluidSimulation::_applyViscosity (fluidsimulation.cpp:170-196) with a
        // ViscositySolver whose cap / tolerance members are overridden
        bool nonzero = false;
        float *v = s._viscosity.getRawArray();
        for (int i = 0; i < s._viscosity.getNumElements(); i++) if (v[i] > 0.0) nonzero = true;
        if (nonzero) {
            ViscositySolverParameters params;
            params.cellwidth = s._dx;
            params.deltaTime = dt;
            params.velocityField = &s._MACVelocity;
            params.liquidSDF = &s._liquidSDF;
            params.solidSDF = &s._solidSDF;
            params.viscosity = &s._viscosity;
            ViscositySolver vs;
            if (r->visc_maxiter > 0) vs._maxSolverIterations = r->visc_maxiter;
            if (r->visc_tol > 0) vs._solverTolerance = r->visc_tol;
            ok = vs.applyViscosityToVelocityField(params);
        }
    }
    std::string log = cap.end();
    double v;
    r->visc_iters = parse_after(log, "Viscosity Solver Iterations: ", &v) ? (int)v : -1;
    r->visc_err = parse_after(log, "Estimated Error: ", &v) ? v : -1;
    r->visc_failed = (log.find("FAILED") != std::string::npos) || !ok;
    return r->visc_failed;
}

void ref_compute_weights(void *h) { ((Ref *)h)->sim._computeWeights(); }

int ref_solve_pressure(void *h, float dt) {
    Ref *r = (Ref *)h;
    StdoutCapture cap; cap.begin();
    r->pressure = r->sim._solvePressure(dt);
    std::string log = cap.end();
    double v;
    r->pres_iters = parse_after(log, "Pressure Solver Iterations: ", &v) ? (int)v : -1;
    r->pres_err = parse_after(log, "Estimated Error: ", &v) ? v : -1;
    r->pres_failed = log.find("FAILED") != std::string::npos;
    return r->pres_failed;
}

void ref_apply_pressure(void *h, float dt) {
    Ref *r = (Ref *)h;
    r->sim._applyPressure(dt, r->pressure);
}
void ref_extrapolate(void *h) {
    Ref *r = (Ref *)h;
    r->sim._extrapolateVelocityField(r->sim._MACVelocity, r->sim._validVelocities);
}
void ref_constrain(void *h) { ((Ref *)h)->sim._constrainVelocityField(); }
void ref_advect_particles(void *h, float dt) { ((Ref *)h)->sim._advectFluidParticles(dt); }
void ref_update_particle_velocities(void *h) { ((Ref *)h)->sim._updateFluidParticleVelocities(); }

void ref_get_solver_stats(void *h, int *visc_iters, double *visc_err, int *pres_iters, double *pres_err) {
    Ref *r = (Ref *)h;
    *visc_iters = r->visc_iters; *visc_err = r->visc_err;
    *pres_iters = r->pres_iters; *pres_err = r->pres_err;
}

/* One substep of size dt with the phase order of advance(); seconds[7] =
 * {sdf, p2g+extrapolate, bodyforce, viscosity, project, constrain, advect}. */
void ref_substep(void *h, float dt, double *seconds) {
    Ref *r = (Ref *)h;
    typedef std::chrono::steady_clock clk;
    auto t0 = clk::now();
    auto lap = [&](int idx) {
        auto t1 = clk::now();
        if (seconds) seconds[idx] = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    };
    ref_update_liquid_sdf(h); lap(0);
    ref_advect_velocity_field(h); lap(1);
    ref_add_body_force(h, dt); lap(2);
    ref_apply_viscosity(h, dt); lap(3);
    ref_compute_weights(h);
    ref_solve_pressure(h, dt);
    ref_apply_pressure(h, dt);
    ref_extrapolate(h); lap(4);
    ref_constrain(h); lap(5);
    ref_advect_particles(h, dt); lap(6);
    (void)r;
}

/* The stock advance(dt) including its CFL loop; returns the number of substeps taken. */
int ref_advance(void *h, float dt) {
    Ref *r = (Ref *)h;
    StdoutCapture cap; cap.begin();
    r->sim.advance(dt);
    std::string log = cap.end();
    int n = 0;
    size_t p = 0;
    while ((p = log.find("Taking substep", p)) != std::string::npos) { n++; p++; }
    return n;
}

/* scalar helpers used as known-answer pins for the restatement (levelsetutils.cpp) */
float ref_fraction_inside2(float a, float b) { return LevelsetUtils::fractionInside(a, b); }
float ref_fraction_inside4(float bl, float br, float tl, float tr) { return LevelsetUtils::fractionInside(bl, br, tl, tr); }
float ref_volume_fraction8(const float *p) {
    return LevelsetUtils::volumeFraction(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
}

}  // extern "C"
