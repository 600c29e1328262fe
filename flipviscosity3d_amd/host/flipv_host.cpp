#include "flipv_host.h"

#include <cstdlib>
#include <cstring>

#include "fluidsimulation.h"

struct flipvh_sim {
    FluidSimulation sim;
};

namespace {
TriangleMesh makeMesh(const float *verts, int nv, const int *tris, int nt) {
    TriangleMesh m;
    m.vertices.resize((size_t)nv);
    for (int i = 0; i < nv; i++) m.vertices[i] = vmath::vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    m.triangles.resize((size_t)nt);
    for (int i = 0; i < nt; i++) m.triangles[i] = Triangle(tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]);
    return m;
}
bool insideDomain(FluidSimulation &s, const TriangleMesh &m) {  // the check addBoundary/addLiquid assert on
    int I, J, K;
    s.getGridDimensions(&I, &J, &K);
    const float dx = s.getCellSize();
    const double lx = I * dx, ly = J * dx, lz = K * dx;
    for (const vmath::vec3 &p : m.vertices)
        if (!(p.x >= 0 && p.y >= 0 && p.z >= 0 && p.x + 1e-9 < lx && p.y + 1e-9 < ly && p.z + 1e-9 < lz)) return false;
    return !m.vertices.empty();
}
}  // namespace

extern "C" {

flipvh_sim *flipvh_create_ex(int I, int J, int K, float dx, int setup_on_device) {
    if (I < 1 || J < 1 || K < 1 || !(dx > 0)) return nullptr;
    flipvh_sim *s = new flipvh_sim();
    s->sim.setQuiet(true);
    if (setup_on_device) {
        s->sim.setSetupOnDevice(true);
        s->sim.setSeeding(FluidSimulation::SEED_COUNTER, 0);
    }
    s->sim.initialize(I, J, K, dx);
    return s;
}
flipvh_sim *flipvh_create(int I, int J, int K, float dx) { return flipvh_create_ex(I, J, K, dx, 0); }
void flipvh_destroy(flipvh_sim *s) { delete s; }

int flipvh_add_boundary(flipvh_sim *s, const float *verts, int nv, const int *tris, int nt, int inverted) {
    TriangleMesh m = makeMesh(verts, nv, tris, nt);
    if (!insideDomain(s->sim, m)) return -1;
    s->sim.addBoundary(m, inverted != 0);
    return 0;
}
void flipvh_reset_boundary(flipvh_sim *s) { s->sim.resetBoundary(); }
void flipvh_set_seeding(flipvh_sim *s, int mode, unsigned long long seed) {
    s->sim.setSeeding(mode ? FluidSimulation::SEED_COUNTER : FluidSimulation::SEED_LIBC_RAND, seed);
}
int flipvh_add_liquid(flipvh_sim *s, const float *verts, int nv, const int *tris, int nt) {
    TriangleMesh m = makeMesh(verts, nv, tris, nt);
    if (!insideDomain(s->sim, m)) return -1;
    s->sim.addLiquid(m);
    return 0;
}
int flipvh_set_viscosity(flipvh_sim *s, float v) {
    if (!(v >= 0)) return -1;
    s->sim.setViscosity(v);
    return 0;
}
int flipvh_set_viscosity_grid(flipvh_sim *s, const float *nodes) {
    int I, J, K;
    s->sim.getGridDimensions(&I, &J, &K);
    Array3d<float> g(I + 1, J + 1, K + 1);
    std::memcpy(g.getRawArray(), nodes, g.size() * sizeof(float));
    for (size_t t = 0; t < g.size(); t++)
        if (!(nodes[t] >= 0)) return -1;
    s->sim.setViscosity(g);
    return 0;
}
void flipvh_set_gravity(flipvh_sim *s, float gx, float gy, float gz) { s->sim.setGravity(gx, gy, gz); }
size_t flipvh_num_particles(flipvh_sim *s) { return s->sim.particles.size(); }
void flipvh_get_particles(flipvh_sim *s, float *aos6) {
    if (!s->sim.particles.empty()) std::memcpy(aos6, &s->sim.particles[0], s->sim.particles.size() * sizeof(FluidParticle));
}
void flipvh_set_particles(flipvh_sim *s, const float *aos6, size_t n) {
    s->sim.particles.resize(n);
    if (n) std::memcpy((void *)&s->sim.particles[0], aos6, n * sizeof(FluidParticle));
}
void flipvh_get_solid_sdf(flipvh_sim *s, float *nodes) {
    int I, J, K;
    s->sim.getGridDimensions(&I, &J, &K);
    std::memcpy(nodes, s->sim.solidSDF().getRawArray(), (size_t)(I + 1) * (J + 1) * (K + 1) * sizeof(float));
}
void flipvh_get_dims(flipvh_sim *s, int *ijk) { s->sim.getGridDimensions(&ijk[0], &ijk[1], &ijk[2]); }
int flipvh_save_state(flipvh_sim *s, const char *path) { return s->sim.saveState(path) ? 0 : -1; }
int flipvh_load_state(flipvh_sim *s, const char *path) { return s->sim.loadState(path) ? 0 : -1; }
int flipvh_advance(flipvh_sim *s, float dt, flipv_stats *stats) {
    s->sim.advance(dt);
    if (stats) *stats = s->sim.lastStats();
    return 0;
}
flipv_context *flipvh_context(flipvh_sim *s) { return s->sim.context(); }

void flipvh_mesh_sdf(int I, int J, int K, float dx, const float *verts, int nv, const int *tris, int nt, int band,
                     float *phi, int *closest) {
    MeshLevelSet ls(I, J, K, dx);
    TriangleMesh m = makeMesh(verts, nv, tris, nt);
    ls.calculateSignedDistanceField(m, band);
    const size_t n = (size_t)(I + 1) * (J + 1) * (K + 1);
    std::memcpy(phi, ls.getRawArray(), n * sizeof(float));
    if (closest) std::memcpy(closest, ls.getClosestRawArray(), n * sizeof(int));
}

int flipvh_load_ply(const char *path, int *nv, int *nt, float *verts, int *tris) {
    TriangleMesh m;
    if (!m.loadPLY(path)) return -1;
    *nv = m.numVertices();
    *nt = m.numTriangles();
    if (verts) std::memcpy(verts, m.vertices.data(), (size_t)*nv * 12);
    if (tris)
        for (int i = 0; i < *nt; i++) std::memcpy(tris + 3 * i, m.triangles[i].tri, 12);
    return 0;
}

}  // extern "C"
