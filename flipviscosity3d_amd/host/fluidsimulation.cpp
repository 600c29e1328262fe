#include "fluidsimulation.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// Precondition failures print and abort() like FLUIDSIM_ASSERT (reference fluidsimassert.h:24-37).
#define FLIPV_HOST_ASSERT(cond)                                                                         \
    do {                                                                                                \
        if (!(cond)) {                                                                                  \
            std::fprintf(stderr, "Assertion failed: %s, file %s, line %d\n", #cond, __FILE__, __LINE__); \
            std::abort();                                                                               \
        }                                                                                               \
    } while (0)

#define FLIPV_CALL(call)                                                                                  \
    do {                                                                                                  \
        int rc_ = (call);                                                                                 \
        if (rc_ < 0) {                                                                                    \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, flipv_last_error(_ctx));              \
            std::abort();                                                                                 \
        }                                                                                                 \
    } while (0)

namespace {

struct Box {  // the three AABB operations the simulation uses (reference aabb.cpp:27-29, 118-129, 213-219)
    vmath::vec3 position;
    double width = 0, height = 0, depth = 0;
    Box(double x, double y, double z, double w, double h, double d) : position((float)x, (float)y, (float)z), width(w), height(h), depth(d) {}
    explicit Box(const std::vector<vmath::vec3> &pts) {  // reference aabb.cpp:50-84
        if (pts.empty()) return;
        double minx = pts[0].x, miny = pts[0].y, minz = pts[0].z, maxx = minx, maxy = miny, maxz = minz;
        for (const vmath::vec3 &p : pts) {
            minx = std::fmin(p.x, minx); miny = std::fmin(p.y, miny); minz = std::fmin(p.z, minz);
            maxx = std::fmax(p.x, maxx); maxy = std::fmax(p.y, maxy); maxz = std::fmax(p.z, maxz);
        }
        const double eps = 1e-9;
        position = vmath::vec3((float)minx, (float)miny, (float)minz);
        width = maxx - minx + eps; height = maxy - miny + eps; depth = maxz - minz + eps;
    }
    void expand(double v) {
        const double h = 0.5 * v;
        position -= vmath::vec3((float)h, (float)h, (float)h);
        width += v; height += v; depth += v;
    }
    bool isPointInside(vmath::vec3 p) const {
        return p.x >= position.x && p.y >= position.y && p.z >= position.z && p.x < position.x + width &&
               p.y < position.y + height && p.z < position.z + depth;
    }
    vmath::vec3 minPoint() const { return position; }
    vmath::vec3 maxPoint() const { return position + vmath::vec3((float)width, (float)height, (float)depth); }
};

inline unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace

FluidSimulation::FluidSimulation() { std::memset(&_stats, 0, sizeof(_stats)); }
FluidSimulation::~FluidSimulation() { _destroy(); }

void FluidSimulation::_destroy() {
    if (_ctx) flipv_destroy(_ctx);
    _ctx = nullptr;
}

void FluidSimulation::initialize(int i, int j, int k, float dx) {  // reference fluidsimulation.cpp:26-43
    _destroy();
    _isize = i; _jsize = j; _ksize = k; _dx = dx;
    particles.clear();
    _viscosityUniform = 1.0f;
    _viscosityGrid.clear();
    _gravity = vmath::vec3(0.0f, -9.81f, 0.0f);
    for (int m = 0; m < 3; m++) std::vector<float>().swap(_restoreVel[m]);
    _solidDirty = _viscosityDirty = _gravityDirty = true;
    _initializeBoundary();
}

// The device context is created on first use so that scene setup (mesh level sets, seeding) also works on a
// host without a GPU; advance() needs one and fails loudly without it (there is no CPU substep).
void FluidSimulation::_ensureContext() {
    FLIPV_HOST_ASSERT(_isize > 0 && _jsize > 0 && _ksize > 0);
    if (!_ctx) {
        int rc = flipv_create(_isize, _jsize, _ksize, _dx, &_ctx);
        if (rc != FLIPV_OK) {
            std::fprintf(stderr, "flipv_create failed (%d): %s\n", rc, flipv_last_error(nullptr));
            std::abort();
        }
        _solidDirty = _viscosityDirty = _gravityDirty = true;
        if (!_restoreVel[0].empty()) {  // resumed from a checkpoint: the MAC field the last substep left (read by _cfl)
            for (int m = 0; m < 3; m++) {
                FLIPV_CALL(flipv_write_grid(_ctx, FLIPV_GRID_U + m, _restoreVel[m].data()));
                std::vector<float>().swap(_restoreVel[m]);
            }
        }
    }
    if (_solidDirty) FLIPV_CALL(flipv_set_solid_sdf(_ctx, _solidSDF.getRawArray()));
    if (_viscosityDirty) {
        if (_viscosityGrid.empty()) FLIPV_CALL(flipv_set_viscosity_uniform(_ctx, _viscosityUniform));
        else FLIPV_CALL(flipv_set_viscosity(_ctx, _viscosityGrid.data()));
    }
    if (_gravityDirty) FLIPV_CALL(flipv_set_gravity(_ctx, _gravity.x, _gravity.y, _gravity.z));
    _solidDirty = _viscosityDirty = _gravityDirty = false;
}

MeshLevelSet &FluidSimulation::solidSDF() {
    if (_solidHostStale) {
        FLIPV_CALL(flipv_read_grid(_ctx, FLIPV_GRID_SOLID_PHI, _solidSDF.getRawArray()));
        _solidHostStale = false;
    }
    return _solidSDF;
}

void FluidSimulation::_initializeBoundary() {  // reference fluidsimulation.cpp:198-239
    if (_setupOnDevice) {
        _solidSDF = MeshLevelSet(_isize, _jsize, _ksize, _dx);
        _solidDirty = false;
        _ensureContext();
        FLIPV_CALL(flipv_reset_boundary(_ctx));
        _solidHostStale = true;
        return;
    }
    const double eps = 1e-6;
    Box dom(0.0, 0.0, 0.0, _isize * _dx, _jsize * _dx, _ksize * _dx);
    dom.expand(-3 * _dx - eps);
    const vmath::vec3 p = dom.position;
    const float w = (float)dom.width, h = (float)dom.height, d = (float)dom.depth;
    TriangleMesh m;
    m.vertices = {vmath::vec3(p.x, p.y, p.z),         vmath::vec3(p.x + w, p.y, p.z),         vmath::vec3(p.x + w, p.y, p.z + d),
                  vmath::vec3(p.x, p.y, p.z + d),     vmath::vec3(p.x, p.y + h, p.z),         vmath::vec3(p.x + w, p.y + h, p.z),
                  vmath::vec3(p.x + w, p.y + h, p.z + d), vmath::vec3(p.x, p.y + h, p.z + d)};
    m.triangles = {Triangle(0, 1, 2), Triangle(0, 2, 3), Triangle(4, 7, 6), Triangle(4, 6, 5), Triangle(0, 3, 7), Triangle(0, 7, 4),
                   Triangle(1, 5, 6), Triangle(1, 6, 2), Triangle(0, 4, 5), Triangle(0, 5, 1), Triangle(3, 2, 6), Triangle(3, 6, 7)};
    _solidSDF = MeshLevelSet(_isize, _jsize, _ksize, _dx);
    _solidSDF.calculateSignedDistanceField(m, _meshLevelSetExactBand);
    _solidSDF.negate();
    _solidDirty = true;
}

void FluidSimulation::addBoundary(TriangleMesh &boundary, bool isInverted) {  // reference fluidsimulation.cpp:45-58
    Box domain(0.0, 0.0, 0.0, _isize * _dx, _jsize * _dx, _ksize * _dx);
    Box bbox(boundary.vertices);
    FLIPV_HOST_ASSERT(domain.isPointInside(bbox.minPoint()) && domain.isPointInside(bbox.maxPoint()));
    if (_setupOnDevice) {
        FLIPV_HOST_ASSERT(!_solidDirty);  // solidSDF() was not edited by hand since the last device operation
        _ensureContext();
        FLIPV_CALL(flipv_add_boundary_mesh(_ctx, &boundary.vertices[0].x, boundary.vertices.size(), &boundary.triangles[0].tri[0],
                                           boundary.triangles.size(), isInverted ? 1 : 0));
        _solidHostStale = true;
        return;
    }
    MeshLevelSet sdf(_isize, _jsize, _ksize, _dx);
    sdf.calculateSignedDistanceField(boundary, _meshLevelSetExactBand);
    if (isInverted) sdf.negate();
    _solidSDF.calculateUnion(sdf);
    _solidDirty = true;
}

void FluidSimulation::resetBoundary() { _initializeBoundary(); }  // reference fluidsimulation.cpp:60-62

void FluidSimulation::addLiquid(TriangleMesh &mesh) {  // reference fluidsimulation.cpp:64-97
    Box domain(0.0, 0.0, 0.0, _isize * _dx, _jsize * _dx, _ksize * _dx);
    Box bbox(mesh.vertices);
    FLIPV_HOST_ASSERT(domain.isPointInside(bbox.minPoint()) && domain.isPointInside(bbox.maxPoint()));
    if (_setupOnDevice) {
        FLIPV_HOST_ASSERT(_seedMode == SEED_COUNTER);
        _ensureContext();  // uploads a hand-edited solid SDF if there is one
        FLIPV_CALL(flipv_upload_particles(_ctx, particles.empty() ? nullptr : &particles[0].position.x, particles.size()));
        size_t added = 0, n = 0;
        FLIPV_CALL(flipv_add_liquid_mesh(_ctx, &mesh.vertices[0].x, mesh.vertices.size(), &mesh.triangles[0].tri[0], mesh.triangles.size(),
                                         _seed, &added));
        particles.resize(particles.size() + added);
        FLIPV_CALL(flipv_download_particles(_ctx, particles.empty() ? nullptr : &particles[0].position.x, particles.size(), &n));
        return;
    }
    MeshLevelSet meshSDF(_isize, _jsize, _ksize, _dx);
    meshSDF.calculateSignedDistanceField(mesh, _meshLevelSetExactBand);
    const double dx = _dx;
    const float *nodes = meshSDF.getRawArray();
    const int nw = _isize + 1, nh = _jsize + 1;
    for (int k = 0; k < _ksize; k++)
        for (int j = 0; j < _jsize; j++)
            for (int i = 0; i < _isize; i++) {
                if (_seedMode == SEED_COUNTER) {
                    // no sample of this cell can be inside the mesh if all eight corner distances are >= 0
                    bool anyNeg = false;
                    for (int c = 0; c < 8 && !anyNeg; c++)
                        anyNeg = nodes[(size_t)(i + (c & 1)) + (size_t)nw * ((size_t)(j + ((c >> 1) & 1)) + (size_t)nh * (size_t)(k + (c >> 2)))] < 0.0f;
                    if (!anyNeg) continue;
                }
                const vmath::vec3 gpos((float)(i * dx), (float)(j * dx), (float)(k * dx));
                const unsigned long long cell = (unsigned long long)i + (unsigned long long)_isize * ((unsigned long long)j + (unsigned long long)_jsize * (unsigned long long)k);
                for (int s = 0; s < 8; s++) {
                    float jit[3];
                    for (int a = 0; a < 3; a++) {
                        double u;
                        if (_seedMode == SEED_LIBC_RAND) {
                            // _randomDouble(0, dx) (reference fluidsimulation.h:100-102)
                            u = 0.0 + (double)std::rand() / ((double)RAND_MAX / (dx - 0.0));
                        } else {
                            const unsigned long long hsh = splitmix64(splitmix64(_seed) ^ (cell * 24ull + (unsigned long long)(s * 3 + a)));
                            u = (double)(hsh >> 11) * (1.0 / 9007199254740992.0) * dx;
                        }
                        jit[a] = (float)u;
                    }
                    const vmath::vec3 pos = gpos + vmath::vec3(jit[0], jit[1], jit[2]);
                    if (meshSDF.trilinearInterpolate(pos) < 0.0) {
                        const float solid_phi = _solidSDF.trilinearInterpolate(pos);
                        if (solid_phi >= 0) particles.push_back(FluidParticle(pos));
                    }
                }
            }
}

// ---- checkpoint file "FLIPVCK2": int32 I,J,K, float dx, float gravity[3], int32 viscosity kind (0 uniform, 1 grid),
// float uniform viscosity, uint64 particle count, int32 hasVelocity, then solid SDF nodes, [viscosity nodes], particles
// (6 floats each), [U (I+1,J,K), V (I,J+1,K), W (I,J,K+1)].  The MAC field is state: _cfl() of the next advance() reads
// the velocities the previous substep left (reference fluidsimulation.cpp:139, 241-269), so a resumed run only takes the
// same substeps as the uninterrupted one if the field comes back with the particles.
bool FluidSimulation::saveState(const std::string &path) {
    const MeshLevelSet &sdf = solidSDF();
    const size_t nodes = (size_t)(_isize + 1) * (_jsize + 1) * (_ksize + 1);
    const size_t nf[3] = {(size_t)(_isize + 1) * _jsize * _ksize, (size_t)_isize * (_jsize + 1) * _ksize, (size_t)_isize * _jsize * (_ksize + 1)};
    std::vector<float> vel[3];
    int hasVel = 0;
    if (_ctx) {  // a context exists once a frame has run (or device setup was used): fetch its field
        hasVel = 1;
        for (int m = 0; m < 3; m++) {
            vel[m].resize(nf[m]);
            if (flipv_read_grid(_ctx, FLIPV_GRID_U + m, vel[m].data()) < 0) return false;
        }
    } else if (!_restoreVel[0].empty()) {  // loaded but not advanced yet: pass the loaded field on
        hasVel = 1;
        for (int m = 0; m < 3; m++) vel[m] = _restoreVel[m];
    }
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const int dims[3] = {_isize, _jsize, _ksize};
    const float g[3] = {_gravity.x, _gravity.y, _gravity.z};
    const int kind = _viscosityGrid.empty() ? 0 : 1;
    const unsigned long long np = particles.size();
    bool ok = std::fwrite("FLIPVCK2", 1, 8, f) == 8 && std::fwrite(dims, sizeof(int), 3, f) == 3 && std::fwrite(&_dx, sizeof(float), 1, f) == 1 &&
              std::fwrite(g, sizeof(float), 3, f) == 3 && std::fwrite(&kind, sizeof(int), 1, f) == 1 &&
              std::fwrite(&_viscosityUniform, sizeof(float), 1, f) == 1 && std::fwrite(&np, sizeof(np), 1, f) == 1 &&
              std::fwrite(&hasVel, sizeof(int), 1, f) == 1;
    ok = ok && std::fwrite(sdf.getRawArray(), sizeof(float), nodes, f) == nodes;
    if (kind) ok = ok && std::fwrite(_viscosityGrid.data(), sizeof(float), nodes, f) == nodes;
    if (np) ok = ok && std::fwrite(&particles[0].position.x, sizeof(FluidParticle), (size_t)np, f) == (size_t)np;
    if (hasVel)
        for (int m = 0; m < 3; m++) ok = ok && std::fwrite(vel[m].data(), sizeof(float), nf[m], f) == nf[m];
    return std::fclose(f) == 0 && ok;
}

// The header is not trusted: dimensions are capped, the payload the header promises is checked against the file length
// before anything is allocated, and no exception leaves this function (it sits behind an extern "C" wrapper).
bool FluidSimulation::loadState(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    try {
        char magic[8];
        int dims[3], kind = 0, hasVel = 0;
        float dx = 0, g[3], nu = 0;
        unsigned long long np = 0;
        // "FLIPVCK1" (the previous layout) is the same file without the hasVelocity word and the MAC field: still loadable
        bool ok = std::fread(magic, 1, 8, f) == 8 && (std::memcmp(magic, "FLIPVCK2", 8) == 0 || std::memcmp(magic, "FLIPVCK1", 8) == 0);
        const bool v1 = ok && std::memcmp(magic, "FLIPVCK1", 8) == 0;
        ok = ok && std::fread(dims, sizeof(int), 3, f) == 3 &&
             std::fread(&dx, sizeof(float), 1, f) == 1 && std::fread(g, sizeof(float), 3, f) == 3 && std::fread(&kind, sizeof(int), 1, f) == 1 &&
             std::fread(&nu, sizeof(float), 1, f) == 1 && std::fread(&np, sizeof(np), 1, f) == 1 && (v1 || std::fread(&hasVel, sizeof(int), 1, f) == 1);
        const int maxDim = 1 << 14;  // (2^14+1)^3 nodes still fit a size_t product by a wide margin
        ok = ok && dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[0] <= maxDim && dims[1] <= maxDim && dims[2] <= maxDim &&
             dx > 0 && (kind == 0 || kind == 1) && nu >= 0 && (hasVel == 0 || hasVel == 1);
        if (!ok) { std::fclose(f); return false; }
        const size_t I = (size_t)dims[0], J = (size_t)dims[1], K = (size_t)dims[2];
        const size_t nodes = (I + 1) * (J + 1) * (K + 1);
        const size_t nf[3] = {(I + 1) * J * K, I * (J + 1) * K, I * J * (K + 1)};
        const long here = std::ftell(f);
        ok = here >= 0 && std::fseek(f, 0, SEEK_END) == 0;
        const long end = ok ? std::ftell(f) : -1;
        ok = ok && end >= here && std::fseek(f, here, SEEK_SET) == 0;
        if (ok) {
            const unsigned long long remaining = (unsigned long long)(end - here);
            const unsigned long long gridBytes = 4ull * (nodes * (unsigned long long)(1 + kind) + (hasVel ? nf[0] + nf[1] + nf[2] : 0));
            ok = gridBytes <= remaining && np <= (remaining - gridBytes) / sizeof(FluidParticle) &&
                 gridBytes + np * sizeof(FluidParticle) == remaining;
        }
        if (!ok) { std::fclose(f); return false; }
        std::vector<float> solid(nodes), visc(kind ? nodes : 0), vel[3];
        std::vector<FluidParticle> parts((size_t)np);
        ok = std::fread(solid.data(), sizeof(float), nodes, f) == nodes;
        if (kind) ok = ok && std::fread(visc.data(), sizeof(float), nodes, f) == nodes;
        if (np) ok = ok && std::fread(&parts[0].position.x, sizeof(FluidParticle), (size_t)np, f) == (size_t)np;
        if (hasVel)
            for (int m = 0; m < 3; m++) {
                vel[m].resize(nf[m]);
                ok = ok && std::fread(vel[m].data(), sizeof(float), nf[m], f) == nf[m];
            }
        std::fclose(f);
        f = nullptr;
        if (!ok) return false;
        const bool dev = _setupOnDevice;
        _setupOnDevice = false;  // the solid SDF comes from the file, no boundary to build
        _destroy();
        _isize = dims[0]; _jsize = dims[1]; _ksize = dims[2]; _dx = dx;
        _solidSDF = MeshLevelSet(_isize, _jsize, _ksize, _dx);
        std::memcpy(_solidSDF.getRawArray(), solid.data(), nodes * sizeof(float));
        _solidHostStale = false;
        _viscosityUniform = nu;
        _viscosityGrid.swap(visc);
        _gravity = vmath::vec3(g[0], g[1], g[2]);
        particles.swap(parts);
        for (int m = 0; m < 3; m++) _restoreVel[m].swap(vel[m]);   // written into the context when it is created
        _solidDirty = _viscosityDirty = _gravityDirty = true;
        _setupOnDevice = dev;
        return true;
    } catch (...) {  // bad_alloc / length_error from a header that passed the checks but cannot be served
        if (f) std::fclose(f);
        return false;
    }
}

void FluidSimulation::setViscosity(float value) {  // reference fluidsimulation.cpp:99-108
    FLIPV_HOST_ASSERT(value >= 0.0);
    _viscosityUniform = value;
    _viscosityGrid.clear();
    _viscosityDirty = true;
}

void FluidSimulation::setViscosity(Array3d<float> &vgrid) {  // reference fluidsimulation.cpp:110-124
    FLIPV_HOST_ASSERT(vgrid.width == _isize + 1 && vgrid.height == _jsize + 1 && vgrid.depth == _ksize + 1);
    const float *raw = vgrid.getRawArray();
    for (size_t t = 0; t < vgrid.size(); t++) FLIPV_HOST_ASSERT(raw[t] >= 0.0);
    _viscosityGrid.assign(raw, raw + vgrid.size());
    _viscosityDirty = true;
}

void FluidSimulation::setGravity(vmath::vec3 g) {  // reference fluidsimulation.cpp:126-128
    _gravity = g;
    _gravityDirty = true;
}
void FluidSimulation::setGravity(float gx, float gy, float gz) { setGravity(vmath::vec3(gx, gy, gz)); }

void FluidSimulation::advance(float dt) {  // reference fluidsimulation.cpp:135-168
    _ensureContext();
    // `particles` is public and may have been edited since the last frame: it is the source of truth
    FLIPV_CALL(flipv_upload_particles(_ctx, particles.empty() ? nullptr : &particles[0].position.x, particles.size()));
    FLIPV_CALL(flipv_advance(_ctx, dt, &_stats));
    size_t n = 0;
    FLIPV_CALL(flipv_download_particles(_ctx, particles.empty() ? nullptr : &particles[0].position.x, particles.size(), &n));
    if (!_quiet) {
        std::printf("advance(%g): %d substep(s), %.3f ms on the GPU; viscosity %d its (res %.3e), pressure %d its (res %.3e)\n",
                    dt, _stats.substeps, _stats.total_ms, _stats.viscosity.iterations, _stats.viscosity.residual,
                    _stats.pressure.iterations, _stats.pressure.residual);
    }
}
