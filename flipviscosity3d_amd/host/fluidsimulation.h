// fluidsimulation.h -- host-side FluidSimulation: the reference's public class surface
// (reference fluidsimulation.h:39-63) in front of the HIP substep (include/flipv.h).
//
// A frame loop written against the reference -- initialize / addBoundary / addLiquid / setViscosity /
// setGravity / advance, reading and writing the public `particles` vector between frames
// (reference main.cpp:42-90) -- compiles and runs unchanged against this header.  Setup-time geometry
// (mesh level sets, seeding) is host C++; every per-substep phase runs on the GPU through the C-ABI.
#pragma once
#include <string>
#include <vector>

#include "../../include/flipv.h"
#include "array3d.h"
#include "meshlevelset.h"
#include "trianglemesh.h"
#include "vmath.h"

struct FluidParticle {  // reference fluidsimulation.h:39-48; 6 packed floats = one C-ABI particle record
    vmath::vec3 position;
    vmath::vec3 velocity;
    FluidParticle() {}
    FluidParticle(vmath::vec3 p) : position(p) {}
    FluidParticle(vmath::vec3 p, vmath::vec3 v) : position(p), velocity(v) {}
};
static_assert(sizeof(FluidParticle) == 24, "FluidParticle must be 6 packed floats");

class FluidSimulation {
public:
    FluidSimulation();
    ~FluidSimulation();
    FluidSimulation(const FluidSimulation &) = delete;
    FluidSimulation &operator=(const FluidSimulation &) = delete;

    // ---- the reference's API (reference fluidsimulation.h:53-63)
    void initialize(int i, int j, int k, float dx);
    void addBoundary(TriangleMesh &boundary, bool isInverted = false);
    void resetBoundary();
    void addLiquid(TriangleMesh &mesh);
    void setViscosity(float value);
    void setViscosity(Array3d<float> &vgrid);
    void setGravity(vmath::vec3 gravity);
    void setGravity(float gx, float gy, float gz);
    void advance(float dt);

    std::vector<FluidParticle> particles;

    // ---- the names BASELINE.json's north_star uses for the same calls
    void update(float dt) { advance(dt); }
    void addSolid(TriangleMesh &solid, bool isInverted = false) { addBoundary(solid, isInverted); }

    // ---- additions (no reference counterpart)
    enum SeedingMode {
        SEED_LIBC_RAND = 0,  // the reference's stream: three rand() draws per sample (fluidsimulation.cpp:79-84, .h:100-102)
        SEED_COUNTER = 1     // counter-based hash of (seed, cell, sample, axis): platform independent, skips empty cells
    };
    void setSeeding(SeedingMode mode, unsigned long long seed = 0) { _seedMode = mode; _seed = seed; }
    // Scene setup on the device (flipv_reset_boundary / flipv_add_boundary_mesh / flipv_add_liquid_mesh): mesh level
    // sets and seeding run as HIP kernels instead of host C++.  Call before initialize().  Needs SEED_COUNTER seeding
    // (libc rand() has no device counterpart); band values, signs and the seeded particles are identical to the host
    // path, far-field distances of the solid SDF are the relaxed ones (include/flipv.h).
    void setSetupOnDevice(bool on) { _setupOnDevice = on; }
    // Checkpoint (the reference cannot stop and resume a run): grid size, cell width, gravity, viscosity,
    // solid SDF, the particles (positions + velocities) and the MAC velocity field of the last substep -- _cfl() of the
    // next frame reads it (reference fluidsimulation.cpp:139, 241-269) -- in one little-endian binary file.  loadState()
    // re-initialises the simulation from the file; a run resumed from a checkpoint takes the same substeps (the same CFL steps from the
    // same particles and MAC field) as the uninterrupted one.  What the library's solvers remember between solves -- the AUTO viscosity
    // preconditioner's iteration history, the solver array layout -- is NOT in the file and need not be: every path AUTO can take converges to
    // the same tolerance, so a resumed run's velocities agree with the uninterrupted run's to solver tolerance, not bit for bit.
    // Both return false on I/O or format errors (a header that does not match the file length included).
    bool saveState(const std::string &path);
    bool loadState(const std::string &path);
    void setQuiet(bool q) { _quiet = q; }                  // the reference prints phase banners on stdout
    const flipv_stats &lastStats() const { return _stats; }
    flipv_context *context() { _ensureContext(); return _ctx; }   // created lazily: setup needs no GPU, advance() does
    MeshLevelSet &solidSDF();
    void getGridDimensions(int *i, int *j, int *k) const { *i = _isize; *j = _jsize; *k = _ksize; }
    float getCellSize() const { return _dx; }
    void markSolidDirty() { _solidDirty = true; }          // after editing solidSDF() by hand

private:
    void _initializeBoundary();
    void _destroy();
    void _ensureContext();

    int _isize = 0, _jsize = 0, _ksize = 0;
    float _dx = 0.0f;
    int _meshLevelSetExactBand = 3;  // reference fluidsimulation.h:121
    MeshLevelSet _solidSDF;
    flipv_context *_ctx = nullptr;
    flipv_stats _stats;
    SeedingMode _seedMode = SEED_LIBC_RAND;
    unsigned long long _seed = 0;
    bool _quiet = false;
    bool _setupOnDevice = false;
    bool _solidHostStale = false;   // device setup: the context holds the solid SDF, the host copy is fetched on demand
    bool _solidDirty = true, _viscosityDirty = true, _gravityDirty = true;
    float _viscosityUniform = 1.0f;          // reference fluidsimulation.cpp:39
    std::vector<float> _viscosityGrid;       // non-empty after setViscosity(Array3d<float>&)
    std::vector<float> _restoreVel[3];       // U, V, W of a loaded checkpoint until the context exists
    vmath::vec3 _gravity = vmath::vec3(0.0f, -9.81f, 0.0f);  // reference fluidsimulation.cpp:40
};
