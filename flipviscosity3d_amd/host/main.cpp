// fluidsim_mi355x -- the reference's example driver (reference main.cpp:42-90: bunny dropped inside a
// spherical container, viscosity 5, gravity -9.81 y, 300 frames of 0.01 s, one particle dump per frame) on
// top of the MI355X-native FluidSimulation.  Usage:
//   fluidsim_mi355x [--size N] [--frames F] [--dt T] [--viscosity V] [--boundary file.ply[:inverted]]
//                   [--liquid file.ply] [--mesh-dir DIR] [--no-export] [--ply] [--gpu-setup]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>

#include "fluidsimulation.h"
#include "trianglemesh.h"

static void export_particles(int frame, const std::vector<FluidParticle> &particles, bool obj, bool ply) {
    TriangleMesh mesh;
    mesh.vertices.reserve(particles.size());
    for (const FluidParticle &p : particles) mesh.vertices.push_back(p.position);
    char name[32];
    std::snprintf(name, sizeof(name), "%04d", frame);
    if (obj) mesh.writeMeshToOBJ(std::string(name) + ".obj");
    if (ply) mesh.writeMeshToPLY(std::string(name) + ".ply");
}

int main(int argc, char **argv) {
    int size = 64, frames = 300;
    float timestep = 0.01f, viscosity = 5.0f;
    std::string meshDir = "sample_meshes", boundary = "sphere_large.ply", liquid = "stanford_bunny.ply";
    bool inverted = true, exportObj = true, exportPly = false, gpuSetup = false;
    for (int a = 1; a < argc; a++) {
        const std::string s = argv[a];
        auto next = [&]() -> const char * { return a + 1 < argc ? argv[++a] : ""; };
        if (s == "--size") size = std::atoi(next());
        else if (s == "--frames") frames = std::atoi(next());
        else if (s == "--dt") timestep = (float)std::atof(next());
        else if (s == "--viscosity") viscosity = (float)std::atof(next());
        else if (s == "--mesh-dir") meshDir = next();
        else if (s == "--liquid") liquid = next();
        else if (s == "--boundary") {
            boundary = next();
            const size_t c = boundary.find(":inverted");
            inverted = c != std::string::npos;
            if (inverted) boundary = boundary.substr(0, c);
        } else if (s == "--no-export") exportObj = exportPly = false;
        else if (s == "--ply") { exportPly = true; exportObj = false; }
        else if (s == "--gpu-setup") gpuSetup = true;  // mesh level sets + seeding as HIP kernels (counter-based jitter)
        else { std::fprintf(stderr, "unknown option %s\n", s.c_str()); return 2; }
    }
    const float dx = 1.0f / (float)size;  // reference main.cpp:53
    FluidSimulation fluidsim;
    if (gpuSetup) {
        fluidsim.setSetupOnDevice(true);
        fluidsim.setSeeding(FluidSimulation::SEED_COUNTER, 0);
    }
    fluidsim.initialize(size, size, size, dx);

    TriangleMesh boundaryMesh;
    if (!boundary.empty() && boundary != "none") {
        if (!boundaryMesh.loadPLY(meshDir + "/" + boundary)) { std::cout << "Error loading boundary mesh: " << boundary << std::endl; return 1; }
        fluidsim.addBoundary(boundaryMesh, inverted);
    }
    TriangleMesh liquidMesh;
    if (!liquidMesh.loadPLY(meshDir + "/" + liquid)) { std::cout << "Error loading liquid mesh: " << liquid << std::endl; return 1; }
    fluidsim.addLiquid(liquidMesh);
    fluidsim.setViscosity(viscosity);
    fluidsim.setGravity(0.0f, -9.81f, 0.0f);
    std::cout << "particles: " << fluidsim.particles.size() << std::endl;
    for (int frame = 0; frame < frames; frame++) {
        if (exportObj || exportPly) export_particles(frame, fluidsim.particles, exportObj, exportPly);
        fluidsim.advance(timestep);
    }
    return 0;
}
