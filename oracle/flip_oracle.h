/*
 * oracle/flip_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, single thread) of the FLIP substep of rlguy/FLIPViscosity3D,
 * FluidSimulation::advance() (reference fluidsimulation.cpp:135-168) and everything below
 * it.  It is the checker for the HIP path and the "port" CPU baseline; it is never linked
 * into or called by the product library.  Parity status: PINNED -- every function is compared
 * bit-for-bit (solvers: iteration-for-iteration) against the compiled reference
 * (oracle/_ref) in tests/test_oracle_vs_reference.py and against the committed fixtures in
 * tests/golden/ (tests/test_oracle_golden.py).
 *
 * All arrays are host arrays in the reference's Array3d layout: flat = i + w*(j + h*k)
 * (reference array3d.h:397-400).  For an I x J x K grid:
 *   U (I+1,J,K)  V (I,J+1,K)  W (I,J,K+1)   liquid phi (I,J,K)   solid phi / viscosity (I+1,J+1,K+1)
 * Particles are AoS {px,py,pz,vx,vy,vz} float (reference fluidsimulation.h:39-48).
 */
#ifndef FLIP_ORACLE_H
#define FLIP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- level-set helpers (reference levelsetutils.cpp) ---- */
float oracle_fraction_inside2(float phi_left, float phi_right);
float oracle_fraction_inside4(float bl, float br, float tl, float tr);
float oracle_volume_fraction8(const float p[8]); /* order 000,100,010,110,001,101,011,111 */

/* ---- per-phase operators ---- */
/* K1+K2  ParticleLevelSet::calculateSignedDistanceField (particlelevelset.cpp:77-139) */
void oracle_particle_sdf(int I, int J, int K, float dx, const float *aos6, size_t n,
                         const float *solid_nodes, float *phi);

/* K3  _computeVelocityScalarField for one component (fluidsimulation.cpp:364-438) */
void oracle_p2g_component(int I, int J, int K, float dx, const float *aos6, size_t n, int dir,
                          float *field, uint8_t *isset);

/* K3+K4  _advectVelocityFieldU/V/W (fluidsimulation.cpp:440-498): clears U,V,W and valid */
void oracle_p2g(int I, int J, int K, float dx, const float *aos6, size_t n, const float *phi,
                float *U, float *V, float *W, uint8_t *validU, uint8_t *validV, uint8_t *validW);

/* K5  MACVelocityField::_extrapolateGrid (macvelocityfield.cpp:580-687) on one w x h x d grid */
void oracle_extrapolate_grid(int w, int h, int d, float *grid, const uint8_t *valid, int layers);

/* K6  _addBodyForce (fluidsimulation.cpp:271-312) */
void oracle_body_force(int I, int J, int K, const float *phi, float *U, float *V, float *W,
                       float gx, float gy, float gz, float dt);

/* K16 _cfl (fluidsimulation.cpp:241-269) */
float oracle_cfl(int I, int J, int K, float dx, const float *U, const float *V, const float *W,
                 float cfl_number);

/* K10 _computeWeights (fluidsimulation.cpp:549-582) */
void oracle_compute_weights(int I, int J, int K, const float *solid_nodes, float *wU, float *wV, float *wW);

typedef struct {
    int iterations;      /* as the reference counts them (pressure: zero-based index of the converging
                            iteration, pressuresolver.cpp:549-552; viscosity: count, pcgsolver.h:272) */
    double residual;     /* final inf-norm residual */
    int status;          /* 0 converged, 1 cap reached (solution still applied), 2 failed (field untouched),
                            3 trivial (rhs == 0 / below tolerance) */
    int rows;            /* number of unknowns */
    long nnz;            /* matrix non-zeros (viscosity) */
} oracle_solve_info;

/* K11+K12 PressureSolver::solve (pressuresolver.cpp:166-567).  pressure (I,J,K) out. */
void oracle_pressure_solve(int I, int J, int K, float dx, float dt, const float *U, const float *V,
                           const float *W, const float *wU, const float *wV, const float *wW,
                           const float *phi, float minfrac, double tol, int maxiter,
                           float *pressure, oracle_solve_info *info);

/* K13 _applyPressure (fluidsimulation.cpp:598-688) */
void oracle_apply_pressure(int I, int J, int K, float dx, float dt, const float *pressure,
                           const float *phi, const float *wU, const float *wV, const float *wW,
                           float minfrac, float *U, float *V, float *W, uint8_t *validU,
                           uint8_t *validV, uint8_t *validW);

/* K14 _constrainVelocityField (fluidsimulation.cpp:696-729) */
void oracle_constrain(int I, int J, int K, const float *wU, const float *wV, const float *wW,
                      float *U, float *V, float *W, float *sU, float *sV, float *sW);

/* K7-K9 ViscositySolver::applyViscosityToVelocityField (viscositysolver.cpp:41-727) with
 * PCGSolver<double> (pcgsolver/pcgsolver.h).  U,V,W updated in place unless status == 2.
 * If the viscosity grid is all zero the call is a no-op (fluidsimulation.cpp:171-184). */
void oracle_viscosity_solve(int I, int J, int K, float dx, float dt, float *U, float *V, float *W,
                            const float *phi, const float *solid_nodes, const float *viscosity_nodes,
                            double tol, int maxiter, double accept_tol, oracle_solve_info *info);

/* The seven volume-fraction lattices of ViscositySolver::_computeVolumeGrid
 * (viscositysolver.cpp:135-270); outputs sized center (I,J,K), U (I+1,J,K), V, W,
 * edgeU (I,J+1,K+1), edgeV (I+1,J,K+1), edgeW (I+1,J+1,K).  Any output may be NULL. */
/* research hook: path != NULL makes oracle_viscosity_solve also dump the assembled system (header {rows, ROWCAP, faces, 0}
 * as int64, then counts, columns, values, rhs, face->row table); the pointer must stay valid */
void oracle_viscosity_dump_to(const char *path);
/* research hook: replace the derived control volumes (7 grids) / face states (3 grids, 1 = FLUID, 2 = SOLID) */
void oracle_viscosity_override(const float *const *vols7, const unsigned char *const *states3);
void oracle_viscosity_volumes(int I, int J, int K, float dx, const float *phi, float *center, float *volU,
                              float *volV, float *volW, float *edgeU, float *edgeV, float *edgeW);

/* K15 _advectFluidParticles incl. _updateFluidParticleVelocities (fluidsimulation.cpp:315-352) */
void oracle_update_particle_velocities(int I, int J, int K, float dx, float *aos6, size_t n,
                                       const float *U, const float *V, const float *W, const float *sU,
                                       const float *sV, const float *sW, float pic_ratio);
void oracle_advect_particles(int I, int J, int K, float dx, float dt, float *aos6, size_t n,
                             const float *U, const float *V, const float *W, const float *sU,
                             const float *sV, const float *sW, const float *solid_nodes, float pic_ratio);

/* ---- whole simulation state (for end-to-end parity and the CPU baseline) ---- */
typedef struct oracle_sim oracle_sim;

oracle_sim *oracle_sim_create(int I, int J, int K, float dx);
void oracle_sim_destroy(oracle_sim *s);
void oracle_sim_set_solid(oracle_sim *s, const float *solid_nodes);
void oracle_sim_set_viscosity(oracle_sim *s, const float *viscosity_nodes);
void oracle_sim_set_gravity(oracle_sim *s, float gx, float gy, float gz);
void oracle_sim_set_particles(oracle_sim *s, const float *aos6, size_t n);
size_t oracle_sim_num_particles(oracle_sim *s);
void oracle_sim_get_particles(oracle_sim *s, float *aos6);
void oracle_sim_set_solver_limits(oracle_sim *s, double ptol, int pmaxiter, double vtol, int vmaxiter);
/* which: the flipv_grid ids of include/flipv.h; valid masks are returned as 0/1 floats */
int oracle_sim_get_grid(oracle_sim *s, int which, float *out);
int oracle_sim_set_grid(oracle_sim *s, int which, const float *in);
/* one substep; seconds[7] = {sdf, p2g+extrapolate, bodyforce, viscosity, project, constrain, advect} */
void oracle_sim_substep(oracle_sim *s, float dt, double *seconds, oracle_solve_info *visc, oracle_solve_info *pres);
/* advance(dt) with the CFL loop (fluidsimulation.cpp:135-168); returns the number of substeps */
int oracle_sim_advance(oracle_sim *s, float dt);

#ifdef __cplusplus
}
#endif
#endif
