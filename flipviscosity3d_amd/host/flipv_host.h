/* flipv_host.h -- C wrapper around the host-side FluidSimulation class (fluidsimulation.h) so that
 * non-C++ hosts (the ctypes test driver, bench.py) can build scenes and run frames.  Scene setup
 * (mesh level sets, seeding) is host code and works without a GPU; flipvh_advance and
 * flipvh_context need a HIP device. */
#ifndef FLIPV_HOST_H
#define FLIPV_HOST_H
#include <stddef.h>
#include "../../include/flipv.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct flipvh_sim flipvh_sim;

flipvh_sim *flipvh_create(int isize, int jsize, int ksize, float dx);      /* FluidSimulation::initialize */
flipvh_sim *flipvh_create_ex(int isize, int jsize, int ksize, float dx, int setup_on_device); /* + setSetupOnDevice (needs a GPU; counter seeding) */
void flipvh_destroy(flipvh_sim *s);
/* 0 ok, -1 mesh bounding box outside the domain (the reference asserts, fluidsimulation.cpp:46-49) */
int flipvh_add_boundary(flipvh_sim *s, const float *verts, int nv, const int *tris, int nt, int inverted);
void flipvh_reset_boundary(flipvh_sim *s);
void flipvh_set_seeding(flipvh_sim *s, int mode, unsigned long long seed); /* 0 libc rand() stream, 1 counter hash */
int flipvh_add_liquid(flipvh_sim *s, const float *verts, int nv, const int *tris, int nt);
int flipvh_set_viscosity(flipvh_sim *s, float value);
int flipvh_set_viscosity_grid(flipvh_sim *s, const float *nodes);
void flipvh_set_gravity(flipvh_sim *s, float gx, float gy, float gz);
size_t flipvh_num_particles(flipvh_sim *s);
void flipvh_get_particles(flipvh_sim *s, float *aos6);
void flipvh_set_particles(flipvh_sim *s, const float *aos6, size_t n);
void flipvh_get_solid_sdf(flipvh_sim *s, float *nodes);                   /* (I+1)(J+1)(K+1) */
void flipvh_get_dims(flipvh_sim *s, int *ijk3);
int flipvh_save_state(flipvh_sim *s, const char *path);                    /* FluidSimulation::saveState: 0 ok */
int flipvh_load_state(flipvh_sim *s, const char *path);                    /* FluidSimulation::loadState: 0 ok */
int flipvh_advance(flipvh_sim *s, float dt, flipv_stats *stats);          /* FluidSimulation::advance; needs a GPU */
flipv_context *flipvh_context(flipvh_sim *s);                             /* the underlying C-ABI context; needs a GPU */

/* MeshLevelSet::calculateSignedDistanceField on its own (reference meshlevelset.cpp:138-150) */
void flipvh_mesh_sdf(int isize, int jsize, int ksize, float dx, const float *verts, int nv, const int *tris, int nt,
                     int band, float *phi_nodes, int *closest_nodes);
/* TriangleMesh::loadPLY; returns 0 on success and the counts; call twice (first with NULL buffers) */
int flipvh_load_ply(const char *path, int *nv, int *nt, float *verts, int *tris);

#ifdef __cplusplus
}
#endif
#endif
