// flipv_comm.h -- communication layer of the slab decomposition (the reference is single-process,
// there is nothing to translate).
//
// Decomposition: slabs along k, the slowest axis of the Array3d layout, so a halo is a whole number of
// contiguous PX*PY planes.  Every rank indexes the GLOBAL grid (its arrays span the whole index space, only its
// own planes [k0,k1) plus a few halo planes ever hold data), which keeps every kernel identical to the single-GPU
// one; 288 GB per GPU makes the replicated allocation affordable (8 x 256^3 stacked: ~50 GB per rank).
//
// Three exchange patterns, all enqueued on the context's stream:
//   halo copy    owner -> neighbour copies of the H boundary planes of a set of arrays      (s before every SpMV,
//                velocities/valid masks per extrapolation layer, phi, pressure, ...)
//   halo reduce  neighbour -> owner contributions on planes a rank scattered into but does not own, combined
//                with min (particle SDF) or + (P2G accumulators)
//   all-reduce   the PCG scalars.  Ranks accumulate into DISJOINT slots of the per-iteration slot block, so one
//                sum all-reduce merges sums and maxima alike: ONE small all-reduce per PCG iteration (pcg_common.h)
//   migration    particles that left the slab go to the neighbour owning their cell (counts first, then records)
//
// Backends: RCCL (one process per GPU, ncclSend/ncclRecv grouped per exchange, ncclAllReduce; librccl is
// dlopen'ed on first use so single-GPU runs never load it) and an in-process backend (N contexts in one process on
// one device, one host thread per rank, rendezvous through host memory) that exists so the decomposition can be
// verified against the single-domain result on a one-GPU box.
#pragma once
#include "flipv_internal.h"

struct Comm {
    int rank = 0, nranks = 1;
    virtual ~Comm() {}
    virtual int begin(flipv_context *c) = 0;   // open a group of point-to-point operations
    // exchange with rank `peer` (rank-1 or rank+1): device buffers; either side may be empty (bytes == 0)
    virtual int sendrecv(flipv_context *c, int peer, const void *sendbuf, size_t sbytes, void *recvbuf, size_t rbytes) = 0;
    virtual int end(flipv_context *c) = 0;     // close the group: after it returns the operations are enqueued
    virtual int allreduce_sum(flipv_context *c, double *dev, size_t n) = 0;
    virtual int allreduce_sum_f32(flipv_context *c, float *dev, size_t n) = 0;   // same result on every rank (the viscosity multigrid's global levels rely on it)
    virtual int barrier(flipv_context *c) = 0;
};

// halo helpers (flipv_comm.hip)
// lay: 0 = the plain layout (gidx), 1 = the brick layout of the viscosity solver's arrays (bidx over c->LB).  What travels is box-shaped
// staging either way, so the two ends of an exchange need not use the same layout.
struct HaloArray { void *p; size_t elem; int lay = 0; };
int fv_halo_copy(flipv_context *c, const HaloArray *arr, int n, int H);
int fv_halo_copy_begin(flipv_context *c, const HaloArray *arr, int n, int H);  // on the communication stream, overlapping c->stream
int fv_halo_wait(flipv_context *c);                                             // c->stream waits for that exchange
enum { HALO_MIN_F32 = 0, HALO_ADD_F32 = 1 };
int fv_halo_reduce(flipv_context *c, float *const *arr, int n, int Hlo, int Hhi, int op);
// The same two exchanges for fp32 arrays of a DISTRIBUTED coarse level of the viscosity multigrid (cidx layout over the level's global Lay LC): this rank owns
// the level's indices [olo, ohi) -- the boxes of the ranks tile the level like their blocks tile the domain --, H entries travel.  add = 0: owner -> neighbours
// (copy into their halo entries); add = 1: neighbours -> owner (what they accumulated in their halo entries is added to the owner's).  n <= 6 arrays per call.
int fv_halo_level(flipv_context *c, const Lay &LC, const int olo[3], const int ohi[3], float *const *arr, int n, int H, int add);
int fv_allreduce_scalars(flipv_context *c, double *dev, size_t n);
int fv_allreduce_f32(flipv_context *c, float *dev, size_t n);    // in-place sum over the ranks, on c->stream
int fv_migrate_particles(flipv_context *c);
int fv_allreduce_max_f32(flipv_context *c, float *value);  // host value in/out (synchronises)
// every rank's n <= FV_GATHER_MAX host values on every rank, all[r * n + q] = rank r's q-th value (one small all-reduce + one synchronisation,
// whatever n: the per-solve decisions that must come out alike on all ranks are taken from ONE such exchange, not from one round trip per value).
// Without a communicator: all = mine.
constexpr int FV_GATHER_MAX = 8;
int fv_allgather_f64(flipv_context *c, const double *mine, int n, double *all);
// thinnest slab a multi-rank run accepts: the widest exchange moves ceil(cfl) + 3 of a rank's own planes
static inline int fv_min_slab_planes(float cfl_number) { return (int)ceilf(cfl_number) + 3; }
// FLIPV_ERR_INVALID (with c->err set) if the block is thinner than that along an axis on which it has neighbours
int fv_check_block_thickness(flipv_context *c, float cfl_number, const char *who);
