// flipv_internal.h -- shared host/device declarations of libflipv (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flipv.h"

// ---------------------------------------------------------------------------------------------
// Device layout.
//
// At the ABI every grid is the reference's Array3d (x fastest, its own width/height).  On the device all
// lattices of one simulation -- cells (I,J,K), U/V/W faces, nodes (I+1,J+1,K+1) and the three edge
// families -- live in ONE index space of roundup8(I+1) x roundup4(J+1) x (K+1) entries, of which a context ALLOCATES a box:
// the whole space on a single GPU; the block of indices a rank owns plus FV_HALO entries on every side that has a
// neighbour (rounded out to multiples of 8 in i and 4 in j) in a block-decomposed run.  With (ox,oy,oz) the first
// allocated index and PX x PY x PZ the allocated extents
//     g(i,j,k) = (i-ox) + PX*((j-oy) + PY*(k-oz))      for every array,  i, j, k always GLOBAL indices,
// so that every kernel states the domain's boundary conditions (i == 0, i == I, ...) the same way on every rank.
// Consequences: a single index and a single set of neighbour offsets (1, PX, PX*PY) address every
// field; rows start 16-byte aligned so a lane can move 4 consecutive i with one dwordx4 access (a wave:
// 1 KiB per instruction); a k-plane is one contiguous PX*PY block (the halo unit of a slab decomposition).
// Entries outside a lattice's logical range are zero and never written.  Each allocation carries a guard
// zone of one plane + one row on both sides so that the +-1 neighbours of any in-range index are valid
// addresses.  Conversion to/from Array3d happens in flipv_read_grid / flipv_write_grid (k_pack/k_unpack).
// ---------------------------------------------------------------------------------------------
struct Lay {
    int I, J, K;     // cells of the GLOBAL grid
    int PX, PY, PZ;  // extents of the allocated box (8 | PX, 4 | PY)
    long sy, sz;     // strides of j and k
    size_t n;        // PX*PY*PZ
    size_t guard;    // floats of guard zone in front of / behind every array
    int ox, oy, oz;  // global index of the first allocated entry (8 | ox, 4 | oy; 0 on a single GPU)
    int ib, ie, jb, je, kb, ke;   // box of indices a pointwise launch covers (GRID3 / IJK_OR_RETURN), half-open
    int olo[3], ohi[3];           // box of indices this rank OWNS, half-open (the whole allocated box on a single GPU; on the last
                                  // rank of an axis it runs to the padded end, i.e. includes the closing face / node plane)
};

__host__ __device__ __forceinline__ size_t gidx(const Lay &L, int i, int j, int k) {
    return (size_t)(i - L.ox) + (size_t)L.PX * ((size_t)(j - L.oy) + (size_t)L.PY * (size_t)(k - L.oz));
}
__host__ __device__ __forceinline__ bool d_owned(const Lay &L, int i, int j, int k) {
    return i >= L.olo[0] && i < L.ohi[0] && j >= L.olo[1] && j < L.ohi[1] && k >= L.olo[2] && k < L.ohi[2];
}
// first entry of allocated plane k (whole planes are contiguous: copies, fills and reductions over plane ranges)
__host__ __device__ __forceinline__ size_t plane_off(const Lay &L, int k) { return (size_t)(k - L.oz) * (size_t)L.sz; }
// Swizzled plane layout (x, r, q and the diagonal of the viscosity PCG, 16-lane tile geometry): the four rows of an aligned row
// group are interleaved in pieces of 8 indices, so that a 128-byte line holds an 8 x 4 patch of a k-plane instead of a
// 32 x 1 stick.  On a compact liquid body the sticks are 1.50x over-fetched (both ends of every i-run), the patches 1.20x
// (counted on the 256^3 bunny); a wave of 16 x 4 lanes still moves one contiguous 1 KB per access.  PX % 8 == 0 and
// PY % 4 == 0 make it a bijection of each plane; j = -1 aliases the last row of the previous plane as gidx does.
__host__ __device__ __forceinline__ size_t sidx(const Lay &L, int i, int j, int k) {
    i -= L.ox; j -= L.oy;   // 8 | ox and 4 | oy: patches of the box are patches of the global index space
    return (size_t)((long)(k - L.oz) * L.sz + (long)(j >> 2) * (4 * L.PX) + (long)(i >> 3) * 32 + (long)(((j & 3) << 3) + (i & 7)));
}

// Brick layout of the viscosity solver's arrays on sparse liquids (k_viscosity_brick.hip).  The context's ALLOCATED box (the whole index
// space on a single GPU; the rank's owned box + halo on a block context: 8 | ox, 4 | oy, oz arbitrary) is cut
// into bricks of 8 x 4 x 2 indices = 64 entries = 256 contiguous bytes, each brick two 128-byte lines of 4 x 4 x 2 indices side by side
// in i, bricks in x-fastest order with ONE padding brick on every side (box-local indices -8.. and up to the padded end + 7 are addressable:
// stencil neighbours of any in-range index need no guard zone and no bounds test).  Indices are GLOBAL, as everywhere: bidx subtracts the box origin.  On the reference's scenes the liquid fills a
// few per cent of the box as a compact body: a 128-byte line that is a 32 x 1 stick of a k-plane (the plain layout) is 1.5x
// over-fetched at the ends of the liquid's i-runs, a 4 x 4 x 2 block 1.1x (counted on the 256^3 bunny).  A wave owns one brick, a lane
// one index; the address is SEPARABLE, bidx = f(i) + g(j) + h(k), so the offset to a neighbour along an axis depends on the lane's own
// position along that axis only (NbOff: six per-lane constants address the whole 15-point coupled stencil, diagonals by addition).
// For a Lay describing this layout sy / sz hold the BRICK strides in bricks (bricks per row, bricks per plane), as for the multigrid's
// coarse levels (k_viscosity_mg.hip: cidx); brick_lay() derives it from the context's plain Lay.
__host__ __device__ __forceinline__ size_t bidx(const Lay &B, int i, int j, int k) {
    const int ip = i - B.ox + 8, jp = j - B.oy + 4, kp = k - B.oz + 2;
    return ((size_t)((long)(kp >> 1) * B.sz + (long)(jp >> 2) * B.sy + (long)(ip >> 3)) << 6) +
           (size_t)((((ip >> 2) & 1) << 5) + ((kp & 1) << 4) + ((jp & 3) << 2) + (ip & 3));
}
static inline Lay brick_lay(const Lay &L) {   // (keeps L's origin and owned box: d_owned and the lattice tests work on it as on L)
    Lay B = L;
    const long nbx = L.PX / 8 + 2, nby = L.PY / 4 + 2, nbz = (L.PZ + 1) / 2 + 2;
    B.sy = nbx; B.sz = nbx * nby;
    B.n = (size_t)(nbx * nby * nbz) * 64;
    return B;
}
// element offsets from an index to its six axis neighbours (plain layout: +-1, +-sy, +-sz for every index; brick layout: per index)
// A pair of rows of the viscosity system whose coupling is >= 0.7 of the geometric mean of their diagonals (k_visc_pairs_find): indices of the two rows in the layout
// of r (own-index arrays) and of the multigrid's sweep vectors, their components (2 bits each), and the inverse of their 2 x 2 block.
struct VPair { unsigned ir0, ir1, iz0, iz1, comps; float i00, i01, i11; };
constexpr int FV_PAIR_CAP = 16384;
struct NbOff { int xm, xp, ym, yp, zm, zp; };
__host__ __device__ __forceinline__ NbOff nb_plain(const Lay &L) {
    NbOff o; o.xm = -1; o.xp = 1; o.ym = -(int)L.sy; o.yp = (int)L.sy; o.zm = -(int)L.sz; o.zp = (int)L.sz;
    return o;
}
__host__ __device__ __forceinline__ NbOff nb_brick(const Lay &B, int i, int j, int k) {
    const int sby = (int)B.sy * 64, sbz = (int)B.sz * 64;
    NbOff o;
    o.xp = (i & 3) < 3 ? 1 : 29;          o.xm = (i & 3) > 0 ? -1 : -29;            // into the other half-brick / the next brick: +32 - 3
    o.yp = (j & 3) < 3 ? 4 : sby - 12;    o.ym = (j & 3) > 0 ? -4 : -(sby - 12);
    const int kl = (k - B.oz) & 1;        // (i & 3, j & 3 are box-local as they are: 8 | ox, 4 | oy)
    o.zp = kl == 0 ? 16 : sbz - 16;       o.zm = kl == 1 ? -16 : -(sbz - 16);
    return o;
}
// Coarse levels of the viscosity multigrid (k_viscosity_mg.hip) are stored in bricks of 8 x 4 x 2 indices too, whole rows of 8 along i in a 128-byte line,
// over the level's GLOBAL index space (origin 0) with one brick of padding on every side; sy / sz of such a Lay are brick strides.  (Here because the
// halo exchange of a DISTRIBUTED coarse level addresses it: flipv_comm.h: fv_halo_level.)
__host__ __device__ __forceinline__ size_t cidx(const Lay &L, int i, int j, int k) {
    const int ip = i + 8, jp = j + 4, kp = k + 2;
    return ((size_t)((long)(kp >> 1) * L.sz + (long)(jp >> 2) * L.sy + (long)(ip >> 3)) << 6) + (size_t)(((kp & 1) << 5) + ((jp & 3) << 3) + (ip & 7));
}
// layout of the viscosity solver's arrays for the current solve
enum { VLAYOUT_PLAIN = 0, VLAYOUT_SWZ = 1, VLAYOUT_BRICK = 2 };

// lattice ids: logical extents inside the shared index space
enum { LAT_CELL = 0, LAT_U = 1, LAT_V = 2, LAT_W = 3, LAT_NODE = 4, LAT_EU = 5, LAT_EV = 6, LAT_EW = 7 };
__host__ __device__ __forceinline__ void lat_dims(const Lay &L, int lat, int &w, int &h, int &d) {
    w = L.I + (lat == LAT_U || lat == LAT_NODE || lat == LAT_EV || lat == LAT_EW);
    h = L.J + (lat == LAT_V || lat == LAT_NODE || lat == LAT_EU || lat == LAT_EW);
    d = L.K + (lat == LAT_W || lat == LAT_NODE || lat == LAT_EU || lat == LAT_EV);
}

// Solver tiles: (ROWL N) x TY x 1 indices, one 256-thread block (64, 4, 1) per tile; see pcg_geo.inc.  ROWL (16 or 64 lanes
// of a wave along i) is chosen per solve from how full the tiles are; TY = 4 * 64 / ROWL.
constexpr int VW_P = 4;          // 7-point stencil, 72 VGPRs: 16-byte accesses
constexpr int VW_V = 2;          // narrowest lane width of the viscosity kernels (2 or 4 is chosen per solve)
static inline int geo_ty(int rowl) { return 4 * (64 / rowl); }

struct TileGrid {
    int ntx, nty, ntz;
    int rowl;   // 16 or 64: which geometry the grid (and the tile list built on it) belongs to
    int ox, oy, oz;   // global index of tile (0,0,0)'s first entry = the first index the rank owns
    __host__ __device__ int count() const { return ntx * nty * ntz; }
};
// tiles cover the indices the rank owns (the whole index space on a single GPU)
static inline TileGrid make_tile_grid(const Lay &L, int rowl, int vw) {
    TileGrid tg;
    tg.rowl = rowl;
    tg.ox = L.olo[0]; tg.oy = L.olo[1]; tg.oz = L.olo[2];
    tg.ntx = (L.ohi[0] - L.olo[0] + rowl * vw - 1) / (rowl * vw);
    tg.nty = (L.ohi[1] - L.olo[1] + geo_ty(rowl) - 1) / geo_ty(rowl);
    tg.ntz = L.ohi[2] - L.olo[2];
    return tg;
}

struct Comm;
// partial-sum slots per reduced PCG scalar (pcg_common.h); ranks of a communicator own disjoint slot ranges, so this is
// also the largest rank count a communicator accepts
constexpr int NSLOT = 32;

// widest halo any exchange uses: the velocity halo of particle advection, ceil(cfl_number) + 3 planes at the default CFL
// number of 5.  A block context allocates this many entries around its owned box; flipv_set_params rejects a larger CFL
// number on such a context.
constexpr int FV_HALO = 8;

struct FvParams : flipv_params, flipv_debug_params {};

struct flipv_context {
    Lay L;           // launch box = the whole allocated box; per-launch ranges come from fv_range()
    // Block decomposition: this rank owns cells [cell0, cell1) per axis, i.e. indices [L.olo, L.ohi) (the closing face / node
    // plane belongs to the last rank of the axis).  Single GPU: everything.  k0/k1 = L.olo[2]/L.ohi[2], kept under their
    // old names where only the k-range matters.
    int cell0[3], cell1[3];
    int k0, k1;
    int setupOnly;   // created by flipv_create_setup: only the arrays the scene-setup entry points touch exist (solid SDF, viscosity, liquid phi,
                     // particles, staging); every substep entry point refuses it
    // Where the liquid is.  Inside flipv_substep / flipv_advance the sweeps whose result is trivial away from the liquid (liquid SDF
    // reset, P2G finalisation, extrapolation set-up, body force, the set-up kernels of both solves, the pressure gradient, the
    // velocity copies) cover only the box of the particle bins that hold particles, widened by LIQ_MARGIN cells, united with the
    // previous substep's box: what lay outside both was written with its trivial value (phi = 3 dx, velocity 0, valid 0,
    // coefficients 0, band 0) when it was last covered and has not been touched since.  Any grid written through the ABI, or an
    // operator called on its own, resets this to "everywhere".
    int inSubstep;
    int liqValid, liqPrevValid;
    int liqLo[3], liqHi[3], liqPrevLo[3], liqPrevHi[3];   // index boxes, half-open
    int isBlock;     // created by flipv_create_block with a box smaller than the domain: scene setup entry points refuse it
    int pgrid[3], pcoord[3];   // process grid and this rank's place in it (set by flipv_comm_init_*; {1,1,1} / {0,0,0} without)
    Comm *comm;      // nullptr on a single GPU
    // communication stream state: point-to-point operations are enqueued on `xs` (normally = stream; = commStream
    // while a halo exchange overlaps interior work, see pcg_common.h), the scalar all-reduces always on `stream`
    hipStream_t xs, commStream;
    hipEvent_t evMain, evHalo;
    hipEvent_t evPoll[2];  // stop-flag read-backs of the PCG loop (two in flight)
    // small device values the host waits for (fv_read_small): mapped host block [0] = sequence word, [FV_PUB_DATA ..) = data; the device's counter; the host's expectation
    int *h_pub, *d_pubMap, *d_pubSeq;
    int pubSeq, pubUsed, pubPendingN;
    struct PubPending { void *host; int at, words; } pubPending[16];
    float dx;
    int device;
    hipStream_t stream;
    FvParams prm;   // flipv_params + flipv_debug_params (distinct field names: c->prm.x reaches either)
    float gravity[3];
    std::string err;
    std::vector<void *> allocs;

    // persistent grids (device layout above)
    float *U, *V, *W, *sU, *sV, *sW, *wU, *wV, *wW, *phi, *solid, *visc, *pressure;
    uint8_t *vU, *vV, *vW;
    // particles
    float *particles;  // AoS 6 floats, caller's order
    size_t np, pcap;
    float *pScratch;   // migration staging (same capacity)
    size_t pScratchCap;
    float *haloBuf;    // (unused since the box exchanges; kept for layout stability of older tools)
    size_t haloCap;
    char *xbuf = nullptr;   // staging of the packed halo boxes: send lower | send upper | receive lower | receive upper
    size_t xbufCap = 0;
    // particle bins (k_particles.hip): particle indices grouped by tile of BIN_T^3 cells, rebuilt by fv_bin_particles
    unsigned *binIdx;      // np particle indices, tile by tile
    size_t binIdxCap;
    int *binCnt, *binOff, *binCur, *binList;  // per tile: count, first entry, fill cursor; compacted list of non-empty tiles
    int *binNList;         // device: number of non-empty tiles
    int binTilesCap;
    int nbx, nby, nbz;     // tile grid of the owned slab
    int binsValid;         // bins match the current particle positions
    // P2G accumulators (value, weight) per component
    float *accU, *accV, *accW, *wgtU, *wgtV, *wgtW;
    // extrapolation stamps + activity blocks (2 x ceil(PX/8) ceil(PY/8) ceil(PZ/8) bytes)
    uint8_t *actFlags;
    int *actList = nullptr;   // active extrapolation blocks, compacted ([0] = count, ids from [1])
    uint8_t *stampU, *stampV, *stampW;
    // staging for Array3d <-> device layout conversion
    float *stage;
    size_t stageCap;
    // scalars
    double *d_scal;   // device scalar scratch (PCG)
    int geoMemoP = 0, geoMemoV = 0;   // fv_build_tiles: a tile geometry tried and turned down (pressure, viscosity)
    double tileFillP = 0.0; // ... of the pressure solve's list (launch_pressure_spmv picks its kernel by it)
    double tileFill;       // fv_build_tiles: unknowns per index of the listed tiles inside the lattices' extent (the last list built)
    double *d_scal_small;  // 64 doubles: communication scratch (counts, CFL max, barrier)
    double *d_gather = nullptr;   // NSLOT x FV_GATHER_MAX doubles: fv_allgather_f64
    double *h_scal;   // pinned host mirror
    size_t scalCap;
    int *d_flags;     // device int scratch: [0] conv, [1] tile count, [2] row count, [3] cfl bits, [4,5] graph iteration counters, [6] interior tile count, [7] in-domain indices of the listed tiles / 4
    int *h_flags;     // pinned host mirror
    int viscosity_nonzero;  // cached host-side: any viscosity node > 0 (in this rank's box)
    float viscosity_min = 1.0f;     // smallest viscosity node value in this rank's box (min != max: a variable field; k_viscosity_brick.hip: the per-row factors of the fp64 residual)
    float vFactorNow = 0.0f;       // dt / dx^2 of the running viscosity solve (viscositysolver.cpp:379-380)
    int vZeroRegion = 0;           // the viscosity field is exactly zero on part of the nodes and positive elsewhere (two correction stages: k_viscosity.hip)
    int vPerRowFactors = 0;        // this solve's fp64 residual on the reference's operator forms every row's six factors in the reference's own arithmetic
    float viscosity_max = 1.0f;     // largest viscosity node value in this rank's box; viscosity_max_any: over all ranks (the a-priori stiffness
    float viscosity_max_any = 1.0f; // estimate nu dt/dx^2 of fv_visc_auto_pick; all-reduced at the start of every viscosity solve)
    int viscosity_nonzero_any = 1;  // ... on any rank of the communicator (all-reduced at the start of every viscosity solve)
    int vForceMultigridOnce = 0;    // set while a diagonal solve AUTO picked and that ran into the cap is being repeated with the multigrid
    int vmgSweeps = 16;             // Jacobi sweeps on the multigrid's LDS-resident coarsest level for the current solve (viscosity_solve_t picks)
    int vMixed64 = 0;           // the current viscosity solve is precision = FP64 under the multigrid: fp32 Krylov loops refined to the fp64 tolerance
    double vRowsAll = 0.0;      // rows of the current viscosity system over all ranks (viscosity_solve_t's all-gather)
    unsigned long long *polishList = nullptr;   // k_visc_massless_find's list of edges (k_viscosity.hip)
    unsigned long long *polishRow = nullptr;    // k_visc_massless_polish<false>: (component << 62 | index) and value of up to four rows per listed edge, stored by k_visc_massless_write
    float *polishVal = nullptr;
    int nPolishEdges = 0;                       // edges listed before this solve (may exceed FV_POLISH_CAP: flipv_solve_info::massless_cluster_edges)
    unsigned long long *elimList = nullptr;     // k_visc_singular_find's list of faces (k_viscosity.hip)
    unsigned long long *floatList = nullptr;    // ... its list of massless rows not grounded at once, and the marks of k_visc_floating (one byte per index of the allocated box)
    uint8_t *groundMark = nullptr;
    struct VPair *pairList = nullptr;           // k_visc_pairs_find's list of strongly coupled row pairs; the first 16 bytes of the allocation hold their number (k_viscosity.hip)
    int nElim = 0;                              // rows k_visc_singular_find took out of this solve's system (flipv_solve_info::eliminated_rows)
    int nPairs = 0;                             // ... as the host read it after the set-up (0: the multigrid loop launches no pair kernel)
    long nExchanges = 0, nAllReduces = 0;   // neighbour exchanges (halo copies / reductions, one per call whatever the number of neighbours) and all-reduces issued so far (flipv_comm.hip)
    int exchIter = 0, allrIter = 0;         // ... by ONE iteration of the current solve's loop (flipv_solve_info::halo_exchanges_per_iteration, allreduces_per_iteration)
    double commBytesSetup = 0.0, commBytesIter = 0.0;   // what the current solve's multigrid all-reduces: once, and per iteration (flipv_solve_info::comm_bytes_*)
    int vmgPackedRows = 1;          // ... and whether its cycle reads the coarse rows in the packed fp16 form (k_viscosity_mg.hip: d_row_dot) or the fp32 grids
    int facValid = 0;               // the factor arrays hold the current layout's values wherever the band was

    // solver tiles
    TileGrid tgP, tgV;
    int vSwz = 0;        // the viscosity PCG arrays of the current solve are in the swizzled layout (vLayout == VLAYOUT_SWZ)
    int vLayout = 0;     // VLAYOUT_*: layout of the viscosity solver's arrays (factors, own volumes, diagonal, row mask copy, x, r, q, s, b) for the current solve
    Lay LB;              // the brick layout of this context's allocated box (brick_lay(L))
    size_t solverCap = 0; // entries every solver array holds behind its pointer: max(plain layout + guard, brick layout)
    int *brickList = nullptr, *brickFlag = nullptr;   // active bricks of the current solve (linear brick ids, x fastest), flags of the box's bricks
    size_t brickCap = 0;
    int nBricks = 0;
    uint8_t *vMaskB = nullptr;   // the row mask in the brick layout (vRowMask stays plain: the setup's "was a row" memory)
    float *vB[3] = {nullptr, nullptr, nullptr};      // right-hand side of the current solve (residual replacement recomputes r = b - A x from it)
    double *vXacc[3] = {nullptr, nullptr, nullptr};  // fp64 accumulator of the solution (group-wise update: x of the PCG loop is flushed into it at every residual replacement)
    double *vQ64[3] = {nullptr, nullptr, nullptr};   // plane layouts: A xacc in fp64 (the residual's scratch, k_viscosity.hip: fv_plane_refine); allocated on first use
    int *tileListP, *tileListV;
    int *tileFlag;
    int nActiveP, nActiveV;
    int nIntP, nIntV;  // multi-rank: the first nInt* list entries are tiles of interior planes, the rest of the slab's two boundary planes
    int vwV;         // lane width chosen for the current viscosity solve (2 or 4)
    int vPred;       // sparse liquid: the SpMV predicates its loads per lane

    // pressure system (zero outside pressure cells)
    float *pDiag, *pPi, *pPj, *pPk;
    uint8_t *pMask;  // 1 where the cell is a pressure cell
    unsigned *mlistP = nullptr, *mlistV = nullptr;   // mask words of the listed tiles in list order (256 per tile), grown on demand
    size_t mlistCapP = 0, mlistCapV = 0;             // capacity in tiles
    // k-marching work units of the two SpMV kernels (pcg_geo.inc): runs, their lane masks, the candidates' scratch
    struct Run *runsP = nullptr, *runsV = nullptr, *runCand = nullptr;
    unsigned *rmaskP = nullptr, *rmaskV = nullptr;
    size_t runCapP = 0, runCapV = 0, runCandCap = 0, rmaskCapP = 0, rmaskCapV = 0;
    int nRunsP = 0, nRunsV = 0;          // 0: the tile-at-a-time kernels run
    int runLenP = 0, runLenV = 0;
    void *pX, *pR, *pZ, *pS;  // vectors (float or double per precision)
    // viscosity system
    float *scp;                                                 // solid phi at cell centres
    float *volC, *volU, *volV, *volW, *volEU, *volEV, *volEW;  // control volumes (kept for parity reads)
    float *fC, *fEU, *fEV, *fEW;                               // factor lattices
    float *vDiagU, *vDiagV, *vDiagW;
    float *vmU, *vmV, *vmW;                                    // own volume of a row, -1 elsewhere (SpMV row mask): the exact operator (multigrid sweeps and coarse operators)
    float *vrU, *vrV, *vrW;                                    // the same plus the rounding defect of the reference's float diagonal (k_viscosity.hip: d_ref_volume): the operator the PCG solves with
    uint8_t *vRowMask;                                         // bit m set: component m has a row at this index
    uint8_t *stU, *stV, *stW;
    void *vX[3], *vR[3], *vZ[3], *vS[3];
    uint8_t *validCells, *validTmp;
    uint8_t *bandPrev;   // validCells of the previous viscosity solve
    int bandPrevValid;
    void *mgState;       // pressure multigrid hierarchy (k_pressure_mg.hip), created on first use
    void *vmgState;      // viscosity multigrid hierarchy (k_viscosity_mg.hip), created on first use
    // executables of the PCG loops' captured graphs, kept between solves: a new capture with the same topology updates the cached
    // executable in place (hipGraphExecUpdate) instead of instantiating a new one.  Slots: FV_GE_*
    hipGraphExec_t geCache[6];
    unsigned *surfList;  // indices whose control volumes need the sampling path (+ the counter at [L.n]); allocated on first use

    // kernel timing
    flipv_kernel_stats kstats;
    std::vector<hipEvent_t> evPool;
    size_t evUsed;
    struct EvSpan { size_t a, b; int which; double cells; };
    std::vector<EvSpan> evSpans;
    hipEvent_t phaseEv[FLIPV_PHASE_COUNT + 1];

    float lastDt;
    int pressureReady, viscosityReady;
    int solidVersion, weightsVersion, faceStateVersion;  // products of the solid SDF (weights; solid phi at cell centres + face states) are reused while it is unchanged
    int viscStateValid, viscStatePrec;  // k_visc_setup's off-row values are in place for this vector precision
    // what the previous viscosity solve did (flipv_params.viscosity_preconditioner = AUTO picks the next one's preconditioner from it,
    // fv_visc_auto_pick): 0 nothing yet, 1 diagonal, 2 multigrid
    long viscSolves;     // viscosity solves so far (the multigrid hierarchy is dated with it)
    int vNoMultigridOnce; // set while a failed multigrid solve is being repeated with the diagonal
    int vOperatorExact;   // 1: the viscosity SpMV applies the exact operator (vm*), 0: the reference's float-rounded one (vr*); set per solve
    int vLastPrec, vLastIts, vLastConverged;
    double vLastRelRes;
    int pressurePrec, viscosityPrec;
};

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return FLIPV_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// launch range: the owned box widened by `halo` entries on each side, clipped to the allocated box
static inline Lay fv_range(const flipv_context *c, int halo) {
    Lay L = c->L;
    const int o[3] = {L.ox, L.oy, L.oz}, P[3] = {L.PX, L.PY, L.PZ};
    int lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = L.olo[a] - halo < o[a] ? o[a] : L.olo[a] - halo;
        hi[a] = L.ohi[a] + halo > o[a] + P[a] ? o[a] + P[a] : L.ohi[a] + halo;
    }
    L.ib = lo[0]; L.ie = hi[0]; L.jb = lo[1]; L.je = hi[1]; L.kb = lo[2]; L.ke = hi[2];
    return L;
}

// 7 extrapolation layers + the particles' reach (2) + the viscosity band's two dilations and the 4^3 neighbourhood of the volume
// classification (3): nothing further from a particle than this is ever non-trivial
constexpr int LIQ_MARGIN = 12;
// fv_range clipped to where the liquid is or was one substep ago (see flipv_context::liqValid)
static inline Lay fv_range_liquid(const flipv_context *c, int halo, int site = 31) {
    Lay L = fv_range(c, halo);
    (void)site;
    if (!c->inSubstep || c->prm.no_liquid_box || !c->liqValid || !c->liqPrevValid) return L;
    int *lo[3] = {&L.ib, &L.jb, &L.kb}, *hi[3] = {&L.ie, &L.je, &L.ke};
    for (int a = 0; a < 3; a++) {
        const int l = c->liqLo[a] < c->liqPrevLo[a] ? c->liqLo[a] : c->liqPrevLo[a];
        const int h = c->liqHi[a] > c->liqPrevHi[a] ? c->liqHi[a] : c->liqPrevHi[a];
        if (l > *lo[a]) *lo[a] = l;
        if (h < *hi[a]) *hi[a] = h;
        if (*hi[a] <= *lo[a]) *hi[a] = *lo[a] + 1;   // (never empty: a launch needs a block)
    }
    return L;
}

// pointwise kernels: one thread per index of the launch box; a wave = 64 consecutive i of one row
#define GRID3(L) dim3(cdiv((L).ie - (L).ib, 64), cdiv((L).je - (L).jb, 4), (unsigned)((L).ke - (L).kb)), dim3(64, 4, 1)
#define IJK_OF_THREAD(L) \
    const int i = (L).ib + blockIdx.x * 64 + threadIdx.x, j = (L).jb + blockIdx.y * 4 + threadIdx.y, k = blockIdx.z + (L).kb
#define IJK_OR_RETURN(L)                                                                              \
    IJK_OF_THREAD(L);                                                                                  \
    if (i >= (L).ie || j >= (L).je) return;                                                            \
    const size_t c = gidx((L), i, j, k);                                                                \
    (void)c

// ---------------------------------------------------------------------------------------------
// device helpers shared by all kernels
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool d_in_range(int i, int j, int k, int w, int h, int d) {
    return i >= 0 && j >= 0 && k >= 0 && i < w && j < h && k < d;
}
// The two corners (i, j, k), (i + 1, j, k) of an interpolation cell with ONE 8-byte load where both are inside the array (global loads of
// two dwords need 4-byte alignment only): the particle kernels are bound by the number of per-lane gather instructions, not by bytes.
// A corner outside the array contributes 0, as in the reference.
struct __attribute__((packed, aligned(4))) FloatPair { float a, b; };
struct __attribute__((packed, aligned(4))) FloatQuad { float a, b, c, d; };
__device__ __forceinline__ void d_corner_pair(const float *__restrict__ g, const Lay &L, int i, int j, int k, int w, int h, int d, float &a, float &b) {
    a = 0.0f; b = 0.0f;
    if (j < 0 || j >= h || k < 0 || k >= d) return;
    const bool r0 = i >= 0 && i < w, r1 = i + 1 >= 0 && i + 1 < w;
    if (r0 && r1) { const FloatPair v = *reinterpret_cast<const FloatPair *>(g + gidx(L, i, j, k)); a = v.a; b = v.b; }
    else if (r0) a = g[gidx(L, i, j, k)];
    else if (r1) b = g[gidx(L, i + 1, j, k)];
}

// LevelsetUtils::fractionInside, 2-point (reference levelsetutils.cpp:15-27)
__device__ __forceinline__ float d_frac2(float l, float r) {
    if (l < 0 && r < 0) return 1.0f;
    if (l < 0 && r >= 0) return l / (l - r);
    if (l >= 0 && r < 0) return r / (r - l);
    return 0.0f;
}

// LevelsetUtils::fractionInside, 4-point marching squares (reference levelsetutils.cpp:38-119).
// The reference rotates a cyclic list; here the rotation is resolved to a start index s so that
// l0 = c[s], l1 = c[s+1], ... (cyclic list order bl, br, tr, tl).
__device__ __forceinline__ float d_frac4(float bl, float br, float tl, float tr) {
    float c[4] = {bl, br, tr, tl};
    int n = (bl < 0) + (br < 0) + (tl < 0) + (tr < 0);
    if (n == 4) return 1.0f;
    if (n == 0) return 0.0f;
    int s = 0;
    if (n == 3) {
        while (c[s & 3] < 0) s++;
    } else if (n == 1) {
        while (c[s & 3] >= 0) s++;
    } else {
        while (c[s & 3] >= 0 || !(c[(s + 1) & 3] < 0 || c[(s + 2) & 3] < 0)) s++;
    }
    float l0 = c[s & 3], l1 = c[(s + 1) & 3], l2 = c[(s + 2) & 3], l3 = c[(s + 3) & 3];
    if (n == 3) {
        float s0 = 1.0f - d_frac2(l0, l3), s1 = 1.0f - d_frac2(l0, l1);
        return 1.0f - 0.5f * s0 * s1;
    }
    if (n == 1) {
        return 0.5f * d_frac2(l0, l3) * d_frac2(l0, l1);
    }
    if (l1 < 0) return 0.5f * (d_frac2(l0, l3) + d_frac2(l1, l2));
    float mid = 0.25f * (l0 + l1 + l2 + l3);
    if (mid < 0) {
        float area = 0.5f * (1.0f - d_frac2(l0, l3)) * (1.0f - d_frac2(l2, l3));
        area += 0.5f * (1.0f - d_frac2(l0, l1)) * (1.0f - d_frac2(l2, l1));
        return 1.0f - area;
    }
    float area = 0.5f * d_frac2(l0, l1) * d_frac2(l0, l3);
    area += 0.5f * d_frac2(l2, l1) * d_frac2(l2, l3);
    return area;
}

// MeshLevelSet::getDistanceAtCellCenter (reference meshlevelset.cpp:66-76); same summation order
__device__ __forceinline__ float d_solid_center(const float *__restrict__ s, const Lay &L, size_t c) {
    return 0.125f * (s[c] + s[c + 1] + s[c + L.sy] + s[c + 1 + L.sy] + s[c + L.sz] + s[c + 1 + L.sz] +
                     s[c + L.sy + L.sz] + s[c + 1 + L.sy + L.sz]);
}

// Grid3d::isFaceBorderingValueU/V/W on the predicate phi<0 (reference grid3d.h:496-530).
// (i,j,k) must be a valid face index of component dir.
__device__ __forceinline__ bool d_face_borders_fluid(int dir, int i, int j, int k, const Lay &L,
                                                     const float *__restrict__ phi) {
    const int n = dir == 0 ? L.I : (dir == 1 ? L.J : L.K);
    const int cd = dir == 0 ? i : (dir == 1 ? j : k);
    const long back = dir == 0 ? 1 : (dir == 1 ? L.sy : L.sz);
    const size_t c = gidx(L, i, j, k);
    if (cd == n) return phi[c - back] < 0.0f;
    if (cd > 0) return phi[c] < 0.0f || phi[c - back] < 0.0f;
    return phi[c] < 0.0f;
}

// wave64 reductions (no LDS memory): butterfly over the 64 lanes
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// block (256 threads = 4 waves) reduction to thread 0 via one LDS slot per wave
__device__ __forceinline__ double block_sum_256(double v, double *lds4) {
    v = wave_sum(v);
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) lds4[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) r = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max_256(double v, double *lds4) {
    v = wave_max(v);
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) lds4[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) r = fmax(fmax(lds4[0], lds4[1]), fmax(lds4[2], lds4[3]));
    __syncthreads();
    return r;
}

// atomic max of a non-negative double (bit pattern is monotone for x >= 0)
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
    atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// XCD-aware tile fetch: blocks b, b+8, b+16... run on the same XCD (block b -> XCD b%8); give each XCD a
// contiguous chunk of the (spatially ordered) tile list so that j/k-neighbour tiles share an L2.
__device__ __forceinline__ int d_tile_slot(int b, int n) {
    const int per = (n + 7) >> 3;
    return (b & 7) * per + (b >> 3);  // may be >= n: caller checks
}

// ---------------------------------------------------------------------------------------------
// cross-file entry points (host side)
// ---------------------------------------------------------------------------------------------
enum { FV_GE_VISCOSITY = 0, FV_GE_PRESSURE = 1, FV_GE_PRESSURE_MG = 2, FV_GE_VISCOSITY_MG = 3, FV_GE_VISCOSITY_MID = 4, FV_GE_PRESSURE_MID = 5 };
void fv_mg_free(flipv_context *c);
void fv_vmg_free(flipv_context *c);
int fv_particle_sdf(flipv_context *c);
int fv_p2g(flipv_context *c);
int fv_extrapolate(flipv_context *c);
int fv_body_force(flipv_context *c, float dt);
int fv_compute_weights(flipv_context *c);
int fv_apply_pressure(flipv_context *c, float dt);
int fv_constrain(flipv_context *c);
int fv_cfl(flipv_context *c, float *dt_out);
int fv_update_particle_velocities(flipv_context *c);
int fv_advect_particles(flipv_context *c, float dt);
int fv_pressure_solve(flipv_context *c, float dt, flipv_solve_info *info);
int fv_viscosity_solve(flipv_context *c, float dt, flipv_solve_info *info);
// device layout <-> a box [lo, hi) of lattice `lat` in Array3d order (x fastest, box-shaped); unpack zeroes the allocated
// entries outside the lattice
int fv_pack(flipv_context *c, int lat, const float *src_f32, const uint8_t *src_u8, float *linear, const int lo[3], const int hi[3]);
int fv_unpack(flipv_context *c, int lat, const float *linear, float *dst_f32, uint8_t *dst_u8, const int lo[3], const int hi[3]);
int fv_fill(flipv_context *c, float *p, size_t n, float v);
// Several memsets as ONE launch on c->stream (or `st`).  hipMemsetAsync costs a dispatch of ~4.7 us on this device however small -- and TWO where address or size
// is not a multiple of 8 --, and a substep issued ~100 of them (flags, counters, scalar blocks, accumulators).  byte = the value every byte takes, like memset.
struct FillJob { void *p; size_t bytes; int byte; };
int fv_fill_list(flipv_context *c, const FillJob *jobs, int n, hipStream_t st = nullptr);
// Small device values the host waits for (stop flags, counts, maxima) WITHOUT a copy dispatch: a one-wave kernel stores them into mapped host memory and then bumps a
// sequence word there; the host spins on that word.  (A 4-byte hipMemcpyAsync device -> host is a blit kernel of ~12 us on this device, and hipStreamSynchronize wakes the
// host some 10 us after it; a substep issued ~55 such reads.)  words: 4-byte words; <= 6 jobs per call, <= FV_PUB_WORDS words between two waits.
constexpr int FV_PUB_REPLAY = 8, FV_PUB_DATA = 16, FV_PUB_WORDS = 1008;   // h_pub: [0] sequence | [8, 16) a replayed launch's values | [16 ..) fv_read_small's
struct ReadJob { void *host; const void *dev; int words; };
int fv_read_small(flipv_context *c, const ReadJob *jobs, int n);   // enqueue on c->stream
int fv_read_wait(flipv_context *c);                                 // everything enqueued so far has arrived and sits in the jobs' host locations
int fv_read_wait_seq(flipv_context *c, int seq);                    // the publication number `seq` (c->pubSeq at the time it was enqueued) has arrived
// the same for a launch that is CAPTURED and replayed: `words` (<= 8) words from dev to h_pub[FV_PUB_REPLAY ..) at every replay; the host calls fv_read_replayed after each
// hipGraphLaunch of that graph and then fv_read_wait; the values are then at c->h_pub + FV_PUB_REPLAY
// the stream's synchronisation point: pending small reads are waited for by spinning (and copied out), then the stream itself (returns at once when it has drained)
int fv_sync(flipv_context *c);
#define FV_SYNC(ctx) do { const int rcs_ = fv_sync(ctx); if (rcs_) return rcs_; } while (0)
#define FV_READ(ctx, host_, dev_, bytes_) do { const ReadJob rj_{(void *)(host_), (const void *)(dev_), (int)((bytes_) / 4)}; const int rcr_ = fv_read_small(ctx, &rj_, 1); if (rcr_) return rcr_; } while (0)
// (several adjacent reads as ONE publishing launch)
#define FV_READ_JOBS(ctx, ...) do { const ReadJob rjs_[] = {__VA_ARGS__}; const int rcr_ = fv_read_small(ctx, rjs_, (int)(sizeof(rjs_) / sizeof(rjs_[0]))); if (rcr_) return rcr_; } while (0)
#define FV_JOB(host_, dev_, bytes_) ReadJob{(void *)(host_), (const void *)(dev_), (int)((bytes_) / 4)}
inline int fv_read_now(flipv_context *c, void *host, const void *dev, int words) { const ReadJob j{host, dev, words}; const int rc = fv_read_small(c, &j, 1); return rc ? rc : fv_read_wait(c); }
int fv_read_capture(flipv_context *c, const void *dev, int words);
void fv_read_replayed(flipv_context *c);
int fv_fill_cells(flipv_context *c, float *p, float v, int halo);
int fv_fill_cells_liquid(flipv_context *c, float *p, float v, int halo, int site);   // the same over fv_range_liquid

// event-pool helpers for kernel timing
void fv_ev_begin(flipv_context *c, int which, double cells);
void fv_ev_end(flipv_context *c);
void fv_ev_collect(flipv_context *c);
