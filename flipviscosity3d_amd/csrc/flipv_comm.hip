// flipv_comm.hip -- halo exchange, PCG scalar all-reduce and particle migration of the slab decomposition
// (see flipv_comm.h), with an RCCL backend and an in-process verification backend.
#include "flipv_comm.h"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <mutex>

// ================================================================================================ kernels
__global__ void k_halo_combine(float *__restrict__ dst, const float *__restrict__ src, size_t n, int op) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const float a = dst[t], b = src[t];
        dst[t] = op == HALO_MIN_F32 ? fminf(a, b) : a + b;
    }
}

// Boxes of the index space <-> a contiguous staging buffer, for up to HALO_MAXARR arrays at once (array a's part of a box's
// staging starts at byte offset a * boxcount * 8, whatever its element size: keeps every part 8-byte aligned): k_halo_dirs.
constexpr int HALO_MAXARR = 6;
struct HaloSet { void *p[HALO_MAXARR]; int elem[HALO_MAXARR]; int lay[HALO_MAXARR]; int n; };
struct HBox { int lo[3], hi[3]; };
// particle -> destination along one axis by the index of its cell on that axis: 0 stay, 1 previous rank, 2 next rank.
// Output slots come from ONE global atomic per block and destination (nearly every particle stays: one atomic per
// particle on the same address serialised 4.7 M operations into 58 ms).  Order inside a destination follows the order in which the blocks arrive: it is
// arbitrary, as it is after any migration.
__global__ __launch_bounds__(256) void k_migrate_classify(const float *__restrict__ aos6, size_t n, float *__restrict__ stay,
                                                          float *__restrict__ toPrev, float *__restrict__ toNext,
                                                          unsigned long long *__restrict__ counts, float dx, int axis, int kc0, int kc1) {
    __shared__ unsigned wcount[4][3];            // per wave, per destination
    __shared__ unsigned long long bbase[3];      // this block's first slot per destination
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float q[6] = {0, 0, 0, 0, 0, 0};
    int dest = -1;
    if (p < n) {
#pragma unroll
        for (int e = 0; e < 6; e++) q[e] = aos6[6 * p + e];
        const float pos = axis == 0 ? q[0] : (axis == 1 ? q[1] : q[2]);
        const int k = (int)floor((double)pos * (1.0 / (double)dx));
        dest = k < kc0 ? 1 : (k >= kc1 ? 2 : 0);
    }
    unsigned rank = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const unsigned long long m = __ballot(dest == d);
        if (dest == d) rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wv][d] = (unsigned)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned tot = wcount[0][threadIdx.x] + wcount[1][threadIdx.x] + wcount[2][threadIdx.x] + wcount[3][threadIdx.x];
        bbase[threadIdx.x] = tot ? atomicAdd(&counts[threadIdx.x], (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    if (dest >= 0) {
        unsigned before = 0;
        for (int w = 0; w < wv; w++) before += wcount[w][dest];
        const unsigned long long slot = bbase[dest] + before + rank;
        float *out = dest == 0 ? stay : (dest == 1 ? toPrev : toNext);
#pragma unroll
        for (int e = 0; e < 6; e++) out[6 * slot + e] = q[e];
    }
}

static unsigned grid1d(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}

// ================================================================================================ process grid
static inline int nbr_rank(const flipv_context *c, int axis, int dir) {   // dir -1 / +1; -1 if there is no such neighbour
    int co[3] = {c->pcoord[0], c->pcoord[1], c->pcoord[2]};
    co[axis] += dir;
    if (co[axis] < 0 || co[axis] >= c->pgrid[axis]) return -1;
    return co[0] + c->pgrid[0] * (co[1] + c->pgrid[1] * co[2]);
}

int fv_check_block_thickness(flipv_context *c, float cfl_number, const char *who) {
    const int need = fv_min_slab_planes(cfl_number);
    for (int a = 0; a < 3; a++)
        if (c->pgrid[a] > 1 && c->cell1[a] - c->cell0[a] < need) {
            c->err = std::string(who) + ": the block is " + std::to_string(c->cell1[a] - c->cell0[a]) + " cells thick along axis " + std::to_string(a) +
                     ", thinner than the widest halo (" + std::to_string(need) + " = ceil(cfl_number) + 3)";
            return FLIPV_ERR_INVALID;
        }
    return FLIPV_OK;
}

// staging for the packed faces: [send to lower | send to upper | receive from lower | receive from upper], each `bytes` long
static int xbuf_reserve(flipv_context *c, size_t bytes) {
    const size_t need = 4 * bytes + 64;
    if (need <= c->xbufCap) return FLIPV_OK;
    if (c->xbuf) { HIPCHK(c, hipStreamSynchronize(c->xs)); FV_SYNC(c); (void)hipFree(c->xbuf); c->xbuf = nullptr; c->xbufCap = 0; }
    const size_t cap = need + need / 4;
    HIPCHK(c, hipMalloc((void **)&c->xbuf, cap));
    c->xbufCap = cap;
    return FLIPV_OK;
}

static size_t hbox_count(const HBox &b) { return (size_t)(b.hi[0] - b.lo[0]) * (size_t)(b.hi[1] - b.lo[1]) * (size_t)(b.hi[2] - b.lo[2]); }

// ---- direct exchange with the (up to 26) neighbouring blocks: ONE pack kernel, ONE group of sends / receives, ONE unpack
// kernel per exchange, whatever the decomposition -- faces, edges and corners each go straight to the rank that needs them.
// (An axis-by-axis exchange forwards edges and corners in two or three hops and costs three groups and up to twelve small
// kernels; the PCG exchanges its search direction every iteration, so the count of dependent steps is what matters.)
constexpr int HALO_MAXDIR = 26;
struct DirSet {
    int n;
    int peer[HALO_MAXDIR];
    HBox sbox[HALO_MAXDIR], rbox[HALO_MAXDIR];
    unsigned long long soff[HALO_MAXDIR], roff[HALO_MAXDIR];   // byte offsets of the direction's part of the send / receive staging
    unsigned long long scnt[HALO_MAXDIR], rcnt[HALO_MAXDIR];   // entries of the boxes
};
// blockIdx.y = direction; recv = 0: sbox -> staging + soff (pack); recv = 1: staging + roff -> rbox with `mode` (1 copy, 2 min, 3 add)
__global__ void k_halo_dirs(HaloSet hs, Lay L, Lay LB, DirSet D, char *__restrict__ staging, int recv, int mode) {   // (hs.lay: 0 gidx(L), 1 bidx(LB), 2 cidx(LB): a coarse level's Lay travels in LB's place)
    const int q = blockIdx.y;
    const HBox b = recv ? D.rbox[q] : D.sbox[q];
    const int w = b.hi[0] - b.lo[0], h = b.hi[1] - b.lo[1], d = b.hi[2] - b.lo[2];
    const size_t cnt = (size_t)w * h * d;
    char *base = staging + (recv ? D.roff[q] : D.soff[q]);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < cnt; t += stride) {
        const int i = b.lo[0] + (int)(t % (size_t)w), j = b.lo[1] + (int)((t / (size_t)w) % (size_t)h), k = b.lo[2] + (int)(t / ((size_t)w * h));
        const size_t cp = hs.lay[0] == 2 ? 0 : gidx(L, i, j, k), cb = hs.lay[0] == 2 ? cidx(LB, i, j, k) : bidx(LB, i, j, k);   // (an exchange is all-level or all-fine)
        for (int a = 0; a < hs.n; a++) {
            const size_t c = hs.lay[a] ? cb : cp;
            char *st = base + (size_t)a * cnt * 8;
            if (hs.elem[a] == 4) {
                float *g = (float *)hs.p[a] + c, *p = (float *)st + t;
                // a reduction's receive boxes overlap where faces meet (an edge cell takes contributions from up to three, a corner
                // cell from up to seven neighbours, each handled by another blockIdx.y): combine atomically
                if (!recv) *p = *g;
                else if (mode == 1) *g = *p;
                else if (mode == 2) {   // float min through integer atomics: by the SIGN BIT (-0.0f must not take the signed-int path: its pattern is INT_MIN)
                    const float v = *p;
                    if (__float_as_int(v) >= 0) atomicMin((int *)g, __float_as_int(v)); else atomicMax((unsigned *)g, __float_as_uint(v));
                }
                else atomicAdd(g, *p);
            } else if (hs.elem[a] == 8) {
                double *g = (double *)hs.p[a] + c, *p = (double *)st + t;
                if (!recv) *p = *g; else *g = *p;
            } else {
                uint8_t *g = (uint8_t *)hs.p[a] + c, *p = (uint8_t *)st + t;
                if (!recv) *p = *g; else *g = *p;
            }
        }
    }
}

// The directions of an exchange.  Along an axis with offset -1 / +1 the send range is s0 / s1 and the receive range r0 / r1 (given
// per axis as [lo, hi) pairs by the caller: a copy sends owned entries and receives halo entries, a reduction the other way
// round); with offset 0 both are the owned range.
struct AxisRanges { int sLo[2], sHi[2], rLo[2], rHi[2]; };   // [lo, hi) towards the lower / upper neighbour
static void build_dirs(const flipv_context *c, const AxisRanges R[3], int narr, DirSet *D, size_t *sendBytes, size_t *recvBytes, const int *ownLo = nullptr, const int *ownHi = nullptr) {
    Lay L = c->L;
    if (ownLo) for (int a = 0; a < 3; a++) { L.olo[a] = ownLo[a]; L.ohi[a] = ownHi[a]; }   // (a coarse level's owned box)
    D->n = 0;
    size_t so = 0, ro = 0;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int dd[3] = {dx, dy, dz};
                if (!dx && !dy && !dz) continue;
                int co[3];
                bool ok = true;
                for (int a = 0; a < 3; a++) {
                    co[a] = c->pcoord[a] + dd[a];
                    if (co[a] < 0 || co[a] >= c->pgrid[a]) ok = false;
                }
                if (!ok) continue;
                const int q = D->n++;
                D->peer[q] = co[0] + c->pgrid[0] * (co[1] + c->pgrid[1] * co[2]);
                for (int a = 0; a < 3; a++) {
                    if (dd[a] < 0) { D->sbox[q].lo[a] = R[a].sLo[0]; D->sbox[q].hi[a] = R[a].sLo[1]; D->rbox[q].lo[a] = R[a].rLo[0]; D->rbox[q].hi[a] = R[a].rLo[1]; }
                    else if (dd[a] > 0) { D->sbox[q].lo[a] = R[a].sHi[0]; D->sbox[q].hi[a] = R[a].sHi[1]; D->rbox[q].lo[a] = R[a].rHi[0]; D->rbox[q].hi[a] = R[a].rHi[1]; }
                    else { D->sbox[q].lo[a] = D->rbox[q].lo[a] = L.olo[a]; D->sbox[q].hi[a] = D->rbox[q].hi[a] = L.ohi[a]; }
                }
                D->scnt[q] = hbox_count(D->sbox[q]); D->rcnt[q] = hbox_count(D->rbox[q]);
                D->soff[q] = so; D->roff[q] = ro;
                so += D->scnt[q] * 8 * (size_t)narr;
                ro += D->rcnt[q] * 8 * (size_t)narr;
            }
    *sendBytes = so; *recvBytes = ro;
    for (int q = 0; q < D->n; q++) D->roff[q] += so;   // the receive staging follows the send staging
}
static int exchange_dirs(flipv_context *c, const HaloSet &hs, DirSet &D, size_t sendBytes, size_t recvBytes, int mode, const Lay *LC = nullptr) {
    Comm *cm = c->comm;
    int rc;
    if (D.n == 0) return FLIPV_OK;
    if ((rc = xbuf_reserve(c, (sendBytes + recvBytes + 3) / 4))) return rc;   // (xbuf_reserve sizes four parts)
    size_t maxs = 1, maxr = 1;
    for (int q = 0; q < D.n; q++) { if (D.scnt[q] > maxs) maxs = D.scnt[q]; if (D.rcnt[q] > maxr) maxr = D.rcnt[q]; }
    unsigned gs = grid1d(maxs), gr = grid1d(maxr);
    if (gs > 512) gs = 512;
    if (gr > 512) gr = 512;
    hipLaunchKernelGGL(k_halo_dirs, dim3(gs, D.n), dim3(256), 0, c->xs, hs, c->L, LC ? *LC : c->LB, D, c->xbuf, 0, 0);
    if ((rc = cm->begin(c))) return rc;
    for (int q = 0; q < D.n; q++)
        if ((rc = cm->sendrecv(c, D.peer[q], c->xbuf + D.soff[q], D.scnt[q] * 8 * (size_t)hs.n, c->xbuf + D.roff[q], D.rcnt[q] * 8 * (size_t)hs.n))) return rc;
    if ((rc = cm->end(c))) return rc;
    hipLaunchKernelGGL(k_halo_dirs, dim3(gr, D.n), dim3(256), 0, c->xs, hs, c->L, LC ? *LC : c->LB, D, c->xbuf, 1, mode);
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

// ================================================================================================ halo helpers
// owner -> neighbour copies of the H boundary entries of every array, edges and corners included (the coupled viscosity
// stencil has cross terms like V(i-1, j+1); particles sample the fields trilinearly).
int fv_halo_copy(flipv_context *c, const HaloArray *arr, int n, int H) {
    Comm *cm = c->comm;
    if (!cm || H <= 0) return FLIPV_OK;
    if (n > HALO_MAXARR) { c->err = "fv_halo_copy: too many arrays"; return FLIPV_ERR_INVALID; }
    c->nExchanges++;
    const Lay &L = c->L;
    int rc;
    bool anyBrick = false;
    for (int a = 0; a < n; a++) anyBrick = anyBrick || arr[a].lay != 0;
    if (c->pgrid[0] == 1 && c->pgrid[1] == 1 && !anyBrick) {
        // slabs along k: whole contiguous planes travel straight from / into the arrays (plain layout only: a brick array's planes are not contiguous)
        if (c->pgrid[2] <= 1) return FLIPV_OK;
        const int lower = nbr_rank(c, 2, -1), upper = nbr_rank(c, 2, +1);
        const int own0 = L.olo[2], own1 = L.ohi[2];
        const size_t plane = (size_t)L.sz;
        if ((rc = cm->begin(c))) return rc;
        for (int a = 0; a < n; a++) {
            char *base = (char *)arr[a].p;
            const size_t pb = plane * arr[a].elem;
            if (lower >= 0 && (rc = cm->sendrecv(c, lower, base + plane_off(L, own0) * arr[a].elem, (size_t)H * pb, base + plane_off(L, own0 - H) * arr[a].elem, (size_t)H * pb))) return rc;
            if (upper >= 0 && (rc = cm->sendrecv(c, upper, base + plane_off(L, own1 - H) * arr[a].elem, (size_t)H * pb, base + plane_off(L, own1) * arr[a].elem, (size_t)H * pb))) return rc;
        }
        return cm->end(c);
    }
    HaloSet hs;
    hs.n = n;
    for (int a = 0; a < n; a++) { hs.p[a] = arr[a].p; hs.elem[a] = (int)arr[a].elem; hs.lay[a] = arr[a].lay; }
    AxisRanges R[3];
    for (int a = 0; a < 3; a++) {
        const int o0 = L.olo[a], o1 = L.ohi[a];
        R[a].sLo[0] = o0; R[a].sLo[1] = o0 + H; R[a].rLo[0] = o0 - H; R[a].rLo[1] = o0;
        R[a].sHi[0] = o1 - H; R[a].sHi[1] = o1; R[a].rHi[0] = o1; R[a].rHi[1] = o1 + H;
    }
    DirSet D;
    size_t sb, rb;
    build_dirs(c, R, n, &D, &sb, &rb);
    return exchange_dirs(c, hs, D, sb, rb, 1);
}

// The same exchange on the communication stream: it starts once everything enqueued on c->stream so far has finished and
// runs beside whatever c->stream is given next; fv_halo_wait() makes c->stream wait for it.
int fv_halo_copy_begin(flipv_context *c, const HaloArray *arr, int n, int H) {
    if (!c->comm || H <= 0) return FLIPV_OK;
    if (c->prm.no_comm_overlap) return fv_halo_copy(c, arr, n, H);
    HIPCHK(c, hipEventRecord(c->evMain, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->commStream, c->evMain, 0));
    c->xs = c->commStream;
    const int rc = fv_halo_copy(c, arr, n, H);
    c->xs = c->stream;
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->evHalo, c->commStream));
    return FLIPV_OK;
}
int fv_halo_wait(flipv_context *c) {
    if (!c->comm || c->prm.no_comm_overlap) return FLIPV_OK;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->evHalo, 0));
    return FLIPV_OK;
}

// neighbour -> owner: contributions a rank scattered into entries it does not own (Hlo below its box, Hhi above) are combined
// (min or +) into the owner's entries.  Every region outside the owned box -- face, edge or corner -- belongs to exactly one
// neighbour and goes straight to it.
int fv_halo_reduce(flipv_context *c, float *const *arr, int n, int Hlo, int Hhi, int op) {
    Comm *cm = c->comm;
    if (!cm) return FLIPV_OK;
    if (n > HALO_MAXARR) { c->err = "fv_halo_reduce: too many arrays"; return FLIPV_ERR_INVALID; }
    c->nExchanges++;
    const Lay &L = c->L;
    HaloSet hs;
    hs.n = n;
    for (int a = 0; a < n; a++) { hs.p[a] = arr[a]; hs.elem[a] = 4; hs.lay[a] = 0; }
    AxisRanges R[3];
    for (int a = 0; a < 3; a++) {
        const int o0 = L.olo[a], o1 = L.ohi[a];
        // to the lower neighbour: what I scattered into [o0-Hlo, o0); from it: what it scattered into my [o0, o0+Hhi)
        R[a].sLo[0] = o0 - Hlo; R[a].sLo[1] = o0; R[a].rLo[0] = o0; R[a].rLo[1] = o0 + Hhi;
        R[a].sHi[0] = o1; R[a].sHi[1] = o1 + Hhi; R[a].rHi[0] = o1 - Hlo; R[a].rHi[1] = o1;
    }
    DirSet D;
    size_t sb, rb;
    build_dirs(c, R, n, &D, &sb, &rb);
    return exchange_dirs(c, hs, D, sb, rb, op == HALO_MIN_F32 ? 2 : 3);
}

int fv_halo_level(flipv_context *c, const Lay &LC, const int olo[3], const int ohi[3], float *const *arr, int n, int H, int add) {
    Comm *cm = c->comm;
    if (!cm || H <= 0) return FLIPV_OK;
    if (n > HALO_MAXARR) { c->err = "fv_halo_level: too many arrays"; return FLIPV_ERR_INVALID; }
    c->nExchanges++;
    HaloSet hs;
    hs.n = n;
    for (int a = 0; a < n; a++) { hs.p[a] = arr[a]; hs.elem[a] = 4; hs.lay[a] = 2; }
    AxisRanges R[3];
    for (int a = 0; a < 3; a++) {
        const int o0 = olo[a], o1 = ohi[a];
        if (!add) {   // copy: owned entries out, halo entries in
            R[a].sLo[0] = o0; R[a].sLo[1] = o0 + H; R[a].rLo[0] = o0 - H; R[a].rLo[1] = o0;
            R[a].sHi[0] = o1 - H; R[a].sHi[1] = o1; R[a].rHi[0] = o1; R[a].rHi[1] = o1 + H;
        } else {      // reduction: halo entries out, owned entries in
            R[a].sLo[0] = o0 - H; R[a].sLo[1] = o0; R[a].rLo[0] = o0; R[a].rLo[1] = o0 + H;
            R[a].sHi[0] = o1; R[a].sHi[1] = o1 + H; R[a].rHi[0] = o1 - H; R[a].rHi[1] = o1;
        }
    }
    DirSet D;
    size_t sb, rb;
    build_dirs(c, R, n, &D, &sb, &rb, olo, ohi);
    return exchange_dirs(c, hs, D, sb, rb, add ? 3 : 1, &LC);
}

int fv_allreduce_scalars(flipv_context *c, double *dev, size_t n) {
    Comm *cm = c->comm;
    if (!cm) return FLIPV_OK;
    c->nAllReduces++;
    return cm->allreduce_sum(c, dev, n);
}

int fv_allreduce_f32(flipv_context *c, float *dev, size_t n) {
    Comm *cm = c->comm;
    if (!cm || n == 0) return FLIPV_OK;
    c->nAllReduces++;
    return cm->allreduce_sum_f32(c, dev, n);
}

int fv_allreduce_max_f32(flipv_context *c, float *value) {
    Comm *cm = c->comm;
    if (!cm) return FLIPV_OK;
    // one slot per rank, merged by a sum all-reduce
    double *buf = c->d_scal_small;
    std::vector<double> h((size_t)cm->nranks, 0.0);
    h[cm->rank] = (double)*value;
    HIPCHK(c, hipMemcpyAsync(buf, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = cm->allreduce_sum(c, buf, h.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(h.data(), buf, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FV_SYNC(c);
    double m = h[0];   // (every rank has written its slot: a true maximum, negative values included)
    for (double v : h) m = v > m ? v : m;
    *value = (float)m;
    return FLIPV_OK;
}

int fv_allgather_f64(flipv_context *c, const double *mine, int n, double *all) {
    Comm *cm = c->comm;
    if (n < 1 || n > FV_GATHER_MAX) { c->err = "fv_allgather_f64: bad count"; return FLIPV_ERR_INVALID; }
    if (!cm) { for (int q = 0; q < n; q++) all[q] = mine[q]; return FLIPV_OK; }
    const size_t tot = (size_t)cm->nranks * (size_t)n;
    double h[NSLOT * FV_GATHER_MAX];
    for (size_t q = 0; q < tot; q++) h[q] = 0.0;
    for (int q = 0; q < n; q++) h[(size_t)cm->rank * n + q] = mine[q];
    HIPCHK(c, hipMemcpyAsync(c->d_gather, h, tot * sizeof(double), hipMemcpyHostToDevice, c->stream));
    FV_SYNC(c);   // (h is pageable stack memory: the copy must have left it before the all-reduce overwrites nothing of it, and before h dies)
    int rc = cm->allreduce_sum(c, c->d_gather, tot);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(all, c->d_gather, tot * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FV_SYNC(c);
    return FLIPV_OK;
}

// Particles whose cell left the block move to the rank that owns it: axis by axis (x, y, z), each phase to the two
// neighbours along that axis -- a particle that crossed an edge or a corner arrives in two or three hops.  (A particle moves
// at most CFL = 5 cells per substep and blocks are at least ceil(CFL) + 3 cells thick, so only adjacent ranks are ever
// destinations.)
static int migrate_axis(flipv_context *c, int axis) {
    Comm *cm = c->comm;
    const int lower = nbr_rank(c, axis, -1), upper = nbr_rank(c, axis, +1);
    const size_t np = c->np;
    const size_t need = 3 * (np + 1024) * 6;
    if (need > c->pScratchCap) {
        if (c->pScratch) (void)hipFree(c->pScratch);
        c->pScratch = nullptr; c->pScratchCap = 0;
        HIPCHK(c, hipMalloc((void **)&c->pScratch, need * sizeof(float)));
        c->pScratchCap = need;
    }
    float *stay = c->pScratch, *toPrev = stay + (np + 1024) * 6, *toNext = toPrev + (np + 1024) * 6;
    unsigned long long *cnt = (unsigned long long *)c->d_scal_small;  // [0..2] out counts, [4] from prev, [5] from next
    HIPCHK(c, hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), c->stream));
    // the end ranks keep whatever lies beyond the domain on their side
    const int kc0 = lower < 0 ? -1000000 : c->cell0[axis], kc1 = upper < 0 ? 1000000000 : c->cell1[axis];
    if (np)
        hipLaunchKernelGGL(k_migrate_classify, dim3(cdiv(np, 256)), dim3(256), 0, c->stream, c->particles, np, stay, toPrev, toNext,
                           cnt, c->dx, axis, kc0, kc1);
    int rc = cm->begin(c);
    if (rc) return rc;
    if (lower >= 0) { rc = cm->sendrecv(c, lower, cnt + 1, 8, cnt + 4, 8); if (rc) return rc; }
    if (upper >= 0) { rc = cm->sendrecv(c, upper, cnt + 2, 8, cnt + 5, 8); if (rc) return rc; }
    rc = cm->end(c);
    if (rc) return rc;
    unsigned long long h[8];
    HIPCHK(c, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    FV_SYNC(c);
    const size_t nStay = h[0], nToPrev = h[1], nToNext = h[2], nFromPrev = h[4], nFromNext = h[5];
    const size_t nNew = nStay + nFromPrev + nFromNext;
    if (nNew > c->pcap) {  // grow the particle store; contents are rebuilt below
        if (c->particles) (void)hipFree(c->particles);
        c->particles = nullptr; c->pcap = 0;
        const size_t cap = nNew + nNew / 8 + 1024;
        HIPCHK(c, hipMalloc((void **)&c->particles, cap * 6 * sizeof(float)));
        c->pcap = cap;
    }
    if (nStay) HIPCHK(c, hipMemcpyAsync(c->particles, stay, nStay * 24, hipMemcpyDeviceToDevice, c->stream));
    rc = cm->begin(c);
    if (rc) return rc;
    if (lower >= 0) { rc = cm->sendrecv(c, lower, toPrev, nToPrev * 24, c->particles + nStay * 6, nFromPrev * 24); if (rc) return rc; }
    if (upper >= 0) {
        rc = cm->sendrecv(c, upper, toNext, nToNext * 24, c->particles + (nStay + nFromPrev) * 6, nFromNext * 24);
        if (rc) return rc;
    }
    rc = cm->end(c);
    if (rc) return rc;
    c->np = nNew;
    c->binsValid = 0;
    return FLIPV_OK;
}

int fv_migrate_particles(flipv_context *c) {
    if (!c->comm) return FLIPV_OK;
    for (int axis = 0; axis < 3; axis++) {
        if (c->pgrid[axis] <= 1) continue;
        const int rc = migrate_axis(c, axis);
        if (rc) return rc;
    }
    return FLIPV_OK;
}

// ================================================================================================ RCCL backend
namespace {

typedef int ncclResult_t_;
typedef void *ncclComm_t_;
struct ncclUniqueId_ { char internal[128]; };
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t_ (*GetUniqueId)(ncclUniqueId_ *) = nullptr;
    ncclResult_t_ (*CommInitRank)(ncclComm_t_ *, int, ncclUniqueId_, int) = nullptr;
    ncclResult_t_ (*CommDestroy)(ncclComm_t_) = nullptr;
    ncclResult_t_ (*GroupStart)() = nullptr;
    ncclResult_t_ (*GroupEnd)() = nullptr;
    ncclResult_t_ (*Send)(const void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    ncclResult_t_ (*Recv)(void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    ncclResult_t_ (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t_) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

bool rccl_load(std::string *err) {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.lib) return true;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { *err = std::string("dlopen librccl.so: ") + dlerror(); return false; }
#define SYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(h, name); if (!g_rccl.field) { *err = std::string("dlsym ") + name; return false; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return true;
}

// RCCL enum values (rccl.h): ncclInt8/ncclChar = 0, ncclFloat32/ncclFloat = 7, ncclFloat64/ncclDouble = 8, ncclSum = 0
constexpr int NCCL_CHAR = 0, NCCL_FLOAT = 7, NCCL_DOUBLE = 8, NCCL_SUM = 0;

struct RcclComm : Comm {
    ncclComm_t_ comm = nullptr;
    ~RcclComm() override { if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm); }
    int chk(flipv_context *c, ncclResult_t_ r, const char *what) {
        if (r == 0) return FLIPV_OK;
        c->err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
        return FLIPV_ERR_COMM;
    }
    int begin(flipv_context *c) override { return chk(c, g_rccl.GroupStart(), "ncclGroupStart"); }
    int sendrecv(flipv_context *c, int peer, const void *sb, size_t sbytes, void *rb, size_t rbytes) override {
        if (sbytes) { int rc = chk(c, g_rccl.Send(sb, sbytes, NCCL_CHAR, peer, comm, c->xs), "ncclSend"); if (rc) return rc; }
        if (rbytes) { int rc = chk(c, g_rccl.Recv(rb, rbytes, NCCL_CHAR, peer, comm, c->xs), "ncclRecv"); if (rc) return rc; }
        return FLIPV_OK;
    }
    int end(flipv_context *c) override { return chk(c, g_rccl.GroupEnd(), "ncclGroupEnd"); }
    int allreduce_sum(flipv_context *c, double *dev, size_t n) override {
        return chk(c, g_rccl.AllReduce(dev, dev, n, NCCL_DOUBLE, NCCL_SUM, comm, c->stream), "ncclAllReduce");
    }
    int allreduce_sum_f32(flipv_context *c, float *dev, size_t n) override {
        return chk(c, g_rccl.AllReduce(dev, dev, n, NCCL_FLOAT, NCCL_SUM, comm, c->stream), "ncclAllReduce(float)");
    }
    int barrier(flipv_context *c) override {
        int rc = allreduce_sum(c, c->d_scal_small + 32, 1);
        if (rc) return rc;
        FV_SYNC(c);
        return FLIPV_OK;
    }
};

// ================================================================================================ in-process backend
struct LocalGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    int refs = 0;
    struct Op { int peer; const void *sb; size_t sbytes; void *rb; size_t rbytes; };
    std::vector<std::vector<Op>> ops;
    std::vector<std::vector<double>> red;
    std::vector<std::vector<float>> redf;
    bool broken = false;
    // false: a rank did not arrive within two minutes (it left its substep with an error, or took another sequence of collectives): the group is
    // marked broken and every later rendezvous fails at once -- the verification backend reports FLIPV_ERR_COMM instead of hanging the process
    bool wait() {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned long long g = generation;
        if (++arrived == n) { arrived = 0; generation++; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != g || broken; }) || broken) { broken = true; cv.notify_all(); return false; }
        return true;
    }
};
#define LOCAL_WAIT(c_)  do { if (!g->wait()) { (c_)->err = "local comm: a rank did not arrive at the rendezvous (timeout): another rank failed or took a different sequence of collectives"; return FLIPV_ERR_COMM; } } while (0)

struct LocalComm : Comm {
    LocalGroup *g = nullptr;
    std::vector<LocalGroup::Op> mine;
    ~LocalComm() override {
        bool last;
        { std::lock_guard<std::mutex> lk(g->m); last = --g->refs == 0; }
        if (last) delete g;
    }
    int begin(flipv_context *) override { mine.clear(); return FLIPV_OK; }
    int sendrecv(flipv_context *, int peer, const void *sb, size_t sbytes, void *rb, size_t rbytes) override {
        mine.push_back({peer, sb, sbytes, rb, rbytes});
        return FLIPV_OK;
    }
    int end(flipv_context *c) override {
        HIPCHK(c, hipStreamSynchronize(c->xs));  // my send buffers are complete
        g->ops[rank] = mine;
        LOCAL_WAIT(c);
        // pull: my m-th operation towards peer p matches p's m-th operation towards me
        std::vector<int> seen((size_t)g->n, 0);
        int rc = FLIPV_OK;
        for (const auto &op : mine) {
            const int m = seen[op.peer]++;
            const LocalGroup::Op *match = nullptr;
            int cnt = 0;
            for (const auto &po : g->ops[op.peer])
                if (po.peer == rank && cnt++ == m) { match = &po; break; }
            if (!match || match->sbytes != op.rbytes) { c->err = "local comm: unmatched sendrecv"; rc = FLIPV_ERR_COMM; continue; }
            if (op.rbytes && hipMemcpyAsync(op.rb, match->sb, op.rbytes, hipMemcpyDeviceToDevice, c->xs) != hipSuccess) {
                c->err = "local comm: hipMemcpyAsync failed"; rc = FLIPV_ERR_HIP;
            }
        }
        if (hipStreamSynchronize(c->xs) != hipSuccess) rc = FLIPV_ERR_HIP;
        LOCAL_WAIT(c);  // everybody has read: send buffers may be reused
        return rc;
    }
    int allreduce_sum(flipv_context *c, double *dev, size_t n) override {
        std::vector<double> &h = g->red[rank];
        h.resize(n);
        HIPCHK(c, hipMemcpyAsync(h.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FV_SYNC(c);
        LOCAL_WAIT(c);
        std::vector<double> sum(n, 0.0);
        for (int r = 0; r < g->n; r++)
            for (size_t t = 0; t < n; t++) sum[t] += g->red[r][t];
        LOCAL_WAIT(c);  // everybody has read every contribution
        HIPCHK(c, hipMemcpyAsync(dev, sum.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        FV_SYNC(c);
        return FLIPV_OK;
    }
    int allreduce_sum_f32(flipv_context *c, float *dev, size_t n) override {
        std::vector<float> &h = g->redf[rank];
        h.resize(n);
        HIPCHK(c, hipMemcpyAsync(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        FV_SYNC(c);
        LOCAL_WAIT(c);
        std::vector<float> sum(g->redf[0]);   // rank order on every rank: bitwise the same result everywhere
        for (int r = 1; r < g->n; r++) {
            const float *q = g->redf[r].data();
            for (size_t t = 0; t < n; t++) sum[t] += q[t];
        }
        LOCAL_WAIT(c);  // everybody has read every contribution
        HIPCHK(c, hipMemcpyAsync(dev, sum.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        FV_SYNC(c);
        return FLIPV_OK;
    }
    int barrier(flipv_context *c) override {
        FV_SYNC(c);
        LOCAL_WAIT(c);
        return FLIPV_OK;
    }
};

// ================================================================================================ host-callback backend
// Every exchange is staged through pinned host memory and handed to callbacks of the embedding program (flipv_host_comm, include/flipv.h): one process per rank like the
// RCCL backend -- the same launcher, rendezvous, per-process block contexts, migration and timing reduction -- but the transport is the caller's (bench.py / the tests:
// torch.distributed over gloo), so SEVERAL ranks can share ONE device.  It exists to rehearse the multi-process path on a one-GPU box; it is not a fast path.
struct HostComm : Comm {
    flipv_host_comm cb;
    struct Op { int peer; const void *sb; size_t sbytes; void *rb; size_t rbytes; size_t soff, roff; };
    std::vector<Op> ops;
    char *stage = nullptr;      // pinned: [send payloads | receive payloads] of one group, or an all-reduce's values
    size_t cap = 0;
    ~HostComm() override { if (stage) (void)hipHostFree(stage); }
    int reserve(flipv_context *c, size_t bytes) {
        if (bytes <= cap) return FLIPV_OK;
        if (stage) (void)hipHostFree(stage);
        stage = nullptr; cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        HIPCHK(c, hipHostMalloc((void **)&stage, want));
        cap = want;
        return FLIPV_OK;
    }
    int fail(flipv_context *c, const char *what) { c->err = std::string("host communicator: the ") + what + " callback reported an error"; return FLIPV_ERR_COMM; }
    int begin(flipv_context *) override { ops.clear(); return FLIPV_OK; }
    int sendrecv(flipv_context *, int peer, const void *sb, size_t sbytes, void *rb, size_t rbytes) override {
        ops.push_back({peer, sb, sbytes, rb, rbytes, 0, 0});
        return FLIPV_OK;
    }
    int end(flipv_context *c) override {
        size_t total = 0;
        for (auto &o : ops) { o.soff = total; total += (o.sbytes + 15) & ~(size_t)15; }
        for (auto &o : ops) { o.roff = total; total += (o.rbytes + 15) & ~(size_t)15; }
        int rc = reserve(c, total);
        if (rc) return rc;
        for (auto &o : ops) if (o.sbytes) HIPCHK(c, hipMemcpyAsync(stage + o.soff, o.sb, o.sbytes, hipMemcpyDeviceToHost, c->xs));
        HIPCHK(c, hipStreamSynchronize(c->xs));
        const int n = (int)ops.size();
        std::vector<int> peer((size_t)n);
        std::vector<const void *> sp((size_t)n);
        std::vector<void *> rp((size_t)n);
        std::vector<size_t> sb((size_t)n), rb((size_t)n);
        for (int m = 0; m < n; m++) { peer[m] = ops[m].peer; sp[m] = stage + ops[m].soff; rp[m] = stage + ops[m].roff; sb[m] = ops[m].sbytes; rb[m] = ops[m].rbytes; }
        if (n && cb.exchange(cb.user, n, peer.data(), sp.data(), sb.data(), rp.data(), rb.data()) != 0) return fail(c, "exchange");
        for (auto &o : ops) if (o.rbytes) HIPCHK(c, hipMemcpyAsync(o.rb, stage + o.roff, o.rbytes, hipMemcpyHostToDevice, c->xs));
        HIPCHK(c, hipStreamSynchronize(c->xs));   // (the staging area is reused by the next group)
        return FLIPV_OK;
    }
    int allreduce_sum(flipv_context *c, double *dev, size_t n) override {
        int rc = reserve(c, n * sizeof(double));
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(stage, dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FV_SYNC(c);
        if (cb.allreduce_sum_f64(cb.user, (double *)stage, n) != 0) return fail(c, "allreduce_sum_f64");
        HIPCHK(c, hipMemcpyAsync(dev, stage, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        FV_SYNC(c);
        return FLIPV_OK;
    }
    int allreduce_sum_f32(flipv_context *c, float *dev, size_t n) override {
        int rc = reserve(c, n * sizeof(float));
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(stage, dev, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        FV_SYNC(c);
        if (cb.allreduce_sum_f32(cb.user, (float *)stage, n) != 0) return fail(c, "allreduce_sum_f32");
        HIPCHK(c, hipMemcpyAsync(dev, stage, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        FV_SYNC(c);
        return FLIPV_OK;
    }
    int barrier(flipv_context *c) override {
        FV_SYNC(c);
        return cb.barrier(cb.user) == 0 ? FLIPV_OK : fail(c, "barrier");
    }
};

}  // namespace

// ================================================================================================ C-ABI
extern "C" int flipv_comm_unique_id_bytes(void) { return 128; }

extern "C" int flipv_comm_get_unique_id(void *id_out) {
    std::string err;
    if (!id_out || !rccl_load(&err)) return FLIPV_ERR_COMM;
    ncclUniqueId_ id;
    if (g_rccl.GetUniqueId(&id) != 0) return FLIPV_ERR_COMM;
    memcpy(id_out, &id, 128);
    return FLIPV_OK;
}

// Limits a communicator must respect (checked here, where the rank count becomes known):
//  * nranks <= NSLOT (32): every rank accumulates its PCG partial sums into its own slots of the NSLOT-wide slot block
//    (pcg_common.h), and fv_allreduce_max_f32 / the RCCL barrier use one double per rank of the 64-entry scratch;
//  * a slab must be at least ceil(cfl_number) + 3 planes thick: the widest exchange (the velocity halo of particle
//    advection) sends that many of its OWN planes to a neighbour that posts a receive of exactly that size, and
//    migration / halo reductions only ever talk to the two adjacent ranks.
static int comm_check(flipv_context *c, int nranks) {
    if (nranks > NSLOT) {
        c->err = "flipv_comm_init: at most " + std::to_string(NSLOT) + " ranks per communicator (got " + std::to_string(nranks) + ")";
        return FLIPV_ERR_INVALID;
    }
    return FLIPV_OK;
}
// place the context in the process grid `dims` (rank = x + dims[0] * (y + dims[1] * z)) and check that its box fits that place
static int comm_place(flipv_context *c, int rank, const int *dims) {
    const int nranks = dims[0] * dims[1] * dims[2];
    if (dims[0] < 1 || dims[1] < 1 || dims[2] < 1 || rank < 0 || rank >= nranks) { c->err = "flipv_comm_init: bad process grid"; return FLIPV_ERR_INVALID; }
    int rc = comm_check(c, nranks);
    if (rc) return rc;
    const int N[3] = {c->L.I, c->L.J, c->L.K};
    const int co[3] = {rank % dims[0], (rank / dims[0]) % dims[1], rank / (dims[0] * dims[1])};
    for (int a = 0; a < 3; a++) {
        if ((co[a] == 0) != (c->cell0[a] == 0) || (co[a] == dims[a] - 1) != (c->cell1[a] == N[a])) {
            c->err = "flipv_comm_init: the context's block does not match its place in the process grid (axis " + std::to_string(a) + ")";
            return FLIPV_ERR_INVALID;
        }
        c->pgrid[a] = dims[a]; c->pcoord[a] = co[a];
    }
    if (nranks > 1 && (rc = fv_check_block_thickness(c, c->prm.cfl_number, "flipv_comm_init"))) {
        for (int a = 0; a < 3; a++) { c->pgrid[a] = 1; c->pcoord[a] = 0; }
        return rc;
    }
    return FLIPV_OK;
}

extern "C" int flipv_comm_init_rccl(flipv_context *c, const void *unique_id, int rank, int nranks) {
    const int dims[3] = {1, 1, nranks};   // slabs along k
    return flipv_comm_init_rccl_grid(c, unique_id, rank, dims);
}

extern "C" int flipv_comm_init_rccl_grid(flipv_context *c, const void *unique_id, int rank, const int *dims) {
    if (!c || !unique_id || !dims) return FLIPV_ERR_INVALID;
    if (c->comm) { c->err = "flipv_comm_init: communicator already set"; return FLIPV_ERR_INVALID; }
    { const int rc = comm_place(c, rank, dims); if (rc) return rc; }
    const int nranks = dims[0] * dims[1] * dims[2];
    if (!rccl_load(&c->err)) return FLIPV_ERR_COMM;
    HIPCHK(c, hipSetDevice(c->device));
    RcclComm *cm = new RcclComm();
    cm->rank = rank; cm->nranks = nranks;
    ncclUniqueId_ id;
    memcpy(&id, unique_id, 128);
    ncclResult_t_ r = g_rccl.CommInitRank(&cm->comm, nranks, id, rank);
    if (r != 0) { c->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete cm; return FLIPV_ERR_COMM; }
    c->comm = cm;
    return FLIPV_OK;
}

extern "C" int flipv_comm_init_local(flipv_context **ctxs, int n) {
    const int dims[3] = {1, 1, n};   // slabs along k
    return flipv_comm_init_local_grid(ctxs, dims);
}

extern "C" int flipv_comm_init_local_grid(flipv_context **ctxs, const int *dims) {
    if (!ctxs || !dims) return FLIPV_ERR_INVALID;
    const int n = dims[0] * dims[1] * dims[2];
    if (n < 1) return FLIPV_ERR_INVALID;
    for (int r = 0; r < n; r++) if (!ctxs[r] || ctxs[r]->comm) return FLIPV_ERR_INVALID;
    for (int r = 0; r < n; r++) { const int rc = comm_place(ctxs[r], r, dims); if (rc) return rc; }
    LocalGroup *g = new LocalGroup();
    g->n = n; g->refs = n;
    g->ops.resize((size_t)n);
    g->red.resize((size_t)n);
    g->redf.resize((size_t)n);
    for (int r = 0; r < n; r++) {
        LocalComm *cm = new LocalComm();
        cm->rank = r; cm->nranks = n; cm->g = g;
        ctxs[r]->comm = cm;
    }
    return FLIPV_OK;
}


extern "C" int flipv_comm_init_host_grid(flipv_context *c, const flipv_host_comm *cb, int rank, const int *dims) {
    if (!c || !cb || !dims || !cb->exchange || !cb->allreduce_sum_f64 || !cb->allreduce_sum_f32 || !cb->barrier) return FLIPV_ERR_INVALID;
    if (c->comm) { c->err = "flipv_comm_init: communicator already set"; return FLIPV_ERR_INVALID; }
    { const int rc = comm_place(c, rank, dims); if (rc) return rc; }
    HostComm *cm = new HostComm();
    cm->rank = rank; cm->nranks = dims[0] * dims[1] * dims[2];
    cm->cb = *cb;
    c->comm = cm;
    return FLIPV_OK;
}

extern "C" int flipv_comm_finalize(flipv_context *c) {
    if (!c) return FLIPV_ERR_INVALID;
    if (c->comm) { delete c->comm; c->comm = nullptr; }
    for (int a = 0; a < 3; a++) { c->pgrid[a] = 1; c->pcoord[a] = 0; }
    return FLIPV_OK;
}
