// flipv_comm.hip -- halo exchange, PCG scalar all-reduce and particle migration of the slab decomposition
// (see flipv_comm.h), with an RCCL backend and an in-process verification backend.
#include "flipv_comm.h"

#include <dlfcn.h>

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

// particle -> destination rank by the k index of its cell: 0 stay, 1 previous rank, 2 next rank.
// Output slots come from ONE global atomic per block and destination (nearly every particle stays: one atomic per
// particle on the same address serialised 4.7 M operations into 58 ms).  Order inside a destination follows the order in which the blocks arrive: it is
// arbitrary, as it is after any migration.
__global__ __launch_bounds__(256) void k_migrate_classify(const float *__restrict__ aos6, size_t n, float *__restrict__ stay,
                                                          float *__restrict__ toPrev, float *__restrict__ toNext,
                                                          unsigned long long *__restrict__ counts, float dx, int kc0, int kc1) {
    __shared__ unsigned wcount[4][3];            // per wave, per destination
    __shared__ unsigned long long bbase[3];      // this block's first slot per destination
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float q[6] = {0, 0, 0, 0, 0, 0};
    int dest = -1;
    if (p < n) {
#pragma unroll
        for (int e = 0; e < 6; e++) q[e] = aos6[6 * p + e];
        const int k = (int)floor((double)q[2] * (1.0 / (double)dx));
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

// ================================================================================================ halo helpers
int fv_halo_copy(flipv_context *c, const HaloArray *arr, int n, int H) {
    Comm *cm = c->comm;
    if (!cm || H <= 0) return FLIPV_OK;
    const Lay &L = c->L;
    const size_t plane = (size_t)L.sz;
    int rc = cm->begin(c);
    if (rc) return rc;
    for (int a = 0; a < n; a++) {
        char *base = (char *)arr[a].p;
        const size_t pb = plane * arr[a].elem;
        if (cm->rank > 0) {  // exchange with the previous rank: send my first H planes, receive its last H planes
            const int hs = c->k0 + H <= c->k1 ? H : c->k1 - c->k0, hr = c->k0 - H >= 0 ? H : c->k0;
            rc = cm->sendrecv(c, cm->rank - 1, base + (size_t)c->k0 * pb, (size_t)hs * pb, base + (size_t)(c->k0 - hr) * pb, (size_t)hr * pb);
            if (rc) return rc;
        }
        if (cm->rank < cm->nranks - 1) {
            const int hs = c->k1 - H >= c->k0 ? H : c->k1 - c->k0, hr = c->k1 + H <= L.PZ ? H : L.PZ - c->k1;
            rc = cm->sendrecv(c, cm->rank + 1, base + (size_t)(c->k1 - hs) * pb, (size_t)hs * pb, base + (size_t)c->k1 * pb, (size_t)hr * pb);
            if (rc) return rc;
        }
    }
    return cm->end(c);
}

// The same exchange on the communication stream: it starts once everything enqueued on c->stream so far has finished and
// runs beside whatever c->stream is given next; fv_halo_wait() makes c->stream wait for it.
int fv_halo_copy_begin(flipv_context *c, const HaloArray *arr, int n, int H) {
    if (!c->comm || H <= 0) return FLIPV_OK;
    if (!c->commOverlap) return fv_halo_copy(c, arr, n, H);
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
    if (!c->comm || !c->commOverlap) return FLIPV_OK;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->evHalo, 0));
    return FLIPV_OK;
}

int fv_halo_reduce(flipv_context *c, float *const *arr, int n, int Hlo, int Hhi, int op) {
    Comm *cm = c->comm;
    if (!cm) return FLIPV_OK;
    const Lay &L = c->L;
    const size_t plane = (size_t)L.sz;
    // staging: per array [Hhi planes from prev | Hlo planes from next]
    const size_t need = (size_t)n * (size_t)(Hlo + Hhi) * plane;
    if (need > c->haloCap) {
        if (c->haloBuf) (void)hipFree(c->haloBuf);
        c->haloBuf = nullptr; c->haloCap = 0;
        HIPCHK(c, hipMalloc((void **)&c->haloBuf, need * sizeof(float)));
        c->haloCap = need;
    }
    int rc = cm->begin(c);
    if (rc) return rc;
    for (int a = 0; a < n; a++) {
        float *base = arr[a];
        float *stPrev = c->haloBuf + (size_t)a * (Hlo + Hhi) * plane, *stNext = stPrev + (size_t)Hhi * plane;
        if (cm->rank > 0) {
            // I scattered into planes [k0-Hlo, k0) owned by prev; prev scattered into my planes [k0, k0+Hhi)
            rc = cm->sendrecv(c, cm->rank - 1, base + (size_t)(c->k0 - Hlo) * plane, (size_t)Hlo * plane * 4, stPrev, (size_t)Hhi * plane * 4);
            if (rc) return rc;
        }
        if (cm->rank < cm->nranks - 1) {
            rc = cm->sendrecv(c, cm->rank + 1, base + (size_t)c->k1 * plane, (size_t)Hhi * plane * 4, stNext, (size_t)Hlo * plane * 4);
            if (rc) return rc;
        }
    }
    rc = cm->end(c);
    if (rc) return rc;
    for (int a = 0; a < n; a++) {
        float *base = arr[a];
        float *stPrev = c->haloBuf + (size_t)a * (Hlo + Hhi) * plane, *stNext = stPrev + (size_t)Hhi * plane;
        if (cm->rank > 0)
            hipLaunchKernelGGL(k_halo_combine, dim3(grid1d((size_t)Hhi * plane)), dim3(256), 0, c->stream, base + (size_t)c->k0 * plane,
                               stPrev, (size_t)Hhi * plane, op);
        if (cm->rank < cm->nranks - 1)
            hipLaunchKernelGGL(k_halo_combine, dim3(grid1d((size_t)Hlo * plane)), dim3(256), 0, c->stream,
                               base + (size_t)(c->k1 - Hlo) * plane, stNext, (size_t)Hlo * plane, op);
    }
    HIPCHK(c, hipGetLastError());
    return FLIPV_OK;
}

int fv_allreduce_scalars(flipv_context *c, double *dev, size_t n) {
    Comm *cm = c->comm;
    if (!cm) return FLIPV_OK;
    return cm->allreduce_sum(c, dev, n);
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double m = 0;
    for (double v : h) m = v > m ? v : m;
    *value = (float)m;
    return FLIPV_OK;
}

// particles whose cell left the slab move to the neighbour that owns it (a particle moves at most CFL = 5 cells per
// substep, slabs are much thicker, so only the two neighbours can be destinations)
int fv_migrate_particles(flipv_context *c) {
    Comm *cm = c->comm;
    if (!cm) return FLIPV_OK;
    const Lay &L = c->L;
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
    const int kc0 = c->k0, kc1 = cm->rank == cm->nranks - 1 ? L.K + 1000000 : c->k1;
    const int kc0e = cm->rank == 0 ? -1000000 : kc0;
    if (np)
        hipLaunchKernelGGL(k_migrate_classify, dim3(cdiv(np, 256)), dim3(256), 0, c->stream, c->particles, np, stay, toPrev, toNext,
                           cnt, c->dx, kc0e, kc1);
    int rc = cm->begin(c);
    if (rc) return rc;
    if (cm->rank > 0) { rc = cm->sendrecv(c, cm->rank - 1, cnt + 1, 8, cnt + 4, 8); if (rc) return rc; }
    if (cm->rank < cm->nranks - 1) { rc = cm->sendrecv(c, cm->rank + 1, cnt + 2, 8, cnt + 5, 8); if (rc) return rc; }
    rc = cm->end(c);
    if (rc) return rc;
    unsigned long long h[8];
    HIPCHK(c, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    if (cm->rank > 0) { rc = cm->sendrecv(c, cm->rank - 1, toPrev, nToPrev * 24, c->particles + nStay * 6, nFromPrev * 24); if (rc) return rc; }
    if (cm->rank < cm->nranks - 1) {
        rc = cm->sendrecv(c, cm->rank + 1, toNext, nToNext * 24, c->particles + (nStay + nFromPrev) * 6, nFromNext * 24);
        if (rc) return rc;
    }
    rc = cm->end(c);
    if (rc) return rc;
    c->np = nNew;
    c->binsValid = 0;
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

// RCCL enum values (rccl.h): ncclInt8/ncclChar = 0, ncclFloat64/ncclDouble = 8, ncclSum = 0
constexpr int NCCL_CHAR = 0, NCCL_DOUBLE = 8, NCCL_SUM = 0;

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
    int barrier(flipv_context *c) override {
        int rc = allreduce_sum(c, c->d_scal_small + 32, 1);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
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
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long long g = generation;
        if (++arrived == n) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

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
        g->wait();
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
        g->wait();  // everybody has read: send buffers may be reused
        return rc;
    }
    int allreduce_sum(flipv_context *c, double *dev, size_t n) override {
        std::vector<double> &h = g->red[rank];
        h.resize(n);
        HIPCHK(c, hipMemcpyAsync(h.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        g->wait();
        std::vector<double> sum(n, 0.0);
        for (int r = 0; r < g->n; r++)
            for (size_t t = 0; t < n; t++) sum[t] += g->red[r][t];
        g->wait();  // everybody has read every contribution
        HIPCHK(c, hipMemcpyAsync(dev, sum.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return FLIPV_OK;
    }
    int barrier(flipv_context *c) override {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        g->wait();
        return FLIPV_OK;
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
    const int need = fv_min_slab_planes(c->prm.cfl_number);
    if (nranks > 1 && c->k1 - c->k0 < need) {
        c->err = "flipv_comm_init: slab [" + std::to_string(c->k0) + ", " + std::to_string(c->k1) + ") is thinner than the widest halo (" +
                 std::to_string(need) + " planes = ceil(cfl_number) + 3)";
        return FLIPV_ERR_INVALID;
    }
    return FLIPV_OK;
}

extern "C" int flipv_comm_init_rccl(flipv_context *c, const void *unique_id, int rank, int nranks) {
    if (!c || !unique_id || rank < 0 || rank >= nranks) return FLIPV_ERR_INVALID;
    if (c->comm) { c->err = "flipv_comm_init: communicator already set"; return FLIPV_ERR_INVALID; }
    { const int rc = comm_check(c, nranks); if (rc) return rc; }
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
    if (!ctxs || n < 1) return FLIPV_ERR_INVALID;
    for (int r = 0; r < n; r++) if (!ctxs[r] || ctxs[r]->comm) return FLIPV_ERR_INVALID;
    for (int r = 0; r < n; r++) { const int rc = comm_check(ctxs[r], n); if (rc) return rc; }
    LocalGroup *g = new LocalGroup();
    g->n = n; g->refs = n;
    g->ops.resize((size_t)n);
    g->red.resize((size_t)n);
    for (int r = 0; r < n; r++) {
        LocalComm *cm = new LocalComm();
        cm->rank = r; cm->nranks = n; cm->g = g;
        ctxs[r]->comm = cm;
    }
    return FLIPV_OK;
}

extern "C" int flipv_comm_finalize(flipv_context *c) {
    if (!c) return FLIPV_ERR_INVALID;
    if (c->comm) { delete c->comm; c->comm = nullptr; }
    return FLIPV_OK;
}
