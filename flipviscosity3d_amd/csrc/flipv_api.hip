// flipv_api.hip -- C-ABI entry points of libflipv.so (include/flipv.h): context lifetime, host<->device
// transfers (reference Array3d layout at the ABI, shared padded index space on the device), and the substep
// sequencing of FluidSimulation::advance (reference fluidsimulation.cpp:135-168).
#include "flipv_internal.h"
#include "flipv_comm.h"

#include <new>

static thread_local std::string g_create_error;

// ------------------------------------------------------------------------------------------------
// every grid array: guard zone | n entries | guard zone, all zero-initialised; returns the pointer to entry 0
template <typename T>
static int grid_alloc(flipv_context *c, T **p, size_t elem_bytes = sizeof(T), size_t min_after = 0) {
    size_t n = c->L.n;
    const size_t g = c->L.guard;
    if (min_after > n + g) n = min_after - g;   // (solver arrays: room for the brick layout behind the pointer)
    const size_t bytes = (n + 2 * g) * elem_bytes;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? FLIPV_ERR_OOM : FLIPV_ERR_HIP;
    }
    c->allocs.push_back(q);
    e = hipMemsetAsync(q, 0, bytes, c->stream);
    if (e != hipSuccess) { c->err = std::string("hipMemset: ") + hipGetErrorString(e); return FLIPV_ERR_HIP; }
    *p = (T *)((char *)q + g * elem_bytes);
    return FLIPV_OK;
}

template <typename T>
static int plain_alloc(flipv_context *c, T **p, size_t n) {
    void *q = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? FLIPV_ERR_OOM : FLIPV_ERR_HIP;
    }
    c->allocs.push_back(q);
    e = hipMemsetAsync(q, 0, bytes, c->stream);
    if (e != hipSuccess) { c->err = std::string("hipMemset: ") + hipGetErrorString(e); return FLIPV_ERR_HIP; }
    *p = (T *)q;
    return FLIPV_OK;
}

extern "C" int flipv_abi_version(void) { return FLIPV_VERSION; }
extern "C" int flipv_default_params(flipv_params *p) {
    if (!p) return FLIPV_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    p->cfl_number = 5.0f;
    p->min_frac = 0.01f;
    p->pic_ratio = 0.05f;
    p->extrapolation_layers = 0;
    p->pressure_tolerance = 1e-9;
    p->pressure_rel_tolerance = 1e-6;
    p->pressure_max_iterations = 2000;
    p->viscosity_tolerance = 1e-6;
    p->viscosity_max_iterations = 700;
    p->viscosity_accept_tolerance = 10.0;
    p->precision = FLIPV_PRECISION_FP32;
    p->check_every = 0;
    return FLIPV_OK;
}
extern "C" int flipv_default_debug_params(flipv_debug_params *p) {
    if (!p) return FLIPV_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    return FLIPV_OK;
}

extern "C" const char *flipv_last_error(flipv_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int flipv_create_slab(int I, int J, int K, float dx, int dev, int kbegin, int kend, flipv_context **out) {
    const int lo[3] = {0, 0, kbegin}, hi[3] = {I, J, kend};
    return flipv_create_block(I, J, K, dx, dev, lo, hi, out);
}

static int create_context(int I, int J, int K, float dx, int dev, const int *cell_lo, const int *cell_hi, int setupOnly, flipv_context **out);

extern "C" int flipv_create_block(int I, int J, int K, float dx, int dev, const int *cell_lo, const int *cell_hi, flipv_context **out) {
    return create_context(I, J, K, dx, dev, cell_lo, cell_hi, 0, out);
}
extern "C" int flipv_create_setup(int I, int J, int K, float dx, int dev, flipv_context **out) {
    const int lo[3] = {0, 0, 0}, hi[3] = {I, J, K};
    return create_context(I, J, K, dx, dev, lo, hi, 1, out);
}

static int create_context(int I, int J, int K, float dx, int dev, const int *cell_lo, const int *cell_hi, int setupOnly, flipv_context **out) {
    if (!out) return FLIPV_ERR_INVALID;
    *out = nullptr;
    if (I < 1 || J < 1 || K < 1 || !(dx > 0.0f)) {
        g_create_error = "flipv_create: grid dimensions must be >= 1 and dx > 0";
        return FLIPV_ERR_INVALID;
    }
    const int N[3] = {I, J, K};
    if (!cell_lo || !cell_hi) { g_create_error = "flipv_create_block: null box"; return FLIPV_ERR_INVALID; }
    for (int a = 0; a < 3; a++)
        if (cell_lo[a] < 0 || cell_hi[a] > N[a] || cell_lo[a] >= cell_hi[a]) {
            g_create_error = "flipv_create_block: need 0 <= lo < hi <= size on every axis";
            return FLIPV_ERR_INVALID;
        }
    if (cell_lo[0] % 8 != 0 || (cell_hi[0] != I && cell_hi[0] % 8 != 0)) {
        // a lane of the solver kernels owns 4 consecutive i and the swizzled layout interleaves patches of 8: a cut along i
        // must not split either
        g_create_error = "flipv_create_block: block boundaries along i must be multiples of 8";
        return FLIPV_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "flipv_create: no HIP device visible (libflipv has no CPU path)";
        return FLIPV_ERR_NO_DEVICE;
    }
    if (dev < 0 || dev >= ndev) {
        g_create_error = "flipv_create: invalid device ordinal";
        return FLIPV_ERR_INVALID;
    }
    flipv_context *c = new (std::nothrow) flipv_context();
    if (!c) return FLIPV_ERR_OOM;
    Lay &L = c->L;
    L.I = I; L.J = J; L.K = K;
    {
        // the global index space: 8 | PXg and 4 | PYg (the swizzled plane layout needs whole 8 x 4 patches)
        const int Pg[3] = {((I + 1 + 7) / 8) * 8, ((J + 1 + 3) / 4) * 4, K + 1};
        const int align[3] = {8, 4, 1};
        int o[3], e[3];
        c->isBlock = 0;
        for (int a = 0; a < 3; a++) {
            c->cell0[a] = cell_lo[a]; c->cell1[a] = cell_hi[a];
            L.olo[a] = cell_lo[a];
            L.ohi[a] = cell_hi[a] == N[a] ? Pg[a] : cell_hi[a];   // the last block of an axis also owns the closing plane (and the padding)
            o[a] = cell_lo[a] == 0 ? 0 : cell_lo[a] - FV_HALO;
            if (o[a] < 0) o[a] = 0;
            o[a] -= o[a] % align[a];
            e[a] = cell_hi[a] == N[a] ? Pg[a] : cell_hi[a] + FV_HALO;
            e[a] = (e[a] + align[a] - 1) / align[a] * align[a];
            if (e[a] > Pg[a]) e[a] = Pg[a];
            if (cell_lo[a] != 0 || cell_hi[a] != N[a]) c->isBlock = 1;
            c->pgrid[a] = 1; c->pcoord[a] = 0;
        }
        L.ox = o[0]; L.oy = o[1]; L.oz = o[2];
        L.PX = e[0] - o[0]; L.PY = e[1] - o[1]; L.PZ = e[2] - o[2];
    }
    L.sy = L.PX; L.sz = (long)L.PX * L.PY;
    L.n = (size_t)L.sz * L.PZ;
    L.guard = (((size_t)L.sz + (size_t)L.sy + 8) + 63) / 64 * 64;
    L.ib = L.ox; L.ie = L.ox + L.PX; L.jb = L.oy; L.je = L.oy + L.PY; L.kb = L.oz; L.ke = L.oz + L.PZ;
    c->k0 = L.olo[2];
    c->k1 = L.ohi[2];
    c->LB = brick_lay(L);
    c->solverCap = L.n + L.guard;
    if (c->LB.n > c->solverCap) c->solverCap = c->LB.n;
    c->comm = nullptr;
    c->pScratch = nullptr; c->pScratchCap = 0;
    c->binIdx = nullptr; c->binIdxCap = 0;
    c->surfList = nullptr;
    c->mgState = nullptr;
    c->vmgState = nullptr;
    c->binCnt = c->binOff = c->binCur = c->binList = c->binNList = nullptr;
    c->binTilesCap = 0; c->nbx = c->nby = c->nbz = 0; c->binsValid = 0;
    c->haloBuf = nullptr; c->haloCap = 0;
    c->d_scal_small = nullptr;
    c->dx = dx;
    c->device = dev;
    c->np = c->pcap = 0;
    c->particles = nullptr;
    c->stage = nullptr; c->stageCap = 0;
    c->d_scal = nullptr; c->h_scal = nullptr; c->scalCap = 0;
    c->h_flags = nullptr;
    c->evUsed = 0;
    c->pressureReady = c->viscosityReady = 0;
    c->viscStateValid = 0; c->viscStatePrec = 0;
    c->bandPrevValid = 0;
    c->solidVersion = 1; c->weightsVersion = 0; c->faceStateVersion = 0;
    c->nActiveP = c->nActiveV = 0;
    c->vwV = 2;
    c->viscosity_nonzero = 1;
    memset(&c->kstats, 0, sizeof(c->kstats));
    flipv_default_params(&c->prm);
    flipv_default_debug_params(&c->prm);
    c->gravity[0] = 0.0f; c->gravity[1] = -9.81f; c->gravity[2] = 0.0f;  // fluidsimulation.cpp:40
#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(e_); flipv_destroy(c); return FLIPV_ERR_HIP; } } while (0)
#define GALLOC(ptr) do { int rc_ = grid_alloc(c, &(ptr)); if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; } } while (0)
#define VALLOC(ptr) do { double *t_ = nullptr; int rc_ = grid_alloc(c, &t_, sizeof(double), c->solverCap); if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; } (ptr) = t_; } while (0)
#define SALLOC(ptr) do { int rc_ = grid_alloc(c, &(ptr), sizeof(*(ptr)), c->solverCap); if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; } } while (0)
    c->stream = c->xs = c->commStream = nullptr;
    c->evMain = c->evHalo = nullptr;
    c->evPoll[0] = c->evPoll[1] = nullptr;
    c->h_pub = c->d_pubMap = c->d_pubSeq = nullptr;
    c->pubSeq = c->pubUsed = c->pubPendingN = 0;
    c->nIntP = c->nIntV = 0;
    for (int q = 0; q <= FLIPV_PHASE_COUNT; q++) c->phaseEv[q] = nullptr;
    CHK(hipSetDevice(dev));
    CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&c->commStream, hipStreamNonBlocking));
    CHK(hipEventCreateWithFlags(&c->evMain, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&c->evHalo, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&c->evPoll[0], hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&c->evPoll[1], hipEventDisableTiming));
    c->xs = c->stream;
    for (int q = 0; q <= FLIPV_PHASE_COUNT; q++) CHK(hipEventCreate(&c->phaseEv[q]));
    c->setupOnly = setupOnly;
    GALLOC(c->phi); GALLOC(c->solid); GALLOC(c->visc);
    if (!setupOnly) {
    GALLOC(c->U); GALLOC(c->V); GALLOC(c->W);
    GALLOC(c->sU); GALLOC(c->sV); GALLOC(c->sW);
    GALLOC(c->wU); GALLOC(c->wV); GALLOC(c->wW);
    GALLOC(c->vU); GALLOC(c->vV); GALLOC(c->vW);
    GALLOC(c->pressure);
    GALLOC(c->accU); GALLOC(c->accV); GALLOC(c->accW);
    GALLOC(c->wgtU); GALLOC(c->wgtV); GALLOC(c->wgtW);
    GALLOC(c->stampU); GALLOC(c->stampV); GALLOC(c->stampW);
    }
    {
        int rc_ = plain_alloc(c, &c->d_flags, 16);
        if (!rc_) rc_ = plain_alloc(c, &c->d_scal_small, 64);
        if (!rc_) rc_ = plain_alloc(c, &c->d_gather, 8 * NSLOT);
        if (!rc_ && !setupOnly) rc_ = plain_alloc(c, &c->actFlags, (size_t)3 * ((L.PX + 7) / 8) * ((L.PY + 7) / 8) * ((L.PZ + 7) / 8));
        if (!rc_ && !setupOnly) rc_ = plain_alloc(c, &c->actList, (size_t)((L.PX + 7) / 8) * ((L.PY + 7) / 8) * ((L.PZ + 7) / 8) + 16);
        if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; }
    }
    CHK(hipHostMalloc((void **)&c->h_flags, 16 * sizeof(int)));
    memset(c->h_flags, 0, 16 * sizeof(int));
    CHK(hipHostMalloc((void **)&c->h_pub, (FV_PUB_DATA + FV_PUB_WORDS) * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_pub, 0, (FV_PUB_DATA + FV_PUB_WORDS) * sizeof(int));
    CHK(hipHostGetDevicePointer((void **)&c->d_pubMap, c->h_pub, 0));
    CHK(hipMalloc((void **)&c->d_pubSeq, sizeof(int)));
    CHK(hipMemset(c->d_pubSeq, 0, sizeof(int)));
    // solver tiles over the shared index space
    // the geometry is chosen per solve (fv_build_tiles); flipv_params.tile_rows pins it
    c->tgP = make_tile_grid(L, 64, VW_P);
    c->tgV = make_tile_grid(L, 64, VW_V);
    if (!setupOnly) {
        size_t ntmax = 0;   // the largest tile grid of any geometry and lane width; the virtual enumeration pads nty to a multiple of 4
        for (int rl = 16; rl <= 64; rl *= 4)
            for (int vw = 2; vw <= 4; vw *= 2) {
                const TileGrid g = make_tile_grid(L, rl, vw);
                const size_t n = (size_t)g.ntx * (size_t)((g.nty + 3) / 4 * 4) * (size_t)g.ntz + 64;
                if (n > ntmax) ntmax = n;
            }
        int rc_ = plain_alloc(c, &c->tileListP, ntmax);
        if (!rc_) rc_ = plain_alloc(c, &c->tileListV, ntmax);
        if (!rc_) rc_ = plain_alloc(c, &c->tileFlag, ntmax);
        if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; }
    }
    if (!setupOnly) {
    // pressure system
    GALLOC(c->pDiag); GALLOC(c->pPi); GALLOC(c->pPj); GALLOC(c->pPk);
    VALLOC(c->pX); VALLOC(c->pR); VALLOC(c->pZ); VALLOC(c->pS);
    // viscosity system
    GALLOC(c->scp);
    GALLOC(c->volC); GALLOC(c->volU); GALLOC(c->volV); GALLOC(c->volW);
    GALLOC(c->volEU); GALLOC(c->volEV); GALLOC(c->volEW);
    SALLOC(c->fC); SALLOC(c->fEU); SALLOC(c->fEV); SALLOC(c->fEW);
    SALLOC(c->vDiagU); SALLOC(c->vDiagV); SALLOC(c->vDiagW);
    SALLOC(c->vmU); SALLOC(c->vmV); SALLOC(c->vmW);
    SALLOC(c->vrU); SALLOC(c->vrV); SALLOC(c->vrW);
    GALLOC(c->stU); GALLOC(c->stV); GALLOC(c->stW);
    GALLOC(c->vRowMask);
    SALLOC(c->vMaskB);
    for (int q = 0; q < 3; q++) { SALLOC(c->vB[q]); SALLOC(c->vXacc[q]); }
    {
        c->brickCap = c->LB.n / 64;
        int rc_ = plain_alloc(c, &c->brickList, c->brickCap + 64);
        if (!rc_) rc_ = plain_alloc(c, &c->brickFlag, c->brickCap + 64);
        if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; }
    }
    GALLOC(c->bandPrev);
    GALLOC(c->pMask);
    GALLOC(c->validCells); GALLOC(c->validTmp);
    for (int q = 0; q < 3; q++) { VALLOC(c->vX[q]); VALLOC(c->vR[q]); VALLOC(c->vZ[q]); VALLOC(c->vS[q]); }
    }
    // staging buffer for layout conversion: one allocated box worth of floats
    {
        const size_t cap = L.n;
        int rc_ = plain_alloc(c, &c->stage, cap);
        if (rc_) { g_create_error = c->err; flipv_destroy(c); return rc_; }
        c->stageCap = cap;
    }
    CHK(hipStreamSynchronize(c->stream));
#undef CHK
#undef GALLOC
#undef VALLOC
#undef SALLOC
    // defaults of initialize(): viscosity 1.0 at every node (fluidsimulation.cpp:39), liquid phi = 3 dx
    int rc = flipv_set_viscosity_uniform(c, 1.0f);
    if (rc == FLIPV_OK) {
        rc = fv_fill_cells(c, c->phi, 3.0f * dx, 1 << 20);   // the initial liquid SDF is "far" everywhere (the whole allocated box)
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = FLIPV_ERR_HIP;
    }
    if (rc != FLIPV_OK) { g_create_error = c->err; flipv_destroy(c); return rc; }
    *out = c;
    return FLIPV_OK;
}

extern "C" int flipv_create_on_device(int I, int J, int K, float dx, int dev, flipv_context **out) {
    return flipv_create_slab(I, J, K, dx, dev, 0, K, out);
}

extern "C" int flipv_create(int I, int J, int K, float dx, flipv_context **out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return flipv_create_on_device(I, J, K, dx, dev, out);
}

extern "C" int flipv_slab_range(flipv_context *c, int *kbegin, int *kend) {
    if (!c || !kbegin || !kend) return FLIPV_ERR_INVALID;
    *kbegin = c->cell0[2];
    *kend = c->cell1[2];
    return FLIPV_OK;
}
extern "C" int flipv_block_range(flipv_context *c, int *cell_lo, int *cell_hi) {
    if (!c || !cell_lo || !cell_hi) return FLIPV_ERR_INVALID;
    for (int a = 0; a < 3; a++) { cell_lo[a] = c->cell0[a]; cell_hi[a] = c->cell1[a]; }
    return FLIPV_OK;
}

extern "C" int flipv_destroy(flipv_context *c) {
    if (!c) return FLIPV_OK;
    (void)hipSetDevice(c->device);
    if (c->commStream) (void)hipStreamSynchronize(c->commStream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { delete c->comm; c->comm = nullptr; }
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->particles) (void)hipFree(c->particles);
    if (c->pScratch) (void)hipFree(c->pScratch);
    if (c->binIdx) (void)hipFree(c->binIdx);
    if (c->surfList) (void)hipFree(c->surfList);
    if (c->mlistP) (void)hipFree(c->mlistP);
    if (c->mlistV) (void)hipFree(c->mlistV);
    if (c->runsP) (void)hipFree(c->runsP);
    if (c->runsV) (void)hipFree(c->runsV);
    if (c->runCand) (void)hipFree(c->runCand);
    if (c->rmaskP) (void)hipFree(c->rmaskP);
    if (c->rmaskV) (void)hipFree(c->rmaskV);
    fv_mg_free(c);
    fv_vmg_free(c);
    for (auto &ge : c->geCache) if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
    if (c->binCnt) (void)hipFree(c->binCnt);
    if (c->haloBuf) (void)hipFree(c->haloBuf);
    if (c->xbuf) (void)hipFree(c->xbuf);
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->h_pub) (void)hipHostFree(c->h_pub);
    if (c->d_pubSeq) (void)hipFree(c->d_pubSeq);
    if (c->polishList) (void)hipFree(c->polishList);
    if (c->polishRow) (void)hipFree(c->polishRow);
    if (c->polishVal) (void)hipFree(c->polishVal);
    if (c->elimList) (void)hipFree(c->elimList);
    if (c->floatList) (void)hipFree(c->floatList);
    if (c->groundMark) (void)hipFree(c->groundMark - c->L.guard);
    if (c->pairList) (void)hipFree(c->pairList);
    for (hipEvent_t e : c->evPool) (void)hipEventDestroy(e);
    for (int q = 0; q <= FLIPV_PHASE_COUNT; q++) if (c->phaseEv[q]) (void)hipEventDestroy(c->phaseEv[q]);
    if (c->evMain) (void)hipEventDestroy(c->evMain);
    if (c->evHalo) (void)hipEventDestroy(c->evHalo);
    for (int q = 0; q < 2; q++) if (c->evPoll[q]) (void)hipEventDestroy(c->evPoll[q]);
    if (c->commStream) (void)hipStreamDestroy(c->commStream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FLIPV_OK;
}

extern "C" int flipv_device_name(flipv_context *c, char *buf, size_t len) {
    if (!c || !buf || !len) return FLIPV_ERR_INVALID;
    hipDeviceProp_t p;
    HIPCHK(c, hipGetDeviceProperties(&p, c->device));
    snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return FLIPV_OK;
}

extern "C" int flipv_set_params(flipv_context *c, const flipv_params *p) {
    if (!c || !p) return FLIPV_ERR_INVALID;
    if (p->pressure_max_iterations < 1 || p->viscosity_max_iterations < 1 || !(p->cfl_number > 0) ||
        (p->precision != FLIPV_PRECISION_FP32 && p->precision != FLIPV_PRECISION_FP64)) {
        c->err = "flipv_set_params: invalid parameter";
        return FLIPV_ERR_INVALID;
    }
    if (c->isBlock && fv_min_slab_planes(p->cfl_number) > FV_HALO) {
        c->err = "flipv_set_params: cfl_number needs a halo of " + std::to_string(fv_min_slab_planes(p->cfl_number)) + " entries, a block context allocates " +
                 std::to_string(FV_HALO);
        return FLIPV_ERR_INVALID;
    }
    if (c->comm && c->comm->nranks > 1) {
        const int rcT = fv_check_block_thickness(c, p->cfl_number, "flipv_set_params");
        if (rcT) return rcT;
    }
    // every field has an error path: nothing out of its documented range is silently ignored or used as given
    {
        const char *bad = nullptr;
        auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
        auto fin = [](float v, float lo, float hi) { return v >= lo && v <= hi; };   // (false for NaN)
        if (!(p->min_frac > 0.0f && p->min_frac <= 1.0f)) bad = "min_frac";
        else if (!fin(p->pic_ratio, 0.0f, 1.0f)) bad = "pic_ratio";
        else if (!in(p->extrapolation_layers, 0, 64)) bad = "extrapolation_layers";
        else if (!(p->pressure_tolerance >= 0.0) || !(p->pressure_rel_tolerance >= 0.0) || !(p->viscosity_tolerance > 0.0) || !(p->viscosity_accept_tolerance >= 0.0)) bad = "a tolerance";
        else if (!in(p->check_every, 0, 4096)) bad = "check_every";
        else if (!in(p->pressure_preconditioner, 0, 2) || !in(p->viscosity_preconditioner, 0, 2)) bad = "a preconditioner";
        else if (!in(p->exact_viscosity_operator, 0, 1)) bad = "exact_viscosity_operator";
        else if (!in(p->viscosity_layout, 0, 3)) bad = "viscosity_layout";
        else if (!in(p->verbose, 0, 2) || !in(p->multigrid_rank_local, 0, 1)) bad = "verbose / multigrid_rank_local";
        else if (!in(p->multigrid_distributed_levels, -1, 1)) bad = "multigrid_distributed_levels";
        else if (!(p->viscosity_stage1_factor == 0.0f || fin(p->viscosity_stage1_factor, 1.0f, 1e6f))) bad = "viscosity_stage1_factor (0 or >= 1)";
        else if (!fin(p->viscosity_stage2_factor, 0.0f, 0.5f)) bad = "viscosity_stage2_factor (0 ... 0.5: a stage that reduces nothing is no stage)";
        else if (!in(p->viscosity_stage2_max_iterations, 0, 1 << 20) || !in(p->viscosity_stage2_rounds, 0, 16)) bad = "viscosity_stage2_max_iterations / viscosity_stage2_rounds";
        else if (p->viscosity_stage2_max_iterations > p->viscosity_max_iterations) bad = "viscosity_stage2_max_iterations (beyond viscosity_max_iterations)";
        else if (!(p->viscosity_two_stage_max_stiffness >= 0.0f)) bad = "viscosity_two_stage_max_stiffness";
        else if (!in(p->viscosity_defect_predictor, -1, 0)) bad = "viscosity_defect_predictor";
        else if (!(p->viscosity_velocity_tolerance == -1.0f || fin(p->viscosity_velocity_tolerance, 0.0f, 1.0f)) || !in(p->viscosity_velocity_window, 0, 8)) bad = "viscosity_velocity_tolerance (-1, or 0 ... 1) / viscosity_velocity_window (0 ... 8)";
        else if (!(p->viscosity_mass_scale == -1.0f || fin(p->viscosity_mass_scale, 0.0f, 1e9f))) bad = "viscosity_mass_scale (-1, or >= 0)";
        else if (!fin(p->viscosity_velocity_stall_ratio, 0.0f, 1.0f)) bad = "viscosity_velocity_stall_ratio (0 .. 1)";
        else if (!fin(p->viscosity_mass_floor, 0.0f, 1.0f)) bad = "viscosity_mass_floor (0 .. 1)";
        else if (!in(p->viscosity_massless_polish, -1, 0)) bad = "viscosity_massless_polish (0 on, -1 off)";
        else if (!in(p->viscosity_pair_correction, -1, 1)) bad = "viscosity_pair_correction (0 on for a viscosity field, 1 on, -1 off)";
        if (bad) { c->err = std::string("flipv_set_params: out of range: ") + bad; return FLIPV_ERR_INVALID; }
    }
    static_cast<flipv_params &>(c->prm) = *p;
    return FLIPV_OK;
}
extern "C" int flipv_set_debug_params(flipv_context *c, const flipv_debug_params *p) {
    if (!c || !p) return FLIPV_ERR_INVALID;
    const char *bad = nullptr;
    auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
    auto fin = [](float v, float lo, float hi) { return v >= lo && v <= hi; };
    auto sweeps = [](int v) { return v >= 0 && v <= 64; };
    if (!in(p->kernel_timing, 0, 1)) bad = "kernel_timing";
    else if (p->tile_rows != 0 && p->tile_rows != 16 && p->tile_rows != 64) bad = "tile_rows";
    else if (!sweeps(p->viscosity_mg_coarsest_sweeps) || !sweeps(p->pressure_mg_coarsest_sweeps)) bad = "a coarsest-level sweep count (0 ... 64)";
    else if (!in(p->viscosity_mg_min_dim, 0, 4096)) bad = "viscosity_mg_min_dim";
    else if (!fin(p->pressure_mg_omega, 0.0f, 2.0f) || !fin(p->pressure_mg_overcorrection, 0.0f, 4.0f)) bad = "pressure_mg_omega / pressure_mg_overcorrection";
    else if (!fin(p->viscosity_mg_omega_first, 0.0f, 2.0f) || !fin(p->viscosity_mg_omega_second, 0.0f, 2.0f)) bad = "viscosity_mg_omega_*";
    else if (!in(p->no_liquid_box, 0, 1) || !in(p->no_comm_overlap, 0, 1) || !in(p->no_graph_replay, 0, 1) || !in(p->unbinned_scatter, 0, 1) || !in(p->beta_from_conjugacy, 0, 1)) bad = "a 0/1 switch";
    else if (!in(p->grid_cap, 0, 1 << 20) || !in(p->viscosity_spmv_grid_cap, 0, 1 << 20) || !in(p->viscosity_update_grid_cap, 0, 1 << 20)) bad = "a grid cap";
    else if (p->viscosity_lane_width != 0 && p->viscosity_lane_width != 2 && p->viscosity_lane_width != 4) bad = "viscosity_lane_width";
    else if (!in(p->spmv_run_length, -2, 64) || p->spmv_run_length == 1) bad = "spmv_run_length";
    else if (!in(p->viscosity_mg_packed_rows, -1, 1)) bad = "viscosity_mg_packed_rows";
    else if (!(p->stall_guard_ratio == 0.0f || fin(p->stall_guard_ratio, 1.0f, 1e30f))) bad = "stall_guard_ratio (0 or >= 1)";
    else if (!fin(p->viscosity_pair_lambda_floor, 0.0f, 1.0f)) bad = "viscosity_pair_lambda_floor (0 ... 1)";
    else if (!in(p->velocity_patience, 0, 1 << 20)) bad = "velocity_patience";
    if (bad) { c->err = std::string("flipv_set_debug_params: out of range: ") + bad; return FLIPV_ERR_INVALID; }
    static_cast<flipv_debug_params &>(c->prm) = *p;
    return FLIPV_OK;
}
extern "C" int flipv_get_debug_params(flipv_context *c, flipv_debug_params *p) {
    if (!c || !p) return FLIPV_ERR_INVALID;
    *p = static_cast<const flipv_debug_params &>(c->prm);
    return FLIPV_OK;
}
extern "C" int flipv_get_params(flipv_context *c, flipv_params *p) {
    if (!c || !p) return FLIPV_ERR_INVALID;
    *p = static_cast<const flipv_params &>(c->prm);
    return FLIPV_OK;
}
extern "C" int flipv_set_gravity(flipv_context *c, float gx, float gy, float gz) {
    if (!c) return FLIPV_ERR_INVALID;
    c->gravity[0] = gx; c->gravity[1] = gy; c->gravity[2] = gz;
    return FLIPV_OK;
}

// ------------------------------------------------------------------------------------------------ grids
struct GridRef { float *f; uint8_t *m; int lat; };
static bool grid_ref(flipv_context *c, int which, GridRef *g) {
    g->f = nullptr; g->m = nullptr;
    switch (which) {
        case FLIPV_GRID_U: g->f = c->U; g->lat = LAT_U; return true;
        case FLIPV_GRID_V: g->f = c->V; g->lat = LAT_V; return true;
        case FLIPV_GRID_W: g->f = c->W; g->lat = LAT_W; return true;
        case FLIPV_GRID_SAVED_U: g->f = c->sU; g->lat = LAT_U; return true;
        case FLIPV_GRID_SAVED_V: g->f = c->sV; g->lat = LAT_V; return true;
        case FLIPV_GRID_SAVED_W: g->f = c->sW; g->lat = LAT_W; return true;
        case FLIPV_GRID_VALID_U: g->m = c->vU; g->lat = LAT_U; return true;
        case FLIPV_GRID_VALID_V: g->m = c->vV; g->lat = LAT_V; return true;
        case FLIPV_GRID_VALID_W: g->m = c->vW; g->lat = LAT_W; return true;
        case FLIPV_GRID_LIQUID_PHI: g->f = c->phi; g->lat = LAT_CELL; return true;
        case FLIPV_GRID_SOLID_PHI: g->f = c->solid; g->lat = LAT_NODE; return true;
        case FLIPV_GRID_WEIGHT_U: g->f = c->wU; g->lat = LAT_U; return true;
        case FLIPV_GRID_WEIGHT_V: g->f = c->wV; g->lat = LAT_V; return true;
        case FLIPV_GRID_WEIGHT_W: g->f = c->wW; g->lat = LAT_W; return true;
        case FLIPV_GRID_VISCOSITY: g->f = c->visc; g->lat = LAT_NODE; return true;
        case FLIPV_GRID_PRESSURE: g->f = c->pressure; g->lat = LAT_CELL; return true;
    }
    return false;
}
static size_t lat_count(const Lay &L, int lat) {
    int w, h, d;
    lat_dims(L, lat, w, h, d);
    return (size_t)w * h * d;
}

extern "C" size_t flipv_grid_elements(flipv_context *c, int which) {
    GridRef g;
    if (!c || !grid_ref(c, which, &g)) return 0;
    return lat_count(c->L, g.lat);
}

// The part of lattice `lat` inside a box of the index space: kind 0 = the indices this rank owns, 1 = the indices it allocates
// (owned + halo).  Returns false if the intersection is empty.
static bool lat_box(const flipv_context *c, int lat, int kind, int lo[3], int hi[3]) {
    const Lay &L = c->L;
    int dims[3];
    lat_dims(L, lat, dims[0], dims[1], dims[2]);
    const int o[3] = {L.ox, L.oy, L.oz}, P[3] = {L.PX, L.PY, L.PZ};
    bool any = true;
    for (int a = 0; a < 3; a++) {
        lo[a] = kind == 0 ? L.olo[a] : o[a];
        hi[a] = kind == 0 ? L.ohi[a] : o[a] + P[a];
        if (hi[a] > dims[a]) hi[a] = dims[a];
        if (hi[a] <= lo[a]) any = false;
    }
    return any;
}
static size_t box_count(const int lo[3], const int hi[3]) { return (size_t)(hi[0] - lo[0]) * (size_t)(hi[1] - lo[1]) * (size_t)(hi[2] - lo[2]); }

// device -> host: the OWNED part of the lattice, box-shaped (x fastest) into `out`
static int read_lattice_box(flipv_context *c, int lat, const float *srcf, const uint8_t *srcb, float *out) {
    int lo[3], hi[3];
    if (!lat_box(c, lat, 0, lo, hi)) return FLIPV_OK;
    int rc = fv_pack(c, lat, srcf, srcb, c->stage, lo, hi);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->stage, box_count(lo, hi) * 4, hipMemcpyDeviceToHost, c->stream));
    FV_SYNC(c);
    return FLIPV_OK;
}
// host -> device: the ALLOCATED part of the lattice, box-shaped in `in`; entries of the allocated box outside the lattice are zeroed
static int write_lattice_box(flipv_context *c, int lat, const float *in, float *dstf, uint8_t *dstb) {
    int lo[3], hi[3];
    if (!lat_box(c, lat, 1, lo, hi)) return FLIPV_OK;
    HIPCHK(c, hipMemcpyAsync(c->stage, in, box_count(lo, hi) * 4, hipMemcpyHostToDevice, c->stream));
    int rc = fv_unpack(c, lat, c->stage, dstf, dstb, lo, hi);
    if (rc) return rc;
    FV_SYNC(c);
    return FLIPV_OK;
}
// Full-size Array3d at the ABI.  A single-domain context's box IS the lattice; a block context moves its box only: a read
// fills the entries this rank owns and leaves the rest of `out` untouched, a write takes the entries it allocates.
static int read_lattice(flipv_context *c, int lat, const float *srcf, const uint8_t *srcb, float *out) {
    if (!c->isBlock) return read_lattice_box(c, lat, srcf, srcb, out);
    int lo[3], hi[3];
    if (!lat_box(c, lat, 0, lo, hi)) return FLIPV_OK;
    std::vector<float> tmp(box_count(lo, hi));
    int rc = read_lattice_box(c, lat, srcf, srcb, tmp.data());
    if (rc) return rc;
    int w, h, d;
    lat_dims(c->L, lat, w, h, d);
    const size_t bw = (size_t)(hi[0] - lo[0]), bh = (size_t)(hi[1] - lo[1]);
    for (int k = lo[2]; k < hi[2]; k++)
        for (int j = lo[1]; j < hi[1]; j++)
            memcpy(out + (size_t)lo[0] + (size_t)w * ((size_t)j + (size_t)h * (size_t)k), tmp.data() + bw * ((size_t)(j - lo[1]) + bh * (size_t)(k - lo[2])), bw * 4);
    return FLIPV_OK;
}
static int write_lattice(flipv_context *c, int lat, const float *in, float *dstf, uint8_t *dstb) {
    if (!c->isBlock) return write_lattice_box(c, lat, in, dstf, dstb);
    int lo[3], hi[3];
    if (!lat_box(c, lat, 1, lo, hi)) return FLIPV_OK;
    std::vector<float> tmp(box_count(lo, hi));
    int w, h, d;
    lat_dims(c->L, lat, w, h, d);
    const size_t bw = (size_t)(hi[0] - lo[0]), bh = (size_t)(hi[1] - lo[1]);
    for (int k = lo[2]; k < hi[2]; k++)
        for (int j = lo[1]; j < hi[1]; j++)
            memcpy(tmp.data() + bw * ((size_t)(j - lo[1]) + bh * (size_t)(k - lo[2])), in + (size_t)lo[0] + (size_t)w * ((size_t)j + (size_t)h * (size_t)k), bw * 4);
    return write_lattice_box(c, lat, tmp.data(), dstf, dstb);
}

#define NOT_SETUP_ONLY(c) do { if ((c)->setupOnly) { (c)->err = "this context was created by flipv_create_setup: it only serves the scene-setup entry points"; return FLIPV_ERR_INVALID; } } while (0)
#define ENTER(c) do { if (!(c)) return FLIPV_ERR_INVALID; hipError_t e_ = hipSetDevice((c)->device); if (e_ != hipSuccess) { (c)->err = hipGetErrorString(e_); return FLIPV_ERR_HIP; } } while (0)
#define SYNC_RET(c, rc) do { const int rc__ = (rc); const int rs__ = fv_sync(c); return rs__ ? rs__ : rc__; } while (0)

extern "C" int flipv_read_grid(flipv_context *c, int which, float *out) {
    ENTER(c);
    GridRef g;
    if (!out || !grid_ref(c, which, &g)) { c->err = "flipv_read_grid: bad grid id"; return FLIPV_ERR_INVALID; }
    return read_lattice(c, g.lat, g.f, g.m, out);
}

extern "C" int flipv_write_grid(flipv_context *c, int which, const float *in) {
    ENTER(c);
    c->liqValid = c->liqPrevValid = 0;   // grids edited from outside: the next substep sweeps everything
    GridRef g;
    if (!in || !grid_ref(c, which, &g)) { c->err = "flipv_write_grid: bad grid id"; return FLIPV_ERR_INVALID; }
    if (which == FLIPV_GRID_SOLID_PHI) c->solidVersion++;
    if (which == FLIPV_GRID_WEIGHT_U || which == FLIPV_GRID_WEIGHT_V || which == FLIPV_GRID_WEIGHT_W) c->weightsVersion = -1;
    if (which == FLIPV_GRID_VISCOSITY) {
        const size_t n = lat_count(c->L, g.lat);
        float vmax = 0.0f, vmin = n ? in[0] : 0.0f;
        for (size_t t = 0; t < n; t++) { if (in[t] > vmax) vmax = in[t]; if (in[t] < vmin) vmin = in[t]; }
        c->viscosity_nonzero = vmax > 0.0f;
        c->viscosity_max = vmax; c->viscosity_min = vmin;
    }
    return write_lattice(c, g.lat, in, g.f, g.m);
}

extern "C" int flipv_grid_box(flipv_context *c, int which, int kind, int *lo, int *hi) {
    GridRef g;
    if (!c || !lo || !hi || (kind != 0 && kind != 1) || !grid_ref(c, which, &g)) return FLIPV_ERR_INVALID;
    if (!lat_box(c, g.lat, kind, lo, hi)) { for (int a = 0; a < 3; a++) hi[a] = lo[a]; }
    return FLIPV_OK;
}
extern "C" int flipv_read_grid_box(flipv_context *c, int which, float *out) {
    ENTER(c);
    GridRef g;
    if (!out || !grid_ref(c, which, &g)) { c->err = "flipv_read_grid_box: bad grid id"; return FLIPV_ERR_INVALID; }
    return read_lattice_box(c, g.lat, g.f, g.m, out);
}
extern "C" int flipv_read_grid_region(flipv_context *c, int which, const int *lo, const int *hi, float *out) {
    ENTER(c);
    GridRef g;
    if (!out || !lo || !hi || !grid_ref(c, which, &g)) { c->err = "flipv_read_grid_region: bad argument"; return FLIPV_ERR_INVALID; }
    int alo[3], ahi[3];
    if (!lat_box(c, g.lat, 1, alo, ahi)) { c->err = "flipv_read_grid_region: the context holds nothing of this grid"; return FLIPV_ERR_INVALID; }
    for (int a = 0; a < 3; a++)
        if (lo[a] < alo[a] || hi[a] > ahi[a] || hi[a] <= lo[a]) { c->err = "flipv_read_grid_region: box outside what the context allocates"; return FLIPV_ERR_INVALID; }
    const int rc = fv_pack(c, g.lat, g.f, g.m, c->stage, lo, hi);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->stage, box_count(lo, hi) * 4, hipMemcpyDeviceToHost, c->stream));
    FV_SYNC(c);
    return FLIPV_OK;
}
extern "C" int flipv_write_grid_box(flipv_context *c, int which, const float *in) {
    ENTER(c);
    c->liqValid = c->liqPrevValid = 0;
    GridRef g;
    if (!in || !grid_ref(c, which, &g)) { c->err = "flipv_write_grid_box: bad grid id"; return FLIPV_ERR_INVALID; }
    if (which == FLIPV_GRID_SOLID_PHI) c->solidVersion++;
    if (which == FLIPV_GRID_WEIGHT_U || which == FLIPV_GRID_WEIGHT_V || which == FLIPV_GRID_WEIGHT_W) c->weightsVersion = -1;
    if (which == FLIPV_GRID_VISCOSITY) {
        int lo[3], hi[3];
        float vmax = 0.0f, vmin = 0.0f;
        if (lat_box(c, g.lat, 1, lo, hi)) { const size_t n = box_count(lo, hi); vmin = n ? in[0] : 0.0f; for (size_t t = 0; t < n; t++) { if (in[t] > vmax) vmax = in[t]; if (in[t] < vmin) vmin = in[t]; } }
        c->viscosity_nonzero = vmax > 0.0f;   // this rank's box; the solve all-reduces it (k_viscosity.hip) so that every rank takes the same path
        c->viscosity_max = vmax; c->viscosity_min = vmin;
    }
    return write_lattice_box(c, g.lat, in, g.f, g.m);
}

extern "C" int flipv_set_solid_sdf(flipv_context *c, const float *nodes) { return flipv_write_grid(c, FLIPV_GRID_SOLID_PHI, nodes); }
extern "C" int flipv_set_viscosity(flipv_context *c, const float *nodes) {
    if (!c || !nodes) return FLIPV_ERR_INVALID;
    const size_t n = lat_count(c->L, LAT_NODE);
    for (size_t t = 0; t < n; t++)
        if (!(nodes[t] >= 0.0f)) { c->err = "flipv_set_viscosity: negative viscosity"; return FLIPV_ERR_INVALID; }  // fluidsimulation.cpp:119
    return flipv_write_grid(c, FLIPV_GRID_VISCOSITY, nodes);
}
extern "C" int flipv_set_viscosity_uniform(flipv_context *c, float value) {
    if (!c) return FLIPV_ERR_INVALID;
    if (!(value >= 0.0f)) { c->err = "flipv_set_viscosity_uniform: negative viscosity"; return FLIPV_ERR_INVALID; }  // fluidsimulation.cpp:100
    int lo[3], hi[3];
    if (!lat_box(c, LAT_NODE, 1, lo, hi)) return FLIPV_OK;
    std::vector<float> v(box_count(lo, hi), value);
    const int rc = flipv_write_grid_box(c, FLIPV_GRID_VISCOSITY, v.data());
    c->viscosity_nonzero = value > 0.0f;
    c->viscosity_max = value; c->viscosity_min = value;
    return rc;
}

// ------------------------------------------------------------------------------------------------ scene setup (k_meshsdf.hip)
int fv_mesh_level_set(flipv_context *c, const float *verts, size_t nverts, const int *tris, size_t ntris, int band, float *phi,
                      int *closest_out);
int fv_mesh_negate(flipv_context *c, float *phi);
int fv_mesh_union(flipv_context *c, float *into, const float *other);
int fv_seed_particles(flipv_context *c, const float *meshphi, unsigned long long seed, size_t *added);

namespace {
// a temporary grid in the shared index space (guard zones included), zero-initialised
struct TempGrid {
    void *base = nullptr;
    float *p = nullptr;
    int alloc(flipv_context *c) {
        const size_t tot = c->L.n + 2 * c->L.guard;
        hipError_t e = hipMalloc(&base, tot * 4);
        if (e != hipSuccess) { c->err = std::string("hipMalloc(temporary grid): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        e = hipMemsetAsync(base, 0, tot * 4, c->stream);
        if (e != hipSuccess) { c->err = hipGetErrorString(e); return FLIPV_ERR_HIP; }
        p = (float *)base + c->L.guard;
        return FLIPV_OK;
    }
    ~TempGrid() { if (base) (void)hipFree(base); }
};
int mesh_inside_domain(flipv_context *c, const float *v, size_t nverts, const char *who) {  // fluidsimulation.cpp:46-49, 65-68
    if (c->comm || c->isBlock) { c->err = std::string(who) + ": scene setup runs on a single-domain context"; return FLIPV_ERR_INVALID; }
    if (!v || nverts == 0) { c->err = std::string(who) + ": empty mesh"; return FLIPV_ERR_INVALID; }
    float lo[3] = {v[0], v[1], v[2]}, hi[3] = {v[0], v[1], v[2]};
    for (size_t t = 0; t < nverts; t++)
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], v[3 * t + a]); hi[a] = fmaxf(hi[a], v[3 * t + a]); }
    const double ext[3] = {c->L.I * (double)c->dx, c->L.J * (double)c->dx, c->L.K * (double)c->dx};
    for (int a = 0; a < 3; a++)
        if (!(lo[a] >= 0.0 && hi[a] < ext[a])) { c->err = std::string(who) + ": mesh bounding box outside the domain"; return FLIPV_ERR_INVALID; }
    return FLIPV_OK;
}
}  // namespace

extern "C" int flipv_mesh_level_set(flipv_context *c, const float *vertices, size_t nvertices, const int *triangles, size_t ntriangles,
                                    int bandwidth, float *phi_out, int *closest_out) {
    ENTER(c);
    if (!phi_out) return FLIPV_ERR_INVALID;
    if (c->comm || c->isBlock) { c->err = "flipv_mesh_level_set: scene setup runs on a single-domain context"; return FLIPV_ERR_INVALID; }
    TempGrid phi, clo;
    int rc = phi.alloc(c);
    if (rc) return rc;
    if (closest_out && (rc = clo.alloc(c))) return rc;
    rc = fv_mesh_level_set(c, vertices, nvertices, triangles, ntriangles, bandwidth, phi.p, closest_out ? (int *)clo.p : nullptr);
    if (rc) return rc;
    rc = read_lattice(c, LAT_NODE, phi.p, nullptr, phi_out);
    if (rc) return rc;
    if (closest_out) rc = read_lattice(c, LAT_NODE, clo.p, nullptr, (float *)closest_out);  // 4-byte values, copied bit for bit
    return rc;
}

extern "C" int flipv_add_boundary_mesh(flipv_context *c, const float *vertices, size_t nvertices, const int *triangles,
                                       size_t ntriangles, int inverted) {
    ENTER(c);
    int rc = mesh_inside_domain(c, vertices, nvertices, "flipv_add_boundary_mesh");
    if (rc) return rc;
    TempGrid phi;
    if ((rc = phi.alloc(c))) return rc;
    if ((rc = fv_mesh_level_set(c, vertices, nvertices, triangles, ntriangles, 3, phi.p, nullptr))) return rc;  // _meshLevelSetExactBand (fluidsimulation.h:121)
    if (inverted && (rc = fv_mesh_negate(c, phi.p))) return rc;
    rc = fv_mesh_union(c, c->solid, phi.p);
    c->solidVersion++;
    SYNC_RET(c, rc);
}

extern "C" int flipv_reset_boundary(flipv_context *c) {  // _initializeBoundary (fluidsimulation.cpp:198-239)
    ENTER(c);
    if (c->comm || c->isBlock) { c->err = "flipv_reset_boundary: scene setup runs on a single-domain context"; return FLIPV_ERR_INVALID; }
    const double dx = (double)c->dx, eps = 1e-6;
    // AABB(0,0,0, I dx, J dx, K dx).expand(-3 dx - eps): the constructor and expand() work in double, the corners are
    // stored as float (aabb.cpp:118-124)
    const double v = -3 * dx - eps, hh = 0.5 * v;
    const float px = (float)(0.0 - hh), py = (float)(0.0 - hh), pz = (float)(0.0 - hh);
    const float w = (float)(c->L.I * dx + v), h = (float)(c->L.J * dx + v), d = (float)(c->L.K * dx + v);
    const float verts[24] = {px, py, pz,         px + w, py, pz,         px + w, py, pz + d,     px, py, pz + d,
                             px, py + h, pz,     px + w, py + h, pz,     px + w, py + h, pz + d, px, py + h, pz + d};
    const int tris[36] = {0, 1, 2, 0, 2, 3, 4, 7, 6, 4, 6, 5, 0, 3, 7, 0, 7, 4, 1, 5, 6, 1, 6, 2, 0, 4, 5, 0, 5, 1, 3, 2, 6, 3, 6, 7};
    int rc = fv_mesh_level_set(c, verts, 8, tris, 12, 3, c->solid, nullptr);
    if (rc) return rc;
    rc = fv_mesh_negate(c, c->solid);
    c->solidVersion++;
    SYNC_RET(c, rc);
}

extern "C" int flipv_add_liquid_mesh(flipv_context *c, const float *vertices, size_t nvertices, const int *triangles,
                                     size_t ntriangles, unsigned long long seed, size_t *added_out) {
    ENTER(c);
    int rc = mesh_inside_domain(c, vertices, nvertices, "flipv_add_liquid_mesh");
    if (rc) return rc;
    TempGrid phi;
    if ((rc = phi.alloc(c))) return rc;
    if ((rc = fv_mesh_level_set(c, vertices, nvertices, triangles, ntriangles, 3, phi.p, nullptr))) return rc;
    return fv_seed_particles(c, phi.p, seed, added_out);
}

// ------------------------------------------------------------------------------------------------ particles
extern "C" int flipv_upload_particles(flipv_context *c, const float *aos6, size_t n) {
    ENTER(c);
    if (n && !aos6) return FLIPV_ERR_INVALID;
    if (n > c->pcap) {
        if (c->particles) (void)hipFree(c->particles);
        c->particles = nullptr;
        c->pcap = 0;
        const size_t cap = n + n / 8 + 1024;
        hipError_t e = hipMalloc((void **)&c->particles, cap * 6 * sizeof(float));
        if (e != hipSuccess) { c->err = std::string("hipMalloc(particles): ") + hipGetErrorString(e); return FLIPV_ERR_OOM; }
        c->pcap = cap;
    }
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->particles, aos6, n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        FV_SYNC(c);
    }
    c->np = n;
    c->binsValid = 0;
    return FLIPV_OK;
}
extern "C" int flipv_download_particles(flipv_context *c, float *aos6, size_t capacity, size_t *n_out) {
    ENTER(c);
    if (n_out) *n_out = c->np;
    if (capacity < c->np) { c->err = "flipv_download_particles: buffer too small"; return FLIPV_ERR_INVALID; }
    if (c->np) {
        if (!aos6) return FLIPV_ERR_INVALID;
        HIPCHK(c, hipMemcpyAsync(aos6, c->particles, c->np * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        FV_SYNC(c);
    }
    return FLIPV_OK;
}
extern "C" size_t flipv_num_particles(flipv_context *c) { return c ? c->np : 0; }

// ------------------------------------------------------------------------------------------------ operators
extern "C" int flipv_cfl(flipv_context *c, float *dt_out) { ENTER(c); NOT_SETUP_ONLY(c); if (!dt_out) return FLIPV_ERR_INVALID; return fv_cfl(c, dt_out); }
extern "C" int flipv_particle_sdf(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_particle_sdf(c)); }
extern "C" int flipv_p2g(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_p2g(c)); }
extern "C" int flipv_extrapolate(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_extrapolate(c)); }

static int save_velocity(flipv_context *c) {
    const Lay R = fv_range_liquid(c, 1, 8);   // (whole planes of its k-range)
    const size_t off = plane_off(c->L, R.kb), bytes = (size_t)(R.ke - R.kb) * c->L.sz * 4;   // whole allocated planes
    HIPCHK(c, hipMemcpyAsync(c->sU + off, c->U + off, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sV + off, c->V + off, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sW + off, c->W + off, bytes, hipMemcpyDeviceToDevice, c->stream));
    return FLIPV_OK;
}
extern "C" int flipv_save_velocity(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, save_velocity(c)); }

static int advect_velocity_field(flipv_context *c) {  // fluidsimulation.cpp:500-519
    int rc = fv_p2g(c);
    if (rc) return rc;
    rc = fv_extrapolate(c);
    if (rc) return rc;
    return save_velocity(c);
}
extern "C" int flipv_advect_velocity_field(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, advect_velocity_field(c)); }
extern "C" int flipv_body_force(flipv_context *c, float dt) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_body_force(c, dt)); }
extern "C" int flipv_viscosity_solve(flipv_context *c, float dt, flipv_solve_info *info) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_viscosity_solve(c, dt, info)); }
extern "C" int flipv_compute_weights(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_compute_weights(c)); }
extern "C" int flipv_pressure_solve(flipv_context *c, float dt, flipv_solve_info *info) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_pressure_solve(c, dt, info)); }
extern "C" int flipv_apply_pressure(flipv_context *c, float dt) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_apply_pressure(c, dt)); }
extern "C" int flipv_constrain(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_constrain(c)); }
extern "C" int flipv_update_particle_velocities(flipv_context *c) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_update_particle_velocities(c)); }
extern "C" int flipv_advect_particles(flipv_context *c, float dt) { ENTER(c); NOT_SETUP_ONLY(c); SYNC_RET(c, fv_advect_particles(c, dt)); }

extern "C" int flipv_read_viscosity_volume(flipv_context *c, int which, float *out) {
    ENTER(c);
    NOT_SETUP_ONLY(c);
    if (!out || which < 0 || which > 6) return FLIPV_ERR_INVALID;
    const float *src[7] = {c->volC, c->volU, c->volV, c->volW, c->volEU, c->volEV, c->volEW};
    const int lat[7] = {LAT_CELL, LAT_U, LAT_V, LAT_W, LAT_EU, LAT_EV, LAT_EW};
    return read_lattice(c, lat[which], src[which], nullptr, out);
}

// ------------------------------------------------------------------------------------------------ substep
static int substep_inner(flipv_context *c, float dt, flipv_stats *st);
static int substep(flipv_context *c, float dt, flipv_stats *st) {
    c->inSubstep = 1;   // the sweeps may restrict themselves to where the liquid is (flipv_context::liqValid)
    const int rc = substep_inner(c, dt, st);
    c->inSubstep = 0;
    return rc;
}
static int substep_inner(flipv_context *c, float dt, flipv_stats *st) {
    // phase order of the while-loop body, fluidsimulation.cpp:145-164
    int rc, warn = FLIPV_OK;
    flipv_solve_info vi, pi;
    memset(&vi, 0, sizeof(vi));
    memset(&pi, 0, sizeof(pi));
#define MARK(q) HIPCHK(c, hipEventRecord(c->phaseEv[q], c->stream))
    MARK(0);
    if ((rc = fv_particle_sdf(c)) < 0) return rc;
    MARK(1);
    if ((rc = advect_velocity_field(c)) < 0) return rc;
    MARK(2);
    if ((rc = fv_body_force(c, dt)) < 0) return rc;
    MARK(3);
    if ((rc = fv_viscosity_solve(c, dt, &vi)) < 0) return rc;
    if (rc > warn) warn = rc;
    MARK(4);
    // the face weights depend on the solid SDF only: the reference recomputes them every substep (fluidsimulation.cpp:585),
    // here they are kept until the solid SDF (or, through flipv_write_grid, a weight grid) changes
    if (c->weightsVersion != c->solidVersion) {
        if ((rc = fv_compute_weights(c)) < 0) return rc;
        c->weightsVersion = c->solidVersion;
    }
    if ((rc = fv_pressure_solve(c, dt, &pi)) < 0) return rc;
    if (rc > warn) warn = rc;
    if ((rc = fv_apply_pressure(c, dt)) < 0) return rc;
    if ((rc = fv_extrapolate(c)) < 0) return rc;
    MARK(5);
    if ((rc = fv_constrain(c)) < 0) return rc;
    MARK(6);
    if ((rc = fv_advect_particles(c, dt)) < 0) return rc;
    MARK(7);
#undef MARK
    HIPCHK(c, hipEventSynchronize(c->phaseEv[7]));
    if (st) {
        double tot = 0;
        for (int q = 0; q < FLIPV_PHASE_COUNT; q++) {
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->phaseEv[q], c->phaseEv[q + 1]));
            st->phase_ms[q] = ms;
            tot += ms;
        }
        st->total_ms = tot;
        st->dt = dt;
        st->substeps = 1;
        st->viscosity = vi;
        st->pressure = pi;
    }
    return warn;
}

extern "C" int flipv_substep(flipv_context *c, float dt, flipv_stats *st) {
    ENTER(c);
    NOT_SETUP_ONLY(c);
    if (!(dt > 0.0f)) { c->err = "flipv_substep: dt must be > 0"; return FLIPV_ERR_INVALID; }
    return substep(c, dt, st);
}

extern "C" int flipv_advance(flipv_context *c, float dt, flipv_stats *st) {
    ENTER(c);
    NOT_SETUP_ONLY(c);
    if (!(dt > 0.0f)) { c->err = "flipv_advance: dt must be > 0"; return FLIPV_ERR_INVALID; }
    // fluidsimulation.cpp:135-168
    float t = 0;
    int n = 0, warn = FLIPV_OK;
    flipv_stats acc, one;
    memset(&acc, 0, sizeof(acc));
    while (t < dt) {
        float sub;
        int rc = fv_cfl(c, &sub);
        if (rc < 0) return rc;
        if (!(sub > 0.0f)) { c->err = "flipv_advance: CFL substep is not positive (non-finite velocities?)"; return FLIPV_ERR_INVALID; }
        if (t + sub > dt) sub = dt - t;
        rc = substep(c, sub, &one);
        if (rc < 0) return rc;
        if (rc > warn) warn = rc;
        for (int q = 0; q < FLIPV_PHASE_COUNT; q++) acc.phase_ms[q] += one.phase_ms[q];
        acc.total_ms += one.total_ms;
        acc.viscosity = one.viscosity;
        acc.pressure = one.pressure;
        acc.dt = sub;
        t += sub;
        n++;
    }
    acc.substeps = n;
    if (st) *st = acc;
    return warn;
}

// ------------------------------------------------------------------------------------------------ timing
void fv_ev_begin(flipv_context *c, int which, double cells) {
    if (c->evUsed + 2 > c->evPool.size()) {
        const size_t grow = c->evPool.size() + 512;
        while (c->evPool.size() < grow) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            c->evPool.push_back(e);
        }
    }
    flipv_context::EvSpan s;
    s.a = c->evUsed++; s.b = c->evUsed++; s.which = which; s.cells = cells;
    (void)hipEventRecord(c->evPool[s.a], c->stream);
    c->evSpans.push_back(s);
}
void fv_ev_end(flipv_context *c) {
    if (c->evSpans.empty()) return;
    (void)hipEventRecord(c->evPool[c->evSpans.back().b], c->stream);
}
void fv_ev_collect(flipv_context *c) {
    if (c->evSpans.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (const auto &s : c->evSpans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->evPool[s.a], c->evPool[s.b]) != hipSuccess) continue;
        if (s.which == 0) { c->kstats.pressure_spmv_ms += ms; c->kstats.pressure_spmv_launches++; c->kstats.pressure_spmv_cells += s.cells; }
        else { c->kstats.viscosity_spmv_ms += ms; c->kstats.viscosity_spmv_launches++; c->kstats.viscosity_spmv_cells += s.cells; }
    }
    c->evSpans.clear();
    c->evUsed = 0;
}

extern "C" int flipv_kernel_stats_reset(flipv_context *c) {
    if (!c) return FLIPV_ERR_INVALID;
    memset(&c->kstats, 0, sizeof(c->kstats));
    return FLIPV_OK;
}
extern "C" int flipv_kernel_stats_get(flipv_context *c, flipv_kernel_stats *out) {
    if (!c || !out) return FLIPV_ERR_INVALID;
    fv_ev_collect(c);
    *out = c->kstats;
    return FLIPV_OK;
}
extern "C" int flipv_synchronize(flipv_context *c) {
    ENTER(c);
    FV_SYNC(c);
    return FLIPV_OK;
}

int fv_bench_pressure_spmv(flipv_context *c, int reps, double *ms, double *cells);
int fv_bench_viscosity_spmv(flipv_context *c, int reps, double *ms, double *cells, int mgLoop);
extern "C" int flipv_bench_spmv(flipv_context *c, int which, int reps, double *ms_out, double *cells_out) {
    ENTER(c);
    NOT_SETUP_ONLY(c);
    if (!ms_out || !cells_out || reps < 1) return FLIPV_ERR_INVALID;
    if (which < 0 || which > 2) return FLIPV_ERR_INVALID;
    return which == 0 ? fv_bench_pressure_spmv(c, reps, ms_out, cells_out) : fv_bench_viscosity_spmv(c, reps, ms_out, cells_out, which == 2);
}

extern "C" int flipv_bench_copy(flipv_context *c, size_t bytes, int reps, double *gbps_out) {
    ENTER(c);
    if (!gbps_out || reps < 1 || bytes < 1024) return FLIPV_ERR_INVALID;
    void *a = nullptr, *b = nullptr;
    HIPCHK(c, hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); c->err = "flipv_bench_copy: out of memory"; return FLIPV_ERR_OOM; }
    (void)hipMemsetAsync(a, 1, bytes, c->stream);
    (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps; r++) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    *gbps_out = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    return FLIPV_OK;
}

// Attainable HBM rates with plain streaming kernels ("measure the attainable peak ... and quote both"):
// mode 0 = read-only (sum of 16-byte loads), 1 = copy (16-byte loads + stores), 2 = write-only.  GB/s counts the bytes moved.
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n, int mode, float *__restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        if (mode == 2) { b[t] = make_float4(1.0f, 2.0f, 3.0f, 4.0f); continue; }
        const float4 v = a[t];
        if (mode == 1) b[t] = v;
        else acc += v.x + v.y + v.z + v.w;
    }
    if (mode == 0 && acc == 123.456f) *sink = acc;  // keeps the loads alive
}
// The same three mixes the way a tuned streaming kernel moves them (modes 3 read, 4 copy, 5 = FIVE reads per write: the byte mix of the 7-point pressure
// SpMV, 5 x 4 B read + 4 B written per cell): each lane keeps four independent 16-byte accesses in flight, a block walks contiguous 16 KiB chunks, loads
// and stores are nontemporal (nothing is read twice).  bytes = the size of ONE array; GB/s counts every byte moved.
typedef float fv_v4f __attribute__((ext_vector_type(4)));
template <int NREAD, int NWRITE>
__global__ __launch_bounds__(256) void k_stream_tuned(const fv_v4f *__restrict__ a, fv_v4f *__restrict__ b, size_t n, float *__restrict__ sink) {
    fv_v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t base = (size_t)blockIdx.x * 1024; base < n; base += (size_t)gridDim.x * 1024) {
        fv_v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t t = base + (size_t)u * 256 + threadIdx.x;
            v[u] = fv_v4f{0.0f, 0.0f, 0.0f, 0.0f};
            if (t < n) {
#pragma unroll
                for (int m = 0; m < NREAD; m++) v[u] += __builtin_nontemporal_load(a + (size_t)m * n + t);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t t = base + (size_t)u * 256 + threadIdx.x;
            if (NWRITE) {
                if (t < n) {
#pragma unroll
                    for (int m = 0; m < NWRITE; m++) __builtin_nontemporal_store(v[u], b + (size_t)m * n + t);
                }
            } else acc += v[u];
        }
    }
    if (!NWRITE && acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = acc.x;
}
extern "C" int flipv_bench_stream(flipv_context *c, size_t bytes, int reps, int mode, double *gbps_out) {
    ENTER(c);
    if (!gbps_out || reps < 1 || bytes < 4096 || mode < 0 || mode > 6) return FLIPV_ERR_INVALID;
    void *a = nullptr, *b = nullptr;
    const int nread = mode == 5 ? 5 : (mode == 6 ? 10 : 1), nwrite = mode == 6 ? 3 : 1;
    HIPCHK(c, hipMalloc(&a, bytes * nread));
    if (hipMalloc(&b, bytes * nwrite + 64) != hipSuccess) { (void)hipFree(a); c->err = "flipv_bench_stream: out of memory"; return FLIPV_ERR_OOM; }
    (void)hipMemsetAsync(a, 0, bytes * nread, c->stream);
    (void)hipMemsetAsync(b, 0, bytes + 64, c->stream);
    const size_t n = bytes / 16;
    const unsigned grid = 256 * 8 * 4;  // 32 blocks per CU
    float *sink = (float *)((char *)b + bytes * nwrite);
    auto launch = [&]() {
        if (mode <= 2) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, c->stream, (const float4 *)a, (float4 *)b, n, mode, sink);
        else if (mode == 3) hipLaunchKernelGGL((k_stream_tuned<1, 0>), dim3(256 * 8), dim3(256), 0, c->stream, (const fv_v4f *)a, (fv_v4f *)b, n, sink);
        else if (mode == 4) hipLaunchKernelGGL((k_stream_tuned<1, 1>), dim3(256 * 8), dim3(256), 0, c->stream, (const fv_v4f *)a, (fv_v4f *)b, n, sink);
        else if (mode == 5) hipLaunchKernelGGL((k_stream_tuned<5, 1>), dim3(256 * 8), dim3(256), 0, c->stream, (const fv_v4f *)a, (fv_v4f *)b, n, sink);
        else hipLaunchKernelGGL((k_stream_tuned<10, 3>), dim3(256 * 8), dim3(256), 0, c->stream, (const fv_v4f *)a, (fv_v4f *)b, n, sink);
    };
    launch();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps; r++) launch();
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    const double moved = mode == 1 || mode == 4 ? 2.0 : (mode == 5 ? 6.0 : (mode == 6 ? 13.0 : 1.0));
    *gbps_out = moved * (double)(n * 16) * reps / ((double)ms * 1e-3) / 1e9;
    return FLIPV_OK;
}

// debug only (not part of the ABI): fold the slot-spread PCG scalars of the LAST solve on the host.
// out = 5 arrays of n doubles: sig, a, b, c, rmax.
#include "pcg_common.h"
extern "C" int fvdbg_pcg_scalars(flipv_context *c, int cap, int n, double *out) {
    const size_t stride = fv_scal_stride(cap);
    std::vector<double> h(FV_SCAL_BANKS * stride);
    if (hipMemcpy(h.data(), c->d_scal, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const int nbank = c->comm ? 1 : FV_SCAL_BANKS;
    for (int q = 0; q < 5; q++)
        for (int it = 0; it < n; it++) {
            double v = 0;
            for (int bk = 0; bk < nbank; bk++)
            for (int s = 0; s < NSLOT; s++) {
                const double x = h[(size_t)bk * stride + (size_t)it * FV_NSC * NSLOT + (size_t)q * NSLOT + s];   // block `it` = [sig | a | b | c | rmax | step] x NSLOT (PcgScal), per bank
                v = q == 4 ? (x > v ? x : v) : v + x;
            }
            out[(size_t)q * n + it] = v;
        }
    return 0;
}
