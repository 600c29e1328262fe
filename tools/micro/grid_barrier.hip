// Micro-benchmark: what a phase of a latency-bound multi-phase kernel costs when the phases are (a) separate launches replayed from a hipGraph,
// (b) one persistent launch with a device-wide barrier between phases, workgroups on every XCD, (c) the same with the working workgroups on
// one XCD only (workgroup id % 8 == 0; the others exit at once).  A phase: every lane reads 24 floats that other workgroups wrote in the
// previous phase, adds them and writes one float -- the shape of a coarse multigrid level's sweep.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier ; ./grid_barrier [phases]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float phase_work(const float *__restrict__ in, int n, int e) {
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 24; q++) { int o = e + (q - 12) * 97; o = o < 0 ? o + n : (o >= n ? o - n : o); s += __builtin_nontemporal_load(in + o) * (1.0f / 24.0f); }
    return s;
}
__global__ __launch_bounds__(256) void k_phase(const float *__restrict__ in, float *__restrict__ out, int n) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) out[e] = phase_work(in, n, e) + 1.0f;
}
// ctr: arrivals; every workgroup passes `gen` (a multiple of the number of working workgroups) barriers in step
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned nwg, unsigned &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += nwg;
        __threadfence();
        atomicAdd(ctr, 1u);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}
template <int STRIDE>
__global__ __launch_bounds__(256) void k_persistent(float *a, float *b, int n, int phases, unsigned *ctr) {
    if (STRIDE > 1 && (blockIdx.x % STRIDE) != 0) return;
    const int wg = (int)blockIdx.x / STRIDE, nwg = (int)gridDim.x / STRIDE;
    unsigned gen = 0;
    for (int p = 0; p < phases; p++) {
        const float *in = (p & 1) ? b : a;
        float *out = (p & 1) ? a : b;
        for (int e = wg * 256 + threadIdx.x; e < n; e += nwg * 256) out[e] = phase_work(in, n, e) + 1.0f;
        grid_barrier(ctr, (unsigned)nwg, gen);
    }
    // the last workgroup to leave re-arms the counter for the next launch
    if (threadIdx.x == 0 && atomicAdd(ctr + 1, 1u) == (unsigned)nwg - 1u) { ctr[0] = 0u; ctr[1] = 0u; }
}

int main(int argc, char **argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 10;
    const int reps = 200;
    for (int rows : {6144, 24576, 131072}) {
        const int n = rows;
        float *a, *b; unsigned *ctr;
        CHK(hipMalloc(&a, n * 4)); CHK(hipMalloc(&b, n * 4)); CHK(hipMalloc(&ctr, 8));
        std::vector<float> h0(n, 0.0f), ref(n), got(n);
        hipStream_t st; CHK(hipStreamCreate(&st));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        // (a) separate launches in a graph
        const int nb = (n + 255) / 256;
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < phases; p++) hipLaunchKernelGGL(k_phase, dim3(nb), dim3(256), 0, st, (const float *)((p & 1) ? b : a), (p & 1) ? a : b, n);
        CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHK(hipMemcpy(a, h0.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemset(b, 0, n * 4));
        CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
        CHK(hipMemcpy(ref.data(), (phases & 1) ? b : a, n * 4, hipMemcpyDeviceToHost));
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) CHK(hipGraphLaunch(ge, st));
        CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%7d rows, %d phases: graph of launches      %6.2f us per phase\n", n, phases, ms * 1e3 / reps / phases);
        // (b), (c) persistent
        for (int variant = 0; variant < 2; variant++)
            for (int nwg : {8, 16, 32, 64, 128, 256}) {
                if (nwg * 256 > n * 2) continue;
                CHK(hipMemcpy(a, h0.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemset(b, 0, n * 4)); CHK(hipMemset(ctr, 0, 8));
                auto launch = [&]() {
                    if (variant == 0) hipLaunchKernelGGL(k_persistent<1>, dim3(nwg), dim3(256), 0, st, a, b, n, phases, ctr);
                    else hipLaunchKernelGGL(k_persistent<8>, dim3(nwg * 8), dim3(256), 0, st, a, b, n, phases, ctr);
                };
                if (variant == 1 && nwg > 32 * 4) continue;   // 32 CUs of an XCD, a few workgroups each
                launch(); CHK(hipStreamSynchronize(st));
                CHK(hipMemcpy(got.data(), (phases & 1) ? b : a, n * 4, hipMemcpyDeviceToHost));
                int bad = 0;
                for (int e = 0; e < n; e++) if (got[e] != ref[e]) bad++;
                CHK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; r++) launch();
                CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms, e0, e1));
                printf("   persistent, %3d workgroups %-12s %6.2f us per phase (%.1f us per launch)%s\n", nwg, variant ? "on one XCD" : "on all XCDs", ms * 1e3 / reps / phases, ms * 1e3 / reps, bad ? "  WRONG RESULT" : "");
            }
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
        CHK(hipFree(a)); CHK(hipFree(b)); CHK(hipFree(ctr));
    }
    return 0;
}
