// Micro-benchmark: cost of a hand-written grid-wide barrier (agent-scope atomics + fences) on gfx950, with a
// cross-XCD visibility check.  hipcc --offload-arch=gfx950 -O3 gridbar_bench.hip -o gridbar_bench ; ./gridbar_bench [blocks] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifndef GROUP
#define GROUP 32
#endif
// two-level arrive counters (GROUP blocks per first-level counter, one 128-byte line each), a separate release flag that
// everybody polls: the polls do not disturb the arrive atomics.   cnt[0]: flag, cnt[32]: top counter, cnt[64 + 32 g]: group g
__device__ __forceinline__ void grid_barrier(unsigned *cnt, unsigned nblocks, unsigned &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned g = blockIdx.x / GROUP, ngroups = (nblocks + GROUP - 1) / GROUP;
        const unsigned gsize = min((unsigned)GROUP, nblocks - g * GROUP);
        const unsigned old = __hip_atomic_fetch_add(cnt + 64 + 32 * g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * gsize) {
            const unsigned o2 = __hip_atomic_fetch_add(cnt + 32, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == gen * ngroups) __hip_atomic_store(cnt, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// mode 0: barriers only; mode 1: every block writes 4 KB, barrier, reads another block's 4 KB and checks it
__global__ __launch_bounds__(256) void k_bar(unsigned *cnt, float *buf, int iters, int mode, unsigned *errors) {
    unsigned gen = 0;
    const unsigned nb = gridDim.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; it++) {
        if (mode) {
            float4 v = make_float4((float)(it * 1000 + blockIdx.x), 1.f, 2.f, 3.f);
            reinterpret_cast<float4 *>(buf)[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
        }
        grid_barrier(cnt, nb, gen);
        if (mode) {
            const unsigned other = (blockIdx.x + 37u) % nb;
            const float4 v = reinterpret_cast<const float4 *>(buf)[(size_t)other * 256 + threadIdx.x];
            if (v.x != (float)(it * 1000 + other)) bad++;
            grid_barrier(cnt, nb, gen);   // nobody overwrites before everybody has read
        }
    }
    if (bad) atomicAdd(errors, bad);
}

int main(int argc, char **argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 1000;
    unsigned *cnt, *err; float *buf;
    CHK(hipMalloc(&cnt, 65536)); CHK(hipMalloc(&err, 4)); CHK(hipMalloc(&buf, (size_t)blocks * 4096));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipMemset(cnt, 0, 65536)); CHK(hipMemset(err, 0, 4));
            void *args[] = {&cnt, &buf, &iters, &mode, &err};
            CHK(hipEventRecord(e0));
            CHK(hipLaunchCooperativeKernel((const void *)k_bar, dim3(blocks), dim3(256), args, 0, 0));
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h; CHK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            const int nbar = iters * (mode ? 2 : 1);
            printf("blocks %d mode %d: %.3f ms for %d barriers = %.2f us/barrier, errors %u\n", blocks, mode, ms, nbar, 1000.0 * ms / nbar, h);
        }
    }
    return 0;
}
