// Round trip of "a kernel produced a flag, the host needs it": (a) hipMemcpyAsync of 4 bytes to pinned memory + hipStreamSynchronize, (b) a one-wave kernel that stores the
// value and a sequence word into mapped host memory, the host spinning on the sequence word (flipv_internal.h: fv_read_small).  Also the gap a following kernel sees.
// hipcc -O3 --offload-arch=gfx950 small_read_latency.hip -o small_read_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(int *flag, float *buf, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) buf[i] = buf[i] * 1.0001f + 1.0f; if (i == 0) *flag += 1; }
__global__ void k_publish(const int *src, int *host, int *seq) {
    if (threadIdx.x == 0) __hip_atomic_store(host + 8, *src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (threadIdx.x == 0) { const int s = *seq + 1; *seq = s; __hip_atomic_store(host, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    int *dflag, *dseq, *hpin, *hmap, *dmap; float *buf; const int n = 1 << 20;
    CK(hipMalloc(&dflag, 4)); CK(hipMalloc(&dseq, 4)); CK(hipMalloc(&buf, n * 4)); CK(hipMemset(dflag, 0, 4)); CK(hipMemset(dseq, 0, 4)); CK(hipMemset(buf, 0, n * 4));
    CK(hipHostMalloc(&hpin, 64)); CK(hipHostMalloc(&hmap, 256, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void **)&dmap, hmap, 0));
    hmap[0] = 0;
    const int reps = 2000;
    for (int mode = 0; mode < 4; mode++) {
        CK(hipDeviceSynchronize());
        int seq = __atomic_load_n(hmap, __ATOMIC_ACQUIRE);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, dflag, buf, n);
            if (mode == 0) { CK(hipMemcpyAsync(hpin, dflag, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
            else if (mode == 1) { hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, dflag, dmap, dseq); seq++; while (__atomic_load_n(hmap, __ATOMIC_ACQUIRE) - seq < 0) __builtin_ia32_pause(); }
            else if (mode == 2) { hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, dflag, dmap, dseq); seq++; CK(hipStreamSynchronize(st)); }
            else { CK(hipStreamSynchronize(st)); }
        }
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("%-64s %7.2f us per round trip\n", mode == 0 ? "kernel + 4-byte hipMemcpyAsync D2H + hipStreamSynchronize" : mode == 1 ? "kernel + publish kernel + host spin on mapped word" : mode == 2 ? "kernel + publish kernel + hipStreamSynchronize" : "kernel + hipStreamSynchronize (no read at all)", us);
    }
    return 0;
}
