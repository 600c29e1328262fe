// Micro-benchmark: rate of LDS atomic adds on gfx950 by operand type (ds_add_f32 / ds_add_u32 / ds_add_u64) and by how the lanes'
// addresses collide.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct casf { float v; };   // float added through a compare-and-swap loop on the word (ds_cmpst_rtn_b32) instead of ds_add_f32
__device__ __forceinline__ void lds_add(float *a, float v) { atomicAdd(a, v); }
__device__ __forceinline__ void lds_add(unsigned *a, unsigned v) { atomicAdd(a, v); }
__device__ __forceinline__ void lds_add(unsigned long long *a, unsigned long long v) { atomicAdd(a, v); }
__device__ __forceinline__ void lds_add(casf *a, casf v) {
    unsigned *w = (unsigned *)a;
    unsigned old = *w, assumed;
    do {
        assumed = old;
        old = atomicCAS(w, assumed, __float_as_uint(__uint_as_float(assumed) + v.v));
    } while (old != assumed);
}
template <typename T> __device__ __forceinline__ T one() { return (T)1; }
template <> __device__ __forceinline__ casf one<casf>() { return casf{1.0f}; }
template <typename T> __device__ __forceinline__ T zero() { return (T)0; }
template <> __device__ __forceinline__ casf zero<casf>() { return casf{0.0f}; }
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k(int iters, T *out) {
    __shared__ T sh[8192];
    for (int e = threadIdx.x; e < 8192; e += 256) sh[e] = zero<T>();
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        int a;
        if (MODE == 0) a = (lane + 64 * wv + 256 * (it & 31)) & 8191;          // distinct banks, distinct addresses
        else if (MODE == 1) a = (h >> 8) & 8191;                                // random addresses
        else if (MODE == 2) a = ((lane >> 3) + 8 * wv + 64 * (it & 127)) & 8191; // 8 lanes per address
        else a = (wv + 4 * (it & 2047)) & 8191;                                  // all 64 lanes one address
        lds_add(&sh[a], one<T>());
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}
template <typename T, int MODE>
void run(const char *name) {
    T *out; hipMalloc((void **)&out, 4096 * sizeof(T));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096, blocks = 2048;
    hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, 16, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)iters * blocks * 256;
    printf("%-10s mode %d: %8.3f ms  %8.1f G lane-atomics/s  (%.2f lane-atomics per CU-cycle at 2.4 GHz, 256 CUs)\n", name, MODE, ms, ops / ms / 1e6, ops / (ms * 1e-3) / (256 * 2.4e9));
    hipFree(out);
}
int main() {
    run<float, 0>("f32"); run<float, 1>("f32"); run<float, 2>("f32"); run<float, 3>("f32");
    run<casf, 0>("f32 cas"); run<casf, 1>("f32 cas"); run<casf, 2>("f32 cas"); run<casf, 3>("f32 cas");
    run<unsigned, 0>("u32"); run<unsigned, 1>("u32"); run<unsigned, 2>("u32"); run<unsigned, 3>("u32");
    run<unsigned long long, 0>("u64"); run<unsigned long long, 1>("u64"); run<unsigned long long, 2>("u64"); run<unsigned long long, 3>("u64");
    return 0;
}
