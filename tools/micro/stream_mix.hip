// Micro-benchmark: what this box's HBM delivers for the byte mix of the stencil SpMV kernels, without any stencil --
// R float arrays read once, one byte array read once, W float arrays written once, float4 per lane, grid-stride.
// The filled-box SpMV roofline fractions in DESIGN.md are read against these rates as well as against the 8 TB/s datasheet peak.
//   hipcc --offload-arch=gfx950 -O3 stream_mix.hip -o stream_mix ; ./stream_mix [million elements] [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int R, int W>
struct Arrays { const float4 *in[R > 0 ? R : 1]; float4 *out[W > 0 ? W : 1]; const uint32_t *mask; };

template <int R, int W, bool MASK, int NT>
__global__ __launch_bounds__(256) void k_mix(Arrays<R, W> A, size_t n4) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        float4 v[R > 0 ? R : 1];
        uint32_t m = 0xffffffffu;
        if (MASK) m = A.mask[e];
#pragma unroll
        for (int a = 0; a < R; a++) { if (NT & 1) { const vf4 t = __builtin_nontemporal_load((const vf4 *)&A.in[a][e]); v[a] = make_float4(t.x, t.y, t.z, t.w); } else v[a] = A.in[a][e]; }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < R; a++) { s.x += v[a].x; s.y += v[a].y; s.z += v[a].z; s.w += v[a].w; }
        if (MASK && m == 0x12345678u) s.x = 1.f;
        if (W > 0) {
#pragma unroll
            for (int a = 0; a < W; a++) { if (NT & 2) { vf4 t; t.x = s.x; t.y = s.y; t.z = s.z; t.w = s.w; __builtin_nontemporal_store(t, (vf4 *)&A.out[a][e]); } else A.out[a][e] = s; }
        } else if (s.x == 123.456f) {
            A.out[0][e] = s;   // never: keeps the loads
        }
    }
}

template <int R, int W, bool MASK, int NT = 0>
static int run(const char *what, size_t n, int launches, int blocks) {
    Arrays<R, W> A;
    float *buf[R + W + 1];
    for (int a = 0; a < R + (W > 0 ? W : 1); a++) { CHK(hipMalloc(&buf[a], n * 4)); CHK(hipMemset(buf[a], 0, n * 4)); }
    uint8_t *mask; CHK(hipMalloc(&mask, n)); CHK(hipMemset(mask, 1, n));
    for (int a = 0; a < R; a++) A.in[a] = (const float4 *)buf[a];
    for (int a = 0; a < (W > 0 ? W : 1); a++) A.out[a] = (float4 *)buf[R + a];
    A.mask = (const uint32_t *)mask;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) k_mix<R, W, MASK, NT><<<blocks, 256>>>(A, n / 4);
    CHK(hipEventRecord(e0));
    for (int l = 0; l < launches; l++) k_mix<R, W, MASK, NT><<<blocks, 256>>>(A, n / 4);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n * (4.0 * (R + W) + (MASK ? 1.0 : 0.0)), us = ms * 1e3 / launches;
    printf("%-44s %6.1f M elements  %2d B/element  %8.1f us  %6.0f GB/s = %.3f of 8 TB/s\n", what, n / 1e6, 4 * (R + W) + (MASK ? 1 : 0), us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
    for (int a = 0; a < R + (W > 0 ? W : 1); a++) CHK(hipFree(buf[a]));
    CHK(hipFree(mask));
    return 0;
}

int main(int argc, char **argv) {
    const double M = argc > 1 ? atof(argv[1]) : 16.8;
    const int launches = argc > 2 ? atoi(argv[2]) : 30;
    const size_t n = ((size_t)(M * 1e6) / 1024) * 1024;
    for (int blocks : {2048, 8192}) {
        printf("-- %d blocks of 256\n", blocks);
        if (run<6, 0, false>("read only, 6 arrays", n, launches, blocks)) return 1;
        if (run<1, 1, false>("copy", n, launches, blocks)) return 1;
        if (run<5, 1, true>("pressure SpMV mix: 5 read + mask, 1 written", n, launches, blocks)) return 1;
        if (run<10, 3, true>("viscosity SpMV mix: 10 read + mask, 3 written", n, launches, blocks)) return 1;
        if (run<4, 4, false>("PCG update mix: 4 read, 4 written (x r s z)", n, launches, blocks)) return 1;
        if (run<5, 1, true, 2>("pressure mix, nontemporal stores", n, launches, blocks)) return 1;
        if (run<5, 1, true, 3>("pressure mix, nontemporal loads + stores", n, launches, blocks)) return 1;
        if (run<10, 3, true, 2>("viscosity mix, nontemporal stores", n, launches, blocks)) return 1;
        if (run<4, 4, false, 2>("update mix, nontemporal stores", n, launches, blocks)) return 1;
        if (run<4, 4, false, 3>("update mix, nontemporal loads + stores", n, launches, blocks)) return 1;
    }
    return 0;
}
