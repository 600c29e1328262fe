// Stand-alone reproduction attempt for the rocprofv3 --kernel-trace crash seen on long runs of the library (profiles/README.md): nothing of
// libflipv here, only the pattern of its PCG loops -- per "solve" a stream capture of `nodes` small kernel launches, the cached executable
// updated from the new capture (hipGraphExecUpdate) or re-instantiated, `replays` launches of it, one memcpy + synchronise per solve.
//   hipcc --offload-arch=gfx950 -O2 graph_replay_rocprof.hip -o graph_replay ; ./graph_replay [solves] [nodes] [replays] [vary topology 0|1]
//   rocprofv3 --kernel-trace --stats -d out -- ./graph_replay 3000 30 12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_axpy(float *p, int n, float a) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * a + 1.0f; }
int main(int argc, char **argv) {
    const int solves = argc > 1 ? atoi(argv[1]) : 3000, nodes = argc > 2 ? atoi(argv[2]) : 30, replays = argc > 3 ? atoi(argv[3]) : 12, vary = argc > 4 ? atoi(argv[4]) : 1;
    const int n = 1 << 16;
    float *d; int *flag, *h;
    CHK(hipMalloc(&d, n * sizeof(float))); CHK(hipMemset(d, 0, n * sizeof(float)));
    CHK(hipMalloc(&flag, 64)); CHK(hipMemset(flag, 0, 64)); CHK(hipHostMalloc((void **)&h, 64));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipGraphExec_t exec = nullptr;
    long launches = 0;
    for (int s = 0; s < solves; s++) {
        hipGraph_t g;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int q = 0; q < nodes + (vary ? s % 3 : 0); q++) k_axpy<<<n / 256, 256, 0, st>>>(d, n - (s & 7), 0.5f);   // topology changes now and then
        CHK(hipMemcpyAsync(h, flag, 16, hipMemcpyDeviceToHost, st));
        CHK(hipStreamEndCapture(st, &g));
        bool ok = false;
        if (exec) {
            hipGraphNode_t bad; hipGraphExecUpdateResult res;
            ok = hipGraphExecUpdate(exec, g, &bad, &res) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); CHK(hipGraphExecDestroy(exec)); exec = nullptr; }
        }
        if (!ok) CHK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        for (int r = 0; r < replays; r++) { CHK(hipGraphLaunch(exec, st)); launches += nodes; }
        CHK(hipStreamSynchronize(st));
        CHK(hipGraphDestroy(g));
    }
    printf("done: %d solves, %ld kernel launches through graph replays\n", solves, launches);
    return 0;
}
