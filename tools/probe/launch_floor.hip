// What does one more dependent launch cost inside a captured hipGraph on this box?  (the short levels are
// ~160 launches of 8-14 us each.)  Chains of N dependent kernels, captured from one stream and replayed: an empty kernel, a
// kernel of 256 blocks that touches one cache line each, and the same with a 384-byte argument block (the size of ConvArgs).
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/launch_floor.hip -o tools/probe/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { unsigned long long v[48]; };
__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0) p[blockIdx.x * 32] += 1.f; }
__global__ void k_touch_big(float* p, Big b) { if (threadIdx.x == 0) p[blockIdx.x * 32] += (float)b.v[blockIdx.x % 48]; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Big big = {};
    const int N = 200, REP = 20;
    for (int kind = 0; kind < 4; ++kind) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) {
            if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
            else if (kind == 1) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
            else if (kind == 2) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, d);
            else hipLaunchKernelGGL(k_touch_big, dim3(256), dim3(256), 0, st, d, big);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const char* names[] = {"empty kernel, 1 block x 64", "empty kernel, 256 blocks x 256", "256 blocks, one line each", "256 blocks, one line each, 384 B of arguments"};
        printf("%-48s %6.2f us per dependent launch (graph of %d, %d replays)\n", names[kind], ms * 1e3 / (N * REP), N, REP);
        // the same chain as plain stream launches
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r)
            for (int i = 0; i < N; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
                else if (kind == 1) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
                else if (kind == 2) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, d);
                else hipLaunchKernelGGL(k_touch_big, dim3(256), dim3(256), 0, st, d, big);
            }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-48s %6.2f us per launch, plain stream\n", "", ms * 1e3 / (N * REP));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
