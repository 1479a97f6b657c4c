// Micro-benchmark: what one more kernel in a chain of dependent launches costs (a captured hipGraph of N launches on one stream, replayed):
//   empty kernel; one workgroup that loads a word and stores it back; 256 workgroups doing the same.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_touch(int* p) { if (threadIdx.x == 0) p[blockIdx.x] = p[blockIdx.x] + 1; }
static double run(int kind, int n, int* buf) {
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) {
        if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
        else if (kind == 1) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, buf);
        else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, buf);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / (10.0 * n);
}
int main() {
    int* buf; hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096);
    printf("empty kernel:                         %.2f us per launch (graph of 200)\n", run(0, 200, buf));
    printf("1 workgroup, load + store:            %.2f us per launch\n", run(1, 200, buf));
    printf("256 workgroups x 256, load + store:   %.2f us per launch\n", run(2, 200, buf));
    return 0;
}
