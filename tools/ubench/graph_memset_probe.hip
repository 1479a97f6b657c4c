// Probe: do hipMemsetAsync / hipMemcpyAsync(D2D) captured into a hipGraph write exactly the extent they write eagerly?
// (Round 3 replaced the runtime's memset / copy graph NODES in the captured training step by kernels after NaN losses at 16 RNAs and
// memory faults at 64; this checks the primitive itself with guard bytes around the target, at the sizes that step used.)
// Build: hipcc --offload-arch=gfx950 -O2 -o graph_memset_probe graph_memset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill(unsigned char* p, size_t n, unsigned char v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// first / last byte (relative to the target start) that differs from the guard value; -1 -1 if none
static void changed_range(const std::vector<unsigned char>& h, size_t guard, unsigned char gv, long long* lo, long long* hi) {
    *lo = -1; *hi = -1;
    for (size_t i = 0; i < h.size(); ++i) if (h[i] != gv) { if (*lo < 0) *lo = (long long)i - (long long)guard; *hi = (long long)i - (long long)guard; }
}

__global__ void set_u32(unsigned* p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void add_u32(unsigned* p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += v;
}
// ORDER test: a captured chain of [kernel: p = 7][memset node / copy node: p = 0][kernel: p += 1] triples on one stream.  Every word must end as 1:
// 8 = the memset ran before the first kernel, 0 = after the last one, 7 / other = lost or partial.
static int order_test(hipStream_t s, size_t words, int triples, bool use_copy, unsigned long long inst_flags = 0) {
    unsigned *buf, *zeros;
    CK(hipMalloc(&buf, words * 4 * triples)); CK(hipMalloc(&zeros, words * 4));
    CK(hipMemset(zeros, 0, words * 4));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int t = 0; t < triples; ++t) {
        unsigned* p = buf + (size_t)t * words;
        set_u32<<<256, 256, 0, s>>>(p, words, 7u);
        if (use_copy) CK(hipMemcpyAsync(p, zeros, words * 4, hipMemcpyDeviceToDevice, s)); else CK(hipMemsetAsync(p, 0, words * 4, s));
        add_u32<<<256, 256, 0, s>>>(p, words, 1u);
    }
    CK(hipStreamEndCapture(s, &g));
    if (inst_flags) CK(hipGraphInstantiateWithFlags(&ge, g, inst_flags)); else CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int bad = 0;
    std::vector<unsigned> h(words * triples);
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
        size_t wrong = 0; unsigned sample = 1;
        for (size_t i = 0; i < h.size(); ++i) if (h[i] != 1u) { ++wrong; sample = h[i]; }
        printf("order test%s: %3d x [kernel, %s node, kernel], %9zu words each, replay %d: %zu wrong words%s\n", inst_flags ? " (instantiated AutoFreeOnLaunch, as torch.cuda.graph does)" : "", triples, use_copy ? "copy" : "memset", words, rep, wrong,
               wrong ? (sample == 8u ? " (value 8: node ran BEFORE its predecessor)" : sample == 0u ? " (value 0: node ran AFTER its successor)" : " (other)") : "");
        bad += wrong != 0;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(buf)); CK(hipFree(zeros));
    return bad;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t sizes[] = {256, 512, 1026, 4096 + 2, 573440, 4587520, 14151680, 17203200, 68812800, 275251200};
    const size_t guard = 8u << 20;
    int bad = 0;
    for (size_t n : sizes) {
        unsigned char* buf; CK(hipMalloc(&buf, n + 2 * guard));
        unsigned char* src; CK(hipMalloc(&src, n));
        std::vector<unsigned char> h(n + 2 * guard);
        for (int mode = 0; mode < 4; ++mode) {          // 0 eager memset, 1 captured memset, 2 eager D2D copy, 3 captured D2D copy
            fill<<<1024, 256, 0, s>>>(buf, n + 2 * guard, 0xAB);
            fill<<<1024, 256, 0, s>>>(src, n, 0x00);
            CK(hipStreamSynchronize(s));
            if (mode & 1) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                if (mode == 1) CK(hipMemsetAsync(buf + guard, 0, n, s)); else CK(hipMemcpyAsync(buf + guard, src, n, hipMemcpyDeviceToDevice, s));
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            } else {
                if (mode == 0) CK(hipMemsetAsync(buf + guard, 0, n, s)); else CK(hipMemcpyAsync(buf + guard, src, n, hipMemcpyDeviceToDevice, s));
                CK(hipStreamSynchronize(s));
            }
            CK(hipMemcpy(h.data(), buf, n + 2 * guard, hipMemcpyDeviceToHost));
            long long lo, hi; changed_range(h, guard, 0xAB, &lo, &hi);
            const bool ok = lo == 0 && hi == (long long)n - 1;
            size_t holes = 0; if (ok) for (size_t i = 0; i < n; ++i) holes += h[guard + i] != 0;
            printf("n = %10zu  %-22s wrote bytes [%lld, %lld]  %s%s\n", n, mode == 0 ? "eager memset" : mode == 1 ? "CAPTURED memset node" : mode == 2 ? "eager D2D copy" : "CAPTURED D2D copy node",
                   lo, hi, ok && !holes ? "ok" : "WRONG EXTENT", holes ? " (holes inside)" : "");
            bad += !(ok && !holes);
        }
        CK(hipFree(buf)); CK(hipFree(src));
    }
    printf(bad ? "%d case(s) wrong\n" : "all extents exact\n", bad);
    int obad = 0;
    for (size_t words : {(size_t)64, (size_t)2240, (size_t)2048, (size_t)8192, (size_t)143360, (size_t)4300800})
        for (int copy = 0; copy < 2; ++copy) obad += order_test(s, words, words > 1000000 ? 40 : 200, copy != 0);
    printf(obad ? "ORDER VIOLATIONS in %d run(s)\n" : "graph nodes kept stream order\n", obad);
    int fbad = 0;
    for (size_t words : {(size_t)64, (size_t)2240, (size_t)143360, (size_t)3537920})
        for (int copy = 0; copy < 2; ++copy) fbad += order_test(s, words, words > 1000000 ? 4 : 50, copy != 0, hipGraphInstantiateFlagAutoFreeOnLaunch);
    printf(fbad ? "WRONG with hipGraphInstantiateFlagAutoFreeOnLaunch in %d run(s)\n" : "AutoFreeOnLaunch: fine\n", fbad);
    return 0;
}
