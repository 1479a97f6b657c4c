// Micro-benchmark: what HBM rate does the fused kernel's e traffic pattern reach when nothing else is going on?
//   A: plain copy, one 16-byte element per thread, huge grid.
//   B: the kernel's pattern - 256 workgroups x 8 waves, each wave walks 8 KiB blocks (8 x 1 KiB loads, then 8 x 1 KiB
//      stores of the same block), XCD-chunked block order, DEPTH blocks in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void copy_plain(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

template <int DEPTH, bool INPLACE>
__global__ void __launch_bounds__(512) copy_blocks(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int nblocks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = (nblocks + 7) >> 3, x = blockIdx.x & 7;
    const int end = min(nblocks, (x + 1) * chunk), stride = (gridDim.x >> 3) * 8;
    int blk = x * chunk + (blockIdx.x >> 3) * 8 + wave;
    u32x4 buf[DEPTH][8];
    int cur[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        cur[d] = blk + d * stride;
        if (cur[d] < end) {
#pragma unroll
            for (int s = 0; s < 8; ++s) buf[d][s] = src[(size_t)cur[d] * 512 + s * 64 + lane];
        }
    }
    while (cur[0] < end) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (cur[d] < end) {
                u32x4* o = (INPLACE ? const_cast<u32x4*>(src) : dst) + (size_t)cur[d] * 512 + lane;
#pragma unroll
                for (int s = 0; s < 8; ++s) { u32x4 v = buf[d][s]; v[0] += 1u; o[s * 64] = v; }
                cur[d] += DEPTH * stride;
                if (cur[d] < end) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) buf[d][s] = src[(size_t)cur[d] * 512 + s * 64 + lane];
                }
            }
        }
    }
}

template <typename F>
static double time_us(F&& launch, int reps = 20) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}

int main() {
    const int nblocks = 30559;                       // C2: one 8 KiB block of e per residue
    const size_t bytes = (size_t)nblocks * 8192, n16 = bytes / 16;
    u32x4 *src, *dst;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&dst, bytes);
    (void)hipMemset(src, 1, bytes); (void)hipMemset(dst, 0, bytes);
    double t;
    t = time_us([&] { hipLaunchKernelGGL(copy_plain, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, src, dst, n16); });
    printf("A plain copy            : %7.1f us  %6.2f TB/s (read+write)\n", t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { hipLaunchKernelGGL((copy_blocks<1, false>), dim3(256), dim3(512), 0, 0, src, dst, nblocks); });
    printf("B blocks, depth 1       : %7.1f us  %6.2f TB/s\n", t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { hipLaunchKernelGGL((copy_blocks<2, false>), dim3(256), dim3(512), 0, 0, src, dst, nblocks); });
    printf("B blocks, depth 2       : %7.1f us  %6.2f TB/s\n", t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { hipLaunchKernelGGL((copy_blocks<4, false>), dim3(256), dim3(512), 0, 0, src, dst, nblocks); });
    printf("B blocks, depth 4       : %7.1f us  %6.2f TB/s\n", t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { hipLaunchKernelGGL((copy_blocks<1, true>), dim3(256), dim3(512), 0, 0, src, dst, nblocks); });
    printf("B in place, depth 1     : %7.1f us  %6.2f TB/s\n", t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { hipLaunchKernelGGL((copy_blocks<2, true>), dim3(256), dim3(512), 0, 0, src, dst, nblocks); });
    printf("B in place, depth 2     : %7.1f us  %6.2f TB/s\n", t, 2.0 * bytes / t / 1e6);
    t = time_us([&] { hipLaunchKernelGGL((copy_blocks<2, true>), dim3(512), dim3(512), 0, 0, src, dst, nblocks); });
    printf("B in place, depth 2, 2 WG/CU: %7.1f us  %6.2f TB/s\n", t, 2.0 * bytes / t / 1e6);
    return 0;
}
