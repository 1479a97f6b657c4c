// bf16 MFMA fast path for gfx950 - see kernels_bf16.h.
#include "kernels_bf16.h"

__device__ __forceinline__ bf16_t f2bf(float x) {            // round-to-nearest-even, NaN kept
    return __builtin_bit_cast(bf16_t, (__bf16)x);
}
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

__global__ void k_convert_rows_bf16(const float* __restrict__ src, int ld_src, int rows, int cols, int cols_pad,
                                    bf16_t* __restrict__ dst) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * cols_pad) return;
    int r = id / cols_pad, c = id - r * cols_pad;
    dst[id] = c < cols ? f2bf(src[(size_t)r * ld_src + c]) : (bf16_t)0;
}
void launch_convert_rows_bf16(const float* src, int ld_src, int rows, int cols, int cols_pad, bf16_t* dst, hipStream_t s) {
    int total = rows * cols_pad;
    hipLaunchKernelGGL(k_convert_rows_bf16, dim3((total + 255) / 256), dim3(256), 0, s, src, ld_src, rows, cols, cols_pad, dst);
}

__global__ void k_bf16_to_f32(const bf16_t* __restrict__ src, float* __restrict__ dst, const int* __restrict__ ntot, int per_row) {
    size_t n = (size_t)(*ntot) * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = bf2f(src[i]);
}
__global__ void k_f32_to_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, const int* __restrict__ ntot, int per_row) {
    size_t n = (size_t)(*ntot) * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}
static unsigned conv_grid(size_t max_elems) { size_t g = (max_elems + 255) / 256; return (unsigned)(g < 8192 ? (g ? g : 1) : 8192); }
void launch_bf16_to_f32(const bf16_t* src, float* dst, size_t max_elems, const int* ntot, int per_row, hipStream_t s) {
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(conv_grid(max_elems)), dim3(256), 0, s, src, dst, ntot, per_row);
}
void launch_f32_to_bf16(const float* src, bf16_t* dst, size_t max_elems, const int* ntot, int per_row, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_to_bf16, dim3(conv_grid(max_elems)), dim3(256), 0, s, src, dst, ntot, per_row);
}

// ---- heavy kernels: filled in below (first bring-up uses the f32 path only) -----------------
void launch_build_mlp_image(const float*, int, const float*, int, const float*, bf16_t*, float*, hipStream_t) {}
void launch_build_embed_image(const float*, const float*, bf16_t*, hipStream_t) {}
void launch_gemm_bf16(const int*, int, const float*, int, int, const float*, int, int, const bf16_t*, const float*, int, int,
                      const float*, int, float*, int, hipStream_t) {}
void launch_edge_embed_bf16(const PackInfo&, int, const float*, const int*, const bf16_t*, const float*, const float*, bf16_t*, hipStream_t) {}
void launch_mpnn_bf16(const PackInfo&, int, bool, bool, const int*, bf16_t*, const float*, const float*, MpnnWB, MpnnWB,
                      const float*, float*, float*, hipStream_t) {}
