// Probe: lane mapping of v_permlane16_swap_b32 and the half-wave sum built on it (k_node_update_rna).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf));
}
__device__ __forceinline__ float half_wave_sum2(float a, float b) {
    // (inline asm: this compiler folds the builtin's two results into one register - tools/ubench/permlane_probe.hip; the s_nops are the
    //  VALU-write -> permlane-read and permlane-write -> DPP-read wait states the compiler would have inserted)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float v = a + b;
    v += dpp_f32<0xB1, 0xf>(v);
    v += dpp_f32<0x4E, 0xf>(v);
    v += dpp_f32<0x141, 0xf>(v);
    v += dpp_f32<0x140, 0xf>(v);
    return v;
}
__global__ void k(unsigned* o, float* f) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    f[threadIdx.x] = half_wave_sum2((float)(threadIdx.x & 31), 100.f + (float)(threadIdx.x >> 5));
}
int main() {
    unsigned* o; float* f; hipMalloc(&o, 512); hipMalloc(&f, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, f);
    unsigned h[128]; float hf[64]; hipMemcpy(h, o, 512, hipMemcpyDeviceToHost); hipMemcpy(hf, f, 256, hipMemcpyDeviceToHost);
    printf("a':"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\nb':"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]);
    printf("\nsum2 (expect 496 on lanes 0-15 / 32-47, 3200 / 3232 on lanes 16-31 / 48-63):"); for (int i = 0; i < 64; ++i) printf(" %g", hf[i]); printf("\n");
    return 0;
}
