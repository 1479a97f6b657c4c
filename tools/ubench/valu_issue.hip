// Micro-benchmark: issue cost (cycles per instruction per wave) of the VALU instructions the fused kernel leans on,
// with 1, 2 and 3 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(INSTR) \
    unsigned long long t0, t1; \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    for (int it = 0; it < 64; ++it) { REP8(REP8(INSTR)) } \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;

template <int OP>
__global__ void k(unsigned long long* out, float* sink, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5f, c = seed * 0.25f;
    if (OP == 0) { BODY(asm volatile("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 1) { BODY(asm volatile("v_pk_fma_f16 %0, %4, %5, %0\n\tv_pk_fma_f16 %1, %4, %5, %1\n\tv_pk_fma_f16 %2, %4, %5, %2\n\tv_pk_fma_f16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 2) { BODY(asm volatile("v_pk_mul_f16 %0, %4, %0\n\tv_pk_mul_f16 %1, %4, %1\n\tv_pk_mul_f16 %2, %4, %2\n\tv_pk_mul_f16 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 3) { BODY(asm volatile("v_cvt_pk_f16_f32 %0, %4, %0\n\tv_cvt_pk_f16_f32 %1, %4, %1\n\tv_cvt_pk_f16_f32 %2, %4, %2\n\tv_cvt_pk_f16_f32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 4) { BODY(asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[0,1,0]\n\tv_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[0,1,0]\n\tv_fma_mix_f32 %2, %4, %5, %2 op_sel_hi:[0,1,0]\n\tv_fma_mix_f32 %3, %4, %5, %3 op_sel_hi:[0,1,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 5) { BODY(asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 6) { BODY(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %0\n\tv_cvt_pk_bf16_f32 %1, %4, %1\n\tv_cvt_pk_bf16_f32 %2, %4, %2\n\tv_cvt_pk_bf16_f32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 7) { BODY(asm volatile("v_pk_fma_f16 %0, %0, %5, %4\n\tv_pk_fma_f16 %0, %0, %5, %4\n\tv_pk_fma_f16 %0, %0, %5, %4\n\tv_pk_fma_f16 %0, %0, %5, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }   // dependent chain
    if (OP == 8) { BODY(asm volatile("v_pk_min_f16 %0, %4, %0\n\tv_pk_min_f16 %1, %4, %1\n\tv_pk_min_f16 %2, %4, %2\n\tv_pk_min_f16 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 9) { BODY(asm volatile("v_pk_fma_f16 %0, %4, %5, %0 clamp\n\tv_pk_fma_f16 %1, %4, %5, %1 clamp\n\tv_pk_fma_f16 %2, %4, %5, %2 clamp\n\tv_pk_fma_f16 %3, %4, %5, %3 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 10) { BODY(asm volatile("v_fma_f32 %0, %0, %5, %4\n\tv_fma_f32 %0, %0, %5, %4\n\tv_fma_f32 %0, %0, %5, %4\n\tv_fma_f32 %0, %0, %5, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }   // dependent f32 chain
}

template <int OP>
static void run(const char* name) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    printf("%-34s", name);
    for (int waves_per_simd = 1; waves_per_simd <= 3; ++waves_per_simd) {
        int threads = 256 * waves_per_simd;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, sink, 1.0f);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, sink, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        double per_instr = s / h.size() / (64.0 * 64 * 4);
        printf("  %dw/SIMD: %5.2f cyc/instr/wave (%5.2f SIMD cyc/instr)", waves_per_simd, per_instr, per_instr / waves_per_simd);
    }
    printf("\n");
    hipFree(out); hipFree(sink);
}

int main() {
    run<0>("v_fma_f32 (independent)");
    run<10>("v_fma_f32 (dependent chain)");
    run<1>("v_pk_fma_f16 (independent)");
    run<7>("v_pk_fma_f16 (dependent chain)");
    run<9>("v_pk_fma_f16 clamp");
    run<2>("v_pk_mul_f16");
    run<8>("v_pk_min_f16");
    run<3>("v_cvt_pk_f16_f32");
    run<6>("v_cvt_pk_bf16_f32");
    run<4>("v_fma_mix_f32");
    run<5>("v_exp_f32");
    return 0;
}
