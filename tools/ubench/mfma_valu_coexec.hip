// Micro-benchmark: how far the matrix pipe and the vector ALU of ONE SIMD overlap when its co-resident waves run a
// "chain of 8 dependent v_mfma_f32_32x32x16_f16, then ~64 packed-f16 vector instructions on that tile" stream - the shape of the fused
// ResMPNN kernel's work - in different arrangements.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coexec mfma_valu_coexec.hip
//   mode 0  every wave: MFMA chains only          mode 1  every wave: vector blocks only
//   mode 2  waves 0..3 MFMA only, 4..7 vector only (perfect role split on each SIMD)
//   mode 3  every wave alternates [8 MFMA][64 VALU], all waves start in the same phase
//   mode 4  the same, the second wave of each SIMD starts with the vector phase (anti-phase)
//   mode 5  every wave: [1 MFMA, 8 VALU] x 8 (fine interleave inside the wave)
//   mode 7-9  mode 3 with s_setprio flips per phase
//   mode 6  mode 3 with a workgroup barrier after every phase and the second half of the waves shifted by one phase (ping-pong)
// Reported: cycles per (chain + block) unit per wave, and per SIMD (= per wave / waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define V4 "v_pk_fma_f16 %0, %0, %4, %5\n\tv_pk_fma_f16 %1, %1, %4, %5\n\tv_pk_fma_f16 %2, %2, %4, %5\n\tv_pk_fma_f16 %3, %3, %4, %5\n\t"
#define VALU8() asm volatile(V4 V4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c))
#define VALU64() do { VALU8(); VALU8(); VALU8(); VALU8(); VALU8(); VALU8(); VALU8(); VALU8(); } while (0)
#define MFMA1() T = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, T, 0, 0, 0)
#define PIN() asm volatile("" : "+v"(T))
#define CHAIN8() do { MFMA1(); MFMA1(); MFMA1(); MFMA1(); MFMA1(); MFMA1(); MFMA1(); MFMA1(); PIN(); } while (0)
// the vector phase starts by reading the tile (the MFMA -> VALU hazard is part of the real stream)
#define READ_T() do { a0 += T[0]; a1 += T[5]; PIN(); } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, float* sink, float seed, int units) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5f, c = seed * 0.25f;
    f32x16 T;
    for (int i = 0; i < 16; ++i) T[i] = seed * i;
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(seed * 0.01f * (i + threadIdx.x % 7)); fb[i] = (_Float16)(seed * 0.02f * (i + 1)); }
    const bool second = wave >= nw / 2;                      // (waves w and w + nw/2 share a SIMD)
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) { for (int u = 0; u < units; ++u) CHAIN8(); }
    if (MODE == 1) { for (int u = 0; u < units; ++u) VALU64(); }
    if (MODE == 2) { if (!second) { for (int u = 0; u < units; ++u) CHAIN8(); } else { for (int u = 0; u < units; ++u) VALU64(); } }
    if (MODE == 3) { for (int u = 0; u < units; ++u) { CHAIN8(); READ_T(); VALU64(); } }
    if (MODE == 4) {
        if (second) VALU64();
        for (int u = 0; u < units; ++u) { CHAIN8(); READ_T(); VALU64(); }
    }
    if (MODE == 5) {
        for (int u = 0; u < units; ++u) {
            MFMA1(); VALU8(); MFMA1(); VALU8(); MFMA1(); VALU8(); MFMA1(); VALU8(); MFMA1(); VALU8(); MFMA1(); VALU8(); MFMA1(); VALU8(); MFMA1(); PIN(); READ_T(); VALU8();
        }
    }
    if (MODE == 7) {      // mode 3 + the wave raises its priority for the MFMA phase: an older wave in its vector phase no longer starves the partner's MFMAs
        for (int u = 0; u < units; ++u) { __builtin_amdgcn_s_setprio(1); CHAIN8(); __builtin_amdgcn_s_setprio(0); READ_T(); VALU64(); }
    }
    if (MODE == 8) {      // mode 7, priority dropped only after the tile has been read (the hazard wait of the chain's last MFMA stays prioritised)
        for (int u = 0; u < units; ++u) { __builtin_amdgcn_s_setprio(1); CHAIN8(); READ_T(); __builtin_amdgcn_s_setprio(0); VALU64(); }
    }
    if (MODE == 9) {      // mode 7 with the opposite sign: the vector phase is prioritised
        for (int u = 0; u < units; ++u) { __builtin_amdgcn_s_setprio(0); CHAIN8(); __builtin_amdgcn_s_setprio(1); READ_T(); VALU64(); }
    }
    if (MODE == 6) {
        if (second) __builtin_amdgcn_s_barrier();
        for (int u = 0; u < units; ++u) {
            CHAIN8(); __builtin_amdgcn_s_barrier();
            READ_T(); VALU64(); __builtin_amdgcn_s_barrier();
        }
        if (!second) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x % 64 == 0) out[blockIdx.x * nw + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + T[0] + T[7];
}

template <int MODE>
static void run(const char* name, int waves_per_simd) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    const int units = 512, threads = 256 * waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, sink, 1.0f, units);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * threads / 64);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0; for (auto v : h) { s += (double)v; if ((double)v > mx) mx = (double)v; }
    const double per_unit = s / h.size() / units;
    printf("%-58s %dw/SIMD: %7.1f cyc/unit/wave  %7.1f cyc/unit/SIMD  (max wave %7.1f)\n", name, waves_per_simd, per_unit, per_unit / waves_per_simd, mx / units);
    hipFree(out); hipFree(sink);
}

int main() {
    run<0>("0 MFMA chains only (8 dependent 32x32x16)", 1);
    run<0>("0 MFMA chains only", 2);
    run<1>("1 vector blocks only (64 v_pk_fma_f16, 4 chains)", 1);
    run<1>("1 vector blocks only", 2);
    run<1>("1 vector blocks only", 3);
    run<2>("2 role split: waves 0-3 MFMA, 4-7 vector (unit = either)", 2);
    run<3>("3 [8 MFMA][64 VALU] per wave, same phase", 1);
    run<3>("3 [8 MFMA][64 VALU] per wave, same phase", 2);
    run<3>("3 [8 MFMA][64 VALU] per wave, same phase", 3);
    run<4>("4 ... second wave of a SIMD starts in the vector phase", 2);
    run<5>("5 [1 MFMA, 8 VALU] x 8 per wave", 1);
    run<5>("5 [1 MFMA, 8 VALU] x 8 per wave", 2);
    run<6>("6 ping-pong: barrier after every phase, halves shifted", 2);
    run<7>("7 mode 3 + s_setprio 1 during the MFMA phase", 2);
    run<7>("7 mode 3 + s_setprio 1 during the MFMA phase", 3);
    run<8>("8 mode 7, priority kept through the tile read", 2);
    run<8>("8 mode 7, priority kept through the tile read", 3);
    run<9>("9 mode 3 + s_setprio 1 during the VECTOR phase", 2);
    return 0;
}
