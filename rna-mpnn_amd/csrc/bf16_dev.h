// Device-side helpers shared by the bf16 / f16 MFMA kernels (kernels_bf16.hip, kernels_mpnn.hip): vector types, conversions,
// the packed-f16 activation arithmetic, LDS staging of weight images, and the channel orders of a 32 x 32 accumulator tile.
#pragma once
#include "rnampnn_internal.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

static constexpr float kSEPS = 1.0e-6f;
#define RN_E_F16 1                // the edge tensor e is stored as f16 of a e (bf16 in rounds 1 - 2)
#define RN_P_F16 1                // the P / Q tables are f16
#define RN_PHI_DEG 4              // coefficients of Phi's polynomial in the per-edge kernels

__device__ __forceinline__ bf16_t f2bf(float x) { return __builtin_bit_cast(bf16_t, (__bf16)x); }   // RNE, NaN kept
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ unsigned pack2(float a, float b) {        // one v_cvt_pk_bf16_f32
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_bf(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// GELU for the bf16 path: x * sigmoid(x * (c0 + c1 x^2)), coefficients minimax-fitted to the exact
// erf form (max |err| 2.7e-4 at |x| ~ 2-3, 15-30x below the bf16 rounding of the result there);
// monotone argument, so no clamp: 7 VALU instructions, 2 of them transcendental.
// The f32 path and the node-level GEMM epilogues keep erff().
__device__ __forceinline__ float gelu_fast(float x) {
    float t = x * x;
    float p = fmaf(t, -0.10012571f, -2.3087657f);          // -log2(e) * (c0 + c1 t), c0 = 1.60031416, c1 = 0.06940179
    float ex = __builtin_amdgcn_exp2f(x * p);              // exp(-x (c0 + c1 t))
    return x * __builtin_amdgcn_rcpf(1.0f + ex);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// Packed-f16 activation arithmetic of the fused ResMPNN kernel: the kernel is VALU-issue
// bound on GELU, and v_pk_*_f16 evaluates two activations per instruction with no transcendental.
//   GELU(x) = x * Phi(x),  Phi(x) ~ clamp01(0.5 + x * q(min(x^2, S)))   (odd polynomial for Phi - 1/2; the clamp of
//   x^2 makes the argument monotone, so beyond sqrt(S) the form saturates to exactly 0 / 1)
// Hidden activations then stay in f16 (11-bit significand, finer than the bf16 they replace) and feed
// v_mfma_f32_32x32x16_f16; where the result is consumed in f32 (residual, mean) the final x * Phi is a mixed-precision
// FMA on the f32 accumulator, so only Phi itself is rounded to f16.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ f16x2 cvt_h2(float a, float b) {          // one v_cvt_pk_f16_f32 (RNE)
    f32x2 v = {a, b};
    return __builtin_convertvector(v, f16x2);
}
__device__ __forceinline__ f16x2 h2(float v) { return f16x2{(_Float16)v, (_Float16)v}; }
__device__ __forceinline__ f16x2 phi2(f16x2 x) {
    f16x2 s = __builtin_elementwise_min(x * x, h2(9.5f));
    f16x2 q = __builtin_elementwise_fma(s, h2(-0.00017380498f), h2(0.0048129941f));
    q = __builtin_elementwise_fma(q, s, h2(-0.05394074f));
    q = __builtin_elementwise_fma(q, s, h2(0.38869277f));
    f16x2 p = __builtin_elementwise_fma(x, q, h2(0.5f));
    return __builtin_elementwise_min(__builtin_elementwise_max(p, h2(0.f)), h2(1.f));
}
__device__ __forceinline__ unsigned gelu_h2(float a, float b) {      // two activations -> packed f16 GELU
    f16x2 x = cvt_h2(a, b);
    return __builtin_bit_cast(unsigned, x * phi2(x));
}
// Four activations at a time: the two packed chains are independent, so the compiler interleaves them and the
// one-wait-state hazard between dependent VOP3P instructions (an s_nop 0 = 4 issue cycles each) disappears.
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ f16x4 h4(float v) { return f16x4{(_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v}; }
__device__ __forceinline__ f16x4 cvt_h4(float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    return __builtin_convertvector(v, f16x4);
}
__device__ __forceinline__ f16x4 phi4(f16x4 x) {
    f16x4 s = __builtin_elementwise_min(x * x, h4(9.5f));
    f16x4 q = __builtin_elementwise_fma(s, h4(-0.00017380498f), h4(0.0048129941f));
    q = __builtin_elementwise_fma(q, s, h4(-0.05394074f));
    q = __builtin_elementwise_fma(q, s, h4(0.38869277f));
    f16x4 p = __builtin_elementwise_fma(x, q, h4(0.5f));
    return __builtin_elementwise_min(__builtin_elementwise_max(p, h4(0.f)), h4(1.f));
}
// Phi with FIVE coefficients (max |x Phi - gelu| 1.2e-3 against 3.1e-3): one more packed fma per four activations.  Used where an error is
// amplified downstream (the node-level FFN chains feed a GraphNormalization, which removes the common part of its input).
__device__ __forceinline__ f16x4 phi4_hi(f16x4 x) {
    f16x4 s = __builtin_elementwise_min(x * (x * h4(0.25f)), h4(11.5f * 0.25f));
    f16x4 q = __builtin_elementwise_fma(s, h4(1.066712254e-05f * 256.f), h4(-0.00041787775f * 64.f));
    q = __builtin_elementwise_fma(q, s, h4(0.00673485407f * 16.f));
    q = __builtin_elementwise_fma(q, s, h4(-0.05988154784f * 4.f));
    q = __builtin_elementwise_fma(q, s, h4(0.39435085475f));
    f16x4 p = __builtin_elementwise_fma(x, q, h4(0.5f));
    return __builtin_elementwise_min(__builtin_elementwise_max(p, h4(0.f)), h4(1.f));
}
// The two edge kernels work in a SCALED activation domain y = a x, a = 1 / sqrt(9.5) (kGA): then min(x^2, 9.5) / 9.5 = clamp01(y^2) and the
// clamp is a free output modifier of the multiply - one instruction less per pair of activations (8 -> 7; both kernels are bound by the
// issue of exactly these instructions).  Phi(x) = clamp01(1/2 + y Q(clamp01(y^2))), Q's coefficients = those of q times 9.5^i / a.  The factor
// is folded into what feeds the first Linears (P / Q tables, W0 and b0 of the embedding, the biases of the second Linears) and the edge
// tensor itself is stored scaled (e_hat = a e: e_hat' = e_hat + y Phi(y) needs no rescaling); the message mean, the taps and the stage API
// divide it out.  phi*s take y.
static constexpr float kGA = 0.324442842f, kGAi = 3.082207001f;
#define RN_QS0 1.19803158f
#define RN_QS1 -1.57943700f
#define RN_QS2 1.33882663f
#define RN_QS3 -0.45929830f
__device__ __forceinline__ f16x4 clamp01h(f16x4 v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, h4(0.f)), h4(1.f)); }
__device__ __forceinline__ f16x2 clamp01h(f16x2 v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, h2(0.f)), h2(1.f)); }
__device__ __forceinline__ f16x4 phi4s(f16x4 y) {
    const f16x4 s = clamp01h(y * y);
    f16x4 q = __builtin_elementwise_fma(s, h4(RN_QS3), h4(RN_QS2));
    q = __builtin_elementwise_fma(q, s, h4(RN_QS1));
    q = __builtin_elementwise_fma(q, s, h4(RN_QS0));
    return clamp01h(__builtin_elementwise_fma(y, q, h4(0.5f)));
}
// The same in the scaled domain of the node FFN chains with FIVE coefficients (the accuracy of phi4_hi at the instruction count of phi4s + 1):
// y = a x, a = 1 / sqrt(11.5) (kGAn), Phi(x) = clamp01(1/2 + y Qn(clamp01(y^2))), Qn_i = c_i 11.5^i / a with phi4_hi's c_i.  The chain's weight
// image carries the factors (first Linear x a, last Linear / a, biases of all but the last x a: api.cpp finalize_chain, k_ffn_chain).
static constexpr float kGAn = 0.294883912f, kGAni = 3.391164992f;
#define RN_QN0 1.33730881f
#define RN_QN1 -2.33529568f
#define RN_QN2 3.02046204f
#define RN_QN3 -2.15521932f
#define RN_QN4 0.63268071f
__device__ __forceinline__ f16x4 phi5n(f16x4 y) {
    const f16x4 s = clamp01h(y * y);
    f16x4 q = __builtin_elementwise_fma(s, h4(RN_QN4), h4(RN_QN3));
    q = __builtin_elementwise_fma(q, s, h4(RN_QN2));
    q = __builtin_elementwise_fma(q, s, h4(RN_QN1));
    q = __builtin_elementwise_fma(q, s, h4(RN_QN0));
    return clamp01h(__builtin_elementwise_fma(y, q, h4(0.5f)));
}
__device__ __forceinline__ f16x2 phi2s(f16x2 y) {
    const f16x2 s = clamp01h(y * y);
    f16x2 q = __builtin_elementwise_fma(s, h2(RN_QS3), h2(RN_QS2));
    q = __builtin_elementwise_fma(q, s, h2(RN_QS1));
    q = __builtin_elementwise_fma(q, s, h2(RN_QS0));
    return clamp01h(__builtin_elementwise_fma(y, q, h2(0.5f)));
}
__device__ __forceinline__ f16x2 lo2(f16x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f16x2 hi2(f16x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}
// Copy a 64 KiB fragment image (4096 x 16 B) into LDS with NT threads: all of a thread's loads are issued before its
// first LDS write (a plain copy loop compiles to load -> wait -> write per iteration, i.e. one L2 round trip each).
template <int NT>
__device__ __forceinline__ void stage_image(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int tid) {
    constexpr int PER = (4096 + NT - 1) / NT;
    constexpr int BATCH = PER < 8 ? PER : 8;
    constexpr bool EXACT = 4096 % NT == 0;
#pragma unroll
    for (int b0 = 0; b0 < PER; b0 += BATCH) {
        u32x4 t[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) if (b0 + i < PER) t[i] = src[EXACT ? tid + (b0 + i) * NT : min(tid + (b0 + i) * NT, 4095)];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) if (b0 + i < PER && (EXACT || tid + (b0 + i) * NT < 4096)) dst[tid + (b0 + i) * NT] = t[i];
    }
}
// The same copy by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, every piece of a thread in flight at once; a wave's
// piece is 1 KiB: wave-uniform LDS base, the DMA adds lane * 16).  The caller waits with s_waitcnt vmcnt(0) + a workgroup barrier
// before the first read (the compiler does not know these loads write LDS).  TOTAL: 16-byte units, whole waves per piece.
#define RN_DMA_STAGE 1
template <int NT, int TOTAL = 4096>
__device__ __forceinline__ void stage_image_dma(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int tid) {
    static_assert(TOTAL % 64 == 0 && NT % 64 == 0, "whole waves per DMA piece");
    constexpr int PER = (TOTAL + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (TOTAL % NT == 0 || tid + i * NT < TOTAL)       // (wave-uniform: NT and TOTAL are multiples of 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tid + i * NT),
                                             (__attribute__((address_space(3))) void*)(dst + (tid & ~63) + i * NT), 16, 0, 0);
}
__device__ __forceinline__ void dma_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// x as a (hi, lo) pair of bf16 in one word (hi in the low half): x = hi + lo to ~16 bits.  Operand of the k = 2 MFMAs that
// add a per-row constant (bias, P row) to an accumulator tile.
__device__ __forceinline__ unsigned split_word(float x) {
    const unsigned hi = pack2(x, 0.f) & 0xffffu;
    return hi | (pack2(x - __uint_as_float(hi << 16), 0.f) << 16);
}
// a * f16(lo / hi half of hp) + c in one mixed-precision FMA (f32 result)
__device__ __forceinline__ float fma_mix_lo(float a, f16x2 hp, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(hp), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(float a, f16x2 hp, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(hp), "v"(c));
    return d;
}
__device__ __forceinline__ f32x16 mfma32h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// Storage format of the per-edge tensor e of this path (fragment-major, see efrag_ptr): f16 of a e.  |a e| stays far inside the f16 range,
// the 11 mantissa bits beat bf16's 8, and - the reason - the residual update e <- e + GELU(.) of the fused kernel becomes ONE packed fma on
// the fragment words as loaded (bf16 storage cost an unpack, an f32 mixed fma and a repack per element: 10 vector instructions per four
// elements against 2, 128 of the ~1,460 a block issues), and the products e . Wc run as f16 MFMAs on the words as loaded.
// RN_E_F16=0 builds the bf16-storage form (A/B).
__device__ __forceinline__ bf16_t e_enc(float x) { return RN_E_F16 ? __builtin_bit_cast(bf16_t, (_Float16)x) : f2bf(x); }
__device__ __forceinline__ float e_dec(bf16_t v) { return RN_E_F16 ? (float)__builtin_bit_cast(_Float16, v) : bf2f(v); }
__device__ __forceinline__ f32x16 mfma_e(u32x4 a, u32x4 b, f32x16 c) { return RN_E_F16 ? mfma32h(a, b, c) : mfma32(a, b, c); }
// Storage format of the per-residue P tables (k_node_update -> fused kernel): f16 of a P, one halfword per entry in the SAME entry order as the
// (hi, lo) bf16 words of rounds 1-2 (entry 32 mb + m <-> accumulator row m of channel block mb).  The injection MFMA then is an f16 product of
// (P, 0) against (1, 1); 11 significand bits against the 8 of the gathered Q rows beside it.  Halves the P bytes k_node_update writes (it is
// HBM-bound on its table writes) and the fused kernel reads.  RN_P_F16=0 builds the word form (A/B).
__device__ __forceinline__ unsigned p_pack2(float a, float b) {       // two P entries -> one word of two f16
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// channel held by accumulator row m of a 32-row block when the output order is "natural per lane":
// lane half h = (m>>2)&1, register i = (m&3) + 4*(m>>3)  ->  channel 32*blk + 16*h + i
__host__ __device__ __forceinline__ int ch_nat(int blk, int m) { return 32 * blk + 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }
// ... when the output order must equal the e B-fragment layout (lane half h holds channels
// 32*blk + 8h + {0..7} and 32*blk + 16 + 8h + {0..7})
__host__ __device__ __forceinline__ int ch_efrag(int blk, int m) {
    int h = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
    return i < 8 ? 32 * blk + 8 * h + i : 32 * blk + 16 + 8 * h + (i - 8);
}

// Edge-embedding MLP image.  The 90 raw features are computed per lane half in a kernel-private
// order (EMB_SLOTS slots per half, 7 k-steps of 16): slot p of half h sits at k = 16*(p>>3) + 8h + (p&7).
#define EMB_KS 7
#define EMB_SLOTS 56
// Lane half h owns the NEIGHBOUR items 4h..4h+3 (atoms), 3h..3h+2 (unit bonds), 2h, 2h+1 (unit normals) - what its 28
// floats of the half-split geometry record hold (kernels_f32.hip: geomh_record) - against ALL central items:
//   p <  28: distance  central atom a = p >> 2   to neighbour atom   4h + (p & 3)        -> feature a*7 + b        (feature.py:414-418)
//   p <  43: cosine    central bond a = (p-28)/3 to neighbour bond   3h + (p-28) % 3     -> feature 49 + a*5 + b   (feature.py:451-464)
//   p <  51: cosine    central normal a = (p-43)>>1 to neighbour normal 2h + ((p-43)&1)  -> feature 74 + a*4 + b   (feature.py:493-512)
// slots of items a half does not have (atom 7, bond 5) and p >= 51 carry zero weights.
__host__ __device__ __forceinline__ int emb_feature_of_slot(int h, int p) {      // -> original feature id or -1
    if (p < 28) { int a = p >> 2, b = 4 * h + (p & 3); return b <= 6 ? a * 7 + b : -1; }
    if (p < 43) { int q = p - 28, a = q / 3, b = 3 * h + q % 3; return b <= 4 ? 49 + a * 5 + b : -1; }
    if (p < 51) { int q = p - 43, a = q >> 1, b = 2 * h + (q & 1); return 74 + a * 4 + b; }
    return -1;
}
