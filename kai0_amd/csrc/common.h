// kai0_amd/csrc/common.h — shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
// Everything here is written for wave64 / MFMA / 160 KiB LDS; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KAI0_API extern "C" __attribute__((visibility("default")))

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define GLB_PTR(T) __attribute__((address_space(1))) T*

// ---- error plumbing (host side) -------------------------------------------------------------
void kai0_set_error(const char* fmt, ...);
int kai0_check_launch(const char* what);

#define KAI0_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            kai0_set_error(__VA_ARGS__);   \
            return -1;                     \
        }                                  \
    } while (0)

// ---- bf16 <-> f32 ----------------------------------------------------------------------------
// f32 -> bf16 is round-to-nearest-even (what torch does); the compiler lowers the cast to
// v_cvt_pk_bf16_f32 on gfx950.
__device__ __forceinline__ float bf2f(bf16_t x) { return static_cast<float>(x); }
__device__ __forceinline__ bf16_t f2bf(float x) { return static_cast<bf16_t>(x); }
// round an f32 value to bf16 precision and come back (emulates a bf16-typed torch op result)
__device__ __forceinline__ float rbf(float x) { return static_cast<float>(static_cast<bf16_t>(x)); }

// tanh-approximated GELU, torch.nn.functional.gelu(approximate="tanh") in f32:
//   gelu(x) = 0.5 x (1 + tanh(u)),  u = sqrt(2/pi) (x + 0.044715 x^3)
// computed as x * sigma(2u) with sigma(2u) = 1 - 1 / (exp(2u) + 1)  [0.5 (1 + tanh u) = sigma(2u)]: one v_exp_f32 and one reciprocal per
// element (libm's tanhf costs ~10x the instructions), everything else on the packed-f32 VALU (v_pk_mul / v_pk_add / v_pk_fma: one
// instruction per two elements).  |error| <= ~2e-7 absolute, far below the bf16 rounding every user applies next.
// The fused GeGLU / GELU epilogues of the GEMMs are VALU-bound — 46 % of the GeGLU-backward GEMM's time with nothing overlapping it
// (profiles/r05_gemm_persistent_phases.txt) — so the forms below are written for instruction count: the exponent's argument is ONE
// polynomial in x^2 times x (constants pre-multiplied by 2 log2 e), sigma = 1 - r and sigma (1 - sigma) = sigma r reuse the reciprocal r,
// and the derivative  sigma + x sigma (1 - sigma) 2 u'  is two fused multiply-adds.  (1 - r, not exp * r: exp may be +inf.)
// Every GELU in this library — packed or scalar, GEMM epilogue, elementwise kernel, in-block GEMM — goes through gelu_sigma2, so they
// agree bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two values to bf16 precision and back: ONE v_cvt_pk_bf16_f32, a shift and a mask (the same round-to-nearest-even as rbf, per element)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 rbf2(f32x2 x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t));
    return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// sigma(2u) and r = 1 - sigma(2u) for u = beta (x + kappa x^3); xx = x * x
__device__ __forceinline__ void gelu_sigma2(f32x2 x, f32x2 xx, f32x2& sig, f32x2& r) {
    constexpr float kBeta = 0.7978845608028654f, kKappa = 0.044715f, kLog2e = 1.4426950408889634f;
    constexpr float a0 = 2.0f * kBeta * kLog2e, a1 = 2.0f * kBeta * kKappa * kLog2e;
    const f32x2 arg = x * pk_fma(xx, f32x2{a1, a1}, f32x2{a0, a0});  // 2 u log2(e)
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} + 1.0f;
    r = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    sig = 1.0f - r;
}
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    f32x2 sig, r;
    gelu_sigma2(x, x * x, sig, r);
    return x * sig;
}
// value and derivative from one exponential:  gelu' = sigma + x sigma (1 - sigma) 2 beta (1 + 3 kappa x^2)
__device__ __forceinline__ void gelu_tanh_both2(f32x2 x, f32x2& val, f32x2& grad) {
    constexpr float kBeta = 0.7978845608028654f, kKappa = 0.044715f;
    constexpr float b0 = 2.0f * kBeta, b1 = 6.0f * kBeta * kKappa;
    const f32x2 xx = x * x;
    f32x2 sig, r;
    gelu_sigma2(x, xx, sig, r);
    val = x * sig;
    grad = pk_fma(x * (sig * r), pk_fma(xx, f32x2{b1, b1}, f32x2{b0, b0}), sig);
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return gelu_tanh2(f32x2{x, x})[0]; }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
    f32x2 v, g;
    gelu_tanh_both2(f32x2{x, x}, v, g);
    return g[0];
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ---- wave64 reductions -----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// block-wide sum for blocks of NW waves (NW*64 threads); `red` is an LDS scratch of >= NW floats
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += red[i];
    return r;
}
