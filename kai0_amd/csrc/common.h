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

// tanh-approximated GELU exactly as torch.nn.functional.gelu(approximate="tanh") computes it in f32:
//   0.5 * x * (1 + tanh(sqrt(2/pi) * (x + 0.044715 x^3)))
// tanh via one v_exp_f32 and one reciprocal: 1 - 2/(exp(2x)+1).  |error| <= ~2e-7 absolute, which is far below the
// bf16 rounding every user applies next; libm's tanhf costs ~10x the instructions and dominated the GELU epilogue.
__device__ __forceinline__ float fast_tanhf(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - __fdividef(2.0f, e + 1.0f);
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float kBeta = 0.7978845608028654f;  // sqrt(2/pi)
    const float kKappa = 0.044715f;
    float inner = kBeta * (x + kKappa * x * x * x);
    return 0.5f * x * (1.0f + fast_tanhf(inner));
}
// d/dx gelu_tanh(x)
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
    const float kBeta = 0.7978845608028654f;
    const float kKappa = 0.044715f;
    float x2 = x * x;
    float inner = kBeta * (x + kKappa * x * x2);
    float t = fast_tanhf(inner);
    float left = 0.5f * (1.0f + t);
    float right = 0.5f * x * (1.0f - t * t) * kBeta * (1.0f + 3.0f * kKappa * x2);
    return left + right;
}
// Two values at a time on the packed-f32 VALU (v_pk_mul_f32 / v_pk_add_f32: one instruction per two lanes' worth of
// elements); only the exponential and the reciprocal stay scalar (quarter rate).  The fused GeGLU epilogues are VALU-bound
// (a 256x256 tile is 128 elements per thread with nothing else running on the CU), so instruction count is their time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 rbf2(f32x2 x) { return f32x2{rbf(x[0]), rbf(x[1])}; }
__device__ __forceinline__ f32x2 fast_tanh2(f32x2 x) {
    const f32x2 x2 = x * 2.0f;
    const f32x2 d = f32x2{__expf(x2[0]), __expf(x2[1])} + 1.0f;
    return 1.0f - f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])} * 2.0f;
}
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    const f32x2 inner = (x + x * x * x * 0.044715f) * 0.7978845608028654f;
    return x * 0.5f * (fast_tanh2(inner) + 1.0f);
}
// value and derivative from one tanh
__device__ __forceinline__ void gelu_tanh_both2(f32x2 x, f32x2& val, f32x2& grad) {
    const f32x2 xx = x * x;
    const f32x2 inner = (x + x * xx * 0.044715f) * 0.7978845608028654f;
    const f32x2 t = fast_tanh2(inner);
    const f32x2 half1pt = (t + 1.0f) * 0.5f;
    val = x * half1pt;
    grad = half1pt + x * (1.0f - t * t) * (xx * (3.0f * 0.044715f) + 1.0f) * (0.5f * 0.7978845608028654f);
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
