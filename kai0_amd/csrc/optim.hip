// optim.hip — gradient-norm clip + fused AdamW over flat shards (train_pytorch.py:469-475,557-561;
// optimizer.py:15-85).  HBM-bound: 16 B/param of f32 state (master, m, v read+write) + grad + bf16 copy.
#include "common.h"
#include "../../include/kai0hip.h"

namespace {

template <bool GF32>
__global__ __launch_bounds__(256) void sumsq_kernel(const void* __restrict__ g, int64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    if constexpr (GF32) {
        const float* p = reinterpret_cast<const float*>(g);
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p + i * 4);
            acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        if (blockIdx.x == 0)
            for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += p[i] * p[i];
    } else {
        const bf16_t* p = reinterpret_cast<const bf16_t*>(g);
        const int64_t n8 = n >> 3;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(p + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf2f(v[e]);
                acc += f * f;
            }
        }
        if (blockIdx.x == 0)
            for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += 256) {
                const float f = bf2f(p[i]);
                acc += f * f;
            }
    }
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;  // per-block partial; sumsq_finish_kernel adds them in a fixed order
}

// out[0] += sum of `n` block partials (n <= 4096), always in the same order: the global gradient norm, and with it the whole
// training trajectory, is reproducible bit for bit (f32 atomics across blocks were not)
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) out[0] += acc;
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
    const float nrm = sqrtf(sumsq[0]);
    if (norm_out) norm_out[0] = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    coef[0] = c < 1.0f ? c : 1.0f;
}

// dst[i] = round(sum_j src[j * stride + i]), j = 0 .. chunks-1 in that order, summed in f32: the local half of the all-pairs
// reduce-scatter (sharded.py rs_algo "alltoall": every peer's copy of this rank's gradient slice arrives over its own xGMI
// link, the sum happens here, once, instead of hop by hop around a ring with a bf16 rounding per hop)
template <bool F32>
__global__ __launch_bounds__(256) void sum_chunks_kernel(const void* __restrict__ src, int chunks, int64_t stride, int64_t n,
                                                         void* __restrict__ dst) {
    constexpr int V = F32 ? 4 : 8;
    const int64_t nv = n / V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int j = 0; j < chunks; ++j) {
            if constexpr (F32) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + j * stride + i * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            } else {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(src) + j * stride + i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
            }
        }
        if constexpr (F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dst) + i * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        } else {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(dst) + i * 8) = o;
        }
    }
}

template <bool GF32, bool PF32>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const void* __restrict__ grad, void* __restrict__ param, int64_t n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2, const float* __restrict__ coef) {
    const float cc = coef ? coef[0] : 1.0f;
    const float inv_bc1 = 1.0f / bc1;
    const float inv_sqrt_bc2 = rsqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float g = GF32 ? reinterpret_cast<const float*>(grad)[i] : bf2f(reinterpret_cast<const bf16_t*>(grad)[i]);
        g *= cc;
        float p = master[i];
        float mi = m[i] * b1 + (1.0f - b1) * g;
        float vi = v[i] * b2 + (1.0f - b2) * g * g;
        // torch.optim.AdamW: p *= 1 - lr*wd ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
        p = p * (1.0f - lr * wd);
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p = p - (lr * inv_bc1) * (mi / denom);
        master[i] = p;
        m[i] = mi;
        v[i] = vi;
        if (PF32) reinterpret_cast<float*>(param)[i] = p;
        else reinterpret_cast<bf16_t*>(param)[i] = f2bf(p);
    }
}

// The same update for a [rows][row_len] parameter whose gradient is zero in most rows (the 257152 x 2048 embedding table: a step
// touches at most B x 200 of its rows).  A row whose moments are exactly zero and whose gradient is zero is a FIXED POINT of the
// update above when 1 - lr*wd rounds to 1 (the host checks lr*wd < 2^-25): m and v stay 0, the step is 0/eps = 0, p * 1 = p.  Such
// rows are skipped after reading only their gradient; `active[row]` (persistent, uint8) records that a row has ever seen a nonzero
// gradient, i.e. that its moments may be nonzero.  Every element that IS updated goes through exactly adamw_kernel's arithmetic,
// so the result is bit-identical to the dense pass — at 2 B instead of 28 B per element of an idle row.
template <bool GF32, bool PF32>
__global__ __launch_bounds__(256) void adamw_rows_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                         const void* __restrict__ grad, void* __restrict__ param, int64_t n_rows,
                                                         int row_len, unsigned char* __restrict__ active, float lr, float b1, float b2,
                                                         float eps, float wd, float bc1, float bc2, const float* __restrict__ coef) {
    const float cc = coef ? coef[0] : 1.0f;
    const float inv_bc1 = 1.0f / bc1;
    const float inv_sqrt_bc2 = rsqrtf(bc2);
    for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const int64_t base = row * row_len;
        int nz = 0;
        if (!GF32 && (row_len & 7) == 0 && ((uintptr_t)grad & 15) == 0) {
            // the scan is what an idle row costs: 16 B per lane, "nonzero" = any bit besides the signs (-0.0 is a zero gradient)
            for (int c = threadIdx.x * 8; c < row_len; c += 256 * 8) {
                const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(grad) + base + c);
                nz |= ((u.x | u.y | u.z | u.w) & 0x7fff7fffu) != 0u;
            }
        } else {
            for (int c = threadIdx.x; c < row_len; c += 256) {
                const float g = GF32 ? reinterpret_cast<const float*>(grad)[base + c] : bf2f(reinterpret_cast<const bf16_t*>(grad)[base + c]);
                nz |= (g != 0.0f);
            }
        }
        const int any = __syncthreads_or(nz);
        if (!any && !active[row]) continue;  // (block-uniform)
        if (any && threadIdx.x == 0) active[row] = 1;
        for (int c = threadIdx.x; c < row_len; c += 256) {
            const int64_t i = base + c;
            float g = GF32 ? reinterpret_cast<const float*>(grad)[i] : bf2f(reinterpret_cast<const bf16_t*>(grad)[i]);
            g *= cc;
            float p = master[i];
            float mi = m[i] * b1 + (1.0f - b1) * g;
            float vi = v[i] * b2 + (1.0f - b2) * g * g;
            p = p * (1.0f - lr * wd);
            const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
            p = p - (lr * inv_bc1) * (mi / denom);
            master[i] = p;
            m[i] = mi;
            v[i] = vi;
            if (PF32) reinterpret_cast<float*>(param)[i] = p;
            else reinterpret_cast<bf16_t*>(param)[i] = f2bf(p);
        }
    }
}

inline int opt_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

KAI0_API int kai0_sumsq(const void* g, int g_f32, int64_t n, float* out, float* scratch, kai0_stream_t stream) {
    if (n <= 0) return 0;
    KAI0_REQUIRE(scratch != nullptr, "kai0_sumsq: needs a scratch buffer of 4096 floats");
    KAI0_REQUIRE(((uintptr_t)g % 16) == 0, "kai0_sumsq: unaligned buffer");
    const int grid = opt_grid(g_f32 ? n / 4 + 1 : n / 8 + 1);
    if (g_f32) hipLaunchKernelGGL((sumsq_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, scratch);
    else hipLaunchKernelGGL((sumsq_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, scratch);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, grid, out);
    return kai0_check_launch("kai0_sumsq");
}

KAI0_API int kai0_sum_chunks(const void* src, int is_f32, int chunks, int64_t chunk_stride, int64_t n, void* dst,
                            kai0_stream_t stream) {
    if (n <= 0) return 0;
    KAI0_REQUIRE(src && dst && chunks >= 1, "kai0_sum_chunks: null buffer or no chunks");
    const int V = is_f32 ? 4 : 8;
    KAI0_REQUIRE((n % V) == 0 && (chunk_stride % V) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                 "kai0_sum_chunks: n=%lld, stride=%lld must be multiples of %d elements and the buffers 16-byte aligned",
                 (long long)n, (long long)chunk_stride, V);
    const int grid = opt_grid(n / V);
    if (is_f32) hipLaunchKernelGGL((sum_chunks_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src, chunks, chunk_stride, n, dst);
    else hipLaunchKernelGGL((sum_chunks_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src, chunks, chunk_stride, n, dst);
    return kai0_check_launch("kai0_sum_chunks");
}

KAI0_API int kai0_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, kai0_stream_t stream) {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef, norm_out);
    return kai0_check_launch("kai0_clip_coef");
}

KAI0_API int kai0_adamw(float* master, float* m, float* v, const void* grad, int grad_f32, void* model_param,
                        int param_f32, int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bias_c1,
                        float bias_c2, const float* clip_coef, kai0_stream_t stream) {
    if (n <= 0) return 0;
    KAI0_REQUIRE(master && m && v && grad && model_param, "kai0_adamw: null buffer");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(opt_grid(n)), block(256);
#define LAUNCH(G, P)                                                                                              \
    hipLaunchKernelGGL((adamw_kernel<G, P>), grid, block, 0, s, master, m, v, grad, model_param, n, lr, beta1, beta2, \
                       eps, wd, bias_c1, bias_c2, clip_coef)
    if (grad_f32 && param_f32) LAUNCH(true, true);
    else if (grad_f32) LAUNCH(true, false);
    else if (param_f32) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    return kai0_check_launch("kai0_adamw");
}

KAI0_API int kai0_adamw_rows(float* master, float* m, float* v, const void* grad, int grad_f32, void* model_param, int param_f32,
                             int64_t n_rows, int row_len, unsigned char* row_active, float lr, float beta1, float beta2, float eps,
                             float wd, float bias_c1, float bias_c2, const float* clip_coef, kai0_stream_t stream) {
    if (n_rows <= 0) return 0;
    KAI0_REQUIRE(master && m && v && grad && model_param && row_active && row_len > 0, "kai0_adamw_rows: null buffer");
    KAI0_REQUIRE(1.0f - lr * wd == 1.0f, "kai0_adamw_rows: lr * wd = %g does not round away (1 - lr*wd must be 1.0f): idle rows are not fixed points, use kai0_adamw",
                 (double)lr * (double)wd);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(n_rows < 16384 ? n_rows : 16384)), block(256);
#define LAUNCH(G, P)                                                                                                          \
    hipLaunchKernelGGL((adamw_rows_kernel<G, P>), grid, block, 0, s, master, m, v, grad, model_param, n_rows, row_len, row_active, lr, \
                       beta1, beta2, eps, wd, bias_c1, bias_c2, clip_coef)
    if (grad_f32 && param_f32) LAUNCH(true, true);
    else if (grad_f32) LAUNCH(true, false);
    else if (param_f32) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    return kai0_check_launch("kai0_adamw_rows");
}
