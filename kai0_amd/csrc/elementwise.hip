// elementwise.hip — the HBM-bound glue of the pi0.5 path: RoPE, masked softmax, GeGLU, gated residual,
// embedding gather, casts, patch im2col, flow-matching mix / MSE / Euler.  Every kernel moves 16 B per lane
// per access where the layout allows and does its arithmetic in f32, rounding to bf16 exactly where the
// reference's bf16-typed torch ops do.
#include "common.h"
#include "../../include/kai0hip.h"
#include <limits.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;

namespace {

__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x8*>(p) = t;
}

inline int ew_grid(int64_t n_items, int per_block) {
    int64_t b = (n_items + per_block - 1) / per_block;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------------------------------------- RoPE
// dst[b][s][h][:] = rope(src[b][s][h][:], pos[b][s]) for B x S rows of H heads (in place when dst == src and the strides
// agree).  A thread owns 4 consecutive frequency indices i4..i4+3 of one (b, s) row and walks the heads, so the sin / cos of
// a position is computed once for all heads.  Strides (elements): *_bs between batch entries, *_ld between rows; pos_bs
// between the position rows of consecutive batch entries.  The strided form lets the joint-attention assembly (scatter a
// segment's q / k into the joint buffer, or gather its gradient back out) apply the rotation on the way instead of in a
// separate in-place pass over the joint buffer.
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ dst, const bf16_t* __restrict__ src,
                                                   const int32_t* __restrict__ pos, const float* __restrict__ inv_freq, int B,
                                                   int S, int64_t dst_bs, int64_t dst_ld, int64_t src_bs, int64_t src_ld,
                                                   int64_t pos_bs, int H, int HD, int inverse, bf16_t* __restrict__ x2 = nullptr,
                                                   int64_t x2_bs = 0, int64_t x2_ld = 0, int H2 = 0) {
    // x2 (optional): a second tensor rotated in place with the SAME positions (k next to q: one launch, the trigonometry once)
    const int tpr = HD >> 3;              // threads per row (HD/2 freqs, 4 per thread)
    const int rpb = 256 / tpr;            // rows per block
    const int rl = threadIdx.x / tpr;
    const int i4 = (threadIdx.x - rl * tpr) * 4;
    if (rl >= rpb) return;
    const int64_t nrows = (int64_t)B * S;
    for (int64_t r = (int64_t)blockIdx.x * rpb + rl; r < nrows; r += (int64_t)gridDim.x * rpb) {
        const int b = (int)(r / S), s = (int)(r - (int64_t)b * S);
        const float p = (float)pos[(int64_t)b * pos_bs + s];
        float c[4], sn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ang = inv_freq[i4 + e] * p;
            c[e] = rbf(cosf(ang));
            sn[e] = rbf(sinf(ang));
            if (inverse) sn[e] = -sn[e];
        }
        const bf16_t* xr = src + (int64_t)b * src_bs + (int64_t)s * src_ld;
        bf16_t* yr = dst + (int64_t)b * dst_bs + (int64_t)s * dst_ld;
        for (int h = 0; h < H; ++h) {
            const bf16x4 a = *reinterpret_cast<const bf16x4*>(xr + h * HD + i4);
            const bf16x4 bq = *reinterpret_cast<const bf16x4*>(xr + h * HD + i4 + (HD >> 1));
            bf16x4 o1, o2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x1 = bf2f(a[e]), x2 = bf2f(bq[e]);
                // q*cos + rotate_half(q)*sin with every product and the sum rounded to bf16
                o1[e] = f2bf(rbf(x1 * c[e]) + rbf(-x2 * sn[e]));
                o2[e] = f2bf(rbf(x2 * c[e]) + rbf(x1 * sn[e]));
            }
            *reinterpret_cast<bf16x4*>(yr + h * HD + i4) = o1;
            *reinterpret_cast<bf16x4*>(yr + h * HD + i4 + (HD >> 1)) = o2;
        }
        if (x2 != nullptr) {
            bf16_t* zr = x2 + (int64_t)b * x2_bs + (int64_t)s * x2_ld;
            for (int h = 0; h < H2; ++h) {
                const bf16x4 a = *reinterpret_cast<const bf16x4*>(zr + h * HD + i4);
                const bf16x4 bq = *reinterpret_cast<const bf16x4*>(zr + h * HD + i4 + (HD >> 1));
                bf16x4 o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x1 = bf2f(a[e]), x2v = bf2f(bq[e]);
                    o1[e] = f2bf(rbf(x1 * c[e]) + rbf(-x2v * sn[e]));
                    o2[e] = f2bf(rbf(x2v * c[e]) + rbf(x1 * sn[e]));
                }
                *reinterpret_cast<bf16x4*>(zr + h * HD + i4) = o1;
                *reinterpret_cast<bf16x4*>(zr + h * HD + i4 + (HD >> 1)) = o2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------- masked softmax
constexpr int SM_MAXC = 8;  // ld <= 4096
// NCH = 16-B chunks per lane (ld <= NCH * 512): instantiated for 1, 2, 4, 8 so that a 1024-wide row keeps 16 values in
// registers, not 64 (the single 8-chunk version needed 255 VGPRs -> 2 waves/SIMD and ran at 1.8 TB/s)
template <int NCH>
__global__ __launch_bounds__(256) void softmax_mask_fwd_kernel(const bf16_t* scores, bf16_t* probs,
                                                               const int32_t* __restrict__ qcode,
                                                               const int32_t* __restrict__ kcode, int B, int Sq, int H,
                                                               int Sk, int64_t ld, int64_t bstride, int q0,
                                                               int64_t qld, int64_t kld) {
    const int lane = threadIdx.x & 63;
    const int64_t rows_pb = (int64_t)Sq * H;
    const int64_t rows = rows_pb * B;
    const int nchunk = (int)(ld >> 3);
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row / rows_pb);
        const int64_t rr = row - (int64_t)b * rows_pb;
        const int s = q0 + (int)(rr / H);
        const int qc = qcode ? qcode[(int64_t)b * qld + s] : INT_MAX;
        const bf16_t* sp = scores + (int64_t)b * bstride + rr * ld;
        bf16_t* pp = probs + (int64_t)b * bstride + rr * ld;
        float v[NCH][8];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                ld8(sp + ci * 8, v[c]);
                int kc8[8];
                if (qcode) {  // key codes of this chunk: two 16-B loads when whole and aligned, else clamped scalars
                    const int32_t* kp = kcode + (int64_t)b * kld + ci * 8;
                    if (ci * 8 + 8 <= Sk && (((uintptr_t)kp) & 15) == 0) {
                        const i32x4 k0 = *reinterpret_cast<const i32x4*>(kp);
                        const i32x4 k1 = *reinterpret_cast<const i32x4*>(kp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { kc8[e] = k0[e]; kc8[4 + e] = k1[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) kc8[e] = kcode[(int64_t)b * kld + min(ci * 8 + e, Sk - 1)];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = ci * 8 + e;
                    bool ok = j < Sk;
                    if (qcode) ok = ok && kc8[e] <= qc;
                    v[c][e] = ok ? v[c][e] : -INFINITY;
                    m = fmaxf(m, v[c][e]);
                }
            }
        }
        m = wave_max(m);
        float sum = 0.f;
        if (m > -INFINITY) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ci = c * 64 + lane;
                if (ci < nchunk) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[c][e] = __expf(v[c][e] - m);  // exp(-inf) = 0 for masked columns; v_exp_f32 path (as attn_fwd):
                                                        // libm expf made this kernel VALU-bound at 1.8 TB/s
                        sum += v[c][e];
                    }
                }
            }
        }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (m > -INFINITY) ? v[c][e] * inv : 0.f;
                st8(pp + ci * 8, o);
            }
        }
    }
}

// out[r] = sum_d a[r][d] * b[r][d]: 8 bf16 per lane, D/8 lanes per row, rows packed into the wave
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                     float* __restrict__ out, int64_t rows, int D) {
    const int lpr = D >> 3;  // lanes per row (<= 64 enforced by the host)
    int span = 1;
    while (span < lpr) span <<= 1;  // lanes reserved per row (power of two, so xor-shuffles stay inside the row)
    const int rpw = 64 / span;      // rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane / span, li = lane - sub * span;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r0 = wave_id * rpw; r0 < rows; r0 += nwaves * rpw) {
        const int64_t r = r0 + sub;
        float acc = 0.f;
        if (r < rows && li < lpr) {
            float x[8], y[8];
            ld8(a + r * D + li * 8, x);
            ld8(b + r * D + li * 8, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += x[e] * y[e];
        }
        for (int off = span >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (r < rows && li == 0) out[r] = acc;
    }
}

template <bool DP_F32>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ probs, const void* dprobs,
                                                          bf16_t* dscores, int64_t rows, int Sk, int64_t ld,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const int nchunk = (int)(ld >> 3);
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        float p[SM_MAXC][8], dp[SM_MAXC][8];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                ld8(probs + row * ld + ci * 8, p[c]);
                if constexpr (DP_F32) {
                    const float* dpp = reinterpret_cast<const float*>(dprobs) + row * ld + ci * 8;
                    f32x4 a = *reinterpret_cast<const f32x4*>(dpp);
                    f32x4 b = *reinterpret_cast<const f32x4*>(dpp + 4);
                    dp[c][0] = a[0]; dp[c][1] = a[1]; dp[c][2] = a[2]; dp[c][3] = a[3];
                    dp[c][4] = b[0]; dp[c][5] = b[1]; dp[c][6] = b[2]; dp[c][7] = b[3];
                } else {
                    ld8(reinterpret_cast<const bf16_t*>(dprobs) + row * ld + ci * 8, dp[c]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (ci * 8 + e >= Sk) { p[c][e] = 0.f; dp[c][e] = 0.f; }
                    dot += p[c][e] * dp[c][e];
                }
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (p[c][e] * (dp[c][e] - dot)) * scale;
                st8(dscores + row * ld + ci * 8, o);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------- GeGLU
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* g, const bf16_t* __restrict__ u,
                                                        bf16_t* h, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float gv[8], uv[8], o[8];
        ld8(g + i * 8, gv);
        ld8(u + i * 8, uv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rbf(gelu_tanh_f(gv[e])) * uv[e];
        st8(h + i * 8, o);
    }
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ g,
                                                        const bf16_t* __restrict__ u, bf16_t* __restrict__ dg,
                                                        bf16_t* __restrict__ du, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float dv[8], gv[8], uv[8], og[8], ou[8];
        ld8(dh + i * 8, dv);
        ld8(g + i * 8, gv);
        ld8(u + i * 8, uv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = rbf(gelu_tanh_f(gv[e]));
            ou[e] = dv[e] * a;
            og[e] = rbf(dv[e] * uv[e]) * gelu_tanh_grad_f(gv[e]);
        }
        st8(dg + i * 8, og);
        st8(du + i * 8, ou);
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ pre,
                                                       bf16_t* __restrict__ dx, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float dv[8], pv[8], o[8];
        ld8(dy + i * 8, dv);
        ld8(pre + i * 8, pv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = dv[e] * gelu_tanh_grad_f(pv[e]);
        st8(dx + i * 8, o);
    }
}

__global__ __launch_bounds__(256) void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        y[i] = v / (1.0f + expf(-v));
    }
}
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        const float sg = 1.0f / (1.0f + expf(-v));
        dx[i] = dy[i] * (sg * (1.0f + v * (1.0f - sg)));
    }
}

// gated residual forward (modeling_gemma.py:209-227): out = bf16(x + bf16(y * gate[b]))
__global__ __launch_bounds__(256) void gated_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                        const bf16_t* __restrict__ gate, bf16_t* __restrict__ out,
                                                        int64_t rows, int rpb, int D) {
    const int d8 = D >> 3;
    const int64_t total = rows * d8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / d8;
        const int c0 = (int)(i - row * d8) * 8;
        float xv[8], yv[8], gt[8];
        ld8(x + row * D + c0, xv);
        ld8(y + row * D + c0, yv);
        ld8(gate + (row / rpb) * D + c0, gt);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] += rbf(yv[e] * gt[e]);
        st8(out + row * D + c0, xv);
    }
}

// gated residual backward: block per batch entry, thread owns 8 columns
__global__ __launch_bounds__(256) void gated_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                        const bf16_t* __restrict__ gate, bf16_t* __restrict__ dy,
                                                        bf16_t* __restrict__ dgate, int rpb, int D) {
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c0 = tid * 8; c0 < D; c0 += 256 * 8) {
        float gt[8], acc[8];
        ld8(gate + (int64_t)b * D + c0, gt);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int i = 0; i < rpb; ++i) {
            const int64_t row = (int64_t)b * rpb + i;
            float dv[8], yv[8], o[8];
            ld8(dout + row * D + c0, dv);
            ld8(y + row * D + c0, yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = dv[e] * gt[e];
                acc[e] += dv[e] * yv[e];
            }
            st8(dy + row * D + c0, o);
        }
        st8(dgate + (int64_t)b * D + c0, acc);
    }
}

// ------------------------------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void embed_gather_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ tok,
                                                           bf16_t* __restrict__ out, int T, int D, float scale,
                                                           int64_t out_bs, int64_t out_row0, int64_t out_ld) {
    const int r = blockIdx.x;  // b*T + t
    const int b = r / T, t = r - b * T;
    const bf16_t* src = table + tok[r] * (int64_t)D;
    bf16_t* dst = out + (int64_t)b * out_bs + (out_row0 + t) * out_ld;
    for (int c0 = threadIdx.x * 8; c0 < D; c0 += 256 * 8) {
        float v[8];
        ld8(src + c0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= scale;
        st8(dst + c0, v);
    }
}
// Deterministic dense embedding gradient: block i owns flattened token i; only the FIRST occurrence of a
// token id sums (in f32) the rows of every occurrence and writes the bf16 gradient row once.
__global__ __launch_bounds__(256) void embed_grad_kernel(const bf16_t* __restrict__ dout, const int64_t* __restrict__ tok,
                                                         bf16_t* __restrict__ dtable, int n, int T, int D, float scale,
                                                         int64_t bs, int64_t row0, int64_t ld) {
    __shared__ int earlier;
    const int i = blockIdx.x;
    const int64_t my = tok[i];
    if (threadIdx.x == 0) earlier = 0;
    __syncthreads();
    int found = 0;
    for (int j = threadIdx.x; j < i; j += 256) found |= (tok[j] == my);
    if (found) earlier = 1;  // benign race: all writers store 1
    __syncthreads();
    if (earlier) return;
    // the occurrences of this token among tokens i..n-1 as a bitmap (built by all threads at once), then walked in ascending order:
    // the f32 sum keeps its fixed order, and the walk costs n/32 word tests instead of n token compares per thread
    extern __shared__ unsigned int occ[];
    const int nw = (n - i + 31) >> 5;
    for (int w = threadIdx.x; w < nw; w += 256) occ[w] = 0u;
    __syncthreads();
    for (int j = i + threadIdx.x; j < n; j += 256)
        if (tok[j] == my) atomicOr(&occ[(j - i) >> 5], 1u << ((j - i) & 31));
    __syncthreads();
    for (int c0 = threadIdx.x * 8; c0 < D; c0 += 256 * 8) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int w = 0; w < nw; ++w) {
            unsigned int bits = occ[w];  // (block-uniform)
            while (bits) {
                const int j = i + (w << 5) + __builtin_ctz(bits);
                bits &= bits - 1;
                const int b = j / T, t = j - b * T;
                float v[8];
                ld8(dout + (int64_t)b * bs + (row0 + t) * ld + c0, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += rbf(v[e] * scale);
            }
        }
        st8(dtable + my * (int64_t)D + c0, acc);
    }
}

// ------------------------------------------------------------------------------------ casts / adds / copies
__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8);
        f32x4 b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        st8(y + i * 8, v);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += 256) y[i] = f2bf(x[i]);
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        ld8(x + i * 8, v);
        *reinterpret_cast<f32x4*>(y + i * 8) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(y + i * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += 256) y[i] = bf2f(x[i]);
}
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ o, int64_t n) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        ld8(a + i * 8, x);
        ld8(b + i * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        st8(o + i * 8, x);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += 256) o[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) o[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows,
                                                        int D, int64_t sbs, int64_t sr0, int64_t sld, int64_t dbs,
                                                        int64_t dr0, int64_t dld) {
    const int64_t r = blockIdx.x;  // b*rows + i
    const int b = (int)(r / rows), i = (int)(r - (int64_t)b * rows);
    const bf16_t* s = src + (int64_t)b * sbs + (sr0 + i) * sld;
    bf16_t* d = dst + (int64_t)b * dbs + (dr0 + i) * dld;
    for (int c0 = threadIdx.x * 8; c0 < D; c0 += 256 * 8)
        *reinterpret_cast<bf16x8*>(d + c0) = *reinterpret_cast<const bf16x8*>(s + c0);
}

// Several strided row moves in ONE launch (the joint-attention assembly of a layer: q / k of both segments rotated into the joint
// buffers, v copied, the padding rows cleared — 9 launches of 5-50 us otherwise; likewise the way back out).  Every part moves
// B x rows rows of `cols` elements; mode 1 / 2 applies RoPE / its inverse per head of HD elements with the row's position
// pos[b * pos_bs + pos_off + r], mode 3 writes zeros.  A block owns 256 / (HD/8) consecutive rows of one part: the thread layout
// of rope_kernel (4 frequencies per thread), which for the plain copies is simply 16-B pieces strided over the row.
struct PackArgs {
    kai0_pack_part part[12];
    int first[13];
    const int32_t* pos;
    const float* inv_freq;
    int64_t pos_bs;
    int n, HD;
};
__global__ __launch_bounds__(256) void pack_rows_kernel(PackArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;
    const kai0_pack_part pt = a.part[i];
    const int HD = a.HD;
    const int tpr = HD >> 3, rpb = 256 / tpr;
    const int rl = threadIdx.x / tpr, tl = threadIdx.x - rl * tpr;
    if (rl >= rpb) return;
    const int64_t r = (int64_t)((int)blockIdx.x - a.first[i]) * rpb + rl;
    if (r >= (int64_t)pt.B * pt.rows) return;
    const int b = (int)(r / pt.rows), s = (int)(r - (int64_t)b * pt.rows);
    bf16_t* yr = (bf16_t*)pt.dst + (int64_t)b * pt.dst_bs + (int64_t)s * pt.dst_ld;
    if (pt.mode == 3) {
        for (int c = tl * 8; c < pt.cols; c += tpr * 8) *reinterpret_cast<bf16x8*>(yr + c) = bf16x8{};
        return;
    }
    const bf16_t* xr = (const bf16_t*)pt.src + (int64_t)b * pt.src_bs + (int64_t)s * pt.src_ld;
    if (pt.mode == 0) {
        for (int c = tl * 8; c < pt.cols; c += tpr * 8) *reinterpret_cast<bf16x8*>(yr + c) = *reinterpret_cast<const bf16x8*>(xr + c);
        return;
    }
    const int i4 = tl * 4;
    const float p = (float)a.pos[(int64_t)b * a.pos_bs + pt.pos_off + s];
    float c[4], sn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ang = a.inv_freq[i4 + e] * p;
        c[e] = rbf(cosf(ang));
        sn[e] = rbf(sinf(ang));
        if (pt.mode == 2) sn[e] = -sn[e];
    }
    const int H = pt.cols / HD;
    for (int h = 0; h < H; ++h) {
        const bf16x4 x1v = *reinterpret_cast<const bf16x4*>(xr + h * HD + i4);
        const bf16x4 x2v = *reinterpret_cast<const bf16x4*>(xr + h * HD + i4 + (HD >> 1));
        bf16x4 o1, o2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x1 = bf2f(x1v[e]), x2 = bf2f(x2v[e]);
            o1[e] = f2bf(rbf(x1 * c[e]) + rbf(-x2 * sn[e]));  // the roundings of rope_kernel
            o2[e] = f2bf(rbf(x2 * c[e]) + rbf(x1 * sn[e]));
        }
        *reinterpret_cast<bf16x4*>(yr + h * HD + i4) = o1;
        *reinterpret_cast<bf16x4*>(yr + h * HD + i4 + (HD >> 1)) = o2;
    }
}

// Sampled content checksum of a list of buffers (device array of {pointer, bytes}): every `stride`-th 64-byte unit of each buffer,
// mixed with its position, summed into one 64-bit integer (integer addition: order-free, hence deterministic).  The inference engine
// stamps the tensors its derived weight copies were cut from with it at the end of every action chunk, so that an in-place edit
// of a weight that bypasses autograd's version counters (`p.data.mul_()`, a foreign kernel) is noticed (infer.py `_content_ok`).
__global__ __launch_bounds__(256) void sampled_checksum_kernel(const kai0_ck_item* __restrict__ items, int n, int stride,
                                                               unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (int it = blockIdx.y; it < n; it += gridDim.y) {
        const kai0_ck_item item = items[it];
        const int64_t units = item.nbytes >> 6;  // 64-byte units: four adjacent lanes read one (a first version read lone 16-B chunks 512 B
        const int64_t ns = (units + stride - 1) / stride;  // apart — a cache line per 16 bytes — and took 88 us per action chunk)
        const uint4* base = reinterpret_cast<const uint4*>(item.ptr);
        for (int64_t k = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2; k < ns; k += (int64_t)gridDim.x * 64) {
            int64_t u = k * stride + ((it * 7 + k) % stride);  // (a different phase per sample: no fixed offset is blind)
            if (u >= units) u = units - 1;
            const int64_t c = u * 4 + (threadIdx.x & 3);
            const uint4 w = base[c];
            const unsigned long long lo = ((unsigned long long)w.y << 32) | w.x, hi = ((unsigned long long)w.w << 32) | w.z;
            acc += (lo ^ (hi * 0x9E3779B97F4A7C15ull)) * (2ull * (unsigned long long)(c + it) + 1ull);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// bf16 matrix transpose through LDS: src [R][C] -> dst [C][R]; 64x64 tiles, 16-B global accesses both ways.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, int C,
                                                        int64_t src_ld, int64_t dst_ld, int64_t src_bs, int64_t dst_bs) {
    __shared__ bf16_t tile[64][72];  // +8 pad: column reads hit different banks
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    src += (int64_t)blockIdx.z * src_bs;
    dst += (int64_t)blockIdx.z * dst_bs;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = p * 32 + (t >> 3), c8 = (t & 7) * 8;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(0.f);
        if (r0 + r < R && c0 + c8 < C) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)(r0 + r) * src_ld + c0 + c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][c8 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = p * 32 + (t >> 3), r8 = (t & 7) * 8;
        if (c0 + c < C && r0 + r8 < R) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[r8 + e][c];
            *reinterpret_cast<bf16x8*>(dst + (int64_t)(c0 + c) * dst_ld + r0 + r8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ patch embed
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, float* __restrict__ cols, int n_img,
                                                     int C, int HW, int P) {
    const int G = HW / P;               // patches per side
    const int Kd = C * P * P;
    const int64_t total = (int64_t)n_img * G * G * Kd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % Kd);
        const int64_t pr = i / Kd;
        const int px = (int)(pr % G), py = (int)((pr / G) % G);
        const int n = (int)(pr / ((int64_t)G * G));
        const int c = k / (P * P), kk = k - c * P * P;
        const int ky = kk / P, kx = kk - ky * P;
        cols[i] = img[(((int64_t)n * C + c) * HW + (py * P + ky)) * HW + (px * P + kx)];
    }
}
__global__ __launch_bounds__(256) void add_pos_cast_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                           bf16_t* __restrict__ out, int64_t rows, int n_pos, int D) {
    const int64_t total = rows * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D);
        out[i] = f2bf(x[i] + pos[(r % n_pos) * D + c]);
    }
}

// ------------------------------------------------------------------------------------------ flow matching
// create_sinusoidal_pos_embedding (pi0_pytorch.py:25-42): f64 arithmetic, cast to f32 at the end.
// out[b] = [sin(s_i * t_b) | cos(s_i * t_b)], s_i = 2*pi / (min * (max/min)^(i/(n-1))), i < n = dim/2.
__global__ __launch_bounds__(256) void time_sincos_kernel(const float* __restrict__ time, float* __restrict__ out, int B,
                                                          int dim, double min_period, double max_period) {
    const int half = dim >> 1;
    const int total = B * half;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int b = i / half, j = i - b * half;
        // torch.linspace(0, 1, half, float64): start + step*j for j < half/2, end - step*(half-1-j) otherwise
        const double step = half > 1 ? 1.0 / (double)(half - 1) : 0.0;
        const double frac = (j < half / 2) ? step * (double)j : 1.0 - step * (double)(half - 1 - j);
        const double period = min_period * pow(max_period / min_period, frac);
        const double scaling = 1.0 / period * 2.0 * 3.141592653589793;
        const double x = scaling * (double)time[b];
        out[(int64_t)b * dim + j] = (float)sin(x);
        out[(int64_t)b * dim + half + j] = (float)cos(x);
    }
}
__global__ __launch_bounds__(256) void flow_mix_kernel(const float* __restrict__ noise, const float* __restrict__ act,
                                                       const float* __restrict__ time, float* __restrict__ xt,
                                                       float* __restrict__ ut, int B, int HA) {
    const int64_t total = (int64_t)B * HA;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float t = time[i / HA];
        const float nz = noise[i], a = act[i];
        xt[i] = t * nz + (1.0f - t) * a;
        ut[i] = nz - a;
    }
}
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                      float* __restrict__ loss, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = u[i] - v[i];
        loss[i] = d * d;
    }
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                      const float* __restrict__ dl, float* __restrict__ dv, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dv[i] = -2.0f * (u[i] - v[i]) * dl[i];
}
__global__ __launch_bounds__(256) void euler_kernel(float* __restrict__ x, const float* __restrict__ v, float dt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = x[i] + dt * v[i];
}

}  // namespace

#define S_(st) ((hipStream_t)(st))

KAI0_API int kai0_rope_inplace(void* x, const int32_t* pos, const float* inv_freq, int B, int S, int64_t s_ld_rows,
                               int64_t row0, int H, int HD, int inverse, kai0_stream_t stream) {
    KAI0_REQUIRE(HD % 8 == 0 && HD >= 8 && HD <= 2048, "kai0_rope_inplace: HD=%d unsupported", HD);
    KAI0_REQUIRE(256 % (HD / 8) == 0, "kai0_rope_inplace: HD/8 must divide 256 (HD=%d)", HD);
    if (B * S <= 0) return 0;
    const int rpb = 256 / (HD / 8);
    const int64_t row = (int64_t)H * HD;
    hipLaunchKernelGGL(rope_kernel, dim3(ew_grid((int64_t)B * S, rpb)), dim3(256), 0, S_(stream), (bf16_t*)x + row0 * row,
                       (const bf16_t*)x + row0 * row, pos, inv_freq, B, S, s_ld_rows * row, row, s_ld_rows * row, row, (int64_t)S, H,
                       HD, inverse);
    return kai0_check_launch("kai0_rope_inplace");
}

KAI0_API int kai0_rope_inplace2(void* x, int H, void* x2, int H2, const int32_t* pos, const float* inv_freq, int B, int S,
                                int64_t s_ld_rows, int64_t row0, int HD, kai0_stream_t stream) {
    KAI0_REQUIRE(x && x2 && pos && inv_freq, "kai0_rope_inplace2: null operand");
    KAI0_REQUIRE(HD % 8 == 0 && HD >= 8 && HD <= 2048 && 256 % (HD / 8) == 0, "kai0_rope_inplace2: HD=%d unsupported", HD);
    if (B * S <= 0) return 0;
    const int rpb = 256 / (HD / 8);
    const int64_t r1 = (int64_t)H * HD, r2 = (int64_t)H2 * HD;
    hipLaunchKernelGGL(rope_kernel, dim3(ew_grid((int64_t)B * S, rpb)), dim3(256), 0, S_(stream), (bf16_t*)x + row0 * r1,
                       (const bf16_t*)x + row0 * r1, pos, inv_freq, B, S, s_ld_rows * r1, r1, s_ld_rows * r1, r1, (int64_t)S, H, HD, 0,
                       (bf16_t*)x2 + row0 * r2, s_ld_rows * r2, r2, H2);
    return kai0_check_launch("kai0_rope_inplace2");
}

KAI0_API int kai0_rope_copy(const void* src, void* dst, const int32_t* pos, const float* inv_freq, int B, int S, int H, int HD,
                            int64_t src_bs, int64_t src_ld, int64_t dst_bs, int64_t dst_ld, int64_t pos_bs, int inverse,
                            kai0_stream_t stream) {
    KAI0_REQUIRE(src && dst && pos && inv_freq, "kai0_rope_copy: null operand");
    KAI0_REQUIRE(HD % 8 == 0 && HD >= 8 && HD <= 2048 && 256 % (HD / 8) == 0, "kai0_rope_copy: HD=%d unsupported", HD);
    KAI0_REQUIRE(src_ld % 4 == 0 && dst_ld % 4 == 0 && src_bs % 4 == 0 && dst_bs % 4 == 0 && ((uintptr_t)src % 8) == 0 &&
                     ((uintptr_t)dst % 8) == 0, "kai0_rope_copy: strides / bases must keep 8-byte alignment");
    if (B * S <= 0) return 0;
    const int rpb = 256 / (HD / 8);
    hipLaunchKernelGGL(rope_kernel, dim3(ew_grid((int64_t)B * S, rpb)), dim3(256), 0, S_(stream), (bf16_t*)dst, (const bf16_t*)src,
                       pos, inv_freq, B, S, dst_bs, dst_ld, src_bs, src_ld, pos_bs, H, HD, inverse);
    return kai0_check_launch("kai0_rope_copy");
}

KAI0_API int kai0_softmax_mask_fwd(const void* scores, void* probs, const int32_t* qcode, const int32_t* kcode, int B,
                                   int Sq, int H, int Sk, int64_t ld, int64_t batch_stride, int q0, int64_t qcode_ld,
                                   int64_t kcode_ld, kai0_stream_t stream) {
    KAI0_REQUIRE(ld % 8 == 0 && ld <= 4096 && Sk <= ld, "kai0_softmax_mask_fwd: ld=%lld (Sk=%d) unsupported",
                 (long long)ld, Sk);
    KAI0_REQUIRE((qcode == nullptr) == (kcode == nullptr), "kai0_softmax_mask_fwd: qcode/kcode must both be set");
    const int64_t rows = (int64_t)B * Sq * H;
    if (rows <= 0) return 0;
#define KAI0_SM_LAUNCH(N)                                                                                          \
    hipLaunchKernelGGL(softmax_mask_fwd_kernel<N>, dim3(ew_grid(rows, 4)), dim3(256), 0, S_(stream),              \
                       (const bf16_t*)scores, (bf16_t*)probs, qcode, kcode, B, Sq, H, Sk, ld, batch_stride, q0,    \
                       qcode_ld, kcode_ld)
    if (ld <= 512) KAI0_SM_LAUNCH(1);
    else if (ld <= 1024) KAI0_SM_LAUNCH(2);
    else if (ld <= 2048) KAI0_SM_LAUNCH(4);
    else KAI0_SM_LAUNCH(8);
#undef KAI0_SM_LAUNCH
    return kai0_check_launch("kai0_softmax_mask_fwd");
}

KAI0_API int kai0_rowdot_bf16(const void* a, const void* b, float* out, int64_t rows, int D, kai0_stream_t stream) {
    KAI0_REQUIRE(D % 8 == 0 && D >= 8 && D <= 512, "kai0_rowdot_bf16: D=%d must be a multiple of 8, <= 512", D);
    if (rows <= 0) return 0;
    int span = 1;
    while (span < D / 8) span <<= 1;
    hipLaunchKernelGGL(rowdot_kernel, dim3(ew_grid(rows, 4 * (64 / span))), dim3(256), 0, S_(stream), (const bf16_t*)a,
                       (const bf16_t*)b, out, rows, D);
    return kai0_check_launch("kai0_rowdot_bf16");
}

KAI0_API int kai0_softmax_bwd(const void* probs, const void* dprobs, int dprobs_f32, void* dscores, int64_t rows, int Sk,
                              int64_t ld, float scale, kai0_stream_t stream) {
    KAI0_REQUIRE(ld % 8 == 0 && ld <= 4096 && Sk <= ld, "kai0_softmax_bwd: ld=%lld unsupported", (long long)ld);
    if (rows <= 0) return 0;
    if (dprobs_f32)
        hipLaunchKernelGGL((softmax_bwd_kernel<true>), dim3(ew_grid(rows, 4)), dim3(256), 0, S_(stream),
                           (const bf16_t*)probs, dprobs, (bf16_t*)dscores, rows, Sk, ld, scale);
    else
        hipLaunchKernelGGL((softmax_bwd_kernel<false>), dim3(ew_grid(rows, 4)), dim3(256), 0, S_(stream),
                           (const bf16_t*)probs, dprobs, (bf16_t*)dscores, rows, Sk, ld, scale);
    return kai0_check_launch("kai0_softmax_bwd");
}

KAI0_API int kai0_geglu_fwd(const void* g, const void* u, void* h, int64_t n, kai0_stream_t stream) {
    KAI0_REQUIRE(n % 8 == 0, "kai0_geglu_fwd: n must be a multiple of 8");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, S_(stream), (const bf16_t*)g,
                       (const bf16_t*)u, (bf16_t*)h, n / 8);
    return kai0_check_launch("kai0_geglu_fwd");
}
KAI0_API int kai0_geglu_bwd(const void* dh, const void* g, const void* u, void* dg, void* du, int64_t n,
                            kai0_stream_t stream) {
    KAI0_REQUIRE(n % 8 == 0, "kai0_geglu_bwd: n must be a multiple of 8");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, S_(stream), (const bf16_t*)dh,
                       (const bf16_t*)g, (const bf16_t*)u, (bf16_t*)dg, (bf16_t*)du, n / 8);
    return kai0_check_launch("kai0_geglu_bwd");
}
KAI0_API int kai0_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, kai0_stream_t stream) {
    KAI0_REQUIRE(n % 8 == 0, "kai0_gelu_bwd: n must be a multiple of 8");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, S_(stream), (const bf16_t*)dy,
                       (const bf16_t*)pre, (bf16_t*)dx, n / 8);
    return kai0_check_launch("kai0_gelu_bwd");
}
KAI0_API int kai0_silu_fwd_f32(const float* x, float* y, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(silu_fwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_(stream), x, y, n);
    return kai0_check_launch("kai0_silu_fwd_f32");
}
KAI0_API int kai0_silu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_(stream), dy, x, dx, n);
    return kai0_check_launch("kai0_silu_bwd_f32");
}
KAI0_API int kai0_gated_fwd(const void* x, const void* y, const void* gate, void* out, int64_t rows,
                            int rows_per_batch, int D, kai0_stream_t stream) {
    KAI0_REQUIRE(D % 8 == 0 && rows_per_batch > 0, "kai0_gated_fwd: bad D=%d / rows_per_batch=%d", D, rows_per_batch);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(gated_fwd_kernel, dim3(ew_grid(rows * (D / 8), 256)), dim3(256), 0, S_(stream), (const bf16_t*)x,
                       (const bf16_t*)y, (const bf16_t*)gate, (bf16_t*)out, rows, rows_per_batch, D);
    return kai0_check_launch("kai0_gated_fwd");
}
KAI0_API int kai0_gated_bwd(const void* dout, const void* y, const void* gate, void* dy, void* dgate, int64_t rows,
                            int rows_per_batch, int D, kai0_stream_t stream) {
    KAI0_REQUIRE(D % 8 == 0, "kai0_gated_bwd: D must be a multiple of 8");
    KAI0_REQUIRE(rows_per_batch > 0 && rows % rows_per_batch == 0, "kai0_gated_bwd: rows %% rows_per_batch != 0");
    const int B = (int)(rows / rows_per_batch);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(gated_bwd_kernel, dim3(B), dim3(256), 0, S_(stream), (const bf16_t*)dout, (const bf16_t*)y,
                       (const bf16_t*)gate, (bf16_t*)dy, (bf16_t*)dgate, rows_per_batch, D);
    return kai0_check_launch("kai0_gated_bwd");
}
KAI0_API int kai0_embed_gather(const void* table, const int64_t* tokens, void* out, int B, int T, int D, float scale,
                               int64_t out_bs, int64_t out_row0, int64_t out_ld, kai0_stream_t stream) {
    KAI0_REQUIRE(D % 8 == 0 && out_ld % 8 == 0, "kai0_embed_gather: D and out_ld must be multiples of 8");
    if (B * T <= 0) return 0;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(B * T), dim3(256), 0, S_(stream), (const bf16_t*)table, tokens,
                       (bf16_t*)out, T, D, scale, out_bs, out_row0, out_ld);
    return kai0_check_launch("kai0_embed_gather");
}
KAI0_API int kai0_embed_grad(const void* dout, const int64_t* tokens, void* dtable, int B, int T, int D, float scale,
                             int64_t dout_bs, int64_t dout_row0, int64_t dout_ld, kai0_stream_t stream) {
    KAI0_REQUIRE(D % 8 == 0 && dout_ld % 8 == 0, "kai0_embed_grad: D and dout_ld must be multiples of 8");
    if (B * T <= 0) return 0;
    KAI0_REQUIRE((int64_t)B * T <= 32 * 16000, "kai0_embed_grad: B*T = %lld tokens exceed the occurrence bitmap (512000)", (long long)B * T);
    hipLaunchKernelGGL(embed_grad_kernel, dim3(B * T), dim3(256), (size_t)((B * T + 31) / 32) * 4, S_(stream), (const bf16_t*)dout, tokens,
                       (bf16_t*)dtable, B * T, T, D, scale, dout_bs, dout_row0, dout_ld);
    return kai0_check_launch("kai0_embed_grad");
}
KAI0_API int kai0_cast_f32_to_bf16(const float* x, void* y, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    KAI0_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "kai0_cast_f32_to_bf16: unaligned");
    hipLaunchKernelGGL(cast_f2b_kernel, dim3(ew_grid(n / 8 + 1, 256)), dim3(256), 0, S_(stream), x, (bf16_t*)y, n);
    return kai0_check_launch("kai0_cast_f32_to_bf16");
}
KAI0_API int kai0_cast_bf16_to_f32(const void* x, float* y, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    KAI0_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "kai0_cast_bf16_to_f32: unaligned");
    hipLaunchKernelGGL(cast_b2f_kernel, dim3(ew_grid(n / 8 + 1, 256)), dim3(256), 0, S_(stream), (const bf16_t*)x, y, n);
    return kai0_check_launch("kai0_cast_bf16_to_f32");
}
KAI0_API int kai0_add_bf16(const void* a, const void* b, void* out, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(ew_grid(n / 8 + 1, 256)), dim3(256), 0, S_(stream), (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)out, n);
    return kai0_check_launch("kai0_add_bf16");
}
KAI0_API int kai0_add_f32(const float* a, const float* b, float* out, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_f32_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_(stream), a, b, out, n);
    return kai0_check_launch("kai0_add_f32");
}
KAI0_API int kai0_copy_rows_bf16(const void* src, void* dst, int B, int rows, int D, int64_t src_bs, int64_t src_row0,
                                 int64_t src_ld, int64_t dst_bs, int64_t dst_row0, int64_t dst_ld,
                                 kai0_stream_t stream) {
    KAI0_REQUIRE(D % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0, "kai0_copy_rows_bf16: D/ld must be multiples of 8");
    if ((int64_t)B * rows <= 0) return 0;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(B * rows), dim3(256), 0, S_(stream), (const bf16_t*)src, (bf16_t*)dst,
                       rows, D, src_bs, src_row0, src_ld, dst_bs, dst_row0, dst_ld);
    return kai0_check_launch("kai0_copy_rows_bf16");
}
KAI0_API int kai0_pack_rows(const kai0_pack_part* parts, int n, const int32_t* pos, int64_t pos_bs, const float* inv_freq, int HD,
                            kai0_stream_t stream) {
    KAI0_REQUIRE(parts != nullptr && n > 0 && n <= 12, "kai0_pack_rows: 1..12 parts (n=%d)", n);
    KAI0_REQUIRE(HD % 8 == 0 && HD >= 8 && HD <= 2048 && 256 % (HD / 8) == 0, "kai0_pack_rows: HD=%d unsupported", HD);
    PackArgs a;
    const int rpb = 256 / (HD / 8);
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const kai0_pack_part& pt = parts[i];
        KAI0_REQUIRE(pt.mode >= 0 && pt.mode <= 3 && pt.dst != nullptr && (pt.mode == 3 || pt.src != nullptr), "kai0_pack_rows: part %d", i);
        KAI0_REQUIRE(pt.cols % 8 == 0 && pt.src_ld % 8 == 0 && pt.dst_ld % 8 == 0 && pt.src_bs % 8 == 0 && pt.dst_bs % 8 == 0 &&
                         ((uintptr_t)pt.src % 16) == 0 && ((uintptr_t)pt.dst % 16) == 0,
                     "kai0_pack_rows: part %d: columns / strides must be multiples of 8 elements, bases 16-byte aligned", i);
        if (pt.mode == 1 || pt.mode == 2)
            KAI0_REQUIRE(pos != nullptr && inv_freq != nullptr && pt.cols % HD == 0, "kai0_pack_rows: part %d rotates: positions, inv_freq, cols %% HD", i);
        a.part[i] = pt;
        a.first[i] = blocks;
        const int64_t rows = (int64_t)pt.B * pt.rows;
        blocks += rows > 0 ? (int)((rows + rpb - 1) / rpb) : 0;
    }
    for (int i = n; i <= 12; ++i) a.first[i] = blocks;
    if (blocks == 0) return 0;
    a.pos = pos; a.inv_freq = inv_freq; a.pos_bs = pos_bs; a.n = n; a.HD = HD;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, S_(stream), a);
    return kai0_check_launch("kai0_pack_rows");
}
KAI0_API int kai0_sampled_checksum(const kai0_ck_item* items_dev, int n, int stride, unsigned long long* out_dev, kai0_stream_t stream) {
    KAI0_REQUIRE(items_dev != nullptr && out_dev != nullptr && n > 0 && stride > 0, "kai0_sampled_checksum: empty");
    hipLaunchKernelGGL(sampled_checksum_kernel, dim3(2, n < 512 ? n : 512), dim3(256), 0, S_(stream), items_dev, n, stride, out_dev);
    return kai0_check_launch("kai0_sampled_checksum");
}
KAI0_API int kai0_transpose_bf16(const void* src, void* dst, int R, int C, kai0_stream_t stream) {
    KAI0_REQUIRE(R % 8 == 0 && C % 8 == 0, "kai0_transpose_bf16: R=%d and C=%d must be multiples of 8", R, C);
    if (R <= 0 || C <= 0) return 0;
    dim3 grid((C + 63) / 64, (R + 63) / 64, 1);
    KAI0_REQUIRE(grid.y <= 65535, "kai0_transpose_bf16: R=%d too large", R);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, S_(stream), (const bf16_t*)src, (bf16_t*)dst, R, C, (int64_t)C,
                       (int64_t)R, (int64_t)0, (int64_t)0);
    return kai0_check_launch("kai0_transpose_bf16");
}
KAI0_API int kai0_transpose_strided_bf16(const void* src, void* dst, int R, int C, int64_t src_ld, int64_t dst_ld, int batch,
                                         int64_t src_bs, int64_t dst_bs, kai0_stream_t stream) {
    KAI0_REQUIRE(R % 8 == 0 && C % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0 && src_bs % 8 == 0 && dst_bs % 8 == 0,
                 "kai0_transpose_strided_bf16: dimensions and strides must be multiples of 8");
    if (R <= 0 || C <= 0 || batch <= 0) return 0;
    dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
    KAI0_REQUIRE(grid.y <= 65535 && batch <= 65535, "kai0_transpose_strided_bf16: R=%d / batch=%d too large", R, batch);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, S_(stream), (const bf16_t*)src, (bf16_t*)dst, R, C, src_ld,
                       dst_ld, src_bs, dst_bs);
    return kai0_check_launch("kai0_transpose_strided_bf16");
}
KAI0_API int kai0_patch_im2col(const float* img, float* cols, int n_img, int C, int HW, int P, kai0_stream_t stream) {
    KAI0_REQUIRE(HW % P == 0, "kai0_patch_im2col: image size %d not divisible by patch %d", HW, P);
    const int64_t total = (int64_t)n_img * (HW / P) * (HW / P) * C * P * P;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(im2col_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S_(stream), img, cols, n_img, C, HW, P);
    return kai0_check_launch("kai0_patch_im2col");
}
KAI0_API int kai0_add_pos_cast(const float* x, const float* pos, void* out, int64_t rows, int n_pos, int D,
                               kai0_stream_t stream) {
    if (rows * D <= 0) return 0;
    hipLaunchKernelGGL(add_pos_cast_kernel, dim3(ew_grid(rows * D, 256)), dim3(256), 0, S_(stream), x, pos,
                       (bf16_t*)out, rows, n_pos, D);
    return kai0_check_launch("kai0_add_pos_cast");
}
KAI0_API int kai0_time_sincos(const float* time, float* out, int B, int dim, double min_period, double max_period,
                              kai0_stream_t stream) {
    KAI0_REQUIRE(dim % 2 == 0, "kai0_time_sincos: dimension (%d) must be divisible by 2", dim);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(time_sincos_kernel, dim3(ew_grid((int64_t)B * dim / 2, 256)), dim3(256), 0, S_(stream), time, out,
                       B, dim, min_period, max_period);
    return kai0_check_launch("kai0_time_sincos");
}
KAI0_API int kai0_flow_mix(const float* noise, const float* actions, const float* time, float* x_t, float* u_t, int B,
                           int HA, kai0_stream_t stream) {
    if ((int64_t)B * HA <= 0) return 0;
    hipLaunchKernelGGL(flow_mix_kernel, dim3(ew_grid((int64_t)B * HA, 256)), dim3(256), 0, S_(stream), noise, actions,
                       time, x_t, u_t, B, HA);
    return kai0_check_launch("kai0_flow_mix");
}
KAI0_API int kai0_mse_fwd(const float* u, const float* v, float* loss, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_(stream), u, v, loss, n);
    return kai0_check_launch("kai0_mse_fwd");
}
KAI0_API int kai0_mse_bwd(const float* u, const float* v, const float* dloss, float* dv, int64_t n,
                          kai0_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_(stream), u, v, dloss, dv, n);
    return kai0_check_launch("kai0_mse_bwd");
}
KAI0_API int kai0_euler_step(float* x, const float* v, float dt, int64_t n, kai0_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(euler_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_(stream), x, v, dt, n);
    return kai0_check_launch("kai0_euler_step");
}

// ---- mask codes + position ids of one request, one launch (kai0hip.h kai0_prefix_codes) -----------------------------------------
namespace {
struct CodesArgs {
    const uint8_t* img[8];
    const uint8_t* lang;
    int ncam, n_img, T, Hs;
};
__global__ __launch_bounds__(256) void prefix_codes_kernel(const CodesArgs a, int32_t* __restrict__ qcode, int32_t* __restrict__ kcode,
                                                           int32_t* __restrict__ pos) {
    __shared__ int wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P_img = a.ncam * a.n_img, P = P_img + a.T, S = P + a.Hs;
    const int per = (S + 255) / 256, j0 = tid * per, j1 = min(S, j0 + per);
    auto padded = [&](int j) -> bool {
        if (j < P_img) return a.img[j / a.n_img][b] != 0;
        if (j < P) return a.lang[(int64_t)b * a.T + (j - P_img)] != 0;
        return true;
    };
    int cnt = 0;
    for (int j = j0; j < j1; ++j) cnt += padded(j) ? 1 : 0;
    // exclusive scan of the threads' counts: within the wave by shuffles, across the four waves through LDS
    int inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - cnt;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int run = base;
    for (int j = j0; j < j1; ++j) {
        const bool pd = padded(j);
        run += pd ? 1 : 0;
        const int cum = j >= P ? 1 : 0;
        const int64_t o = (int64_t)b * S + j;
        qcode[o] = pd ? cum : -1;
        kcode[o] = pd ? cum : INT_MAX;
        pos[o] = run - 1;
    }
}
}  // namespace

KAI0_API int kai0_prefix_codes(const void* const* img_masks, int ncam, const void* lang_mask, int B, int n_img, int T, int Hs,
                               int32_t* qcode, int32_t* kcode, int32_t* pos, kai0_stream_t stream) {
    KAI0_REQUIRE(img_masks && lang_mask && qcode && kcode && pos, "kai0_prefix_codes: null operand");
    KAI0_REQUIRE(ncam >= 1 && ncam <= 8 && n_img >= 1 && T >= 0 && Hs >= 0 && B >= 1, "kai0_prefix_codes: ncam=%d n_img=%d T=%d Hs=%d B=%d",
                 ncam, n_img, T, Hs, B);
    CodesArgs a{};
    for (int c = 0; c < ncam; ++c) {
        KAI0_REQUIRE(img_masks[c] != nullptr, "kai0_prefix_codes: null camera mask %d", c);
        a.img[c] = (const uint8_t*)img_masks[c];
    }
    a.lang = (const uint8_t*)lang_mask;
    a.ncam = ncam; a.n_img = n_img; a.T = T; a.Hs = Hs;
    hipLaunchKernelGGL(prefix_codes_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, qcode, kcode, pos);
    return kai0_check_launch("kai0_prefix_codes");
}
