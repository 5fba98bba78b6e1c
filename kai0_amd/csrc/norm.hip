// norm.hip — RMSNorm / adaRMSNorm / LayerNorm forward+backward and the column reductions they need.
// All HBM-bound: one wave64 per row, 16-B (bf16x8) loads, f32 statistics, rows re-read from L2 only.
// Reference semantics: GemmaRMSNorm (modeling_gemma.py:49-104), nn.LayerNorm in SigLIP
// (modeling_siglip.py:439-441,756).
#include "common.h"
#include "../../include/kai0hip.h"

namespace {

constexpr int MAXC = 4;  // chunks of 8 per lane: D <= 64*8*4 = 2048

__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void loadf8(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// ------------------------------------------------------------------------------------------ RMSNorm
// ADA=false: y = bf16((x*rstd)*(1+w));  ADA=true: y = bf16((x*rstd)*(1+scale_b)+shift_b), gate_out=bf16(gate_b)
template <bool ADA>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ mod, bf16_t* __restrict__ y,
                                                          bf16_t* __restrict__ gate_out, float* __restrict__ rstd_out,
                                                          int64_t rows, int rpb, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int nchunk = D >> 3;
    // plain RMSNorm: the weight does not depend on the row — requested once, together with the first row (not behind its reduction);
    // the row itself with a clamped chunk index instead of a guard (see layernorm_fwd_kernel)
    float wv[MAXC][8];
    if constexpr (!ADA) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) loadf8(w + min(c * 64 + lane, nchunk - 1) * 8, wv[c]);
    }
    for (int64_t row = wg; row < rows; row += nw) {
        const bf16_t* xr = x + row * D;
        float xv[MAXC][8];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) load8(xr + min(c * 64 + lane, nchunk - 1) * 8, xv[c]);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const bool live = c * 64 + lane < nchunk;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += live ? xv[c][e] * xv[c][e] : 0.f;
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)D + eps);
        const float* mrow = nullptr;
        if constexpr (ADA) mrow = mod + (row / rpb) * (int64_t)(3 * D);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float o[8];
                if constexpr (ADA) {
                    float sc[8], sh[8];
                    loadf8(mrow + ci * 8, sc);
                    loadf8(mrow + D + ci * 8, sh);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (xv[c][e] * rstd) * (1.0f + sc[e]) + sh[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (xv[c][e] * rstd) * (1.0f + wv[c][e]);
                }
                store8(y + row * D + ci * 8, o);
                if constexpr (ADA) {
                    if (gate_out != nullptr && (row % rpb) == 0) {
                        float gt[8];
                        loadf8(mrow + 2 * D + ci * 8, gt);
                        store8(gate_out + (row / rpb) * (int64_t)D + ci * 8, gt);
                    }
                }
            }
        }
        if (lane == 0 && rstd_out != nullptr) rstd_out[row] = rstd;
    }
}

// The seam between two Euler steps of the denoise loop in one launch (pi0_pytorch.py:401-461):
//   [final adaRMS norm of the expert's last residual stream -> action_out_proj (f32) -> x_t += dt * v_t]   (closes step s)
//   [action_in_proj (f32) of the new x_t -> bf16 suffix embedding]                                          (opens step s + 1)
// — six launches on the generic path (adarms, cast, f32 GEMM, euler, f32 GEMM, cast) whose data is a 50 x 32 action block.
// One block per row.
// Rounding points as those kernels: y = bf16((x * rstd) * (1 + scale) + shift) with the row statistics summed exactly like
// rmsnorm_fwd_kernel (same lane partials, same wave reduction); v = f32 dot of bf16-rounded y with W_out + b; x_t = x_t + dt * v;
// a = f32 dot + b, stored as bf16.  (The f32 dots run k-ascending per lane and are then summed over the wave — another summation
// order than the MFMA f32 GEMM's, i.e. equal to ~1e-7 relative, not bit for bit.)
__global__ __launch_bounds__(256) void denoise_glue_kernel(const bf16_t* __restrict__ xs, const float* __restrict__ mod, int64_t mod_ld,
                                                           int rpb, float eps, const float* __restrict__ w_out,
                                                           const float* __restrict__ b_out, float* __restrict__ x_t, float dt,
                                                           const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                           bf16_t* __restrict__ xs_next, int64_t rows, int D, int A,
                                                           float* __restrict__ rowsq_next) {
    // one block per action row; the chain is a handful of dependent memory round trips, so everything independent is in flight at
    // once: each of the four waves normalises the row itself (16 elements per lane) and takes A / 4 of the outputs of the first dot
    // with all their weight loads issued up front; the second dot gives every thread D / 256 outputs
    __shared__ float xa_s[64], x0_s[64], bo_s[64], red_s[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    float ssn = 0.f;  // sum of squares of this thread's share of the new bf16 row (the next layer's folded adaRMS statistic)
    const int nchunk = D >> 3;
    if (tid < A) {  // requested at kernel entry, read after the dots
        x0_s[tid] = x_t[row * A + tid];
        bo_s[tid] = xs != nullptr ? b_out[tid] : 0.f;
    }
    __syncthreads();
    if (xs != nullptr) {
        const bf16_t* xr = xs + row * D;
        float xv[MAXC][8];
        const float* mrow = mod + (row / rpb) * mod_ld;
        if (A == 32 && D == 1024) {
            // the row, its scale and its shift: six unconditional 16 / 32-byte loads per lane, all in flight before the statistics
            float sc[2][8], sh[2][8];
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                load8(xr + (c * 64 + lane) * 8, xv[c]);
                loadf8(mrow + (c * 64 + lane) * 8, sc[c]);
                loadf8(mrow + 1024 + (c * 64 + lane) * 8, sh[c]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += xv[c][e] * xv[c][e];
            ss = wave_sum(ss);
            const float rstd = rsqrtf(ss / 1024.0f + eps);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[c][e] = rbf((xv[c][e] * rstd) * (1.0f + sc[c][e]) + sh[c][e]);
        } else {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                load8(xr + ci * 8, xv[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += xv[c][e] * xv[c][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[c][e] = 0.f;
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)D + eps);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float sc[8], sh[8];
                loadf8(mrow + ci * 8, sc);
                loadf8(mrow + D + ci * 8, sh);
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[c][e] = rbf((xv[c][e] * rstd) * (1.0f + sc[e]) + sh[e]);
            }
        }
        }
        // v[j] = <y, W_out[j]> + b_out[j]; wave w takes outputs j = w, w + 4, ... in groups of four
        if (A == 32 && D == 1024) {
            // pi0.5's shape, no run-time conditions around the loads (hipcc branches around a conditional load and waits for each one:
            // sixteen dependent round trips): wave w owns outputs w, w + 4, ..., w + 28; 32 16-B loads per lane in flight
            float acc[8];
            f32x4 wv[8][2][2];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float* wp = w_out + (int64_t)(wave + 4 * u) * 1024 + (c * 64 + lane) * 8;
                    wv[u][c][0] = *reinterpret_cast<const f32x4*>(wp);
                    wv[u][c][1] = *reinterpret_cast<const f32x4*>(wp + 4);
                }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a += xv[c][e] * wv[u][c][e >> 2][e & 3];
                acc[u] = a;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float t = wave_sum(acc[u]);
                const int j = wave + 4 * u;
                if (lane == 0) xa_s[j] = x0_s[j] + dt * (t + bo_s[j]);
            }
        } else
        for (int j0 = wave; j0 < A; j0 += 16) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u;
                if (j < A) {
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int ci = c * 64 + lane;
                        if (ci < nchunk) {
                            float wv[8];
                            loadf8(w_out + (int64_t)j * D + ci * 8, wv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[u] += xv[c][e] * wv[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u;
                const float t = wave_sum(acc[u]);
                if (lane == 0 && j < A) xa_s[j] = x0_s[j] + dt * (t + bo_s[j]);
            }
        }
        __syncthreads();
        if (tid < A) x_t[row * A + tid] = xa_s[tid];
    } else {
        if (tid < A) xa_s[tid] = x0_s[tid];
        __syncthreads();
    }
    if (xs_next == nullptr) return;
    // a[n] = <x_t, W_in[n]> + b_in[n]; A == 32 (pi0.5): the 128-B weight row as eight 16-B loads, four outputs per thread, all
    // loads of a thread in flight before the first multiply
    if (A == 32 && D == 1024) {
        f32x4 w[4][8];
        float bi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = tid + 256 * u;
            bi[u] = b_in[n];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[u][q] = *reinterpret_cast<const f32x4*>(w_in + (int64_t)n * 32 + 4 * q);
        }
        float xa[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) xa[j] = xa_s[j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += xa[4 * q + e] * w[u][q][e];
            const bf16_t o = f2bf(acc + bi[u]);
            xs_next[row * D + tid + 256 * u] = o;
            ssn += bf2f(o) * bf2f(o);
        }
        if (rowsq_next != nullptr) {
            const float t = block_sum<4>(ssn, red_s);
            if (tid == 0) rowsq_next[row] = t;
        }
        return;
    }
    for (int n = tid; n < D; n += 256) {
        const float* wr = w_in + (int64_t)n * A;
        float acc = 0.f;
        for (int j = 0; j < A; ++j) acc += xa_s[j] * wr[j];
        const bf16_t o = f2bf(acc + b_in[n]);
        xs_next[row * D + n] = o;
        ssn += bf2f(o) * bf2f(o);
    }
    if (rowsq_next != nullptr) {
        const float t = block_sum<4>(ssn, red_s);
        if (tid == 0) rowsq_next[row] = t;
    }
}

// split-K combine + gated residual + adaRMS (inference denoise loop): one wave per row
__global__ __launch_bounds__(256) void adarms_combine_kernel(const float* __restrict__ partials, int splits,
                                                             int64_t split_stride, const bf16_t* __restrict__ gate_prev,
                                                             const bf16_t* __restrict__ residual, bf16_t* __restrict__ x_out,
                                                             const float* __restrict__ mod, bf16_t* __restrict__ y,
                                                             bf16_t* __restrict__ gate_out, int64_t rows, int rpb, int D,
                                                             float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = D >> 3;
    const int64_t b = row / rpb;
    float xv[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
        if (ci < nchunk) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* pp = partials + row * D + ci * 8;
            for (int s = 0; s < splits; ++s) {
                float t[8];
                loadf8(pp + (int64_t)s * split_stride, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += t[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = rbf(acc[e]);
            if (gate_prev != nullptr) {
                float g[8];
                load8(gate_prev + b * D + ci * 8, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = rbf(acc[e] * g[e]);
            }
            if (residual != nullptr) {
                float r[8];
                load8(residual + row * D + ci * 8, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = rbf(acc[e] + r[e]);
            }
            store8(x_out + row * D + ci * 8, acc);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xv[c][e] = acc[e];
                ss += acc[e] * acc[e];
            }
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
    const float* mrow = mod + b * (int64_t)(3 * D);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
        if (ci < nchunk) {
            float o[8], sc[8], sh[8];
            loadf8(mrow + ci * 8, sc);
            loadf8(mrow + D + ci * 8, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (xv[c][e] * rstd) * (1.0f + sc[e]) + sh[e];
            store8(y + row * D + ci * 8, o);
            if (gate_out != nullptr && (row % rpb) == 0) {
                float gt[8];
                loadf8(mrow + 2 * D + ci * 8, gt);
                store8(gate_out + b * (int64_t)D + ci * 8, gt);
            }
        }
    }
}

// plain RMSNorm backward: dx and per-wave dw partials [gridDim.x*4][D]
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ rstd_in,
                                                          bf16_t* __restrict__ dx, float* __restrict__ dw_partial,
                                                          const bf16_t* __restrict__ dres, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int nchunk = D >> 3;
    float dwa[MAXC][8];
    float cw[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dwa[c][e] = 0.f; cw[c][e] = 0.f; }
        if (ci < nchunk) {
            loadf8(w + ci * 8, cw[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) cw[c][e] += 1.0f;
        }
    }
    for (int64_t row = wg; row < rows; row += nw) {
        const float rstd = rstd_in[row];
        float xh[MAXC][8], dxh[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float dyv[8];
                load8(dy + row * D + ci * 8, dyv);
                load8(x + row * D + ci * 8, xh[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[c][e] *= rstd;
                    dxh[c][e] = dyv[e] * cw[c][e];
                    s += dxh[c][e] * xh[c][e];
                    dwa[c][e] += dyv[e] * xh[c][e];
                }
            }
        }
        s = wave_sum(s) / (float)D;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float o[8], rs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[e] = 0.f;
                if (dres != nullptr) load8(dres + row * D + ci * 8, rs);  // gradient of the residual branch of x
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (dxh[c][e] - xh[c][e] * s) + rs[e];
                store8(dx + row * D + ci * 8, o);
            }
        }
    }
    // the block's 4 wave partials meet in LDS: one partial row per block (a quarter of the bytes the column reduction reads)
    extern __shared__ float nred[];  // [4][D]
    float* pr = nred + (threadIdx.x >> 6) * D;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
        if (ci < nchunk) {
            *reinterpret_cast<f32x4*>(pr + ci * 8) = f32x4{dwa[c][0], dwa[c][1], dwa[c][2], dwa[c][3]};
            *reinterpret_cast<f32x4*>(pr + ci * 8 + 4) = f32x4{dwa[c][4], dwa[c][5], dwa[c][6], dwa[c][7]};
        }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256)
        dw_partial[(int64_t)blockIdx.x * D + col] = (nred[col] + nred[D + col]) + (nred[2 * D + col] + nred[3 * D + col]);
}

// adaRMS backward: one block of 8 waves per batch entry b (rows b*rpb .. b*rpb+rpb-1); a wave owns whole rows (lane -> chunks
// c*64 + lane of 8 columns, row statistics by wave shuffles, no block barrier per row) and the 8 waves' dscale / dshift sums
// meet in LDS once at the end.   dx, dmod[b] = [dscale | dshift | dgate] (f32)
// (the former one-row-at-a-time block loop took 80 us for 50 rows: two barriers and a dependent load per row)
__global__ __launch_bounds__(512) void adarms_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dgate,
                                                         const bf16_t* __restrict__ x, const float* __restrict__ mod,
                                                         const float* __restrict__ rstd_in, bf16_t* __restrict__ dx,
                                                         float* __restrict__ dmod, const bf16_t* __restrict__ dres, int rpb,
                                                         int D) {
    extern __shared__ float ared[];  // [8 waves][2][D]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = D >> 3;
    const float* mrow = mod + (int64_t)b * 3 * D;
    float c1[MAXC][8], dsc[MAXC][8], dsh[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) { c1[c][e] = 0.f; dsc[c][e] = 0.f; dsh[c][e] = 0.f; }
        if (ci < nchunk) {
            loadf8(mrow + ci * 8, c1[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) c1[c][e] += 1.0f;
        }
    }
    for (int i = wave; i < rpb; i += 8) {
        const int64_t row = (int64_t)b * rpb + i;
        const float rstd = rstd_in[row];
        float xh[MAXC][8], dxh[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float dyv[8];
                load8(dy + row * D + ci * 8, dyv);
                load8(x + row * D + ci * 8, xh[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[c][e] *= rstd;
                    dxh[c][e] = dyv[e] * c1[c][e];
                    s += dxh[c][e] * xh[c][e];
                    dsc[c][e] += dyv[e] * xh[c][e];
                    dsh[c][e] += dyv[e];
                }
            }
        }
        s = wave_sum(s) / (float)D;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float o[8], rs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[e] = 0.f;
                if (dres != nullptr) load8(dres + row * D + ci * 8, rs);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (dxh[c][e] - xh[c][e] * s) + rs[e];
                store8(dx + row * D + ci * 8, o);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
        if (ci < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ared[(wave * 2 + 0) * D + ci * 8 + e] = dsc[c][e];
                ared[(wave * 2 + 1) * D + ci * 8 + e] = dsh[c][e];
            }
        }
    }
    __syncthreads();
    float* dm = dmod + (int64_t)b * 3 * D;
    for (int col = threadIdx.x; col < 2 * D; col += 512) {  // col < D: dscale, else dshift; waves summed in a fixed order
        const int which = col / D, cc = col - which * D;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += ared[(w * 2 + which) * D + cc];
        dm[col] = t;
    }
    for (int col = threadIdx.x; col < D; col += 512) dm[2 * D + col] = dgate != nullptr ? bf2f(dgate[(int64_t)b * D + col]) : 0.f;
}

// ---------------------------------------------------------------------------------------- LayerNorm
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ bsh, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int nchunk = D >> 3;
    // weight and bias do not depend on the row: requested once, with the first row's data (not behind its two reductions); all loads
    // with a clamped chunk index instead of a guard (a guarded load is a branch with its own wait: at B = 1, one row per wave, the
    // kernel was four dependent round trips long)
    float wv[MAXC][8], bv[MAXC][8];
    bool live[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
        live[c] = ci < nchunk;
        const int cc = min(ci, nchunk - 1);
        load8(w + cc * 8, wv[c]);
        load8(bsh + cc * 8, bv[c]);
    }
    for (int64_t row = wg; row < rows; row += nw) {
        float xv[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) load8(x + row * D + min(c * 64 + lane, nchunk - 1) * 8, xv[c]);
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += live[c] ? xv[c][e] : 0.f;
        const float mean = wave_sum(s) / (float)D;
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv[c][e] - mean;
                v += live[c] ? d * d : 0.f;
            }
        const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (xv[c][e] - mean) * rstd * wv[c][e] + bv[c][e];
                store8(y + row * D + ci * 8, o);
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// dx and per-wave partials [gridDim.x*4][2*D] = [dw | db]
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, bf16_t* __restrict__ dx,
                                                            float* __restrict__ partial, const bf16_t* __restrict__ dres,
                                                            int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int nchunk = D >> 3;
    float dwa[MAXC][8], dba[MAXC][8], wv[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dwa[c][e] = 0.f; dba[c][e] = 0.f; wv[c][e] = 0.f; }
        if (ci < nchunk) load8(w + ci * 8, wv[c]);
    }
    for (int64_t row = wg; row < rows; row += nw) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[MAXC][8], dxh[MAXC][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float dyv[8];
                load8(dy + row * D + ci * 8, dyv);
                load8(x + row * D + ci * 8, xh[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[c][e] = (xh[c][e] - mean) * rstd;
                    dxh[c][e] = dyv[e] * wv[c][e];
                    s1 += dxh[c][e];
                    s2 += dxh[c][e] * xh[c][e];
                    dwa[c][e] += dyv[e] * xh[c][e];
                    dba[c][e] += dyv[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c * 64 + lane;
            if (ci < nchunk) {
                float o[8], rs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[e] = 0.f;
                if (dres != nullptr) load8(dres + row * D + ci * 8, rs);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (dxh[c][e] - s1 - xh[c][e] * s2) + rs[e];
                store8(dx + row * D + ci * 8, o);
            }
        }
    }
    extern __shared__ float nred[];  // [4][2 D]: the block's wave partials -> one partial row per block
    float* pr = nred + (threadIdx.x >> 6) * (2 * D);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = c * 64 + lane;
        if (ci < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pr[ci * 8 + e] = dwa[c][e];
                pr[D + ci * 8 + e] = dba[c][e];
            }
        }
    }
    __syncthreads();
    const int W = 2 * D;
    for (int col = threadIdx.x; col < W; col += 256)
        partial[(int64_t)blockIdx.x * W + col] = (nred[col] + nred[W + col]) + (nred[2 * W + col] + nred[3 * W + col]);
}

// ------------------------------------------------------------------------------------ column reductions
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial, int blocks, int ncols,
                                                               int64_t ld, void* __restrict__ out, int out_f32) {
    // 64 columns x 16 row-groups per block; fixed summation order (deterministic)
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < ncols)
        for (int b = rg; b < blocks; b += 16) s += partial[(int64_t)b * ld + col];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < ncols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        if (out_f32) reinterpret_cast<float*>(out)[col] = t;
        else reinterpret_cast<bf16_t*>(out)[col] = f2bf(t);
    }
}

// The same reduction for up to 32 (partials, destination) pairs in one launch: the final sums of the norm-weight and bias
// gradients are ~9 us launches of 2-4 MB each (254 per training step); ops.py queues them during backward and flushes the
// queue before anything reads the gradients.  Block b belongs to the item whose block range [first[i], first[i+1]) holds it.
struct ReduceBatch {
    kai0_reduce_item it[32];
    int first[33];
};
__global__ __launch_bounds__(1024) void reduce_partials_batch_kernel(ReduceBatch rb, int n) {
    __shared__ float red[16][64];
    int i = 0;
    while (i + 1 < n && (int)blockIdx.x >= rb.first[i + 1]) ++i;
    const kai0_reduce_item it = rb.it[i];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = ((int)blockIdx.x - rb.first[i]) * 64 + cl;
    float s = 0.f;
    if (col < it.ncols)
        for (int b = rg; b < it.blocks; b += 16) s += it.partial[(int64_t)b * it.ld + col];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < it.ncols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        if (it.out_f32) reinterpret_cast<float*>(it.out)[col] = t;
        else reinterpret_cast<bf16_t*>(it.out)[col] = f2bf(t);
    }
}

// Column sums of a [M][N] bf16 matrix (bias gradients): block (x, y) covers the 512 columns [512 x, 512 x + 512) — one wave
// is 64 lanes x 8 columns = 1 KiB of a row — and the rows y*4 + w, stepping by 4*gridDim.y; each wave keeps 4 independent
// row loads in flight, the 4 waves' sums meet in LDS, and the block writes one partial row for reduce_partials.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ dy, int64_t M, int N, int64_t ld,
                                                     float* __restrict__ scratch) {
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 8;
    const bool live = col < N;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int64_t step = (int64_t)gridDim.y * 4;
    int64_t m = (int64_t)blockIdx.y * 4 + wave;
    if (live) {
        for (; m + 3 * step < M; m += 4 * step) {
            float v0[8], v1[8], v2[8], v3[8];
            load8(dy + m * ld + col, v0);
            load8(dy + (m + step) * ld + col, v1);
            load8(dy + (m + 2 * step) * ld + col, v2);
            load8(dy + (m + 3 * step) * ld + col, v3);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        }
        for (; m < M; m += step) {
            float v[8];
            load8(dy + m * ld + col, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane][e] = acc[e];
    __syncthreads();
    if (wave == 0 && live) {
        float* sp = scratch + (int64_t)blockIdx.y * N + col;
#pragma unroll
        for (int e = 0; e < 8; ++e) sp[e] = (red[0][lane][e] + red[1][lane][e]) + (red[2][lane][e] + red[3][lane][e]);
    }
}

inline int norm_grid(int64_t rows) {
    int64_t b = (rows + 3) / 4;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define CHECK_D(name, D) KAI0_REQUIRE((D) % 8 == 0 && (D) > 0 && (D) <= 2048, name ": D=%d must be a multiple of 8, <= 2048", (D))

KAI0_API int kai0_rmsnorm_fwd(const void* x, const float* w, void* y, float* rstd, int64_t rows, int D, float eps,
                              kai0_stream_t stream) {
    CHECK_D("kai0_rmsnorm_fwd", D);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL((rmsnorm_fwd_kernel<false>), dim3(norm_grid(rows)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, w, (const float*)nullptr, (bf16_t*)y, (bf16_t*)nullptr, rstd, rows, 1, D, eps);
    return kai0_check_launch("kai0_rmsnorm_fwd");
}

KAI0_API int kai0_rmsnorm_bwd(const void* dy, const void* x, const float* w, const float* rstd, void* dx,
                              float* dw_partial, int dw_blocks, const void* dres, int64_t rows, int D,
                              kai0_stream_t stream) {
    CHECK_D("kai0_rmsnorm_bwd", D);
    KAI0_REQUIRE(dw_blocks > 0, "kai0_rmsnorm_bwd: dw_blocks must be > 0");
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(dw_blocks), dim3(256), 4 * D * sizeof(float), (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, w, rstd, (bf16_t*)dx, dw_partial, (const bf16_t*)dres, rows, D);
    return kai0_check_launch("kai0_rmsnorm_bwd");
}

KAI0_API int kai0_adarms_fwd(const void* x, const float* mod, void* y, void* gate_out, float* rstd, int64_t rows,
                             int rows_per_batch, int D, float eps, kai0_stream_t stream) {
    CHECK_D("kai0_adarms_fwd", D);
    KAI0_REQUIRE(rows_per_batch > 0, "kai0_adarms_fwd: rows_per_batch must be > 0");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL((rmsnorm_fwd_kernel<true>), dim3(norm_grid(rows)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const float*)nullptr, mod, (bf16_t*)y, (bf16_t*)gate_out, rstd, rows,
                       rows_per_batch, D, eps);
    return kai0_check_launch("kai0_adarms_fwd");
}

KAI0_API int kai0_denoise_glue(const void* xs, const float* mod, int64_t mod_ld, int rows_per_batch, float eps, const float* w_out,
                               const float* b_out, float* x_t, float dt, const float* w_in, const float* b_in, void* xs_next,
                               int64_t rows, int D, int A, float* rowsq_next, kai0_stream_t stream) {
    CHECK_D("kai0_denoise_glue", D);
    KAI0_REQUIRE(x_t && A >= 1 && A <= 64 && (xs == nullptr || (mod && w_out && b_out && rows_per_batch > 0 && mod_ld >= 2 * (int64_t)D)) &&
                     (xs_next == nullptr || (w_in && b_in)) && (xs != nullptr || xs_next != nullptr),
                 "kai0_denoise_glue: bad arguments (A <= 64; closing a step needs xs, mod, w_out, b_out; opening one w_in, b_in, xs_next)");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(denoise_glue_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)xs, mod,
                       mod_ld, rows_per_batch, eps, w_out, b_out, x_t, dt, w_in, b_in, (bf16_t*)xs_next, rows, D, A,
                       xs_next != nullptr ? rowsq_next : nullptr);
    return kai0_check_launch("kai0_denoise_glue");
}

KAI0_API int kai0_adarms_combine(const float* partials, int splits, int64_t split_stride, const void* gate_prev,
                                 const void* residual, void* x_out, const float* mod, void* y, void* gate_out,
                                 int64_t rows, int rows_per_batch, int D, float eps, kai0_stream_t stream) {
    CHECK_D("kai0_adarms_combine", D);
    KAI0_REQUIRE(partials && splits >= 1 && x_out && mod && y && rows_per_batch > 0, "kai0_adarms_combine: bad arguments");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(adarms_combine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, partials,
                       splits, split_stride, (const bf16_t*)gate_prev, (const bf16_t*)residual, (bf16_t*)x_out, mod,
                       (bf16_t*)y, (bf16_t*)gate_out, rows, rows_per_batch, D, eps);
    return kai0_check_launch("kai0_adarms_combine");
}

KAI0_API int kai0_adarms_bwd(const void* dy, const void* dgate, const void* x, const float* mod, const float* rstd,
                             void* dx, float* dmod, const void* dres, int64_t rows, int rows_per_batch, int D,
                             kai0_stream_t stream) {
    CHECK_D("kai0_adarms_bwd", D);
    KAI0_REQUIRE(rows_per_batch > 0 && rows % rows_per_batch == 0, "kai0_adarms_bwd: rows %% rows_per_batch != 0");
    const int B = (int)(rows / rows_per_batch);
    if (B <= 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {  // 8 waves x (dscale, dshift) x D f32 = 128 KiB at D = 2048
        hipError_t e = hipFuncSetAttribute((const void*)adarms_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 2048 * 4);
        KAI0_REQUIRE(e == hipSuccess, "kai0_adarms_bwd: cannot reserve LDS: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(adarms_bwd_kernel, dim3(B), dim3(512), 16 * D * sizeof(float), (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)dgate, (const bf16_t*)x, mod, rstd, (bf16_t*)dx, dmod, (const bf16_t*)dres, rows_per_batch, D);
    return kai0_check_launch("kai0_adarms_bwd");
}

KAI0_API int kai0_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                                int64_t rows, int D, float eps, kai0_stream_t stream) {
    CHECK_D("kai0_layernorm_fwd", D);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(norm_grid(rows)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, D, eps);
    return kai0_check_launch("kai0_layernorm_fwd");
}

KAI0_API int kai0_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                                void* dx, float* dwb_partial, int dwb_blocks, const void* dres, int64_t rows, int D,
                                kai0_stream_t stream) {
    CHECK_D("kai0_layernorm_bwd", D);
    KAI0_REQUIRE(dwb_blocks > 0, "kai0_layernorm_bwd: dwb_blocks must be > 0");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(dwb_blocks), dim3(256), 8 * D * sizeof(float), (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, (const bf16_t*)w, mean, rstd, (bf16_t*)dx, dwb_partial, (const bf16_t*)dres, rows, D);
    return kai0_check_launch("kai0_layernorm_bwd");
}

KAI0_API int kai0_reduce_partials(const float* partial, int blocks, int ncols, int64_t ld, void* out, int out_f32,
                                  kai0_stream_t stream) {
    KAI0_REQUIRE(blocks > 0 && ncols > 0, "kai0_reduce_partials: empty");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((ncols + 63) / 64), dim3(1024), 0, (hipStream_t)stream, partial,
                       blocks, ncols, ld, out, out_f32);
    return kai0_check_launch("kai0_reduce_partials");
}

KAI0_API int kai0_reduce_partials_batch(const kai0_reduce_item* items, int n, kai0_stream_t stream) {
    KAI0_REQUIRE(items != nullptr && n > 0, "kai0_reduce_partials_batch: empty");
    for (int base = 0; base < n; base += 32) {
        const int m = n - base < 32 ? n - base : 32;
        ReduceBatch rb;
        int blocks = 0;
        for (int i = 0; i < m; ++i) {
            rb.it[i] = items[base + i];
            KAI0_REQUIRE(rb.it[i].partial != nullptr && rb.it[i].out != nullptr && rb.it[i].blocks > 0 && rb.it[i].ncols > 0,
                         "kai0_reduce_partials_batch: item %d is empty", base + i);
            rb.first[i] = blocks;
            blocks += (rb.it[i].ncols + 63) / 64;
        }
        for (int i = m; i <= 32; ++i) rb.first[i] = blocks;
        hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, rb, m);
        int rc = kai0_check_launch("kai0_reduce_partials_batch");
        if (rc) return rc;
    }
    return 0;
}

KAI0_API int kai0_colsum_partials_bf16(const void* dy, int64_t M, int N, int64_t ld, float* scratch, int scratch_blocks,
                                       int* blocks_used, kai0_stream_t stream) {
    KAI0_REQUIRE(N % 8 == 0 && ld % 8 == 0, "kai0_colsum_bf16: N=%d and ld must be multiples of 8", N);
    KAI0_REQUIRE(scratch_blocks > 0 && blocks_used != nullptr, "kai0_colsum_bf16: scratch_blocks must be > 0");
    int sb = scratch_blocks;
    if ((int64_t)sb > M) sb = (int)M;
    if ((int64_t)sb * 4 > M) sb = (int)((M + 3) / 4);
    *blocks_used = sb;
    dim3 grid((N / 8 + 63) / 64, sb, 1);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, M, N, ld, scratch);
    return kai0_check_launch("kai0_colsum_bf16");
}

KAI0_API int kai0_colsum_bf16(const void* dy, int64_t M, int N, int64_t ld, float* scratch, int scratch_blocks,
                              void* out, int out_f32, kai0_stream_t stream) {
    int sb = 0;
    int rc = kai0_colsum_partials_bf16(dy, M, N, ld, scratch, scratch_blocks, &sb, stream);
    if (rc) return rc;
    return kai0_reduce_partials(scratch, sb, N, N, out, out_f32, stream);
}
