// gemm_f32.hip — exact-f32 MFMA GEMM (v_mfma_f32_16x16x4_f32) for the f32 islands of the path:
// SigLIP patch-embed (f32 weights, gemma_pytorch.py:72-75), adaRMS `dense` (f32 by substring match,
// gemma_pytorch.py:76-78), time MLP and action in/out projections (pi0_pytorch.py:100-105).
// These are < 1 % of the FLOPs; the kernel is fully strided so forward (x W^T), dgrad (dy W) and
// wgrad (dy^T x) are the same code with different strides.
//
// 64x64x16 block tile, 4 waves (2x2), each wave 32x32 = 2x2 MFMA 16x16x4 tiles, 4 k-steps per tile.
// LDS image As[k][m] / Bs[k][n] with an 80-float row pitch so the two k-rows a 32-lane group reads
// hit disjoint banks.
#include "common.h"
#include "../../include/kai0hip.h"

namespace {

constexpr int FBM = 64, FBN = 64, FBK = 16, PITCH = 80;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                       float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       const float* __restrict__ bias, int accumulate, int k_chunk, float* __restrict__ ws) {
    __shared__ float As[FBK * PITCH];
    __shared__ float Bs[FBK * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;
    const int l15 = lane & 15, g = lane >> 4;

    // loader mapping: pick the memory-fastest index as the thread-fastest one (coalescing)
    const bool a_kfast = (sak == 1);
    const bool b_kfast = (sbk == 1);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-K: grid.z slices own k in [z*k_chunk, min(K, (z+1)*k_chunk)) and write raw partial tiles ws[z][M][N]; the
    // reduction kernel below sums them in slice order (deterministic) and applies bias / accumulate
    const int kbeg = blockIdx.z * k_chunk;
    const int kstop = min(K, kbeg + k_chunk);
    const bool split = gridDim.z > 1;
    // Global -> register -> LDS staging, one 64x16 tile of each operand per K-step, software-pipelined: the loads of step
    // t+1 are in flight while step t's MFMAs run.  Operands whose contraction index is the memory-fastest one (forward:
    // x [M][K] and W [N][K]) are read with one 16-B load per thread (thread -> row tid/4, k (tid%4)*4..+3) when rows are
    // 16-B aligned; everything else keeps the strided scalar loads (thread-fastest index = memory-fastest index).
    const bool a_vec = a_kfast && ((sam & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = b_kfast && ((sbn & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const int vr = tid >> 2, vk = (tid & 3) * 4;
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        if (a_vec) {
            const bool rok = m0 + vr < M;
            const float* ap = A + (int64_t)(m0 + vr) * sam + (k0 + vk);
            if (rok && k0 + vk + 3 < kstop) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ap);
                ra[0] = v[0]; ra[1] = v[1]; ra[2] = v[2]; ra[3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[e] = (rok && k0 + vk + e < kstop) ? ap[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = e * 256 + tid;
                int m, k;
                if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
                ra[e] = (m0 + m < M && k0 + k < kstop) ? A[(int64_t)(m0 + m) * sam + (int64_t)(k0 + k) * sak] : 0.f;
            }
        }
        if (b_vec) {
            const bool rok = n0 + vr < N;
            const float* bp = B + (int64_t)(n0 + vr) * sbn + (k0 + vk);
            if (rok && k0 + vk + 3 < kstop) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bp);
                rb[0] = v[0]; rb[1] = v[1]; rb[2] = v[2]; rb[3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = (rok && k0 + vk + e < kstop) ? bp[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = e * 256 + tid;
                int n, kb;
                if (b_kfast) { kb = idx & 15; n = idx >> 4; } else { n = idx & 63; kb = idx >> 6; }
                rb[e] = (n0 + n < N && k0 + kb < kstop) ? B[(int64_t)(k0 + kb) * sbk + (int64_t)(n0 + n) * sbn] : 0.f;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = e * 256 + tid;
            if (a_vec) {
                As[(vk + e) * PITCH + vr] = ra[e];
            } else {
                int m, k;
                if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
                As[k * PITCH + m] = ra[e];
            }
            if (b_vec) {
                Bs[(vk + e) * PITCH + vr] = rb[e];
            } else {
                int n, kb;
                if (b_kfast) { kb = idx & 15; n = idx >> 4; } else { n = idx & 63; kb = idx >> 6; }
                Bs[kb * PITCH + n] = rb[e];
            }
        }
    };
    if (kbeg < kstop) fetch(kbeg);
    for (int k0 = kbeg; k0 < kstop; k0 += FBK) {
        commit();
        __syncthreads();
        if (k0 + FBK < kstop) fetch(k0 + FBK);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[(ks * 4 + g) * PITCH + wm * 32 + t * 16 + l15];
                b[t] = Bs[(ks * 4 + g) * PITCH + wn * 32 + t * 16 + l15];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C layout: col = lane&15, row = 4*(lane>>4)+reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + l15;
            if (col >= N) continue;
            const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + 4 * g + r;
                if (row >= M) continue;
                if (split) {
                    ws[((int64_t)blockIdx.z * M + row) * N + col] = acc[i][j][r];
                    continue;
                }
                float v = acc[i][j][r] + bv;
                float* cp = C + (int64_t)row * ldc + col;
                {
                    if (accumulate) v += *cp;
                    *cp = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void gemm_f32_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ C,
                                                              int64_t ldc, int M, int N, const float* __restrict__ bias,
                                                              int accumulate) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += ws[(int64_t)z * total + i];
        if (bias) v += bias[col];
        float* cp = C + (int64_t)row * ldc + col;
        if (accumulate) v += *cp;
        *cp = v;
    }
}

}  // namespace

KAI0_API int64_t kai0_gemm_f32_workspace_bytes(int M, int N, int split_k) {
    return split_k > 1 ? (int64_t)split_k * M * N * 4 : 0;
}

KAI0_API int kai0_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                           float* C, int64_t ldc, int M, int N, int K, const float* bias, int accumulate, int split_k,
                           void* workspace, int64_t workspace_bytes, kai0_stream_t stream) {
    KAI0_REQUIRE(A && B && C, "kai0_gemm_f32: null operand");
    KAI0_REQUIRE(M > 0 && N > 0 && K > 0, "kai0_gemm_f32: empty problem M=%d N=%d K=%d", M, N, K);
    if (split_k < 1) split_k = 1;
    int k_chunk = ((K + split_k - 1) / split_k + FBK - 1) / FBK * FBK;
    split_k = (K + k_chunk - 1) / k_chunk;
    dim3 grid((N + FBN - 1) / FBN, (M + FBM - 1) / FBM, split_k), block(256, 1, 1);
    KAI0_REQUIRE(grid.y <= 65535, "kai0_gemm_f32: M=%d too large for grid.y", M);
    KAI0_REQUIRE(split_k == 1 || (workspace != nullptr && workspace_bytes >= (int64_t)split_k * M * N * 4),
                 "kai0_gemm_f32: split_k=%d needs a workspace of split_k*M*N*4 bytes", split_k);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, (hipStream_t)stream, A, sam, sak, B, sbk, sbn, C, ldc, M,
                       N, K, bias, accumulate, k_chunk, (float*)workspace);
    if (split_k > 1) {
        int rc = kai0_check_launch("kai0_gemm_f32");
        if (rc) return rc;
        int rb = (int)(((int64_t)M * N + 255) / 256);
        if (rb > 1024) rb = 1024;
        hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3(rb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, split_k,
                           C, ldc, M, N, bias, accumulate);
    }
    return kai0_check_launch("kai0_gemm_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// Few-row f32 Linear (M <= 16): out[m][n] = sum_k x[m][k] W[n][k] + b[n] — a weight-streaming GEMV batch.
// One wave per output column n: the 4 KiB weight row is read once with 16-B loads, the M dot products are finished
// with a wave reduction.  Used for the time-MLP / adaRMS `dense` modulations of the denoise schedule (M = steps*B).
namespace {
constexpr int GEMV_MAXM = 16;
__global__ __launch_bounds__(256) void gemv_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int64_t ldo, int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[GEMV_MAXM];
#pragma unroll
    for (int m = 0; m < GEMV_MAXM; ++m) acc[m] = 0.f;
    const float* wr = W + (int64_t)n * K;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
        for (int m = 0; m < GEMV_MAXM; ++m) {
            if (m < M) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (int64_t)m * K + k);
                acc[m] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
            }
        }
    }
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < GEMV_MAXM; ++m) {
        if (m < M) {
            const float t = wave_sum(acc[m]);
            if (lane == 0) out[(int64_t)m * ldo + n] = t + bv;
        }
    }
}
}  // namespace

KAI0_API int kai0_linear_rows_f32(const float* x, const float* W, const float* bias, float* out, int64_t ldo, int M, int N,
                                  int K, kai0_stream_t stream) {
    KAI0_REQUIRE(M >= 1 && M <= GEMV_MAXM && (K % 4) == 0, "kai0_linear_rows_f32: needs 1 <= M <= 16 and K %% 4 == 0 (M=%d K=%d)",
                 M, K);
    KAI0_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)W % 16) == 0, "kai0_linear_rows_f32: unaligned operands");
    hipLaunchKernelGGL(gemv_rows_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, bias, out, ldo, M, N, K);
    return kai0_check_launch("kai0_linear_rows_f32");
}
