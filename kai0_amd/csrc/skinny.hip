// skinny.hip — few-row (M <= 64 per tile) weight-streaming GEMM for the denoise loop of action-chunk inference:
//   C[M, N] = A[M, K] @ W[N, K]^T with the surrounding element-wise work fused into the epilogue.
//
// At B = 1 the action expert multiplies 50 token rows by 0.6 GB of weights ten times per chunk: the job is to stream
// W once at HBM speed and to launch as few kernels as possible, not to feed MFMA.  gfx950 design:
//   * one block = 64 rows x 32 output-weight rows (two 16-row groups `pair_stride` apart, so the RoPE partner column
//     d + 128 of a head, or the `up` column of a GeGLU pair, lands in the SAME lane and register as its mate);
//   * grid = tiles x split_k (x row tiles): the contraction is cut over blocks until the chip is full; inside the
//     block each of the 4 waves owns a quarter of the block's K range (1 or 2 chunks of 128);
//   * operands never touch LDS: every lane reads its MFMA fragment straight from global as 4 x 16 B = 64 contiguous
//     bytes per row (the contraction index inside a 128-chunk is permuted identically for A and W, which a dot
//     product does not care about), all loads of a chunk in flight before the first MFMA;
//   * wave partials meet in LDS; with split_k > 1 the block writes its raw f32 partial product [split][M][N] and the
//     consumer kernel (kai0_adarms_combine: gated residual + adaRMS, needed anyway) adds the splits in a fixed order.
//     (An in-kernel "last block reduces" scheme was measured at 24-41 us per launch: the agent-scope fences it needs
//     write back / invalidate the whole L2.  The kernel boundary gives the same visibility for free.)
//   * epilogues (rounding points identical to the separate kernels they replace, see kai0hip.h):
//       plain  : bf16 -> (* gate) -> (+ residual)
//       rope   : up to 3 column segments (q | k | v) with their own destination / leading dimension; rotated with
//                precomputed bf16-rounded cos/sin tables
//       geglu  : h = bf16( bf16(gelu_tanh(g)) * u ), W = [gate ; up]
#include "common.h"
#include "../../include/kai0hip.h"
#include <stdlib.h>

namespace {

struct SkRowMap {
    int32_t rpb;
    int64_t bs, off;
    __device__ __forceinline__ int64_t operator()(int r) const {
        if (rpb == 0) return r;
        const int q = r / rpb;
        return (int64_t)q * bs + (r - q * rpb) + off;
    }
};

struct SkSeg {
    bf16_t* dst;
    int64_t ld;
    int32_t n_begin, n_end, rope;
};

struct SkinnyArgs {
    const bf16_t* A;
    const bf16_t* W;
    int64_t lda, ldw;
    int M, N, K;
    int pair_stride, mode, split_k, k_blk;
    SkRowMap amap, cmap;
    SkSeg seg[3];
    int nseg;
    const bf16_t* gate;
    int gate_rpb;
    int64_t gate_ld;
    const bf16_t* residual;
    int64_t ldr;
    const float* rope_cos;
    const float* rope_sin;
    int rope_half;
    float* ws;
    const float* mod;  // adaRMS prologue (skinny2 only): scale = mod[b*mod_ld + k], shift = mod[b*mod_ld + K + k]
    int mod_rpb;
    int64_t mod_ld;
    float eps;
    int w_packed;  // skinny2: W in fragment-major 1-KiB blocks (kai0hip.h)
    // folded adaRMS (skinny2, kai0hip.h `rowsq_in`): the norm's scale is folded into W and its shift into cvec by the caller; the kernel
    // multiplies raw x and applies out = acc * rstd[row] + cvec[n], rstd from the producer's per-column-tile partial sums of squares
    const float* rowsq_in;   // [rowsq_parts][rowsq_ld] f32
    const float* cvec;       // [N] f32
    int rowsq_parts;
    int64_t rowsq_ld;
    float* rowsq_out;        // mode 0 (in-block): this launch's own partials, [N / 16][rowsq_out_ld]
    int64_t rowsq_out_ld;
};

constexpr int TM = 64, TN = 32;
constexpr int RED_LD = TN + 1;  // f32 row stride of a wave's partial tile in LDS (odd: conflict-free column writes)

// The fused epilogue on this thread's 4 + 4 outputs: v0 = columns n0 .. n0+3, v1 = columns n1 .. n1+3 (n1 = n0 + pair_stride; only
// v0 when `pair` is false) of output row `mrow` (f32 sums over the whole contraction).  Rounding points: see kai0hip.h.
__device__ __forceinline__ void sk_epilogue(const SkinnyArgs& p, float (&v0)[4], float (&v1)[4], int mrow, int n0, int n1, bool pair,
                                            int col_tile = 0) {
    const int64_t orow = p.cmap(mrow);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v0[e] = rbf(v0[e]);
        v1[e] = rbf(v1[e]);
    }
    if (p.mode == 2) {  // GeGLU: v0 = gate pre-activation, v1 = up
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(rbf(gelu_tanh_f(v0[e])) * v1[e]);
        *reinterpret_cast<bf16x4*>(p.seg[0].dst + orow * p.seg[0].ld + n0) = o;
        return;
    }
    if (p.mode == 1) {  // column segments, optional rotation
        int si = 0;
        if (p.nseg > 1 && n0 >= p.seg[1].n_begin) si = 1;
        if (p.nseg > 2 && n0 >= p.seg[2].n_begin) si = 2;
        const SkSeg sg = p.seg[si];
        const int c0 = n0 - sg.n_begin;
        bf16x4 o1, o2;
        if (sg.rope == 1) {
            const int d = c0 % (2 * p.rope_half);  // index inside the head, < rope_half by construction
            const float* ct = p.rope_cos + (int64_t)mrow * p.rope_half + d;
            const float* st = p.rope_sin + (int64_t)mrow * p.rope_half + d;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = ct[e], s = st[e];
                o1[e] = f2bf(rbf(v0[e] * c) + rbf(-v1[e] * s));
                o2[e] = f2bf(rbf(v1[e] * c) + rbf(v0[e] * s));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o1[e] = f2bf(v0[e]);
                o2[e] = f2bf(v1[e]);
            }
        }
        if (sg.rope == 2) {
            // transposed destination [batch][cols][ld] (value cache of kai0_attn_decode): element (row, col) -> dst[col][row]
            const int bq = p.cmap.rpb ? mrow / p.cmap.rpb : 0;
            const int64_t srow = p.cmap.rpb ? (int64_t)(mrow - bq * p.cmap.rpb) + p.cmap.off : (int64_t)mrow;
            bf16_t* dp = sg.dst + (int64_t)bq * (sg.n_end - sg.n_begin) * sg.ld + (int64_t)c0 * sg.ld + srow;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dp[(int64_t)e * sg.ld] = o1[e];
                dp[(int64_t)(e + p.pair_stride) * sg.ld] = o2[e];
            }
            return;
        }
        bf16_t* dp = sg.dst + orow * sg.ld + c0;
        *reinterpret_cast<bf16x4*>(dp) = o1;
        *reinterpret_cast<bf16x4*>(dp + p.pair_stride) = o2;
        return;
    }
    // gate and residual are requested together, before either is used: in two dependent blocks (load gate, multiply, load residual,
    // add) the epilogue was two memory round trips long
    const bool has_g = p.gate != nullptr, has_r = p.residual != nullptr;
    bf16x4 g0 = {}, g1 = {}, r0 = {}, r1 = {};
    if (has_g) {
        const bf16_t* gp = p.gate + (int64_t)(mrow / p.gate_rpb) * p.gate_ld;
        g0 = *reinterpret_cast<const bf16x4*>(gp + n0);
        if (pair) g1 = *reinterpret_cast<const bf16x4*>(gp + n1);
    }
    if (has_r) {
        const bf16_t* rp = p.residual + orow * p.ldr;
        r0 = *reinterpret_cast<const bf16x4*>(rp + n0);
        if (pair) r1 = *reinterpret_cast<const bf16x4*>(rp + n1);
    }
    if (has_g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = rbf(v0[e] * bf2f(g0[e]));
            if (pair) v1[e] = rbf(v1[e] * bf2f(g1[e]));
        }
    }
    if (has_r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = rbf(v0[e] + bf2f(r0[e]));
            if (pair) v1[e] = rbf(v1[e] + bf2f(r1[e]));
        }
    }
    bf16x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o1[e] = f2bf(v0[e]);
        o2[e] = f2bf(v1[e]);
    }
    bf16_t* dp = p.seg[0].dst + orow * p.seg[0].ld;
    *reinterpret_cast<bf16x4*>(dp + n0) = o1;
    if (pair) *reinterpret_cast<bf16x4*>(dp + n1) = o2;
    if (p.rowsq_out != nullptr) {
        // sum of squares of the bf16 values just stored over this block's 16 columns of the row (the four threads tid & 3 of a row
        // are neighbours in the wave): the consumer's adaRMS statistic, one partial per column tile, summed there in tile order
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += bf2f(o1[e]) * bf2f(o1[e]);
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        if ((threadIdx.x & 3) == 0) p.rowsq_out[(int64_t)col_tile * p.rowsq_out_ld + mrow] = ss;
    }
}

// NW waves share the block's K range (k_blk / NW each, NC chunks of 128): 4 x 2 chunks or 8 x 1 chunk for k_blk = 1024
// (twice the waves = twice the loads in flight per CU for the launches that cannot split K over blocks), 4 x 1 for 512.
template <int NC, int NW>
__global__ __launch_bounds__(NW * 64, 1) void skinny_kernel(const SkinnyArgs p) {
    __shared__ float red[NW][TM][RED_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, ks = blockIdx.y, mt_blk = blockIdx.z;
    const int per = p.pair_stride >> 4;
    const int n_sub0 = (tile / per) * (2 * p.pair_stride) + (tile % per) * 16;
    const int n_sub1 = n_sub0 + p.pair_stride;
    const int m0 = mt_blk * TM;
    const int kw0 = ks * p.k_blk + wave * (p.k_blk / NW);

    // ---- all fragment loads of this wave's K slice, then the MFMAs --------------------------------
    const bf16_t* w0 = p.W + (int64_t)(n_sub0 + i) * p.ldw + kw0 + 32 * g;
    const bf16_t* w1 = p.W + (int64_t)(n_sub1 + i) * p.ldw + kw0 + 32 * g;
    const bf16_t* arow[4];
    bool aok[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = m0 + mt * 16 + i;
        aok[mt] = r < p.M;
        arow[mt] = p.A + p.amap(aok[mt] ? r : 0) * p.lda + kw0 + 32 * g;
    }
    bf16x8 wf[NC][2][4], af[NC][4][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wf[c][0][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w0 + c * 128 + 8 * j));
            wf[c][1][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + c * 128 + 8 * j));
        }
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                af[c][mt][j] = aok[mt] ? *reinterpret_cast<const bf16x8*>(arow[mt] + c * 128 + 8 * j) : zero8;

    f32x4 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt][0] = acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c][mt][j], wf[c][0][j], acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c][mt][j], wf[c][1][j], acc[mt][1], 0, 0, 0);
            }

    // ---- wave partials -> LDS -> per-thread 4+4 outputs --------------------------------------------
    // C layout of a 16x16 MFMA tile: col = lane & 15 (W row), row = 4*(lane >> 4) + reg (A row)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][mt * 16 + 4 * g + r][s * 16 + i] = acc[mt][s][r];
    __syncthreads();
    if (tid >= 256) return;  // 64 rows x 4 column quads finish the tile
    const int row = tid >> 2, q = tid & 3;
    float v0[4], v1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            s0 += red[w][row][4 * q + e];
            s1 += red[w][row][16 + 4 * q + e];
        }
        v0[e] = s0;
        v1[e] = s1;
    }

    // ---- split-K: raw f32 partial products [split][M][N]; the consumer (kai0_adarms_combine) adds them in order ----
    if (p.split_k > 1) {
        const int mrow_p = m0 + row;
        if (mrow_p < p.M) {
            float* dstp = p.ws + ((int64_t)ks * p.M + mrow_p) * p.N;
            *reinterpret_cast<f32x4*>(dstp + n_sub0 + 4 * q) = f32x4{v0[0], v0[1], v0[2], v0[3]};
            *reinterpret_cast<f32x4*>(dstp + n_sub1 + 4 * q) = f32x4{v1[0], v1[1], v1[2], v1[3]};
        }
        return;
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    const int mrow = m0 + row;
    if (mrow >= p.M) return;
    sk_epilogue(p, v0, v1, mrow, n_sub0 + 4 * q, n_sub1 + 4 * q, true);
}

// skinny2: the whole contraction inside ONE block (no partial products, no combine launch): NW = K / 256 waves (4, 8 or 16),
// each with 256 of K (two chunks of 128), every load of the block in flight at once, wave partials summed through LDS, the full
// epilogue in the same launch.  N / 32 (PAIR: two 16-column groups `pair_stride` apart, modes 0-2) or N / 16 blocks (!PAIR:
// mode 0) — a 1024-column o_proj / down_proj is 64 blocks of 64-128 KB of weights each: the launch is one memory latency long,
// which is what matters for a 50-row GEMM; the former split-K version needed its partials finished by a separate launch.
//   ADA: the A operand is adaRMS-normalised on the fly (K == D, the block's waves hold complete rows): y = bf16((x * rstd) *
//   (1 + scale) + shift), rstd over the row, scale / shift from `mod` — the norm that kai0_adarms_combine applied in a launch
//   of its own (modeling_gemma.py:49-104).
//   MTL = 16-row tiles per block: 4 (all 64 rows; the weight slice is read once) for the wide q|k|v and gate|up launches, 1 for the
//   1024-column o_proj / down_proj, whose 64 column tiles alone would leave three quarters of the chip idle: their four row
//   tiles run as four blocks with the same blockIdx.x, i.e. on the same XCD (block -> XCD is linear id % 8 and gridDim.x % 8 == 0),
//   so the weight slice crosses the fabric once and is shared through that XCD's L2.
//   NC = 128-wide chunks of K per wave (K = NW * NC * 128): 2 normally, 1 for the adaRMS variant (half the fragment registers per
//   wave, twice the waves).
//   ALDS: the A rows of the block go through LDS — read from global as contiguous 1-KiB runs (64 lanes x 16 B along a row), written
//   to LDS with 16 B of row padding, MFMA fragments read back with ds_read_b128.  Every block of a launch reads the same few
//   hundred KB of activations out of L2, and the direct fragment pattern (16 rows x 64 B per wave-instruction) gets 31-34 GB/s
//   per CU there against 80-90 GB/s for contiguous runs (round-2 probe frag_load.hip, git history / profiles/HISTORY.md: 128 KB per CU 4.8 vs 2.8 us, 256 KB 8.2 vs
//   3.3 us): in the 1024-column projections the A operand, not the weight stream, was the longer load.
// diagnostics (a workspace passed with split_k == -1 = phase trace, int64 [blocks][8]): wave 0 stamps the shader clock at the
// phase boundaries of its block
// (compiled in only with -DKAI0_SK2_TRACE: the eight extra branches split the kernel's basic blocks and cost the denoise loop 5 %)
#ifdef KAI0_SK2_TRACE
#define SK2_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (trace != nullptr && tid == 0) trace[(int64_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + (i)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define SK2_STAMP(i) do { (void)trace; } while (0)
#endif

template <int NW, int MTL, bool PAIR, bool ADA, int NC = 2, bool WNT = true, bool ALDS = false, bool FOLD = false>
__global__ __launch_bounds__(NW * 64, 1) void skinny2_kernel(const SkinnyArgs p) {
    static_assert(!(FOLD && ADA), "the folded form has no prologue");
    constexpr int TMB = 16 * MTL;
    constexpr int TN2 = PAIR ? 32 : 16;
    constexpr int LD2 = TN2 + 1;
    constexpr int KB = NW * NC * 128;               // the contraction width this instantiation is built for
    constexpr int A_ROWB = KB * 2 + 16;             // ALDS: bytes per A row in LDS (16 B of padding: conflict-free b128 reads)
    constexpr int A_BYTES = ALDS ? TMB * A_ROWB : 0;
    constexpr int RED_BYTES = (int)sizeof(float) * NW * TMB * LD2;
    extern __shared__ __attribute__((aligned(16))) char sk2_smem[];
    // ALDS: the partial-sum buffer aliases the A tile (a barrier separates the last fragment read from the first partial write)
    float (*red)[TMB][LD2] = reinterpret_cast<float (*)[TMB][LD2]>(sk2_smem);
    float (*ssq)[TMB] = reinterpret_cast<float (*)[TMB]>(sk2_smem + (A_BYTES > RED_BYTES ? A_BYTES : RED_BYTES));
    // ALDS + ADA: the entry's scale | shift vectors (2 K f32) are requested at kernel entry, one 16-B piece per thread, and wait in
    // LDS behind the row statistics (read from global after the statistics they were a second, dependent memory round trip)
    float* mod_lds = reinterpret_cast<float*>(sk2_smem + (A_BYTES > RED_BYTES ? A_BYTES : RED_BYTES) + sizeof(float) * NW * TMB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, mt_blk = blockIdx.z;
    // Every kernel argument the block will need is pulled into SGPRs HERE, in one batch of scalar loads.  Left to itself hipcc loads
    // each field of the by-value struct next to its first use, behind the branch that guards it: eight serial s_load / s_waitcnt
    // round trips (~2700 clocks, 1.3 us) stood in front of the first weight load of every launch (tools/probes/sk2_phases.py).
    asm volatile("" ::"s"(p.A), "s"(p.W), "s"(p.lda), "s"(p.ldw), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.pair_stride), "s"(p.mode),
                 "s"(p.amap.rpb), "s"(p.amap.bs), "s"(p.amap.off), "s"(p.mod), "s"(p.mod_rpb), "s"(p.mod_ld), "s"(p.w_packed),
                 "s"(p.ws));
    long long* trace = reinterpret_cast<long long*>(p.ws);
    SK2_STAMP(0);
    int n_sub0, n_sub1;
    if constexpr (PAIR) {
        const int per = p.pair_stride >> 4;
        n_sub0 = (tile / per) * (2 * p.pair_stride) + (tile % per) * 16;
        n_sub1 = n_sub0 + p.pair_stride;
    } else {
        n_sub0 = tile * 16;
        n_sub1 = n_sub0;
    }
    // ADA: row tiles never straddle two batch entries (one scale / shift vector per tile): tile z covers rows
    // b * rpb + [t * TMB, (t + 1) * TMB) of entry b = z / tiles_per_entry
    int m0 = mt_blk * TMB, m_end = p.M;
    if constexpr (ADA) {
        const int tpe = (p.mod_rpb + TMB - 1) / TMB, b = mt_blk / tpe;
        m0 = b * p.mod_rpb + (mt_blk - b * tpe) * TMB;
        m_end = min(p.M, (b + 1) * p.mod_rpb);
    }
    const int kw0 = wave * (NC * 128);

    // contraction index of (lane group g, load j, element e) inside a 128-chunk: 32 j + 8 g + e — the four lanes of a row fetch one
    // contiguous 64-byte sector per load instruction (the same permutation for A, W and the modulation vectors)
    // W: row-major [N][ldw] (lane (i, g): row i of the 16-row group, 16 B at k = 32 j + 8 g of each 128-chunk), or fragment-major
    // (kai0hip.h w_packed: one contiguous KiB per 16 rows x 32 k — the lane's 16 B at lane * 16 of block (tile, k / 32))
    const int64_t wj = p.w_packed ? 512 : 32, wc = p.w_packed ? 2048 : 128;  // element strides of load j / chunk c
    const bf16_t* w0 = p.w_packed ? p.W + ((int64_t)(n_sub0 >> 4) * (p.K >> 5) + (kw0 >> 5)) * 512 + lane * 8
                                  : p.W + (int64_t)(n_sub0 + i) * p.ldw + kw0 + 8 * g;
    const bf16_t* w1 = p.w_packed ? p.W + ((int64_t)(n_sub1 >> 4) * (p.K >> 5) + (kw0 >> 5)) * 512 + lane * 8
                                  : p.W + (int64_t)(n_sub1 + i) * p.ldw + kw0 + 8 * g;
    const bf16_t* arow[MTL];
    bool aok[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) {
        const int r = m0 + mt * 16 + i;
        aok[mt] = r < m_end;
        arow[mt] = p.A + p.amap(aok[mt] ? r : 0) * p.lda + kw0 + 8 * g;
    }
    bf16x8 wf[NC][PAIR ? 2 : 1][4], af[NC][MTL][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // WNT: streamed once by this block only -> non-temporal; shared by the row-tile blocks of an XCD -> keep it in L2
            if constexpr (WNT) {
                wf[c][0][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w0 + c * wc + j * wj));
                if constexpr (PAIR) wf[c][1][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + c * wc + j * wj));
            } else {
                wf[c][0][j] = *reinterpret_cast<const bf16x8*>(w0 + c * wc + j * wj);
                if constexpr (PAIR) wf[c][1][j] = *reinterpret_cast<const bf16x8*>(w1 + c * wc + j * wj);
            }
        }
    // FOLD: the producer's partial sums of squares of this block's rows, requested here (behind the weight stream), reduced after
    // the MFMAs: P = threads per row, each adds its share of the `rowsq_parts` partials in tile order, the P shares meet by shuffles
    constexpr int FP = FOLD ? (NW * 64) / TMB : 1;      // threads per row (8 .. 32)
    constexpr int FQ = FOLD ? (64 + FP - 1) / FP : 1;   // partials per thread (<= 64 column tiles)
    float fsq[FQ];
    if constexpr (FOLD) {
        const int frow = min(m0 + tid / FP, m_end - 1), fj = tid % FP;
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int part = fj * FQ + q;
            const float v = p.rowsq_in[(int64_t)min(part, p.rowsq_parts - 1) * p.rowsq_ld + frow];  // clamped, not guarded
            fsq[q] = part < p.rowsq_parts ? v : 0.f;
        }
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 mod_piece[(ALDS && ADA) ? (2 * KB / 4 + NW * 64 - 1) / (NW * 64) : 1];
    if constexpr (ALDS && ADA) {
        const float* mrow = p.mod + (int64_t)(m0 / p.mod_rpb) * p.mod_ld;
#pragma unroll
        for (int q = 0; q < (2 * KB / 4 + NW * 64 - 1) / (NW * 64); ++q) {
            // 16-B piece of [scale (K) | shift (K)]; the index is clamped instead of guarded: a conditional load is compiled as a
            // branch with its own vmcnt(0), which held back every load issued after it by one memory round trip
            const int idx = min(q * (NW * 64) + tid, 2 * KB / 4 - 1);
            mod_piece[q] = *reinterpret_cast<const f32x4*>(mrow + idx * 4);
        }
    }
    if constexpr (ALDS) {
        // the block's TMB x K tile of A: 16-B chunks, consecutive lanes on consecutive chunks of a row.  Loaded through a buffer
        // descriptor with 32-bit offsets: a row past the tile's end gets an out-of-range offset (the hardware returns zeros), so
        // all NCH loads are unconditional and issued back to back.  (With `row < m_end ? load : 0` and the 64-bit row map hipcc
        // emitted an exec-masked branch, a signed division and a 64-bit multiply PER CHUNK — ~1300 instructions, ~3 us, in front of
        // the gate | up launch's last load.)  The row map is evaluated once per thread and stepped: a thread's chunks are RPQ rows apart.
        constexpr int CPR = KB / 8;                         // chunks per row
        constexpr int NCH = TMB * CPR / (NW * 64);          // chunks per thread
        constexpr int RPQ = NW * 64 / CPR;                  // rows between a thread's consecutive chunks
        static_assert((TMB * CPR) % (NW * 64) == 0 && (NW * 64) % CPR == 0, "A tile must divide over the block's threads");
        const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)0x80000000u, 0x00020000);
        const int r0 = tid / CPR, c8 = tid - r0 * CPR;
        int quot = 0, rem = m0 + r0;                        // row m0 + r0 = quot * rpb + rem
        if (p.amap.rpb != 0) {
            quot = rem / p.amap.rpb;
            rem -= quot * p.amap.rpb;
        }
        const uint32_t lda2 = (uint32_t)p.lda * 2, bs2 = (uint32_t)p.amap.bs * lda2;
        const uint32_t base = (uint32_t)p.amap.off * lda2 + (uint32_t)c8 * 16;
        bf16x8 ch[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int row = m0 + r0 + q * RPQ;
            uint32_t off = (uint32_t)quot * bs2 + (uint32_t)rem * lda2 + base;
            if (row >= m_end) off = 0x80000000u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off, 0, 0);
            ch[q] = __builtin_bit_cast(bf16x8, v);
            rem += RPQ;
            if (p.amap.rpb != 0) {
                if (p.amap.rpb >= RPQ) {  // one step at most (select, no branch)
                    const bool wrap = rem >= p.amap.rpb;
                    rem -= wrap ? p.amap.rpb : 0;
                    quot += wrap ? 1 : 0;
                } else {  // batch entries shorter than a thread's row step (tests): plain division
                    const int nr = m0 + r0 + (q + 1) * RPQ;
                    quot = nr / p.amap.rpb;
                    rem = nr - quot * p.amap.rpb;
                }
            }
        }
        SK2_STAMP(1);  // every load of the block issued
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int cidx = q * (NW * 64) + tid;
            const int r = cidx / CPR;
            *reinterpret_cast<bf16x8*>(sk2_smem + r * A_ROWB + (cidx - r * CPR) * 16) = ch[q];
        }
        if constexpr (ADA) {
#pragma unroll
            for (int q = 0; q < (2 * KB / 4 + NW * 64 - 1) / (NW * 64); ++q) {
                const int idx = q * (NW * 64) + tid;
                if (idx < 2 * KB / 4) *reinterpret_cast<f32x4*>(mod_lds + idx * 4) = mod_piece[q];
            }
        }
        __syncthreads();
        SK2_STAMP(2);  // A tile in LDS
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    af[c][mt][j] = *reinterpret_cast<const bf16x8*>(sk2_smem + (mt * 16 + i) * A_ROWB + (kw0 + 8 * g + c * 128 + 32 * j) * 2);
    } else {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                af[c][mt][j] = aok[mt] ? *reinterpret_cast<const bf16x8*>(arow[mt] + c * 128 + 32 * j) : zero8;
    }

    if constexpr (ADA) {
        // row statistics: this wave's 256 of the row's K = D elements, then the NW waves' partials through LDS (fixed order)
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = bf2f(af[c][mt][j][e]);
                        ss += x * x;
                    }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (g == 0) ssq[wave][mt * 16 + i] = ss;
        }
        __syncthreads();
        SK2_STAMP(3);  // row statistics exchanged
        float rstd[MTL];
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) {
            float ss = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) ss += ssq[w][mt * 16 + i];
            rstd[mt] = rsqrtf(ss / (float)p.K + p.eps);
        }
        const float* mrow = (ALDS ? mod_lds : p.mod + (int64_t)(m0 / p.mod_rpb) * p.mod_ld) + kw0 + 8 * g;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* mp = mrow + c * 128 + 32 * j;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(mp), a1 = *reinterpret_cast<const f32x4*>(mp + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(mp + p.K), b1 = *reinterpret_cast<const f32x4*>(mp + p.K + 4);
#pragma unroll
                for (int mt = 0; mt < MTL; ++mt) {
                    bf16x8 y;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        y[e] = f2bf((bf2f(af[c][mt][j][e]) * rstd[mt]) * (1.0f + (e < 4 ? a0[e] : a1[e - 4])) + (e < 4 ? b0[e] : b1[e - 4]));
                    af[c][mt][j] = aok[mt] ? y : zero8;
                }
            }
    }

    SK2_STAMP(4);  // A fragments ready (normalised)
    f32x4 acc[MTL][PAIR ? 2 : 1];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
        for (int s2 = 0; s2 < (PAIR ? 2 : 1); ++s2) acc[mt][s2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int s2 = 0; s2 < (PAIR ? 2 : 1); ++s2)
                    acc[mt][s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c][mt][j], wf[c][s2][j], acc[mt][s2], 0, 0, 0);
    if constexpr (ALDS) __syncthreads();  // `red` aliases the A tile: every wave has its fragments in registers by now
    SK2_STAMP(5);  // weights landed, MFMAs issued
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
        for (int s2 = 0; s2 < (PAIR ? 2 : 1); ++s2)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][mt * 16 + 4 * g + r][s2 * 16 + i] = acc[mt][s2][r];
    if constexpr (FOLD) {
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < FQ; ++q) ss += fsq[q];
#pragma unroll
        for (int off = 1; off < FP; off <<= 1) ss += __shfl_xor(ss, off, 64);
        if (tid % FP == 0) ssq[0][tid / FP] = rsqrtf(ss / (float)p.K + p.eps);
    }
    __syncthreads();
    SK2_STAMP(6);  // wave partials in LDS
    if (tid >= TMB * 4) return;  // TMB rows x 4 column quads finish the tile
    const int row = tid >> 2, q = tid & 3;
    float v0[4], v1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            s0 += red[w][row][4 * q + e];
            if constexpr (PAIR) s1 += red[w][row][16 + 4 * q + e];
        }
        v0[e] = s0;
        v1[e] = s1;
    }
    if constexpr (FOLD) {  // out = (x W'^T) * rstd + c: the norm's per-row factor and the folded shift term
        const float rstd = ssq[0][row];
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.cvec + n_sub0 + 4 * q);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.cvec + n_sub1 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = v0[e] * rstd + c0[e];
            v1[e] = v1[e] * rstd + c1[e];
        }
    }
    const int mrow = m0 + row;
    if (mrow < m_end) sk_epilogue(p, v0, v1, mrow, n_sub0 + 4 * q, n_sub1 + 4 * q, PAIR, tile);
    SK2_STAMP(7);  // epilogue stores issued
}

template <bool BF>
__global__ __launch_bounds__(256) void rope_table_kernel(const int32_t* __restrict__ pos, const float* __restrict__ inv_freq,
                                                         void* __restrict__ cos_out, void* __restrict__ sin_out,
                                                         int64_t n, int half) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t r = t / half;
    const int d = (int)(t - r * half);
    const float ang = inv_freq[d] * (float)pos[r];
    if constexpr (BF) {  // the same bf16-rounded values, stored as bf16 (what the RoPE epilogue of kai0_gemm_bf16 reads)
        reinterpret_cast<bf16_t*>(cos_out)[t] = f2bf(cosf(ang));
        reinterpret_cast<bf16_t*>(sin_out)[t] = f2bf(sinf(ang));
    } else {
        reinterpret_cast<float*>(cos_out)[t] = rbf(cosf(ang));
        reinterpret_cast<float*>(sin_out)[t] = rbf(sinf(ang));
    }
}

}  // namespace

KAI0_API int kai0_skinny_desc_size(void) { return (int)sizeof(kai0_skinny_desc); }

KAI0_API int64_t kai0_skinny_workspace_bytes(int M, int N, int split_k) {
    return split_k > 1 ? (int64_t)split_k * M * N * 4 : 0;
}

namespace {

template <int NW, int MTL, bool PAIR, bool ADA, int NC = 2, bool WNT = true, bool ALDS = false, bool FOLD = false>
int launch_skinny2(const SkinnyArgs& a, hipStream_t s) {
    constexpr int TMB = 16 * MTL, LD2 = (PAIR ? 32 : 16) + 1;
    constexpr int RED = (int)sizeof(float) * NW * TMB * LD2, ATILE = ALDS ? TMB * (NW * NC * 128 * 2 + 16) : 0;
    constexpr int LDS = (ATILE > RED ? ATILE : RED) + (int)sizeof(float) * NW * TMB + ((ALDS && ADA) ? 2 * NW * NC * 128 * 4 : 0);
    static_assert(LDS <= 160 * 1024, "skinny2: LDS");
    // ADA: row tiles per batch entry (see the kernel)
    const int mtiles = ADA ? ((a.M + a.mod_rpb - 1) / a.mod_rpb) * ((a.mod_rpb + TMB - 1) / TMB) : (a.M + TMB - 1) / TMB;
    const dim3 grid(a.N / (PAIR ? 32 : 16), 1, mtiles);
    auto kern = skinny2_kernel<NW, MTL, PAIR, ADA, NC, WNT, ALDS, FOLD>;
    if constexpr (LDS > 48 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            KAI0_REQUIRE(e == hipSuccess, "kai0_gemm_skinny_bf16: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), LDS, s, a);
    return kai0_check_launch("kai0_gemm_skinny_bf16 (in-block K)");
}

// split_k == -1: skinny2 (see kai0hip.h)
int skinny_inblock(const kai0_skinny_desc* d, SkinnyArgs& a, hipStream_t s) {
    const int nw = d->K / 256;  // waves of the two-chunk variants
    KAI0_REQUIRE(d->K % 256 == 0 && (nw == 4 || nw == 8 || nw == 16), "kai0_gemm_skinny_bf16: in-block K needs K in {1024, 2048, 4096}, got %d", d->K);
    const bool ada = d->mod != nullptr;
    const bool fold = d->rowsq_in != nullptr;
    if (fold) {
        KAI0_REQUIRE(!ada && d->mode != 0 && nw == 4 && d->cvec && d->rowsq_parts >= 1 && d->rowsq_parts <= 64 && d->rowsq_ld >= d->M &&
                         d->a_rpb == 0 && ((uintptr_t)d->cvec % 16) == 0,
                     "kai0_gemm_skinny_bf16: the folded adaRMS form needs mode 1 / 2, K = 1024, no `mod`, cvec [N] f32 16-byte aligned, "
                     "1 <= rowsq_parts <= 64, rowsq_ld >= M, identity A rows");
        a.rowsq_in = d->rowsq_in;
        a.cvec = d->cvec;
        a.rowsq_parts = d->rowsq_parts;
        a.rowsq_ld = d->rowsq_ld;
        a.eps = d->eps;
        // the shapes of the adaRMS launches they replace: q|k|v as 8 waves x one 16-row tile (320 blocks), gate|up 8 waves x all four
        if (d->mode == 1) return launch_skinny2<8, 1, true, false, 1, false, true, true>(a, s);
        return launch_skinny2<8, 4, true, false, 1, true, true, true>(a, s);
    }
    if (ada) {
        KAI0_REQUIRE(d->mode != 0 && nw == 4 && d->mod_rpb > 0 && d->mod_ld >= 2 * (int64_t)d->K && d->mod_ld % 4 == 0 &&
                         ((uintptr_t)d->mod % 16) == 0 && d->a_rpb == 0,
                     "kai0_gemm_skinny_bf16: the adaRMS prologue needs mode 1 / 2, K = 1024, mod [b][>= 2K] f32 16-byte aligned, identity A rows");
        a.mod = d->mod;
        a.mod_ld = d->mod_ld;
        a.mod_rpb = d->mod_rpb;
        a.eps = d->eps;
    }
    if (d->rowsq_out != nullptr) {
        KAI0_REQUIRE(d->mode == 0 && d->rowsq_out_ld >= d->M && d->N / 16 <= 64, "kai0_gemm_skinny_bf16: rowsq_out needs in-block mode 0, "
                     "rowsq_out_ld >= M, N <= 1024");
        a.rowsq_out = d->rowsq_out;
        a.rowsq_out_ld = d->rowsq_out_ld;
    }
    if (d->mode == 0) {
        KAI0_REQUIRE(d->N % 128 == 0, "kai0_gemm_skinny_bf16: in-block mode 0 needs N %% 128 == 0 (8 column tiles per XCD round), got %d", d->N);
        a.pair_stride = 16;
        // One 16-row tile per block, cached (not non-temporal) weight loads: the four row-tile blocks of a column tile share the
        // slice through their XCD's L2.  Measured in the chunk (us per launch, K = 2048 / 4096): 6.9 / 10.3; with non-temporal loads
        // (each block streams its own copy from the fabric) 8.6 / 13.9; two row tiles per block 8.8 / 14.1; all four in one block
        // (weights once, 1024-thread blocks) 13.4 / 33.
        // (also measured and removed: half the waves with four chunks of K each; A fragments straight from global instead of through
        // LDS — round 2's form, +0.6 ms per chunk)
        if (nw == 4) return launch_skinny2<4, 1, false, false, 2, false, true>(a, s);
        if (nw == 8) return launch_skinny2<8, 1, false, false, 2, false, true>(a, s);
        return launch_skinny2<16, 1, false, false, 2, false, true>(a, s);
    }
    KAI0_REQUIRE(nw == 4, "kai0_gemm_skinny_bf16: in-block modes 1 / 2 are built for K = 1024 (got %d)", d->K);
    // With the A rows staged through LDS (chunk of 10 Euler steps, one box): q|k|v as 8 waves x one 16-row tile per block (320
    // blocks) 8.69 ms of denoise, 8 waves x two tiles 8.60-8.70, 4 waves x two tiles (the choice before the staging) 8.83; gate|up
    // keeps all four row tiles in one block.  The adaRMS-prologue form (`mod`) is the A/B alternative of the folded form above.
    if (ada) {
        // the adaRMS prologue reads A with identity rows (a_rpb == 0), so the coalesced tile load applies as is
        if (d->mode == 1) return launch_skinny2<8, 1, true, true, 1, false, true>(a, s);
        return launch_skinny2<8, 4, true, true, 1, true, true>(a, s);
    }
    return launch_skinny2<4, 4, true, false>(a, s);
}

}  // namespace

KAI0_API int kai0_gemm_skinny_bf16(const kai0_skinny_desc* d, kai0_stream_t stream) {
    KAI0_REQUIRE(d != nullptr && d->A && d->W, "kai0_gemm_skinny_bf16: null operand");
    KAI0_REQUIRE(d->M >= 1 && d->N >= 32 && d->N % 32 == 0, "kai0_gemm_skinny_bf16: M=%d N=%d unsupported", d->M, d->N);
    const bool inblock = d->split_k == -1;
    const int S = d->split_k < 1 ? 1 : d->split_k;
    KAI0_REQUIRE(inblock || (d->K % S == 0 && ((d->K / S) == 512 || (d->K / S) == 1024)),
                 "kai0_gemm_skinny_bf16: K=%d split_k=%d: K/split_k must be 512 or 1024", d->K, S);
    KAI0_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0 && ((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->W % 16) == 0,
                 "kai0_gemm_skinny_bf16: operands must be 16-byte aligned with leading dimensions %% 8 == 0");
    const int ps = (inblock && d->mode == 0) ? 16 : d->pair_stride;
    KAI0_REQUIRE(ps >= 16 && ps % 16 == 0 && d->N % (2 * ps) == 0, "kai0_gemm_skinny_bf16: pair_stride=%d does not tile N=%d",
                 ps, d->N);
    KAI0_REQUIRE(d->mode >= 0 && d->mode <= 2, "kai0_gemm_skinny_bf16: mode=%d", d->mode);
    KAI0_REQUIRE(S > 1 || (d->nseg >= 1 && d->nseg <= 3 && d->seg[0].dst), "kai0_gemm_skinny_bf16: nseg=%d", d->nseg);
    SkinnyArgs a{};
    a.A = (const bf16_t*)d->A;
    a.W = (const bf16_t*)d->W;
    a.lda = d->lda;
    a.ldw = d->ldw;
    a.M = d->M;
    a.N = d->N;
    a.K = d->K;
    a.pair_stride = ps;
    a.mode = d->mode;
    a.split_k = S;
    a.k_blk = d->K / S;
    a.amap = SkRowMap{d->a_rpb, d->a_bs, d->a_off};
    a.cmap = SkRowMap{d->c_rpb, d->c_bs, d->c_off};
    a.nseg = S > 1 ? 0 : d->nseg;
    for (int s = 0; s < a.nseg; ++s) {
        a.seg[s] = SkSeg{(bf16_t*)d->seg[s].dst, d->seg[s].ld, d->seg[s].n_begin, d->seg[s].n_end, d->seg[s].rope};
        KAI0_REQUIRE(d->seg[s].dst && (d->seg[s].ld % 4 == 0 || d->seg[s].rope == 2), "kai0_gemm_skinny_bf16: segment %d destination", s);
    }
    if (d->mode == 1) {
        int expect = 0;
        for (int s = 0; s < d->nseg; ++s) {
            KAI0_REQUIRE(d->seg[s].n_begin == expect && d->seg[s].n_end > expect && (d->seg[s].n_end - expect) % (2 * ps) == 0,
                         "kai0_gemm_skinny_bf16: segments must tile [0, N) in multiples of 2*pair_stride");
            expect = d->seg[s].n_end;
            if (d->seg[s].rope == 1)
                KAI0_REQUIRE(d->rope_cos && d->rope_sin && d->rope_half == ps,
                             "kai0_gemm_skinny_bf16: rope needs cos/sin tables with rope_half == pair_stride");
        }
        KAI0_REQUIRE(expect == d->N, "kai0_gemm_skinny_bf16: segments cover %d of N=%d columns", expect, d->N);
    }
    if (d->mode == 2) KAI0_REQUIRE(ps * 2 == d->N, "kai0_gemm_skinny_bf16: geglu needs W = [gate; up], pair_stride = N/2");
    a.gate = (const bf16_t*)d->gate;
    a.gate_rpb = d->gate_rpb > 0 ? d->gate_rpb : 1;
    a.gate_ld = d->gate_ld;
    a.residual = (const bf16_t*)d->residual;
    a.ldr = d->ldr;
    a.rope_cos = d->rope_cos;
    a.rope_sin = d->rope_sin;
    a.rope_half = d->rope_half > 0 ? d->rope_half : ps;
    KAI0_REQUIRE(!d->w_packed || (inblock && d->K % 32 == 0 && d->N % 16 == 0), "kai0_gemm_skinny_bf16: w_packed needs split_k == -1");
    if (inblock) {  // the in-block kernels address A with 32-bit byte offsets below a 2 GiB buffer descriptor
        const int64_t a_rows = d->a_rpb ? ((int64_t)(d->M / d->a_rpb) + 1) * d->a_bs + d->a_off + d->a_rpb : (int64_t)d->M;
        KAI0_REQUIRE(a_rows * d->lda * 2 < (int64_t)0x7FFF0000, "kai0_gemm_skinny_bf16: A spans more than 2 GiB (%lld rows)", (long long)a_rows);
    }
    a.w_packed = d->w_packed;
    if (inblock) {
        if (d->workspace != nullptr) {  // diagnostics: phase trace, int64 [row tiles x column tiles][8]
            KAI0_REQUIRE(((uintptr_t)d->workspace % 8) == 0 && d->workspace_bytes >= 8 * 8 * 4096, "kai0_gemm_skinny_bf16: trace buffer too small");
            a.ws = (float*)d->workspace;
        }
        return skinny_inblock(d, a, (hipStream_t)stream);
    }
    KAI0_REQUIRE(d->mod == nullptr && d->rowsq_in == nullptr && d->rowsq_out == nullptr,
                 "kai0_gemm_skinny_bf16: the adaRMS prologue / folded form / row statistics need split_k == -1");
    const int tiles = d->N / TN, mtiles = (d->M + TM - 1) / TM;
    if (S > 1) {
        const int64_t need = (int64_t)S * d->M * d->N * 4;
        KAI0_REQUIRE(d->mode == 0 && d->gate == nullptr && d->residual == nullptr,
                     "kai0_gemm_skinny_bf16: split_k > 1 produces raw partial products (mode 0, no gate / residual)");
        KAI0_REQUIRE(d->workspace && d->workspace_bytes >= need && ((uintptr_t)d->workspace % 16) == 0,
                     "kai0_gemm_skinny_bf16: split_k=%d needs %lld workspace bytes", S, (long long)need);
        a.ws = (float*)d->workspace;
    }
    const dim3 grid(tiles, S, mtiles);
    // (8 waves measured identical to 4 on the denoise loop, 24.44 vs 24.45 ms per chunk: removed)
    if (a.k_blk == 512)
        hipLaunchKernelGGL((skinny_kernel<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((skinny_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return kai0_check_launch("kai0_gemm_skinny_bf16");
}

KAI0_API int kai0_rope_table(const int32_t* pos, const float* inv_freq, void* cos_out, void* sin_out, int64_t rows,
                             int half, int out_bf16, kai0_stream_t stream) {
    const int64_t n = rows * half;
    if (n <= 0) return 0;
    if (out_bf16)
        hipLaunchKernelGGL(rope_table_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, inv_freq,
                           cos_out, sin_out, n, half);
    else
        hipLaunchKernelGGL(rope_table_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, inv_freq,
                           cos_out, sin_out, n, half);
    return kai0_check_launch("kai0_rope_table");
}
