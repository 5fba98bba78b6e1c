// gemm_bf16.hip — the bf16 MFMA GEMM under every Linear and attention matmul of the pi0.5 path.
//
// gfx950 design:
//   * two tile configurations of one template: 256x256x64 (8 waves as 2x4, each a 128x64 sub-tile = 8x4 MFMA
//     16x16x32 tiles; 128 KiB LDS, 1 block/CU) for problems that fill the chip, and 128x128x64 (4 waves, 64x64
//     each; 64 KiB, 2 blocks/CU) for small ones.  Measured (KAI0_GEMM_ABLATE): the 128x128 kernel is bound by the
//     L2->LDS staging rate (~14 TB/s chip-wide), not by MFMA or LDS reads, so the lever is bytes staged per
//     FLOP = tile size; 256x256 halves it.
//   * operands go HBM -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction), never
//     through VGPRs; two stages (A+B) double-buffered.
//   * the LDS image is lane-linear (DMA constraint), so the bank-conflict XOR swizzle is applied on the
//     per-lane GLOBAL source address and undone on the ds_read address (same involution both sides).
//   * K-contiguous operands are read with ds_read_b128; contraction-strided operands (dgrad's W,
//     wgrad's dY/X, attention's V) are read with ds_read_b64_tr_b16 (hardware 4x16 transpose), so no
//     operand is ever transposed through HBM.
//   * epilogue: accumulators -> wave-private LDS slab (f32) -> each lane owns 8 consecutive columns of a
//     row: bias/scale/GELU/gate/residual are applied with 16-B vector loads and the tile leaves as 16-B
//     stores. Rounding to bf16 happens exactly where the reference's bf16 torch ops round.
//   * 1-D grid, XCD-aware bijective remap + grouped raster so the blocks sharing an A/B panel sit on one
//     XCD's L2.
#include "common.h"
#include "../../include/kai0hip.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <type_traits>

namespace {

constexpr int BK = 64;

// timing ablations (no DMA inside the K loop: the compute + LDS-read ceiling) exist only in builds made with
// KAI0_HIPCC_FLAGS=-DKAI0_ABLATE (with KAI0_GEMM_ABLATE=1 in the environment; the round-1 driver tools/gemm_ablate.py is in the git history); the shipped kernels carry no ablation branch
#ifdef KAI0_ABLATE
#define KAI0_ABL(p) ((p).ablate)
#else
#define KAI0_ABL(p) 0
#endif

struct RowMap {
    int32_t rpb;
    int64_t bs, off;
    // rows are < 2^31: 32-bit divide (a 64-bit one costs ~100 VALU ops)
    __device__ __forceinline__ int64_t operator()(int r) const {
        if (rpb == 0) return r;
        const int q = r / rpb;
        return (int64_t)q * bs + (r - q * rpb) + off;
    }
};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    int M, N, K;
    int64_t lda, ldb, ldc;
    int batch_inner;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;
    RowMap amap, bmap, cmap;
    const void* bias;
    int bias_f32;
    float scale;
    int act;
    int out_f32;
    bf16_t* pre_out;
    const bf16_t* aux1;  // act 2/3: gate pre-activation g [M][ldc]
    const bf16_t* aux2;  // act 3: up projection u [M][ldc]
    const bf16_t* gate;
    int gate_rpb;
    int accumulate;
    int64_t gate_ld;
    const bf16_t* residual;
    int64_t ldr, sR1, sR2;
    int tiles_m, tiles_n;
    int ablate;  // diagnostics only (KAI0_GEMM_ABLATE=1): no DMA inside the K loop (compute-only ceiling)
    int nt_c;    // C stored non-temporally (kai0hip.h c_nontemporal: weight gradients, read again only by the optimizer)
    int simple_epi;  // the store-with-little-else fast epilogue may be used (kai0_gemm_desc.general_epilogue == 0)
    int nt_pre;  // pre-activation outputs (pre_out / pre_out2 of act 1 and 6: read again only by the backward) stored non-temporally
    const float* rowvec;   // act 4: per-row f32 vector D (softmax backward), index z1*rv_s1 + z2*rv_s2 + row*rv_ld
    int64_t rv_s1, rv_s2, rv_ld;
    int nseg;              // > 0: bf16 output columns are routed to up to 3 destinations (fused q|k|v projection)
    bf16_t* seg_dst[3];
    int64_t seg_ld[3];
    int seg_begin[3];
    float* ws;             // split-K workspace [batch*split_k][M][N] f32
    int split_k, k_chunk;  // split-K: blockIdx.y = z * split_k + s, split s owns k in [s*k_chunk, min(K, (s+1)*k_chunk))
    // act 6 (GeGLU pair): the B operand is TWO weights, gate = B and up = B2, interleaved in 32-row groups along the tile's N
    // axis (logical column 64 c + j: j < 32 -> gate row 32 c + j, else up row 32 c + j - 32), so that every wave's 64-column
    // sub-tile holds gate AND up of the same 32 output columns and GeGLU needs no operand from memory.  N is the LOGICAL width
    // (2 x output columns).
    const bf16_t* B2;
    bf16_t* pre_out2;
    // split-K only: the consumer's norm fused into the reduction launch (kai0hip.h norm_kind)
    bf16_t* norm_out;
    const void* norm_w;
    const bf16_t* norm_b;
    float norm_eps;
    int norm_kind;
    // act 7 (RoPE epilogue, kai0hip.h): bf16 tables [M][rope_half]; permuted columns < rope_n_end are rotated
    const bf16_t* rope_cos;
    const bf16_t* rope_sin;
    int rope_half, rope_n_end;
};

// LDS-DMA through a raw buffer descriptor: 16 B per lane from base + voff (bytes) to lds_dst + lane*16.  An offset at
// or beyond the descriptor's 2 GiB range returns zeros — that is how tile edges and the K tail are zero-filled
// (no second source pointer, no per-lane 64-bit address).
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
constexpr uint32_t OOB = 0x80000000u;
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_PTR(void))lds_dst_wave_uniform, 16, (int)voff, 0, 0, 0);
}

// Swizzle key of a contraction-strided (MC) tile row r: key(r) = (r & 3) | (((r >> 3) & 1) << 2).
// It spreads the 8 k-rows that a 32-lane half of ds_read_b64_tr_b16 touches ({0..3, 8..11} + 4h) over the 8
// distinct 32-B segments of the 256-B bank row.  Both the DMA source address and the read address apply it.

__device__ __forceinline__ void store_pre8(bf16_t* dst, bf16x8 v, int nt) {
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst));
    else *reinterpret_cast<bf16x8*>(dst) = v;
}

// The fused epilogue on 8 consecutive columns [ccol, ccol+8) of output row `row` (v = f32 accumulators), shared by the
// GEMM kernel and the split-K reduction.  Order and rounding points: see kai0hip.h.
__device__ __forceinline__ void epilogue8(const GemmArgs& p, float (&v)[8], int row, int ccol, int64_t cz, int64_t rz,
                                          int64_t vz, void* cbase, bool raw_f32, const bf16x8* res_pre = nullptr) {
    // res_pre: the residual operand already loaded by the caller (requested together with its other loads)
    // raw_f32: f32 output keeps the raw accumulator (no bf16 rounding points): gradients that must not be quantised
    // before a cancelling reduction (softmax backward) and split-K partial tiles.
    const bool rnd = !raw_f32;
    auto R = [rnd](float x) { return rnd ? rbf(x) : x; };
    const int64_t orow = p.cmap(row);
    if (p.act == 4) {
        // softmax backward fused into dP = dO V^T: C <- bf16( (P * (dP - D[row])) * scale ), D = rowsum(dO * O) (= the
        // row's <dP, P>), dP straight from the f32 accumulator (never rounded, never written)
        const bf16x8 pv = *reinterpret_cast<const bf16x8*>(p.aux1 + cz + orow * p.ldc + ccol);
        const float dsum = p.rowvec[vz + (int64_t)row * p.rv_ld];
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = f2bf((bf2f(pv[e]) * (v[e] - dsum)) * p.scale);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(cbase) + cz + orow * p.ldc + ccol) = ov;
        return;
    }
    if (p.bias != nullptr) {
        if (p.bias_f32) {
            const float* bp = reinterpret_cast<const float*>(p.bias) + ccol;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bp[e];
        } else {
            bf16x8 bv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.bias) + ccol);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf2f(bv[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = R(v[e]);
    if (p.scale != 1.0f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = R(v[e] * p.scale);
    }
    if (p.act == 1) {
        if (p.pre_out != nullptr) {
            bf16x8 pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = f2bf(v[e]);
            store_pre8(p.pre_out + cz + orow * p.ldc + ccol, pv, p.nt_pre);
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 a = gelu_tanh2(f32x2{v[e], v[e + 1]});
            v[e] = R(a[0]);
            v[e + 1] = R(a[1]);
        }
    } else if (p.act == 2) {
        // GeGLU forward fused into the up-projection GEMM: v = u; pre_out <- u; C <- bf16(bf16(gelu(g)) * u)
        if (p.pre_out != nullptr) {  // u is only needed by the backward
            bf16x8 uv;
#pragma unroll
            for (int e = 0; e < 8; ++e) uv[e] = f2bf(v[e]);
            *reinterpret_cast<bf16x8*>(p.pre_out + cz + orow * p.ldc + ccol) = uv;
        }
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(p.aux1 + cz + orow * p.ldc + ccol);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 a = gelu_tanh2(f32x2{bf2f(gv[e]), bf2f(gv[e + 1])});
            v[e] = R(R(a[0]) * v[e]);
            v[e + 1] = R(R(a[1]) * v[e + 1]);
        }
    } else if (p.act == 3) {
        // GeGLU backward fused into the down-projection dgrad: v = dh; pre_out <- du = bf16(dh * bf16(gelu(g)));
        // C <- dg = bf16(bf16(dh * u) * gelu'(g))
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(p.aux1 + cz + orow * p.ldc + ccol);
        const bf16x8 uv = *reinterpret_cast<const bf16x8*>(p.aux2 + cz + orow * p.ldc + ccol);
        bf16x8 du;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f32x2 gl, gr;
            gelu_tanh_both2(f32x2{bf2f(gv[e]), bf2f(gv[e + 1])}, gl, gr);
            du[e] = f2bf(v[e] * R(gl[0]));
            du[e + 1] = f2bf(v[e + 1] * R(gl[1]));
            v[e] = R(R(v[e] * bf2f(uv[e])) * gr[0]);
            v[e + 1] = R(R(v[e + 1] * bf2f(uv[e + 1])) * gr[1]);
        }
        *reinterpret_cast<bf16x8*>(p.pre_out + cz + orow * p.ldc + ccol) = du;
    } else if (p.act == 5) {
        // GELU backward fused into the dgrad that produces dh (SigLIP fc2): C <- bf16(dh * gelu_tanh'(pre)), pre = aux1
        const bf16x8 pv = *reinterpret_cast<const bf16x8*>(p.aux1 + cz + orow * p.ldc + ccol);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f32x2 gl, gr;
            gelu_tanh_both2(f32x2{bf2f(pv[e]), bf2f(pv[e + 1])}, gl, gr);
            v[e] = R(v[e] * gr[0]);
            v[e + 1] = R(v[e + 1] * gr[1]);
        }
    }
    if (p.gate != nullptr) {
        bf16x8 gv = *reinterpret_cast<const bf16x8*>(p.gate + (int64_t)(row / p.gate_rpb) * p.gate_ld + ccol);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = R(v[e] * bf2f(gv[e]));
    }
    if (p.residual != nullptr) {
        bf16x8 rv = res_pre != nullptr ? *res_pre : *reinterpret_cast<const bf16x8*>(p.residual + rz + orow * p.ldr + ccol);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = R(v[e] + bf2f(rv[e]));
    }
    if (raw_f32) {
        float* cp = reinterpret_cast<float*>(cbase) + cz + orow * p.ldc + ccol;
        if (p.accumulate) {
            f32x4 o0 = *reinterpret_cast<const f32x4*>(cp);
            f32x4 o1 = *reinterpret_cast<const f32x4*>(cp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += o0[e]; v[4 + e] += o1[e]; }
        }
        *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        bf16_t* cp = reinterpret_cast<bf16_t*>(cbase) + cz + orow * p.ldc + ccol;
        if (p.nseg > 0) {
            int si = 0;
            if (p.nseg > 1 && ccol >= p.seg_begin[1]) si = 1;
            if (p.nseg > 2 && ccol >= p.seg_begin[2]) si = 2;
            cp = p.seg_dst[si] + orow * p.seg_ld[si] + (ccol - p.seg_begin[si]);
        }
        if (p.accumulate) {
            bf16x8 ov = *reinterpret_cast<const bf16x8*>(cp);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf2f(ov[e]);
        }
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = f2bf(v[e]);
        store_pre8(cp, ov, p.nt_c);
    }
}

#define N_ALIGNED8(n) (((n) & 7) == 0)

// block barrier that does NOT drain the LDS-DMA queue behind the compiler's back: the kernel places its own vmcnt.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);  // keep register-only MFMAs on their side of the barrier (they ignore "memory")
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Block tile = (WM*MT*16) x (WN*NT*16) x 64, WM x WN waves, each wave MT x NT MFMA 16x16x32 tiles.
// PP (ping-pong, 8-wave tile only): the waves of tile-row 0 and tile-row 1 (one wave of each per SIMD) run one barrier
//   slot apart, so in every slot one group issues its 32 MFMAs while the other issues LDS-DMA and ds_reads.
// NS = LDS stages of the plain (non-PP) loop.  2: one K-tile of prefetch, enough when a second block on the CU (or the
//   sheer length of a 256x256 tile's MFMA burst) covers the load latency.  4 (128x128 tile, 128 KiB): three K-tiles in
//   flight with a counted vmcnt — for grids of <= 1 block per CU (B = 1 inference GEMMs), where a 2-stage loop runs at
//   one memory latency per K-tile.
// BKT = K-tile depth.  64 everywhere except the ring schedule (PP, NS = 4, BKT = 32): 4 slots of 32 KiB, three
//   32-deep sub-tiles in flight, every load slot carries 2 DMA pieces + 12 fragment reads and every MFMA slot 32 MFMAs
//   + 2 DMA pieces, with nothing conditional inside the loop (tail pieces are issued out of range = zero fill).
// SCH (PP, NS = 2, BKT = 64): 0 = the two-buffer ping-pong below; 1 / 2 = the quadrant schedule ("8 phases" per
//   two K-tiles): every K-tile is four phases of [fragment reads of one half-operand + 2 DMA pieces | barrier | 16 MFMAs of one
//   64x32 quadrant of the wave's 128x64 sub-tile | barrier]; the four half-tiles of a K-tile (A rows / B columns of the two
//   quadrant halves, 16 KiB each) are staged one per phase, 4-6 phases ahead of their first read, into the half-buffer whose
//   previous occupant died earliest, with counted waits (never vmcnt(0) inside the loop).  1: pieces issued after the phase's
//   fragment reads; 2: pieces issued in the middle of the phase's MFMAs.
template <bool A_KC, bool B_KC, int WM, int WN, int MT, int NT, bool PP, int NS = 2, int BKT = 64, int SCH = 0>
__global__ __launch_bounds__(WM * WN * 64, ((WM * WN * 64) / 256) * ((MT == 8 && WM * WN == 4) ? 2 : 1)) void gemm_bf16_kernel(const GemmArgs p) {  // (256 x 128 on four waves: two blocks per CU)
    static_assert(!PP || (WM == 2 && WN == 4), "ping-pong schedule: two 4-wave groups");
    static_assert(SCH == 0 || (SCH == 1 && PP && NS == 2 && BKT == 64 && MT == 8 && NT == 4) || (SCH == 3 && PP && NS == 4), "schedules: 0 plain / two-buffer ping-pong, 1 quadrant (256x256x64), 3 ring (256x256x32)");
    static_assert(!(PP && NS == 4) || SCH == 3, "the ring runs with every DMA piece between the MFMA rows");
    static_assert(NS >= 2 && (!PP || NS == 2 || (NS == 4 && BKT == 32)), "stages");
    static_assert(BKT == 64 || BKT == 32, "K-tile depth");
    constexpr int BK = BKT;
    constexpr int NWAVES = WM * WN;
    constexpr int TBM = WM * MT * 16, TBN = WN * NT * 16;
    constexpr int A_TILE = TBM * BK * 2, B_TILE = TBN * BK * 2;  // bytes: [rows][BK] or [BK][cols] bf16
    constexpr int STAGE = A_TILE + B_TILE;
    constexpr int NA = A_TILE / 1024 / NWAVES;    // A DMA pieces (1 KiB) per wave per K-tile
    constexpr int NB = B_TILE / 1024 / NWAVES;
    constexpr int KC_CPR = BK / 8;                // K-contiguous tile: 16-B chunks per row (8 / 4) ...
    constexpr int KC_RPP = 64 / KC_CPR;           // ... and rows per DMA piece (8 / 16)
    constexpr int A_ROWB = TBM * 2, B_ROWB = TBN * 2;  // bytes per k-row of a contraction-strided tile (256 / 512)
    constexpr int A_LPR = A_ROWB / 16, B_LPR = B_ROWB / 16;  // 16-B chunks (= lanes) per such row
    constexpr int A_RPP = 64 / A_LPR, B_RPP = 64 / B_LPR;    // k-rows per DMA piece
    constexpr int GROUP = TBM == 128 ? 8 : 4;
    static_assert(NA >= 1 && NB >= 1 && NT == 4 && ((MT % 4) == 0 || MT == 2), "unsupported tile configuration");
    constexpr int AH = MT >= 4 ? 4 : MT;  // 16-row MFMA tiles per epilogue pass / A-fragment group (64 rows; 32 for the 8-wave 128 x 128 tile)
    static_assert(A_KC || (64 % A_LPR) == 0, "contraction-strided A needs a power-of-two tile height");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    if constexpr (MT <= 4) {
        // 128 x 128 configurations (the B = 1 inference GEMMs: a launch is ~20 us, ~190 of them per action chunk): every kernel
        // argument the prologue needs is pulled into SGPRs in ONE batch of scalar loads.  Left to itself hipcc loads each field of
        // the by-value struct next to its first use: six serial s_load / s_waitcnt round trips stood in front of the first LDS-DMA.
        asm volatile("" ::"s"(p.A), "s"(p.B), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.lda), "s"(p.ldb), "s"(p.tiles_m), "s"(p.tiles_n),
                     "s"(p.split_k), "s"(p.k_chunk), "s"(p.batch_inner), "s"(p.amap.rpb), "s"(p.bmap.rpb), "s"(p.act), "s"(p.sA1),
                     "s"(p.sB1), "s"(p.sA2), "s"(p.sB2));
    }
    // ---- block -> tile (XCD-aware bijective remap, then grouped raster) -------------------------
    const int nwg = gridDim.x;
    int pid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = pid & 7, idx = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    {
        const int width = GROUP * p.tiles_n;
        const int group = pid / width;
        const int first_m = group * GROUP;
        const int gsz = min(p.tiles_m - first_m, GROUP);
        const int in_g = pid - group * width;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int ks_id = blockIdx.y % p.split_k;
    const int z = blockIdx.y / p.split_k;
    const int kbeg = ks_id * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int z1 = z / p.batch_inner, z2 = z - z1 * p.batch_inner;
    const bf16_t* __restrict__ Ab = p.A + z1 * p.sA1 + z2 * p.sA2;
    const bf16_t* __restrict__ Bb = p.B + z1 * p.sB1 + z2 * p.sB2;
    const int m0 = tile_m * TBM, n0 = tile_n * TBN;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)OOB, 0x00020000);
    const bool pair = p.act == 6;  // (host: A and B K-contiguous, one batch entry, no split-K)
    const __amdgpu_buffer_rsrc_t b2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(pair ? p.B2 : Bb), 0, (int)OOB, 0x00020000);

    // ---- staging source addresses ----------------------------------------------------------------
    // K-contiguous tile [rows][64 k] (128-B rows): DMA piece q covers rows q*8..q*8+7; lane -> row q*8+(lane>>3),
    //   16-B slot lane&7 which holds source chunk (lane&7)^(row&7).
    // contraction-strided tile [64 k][cols] (256- or 512-B rows): piece q covers RPP k-rows; lane -> k-row
    //   q*RPP + lane/LPR, slot lane%LPR holding source chunk slot^(key(row)<<1)  (key: see above).
    uint32_t a_off[NA], b_off[NB];  // byte offsets from the operand base; OOB = this lane's chunk lies outside
    // KC: k offset (elements) of this lane's chunk in the K-tile.  BK = 64: slot lane&7 of row lane>>3 holds source chunk
    // slot ^ (row & 7).  BK = 32 (64-B rows): slot lane&3 of row lane>>2 holds chunk slot ^ (row & 8 ? 3 : 0), which makes
    // the four 16-lane groups of a ds_read_b128 (rows 0-3/12-15 with rows 4-11 of the next chunk) hit 16 distinct slots.
    const int kc_chunk = BK == 64 ? ((lane & 7) ^ (lane >> 3)) * 8 : ((lane & 3) ^ (((lane >> 2) & 8) ? 3 : 0)) * 8;
    if constexpr (A_KC) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int R = m0 + (wave * NA + j) * KC_RPP + lane / KC_CPR;
            a_off[j] = R < p.M ? (uint32_t)((p.amap(R) * p.lda + kc_chunk) * 2) : OOB;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int r = (wave * NA + j) * A_RPP + lane / A_LPR;
            const int key = (r & 3) | (((r >> 3) & 1) << 2);
            const int col = m0 + ((lane % A_LPR) ^ (key << 1)) * 8;
            a_off[j] = col < p.M ? (uint32_t)(col * 2) : OOB;  // + stored_row(k)*lda*2 added per K-tile
        }
    }
    if constexpr (B_KC) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int rt = (wave * NB + j) * KC_RPP + lane / KC_CPR;  // row of the B tile
            int R = n0 + rt;
            bool ok = R < p.N;
            if (pair) {  // 32-row groups alternate gate / up; both index the same weight rows (n0 / 2 + 32 (rt / 64) + rt % 32)
                R = (n0 >> 1) + ((rt >> 6) << 5) + (rt & 31);
                ok = R < (p.N >> 1);
            }
            b_off[j] = ok ? (uint32_t)((p.bmap(R) * p.ldb + kc_chunk) * 2) : OOB;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = (wave * NB + j) * B_RPP + lane / B_LPR;
            const int key = (r & 3) | (((r >> 3) & 1) << 2);
            const int col = n0 + ((lane % B_LPR) ^ (key << 1)) * 8;
            b_off[j] = col < p.N ? (uint32_t)(col * 2) : OOB;
        }
    }
    const uint32_t lda2 = (uint32_t)p.lda * 2, ldb2 = (uint32_t)p.ldb * 2;

    auto stage = [&](int kt, int slot) {
        char* sa = smem + slot * STAGE + wave * (NA * 1024);
        char* sb = smem + slot * STAGE + A_TILE + wave * (NB * 1024);
        const int k0 = kbeg + kt * BK;
        const bool kc_in = (k0 + kc_chunk) < kend;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            uint32_t off;
            if constexpr (A_KC) {
                off = kc_in ? a_off[j] + (uint32_t)k0 * 2 : OOB;
            } else {
                const int kr = k0 + (wave * NA + j) * A_RPP + lane / A_LPR;
                off = kr < kend ? a_off[j] + (uint32_t)p.amap(kr) * lda2 : OOB;
            }
            glds16(a_rsrc, off, sa + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            uint32_t off;
            if constexpr (B_KC) {
                off = kc_in ? b_off[j] + (uint32_t)k0 * 2 : OOB;
            } else {
                const int kr = k0 + (wave * NB + j) * B_RPP + lane / B_LPR;
                off = kr < kend ? b_off[j] + (uint32_t)p.bmap(kr) * ldb2 : OOB;
            }
            // pair: the piece's 8 tile rows lie in one 32-row group -> gate or up weight, wave-uniform
            glds16((B_KC && BK == 64 && pair && (((wave * NB + j) * KC_RPP) & 32)) ? b2_rsrc : b_rsrc, off, sb + j * 1024);
        }
    };

    // one 1-KiB piece of a K-tile (idx < NA: A pieces, then B pieces): source offset (VALU, computed ahead of time) and
    // the bare DMA instruction, so that pieces can be dropped between MFMAs with nothing but an s_mov m0 around them
    auto piece_off = [&](int kt, int idx) -> uint32_t {
        const int k0 = kbeg + kt * BK;
        const bool kc_in = (k0 + kc_chunk) < kend;
        if (idx < NA) {
            if constexpr (A_KC) {
                return kc_in ? a_off[idx] + (uint32_t)k0 * 2 : OOB;
            } else {
                const int kr = k0 + (wave * NA + idx) * A_RPP + lane / A_LPR;
                return kr < kend ? a_off[idx] + (uint32_t)p.amap(kr) * lda2 : OOB;
            }
        } else {
            const int j = idx - NA;
            if constexpr (B_KC) {
                return kc_in ? b_off[j] + (uint32_t)k0 * 2 : OOB;
            } else {
                const int kr = k0 + (wave * NB + j) * B_RPP + lane / B_LPR;
                return kr < kend ? b_off[j] + (uint32_t)p.bmap(kr) * ldb2 : OOB;
            }
        }
    };
    auto piece_issue = [&](uint32_t off, int slot, int idx) {
        if (idx < NA) glds16(a_rsrc, off, smem + slot * STAGE + wave * (NA * 1024) + idx * 1024);
        else glds16(b_rsrc, off, smem + slot * STAGE + A_TILE + wave * (NB * 1024) + (idx - NA) * 1024);
    };

    // ---- fragment read offsets (bytes inside an operand tile), fixed per thread -----------------
    const int l15 = lane & 15, g = lane >> 4;
    // KC: row = w*(T*16) + t*16 + l15 ; chunk = ks*4 + g ; addr = row*128 + ((chunk ^ (row&7)) << 4)
    // MC: cols n = w*(T*16) + t*16 ; k-row r = ks*32 + 8g + 4h + (l15>>2) ; key = (l15>>2)|((g&1)<<2)
    //     chunk = n/8 + ((l15&3)>>1) ; addr = r*ROWB + ((chunk ^ (key<<1)) << 4) + (l15&1)*8
    const int mc_keyv = (l15 >> 2) | ((g & 1) << 2);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_frag = [&](const char* tile, bool kc, int rowb, int col0, int ks) -> bf16x8 {
        if (kc) {
            const int row = col0 + l15;
            const int off = BK == 64 ? row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4)
                                     : row * 64 + ((g ^ ((row & 8) ? 3 : 0)) << 4);
            return *reinterpret_cast<const bf16x8*>(tile + off);
        } else {
            const int chunk = (col0 >> 3) + ((l15 & 3) >> 1);
            const int sw = ((chunk ^ (mc_keyv << 1)) << 4) + (l15 & 1) * 8;
            const int r0 = ks * 32 + 8 * g + (l15 >> 2);
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))(tile + r0 * rowb + sw));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))(tile + (r0 + 4) * rowb + sw));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };

    const int nk = (kend - kbeg + BK - 1) / BK;
    if constexpr (PP && NS == 4) {
        // Ring schedule.  Sub-tile u (32 deep) lives in slot u & 3.  Group g (= wm) runs  L(u) |b| M(u) |b|  at barrier
        // slots 2u+g, 2u+1+g: in every slot one group issues MFMAs while the other reads fragments.
        //   L(u): fragment reads of sub-tile u, then DMA pieces 0,1 of sub-tile u+3;  M(u): 32 MFMAs with pieces 2,3 of
        //   sub-tile u+3 dropped in after MFMA rows 2 and 5 (their offsets were computed in L(u)).
        //  * Why: one LDS-DMA piece costs its wave 100-180 cycles next to ds_reads and ~60 between MFMAs; the former
        //    schedule put all 8 pieces of a 64-deep tile in one load slot (~1300 cycles against the partner's 512 of
        //    MFMA), and the loop without DMA ran at 1.7-2.0 PFLOP/s against 1.1 with it.
        //  * WAR: slot (u+3)&3 held sub-tile u-1, last read in group 1's L(u-1) (slot 2u-1, retired by the lgkmcnt(0)
        //    in front of its barrier); the earliest piece of u+3 is issued in group 0's L(u) (slot 2u).
        //  * RAW: sub-tile u+1 is first read in slot 2u+2, so every wave drains its pieces of u+1 before the barrier that
        //    ends slot 2u+1 — group 0 at the end of M(u) with pieces of u+2, u+3 (8) still in flight, group 1 at the end
        //    of L(u) with those of u+2 and the first two of u+3 (6).  Counts stay exact in the tail because pieces past
        //    the end are still issued (out-of-range offset: zero fill into a slot nobody reads).
        constexpr int NP = NA + NB;
        static_assert(NP == 4 && BK == 32, "ring schedule: 4 pieces per 32-deep sub-tile");
        const int grp = wm;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) piece_issue(piece_off(u, pi), u, pi);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        lds_barrier();
        if (grp == 1) lds_barrier();  // stagger group 1 by one slot
        for (int u = 0; u < nk; ++u) {
            const int slot = u & 3, pslot = (u + 3) & 3;
            const char* ta = smem + slot * STAGE;
            const char* tb = ta + A_TILE;
            bf16x8 bfr[NT], af[MT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bfr[t] = load_frag(tb, B_KC, B_ROWB, wn * (NT * 16) + t * 16, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = load_frag(ta, A_KC, A_ROWB, wm * (MT * 16) + t * 16, 0);
            uint32_t poff[NP];
            if constexpr (SCH == 0) {
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) poff[pi] = KAI0_ABL(p) == 1 ? OOB : piece_off(u + 3, pi);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SCH == 0) {
                piece_issue(poff[0], pslot, 0);
                piece_issue(poff[1], pslot, 1);
                if (grp == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                // SCH == 3: all four pieces of u+3 go out between the MFMA rows (a piece costs its wave ~60 cycles there against
                // 100-180 next to the fragment reads); group 1 therefore ends L(u) with only u+2's four pieces younger than u+1's
                if (grp == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            lds_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                if (SCH == 0 && (i == 2 || i == 5)) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece_issue(poff[i == 2 ? 2 : 3], pslot, i == 2 ? 2 : 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (SCH == 3 && (i & 1) == 0) {  // (the offset arithmetic sits in the MFMA shadow too)
                    __builtin_amdgcn_sched_barrier(0);
                    poff[i >> 1] = KAI0_ABL(p) == 1 ? OOB : piece_off(u + 3, i >> 1);
                    piece_issue(poff[i >> 1], pslot, i >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (grp == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            lds_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing zero-fill pieces must land before the slabs reuse LDS
        if (grp == 0) lds_barrier();  // re-align the two groups
        lds_barrier();
    } else if constexpr (PP && NS == 2 && SCH != 0) {
        // Quadrant schedule.  Wave (wm, wn) owns rows wm*128 + [0, 128), columns wn*64 + [0, 64) of the tile; half-operand
        // A_a = its rows a*64 + [0, 64) (for both wm: 128 tile rows), B_b = its columns b*32 + [0, 32) (for all four wn: 128 tile
        // columns); 16 KiB = 16 DMA pieces each, two per wave.  Tile t (LDS slot t & 1) runs the quadrants
        //     p0: (A0, B0)   p1: (A0, B1)   p2: (A1, B1)   p3: (A1, B0)
        // reading A0 + B0 | B1 | A1 | B0 again, and issues  p0: B0(t+1)  p1: A1(t+1)  p2: A0(t+2)  p3: B1(t+2).
        //  * WAR: each issue targets the half-buffer of a half-tile whose last read lies >= 1 barrier slot back for BOTH groups
        //    (B0(t-1): p3 of t-1; A1(t-1): p2 of t-1; A0(t): p0 of t; B1(t): p1 of t; group 1 runs one slot behind group 0).
        //  * RAW: p0 of t reads A0(t), B0(t): of the pieces issued so far only A1(t), A0(t+1), B1(t+1) are younger than B0(t),
        //    so vmcnt(6) in the load part of p3 of t-1 (both groups, i.e. >= 1 barrier before any group's p0) retires them; p2
        //    reads A1(t): younger are A0(t+1), B1(t+1), B0(t+1), A1(t+1) -> vmcnt(8) in the load part of p1.  The prologue
        //    issues what tiles -2 and -1 would have, in the same order, so the counts hold from the first tile; pieces past
        //    the last tile are issued out of range (zero fill into half-buffers nobody reads), so they hold in the tail too.
        //  * Contraction-strided operands ([K][M] / [K][N] in memory, ds_read_b64_tr_b16 fragments): a half-operand is 128
        //    CONSECUTIVE rows / columns of the tile (a 256-B run per k-row: whole cache lines), staged as an image [64 k][128]
        //    with the usual key swizzle, 4 k-rows per DMA piece; the waves' sub-tiles are interleaved accordingly — wave wm
        //    owns rows a*128 + wm*64 + [0, 64), a = 0, 1 (wave wn: columns b*128 + wn*32 + [0, 32)) — and the epilogue maps
        //    its accumulators back through the same formula (ILM / ILN below).
        const int grp = wm;
        uint32_t ha_off[4], hb_off[4];  // idx = half * 2 + j: byte offset of this lane's 16 B (without the K-tile term)
        int ha_lds[4], hb_lds[4];       // LDS byte offset of the piece inside its operand tile (wave-uniform)
        int ha_kr[4], hb_kr[4];         // contraction-strided: this lane's k-row inside the K-tile
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = 2 * wave + j, x = h * 2 + j;
                if constexpr (A_KC) {
                    const int ra = (q >> 3) * 128 + h * 64 + (q & 7) * 8, Ra = m0 + ra + (lane >> 3);
                    ha_lds[x] = ra * 128;
                    ha_kr[x] = 0;
                    ha_off[x] = Ra < p.M ? (uint32_t)((p.amap(Ra) * p.lda + kc_chunk) * 2) : OOB;
                } else {
                    const int r = q * 4 + (lane >> 4), key = (r & 3) | (((r >> 3) & 1) << 2);
                    const int col = m0 + h * 128 + (((lane & 15) ^ (key << 1)) * 8);
                    ha_lds[x] = h * 16384 + q * 1024;
                    ha_kr[x] = r;
                    ha_off[x] = col < p.M ? (uint32_t)(col * 2) : OOB;
                }
                if constexpr (B_KC) {
                    const int rb = (q >> 2) * 64 + h * 32 + (q & 3) * 8;
                    int Rb = n0 + rb + (lane >> 3);
                    bool ok = Rb < p.N;
                    if (pair) {  // half-operand B_h = 32-column group h of every wave: h = 0 gate rows, h = 1 up rows (issue_half)
                        Rb = (n0 >> 1) + (q >> 2) * 32 + (q & 3) * 8 + (lane >> 3);
                        ok = Rb < (p.N >> 1);
                    }
                    hb_lds[x] = rb * 128;
                    hb_kr[x] = 0;
                    hb_off[x] = ok ? (uint32_t)((p.bmap(Rb) * p.ldb + kc_chunk) * 2) : OOB;
                } else {
                    const int r = q * 4 + (lane >> 4), key = (r & 3) | (((r >> 3) & 1) << 2);
                    const int col = n0 + h * 128 + (((lane & 15) ^ (key << 1)) * 8);
                    hb_lds[x] = h * 16384 + q * 1024;
                    hb_kr[x] = r;
                    hb_off[x] = col < p.N ? (uint32_t)(col * 2) : OOB;
                }
            }
        auto issue_half = [&](bool isb, int h, int t) {
            const int k0 = kbeg + t * BK;
            const bool kc_in = (k0 + kc_chunk) < kend;
            char* base = smem + (t & 1) * STAGE + (isb ? A_TILE : 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int x = h * 2 + j;
                const uint32_t o = isb ? hb_off[x] : ha_off[x];
                uint32_t off = OOB;
                if (isb ? B_KC : A_KC) {
                    if (kc_in && o != OOB) off = o + (uint32_t)k0 * 2;
                } else {
                    const int kr = k0 + (isb ? hb_kr[x] : ha_kr[x]);
                    if (kr < kend && o != OOB) off = o + (uint32_t)(isb ? p.bmap(kr) : p.amap(kr)) * (isb ? ldb2 : lda2);
                }
                if (KAI0_ABL(p) == 1) off = OOB;
                glds16(isb ? ((pair && h == 1) ? b2_rsrc : b_rsrc) : a_rsrc, off, base + (isb ? hb_lds[x] : ha_lds[x]));
            }
        };
        issue_half(false, 0, 0);
        issue_half(true, 1, 0);
        issue_half(true, 0, 0);
        issue_half(false, 1, 0);
        issue_half(false, 0, 1);
        issue_half(true, 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        lds_barrier();
        if (grp == 1) lds_barrier();  // stagger group 1 by one slot
        bf16x8 af[4][2], bq[2][2];
        auto read_a = [&](const char* ta, int h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    af[i][ks] = A_KC ? load_frag(ta, true, A_ROWB, wm * 128 + (h * 4 + i) * 16, ks)
                                     : load_frag(ta + h * 16384, false, 256, wm * 64 + i * 16, ks);
        };
        auto read_b = [&](const char* tb, int h) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    bq[j][ks] = B_KC ? load_frag(tb, true, B_ROWB, wn * 64 + (h * 2 + j) * 16, ks)
                                     : load_frag(tb + h * 16384, false, 256, wn * 32 + j * 16, ks);
        };
        // 16 MFMAs of quadrant (ah, bh); SCH == 2: the phase's two DMA pieces go out after the 8th
        auto quad = [&](auto ahc, auto bhc, bool isb, int h, int t) {
            constexpr int ah = decltype(ahc)::value, bh = decltype(bhc)::value;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][bh * 2 + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bq[j][ks], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
                if (SCH == 2 && ks == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_half(isb, h, t);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        for (int t = 0; t < nk; ++t) {
            const char* ta = smem + (t & 1) * STAGE;
            const char* tb = ta + A_TILE;
            // ---- p0: (A0, B0); stage B0(t+1)
            read_b(tb, 0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(ta, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (SCH == 1) issue_half(true, 0, t + 1);
            lds_barrier();
            quad(I0{}, I0{}, true, 0, t + 1);
            lds_barrier();
            // ---- p1: (A0, B1); stage A1(t+1); A1(t) must have landed one phase from now
            read_b(tb, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (SCH == 1) {
                issue_half(false, 1, t + 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {  // the phase's own pieces are not issued yet: one half-tile fewer in flight
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            lds_barrier();
            quad(I0{}, I1{}, false, 1, t + 1);
            lds_barrier();
            // ---- p2: (A1, B1); stage A0(t+2) (same slot as this tile: A0(t) died in p0)
            read_a(ta, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (SCH == 1) issue_half(false, 0, t + 2);
            lds_barrier();
            quad(I1{}, I1{}, false, 0, t + 2);
            lds_barrier();
            // ---- p3: (A1, B0); stage B1(t+2); A0(t+1), B0(t+1) must have landed one phase from now
            read_b(tb, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (SCH == 1) {
                issue_half(true, 1, t + 2);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            lds_barrier();
            quad(I1{}, I0{}, true, 1, t + 2);
            lds_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing zero-fill pieces must land before the slabs reuse LDS
        if (grp == 0) lds_barrier();  // re-align the two groups
        lds_barrier();
    } else if constexpr (PP) {
        // Two-buffer ping-pong (kept for comparison, kai0_gemm_desc.tile_cfg = 5).  Barrier clock b0, b1, ...: per K-tile t group 0 runs
        // L0(t) |b| M0(t) |b| L1(t) |b| M1(t) |b|  and group 1 the same sequence one barrier later.  Lk = [k = 0: issue the
        // whole DMA of tile t+1] + the 12 fragment reads of k-half k; Mk = its 32 MFMAs.
        //  * RAW: tile t+1 is first read after barrier 4t+3 (group 0's L0(t+1)); every wave drains its own DMA before
        //    that barrier (group 0 at the end of M1(t), group 1 at the end of L1(t)), >= 2 slots after issuing it.
        //  * WAR: the DMA of tile t+1 overwrites the buffer of tile t-1, last read in group 1's L1(t-1) and retired by
        //    the lgkmcnt(0) in front of barrier 4t-1; the earliest issue (group 0, L0(t)) comes after that barrier.
        const int grp = wm;
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (grp == 1) lds_barrier();  // stagger group 1 by one slot
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const char* ta = smem + buf * STAGE;
            const char* tb = ta + A_TILE;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                if (ks == 0 && kt + 1 < nk && KAI0_ABL(p) != 1) stage(kt + 1, buf ^ 1);
                bf16x8 bfr[NT], af[MT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bfr[t] = load_frag(tb, B_KC, B_ROWB, wn * (NT * 16) + t * 16, ks);
#pragma unroll
                for (int t = 0; t < MT; ++t) af[t] = load_frag(ta, A_KC, A_ROWB, wm * (MT * 16) + t * 16, ks);
                if (ks == 1 && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                if (ks == 1 && grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
            }
        }
        if (grp == 0) lds_barrier();  // re-align the two groups
    } else {
        // iteration kt: [tile kt landed (counted vmcnt) | barrier | issue tile kt+NS-1 into the slot tile kt-1 just left |
        // compute tile kt].  The barrier orders both the RAW on tile kt and the WAR on the slot being refilled.
#pragma unroll
        for (int st = 0; st < NS - 1; ++st)
            if (st < nk) stage(st, st);
        int slot_c = 0, slot_p = NS - 1;  // slot being computed / slot being refilled
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (NS == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (NA + NB)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();
            if (kt + NS - 1 < nk && KAI0_ABL(p) != 1) stage(kt + NS - 1, slot_p);
            const char* ta = smem + slot_c * STAGE;
            const char* tb = ta + A_TILE;
    #pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                bf16x8 bfr[NT];
    #pragma unroll
                for (int t = 0; t < NT; ++t) bfr[t] = load_frag(tb, B_KC, B_ROWB, wn * (NT * 16) + t * 16, ks);
    #pragma unroll
                for (int h = 0; h < MT / AH; ++h) {  // A fragments 4 at a time: bounds the live registers of the 8x4 tiling
                    bf16x8 af[AH];
    #pragma unroll
                    for (int t = 0; t < AH; ++t) af[t] = load_frag(ta, A_KC, A_ROWB, wm * (MT * 16) + (h * AH + t) * 16, ks);
    #pragma unroll
                    for (int i = 0; i < AH; ++i)
    #pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[h * AH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[h * AH + i][j], 0, 0, 0);
                }
            }
            slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
            slot_p = slot_p + 1 == NS ? 0 : slot_p + 1;
        }
        lds_barrier();  // the epilogue slabs alias the stage buffers
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // wave-private slab: f32 [64][64] at smem + wave*16 KiB (8-wave 128 x 128 tile: [32][64], 8 KiB), filled MT/4 times (64 rows of the wave's sub-tile each).
    // C layout of a 16x16 tile: col = lane&15, row = 4*(lane>>4) + reg.
    float* slab = reinterpret_cast<float*>(smem + wave * (AH * 4096));  // (f32 [AH * 16][64])
    const int64_t cz = z1 * p.sC1 + z2 * p.sC2;
    const int64_t rz = z1 * p.sR1 + z2 * p.sR2;
    const int64_t vz = z1 * p.rv_s1 + z2 * p.rv_s2;
    // quadrant schedule with a contraction-strided operand: that operand's sub-tiles are interleaved (see the schedule)
    constexpr bool ILM = (SCH == 1 || SCH == 2) && !A_KC, ILN = (SCH == 1 || SCH == 2) && !B_KC;
    const int ccol = ILN ? n0 + ((lane & 7) >> 2) * 128 + wn * 32 + ((lane & 7) & 3) * 8 : n0 + wn * 64 + (lane & 7) * 8;
    auto row_base = [&](int h) { return ILM ? m0 + h * 128 + wm * 64 : m0 + wm * (MT * 16) + h * 64; };
    // 8-wide column groups: when N % 8 != 0 the last group's extra columns hold exact zeros (their B rows are
    // zero-filled) and are stored into the row padding the host guarantees (ldc >= round_up(N, 8)).
    const bool col_ok = ccol < ((p.N + 7) & ~7);
    // one 64-row half of the wave's sub-tile at a time; `h` is a compile-time constant so acc[] keeps static indices
    // (a rolled loop here would turn the accumulators into an indexed array for the whole kernel)
    // Fused GeGLU / softmax-backward epilogues read one or two more [M][N] operands.  In the rolled loop below every
    // iteration would wait out a full global-load latency (16 per tile, +24 us on a 62-us tile); the fast path issues the
    // 8 (x2) operand loads of a 64-row half before the accumulators even go to the slab and consumes them unrolled.
    const bool fused_fast = p.act >= 2 && p.split_k == 1 && p.bias == nullptr && p.gate == nullptr && p.residual == nullptr &&
                            !p.accumulate && !p.out_f32 && p.nseg == 0 && (p.act == 4 || p.scale == 1.0f) &&
                            (p.act != 3 || p.pre_out != nullptr);
    // plain epilogues that read operands (bias and / or residual; optional GELU): see epi_half
    const bool plain_fast = p.act <= 1 && p.split_k == 1 && (p.bias != nullptr || p.residual != nullptr) && p.gate == nullptr &&
                            !p.accumulate && !p.out_f32 && p.nseg == 0 && (N_ALIGNED8(p.N));
    // 256x256 configurations: the store-with-little-else epilogues (act 0 / 1, optional bias, optional residual, optional column routing,
    // optional accumulation into the bf16 destination; no gate / f32 output / scale / row map) without the general path's per-row checks of everything else — measured in
    // the persistent kernel: 9.7 -> 4.6 us per tile for a plain store (profiles/r05_gemm_persistent_phases.txt).  Same order and rounding
    // points as epilogue8; kai0_gemm_desc.general_epilogue = 1 sends these launches through the general path (tests).
    const bool simple_fast = MT == 8 && p.simple_epi && p.act <= 1 && p.split_k == 1 && p.gate == nullptr && !p.out_f32 && p.scale == 1.0f &&
                             p.cmap.rpb == 0 && (N_ALIGNED8(p.N));
    auto epi_half = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        if constexpr (((MT == 4 && WM == 2) || (MT == 2 && WM == 4)) && WN == 2 && A_KC && B_KC) if (p.act == 7) {
            // RoPE epilogue (kai0hip.h act 7): the tile's columns are [64 first-half columns of a head | their 64 partners] (the caller
            // permuted B's rows), i.e. the partner of this wave's column c is column c of the OTHER wave of its tile row (wave ^ 1).
            // Accumulators -> the wave-private slabs, block barrier, then every lane reads its 8 columns from its own slab and the
            // partners from the neighbour's.  The rotation's tables (bf16, the values of kai0_rope_table) for the lane's 8 rows are
            // requested before the barrier.  Rounding points: x = bf16(acc), then exactly rope_kernel's (elementwise.hip).
            const bool rot = n0 < p.rope_n_end;                                  // tile-uniform
            const int fi = ((n0 >> 7) & 1) * 64 + (lane & 7) * 8;                // frequency index of the lane's 8 columns
            const int rcol = rot ? (n0 >> 8) * 256 + wn * 128 + fi : n0 + wn * 64 + (lane & 7) * 8;   // REAL column
            const int rbase = row_base(h) + (lane >> 3);
            bf16x8 cs[AH * 2], sn[AH * 2];
#pragma unroll
            for (int it = 0; it < AH * 2; ++it) {
                const int64_t tr = (int64_t)min(rbase + it * 8, p.M - 1) * p.rope_half + fi;  // (clamped: unconditional loads)
                cs[it] = *reinterpret_cast<const bf16x8*>(p.rope_cos + (rot ? tr : 0));
                sn[it] = *reinterpret_cast<const bf16x8*>(p.rope_sin + (rot ? tr : 0));
            }
#pragma unroll
            for (int i = 0; i < AH; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(i * 16 + 4 * g + r) * 64 + j * 16 + l15] = acc[i][j][r];
            lds_barrier();
            const float* pslab = reinterpret_cast<const float*>(smem + (wave ^ 1) * (AH * 4096));
            int si = 0;
            if (p.nseg > 1 && rcol >= p.seg_begin[1]) si = 1;
            if (p.nseg > 2 && rcol >= p.seg_begin[2]) si = 2;
            const bool rcol_ok = rcol < p.N;
#pragma unroll
            for (int it = 0; it < AH * 2; ++it) {
                const int lr = it * 8 + (lane >> 3);
                const int row = rbase + it * 8;
                const int so = lr * 64 + (lane & 7) * 8;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + so), v1 = *reinterpret_cast<const f32x4*>(slab + so + 4);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(pslab + so), w1 = *reinterpret_cast<const f32x4*>(pslab + so + 4);
                if (row >= p.M || !rcol_ok) continue;
                const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const float y[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xe = rbf(x[e]), ye = rbf(y[e]);
                    const float c = bf2f(cs[it][e]), sv = bf2f(sn[it][e]);
                    // first half (wn == 0): x1 cos - x2 sin;  second half: x2 cos + x1 sin  (rope_kernel's roundings)
                    ov[e] = rot ? f2bf(rbf(xe * c) + rbf((wn == 0 ? -ye : ye) * sv)) : f2bf(xe);
                }
                const int64_t orow = p.cmap(row);
                bf16_t* cp = p.nseg > 0 ? p.seg_dst[si] + orow * p.seg_ld[si] + (rcol - p.seg_begin[si])
                                        : reinterpret_cast<bf16_t*>(p.C) + cz + orow * p.ldc + rcol;
                *reinterpret_cast<bf16x8*>(cp) = ov;
            }
            return;
        }
        if constexpr (MT != 2) if (pair) {
            // GeGLU in registers: slab columns [0, 32) = gate, [32, 64) = up of the wave's 32 output columns n0/2 + wn*32 + ..;
            // each lane takes 8 of them for one row (16 rows per pass).  Rounding points of act 2: g = bf16(acc), u = bf16(acc),
            // h = bf16(bf16(gelu_tanh(g)) * u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(i * 16 + 4 * g + r) * 64 + j * 16 + l15] = acc[h * 4 + i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int ocol = (n0 >> 1) + wn * 32 + (lane & 3) * 8;
            const bool ocol_ok = ocol < (p.N >> 1);
            bf16_t* cb = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int lr = it * 16 + (lane >> 2);
                const int row = row_base(h) + lr;
                const float* sp = slab + lr * 64 + (lane & 3) * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(sp), g1 = *reinterpret_cast<const f32x4*>(sp + 4);
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(sp + 32), u1 = *reinterpret_cast<const f32x4*>(sp + 36);
                if (row >= p.M || !ocol_ok) continue;
                const float gv[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                const float uv[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
                bf16x8 gb, ub, hb;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 gr = rbf2(f32x2{gv[e], gv[e + 1]}), ur = rbf2(f32x2{uv[e], uv[e + 1]});
                    const f32x2 hv = rbf2(gelu_tanh2(gr)) * ur;
                    gb[e] = f2bf(gr[0]);
                    gb[e + 1] = f2bf(gr[1]);
                    ub[e] = f2bf(ur[0]);
                    ub[e + 1] = f2bf(ur[1]);
                    hb[e] = f2bf(hv[0]);
                    hb[e + 1] = f2bf(hv[1]);
                }
                const int64_t o = cz + p.cmap(row) * p.ldc + ocol;
                *reinterpret_cast<bf16x8*>(cb + o) = hb;
                if (p.pre_out != nullptr) store_pre8(p.pre_out + o, gb, p.nt_pre);
                if (p.pre_out2 != nullptr) store_pre8(p.pre_out2 + o, ub, p.nt_pre);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            return;
        }
        if constexpr (MT != 2) if (fused_fast) {
            bf16x8 s0[8], s1[8];
            float ds[8];
            const int rbase = row_base(h) + (lane >> 3);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + it * 8;
                s0[it] = bf16x8{};
                s1[it] = bf16x8{};
                ds[it] = 0.f;
                if (row < p.M && col_ok) {
                    const int64_t o = cz + p.cmap(row) * p.ldc + ccol;
                    s0[it] = *reinterpret_cast<const bf16x8*>(p.aux1 + o);
                    if (p.act == 3) s1[it] = *reinterpret_cast<const bf16x8*>(p.aux2 + o);
                    if (p.act == 4) ds[it] = p.rowvec[vz + (int64_t)row * p.rv_ld];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(i * 16 + 4 * g + r) * 64 + j * 16 + l15] = acc[h * 4 + i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            bf16_t* cb = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + it * 8;
                const float* sp = slab + (it * 8 + (lane >> 3)) * 64 + (lane & 7) * 8;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (row >= p.M || !col_ok) continue;
                const int64_t o = cz + p.cmap(row) * p.ldc + ccol;
                bf16x8 ov;
                if (p.act == 2) {  // same order and rounding points as epilogue8
                    if (p.pre_out != nullptr) {
                        bf16x8 uv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) uv[e] = f2bf(v[e]);
                        *reinterpret_cast<bf16x8*>(p.pre_out + o) = uv;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 r = rbf2(gelu_tanh2(f32x2{bf2f(s0[it][e]), bf2f(s0[it][e + 1])})) * rbf2(f32x2{v[e], v[e + 1]});
                        ov[e] = f2bf(r[0]);
                        ov[e + 1] = f2bf(r[1]);
                    }
                } else if (p.act == 3) {
                    bf16x8 du;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 dh = rbf2(f32x2{v[e], v[e + 1]});
                        f32x2 gl, gr;
                        gelu_tanh_both2(f32x2{bf2f(s0[it][e]), bf2f(s0[it][e + 1])}, gl, gr);
                        const f32x2 a = dh * rbf2(gl);
                        const f32x2 b = rbf2(dh * f32x2{bf2f(s1[it][e]), bf2f(s1[it][e + 1])}) * gr;
                        du[e] = f2bf(a[0]);
                        du[e + 1] = f2bf(a[1]);
                        ov[e] = f2bf(b[0]);
                        ov[e + 1] = f2bf(b[1]);
                    }
                    *reinterpret_cast<bf16x8*>(p.pre_out + o) = du;
                } else if (p.act == 5) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2 gl, gr;
                        gelu_tanh_both2(f32x2{bf2f(s0[it][e]), bf2f(s0[it][e + 1])}, gl, gr);
                        const f32x2 b = rbf2(f32x2{v[e], v[e + 1]}) * gr;
                        ov[e] = f2bf(b[0]);
                        ov[e + 1] = f2bf(b[1]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = f2bf((bf2f(s0[it][e]) * (v[e] - ds[it])) * p.scale);
                }
                *reinterpret_cast<bf16x8*>(cb + o) = ov;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            return;
        }
        if constexpr (MT <= 4) if (plain_fast) {  // (128x128 configurations: in the 256x256 kernels the operand registers went to scratch)
            // bias / GELU / residual epilogue with its operands requested up front: the bias once (a lane's 8 columns are the same for
            // all of its rows), the 8 residual rows of the half before the accumulators go to the slab.  In the rolled loop below every
            // iteration waited out a bias and a residual round trip of its own — most of the fixed cost of the small (B = 1) GEMMs.
            float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr && col_ok) {
                if (p.bias_f32) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.bias) + ccol);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.bias) + ccol + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                } else {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.bias) + ccol);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = bf2f(b[e]);
                }
            }
            bf16x8 rs[AH * 2];
            const int rbase = row_base(h) + (lane >> 3);
#pragma unroll
            for (int it = 0; it < AH * 2; ++it) {
                const int row = rbase + it * 8;
                rs[it] = bf16x8{};
                if (p.residual != nullptr && row < p.M && col_ok) rs[it] = *reinterpret_cast<const bf16x8*>(p.residual + rz + p.cmap(row) * p.ldr + ccol);
            }
#pragma unroll
            for (int i = 0; i < AH; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(i * 16 + 4 * g + r) * 64 + j * 16 + l15] = acc[h * AH + i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            bf16_t* cb = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
            for (int it = 0; it < AH * 2; ++it) {
                const int row = rbase + it * 8;
                const float* sp = slab + (it * 8 + (lane >> 3)) * 64 + (lane & 7) * 8;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                if (row >= p.M || !col_ok) continue;
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const int64_t o = cz + p.cmap(row) * p.ldc + ccol;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + bv[e]);  // same order and rounding points as epilogue8
                if (p.scale != 1.0f) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] * p.scale);
                }
                if (p.act == 1) {
                    if (p.pre_out != nullptr) {
                        bf16x8 pv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pv[e] = f2bf(v[e]);
                        store_pre8(p.pre_out + o, pv, p.nt_pre);
                    }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 a = gelu_tanh2(f32x2{v[e], v[e + 1]});
                        v[e] = rbf(a[0]);
                        v[e + 1] = rbf(a[1]);
                    }
                }
                if (p.residual != nullptr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + bf2f(rs[it][e]));
                }
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = f2bf(v[e]);
                *reinterpret_cast<bf16x8*>(cb + o) = ov;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            return;
        }
        if constexpr (MT == 8) if (simple_fast) {
            // (every argument used inside the unrolled loops is read ONCE here: hipcc reads the argument block from constant memory only
            // while it has fewer than 300 uses in a kernel, beyond that it copies the whole block to scratch)
            const bool has_bias = p.bias != nullptr, has_res = p.residual != nullptr, gelu = p.act == 1;
            const int M_ = p.M, nt_pre_ = p.nt_pre, nt_c_ = p.nt_c;
            const bf16_t* res_ = p.residual;
            bf16_t* pre_ = p.pre_out;
            const int64_t ldr_ = p.ldr, ldc_ = p.ldc;
            float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (has_bias && col_ok) {  // a lane's 8 columns are the same for all of its rows
                if (p.bias_f32) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.bias) + ccol);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.bias) + ccol + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                } else {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.bias) + ccol);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = bf2f(b[e]);
                }
            }
            // destination of the lane's 8 columns (column routing resolved once: the columns do not change with the row)
            bf16_t* cl = reinterpret_cast<bf16_t*>(p.C) + cz + ccol;
            int64_t ldl = p.ldc;
            if (p.nseg > 0) {  // (constant indices only: a run-time index would put the whole argument block into scratch)
                cl = p.seg_dst[0] + (ccol - p.seg_begin[0]);
                ldl = p.seg_ld[0];
                if (p.nseg > 1 && ccol >= p.seg_begin[1]) {
                    cl = p.seg_dst[1] + (ccol - p.seg_begin[1]);
                    ldl = p.seg_ld[1];
                }
                if (p.nseg > 2 && ccol >= p.seg_begin[2]) {
                    cl = p.seg_dst[2] + (ccol - p.seg_begin[2]);
                    ldl = p.seg_ld[2];
                }
            }
            const int rbase = row_base(h) + (lane >> 3);
            auto res_row = [&](int it) -> bf16x8 {
                const int row = rbase + it * 8;
                return (has_res && row < M_ && col_ok) ? *reinterpret_cast<const bf16x8*>(res_ + rz + (int64_t)row * ldr_ + ccol) : bf16x8{};
            };
            const bool accum = p.accumulate != 0;
            auto old_row = [&](int it) -> bf16x8 {  // accumulate: what the destination holds
                const int row = rbase + it * 8;
                return (accum && row < M_ && col_ok) ? *reinterpret_cast<const bf16x8*>(cl + (int64_t)row * ldl) : bf16x8{};
            };
            bf16x8 rnext = res_row(0), onext = old_row(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(i * 16 + 4 * g + r) * 64 + j * 16 + l15] = acc[h * 4 + i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + it * 8;
                const float* sp = slab + (it * 8 + (lane >> 3)) * 64 + (lane & 7) * 8;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                const bf16x8 rv = rnext, ov0 = onext;
                if (it + 1 < 8) {  // (one row ahead: the latency hides behind this row's arithmetic)
                    rnext = res_row(it + 1);
                    onext = old_row(it + 1);
                }
                if (row >= M_ || !col_ok) continue;
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (has_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = rbf(v[e]);
                if (gelu) {
                    if (pre_ != nullptr) {
                        bf16x8 pv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pv[e] = f2bf(v[e]);
                        store_pre8(pre_ + cz + (int64_t)row * ldc_ + ccol, pv, nt_pre_);
                    }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 a = gelu_tanh2(f32x2{v[e], v[e + 1]});
                        v[e] = rbf(a[0]);
                        v[e + 1] = rbf(a[1]);
                    }
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + bf2f(rv[e]));
                }
                if (accum) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf2f(ov0[e]);
                }
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = f2bf(v[e]);
                store_pre8(cl + (int64_t)row * ldl, ov, nt_c_);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            return;
        }
#pragma unroll
        for (int i = 0; i < AH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(i * 16 + 4 * g + r) * 64 + j * 16 + l15] = acc[h * AH + i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int it = 0; it < AH * 2; ++it) {
            const int lr = it * 8 + (lane >> 3);
            const int row = row_base(h) + lr;
            if (row >= p.M || !col_ok) continue;
            const float* sp = slab + lr * 64 + (lane & 7) * 8;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.split_k > 1) {  // raw f32 partial tile into the workspace; the fused epilogue runs in the reduction
                float* wp = reinterpret_cast<float*>(p.ws) + ((int64_t)blockIdx.y * p.M + row) * p.N + ccol;
                *reinterpret_cast<f32x4*>(wp) = v0;
                *reinterpret_cast<f32x4*>(wp + 4) = v1;
            } else {
                epilogue8(p, v, row, ccol, cz, rz, vz, p.C, p.out_f32 != 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    epi_half(std::integral_constant<int, 0>{});
    if constexpr (MT / 4 > 1) epi_half(std::integral_constant<int, 1>{});
    if constexpr (MT / 4 > 2) epi_half(std::integral_constant<int, 2>{});
    static_assert(MT / 4 <= 3, "extend the epilogue parts");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent NT kernel with a DYNAMIC tile queue (round 4): the quadrant schedule above, specialised for K-contiguous
// operands, one batch entry, no split-K, with one resident block per CU that draws 256 x 256 tiles from per-XCD ticket counters until
// none are left.  What it buys over one block per tile:
//   * the next tile's first six half-tiles are staged (LDS-DMA) BEFORE the current tile's epilogue runs, so the K loop restarts with
//     its pipeline full and the epilogue's operand loads / arithmetic / stores overlap with that traffic (the epilogue works out of a
//     32 KiB slab region behind the stage buffers, in 16-row passes, instead of aliasing them);
//   * tiles are handed out by atomic tickets, so a CU whose tile arrived cold or whose epilogue ran long simply draws later — the
//     static assignment of round 2's experiment gave that balance up and lost inside the training step;
//   * the ticket order is the same XCD-grouped raster as the plain launches': counter x hands out XCD x's contiguous run of the
//     swizzled order to the blocks that (by the dispatcher's observed placement, blockIdx % 8) sit on XCD x; a block whose XCD's run
//     is exhausted steals from the next XCD's.  Placement only affects speed, never which tile a ticket means.
// Tickets are drawn two tiles ahead (the one needed next must be known before the current epilogue starts).  The nine counters of a
// launch (8 queues + blocks done) live in a slot of g_ps_ctr; the last block to leave zeroes the slot, so no memset node is needed and
// a captured graph replays correctly.
constexpr int PS_SLOTS = 1024;
__device__ unsigned int g_ps_ctr[PS_SLOTS][16];

// diagnostics (KAI0_HIPCC_FLAGS=-DKAI0_PS_TRACE, tools/probes/persistent_phases.py): per block of the persistent kernel, 100 MHz ticks
// spent in  0 tile start  1 K loop  2 hand-over issue + epilogue passes  3 ticket / drain / barriers  and 4 = tiles run
#ifdef KAI0_PS_TRACE
__device__ long long g_ps_trace[256][8];
#define PS_STAMP(i)                                              \
    do {                                                         \
        const long long n_ = __builtin_amdgcn_s_memrealtime();   \
        t_acc[i] += n_ - t_prev;                                 \
        t_prev = n_;                                             \
    } while (0)
#else
#define PS_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(512, 1) void gemm_nt_persistent_kernel(const GemmArgs p, unsigned int* __restrict__ ctr) {
    constexpr int TBM = 256, TBN = 256, A_TILE = TBM * BK * 2, STAGE = 2 * A_TILE, GROUP = 4, MT = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wm;
    const int l15 = lane & 15, g = lane >> 4;
    int* mbox = reinterpret_cast<int*>(smem + 2 * STAGE);            // two ints at the start of wave 0's slab (free between epilogues)
    float* slab = reinterpret_cast<float*>(smem + 2 * STAGE + wave * 4096);  // wave-private f32 [16][64]
    const int ntile = p.tiles_m * p.tiles_n;
    const int qq = ntile >> 3, rr = ntile & 7;
    // one ticket = one tile id in the swizzled order, or -1 when every queue is empty (thread 0 only).  draw() issues the atomic on
    // the block's own XCD queue and returns its raw result — consumed much later by resolve(), so its round trip hides behind the
    // epilogue; resolve() turns it into a tile id and only when the own queue is exhausted walks the other XCDs' queues.
    const int x0 = blockIdx.x & 7;
    auto draw = [&]() -> unsigned { return __hip_atomic_fetch_add(ctr + x0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto resolve = [&](unsigned t) -> int {
        for (int a = 0; a < 8; ++a) {
            const int x = (x0 + a) & 7;
            const int len = qq + (x < rr ? 1 : 0);
            if (a > 0) t = __hip_atomic_fetch_add(ctr + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)t < len) return (x < rr ? x * (qq + 1) : rr * (qq + 1) + (x - rr) * qq) + (int)t;
        }
        return -1;
    };
    auto next_ticket = [&]() -> int { return resolve(draw()); };
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)OOB, 0x00020000);
    const bool pair = p.act == 6;
    const __amdgpu_buffer_rsrc_t b2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(pair ? p.B2 : p.B), 0, (int)OOB, 0x00020000);
    const int kc_chunk = ((lane & 7) ^ (lane >> 3)) * 8;
    const int nk = (p.K + BK - 1) / BK;

    // ---- per-tile staging state (see the quadrant schedule in gemm_bf16_kernel: same half-tiles, same order, same counts) ----
    // (identity operand row maps only — the host sends remapped operands to the plain kernel — so a tile's 8 source offsets are one
    // multiply-add each: the 64-bit row-map arithmetic of the general kernel spilled here)
    const uint32_t lda2 = (uint32_t)p.lda * 2, ldb2 = (uint32_t)p.ldb * 2, kc2 = (uint32_t)kc_chunk * 2;
    auto arow_fix = [](int) -> uint32_t { return 0u; };
    int m0 = 0, n0 = 0;
    uint32_t ha_off[4], hb_off[4];
    int ha_lds[4], hb_lds[4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 2 * wave + j, x = h * 2 + j;
            ha_lds[x] = ((q >> 3) * 128 + h * 64 + (q & 7) * 8) * 128;
            hb_lds[x] = ((q >> 2) * 64 + h * 32 + (q & 3) * 8) * 128;
        }
    auto setup = [&](int pid) {
        const int width = GROUP * p.tiles_n;
        const int group = pid / width, first_m = group * GROUP;
        const int gsz = min(p.tiles_m - first_m, GROUP);
        const int in_g = pid - group * width;
        m0 = (first_m + in_g % gsz) * TBM;
        n0 = (in_g / gsz) * TBN;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = 2 * wave + j, x = h * 2 + j;
                const int Ra = m0 + (q >> 3) * 128 + h * 64 + (q & 7) * 8 + (lane >> 3);
                ha_off[x] = Ra < p.M ? (uint32_t)Ra * lda2 + kc2 + arow_fix(Ra) : OOB;
                int Rb = n0 + (q >> 2) * 64 + h * 32 + (q & 3) * 8 + (lane >> 3);
                bool ok = Rb < p.N;
                if (pair) {
                    Rb = (n0 >> 1) + (q >> 2) * 32 + (q & 3) * 8 + (lane >> 3);
                    ok = Rb < (p.N >> 1);
                }
                hb_off[x] = ok ? (uint32_t)Rb * ldb2 + kc2 : OOB;
            }
    };
    auto issue_half = [&](bool isb, int h, int t) {
        const int k0 = t * BK;
        const bool kc_in = (k0 + kc_chunk) < p.K;
        char* base = smem + (t & 1) * STAGE + (isb ? A_TILE : 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int x = h * 2 + j;
            const uint32_t o = isb ? hb_off[x] : ha_off[x];
            const uint32_t off = (kc_in && o != OOB) ? o + (uint32_t)k0 * 2 : OOB;
            glds16(isb ? ((pair && h == 1) ? b2_rsrc : b_rsrc) : a_rsrc, off, base + (isb ? hb_lds[x] : ha_lds[x]));
        }
    };
    auto prologue_issue = [&]() {
        issue_half(false, 0, 0);
        issue_half(true, 1, 0);
        issue_half(true, 0, 0);
        issue_half(false, 1, 0);
        issue_half(false, 0, 1);
        issue_half(true, 1, 1);
    };
    auto frag = [&](const char* tile, int row0, int ks) -> bf16x8 {
        const int row = row0 + l15;
        return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4));
    };

    // ---- first two tickets --------------------------------------------------------------------------------------------------
    if (tid == 0) {
        const int t0 = next_ticket();
        mbox[0] = t0;
        mbox[1] = t0 >= 0 ? next_ticket() : -1;
    }
    __syncthreads();
    int cur = mbox[0], nxt = mbox[1];
    __syncthreads();
    if (cur >= 0) {
        setup(cur);
        prologue_issue();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        lds_barrier();
        if (grp == 1) lds_barrier();  // stagger group 1 by one slot
    }
    const int64_t cz = 0, rz = 0, vz = 0;
#ifdef KAI0_PS_TRACE
    long long t_acc[5] = {0, 0, 0, 0, 0};
    long long t_prev = __builtin_amdgcn_s_memrealtime();
#endif
    while (cur >= 0) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 af[4][2], bq[2][2];
        auto read_a = [&](const char* ta, int h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[i][ks] = frag(ta, wm * 128 + (h * 4 + i) * 16, ks);
        };
        auto read_b = [&](const char* tb, int h) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bq[j][ks] = frag(tb, wn * 64 + (h * 2 + j) * 16, ks);
        };
        auto quad = [&](auto ahc, auto bhc) {
            constexpr int ah = decltype(ahc)::value, bh = decltype(bhc)::value;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][bh * 2 + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bq[j][ks], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        PS_STAMP(0);
        for (int t = 0; t < nk; ++t) {
            const char* ta = smem + (t & 1) * STAGE;
            const char* tb = ta + A_TILE;
            read_b(tb, 0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(ta, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue_half(true, 0, t + 1);
            lds_barrier();
            quad(I0{}, I0{});
            lds_barrier();
            read_b(tb, 1);
            __builtin_amdgcn_sched_barrier(0);
            issue_half(false, 1, t + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            lds_barrier();
            quad(I0{}, I1{});
            lds_barrier();
            read_a(ta, 1);
            __builtin_amdgcn_sched_barrier(0);
            issue_half(false, 0, t + 2);
            lds_barrier();
            quad(I1{}, I1{});
            lds_barrier();
            read_b(tb, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue_half(true, 1, t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            lds_barrier();
            quad(I1{}, I0{});
            lds_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing zero-fill pieces have landed: the stage buffers are free
        if (grp == 0) lds_barrier();                      // re-align the two groups
        lds_barrier();

        PS_STAMP(1);
        // ---- hand-over: ticket for the tile after next, next tile's first half-tiles, THEN this tile's epilogue ------------------
        const int m0c = m0, n0c = n0;
        int nn = -1;
        unsigned raw = 0;
        if (tid == 0 && nxt >= 0) raw = draw();
        if (nxt >= 0) {
            setup(nxt);
            prologue_issue();
        }
        const int ccol = n0c + wn * 64 + (lane & 7) * 8;
        const bool col_ok = ccol < ((p.N + 7) & ~7);
        const bool fused_fast = p.act >= 2 && p.act <= 5 && p.bias == nullptr && p.gate == nullptr && p.residual == nullptr && !p.accumulate &&
                                !p.out_f32 && p.nseg == 0 && (p.act == 4 || p.scale == 1.0f) && (p.act != 3 || p.pre_out != nullptr);
        // (the same predicate as the tile kernel's; split_k == 1 is also what the host requires of a persistent launch)
        const bool simple_fast = p.simple_epi && p.act <= 1 && p.split_k == 1 && p.gate == nullptr && !p.out_f32 && p.scale == 1.0f && p.cmap.rpb == 0 &&
                                 (N_ALIGNED8(p.N));
        auto to_slab = [&](int ti) {  // MFMA row-tile ti of the wave's sub-tile (16 rows x 64 columns) -> slab
            __builtin_amdgcn_wave_barrier();
            // (static indexing of acc: callers pass compile-time ti through the unrolled loops below)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(4 * g + r) * 64 + j * 16 + l15] = acc[ti][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        if (pair) {
            const int ocol = (n0c >> 1) + wn * 32 + (lane & 3) * 8;
            const bool ocol_ok = ocol < (p.N >> 1);
            bf16_t* cb = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
            for (int ti = 0; ti < MT; ++ti) {
                to_slab(ti);
                const int row = m0c + wm * 128 + ti * 16 + (lane >> 2);
                const float* sp = slab + (lane >> 2) * 64 + (lane & 3) * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(sp), g1 = *reinterpret_cast<const f32x4*>(sp + 4);
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(sp + 32), u1 = *reinterpret_cast<const f32x4*>(sp + 36);
                if (row < p.M && ocol_ok) {
                    const float gv[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                    const float uv[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
                    bf16x8 gb, ub, hb;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 gr = rbf2(f32x2{gv[e], gv[e + 1]}), ur = rbf2(f32x2{uv[e], uv[e + 1]});
                        const f32x2 hv = rbf2(gelu_tanh2(gr)) * ur;
                        gb[e] = f2bf(gr[0]); gb[e + 1] = f2bf(gr[1]);
                        ub[e] = f2bf(ur[0]); ub[e + 1] = f2bf(ur[1]);
                        hb[e] = f2bf(hv[0]); hb[e + 1] = f2bf(hv[1]);
                    }
                    const int64_t o = p.cmap(row) * p.ldc + ocol;
                    *reinterpret_cast<bf16x8*>(cb + o) = hb;
                    if (p.pre_out != nullptr) store_pre8(p.pre_out + o, gb, p.nt_pre);
                    if (p.pre_out2 != nullptr) store_pre8(p.pre_out2 + o, ub, p.nt_pre);
                }
            }
        } else if (fused_fast) {
            bf16_t* cb = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // the extra [M][N] operands of this 64-row half requested before its accumulators go through the slab
                bf16x8 s0[8], s1[8];
                float ds[8];
                const int rbase = m0c + wm * 128 + h * 64 + (lane >> 3);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = rbase + it * 8;
                    s0[it] = bf16x8{};
                    s1[it] = bf16x8{};
                    ds[it] = 0.f;
                    if (row < p.M && col_ok) {
                        const int64_t o = p.cmap(row) * p.ldc + ccol;
                        s0[it] = *reinterpret_cast<const bf16x8*>(p.aux1 + o);
                        if (p.act == 3) s1[it] = *reinterpret_cast<const bf16x8*>(p.aux2 + o);
                        if (p.act == 4) ds[it] = p.rowvec[(int64_t)row * p.rv_ld];
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    to_slab(h * 4 + i);
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2) {
                        const int it = 2 * i + it2;
                        const int row = rbase + it * 8;
                        const float* sp = slab + (it2 * 8 + (lane >> 3)) * 64 + (lane & 7) * 8;
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        if (row >= p.M || !col_ok) continue;
                        const int64_t o = p.cmap(row) * p.ldc + ccol;
                        bf16x8 ov;
                        if (p.act == 2) {
                            if (p.pre_out != nullptr) {
                                bf16x8 uv;
#pragma unroll
                                for (int e = 0; e < 8; ++e) uv[e] = f2bf(v[e]);
                                *reinterpret_cast<bf16x8*>(p.pre_out + o) = uv;
                            }
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                const f32x2 r = rbf2(gelu_tanh2(f32x2{bf2f(s0[it][e]), bf2f(s0[it][e + 1])})) * rbf2(f32x2{v[e], v[e + 1]});
                                ov[e] = f2bf(r[0]);
                                ov[e + 1] = f2bf(r[1]);
                            }
                        } else if (p.act == 3) {
                            bf16x8 du;
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                const f32x2 dh = rbf2(f32x2{v[e], v[e + 1]});
                                f32x2 gl, gr;
                                gelu_tanh_both2(f32x2{bf2f(s0[it][e]), bf2f(s0[it][e + 1])}, gl, gr);
                                const f32x2 a = dh * rbf2(gl);
                                const f32x2 b = rbf2(dh * f32x2{bf2f(s1[it][e]), bf2f(s1[it][e + 1])}) * gr;
                                du[e] = f2bf(a[0]); du[e + 1] = f2bf(a[1]);
                                ov[e] = f2bf(b[0]); ov[e + 1] = f2bf(b[1]);
                            }
                            *reinterpret_cast<bf16x8*>(p.pre_out + o) = du;
                        } else if (p.act == 5) {
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                f32x2 gl, gr;
                                gelu_tanh_both2(f32x2{bf2f(s0[it][e]), bf2f(s0[it][e + 1])}, gl, gr);
                                const f32x2 b = rbf2(f32x2{v[e], v[e + 1]}) * gr;
                                ov[e] = f2bf(b[0]); ov[e + 1] = f2bf(b[1]);
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) ov[e] = f2bf((bf2f(s0[it][e]) * (v[e] - ds[it])) * p.scale);
                        }
                        *reinterpret_cast<bf16x8*>(cb + o) = ov;
                    }
                }
            }
        } else if (simple_fast) {
            // store-with-little-else (act 0 / 1, optional bias / residual / column routing): see gemm_bf16_kernel's simple_fast
            const bool has_bias = p.bias != nullptr, has_res = p.residual != nullptr, gelu = p.act == 1;
            const int M_ = p.M, nt_pre_ = p.nt_pre, nt_c_ = p.nt_c;  // (read once: see gemm_bf16_kernel)
            const bf16_t* res_ = p.residual;
            bf16_t* pre_ = p.pre_out;
            const int64_t ldr_ = p.ldr, ldc_ = p.ldc;
            float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (has_bias && col_ok) {
                if (p.bias_f32) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.bias) + ccol);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.bias) + ccol + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                } else {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.bias) + ccol);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = bf2f(b[e]);
                }
            }
            bf16_t* cl = reinterpret_cast<bf16_t*>(p.C) + ccol;
            int64_t ldl = p.ldc;
            if (p.nseg > 0) {  // (constant indices only: a run-time index would put the whole argument block into scratch)
                cl = p.seg_dst[0] + (ccol - p.seg_begin[0]);
                ldl = p.seg_ld[0];
                if (p.nseg > 1 && ccol >= p.seg_begin[1]) {
                    cl = p.seg_dst[1] + (ccol - p.seg_begin[1]);
                    ldl = p.seg_ld[1];
                }
                if (p.nseg > 2 && ccol >= p.seg_begin[2]) {
                    cl = p.seg_dst[2] + (ccol - p.seg_begin[2]);
                    ldl = p.seg_ld[2];
                }
            }
            const int rbase = m0c + wm * 128 + (lane >> 3);
            auto res_row = [&](int k) -> bf16x8 {  // k = 16-row pass * 2 + 8-row step
                const int row = rbase + k * 8;
                return (has_res && row < M_ && col_ok) ? *reinterpret_cast<const bf16x8*>(res_ + (int64_t)row * ldr_ + ccol) : bf16x8{};
            };
            const bool accum = p.accumulate != 0;
            auto old_row = [&](int k) -> bf16x8 {
                const int row = rbase + k * 8;
                return (accum && row < M_ && col_ok) ? *reinterpret_cast<const bf16x8*>(cl + (int64_t)row * ldl) : bf16x8{};
            };
            bf16x8 rnext = res_row(0), onext = old_row(0);
#pragma unroll
            for (int ti = 0; ti < MT; ++ti) {
                to_slab(ti);
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2) {
                    const int k = ti * 2 + it2;
                    const int row = rbase + k * 8;
                    const float* sp = slab + (it2 * 8 + (lane >> 3)) * 64 + (lane & 7) * 8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                    const bf16x8 rv = rnext, ov0 = onext;
                    if (k + 1 < 2 * MT) {
                        rnext = res_row(k + 1);
                        onext = old_row(k + 1);
                    }
                    if (row >= M_ || !col_ok) continue;
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if (has_bias) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bv[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rbf(v[e]);
                    if (gelu) {
                        if (pre_ != nullptr) {
                            bf16x8 pv;
#pragma unroll
                            for (int e = 0; e < 8; ++e) pv[e] = f2bf(v[e]);
                            store_pre8(pre_ + (int64_t)row * ldc_ + ccol, pv, nt_pre_);
                        }
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 a = gelu_tanh2(f32x2{v[e], v[e + 1]});
                            v[e] = rbf(a[0]);
                            v[e + 1] = rbf(a[1]);
                        }
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + bf2f(rv[e]));
                    }
                    if (accum) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bf2f(ov0[e]);
                    }
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = f2bf(v[e]);
                    store_pre8(cl + (int64_t)row * ldl, ov, nt_c_);
                }
            }
        } else {
#pragma unroll
            for (int ti = 0; ti < MT; ++ti) {
                to_slab(ti);
#pragma unroll 1
                for (int it2 = 0; it2 < 2; ++it2) {
                    const int lr = it2 * 8 + (lane >> 3);
                    const int row = m0c + wm * 128 + ti * 16 + lr;
                    if (row >= p.M || !col_ok) continue;
                    const float* sp = slab + lr * 64 + (lane & 7) * 8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    epilogue8(p, v, row, ccol, cz, rz, vz, p.C, p.out_f32 != 0);
                }
            }
        }
        PS_STAMP(2);
        // ---- next tile ------------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) mbox[0] = nxt >= 0 ? resolve(raw) : -1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next tile's first half-tiles (and the epilogue's own traffic)
        lds_barrier();
        nn = mbox[0];
        cur = nxt;
        nxt = nn;
        lds_barrier();  // everyone has read the mailbox before wave 0's slab is written again
        if (cur >= 0 && grp == 1) lds_barrier();  // stagger group 1 by one slot
        PS_STAMP(3);
#ifdef KAI0_PS_TRACE
        t_acc[4] += 1;
#endif
    }
#ifdef KAI0_PS_TRACE
    if (tid == 0 && blockIdx.x < 256)
        for (int i = 0; i < 5; ++i) g_ps_trace[blockIdx.x][i] = t_acc[i];
#endif
    // ---- leave: the last block zeroes the launch's counters (the slot is then reusable by a later launch / a graph replay) ---------
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned done = __hip_atomic_fetch_add(ctr + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(ctr + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// split-K reduction: sum the f32 partial tiles of a (batch entry, 8-column group) and run the fused epilogue once.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const int z = blockIdx.y;
    const int z1 = z / p.batch_inner, z2 = z - z1 * p.batch_inner;
    const int n8 = (p.N + 7) >> 3;
    const int64_t total = (int64_t)p.M * n8;
    const float* w0 = p.ws + (int64_t)z * p.split_k * p.M * p.N;
    const int64_t cz = z1 * p.sC1 + z2 * p.sC2, rz = z1 * p.sR1 + z2 * p.sR2, vz = z1 * p.rv_s1 + z2 * p.rv_s2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / n8);
        const int col = (int)(i - (int64_t)row * n8) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bf16x8 rpre = {};
        const bool has_res = p.residual != nullptr && p.act != 4;
        if (has_res) rpre = *reinterpret_cast<const bf16x8*>(p.residual + rz + p.cmap(row) * p.ldr + col);  // with the partials
        if (p.split_k <= 8) {
            // all partial tiles requested before the first add (clamped slice index instead of a loop bound: nothing conditional
            // around the loads); summed in slice order as below
            f32x4 a[8], b[8];
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                const float* wp = w0 + ((int64_t)min(sp, p.split_k - 1) * p.M + row) * p.N + col;
                a[sp] = *reinterpret_cast<const f32x4*>(wp);
                b[sp] = *reinterpret_cast<const f32x4*>(wp + 4);
            }
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                const float m = sp < p.split_k ? 1.0f : 0.0f;  // (x * 1 is exact; slices past the end contribute + 0)
                acc[0] += m * a[sp][0]; acc[1] += m * a[sp][1]; acc[2] += m * a[sp][2]; acc[3] += m * a[sp][3];
                acc[4] += m * b[sp][0]; acc[5] += m * b[sp][1]; acc[6] += m * b[sp][2]; acc[7] += m * b[sp][3];
            }
        } else
        for (int sp = 0; sp < p.split_k; ++sp) {
            const float* wp = w0 + ((int64_t)sp * p.M + row) * p.N + col;
            f32x4 a = *reinterpret_cast<const f32x4*>(wp);
            f32x4 b = *reinterpret_cast<const f32x4*>(wp + 4);
            acc[0] += a[0]; acc[1] += a[1]; acc[2] += a[2]; acc[3] += a[3];
            acc[4] += b[0]; acc[5] += b[1]; acc[6] += b[2]; acc[7] += b[3];
        }
        epilogue8(p, acc, row, col, cz, rz, vz, p.C, p.out_f32 != 0, has_res ? &rpre : nullptr);
    }
}

// split-K reduction + the norm that reads its output (B = 1 inference: SigLIP out_proj -> layer_norm2, fc2 -> the next layer's
// layer_norm1, Gemma down_proj -> the next layer's input RMSNorm): one block per output row (N <= 2048: one 8-column group per
// thread), partials summed and the epilogue run exactly as in splitk_reduce_kernel (x stored), then the row statistics over the
// bf16 values just stored and the normalised row.  KIND 1: y = bf16((x * rstd) * (1 + w)), w f32 (GemmaRMSNorm, modeling_gemma.py:
// 49-104 without cond); KIND 2: y = bf16((x - mean) * rstd * w + b) (nn.LayerNorm, modeling_siglip.py).  Saves the norm's launch and
// its read of x: at B = 1 a launch is worth 4-5 us and there are ~70 of these per action chunk.
template <int KIND>
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(const GemmArgs p) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int n8 = p.N >> 3;
    const bool live = (int)threadIdx.x < n8;
    const int col = min((int)threadIdx.x, n8 - 1) * 8;  // idle threads repeat the last group's loads and store nothing
    // every load of the thread is requested up front: norm weights, residual, partials
    float wv[8], bv[8];
    if constexpr (KIND == 1) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.norm_w) + col);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.norm_w) + col + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wv[e] = w0[e]; wv[4 + e] = w1[e]; bv[e] = bv[4 + e] = 0.f; }
    } else {
        const bf16x8 w8 = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.norm_w) + col);
        const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(p.norm_b + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) { wv[e] = bf2f(w8[e]); bv[e] = bf2f(b8[e]); }
    }
    bf16x8 rpre = {};
    const bool has_res = p.residual != nullptr;
    if (has_res) rpre = *reinterpret_cast<const bf16x8*>(p.residual + p.cmap(row) * p.ldr + col);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.split_k <= 8) {
        f32x4 a[8], b[8];
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            const float* wp = p.ws + ((int64_t)min(sp, p.split_k - 1) * p.M + row) * p.N + col;
            a[sp] = *reinterpret_cast<const f32x4*>(wp);
            b[sp] = *reinterpret_cast<const f32x4*>(wp + 4);
        }
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            const float m = sp < p.split_k ? 1.0f : 0.0f;
            acc[0] += m * a[sp][0]; acc[1] += m * a[sp][1]; acc[2] += m * a[sp][2]; acc[3] += m * a[sp][3];
            acc[4] += m * b[sp][0]; acc[5] += m * b[sp][1]; acc[6] += m * b[sp][2]; acc[7] += m * b[sp][3];
        }
    } else {
        for (int sp = 0; sp < p.split_k; ++sp) {
            const float* wp = p.ws + ((int64_t)sp * p.M + row) * p.N + col;
            const f32x4 a = *reinterpret_cast<const f32x4*>(wp), b = *reinterpret_cast<const f32x4*>(wp + 4);
            acc[0] += a[0]; acc[1] += a[1]; acc[2] += a[2]; acc[3] += a[3];
            acc[4] += b[0]; acc[5] += b[1]; acc[6] += b[2]; acc[7] += b[3];
        }
    }
    if (live) epilogue8(p, acc, row, col, 0, 0, 0, p.C, false, has_res ? &rpre : nullptr);  // stores x; acc = its f32 pre-image
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = live ? rbf(acc[e]) : 0.f;
    float y[8];
    if constexpr (KIND == 1) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
        const float rstd = rsqrtf(block_sum<4>(ss, red) / (float)p.N + p.norm_eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (x[e] * rstd) * (1.0f + wv[e]);
    } else {
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += x[e];
        const float mean = block_sum<4>(sm, red) / (float)p.N;
        float vs = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = x[e] - mean;
            vs += live ? d * d : 0.f;
        }
        const float rstd = rsqrtf(block_sum<4>(vs, red) / (float)p.N + p.norm_eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (x[e] - mean) * rstd * wv[e] + bv[e];
    }
    if (live) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(y[e]);
        *reinterpret_cast<bf16x8*>(p.norm_out + (int64_t)row * p.N + col) = o;
    }
}

template <int WM, int WN, int MT, int NT, bool PP, int NS = 2, int BKT = 64, int SCH = 0>
int launch_cfg(const kai0_gemm_desc* d, GemmArgs& p, int batch, hipStream_t s) {
    constexpr int TBM = WM * MT * 16, TBN = WN * NT * 16;
    constexpr int LDS = NS * (TBM + TBN) * BKT * 2;
    p.tiles_m = (d->M + TBM - 1) / TBM;
    p.tiles_n = (p.N + TBN - 1) / TBN;
    dim3 grid(p.tiles_m * p.tiles_n, batch * p.split_k, 1), block(WM * WN * 64, 1, 1);
#define KAI0_LAUNCH(AK, BK_)                                                                                      \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        auto kern = gemm_bf16_kernel<AK, BK_, WM, WN, MT, NT, PP, NS, BKT, SCH>;                       \
        if (!attr_set) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
            if (e != hipSuccess) {                                                                                \
                kai0_set_error("kai0_gemm_bf16: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));      \
                return -2;                                                                                        \
            }                                                                                                     \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, grid, block, LDS, s, p);                                                         \
    } while (0)
    constexpr bool MC_A_OK = (64 % (TBM * 2 / 16)) == 0;  // contraction-strided A tiles need whole k-rows per DMA piece
    if (d->a_kc && d->b_kc) KAI0_LAUNCH(true, true);
    else if (d->a_kc && !d->b_kc) KAI0_LAUNCH(true, false);
    else if constexpr (MC_A_OK) {
        if (!d->a_kc && d->b_kc) KAI0_LAUNCH(false, true);
        else KAI0_LAUNCH(false, false);
    } else {
        kai0_set_error("kai0_gemm_bf16: this tile configuration needs a K-contiguous A operand");
        return -3;
    }
#undef KAI0_LAUNCH
    return 0;
}

}  // namespace

#ifdef KAI0_PS_TRACE
KAI0_API int kai0_debug_ps_trace(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ps_trace), sizeof(long long) * 256 * 8);
}
#endif

KAI0_API int kai0_gemm_bf16(const kai0_gemm_desc* d, kai0_stream_t stream) {
    KAI0_REQUIRE(d != nullptr, "kai0_gemm_bf16: null descriptor");
    KAI0_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "kai0_gemm_bf16: empty problem M=%d N=%d K=%d", d->M, d->N,
                 d->K);
    KAI0_REQUIRE(d->A && d->B && d->C, "kai0_gemm_bf16: null operand");
    KAI0_REQUIRE((d->lda % 8) == 0 && (d->ldb % 8) == 0 && (d->ldc % 8) == 0,
                 "kai0_gemm_bf16: leading dims must be multiples of 8 (lda=%lld ldb=%lld ldc=%lld)",
                 (long long)d->lda, (long long)d->ldb, (long long)d->ldc);
    KAI0_REQUIRE((d->N % 8) == 0 || (d->b_kc && d->ldc >= ((d->N + 7) & ~7) && !d->bias && !d->gate && !d->residual &&
                                     !d->pre_out && !d->accumulate),
                 "kai0_gemm_bf16: N=%d not a multiple of 8 needs b_kc, padded C rows and a plain epilogue", d->N);
    KAI0_REQUIRE((d->K % 8) == 0 || (!d->a_kc && !d->b_kc), "kai0_gemm_bf16: K=%d must be a multiple of 8 for "
                 "K-contiguous operands", d->K);
    KAI0_REQUIRE(d->a_kc || (d->M % 8) == 0, "kai0_gemm_bf16: M=%d must be a multiple of 8 when A is [K][M]",
                 d->M);
    KAI0_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0 && ((uintptr_t)d->C % 16) == 0,
                 "kai0_gemm_bf16: operands must be 16-byte aligned");
    // batch strides: signed element counts, may be negative or span two allocations (kai0hip.h); every entry must stay 16-B aligned
    KAI0_REQUIRE(d->batch <= 1 || (((d->sA1 | d->sA2 | d->sB1 | d->sB2) & 7) == 0 && ((d->sC1 | d->sC2) & (d->out_f32 ? 3 : 7)) == 0),
                 "kai0_gemm_bf16: batch strides must be multiples of 8 elements (16 B)");
    KAI0_REQUIRE(d->gate == nullptr || d->gate_rpb > 0, "kai0_gemm_bf16: gate needs gate_rpb > 0");
    KAI0_REQUIRE(d->act >= 0 && d->act <= 7, "kai0_gemm_bf16: unknown act %d", d->act);
    KAI0_REQUIRE(d->act != 7 || (d->rope_cos && d->rope_sin && d->rope_half == 128 && d->rope_n_end > 0 && (d->rope_n_end % 256) == 0 &&
                                 d->rope_n_end <= d->N && (d->N % 128) == 0 && d->a_kc && d->b_kc && d->batch <= 1 && d->split_k <= 1 &&
                                 !d->out_f32 && !d->accumulate && !d->bias && !d->gate && !d->residual && !d->pre_out &&
                                 (d->scale == 0.0f || d->scale == 1.0f) && d->b_rpb == 0 && ((uintptr_t)d->rope_cos % 16) == 0 &&
                                 ((uintptr_t)d->rope_sin % 16) == 0),
                 "kai0_gemm_bf16: act=7 (RoPE epilogue) needs bf16 cos / sin tables [M][128], rope_n_end %% 256 == 0, N %% 128 == 0, K-contiguous "
                 "operands, one batch entry, no split-K, a plain bf16 output");
    KAI0_REQUIRE(d->act != 6 || (d->B2 && d->a_kc && d->b_kc && d->batch <= 1 && d->split_k <= 1 && !d->out_f32 && !d->accumulate &&
                                 d->nseg == 0 && !d->bias && !d->gate && !d->residual && (d->scale == 0.0f || d->scale == 1.0f) &&
                                 (d->N % 32) == 0 && d->b_rpb == 0 && ((uintptr_t)d->B2 % 16) == 0),
                 "kai0_gemm_bf16: act=6 (GeGLU pair) needs B2 = up weight, K-contiguous operands, N %% 32 == 0, one batch entry, a "
                 "plain bf16 output");
    KAI0_REQUIRE(d->norm_kind == 0 ||
                     ((d->norm_kind == 1 || d->norm_kind == 2) && d->split_k > 1 && d->norm_out && d->norm_w && (d->norm_kind == 1 || d->norm_b) &&
                      d->batch <= 1 && !d->out_f32 && d->nseg == 0 && (d->N % 8) == 0 && d->N <= 2048 && d->act == 0 && !d->gate &&
                      !d->rowvec && ((uintptr_t)d->norm_out % 16) == 0 && ((uintptr_t)d->norm_w % 16) == 0),
                 "kai0_gemm_bf16: a fused norm (norm_kind=%d) needs split_k > 1, norm_out / norm_w (/ norm_b), one batch entry, a plain "
                 "bf16 output with N %% 8 == 0, N <= 2048, no activation / gate", d->norm_kind);
    KAI0_REQUIRE(d->act != 5 || (d->aux1 && !d->out_f32 && !d->accumulate && d->nseg == 0 && (d->N % 8) == 0),
                 "kai0_gemm_bf16: act=5 (fused GELU backward) needs aux1 = pre-activation, plain bf16 output");
    KAI0_REQUIRE(d->act != 4 || (d->aux1 && d->rowvec && !d->out_f32 && !d->accumulate && d->nseg == 0 && (d->N % 8) == 0),
                 "kai0_gemm_bf16: act=4 (fused softmax backward) needs aux1 = P, rowvec = D, plain bf16 output");
    KAI0_REQUIRE(d->act < 2 || d->act >= 4 || ((d->pre_out || d->act == 2) && d->aux1 && (d->act == 2 || d->aux2) && !d->out_f32 && (d->N % 8) == 0),
                 "kai0_gemm_bf16: act=%d (fused GeGLU) needs pre_out and aux inputs, bf16 output", d->act);
    KAI0_REQUIRE(d->nseg >= 0 && d->nseg <= 3, "kai0_gemm_bf16: nseg=%d", d->nseg);
    if (d->nseg > 0) {
        KAI0_REQUIRE(!d->out_f32 && !d->accumulate && (d->batch <= 1) && (d->act < 2 || d->act == 7) && !d->pre_out && (d->N % 8) == 0,
                     "kai0_gemm_bf16: column segments need a plain bf16 epilogue, batch 1");
        KAI0_REQUIRE(d->act != 4, "kai0_gemm_bf16: column segments and act=4 are exclusive");
        for (int i = 0; i < d->nseg; ++i)
            KAI0_REQUIRE(d->seg[i].dst && (d->seg[i].ld % 8) == 0 && (d->seg[i].n_begin % 8) == 0 &&
                             ((uintptr_t)d->seg[i].dst % 16) == 0 && d->seg[i].n_begin == (i ? d->seg[i].n_begin : 0) &&
                             (i == 0 || d->seg[i].n_begin > d->seg[i - 1].n_begin),
                         "kai0_gemm_bf16: segment %d (dst, ld %% 8, ascending n_begin %% 8, first at 0)", i);
    }
    {
        // operands are addressed with 32-bit byte offsets below a 2 GiB buffer descriptor (per batch entry)
        const int64_t a_rows = d->a_rpb ? ((int64_t)((d->a_kc ? d->M : d->K) / d->a_rpb) + 1) * d->a_bs + d->a_off
                                        : (d->a_kc ? d->M : d->K);
        const int64_t b_rows = d->b_rpb ? ((int64_t)((d->b_kc ? d->N : d->K) / d->b_rpb) + 1) * d->b_bs + d->b_off
                                        : (d->b_kc ? d->N : d->K);
        KAI0_REQUIRE(a_rows * d->lda * 2 < (int64_t)0x7FFF0000 && b_rows * d->ldb * 2 < (int64_t)0x7FFF0000,
                     "kai0_gemm_bf16: an operand spans more than 2 GiB per batch entry (A %lld rows, B %lld rows)",
                     (long long)a_rows, (long long)b_rows);
    }
    const int batch = d->batch > 0 ? d->batch : 1;
    GemmArgs p;
    p.A = (const bf16_t*)d->A;
    p.B = (const bf16_t*)d->B;
    p.C = d->C;
    p.M = d->M; p.N = d->act == 6 ? 2 * d->N : d->N; p.K = d->K;  // act 6: logical width = gate | up interleaved
    p.B2 = (const bf16_t*)d->B2;
    p.pre_out2 = (bf16_t*)d->pre_out2;
    p.norm_out = (bf16_t*)d->norm_out;
    p.norm_w = d->norm_w;
    p.norm_b = (const bf16_t*)d->norm_b;
    p.norm_eps = d->norm_eps;
    p.norm_kind = d->norm_kind;
    p.rope_cos = (const bf16_t*)d->rope_cos;
    p.rope_sin = (const bf16_t*)d->rope_sin;
    p.rope_half = d->rope_half;
    p.rope_n_end = d->rope_n_end;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
    p.sA1 = d->sA1; p.sA2 = d->sA2; p.sB1 = d->sB1; p.sB2 = d->sB2; p.sC1 = d->sC1; p.sC2 = d->sC2;
    p.amap = RowMap{d->a_rpb, d->a_bs, d->a_off};
    p.bmap = RowMap{d->b_rpb, d->b_bs, d->b_off};
    p.cmap = RowMap{d->c_rpb, d->c_bs, d->c_off};
    p.bias = d->bias; p.bias_f32 = d->bias_f32;
    p.scale = d->scale == 0.0f ? 1.0f : d->scale;
    p.act = d->act; p.out_f32 = d->out_f32;
    p.pre_out = (bf16_t*)d->pre_out;
    p.aux1 = (const bf16_t*)d->aux1;
    p.aux2 = (const bf16_t*)d->aux2;
    p.gate = (const bf16_t*)d->gate; p.gate_rpb = d->gate_rpb; p.gate_ld = d->gate_ld;
    p.accumulate = d->accumulate;
    p.residual = (const bf16_t*)d->residual; p.ldr = d->ldr; p.sR1 = d->sR1; p.sR2 = d->sR2;
#ifdef KAI0_ABLATE
    static const int ablate = [] { const char* e = getenv("KAI0_GEMM_ABLATE"); return e ? atoi(e) : 0; }();
    p.ablate = ablate;
#else
    p.ablate = 0;
#endif
    // non-temporal stores for what only the backward / the optimizer reads again: the pre-activation outputs always, C when the
    // caller sets c_nontemporal (weight gradients)
    p.nt_pre = 1;
    p.nt_c = d->c_nontemporal != 0;
    p.simple_epi = d->general_epilogue == 0;
    p.rowvec = d->rowvec; p.rv_s1 = d->rv_s1; p.rv_s2 = d->rv_s2; p.rv_ld = d->rv_ld;
    p.nseg = d->nseg;
    for (int i = 0; i < 3; ++i) {
        p.seg_dst[i] = i < d->nseg ? (bf16_t*)d->seg[i].dst : nullptr;
        p.seg_ld[i] = i < d->nseg ? d->seg[i].ld : 0;
        p.seg_begin[i] = i < d->nseg ? d->seg[i].n_begin : 0;
    }
    p.split_k = 1;
    p.k_chunk = d->K;
    const int split = d->split_k > 1 ? d->split_k : 1;
    p.ws = nullptr;
    if (split > 1) {
        KAI0_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= (int64_t)batch * split * d->M * d->N * 4,
                     "kai0_gemm_bf16: split_k=%d needs a workspace of batch*split*M*N*4 bytes", split);
        KAI0_REQUIRE((d->N % 8) == 0, "kai0_gemm_bf16: split_k needs N %% 8 == 0");
        p.split_k = split;
        p.k_chunk = ((d->K + split - 1) / split + BK - 1) / BK * BK;
        p.ws = (float*)d->workspace;
    }
    // tile configuration: 256x256 (1 block of 8 waves per CU, half the staged bytes per FLOP) when the problem gives
    // (nearly) every CU a block; 128x128 (2 blocks per CU) for small problems.
    const int forced = d->tile_cfg;
    const int64_t big_tiles = (int64_t)((d->M + 255) / 256) * ((p.N + 255) / 256) * batch * (split > 1 ? split : 1);
    const bool big = d->act == 7 ? false : (forced ? forced >= 4 : (big_tiles >= 160 && d->K >= 256));  // (act 7: 128-column tiles)
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // measured on the MLP shapes: the ping-pong schedule wins for NT (+3..8 %) and loses for the transpose-read
    // layouts (their load slot is longer than the MFMA slot), so only NT uses it
    const bool pp = forced ? forced == 5 : (d->a_kc && d->b_kc);
    // few 128x128 tiles (at most one block per CU): nothing else hides the load latency -> 4-stage pipeline
    const int64_t small_blocks = (int64_t)((d->M + 127) / 128) * ((p.N + 127) / 128) * batch * (split > 1 ? split : 1);
    const bool deep = forced ? forced == 2 : (small_blocks <= 256 && p.k_chunk >= 256);
    // 128 x 128 on EIGHT waves (4 x 2 wave tiles of 32 x 64; round 6): with at most one block per CU the four-wave loop has one wave per
    // SIMD, so a K-tile is that wave's 8 LDS-DMA issues + 16 fragment reads + 32 MFMAs one after the other (~1470 clocks for 544 of MFMA);
    // two waves per SIMD let one wave's MFMAs run under the other's DMA issue and read latency.  K-contiguous operands, act 0 / 1 / 7 (the
    // epilogues the narrower wave tile implements); kai0_gemm_desc.small_w8: 0 = this rule, 1 = never, 2 = every eligible 128 x 128 launch.
    const bool w8_ok = !big && d->a_kc && d->b_kc && (d->act <= 1 || d->act == 7) && (forced == 0 || forced == 3);
    const bool w8 = w8_ok && (forced == 3 || d->small_w8 == 2 || (d->small_w8 == 0 && deep));
    // forced (kai0_gemm_desc.tile_cfg, A/B runs): 1 / 2 = 128x128 with 2 / 4 stages, 4 = 256x256 plain loop, 5 = 256x256
    // two-buffer ping-pong for every layout.  Measured (MLP shapes, random data): the 32-deep ring wins +21 % for the transpose-read
    // layout (TN wgrads: 512-B source rows, so a 32-deep sub-tile still moves whole cache lines) and loses up to 17 % for NT (64-B
    // source rows = half lines), which runs the quadrant schedule (+6..10 % over the two-buffer ping-pong, 1.37 PFLOP/s at 8192^3).
    // (Act 6 selects its second weight per DMA piece of a 64-deep K-tile: never on the ring.)  Removed after measurement: a 384x256
    // plain tile (equal to the ping-pong, spilled), the quadrant schedule with its DMA pieces between the MFMAs, the ring with two
    // pieces per slot kind (-0.9 %), the ring for NT.
    const bool ring = !forced && d->act != 6 && !d->a_kc && !d->b_kc;
    // persistent NT kernel with the dynamic tile queue (kai0_gemm_desc.persist: 0 = the rule below, 1 = never, 2 = every eligible NT launch)
    const int persist = d->persist == 1 ? 0 : (d->persist == 2 ? 2 : 1);
    const bool ps_ok = !forced && persist && big && d->a_kc && d->b_kc && batch == 1 && split == 1 &&
                       big_tiles >= (persist == 2 ? 512 : 2048) &&  // (B = 1 prefix MLP, 512 tiles = two per CU: 97 -> 136 us persistent)
                       (p.K % 8) == 0 && !d->rowvec && d->a_rpb == 0 && d->b_rpb == 0;
    // the rule (measured inside the training step with KAI0_GEMM_BREAKDOWN=1, round 4): the wide MLP shapes gain — 30976 x 16384 x 2048 with the GeGLU
    // epilogues 993 -> 1046 TFLOP/s (1056 -> 1165 for the pair GEMM alone), x 2048 x 16384 1371 -> 1385 — while launches of < ~1000
    // tiles (q|k|v, o_proj, SigLIP) lose 1-8 %: their tiles are too few for the queue to pay for its hand-over
    const bool ps_rule = p.N >= 8192 || d->K >= 8192;
    if (ps_ok && (persist == 2 || ps_rule)) {
        constexpr int LDS = 2 * 2 * 256 * 64 * 2 + 8 * 4096;  // two stages + the epilogue slabs = 160 KiB
        // per device: the counters' address (a __device__ symbol has one instance per device), the kernel's LDS attribute, the CU count
        struct PsDev { unsigned int* ctr = nullptr; int cus = 0; };
        static PsDev ps_dev[64];
        static std::atomic<unsigned> next_slot{0};
        int dv = 0;
        KAI0_REQUIRE(hipGetDevice(&dv) == hipSuccess && dv >= 0 && dv < 64, "kai0_gemm_bf16: hipGetDevice");
        PsDev& pd = ps_dev[dv];
        if (pd.ctr == nullptr) {
            void* sym = nullptr;
            hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_ps_ctr));
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            hipDeviceProp_t pr;
            if (e == hipSuccess) e = hipGetDeviceProperties(&pr, dv);
            KAI0_REQUIRE(e == hipSuccess, "kai0_gemm_bf16: persistent kernel setup failed: %s", hipGetErrorString(e));
            pd.cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
            pd.ctr = (unsigned int*)sym;
        }
        unsigned int* ctr_base = pd.ctr;
        p.tiles_m = (d->M + 255) / 256;
        p.tiles_n = (p.N + 255) / 256;
        const int nblk = (int)std::min<int64_t>(big_tiles, pd.cus);
        unsigned int* ctr = ctr_base + (size_t)(next_slot.fetch_add(1) % PS_SLOTS) * 16;
        hipLaunchKernelGGL(gemm_nt_persistent_kernel, dim3(nblk), dim3(512), LDS, s, p, ctr);
        return kai0_check_launch("kai0_gemm_bf16 (persistent)");
    }
    // A/B configurations with TWO blocks per CU (round 6, VERDICT r5 #2 "two tiles in flight per CU": one block's epilogue under the other's
    // K loop): 6 = the eight-wave 128 x 128 tile with two stages (64 KiB), 7 = 256 x 128 x 32 on four waves with three stages (72 KiB)
    if (forced == 6) rc = launch_cfg<4, 2, 2, 4, false, 2>(d, p, batch, s);
    else if (forced == 7) rc = launch_cfg<2, 2, 8, 4, false, 3, 32>(d, p, batch, s);
    // TN: ring with every DMA piece (and its offset arithmetic) between the MFMA rows
    else if (big && ring) rc = launch_cfg<2, 4, 8, 4, true, 4, 32, 3>(d, p, batch, s);
    else if (big && !pp) rc = launch_cfg<2, 4, 8, 4, false>(d, p, batch, s);
    else if (big && d->b_kc && forced != 5) rc = launch_cfg<2, 4, 8, 4, true, 2, 64, 1>(d, p, batch, s);  // NT: quadrant schedule
    else if (big) rc = launch_cfg<2, 4, 8, 4, true>(d, p, batch, s);
    else if (w8) rc = launch_cfg<4, 2, 2, 4, false, 4>(d, p, batch, s);
    else if (deep) rc = launch_cfg<2, 2, 4, 4, false, 4>(d, p, batch, s);
    else rc = launch_cfg<2, 2, 4, 4, false>(d, p, batch, s);
    if (rc) return rc;
    rc = kai0_check_launch("kai0_gemm_bf16");
    if (rc || split == 1) return rc;
    if (d->norm_kind != 0) {
        if (d->norm_kind == 1) hipLaunchKernelGGL(splitk_reduce_norm_kernel<1>, dim3(d->M), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(splitk_reduce_norm_kernel<2>, dim3(d->M), dim3(256), 0, s, p);
        return kai0_check_launch("kai0_gemm_bf16(split-K reduce + norm)");
    }
    const int64_t items = (int64_t)d->M * (d->N / 8);
    int rb = (int)((items + 255) / 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb, batch, 1), dim3(256), 0, s, p);
    return kai0_check_launch("kai0_gemm_bf16(split-K reduce)");
}
