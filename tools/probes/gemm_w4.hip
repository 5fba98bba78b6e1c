// Experiment: NT GEMM tile run by FOUR waves (one per SIMD, up to 512 registers each: the 128 x TN/2 accumulators live in AGPRs) instead
// of eight, software-pipelined inside each wave with ONE block barrier per K-tile:
//     [MFMAs of k-half 0 | fragment reads of k-half 1]  vmcnt(0) lgkmcnt(0) barrier  [MFMAs of k-half 1 | LDS-DMA of tile t+2 | reads of tile t+1, k-half 0]
// TN = 256: the production tile (131 FLOP per staged byte), TN = 384: 157 FLOP per staged byte (two 80 KiB stages = all of LDS).
// Question: does a bigger tile per CU (fewer bytes through the L2 -> LDS path per FLOP) raise the rate, and does hipcc schedule a
// one-wave-per-SIMD loop well enough to find out?
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gemm_w4.hip -Lkai0_amd/lib -lkai0hip -Wl,-rpath,'$ORIGIN/../../kai0_amd/lib' -o tools/probes/gemm_w4.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../include/kai0hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T) __attribute__((address_space(3))) T*
constexpr int BK = 64;

template <int TN>  // tile 256 x TN, waves 2 x 2, each 128 x TN/2
__global__ __launch_bounds__(256, 1) void gemm_w4(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M,
                                                  int N, int K, int tiles_m, int tiles_n) {
    constexpr int TM = 256, NJ = TN / 32;                  // NJ column tiles (16 wide) per wave
    constexpr int A_TILE = TM * 128, B_TILE = TN * 128, STAGE = A_TILE + B_TILE;
    constexpr int NP = STAGE / 1024 / 4;                   // DMA pieces per wave per K-tile (16 / 20)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = pid & 7, idx = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP = 4;
    const int width = GROUP * tiles_n, group = pid / width, first_m = group * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP), in_g = pid - group * width;
    const int m0 = (first_m + in_g % gsz) * TM, n0 = (in_g / gsz) * TN;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    const int nk = K / BK;
    // piece q of a stage = 8 rows x 128 B: pieces [0, TM/8) are A rows, the rest B rows; wave w issues pieces w*NP .. w*NP + NP-1.
    // lane -> row (lane >> 3) of the piece, 16-B chunk (lane & 7) ^ (row & 7) of the K-tile's 128 B (the swizzle of the LDS image)
    uint32_t voff[NP];
    const uint32_t kc2 = (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = wave * NP + j;
        const bool isa = q < TM / 8;
        const int row = (isa ? m0 + q * 8 : n0 + (q - TM / 8) * 8) + (lane >> 3);
        voff[j] = (uint32_t)row * (uint32_t)K * 2 + kc2;
    }
    auto stage = [&](int t, int slot) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = wave * NP + j;  // (wave-uniform; the A / B choice is per piece)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q < TM / 8 ? a_rsrc : b_rsrc, (LDS_PTR(void))(smem + slot * STAGE + q * 1024), 16,
                                                      (int)voff[j], t * (BK * 2), 0, 0);
        }
    };
    const int fa = (wm * 128 + l15) * 128 + ((g ^ (l15 & 7)) << 4);
    const int fb = A_TILE + (wn * (TN / 2) + l15) * 128 + ((g ^ (l15 & 7)) << 4);
    auto frag = [&](int slot, int lane_off, int tile16, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smem + slot * STAGE + ((lane_off + tile16 * 2048) ^ (ks << 6)));
    };
    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[8], b0[NJ], a1[8], b1[NJ];

    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) a0[i] = frag(0, fa, i, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) b0[j] = frag(0, fb, j, 0);

    for (int t = 0; t < nk; ++t) {
        const int slot = t & 1;
        // ---- k-half 0: MFMAs on (a0, b0) while the fragments of k-half 1 are read
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a1[i] = frag(slot, fa, i, 1);
            if (i < NJ) b1[i] = frag(slot, fb, i, 1);
            if (NJ > 8 && i < NJ - 8) b1[8 + i] = frag(slot, fb, 8 + i, 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // every wave has read all of tile t (its slot is free) and its own pieces of tile t+1 have landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-half 1: MFMAs on (a1, b1) while tile t+2 is requested into this slot and tile t+1's k-half 0 fragments are read
        // (nothing conditional: past the end the last tile is simply requested again into a slot nobody reads any more, and the
        // fragment reads fetch bytes that are never used)
        const int tn = min(t + 2, nk - 1) * (BK * 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = i * NP / 8; j < (i + 1) * NP / 8; ++j) {
                const int q = wave * NP + j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(q < TM / 8 ? a_rsrc : b_rsrc, (LDS_PTR(void))(smem + slot * STAGE + q * 1024), 16,
                                                          (int)voff[j], tn, 0, 0);
            }
            a0[i] = frag(slot ^ 1, fa, i, 0);
            if (i < NJ) b0[i] = frag(slot ^ 1, fb, i, 0);
            if (NJ > 8 && i < NJ - 8) b0[8 + i] = frag(slot ^ 1, fb, 8 + i, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- epilogue: accumulators -> wave-private f32 slab [16][16 * NJ] -> 16-B bf16 stores
    float* slab = reinterpret_cast<float*>(smem + wave * (16 * 16 * NJ * 4));
    constexpr int WC = 16 * NJ;  // the wave's columns
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(4 * g + r) * WC + j * 16 + l15] = acc[ti][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < 16 * WC / 8; idx += 64) {
            const int r = idx / (WC / 8), c8 = (idx % (WC / 8)) * 8;
            const float* sp = slab + r * WC + c8;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)sp[e];
            const int row = m0 + wm * 128 + ti * 16 + r, col = n0 + wn * WC + c8;
            *reinterpret_cast<bf16x8*>(C + (int64_t)row * N + col) = o;
        }
    }
}

// The same four-wave loop on 32 x 32 x 16 MFMAs (32 cycles each instead of 16: half as many issue slots per K-tile, twice the room
// behind each one for the ~60-cycle LDS-DMA issue).  Wave tile 128 x 128 = 4 x 4 tiles of 32 x 32; a K-tile = two halves of two
// 16-deep steps; fragment f = tile * 2 + step: lane -> row (lane % 32) of the tile, 16-B chunk 2 * step + lane / 32 of the half.
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256, 1) void gemm_w4_m32(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M,
                                                      int N, int K, int tiles_m, int tiles_n) {
    constexpr int TM = 256, TN = 256, A_TILE = TM * 128, STAGE = 2 * A_TILE, NP = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = pid & 7, idx = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP = 4;
    const int width = GROUP * tiles_n, group = pid / width, first_m = group * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP), in_g = pid - group * width;
    const int m0 = (first_m + in_g % gsz) * TM, n0 = (in_g / gsz) * TN;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    const int nk = K / BK;
    uint32_t voff[NP];
    const uint32_t kc2 = (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = wave * NP + j;
        const int row = (q < 32 ? m0 + q * 8 : n0 + (q - 32) * 8) + (lane >> 3);
        voff[j] = (uint32_t)row * (uint32_t)K * 2 + kc2;
    }
    auto issue = [&](int j, int tbytes, int slot) {
        const int q = wave * NP + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(q < 32 ? a_rsrc : b_rsrc, (LDS_PTR(void))(smem + slot * STAGE + q * 1024), 16, (int)voff[j], tbytes, 0, 0);
    };
    // fragment (tile t, half h, step k2): row = w*128 + t*32 + l31, chunk = h*4 + k2*2 + hi, swizzled by row & 7
    const int fa = (wm * 128 + l31) * 128, fb = A_TILE + (wn * 128 + l31) * 128;
    const int r7 = l31 & 7;
    auto frag = [&](int slot, int base, int t, int h, int k2) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smem + slot * STAGE + base + t * 4096 + (((h * 4 + k2 * 2 + hi) ^ r7) << 4));
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 a0[8], b0[8], a1[8], b1[8];
#pragma unroll
    for (int j = 0; j < NP; ++j) issue(j, 0, 0);
#pragma unroll
    for (int j = 0; j < NP; ++j) issue(j, min(1, nk - 1) * (BK * 2), 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        a0[f] = frag(0, fa, f >> 1, 0, f & 1);
        b0[f] = frag(0, fb, f >> 1, 0, f & 1);
    }
    for (int t = 0; t < nk; ++t) {
        const int slot = t & 1;
#pragma unroll
        for (int f = 0; f < 8; ++f) {  // f = tile * 2 + step
            a1[f] = frag(slot, fa, f >> 1, 1, f & 1);
            b1[f] = frag(slot, fb, f >> 1, 1, f & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[f >> 1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[f], b0[j * 2 + (f & 1)], acc[f >> 1][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int tn = min(t + 2, nk - 1) * (BK * 2);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            issue(2 * f, tn, slot);
            issue(2 * f + 1, tn, slot);
            a0[f] = frag(slot ^ 1, fa, f >> 1, 0, f & 1);
            b0[f] = frag(slot ^ 1, fb, f >> 1, 0, f & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[f >> 1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[f], b1[j * 2 + (f & 1)], acc[f >> 1][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // epilogue: a 32 x 32 accumulator tile: lane -> column l31, register e -> row (e / 4) * 8 + hi * 4 + e % 4
    float* slab = reinterpret_cast<float*>(smem + wave * (32 * 128 * 4));  // [32 rows][128 columns] f32 per wave
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[((e >> 2) * 8 + hi * 4 + (e & 3)) * 128 + j * 32 + l31] = acc[ti][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < 32 * 128 / 8; idx += 64) {
            const int r = idx >> 4, c8 = (idx & 15) * 8;
            const float* sp = slab + r * 128 + c8;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)sp[e];
            *reinterpret_cast<bf16x8*>(C + (int64_t)(m0 + wm * 128 + ti * 32 + r) * N + n0 + wn * 128 + c8) = o;
        }
    }
}

// The transpose-read form (weight gradients): C[M][N] = A^T B with A stored [K][M], B stored [K][N].  LDS image of an operand tile
// = [64 k-rows][256 columns] (512-B rows), 2 k-rows per DMA piece, 16-B slot s of row r holding source chunk s ^ (key(r) << 1),
// key(r) = (r & 3) | (((r >> 3) & 1) << 2); fragments by ds_read_b64_tr_b16 (two per 16 x 32 operand), as in the production kernel.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ __launch_bounds__(256, 1) void gemm_w4_tn(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M,
                                                     int N, int K, int tiles_m, int tiles_n) {
    constexpr int TM = 256, TN = 256, NJ = 8, A_TILE = 64 * 512, STAGE = 2 * A_TILE, NP = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = pid & 7, idx = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP = 4;
    const int width = GROUP * tiles_n, group = pid / width, first_m = group * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP), in_g = pid - group * width;
    const int m0 = (first_m + in_g % gsz) * TM, n0 = (in_g / gsz) * TN;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    const int nk = K / BK;
    // piece q: pieces [0, 32) = A k-rows 2q, 2q+1; [32, 64) = B k-rows; wave w issues pieces 16 w .. 16 w + 15
    // (offsets recomputed per piece — a handful of VALU ops — rather than held in 16 registers: with them hipcc started rotating
    // accumulators between AGPRs and VGPRs, 300 v_accvgpr moves per K-tile)
    auto piece_voff = [&](int j) -> uint32_t {
        const int q = wave * NP + j;
        const bool isa = q < 32;
        const int r = (q & 31) * 2 + (lane >> 5);
        const int key = (r & 3) | (((r >> 3) & 1) << 2);
        const int col = (isa ? m0 : n0) + (((lane & 31) ^ (key << 1)) * 8);
        return (uint32_t)col * 2 + (uint32_t)r * (uint32_t)(isa ? M : N) * 2;
    };
    const int sa_k = BK * M * 2, sb_k = BK * N * 2;  // byte advance per K-tile (soffset)
    auto issue = [&](int j, int t, int slot) {
        const int q = wave * NP + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(q < 32 ? a_rsrc : b_rsrc, (LDS_PTR(void))(smem + slot * STAGE + q * 1024), 16, (int)piece_voff(j),
                                                  t * (q < 32 ? sa_k : sb_k), 0, 0);
    };
    // fragment addresses: operand column group c (16 columns) of the wave's 128: chunk = 16 w + 2 c + b, b = (l15 & 3) >> 1;
    // (2 c + b) ^ (key << 1) = 2 (c ^ key) + b with key = (l15 >> 2) | ((g & 1) << 2); k-row r0 = 32 ks + 8 g + (l15 >> 2), second read r0 + 4
    const int keyv = (l15 >> 2) | ((g & 1) << 2), bb = (l15 & 3) >> 1;
    int adA[8], adB[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int row_part = (8 * g + (l15 >> 2)) * 512 + (l15 & 1) * 8;
        adA[c] = row_part + ((wm * 16 + 2 * (c ^ keyv) + bb) << 4);
        adB[c] = A_TILE + row_part + ((wn * 16 + 2 * (c ^ keyv) + bb) << 4);
    }
    auto frag = [&](int slot, int ad, int ks) -> bf16x8 {
        const char* b = smem + slot * STAGE + ad + ks * 16384;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))(b));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))(b + 2048));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[8], b0[8], a1[8], b1[8];
#pragma unroll
    for (int j = 0; j < NP; ++j) issue(j, 0, 0);
#pragma unroll
    for (int j = 0; j < NP; ++j) issue(j, min(1, nk - 1), 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a0[i] = frag(0, adA[i], 0);
        b0[i] = frag(0, adB[i], 0);
    }
    for (int t = 0; t < nk; ++t) {
        const int slot = t & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a1[i] = frag(slot, adA[i], 1);
            b1[i] = frag(slot, adB[i], 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int tn = min(t + 2, nk - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            issue(2 * i, tn, slot);
            issue(2 * i + 1, tn, slot);
            a0[i] = frag(slot ^ 1, adA[i], 0);
            b0[i] = frag(slot ^ 1, adB[i], 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem + wave * (16 * 128 * 4));
    constexpr int WC = 128;
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(4 * g + r) * WC + j * 16 + l15] = acc[ti][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < 16 * WC / 8; idx += 64) {
            const int r = idx / (WC / 8), c8 = (idx % (WC / 8)) * 8;
            const float* sp = slab + r * WC + c8;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)sp[e];
            const int row = m0 + wm * 128 + ti * 16 + r, col = n0 + wn * WC + c8;
            *reinterpret_cast<bf16x8*>(C + (int64_t)row * N + col) = o;
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main() {
    const int shapes[][3] = {{30720, 15360, 2048}, {30720, 2304, 16384}, {7680, 7680, 8192}};  // multiples of 256 and 384
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
        uint32_t s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        auto gauss = [&]() { return (rnd() + rnd() + rnd() + rnd()) * 1.732f; };
        for (auto& v : hA) v = f2bf(gauss());
        for (auto& v : hB) v = f2bf(gauss() * 0.03f);
        bf16_t *dA, *dB, *dC, *dR;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dR, (size_t)M * N * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](auto fn) {
            for (int i = 0; i < 5; ++i) fn();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) fn();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12;
        };
        kai0_gemm_desc d; memset(&d, 0, sizeof d);
        d.A = dA; d.B = dB; d.C = dR; d.M = M; d.N = N; d.K = K; d.a_kc = 1; d.b_kc = 1; d.lda = K; d.ldb = K; d.ldc = N;
        d.batch = 1; d.batch_inner = 1; d.scale = 1.0f; d.split_k = 1;
        CK(hipFuncSetAttribute((const void*)gemm_w4<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128));
        CK(hipFuncSetAttribute((const void*)gemm_w4<384>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 384) * 128));
        auto prod = [&](int persist) { kai0_gemm_set_persist(persist); return timeit([&] { if (kai0_gemm_bf16(&d, nullptr)) { printf("gemm: %s\n", kai0_last_error()); exit(1); } }); };
        auto w256 = [&] { return timeit([&] { hipLaunchKernelGGL(gemm_w4<256>, dim3((M / 256) * (N / 256)), dim3(256), 2 * 512 * 128, 0, dA, dB, dC, M, N, K, M / 256, N / 256); }); };
        CK(hipFuncSetAttribute((const void*)gemm_w4_m32, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        auto wm32 = [&] { return timeit([&] { hipLaunchKernelGGL(gemm_w4_m32, dim3((M / 256) * (N / 256)), dim3(256), 131072, 0, dA, dB, dC, M, N, K, M / 256, N / 256); }); };
        auto w384 = [&] { return timeit([&] { hipLaunchKernelGGL(gemm_w4<384>, dim3((M / 256) * (N / 384)), dim3(256), 2 * 640 * 128, 0, dA, dB, dC, M, N, K, M / 256, N / 384); }); };
        prod(0);  // clock ramp
        double r[2][4], r32[2];
        size_t bad32 = 0;
        std::vector<uint16_t> hC((size_t)M * N), hR((size_t)M * N);
        size_t bad[2] = {0, 0};
        for (int pass = 0; pass < 2; ++pass) {
            r[pass][0] = prod(0);
            r[pass][1] = prod(2);
            r[pass][2] = w256();
            if (pass == 0) {
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hR.data(), dR, hR.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hC.size(); ++i) bad[0] += hC[i] != hR[i];
            }
            r32[pass] = wm32();
            if (pass == 0) {
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hC.size(); ++i) bad32 += hC[i] != hR[i];
            }
            r[pass][3] = w384();
            if (pass == 0) {
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hC.size(); ++i) bad[1] += hC[i] != hR[i];
            }
        }
        kai0_gemm_set_persist(1);
        CK(hipGetLastError());
        printf("   4 waves 256x256 on 32x32x16 MFMAs %6.0f %6.0f (mismatches %zu)\n", r32[0], r32[1], bad32);
        printf("%6d x %6d x %6d TFLOP/s (two passes): production plain %6.0f %6.0f  persistent %6.0f %6.0f | 4 waves 256x256 %6.0f %6.0f (mismatches %zu)  256x384 %6.0f %6.0f (mismatches %zu)\n",
               M, N, K, r[0][0], r[1][0], r[0][1], r[1][1], r[0][2], r[1][2], bad[0], r[0][3], r[1][3], bad[1]);
        fflush(stdout);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dR);
    }
    // ---- transpose-read (weight-gradient) shapes: C[M][N] = A[K][M]^T B[K][N]
    const int tshapes[][3] = {{16384, 2048, 30976}, {2048, 16384, 30976}, {8192, 8192, 8192}};
    CK(hipFuncSetAttribute((const void*)gemm_w4_tn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (auto& sh : tshapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
        uint32_t s = 777;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        auto gauss = [&]() { return (rnd() + rnd() + rnd() + rnd()) * 1.732f; };
        for (auto& v : hA) v = f2bf(gauss() * 0.05f);
        for (auto& v : hB) v = f2bf(gauss());
        bf16_t *dA, *dB, *dC, *dR;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dR, (size_t)M * N * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](auto fn) {
            for (int i = 0; i < 5; ++i) fn();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) fn();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12;
        };
        kai0_gemm_desc d; memset(&d, 0, sizeof d);
        d.A = dA; d.B = dB; d.C = dR; d.M = M; d.N = N; d.K = K; d.a_kc = 0; d.b_kc = 0; d.lda = M; d.ldb = N; d.ldc = N;
        d.batch = 1; d.batch_inner = 1; d.scale = 1.0f; d.split_k = 1;
        auto prod = [&] { return timeit([&] { if (kai0_gemm_bf16(&d, nullptr)) { printf("gemm: %s\n", kai0_last_error()); exit(1); } }); };
        auto w4 = [&] { return timeit([&] { hipLaunchKernelGGL(gemm_w4_tn, dim3((M / 256) * (N / 256)), dim3(256), 131072, 0, dA, dB, dC, M, N, K, M / 256, N / 256); }); };
        prod();
        double r[2][2];
        size_t bad = 0;
        for (int pass = 0; pass < 2; ++pass) {
            r[pass][0] = prod();
            r[pass][1] = w4();
            if (pass == 0) {
                std::vector<uint16_t> hC((size_t)M * N), hR((size_t)M * N);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hR.data(), dR, hR.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hC.size(); ++i) bad += hC[i] != hR[i];
            }
        }
        CK(hipGetLastError());
        printf("TN %6d x %6d x %6d TFLOP/s (two passes): production ring %6.0f %6.0f | 4 waves 256x256 %6.0f %6.0f (mismatches %zu)\n", M, N, K, r[0][0], r[1][0], r[0][1], r[1][1], bad);
        fflush(stdout);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dR);
    }
    return 0;
}
