// attn_bwd.hip — backward of the masked multi-query attention of the joint Gemma layers (modeling_gemma.py:230-253 through
// autograd), query side: for a block of (token, head) rows
//     D    = rowsum(dO * O)                                   (= <dP, P> of the row)
//     dP   = dO V^T                 (f32, never rounded, never written)
//     dS   = bf16((P * (dP - D)) * scale)                     (written once: dK = dS^T Q still needs it)
//     dQ   = dS K                                              (f32 accumulate, bf16 out)
// in ONE launch.  Round 4 (RC = true, kai0_attn_bwd_dq2): P is not read but RECOMPUTED per key tile — S^T = K Q^T from the K tile
// that is in LDS anyway (the row-major image serves both the b128 row reads of this product and the transpose reads of
// dQ^T += K^T dS^T: with the key swizzle below the 16 lanes of every ds_read_b128 group fall on 16 distinct 16-byte slots), the
// logits rounded and masked exactly as the forward does, P = bf16(exp(s - lse[row])) with the forward's per-row log-sum-exp —
// and WRITTEN once (bf16) next to dS for the two batched GEMMs dV = P^T dO, dK = dS^T Q that follow.  The forward therefore
// stores no probabilities (attention.hip, OP = -1); one extra product here against one fewer pass and one fewer S x S store
// there, and P never sits in HBM between forward and backward (9.6 GB per step at B = 32).
// It replaces kai0_rowdot_bf16 + the K = 256 GEMM with the softmax-backward epilogue (all epilogue: 4 K-tiles
// per 256x256 tile, 277 TFLOP/s) + the dQ GEMM that re-read dS: P is read once, dS written once, dO / V / K stay on chip.
// P is the forward's stored bf16 probabilities — recomputing them (a "flash" backward) would cost 7 GEMM-units of MFMA work
// instead of 4 at HD = 256 and the forward keeps P for dV = P^T dO anyway.
//
// gfx950 mapping (mirrors attention.hip): block = 128 rows on 8 waves (16 rows each, two waves per SIMD); the wave keeps its dO
// rows as MFMA B fragments and its dQ^T accumulators [256 d x 16 rows] in registers; V tiles ([64 keys][64 d] sub-tiles, read
// with ds_read_b128 as A rows of dP^T = V dO^T) and K tiles ([64 keys][HD], read through ds_read_b64_tr_b16 as A = K^T of
// dQ^T += K^T dS^T) arrive by LDS-DMA, double-buffered, zero-filled at the edges by the buffer descriptor.  The A rows of the
// two MFMA tiles of a 32-key group are assigned keys so that lane (row, g) ends up holding dP for keys 8g .. 8g+7 — exactly
// the 16-byte slice of P it needs, the 16-byte slice of dS it writes, and the B fragment of the second product.
#include "common.h"
#include "../../include/kai0hip.h"
#include <limits.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr uint32_t OOB = 0x80000000u;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_PTR(void))lds_dst_wave_uniform, 16, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int AB_KC_MAX = 2048;  // key codes staged in LDS (RC): covers S = 1018 and the estimator's 1786

struct AbArgs {
    const bf16_t* dO;
    const bf16_t* O;
    const bf16_t* P;      // RC: not read
    bf16_t* Pout;         // RC: the recomputed probabilities, same layout as dS
    const bf16_t* Q;      // RC
    const float* lse;     // RC: [batch][s_lse]
    const int32_t* qcode; // RC, optional (with kcode): mask codes, query position of folded row r = r / H
    const int32_t* kcode;
    int64_t s_lse, qcode_ld, kcode_ld;
    int H;
    const bf16_t* K;
    const bf16_t* V;
    bf16_t* dS;
    bf16_t* dQ;
    int rows, Sk, HD;
    int64_t ldo, ldk, ldv, ldp;  // row strides: dO / O / dQ rows, K rows, V rows, P / dS rows
    int64_t sO, sK, sV, sP;      // batch strides
    float scale;
};

// NKS = 64-wide sub-tiles over the head dim (HD <= 64 * NKS)
template <int NKS, bool RC>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_kernel(const AbArgs p) {
    constexpr int KSTEPS = NKS * 2;                // 32-wide contraction steps over the head dim
    constexpr int VT_BYTES = NKS * 8192;           // V: NKS x [64 keys][64 d] bf16, 128-B rows
    constexpr int K_ROWB = NKS * 128;              // K: [64 keys][64 NKS d], one row per key
    constexpr int KT_BYTES = 64 * K_ROWB;
    constexpr int STAGE = VT_BYTES + KT_BYTES;
    constexpr int K_LPR = K_ROWB / 16;             // lanes per K row (32 or 16)
    constexpr int K_RPP = 64 / K_LPR;              // key rows per DMA piece
    constexpr int NVP = NKS * 8 / 8;               // V DMA pieces per wave per tile
    constexpr int NKP = (64 / K_RPP) / 8;          // K DMA pieces per wave per tile
    constexpr int ODT = NKS * 4;                   // 16-wide d tiles of dQ
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int z = blockIdx.y;
    const bf16_t* dOb = p.dO + (int64_t)z * p.sO;
    const bf16_t* Ob = p.O + (int64_t)z * p.sO;
    const bf16_t* Kb = p.K + (int64_t)z * p.sK;
    const bf16_t* Vb = p.V + (int64_t)z * p.sV;
    const bf16_t* Pb = p.P + (int64_t)z * p.sP;
    bf16_t* dSb = p.dS + (int64_t)z * p.sP;
    bf16_t* dQb = p.dQ + (int64_t)z * p.sO;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Pb, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t ds_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dSb, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t po_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(RC ? p.Pout + (int64_t)z * p.sP : dSb), 0, (int)OOB, 0x00020000);
    int* kc_lds = reinterpret_cast<int*>(smem + 2 * STAGE);  // RC: key codes of the whole key range
    const int row = blockIdx.x * 128 + wave * 16 + l15;  // this lane's query row (MFMA column)
    const bool rok = row < p.rows;
    const int ntiles = (p.Sk + 63) / 64;
    const uint32_t ldk2 = (uint32_t)p.ldk * 2, ldv2 = (uint32_t)p.ldv * 2;

    // ---- dO fragments (B operand of dP^T = V dO^T) and D = rowsum(dO * O) ----------------------------------------------
    bf16x8 dof[KSTEPS];
    float dsum = 0.f;
    {
        bf16x8 of[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d = ks * 32 + g * 8;
            bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (rok && d < p.HD) {
                a = *reinterpret_cast<const bf16x8*>(dOb + (int64_t)row * p.ldo + d);
                b = *reinterpret_cast<const bf16x8*>(Ob + (int64_t)row * p.ldo + d);
            }
            dof[ks] = a;
            of[ks] = b;
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += bf2f(dof[ks][e]) * bf2f(of[ks][e]);
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
    }

    // ---- RC: Q fragments (B operand of S^T = K Q^T, same lane layout as dO), the row's log-sum-exp and mask code ---------
    bf16x8 qf[RC ? KSTEPS : 1];
    float lse_row = INFINITY;
    int qc = INT_MAX;
    if constexpr (RC) {
        const bf16_t* Qb = p.Q + (int64_t)z * p.sO;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d = ks * 32 + g * 8;
            bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
            if (rok && d < p.HD) a = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)row * p.ldo + d);
            qf[ks] = a;
        }
        if (rok) lse_row = p.lse[(int64_t)z * p.s_lse + row];
        if (p.qcode != nullptr) qc = rok ? p.qcode[(int64_t)z * p.qcode_ld + row / p.H] : -1;
        const int nkc = ((p.Sk + 63) / 64) * 64;
        for (int i = tid; i < nkc; i += 512) kc_lds[i] = p.kcode == nullptr ? 0 : (i < p.Sk ? p.kcode[(int64_t)z * p.kcode_ld + i] : INT_MAX);
    }

    // ---- staging (attention.hip's layouts with the roles of K and V swapped) ------------------------------------------------
    const int kc_chunk = ((lane & 7) ^ (lane >> 3)) * 8;
    auto stage = [&](int kt, int slot) {
        char* sv = smem + slot * STAGE + wave * (NVP * 1024);
        char* sk = smem + slot * STAGE + VT_BYTES + wave * (NKP * 1024);
        const int key0 = kt * 64;
#pragma unroll
        for (int j = 0; j < NVP; ++j) {
            const int piece = wave * NVP + j;  // 0 .. NKS*8-1
            const int sub = piece >> 3, key = key0 + (piece & 7) * 8 + (lane >> 3);
            const int d = sub * 64 + kc_chunk;
            const uint32_t off = (key < p.Sk && d < p.HD) ? (uint32_t)key * ldv2 + (uint32_t)d * 2 : OOB;
            glds16(v_rsrc, off, sv + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < NKP; ++j) {
            // swizzle key of key-row r: the transpose reads below touch rows {8g .. 8g+3} (then +4) of a 32-key group per 16-lane
            // group, i.e. rows {0-3, 8-11} per 32-lane half: (r & 3) | bit 3 gives them 8 distinct 32-byte slots
            const int r = (wave * NKP + j) * K_RPP + lane / K_LPR;
            const int c = (lane % K_LPR) ^ ((((r & 3) | (((r >> 3) & 1) << 2))) << 1);
            const int key = key0 + r;
            const uint32_t off = (key < p.Sk && c * 8 < p.HD) ? (uint32_t)key * ldk2 + (uint32_t)c * 16 : OOB;
            glds16(k_rsrc, off, sk + j * 1024);
        }
    };
    // this lane's 16-byte slices of P for the two 32-key groups of tile kt (keys 64 kt + 32 hh + 8 g .. +7)
    auto p_off = [&](int kt, int hh) -> uint32_t {
        const int key = kt * 64 + hh * 32 + 8 * g;
        return (rok && key < p.ldp) ? (uint32_t)(((int64_t)row * p.ldp + key) * 2) : OOB;
    };

    f32x4 accq[ODT];
#pragma unroll
    for (int dt = 0; dt < ODT; ++dt) accq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int arow = 8 * (l15 >> 2) + (l15 & 3);  // key (within a 32-key group) fed to A-row l15 of tile 0; tile 1: + 4

    u32x4 pn0 = {0, 0, 0, 0}, pn1 = {0, 0, 0, 0};
    if constexpr (!RC) {
        pn0 = __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, (int)p_off(0, 0), 0, 0);
        pn1 = __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, (int)p_off(0, 1), 0, 0);
    }
    // RC: key tiles in which no key is visible to anybody (all 64 codes INT_MAX: padded prompt slots) contribute nothing to dQ and
    // have P = dS = 0: their two output slices are zero-filled here and the loop walks the live tiles only (as the forward does)
    int* lt = kc_lds + AB_KC_MAX;  // [0] = number of live tiles, [1 + i] = i-th live tile, [40 + t] = flag of tile t
    int nlive = ntiles;
    if constexpr (RC) {
        __syncthreads();  // the codes are in LDS
        for (int t = wave; t < ntiles; t += 8) {
            const bool any = __any(kc_lds[t * 64 + lane] != INT_MAX);
            if (lane == 0) lt[40 + t] = any ? 1 : 0;
        }
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int t = 0; t < ntiles; ++t)
                if (lt[40 + t]) lt[1 + n++] = t;
            lt[0] = n;
        }
        __syncthreads();
        nlive = __builtin_amdgcn_readfirstlane(lt[0]);
        if (nlive < ntiles) {
            const u32x4 zero = {0, 0, 0, 0};
            for (int t = 0; t < ntiles; ++t)
                if (!__builtin_amdgcn_readfirstlane(lt[40 + t])) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        __builtin_amdgcn_raw_buffer_store_b128(zero, po_rsrc, (int)p_off(t, hh), 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(zero, ds_rsrc, (int)p_off(t, hh), 0, 0);
                    }
                }
        }
    }
    auto tile_at = [&](int i) -> int { return RC ? __builtin_amdgcn_readfirstlane(lt[1 + i]) : i; };
    if (nlive > 0) stage(tile_at(0), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int it = 0; it < nlive; ++it) {
        const int kt = tile_at(it);
        const int buf = it & 1;
        const u32x4 pc[2] = {pn0, pn1};
        // next tile's P slices first, then its K / V tiles: the P loads are older than the DMA, so waiting for them later
        // never waits for the tiles
        if (it + 1 < nlive) {
            if constexpr (!RC) {
                pn0 = __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, (int)p_off(kt + 1, 0), 0, 0);
                pn1 = __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, (int)p_off(kt + 1, 1), 0, 0);
            }
            stage(tile_at(it + 1), buf ^ 1);
        }
        const char* tv = smem + buf * STAGE;
        const char* tk = tv + VT_BYTES;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            // dP^T = V dO^T (and, RC, S^T = K Q^T for the same (key, row) pairs): per 32-wide contraction step two V rows (two K rows)
            // of the permuted 32-key group against the row's dO (Q) fragment.  Reads run one step ahead of the MFMAs that consume them
            // (double-buffered; the sched_barriers pin that order — hipcc otherwise waits for every read right in front of its MFMA).
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
            const int r0 = hh * 32 + arow, r1 = r0 + 4;
            const int key0 = ((r0 & 3) | (((r0 >> 3) & 1) << 2)) << 1, key1 = ((r1 & 3) | (((r1 >> 3) & 1) << 2)) << 1;
            // two products, one after the other (not interleaved: two fragment double-buffers at once do not fit the 256 registers
            // of the HD = 256 recompute form): which = 0 -> V rows against dO (a0, a1), which = 1 -> K rows against Q (s0, s1)
            auto product = [&](auto whichc, f32x4& c0, f32x4& c1, const bf16x8 (&bfr)[KSTEPS]) {
                constexpr int which = decltype(whichc)::value;
                bf16x8 fr[2][2];
                auto rd = [&](int ks, bf16x8 (&dst)[2]) {
                    if constexpr (which == 0) {
                        const char* sub = tv + (ks >> 1) * 8192;
                        const int chunk = (ks & 1) * 4 + g;
                        dst[0] = *reinterpret_cast<const bf16x8*>(sub + r0 * 128 + ((chunk ^ (r0 & 7)) << 4));
                        dst[1] = *reinterpret_cast<const bf16x8*>(sub + r1 * 128 + ((chunk ^ (r1 & 7)) << 4));
                    } else {  // K rows: b128 reads of the row-major image (conflict-free with the key swizzle, see the header)
                        const int kch = ks * 4 + g;
                        dst[0] = *reinterpret_cast<const bf16x8*>(tk + r0 * K_ROWB + ((kch ^ key0) << 4));
                        dst[1] = *reinterpret_cast<const bf16x8*>(tk + r1 * K_ROWB + ((kch ^ key1) << 4));
                    }
                };
                rd(0, fr[0]);
                rd(1, fr[1]);
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks += 2) {  // two steps (four reads) ahead
                    __builtin_amdgcn_sched_barrier(0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], bfr[ks], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][1], bfr[ks], c1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 2 < KSTEPS) rd(ks + 2, fr[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], bfr[ks + 1], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][1], bfr[ks + 1], c1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 3 < KSTEPS) rd(ks + 3, fr[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (RC) product(std::integral_constant<int, 1>{}, s0, s1, qf);
            // lane (row, g) holds dP (and S) for keys 64 kt + 32 hh + 8 g + e  (e < 4: a0 / s0, e >= 4: a1 / s1)
            bf16x8 pv;
            if constexpr (RC) {
                const int kbase = kt * 64 + hh * 32 + 8 * g;
                const i32x4 c0 = *reinterpret_cast<const i32x4*>(kc_lds + kbase), c1 = *reinterpret_cast<const i32x4*>(kc_lds + kbase + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = rbf(rbf(e < 4 ? s0[e] : s1[e - 4]) * p.scale);   // the forward's rounding points
                    const bool ok = (kbase + e < p.Sk) && ((e < 4 ? c0[e] : c1[e - 4]) <= qc);
                    pv[e] = f2bf(ok ? __expf(sv - lse_row) : 0.f);                    // lse = +inf (row sees no key): 0
                }
                u32x4 pw;
                __builtin_memcpy(&pw, &pv, 16);
                __builtin_amdgcn_raw_buffer_store_b128(pw, po_rsrc, (int)p_off(kt, hh), 0, 0);  // always issued (OOB = dropped)
            } else {
                const u32x4 pw = pc[hh];
                __builtin_memcpy(&pv, &pw, 16);
            }
            product(std::integral_constant<int, 0>{}, a0, a1, dof);
            bf16x8 ds;
#pragma unroll
            for (int e = 0; e < 8; ++e) ds[e] = f2bf((bf2f(pv[e]) * ((e < 4 ? a0[e] : a1[e - 4]) - dsum)) * p.scale);
            u32x4 dw;
            __builtin_memcpy(&dw, &ds, 16);
            __builtin_amdgcn_raw_buffer_store_b128(dw, ds_rsrc, (int)p_off(kt, hh), 0, 0);  // always issued (OOB = dropped)
            // dQ^T += K^T dS^T : A = K^T fragment [16 d x 32 keys] through the transpose read of the row-major K tile, four d-tiles
            // (8 transpose reads) one batch ahead of their MFMAs
            const int r_lo = hh * 32 + 8 * g + (l15 >> 2), r_hi = r_lo + 4;
            const int kkey = ((r_lo & 3) | (((r_lo >> 3) & 1) << 2)) << 1;  // same for r_hi
            constexpr int QB = (RC && NKS == 4) ? 2 : 4;  // d-tiles per batch (the recompute form at HD = 256 is at the register limit)
            bf16x8 kq[2][QB];
            auto rdq = [&](int b, bf16x8 (&dst)[QB]) {
#pragma unroll
                for (int i = 0; i < QB; ++i) {
                    const int chunk = (b * QB + i) * 2 + ((l15 & 3) >> 1);
                    const int sub8 = (l15 & 1) * 8;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (LDS_PTR(bf16x4))(tk + r_lo * K_ROWB + ((chunk ^ kkey) << 4) + sub8));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (LDS_PTR(bf16x4))(tk + r_hi * K_ROWB + ((chunk ^ kkey) << 4) + sub8));
                    dst[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            rdq(0, kq[0]);
#pragma unroll
            for (int b = 0; b < ODT / QB; ++b) {
                if (b + 1 < ODT / QB) rdq(b + 1, kq[(b + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < QB; ++i) accq[b * QB + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kq[b & 1][i], ds, accq[b * QB + i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the next tile's DMA and P loads must have landed; the two dS (RC: + two P) stores issued after them may still fly
        if constexpr (RC) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        lds_barrier();
    }
    // ---- dQ: lane (row, g) holds d = 16 dt + 4 g + r ---------------------------------------------------------------------
    if (rok) {
#pragma unroll
        for (int dt = 0; dt < ODT; ++dt) {
            const int d = dt * 16 + 4 * g;
            if (d < p.HD) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(accq[dt][e]);
                *reinterpret_cast<bf16x4*>(dQb + (int64_t)row * p.ldo + d) = o;
            }
        }
    }
}

}  // namespace

static int attn_bwd_dq_launch(const void* dO, const void* O, const void* P, const void* K, const void* V, void* dS, void* dQ,
                              int batch, int rows, int Sk, int HD, int64_t ldo, int64_t ldk, int64_t ldv, int64_t ldp,
                              int64_t sO, int64_t sK, int64_t sV, int64_t sP, float scale, kai0_stream_t stream,
                              const void* Q, const float* lse, int64_t s_lse, const int32_t* qcode, int64_t qcode_ld,
                              const int32_t* kcode, int64_t kcode_ld, int H, bool rc) {
    KAI0_REQUIRE(dO && O && P && K && V && dS && dQ, "kai0_attn_bwd_dq: null operand");
    KAI0_REQUIRE(!rc || (Q && lse && s_lse >= rows && H >= 1 && ((qcode == nullptr) == (kcode == nullptr)) && Sk <= AB_KC_MAX),
                 "kai0_attn_bwd_dq2: needs Q, lse (s_lse >= rows), H >= 1, both or neither mask code, Sk <= %d", AB_KC_MAX);
    KAI0_REQUIRE(HD % 8 == 0 && HD > 0 && HD <= 256, "kai0_attn_bwd_dq: HD=%d unsupported", HD);
    KAI0_REQUIRE(ldo % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldp % 8 == 0 && ldp >= Sk,
                 "kai0_attn_bwd_dq: leading dims must be multiples of 8 and ldp >= Sk");
    KAI0_REQUIRE((int64_t)Sk * ldk * 2 < (int64_t)0x7FFF0000 && (int64_t)Sk * ldv * 2 < (int64_t)0x7FFF0000 &&
                     (int64_t)rows * ldp * 2 < (int64_t)0x7FFF0000,
                 "kai0_attn_bwd_dq: an operand spans more than 2 GiB per batch entry");
    if (rows <= 0 || Sk <= 0 || batch <= 0) return 0;
    AbArgs a{(const bf16_t*)dO, (const bf16_t*)O, rc ? nullptr : (const bf16_t*)P, rc ? (bf16_t*)P : nullptr, (const bf16_t*)Q, lse,
             qcode, kcode, s_lse, qcode_ld, kcode_ld, H,
             (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)dS, (bf16_t*)dQ, rows, Sk, HD, ldo, ldk, ldv, ldp, sO, sK, sV, sP, scale};
    const dim3 grid((rows + 127) / 128, batch, 1);
    hipStream_t s = (hipStream_t)stream;
#define KAI0_AB_LAUNCH(NKS, RC)                                                                                          \
    do {                                                                                                               \
        constexpr int LDS = 2 * (NKS * 8192 + 64 * NKS * 128) + (RC ? AB_KC_MAX * 4 + 512 : 0);                                \
        static bool attr_set = false;                                                                                  \
        auto kern = attn_bwd_dq_kernel<NKS, RC>;                                                                       \
        if (!attr_set) {                                                                                               \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);      \
            KAI0_REQUIRE(e == hipSuccess, "kai0_attn_bwd_dq: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e)); \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, grid, dim3(512), LDS, s, a);                                                          \
    } while (0)
    if (rc) {
        if (HD <= 128) KAI0_AB_LAUNCH(2, true);
        else KAI0_AB_LAUNCH(4, true);
    } else {
        if (HD <= 128) KAI0_AB_LAUNCH(2, false);
        else KAI0_AB_LAUNCH(4, false);
    }
#undef KAI0_AB_LAUNCH
    return kai0_check_launch("kai0_attn_bwd_dq");
}

KAI0_API int kai0_attn_bwd_dq(const void* dO, const void* O, const void* P, const void* K, const void* V, void* dS, void* dQ,
                              int batch, int rows, int Sk, int HD, int64_t ldo, int64_t ldk, int64_t ldv, int64_t ldp,
                              int64_t sO, int64_t sK, int64_t sV, int64_t sP, float scale, kai0_stream_t stream) {
    return attn_bwd_dq_launch(dO, O, P, K, V, dS, dQ, batch, rows, Sk, HD, ldo, ldk, ldv, ldp, sO, sK, sV, sP, scale, stream, nullptr,
                              nullptr, 0, nullptr, 0, nullptr, 0, 1, false);
}

KAI0_API int kai0_attn_bwd_desc_size(void) { return (int)sizeof(kai0_attn_bwd_desc); }

KAI0_API int kai0_attn_bwd_dq2(const kai0_attn_bwd_desc* d, kai0_stream_t stream) {
    KAI0_REQUIRE(d != nullptr, "kai0_attn_bwd_dq2: null descriptor");
    return attn_bwd_dq_launch(d->dO, d->O, d->P, d->K, d->V, d->dS, d->dQ, d->batch, d->rows, d->Sk, d->HD, d->ldo, d->ldk, d->ldv,
                              d->ldp, d->sO, d->sK, d->sV, d->sP, d->scale, stream, d->Q, d->lse, d->s_lse, d->qcode, d->qcode_ld,
                              d->kcode, d->kcode_ld, d->H, true);
}
