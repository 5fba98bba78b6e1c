// attention.hip — fused attention forward for the pi0.5 path: logits, prefix-LM / padding mask, f32 softmax, P V.
//
// Replaces eager_attention_forward (modeling_gemma.py:230-253; modeling_siglip.py:325-345) with its exact rounding
// points: logits = bf16(bf16(Q K^T) * scale), softmax in f32, P = bf16(softmax), O = bf16(P V).
//
// gfx950 design:
//   * training (round 4): ONE pass with an online softmax (OP = -1): per key tile the logits are computed once, the running row
//     max m and row sum l are updated, O is rescaled by exp(m_old - m_new) when a row's max moved, P~ = bf16(exp(s - m)) goes
//     straight into P V; at the end O /= l and lse = m + log(l) is written per row.  NO probabilities leave the chip: the
//     backward recomputes them from Q, K and lse (attn_bwd.hip, attn_siglip.hip).  The logits keep the reference's rounding
//     (bf16(bf16(Q K^T) * scale)); the point where P is rounded to bf16 moves from "after the normalisation" to "before" —
//     inside the stated floating-point tolerance, not bit-identical to eager_attention_forward.
//   * the former "two-pass" form (OP = 0: pass 1 row max / row sum, pass 2 recomputes the logits, writes the FINAL normalised P
//     and accumulates O with that bf16 P — bitwise the reference's rounding order) is kept for callers that ask for P and as the
//     A/B alternative (ops.py: KAI0_ATTN_STORE_P=1).
//   * transposed orientation: S^T = K Q^T and O^T = V^T P^T.  The C-layout of S^T (lane = query column, registers =
//     4 consecutive keys) IS the B-operand layout of the second MFMA, so P never leaves registers; softmax statistics
//     are per lane (no cross-lane traffic in the key loop); V^T comes from a row-major V tile through
//     ds_read_b64_tr_b16.
//   * multi-query folding: the caller presents the H query heads of a position as extra rows (rows = Sq*H), so one
//     K/V tile in LDS serves all heads.
//   * block = 4 waves, one per SIMD (512-register budget): each wave owns 32 query rows, Q fragments stay in VGPRs;
//     K (K-contiguous sub-tiles) and V (contraction-strided tile) arrive by LDS-DMA, double-buffered, zero-filled at
//     the edges by the buffer descriptor.
#include "common.h"
#include "../../include/kai0hip.h"
#include <limits.h>
#include <stdlib.h>

namespace {

constexpr uint32_t OOB = 0x80000000u;
// timing ablations (phases switched off; results wrong) exist only in builds made with KAI0_HIPCC_FLAGS=-DKAI0_ABLATE
#ifdef KAI0_ABLATE
#define KAI0_ABL(p) ((p).ablate)
#else
#define KAI0_ABL(p) 0
#endif
constexpr int KC_LDS_MAX = 2048;  // key codes staged in LDS per block (8 KiB): covers S = 1018 and the estimator's 1786

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

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4v;

struct AttnArgs {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* V;
    bf16_t* O;
    bf16_t* P;
    const int32_t* qcode;
    const int32_t* kcode;
    int rows, Sk, HD, H, q0;
    int64_t ldq, ldk, ldv, ldo, ldp;
    int batch_inner;
    int64_t sQ1, sQ2, sK1, sK2, sV1, sV2, sO1, sO2, sP;
    int64_t qcode_ld, kcode_ld;
    float scale;
    int kc_lds_keys;  // > 0: the key codes of the launch's key range are staged in LDS once per block (round_up(Sk, 64) entries)
    float* lse;          // optional [batch][s_lse] f32: log-sum-exp of every query row's logits (+inf for rows that see no key)
    int64_t s_lse;
    int nt_p;    // P stored with the non-temporal hint
    int ablate;  // diagnostics only (KAI0_ATTN_ABLATE bit mask, timing runs): 1 no P store, 2 no pass 1, 4 no P V MFMAs,
                 // 8 no DMA inside the tile loops, 16 no logits MFMAs in pass 2 — results are wrong with any bit set
};

// NKS = number of 64-wide K sub-tiles (HD <= 64*NKS); VC = V tile columns (128 or 256); OMT = VC/16 output d-tiles
// QT = 16-row query tiles per wave.  2: four waves per 128-row block (one per SIMD); 1: eight waves (two per SIMD), so that one
// wave's MFMAs overlap the other's softmax VALU work and LDS reads — the block, its K / V tiles and LDS footprint are the same.
// OP > 0: single pass for Sk <= 64 * OP keys — all OP key tiles (K and V) are staged into LDS at once, the logits of the whole
// row stay in registers (OP x 4 x QT accumulators), so Q K^T is computed once and there is no per-tile barrier; OP = 0: the
// general two-pass form (pass 1: row max / sum, pass 2: recompute the logits, P, P V); OP = -1: one pass, online softmax (no P).
// RB = query rows per block (128; 64: four waves, ONE staging buffer, 80 KiB of LDS -> two independent blocks per CU that cover each
// other's DMA waits and softmax chains instead of eight lockstep waves).
template <int NKS, int VC, int QT, int OP = 0, int RB = 128>
__global__ __launch_bounds__(RB / (16 * QT) * 64, RB == 64 ? 2 : 1) void attn_fwd_kernel(const AttnArgs p) {
    constexpr int WAVES = RB / (16 * QT);
    constexpr int NST = RB == 64 ? 1 : 2;  // staging buffers of the two-pass form
    static_assert(RB == 128 && QT == 1, "128-row blocks of eight waves (the 64-row / four-wave forms were measured slower and removed)");
    constexpr int OMT = VC / 16;
    constexpr int KSTEPS = NKS * 2;                 // 32-wide contraction steps over the head dim
    constexpr int K_BYTES = NKS * 8192;             // NKS x [64 keys][64 d] bf16
    constexpr int V_ROWB = VC * 2;                  // bytes per key row of the V tile
    constexpr int V_BYTES = 64 * V_ROWB;
    constexpr int STAGE = K_BYTES + V_BYTES;
    constexpr int V_LPR = V_ROWB / 16;              // lanes per V row (16 or 32)
    constexpr int V_RPP = 64 / V_LPR;               // key rows per DMA piece (4 or 2)
    constexpr int NKP = NKS * 8 / WAVES;            // K DMA pieces per wave per tile
    constexpr int NVP = (64 / V_RPP) / WAVES;       // V DMA pieces per wave per tile
    static_assert(NKP >= 1 && NVP >= 1, "too many waves for this tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* pbuf_all = smem + (OP > 0 ? OP : NST) * STAGE;  // WAVES x [16 QT rows][64 keys] bf16 (P transposition scratch)
    int* kc_lds = reinterpret_cast<int*>(pbuf_all + WAVES * QT * 2048);  // key codes of all keys (see load_kcodes)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int z = blockIdx.y;
    const int z1 = z / p.batch_inner, z2 = z - z1 * p.batch_inner;
    const bf16_t* Qb = p.Q + z1 * p.sQ1 + z2 * p.sQ2;
    const bf16_t* Kb = p.K + z1 * p.sK1 + z2 * p.sK2;
    const bf16_t* Vb = p.V + z1 * p.sV1 + z2 * p.sV2;
    bf16_t* Ob = p.O + z1 * p.sO1 + z2 * p.sO2;
    bf16_t* Pb = p.P + (int64_t)z * p.sP;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Pb, 0, (int)OOB, 0x00020000);
    const int row0 = blockIdx.x * RB + wave * (16 * QT);   // first query row of this wave
    const int ntiles = (p.Sk + 63) / 64;
    const uint32_t ldk2 = (uint32_t)p.ldk * 2, ldv2 = (uint32_t)p.ldv * 2;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l15, g) holds Q[row][32*ks + 8g .. +8] ----------------
    // through a buffer descriptor: rows past the end / head-dim padding get an out-of-range offset and come back as zeros, so the
    // KSTEPS loads are unconditional and in flight together (a guarded load is compiled as a branch with its own wait)
    const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, (int)OOB, 0x00020000);
    bf16x8 qf[QT][KSTEPS];
#pragma unroll
    for (int nt = 0; nt < QT; ++nt) {
        const int r = row0 + nt * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d = ks * 32 + g * 8;
            const uint32_t off = (r < p.rows && d < p.HD) ? (uint32_t)r * (uint32_t)(p.ldq * 2) + (uint32_t)d * 2 : OOB;
            qf[nt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, (int)off, 0, 0));
        }
    }
    // per-lane query codes (mask): the query position of folded row r is q0 + r / H
    int qc[QT];
#pragma unroll
    for (int nt = 0; nt < QT; ++nt) {
        const int r = row0 + nt * 16 + l15;
        qc[nt] = INT_MAX;
        if (p.qcode != nullptr) qc[nt] = r < p.rows ? p.qcode[z1 * p.qcode_ld + p.q0 + r / p.H] : -1;
    }

    // ---- staging ---------------------------------------------------------------------------------------------------
    // K sub-tile j (d in [64j, 64j+64)): [64 keys][64 d], 128-B rows, DMA piece = 8 key rows; lane -> key row (lane>>3),
    //   slot lane&7 holding source chunk (lane&7)^(row&7).
    // V tile: [64 keys][VC], DMA piece = V_RPP key rows; lane -> row lane/V_LPR, slot lane%V_LPR holding source chunk
    //   slot ^ ((row&7)<<1)  (the tr reads below touch key rows 4g' .. 4g'+3 per 16-lane group: 8 distinct row&7).
    const int kc_chunk = ((lane & 7) ^ (lane >> 3)) * 8;
    auto stage = [&](int kt, int slot, bool with_v) {
        char* sk = smem + slot * STAGE + wave * (NKP * 1024);
        char* sv = smem + slot * STAGE + K_BYTES + wave * (NVP * 1024);
        const int key0 = kt * 64;
#pragma unroll
        for (int j = 0; j < NKP; ++j) {
            const int piece = wave * NKP + j;           // 0 .. NKS*8-1
            const int sub = piece >> 3, key = key0 + (piece & 7) * 8 + (lane >> 3);
            const int d = sub * 64 + kc_chunk;
            const uint32_t off = (key < p.Sk && d < p.HD) ? (uint32_t)key * ldk2 + (uint32_t)d * 2 : OOB;
            glds16(k_rsrc, off, sk + j * 1024);
        }
        if (with_v) {
#pragma unroll
            for (int j = 0; j < NVP; ++j) {
                const int r = (wave * NVP + j) * V_RPP + lane / V_LPR;
                const int c = (lane % V_LPR) ^ ((r & 7) << 1);
                const int key = key0 + r;
                const uint32_t off = (key < p.Sk && c * 8 < p.HD) ? (uint32_t)key * ldv2 + (uint32_t)c * 16 : OOB;
                glds16(v_rsrc, off, sv + j * 1024);
            }
        }
    };

    // S^T tile of this wave: [64 keys][16 QT queries] = 4 x QT MFMA tiles; A = K rows (LDS), B = Q (registers).
    // Fragment reads and MFMAs are issued in BATCHES with the reads one batch ahead (double-buffered kf): left to itself hipcc
    // schedules every ds_read right in front of the MFMA that consumes it and waits for it there (read, s_waitcnt, mfma — 64 times per
    // tile, each one exposing the LDS latency: the listing of round 3's kernels shows exactly that, and it is why their phases "added
    // up").  The sched_barriers pin the order; the compiler's own counted lgkmcnt then waits only for the older batch.
    // `fill(ks)` is called inside MFMA batch ks: independent VALU work the scheduler may interleave with the batch's MFMAs.
    auto logits_f = [&](const char* tk, f32x4 (&s)[4][QT], auto&& fill) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < QT; ++nt) s[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 kf[2][4];
        auto rd = [&](int ks, bf16x8 (&dst)[4]) {
            const char* sub = tk + (ks >> 1) * 8192;
            const int chunk = (ks & 1) * 4 + g;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = mt * 16 + l15;
                dst[mt] = *reinterpret_cast<const bf16x8*>(sub + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
        };
        rd(0, kf[0]);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + 1 < KSTEPS) rd(ks + 1, kf[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < QT; ++nt)
                    s[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks & 1][mt], qf[nt][ks], s[mt][nt], 0, 0, 0);
            fill(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto logits = [&](const char* tk, f32x4 (&s)[4][QT]) { logits_f(tk, s, [](int) {}); };
    // logits with the reference's rounding, masked: element (mt, nt, r) is key kt*64 + mt*16 + 4g + r, query column l15
    // key codes of tile kt for this lane's 16 keys (mt*16 + 4g + r): loaded BEFORE the tile's MFMAs so that the global
    // latency hides behind them (inside finish_logits they cost one exposed load latency per tile and pass)
    // The codes depend on the key only: each block copies the whole row into LDS once (the first barrier below publishes it) and a
    // tile takes its 16 codes with four 16-B LDS reads.  Straight from global they were 16 dword loads per lane per tile and pass,
    // whose latency the tile's 32 MFMAs do not cover (the forward's phases are latency-bound: KAI0_ATTN_ABLATE shows their costs
    // simply adding up).
    if (p.kc_lds_keys > 0) {
        for (int i = tid; i < p.kc_lds_keys; i += WAVES * 64)
            kc_lds[i] = p.kcode == nullptr ? 0 : (i < p.Sk ? p.kcode[z1 * p.kcode_ld + i] : INT_MAX);
    }
    auto load_kcodes = [&](int kt, int (&kc)[4][4]) {
        if (p.kc_lds_keys > 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const i32x4v c = *reinterpret_cast<const i32x4v*>(kc_lds + kt * 64 + mt * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) kc[mt][r] = c[r];
            }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int key = kt * 64 + mt * 16 + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                kc[mt][r] = p.kcode == nullptr ? 0 : ((key + r < p.Sk) ? p.kcode[z1 * p.kcode_ld + key + r] : INT_MAX);
        }
    };
    auto finish_logits = [&](int kt, f32x4 (&s)[4][QT], const int (&kc)[4][4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int key = kt * 64 + mt * 16 + 4 * g;
#pragma unroll
            for (int nt = 0; nt < QT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = rbf(rbf(s[mt][nt][r]) * p.scale);
                    const bool ok = (key + r < p.Sk) && (kc[mt][r] <= qc[nt]);
                    s[mt][nt][r] = ok ? v : -INFINITY;
                }
        }
    };

    f32x4 o[OMT][QT];
#pragma unroll
    for (int mt = 0; mt < OMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) o[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    char* pbuf = pbuf_all + wave * (QT * 2048);
    // one 64-key tile of the output side: final probabilities from the finished logits s (row max m, 1 / row sum inv_l),
    // P written out, O^T += V^T P^T
    // final probabilities of a 64-key tile from the finished logits s (row max m, 1 / row sum inv_l), rounded to bf16 exactly once
    // (what the reference multiplies V with), and written out
    auto make_p = [&](int kt, f32x4 (&s)[4][QT], const float (&m_run)[QT], const float (&inv_l)[QT], bf16x4 (&pb)[4][QT]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < QT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = (m_run[nt] > -INFINITY) ? __expf(s[mt][nt][r] - m_run[nt]) * inv_l[nt] : 0.f;
                    pb[mt][nt][r] = f2bf(e);
                }
        // P tile -> global through a wave-private LDS transposition: write [16 QT q][64 keys] rows, read 16 B per lane
        if (p.P != nullptr && !(KAI0_ABL(p) & 1)) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < QT; ++nt)
                    *reinterpret_cast<bf16x4*>(pbuf + (nt * 16 + l15) * 128 + (mt * 16 + 4 * g) * 2) = pb[mt][nt];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // buffer stores with an out-of-range offset for the lanes that have nothing to write: the instruction is ALWAYS
            // issued, so the wait at the end of the tile can count these 2 QT stores as the youngest memory operations and
            // need not sit out their acknowledgement (vmcnt counts stores too; a plain vmcnt(0) there cost ~1 us per tile)
#pragma unroll
            for (int it = 0; it < 2 * QT; ++it) {
                const int lr = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
                const int r = row0 + lr, key = kt * 64 + c8;
                const u32x4 pv = *reinterpret_cast<const u32x4*>(pbuf + lr * 128 + c8 * 2);
                const uint32_t off = (r < p.rows && key < p.ldp) ? (uint32_t)(((int64_t)r * p.ldp + key) * 2) : OOB;
                // P is read again only by the backward, a whole forward later: non-temporal (aux bit 1 = nt), so that the 0.5 GB
                // of a launch do not displace what the next kernels are about to read from L2 / Infinity Cache
                if (p.nt_p) __builtin_amdgcn_raw_buffer_store_b128(pv, p_rsrc, (int)off, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(pv, p_rsrc, (int)off, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    };
    // O^T += V^T P^T : contraction over the 64 keys in two 32-key steps; step kk uses key blocks 2kk and 2kk+1, lane
    // group g contributing keys {4g..4g+3} of each — the same 8 keys on both operands.  Batched like the logits: the transpose
    // reads of four output d-tiles (8 ds_read_b64_tr_b16) run one batch ahead of the MFMAs that consume them.
    auto pv_tile = [&](const char* tv, const bf16x4 (&pb)[4][QT]) {
        if (KAI0_ABL(p) & 4) return;
        constexpr int NB = OMT / 4;              // batches per 32-key step
        bf16x8 vf[2][4];
        auto rdv = [&](int b, bf16x8 (&dst)[4]) {
            const int kk = b / NB, m0 = (b % NB) * 4;
            const int r_lo = kk * 32 + 4 * g + (l15 >> 2);   // key row this lane addresses for the transpose read
            const int r_hi = r_lo + 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int chunk = (m0 + i) * 2 + ((l15 & 3) >> 1);
                const int sub = (l15 & 1) * 8;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (LDS_PTR(bf16x4))(tv + r_lo * V_ROWB + ((chunk ^ ((r_lo & 7) << 1)) << 4) + sub));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (LDS_PTR(bf16x4))(tv + r_hi * V_ROWB + ((chunk ^ ((r_hi & 7) << 1)) << 4) + sub));
                dst[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        bf16x8 pf[2][QT];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nt = 0; nt < QT; ++nt) pf[kk][nt] = __builtin_shufflevector(pb[2 * kk][nt], pb[2 * kk + 1][nt], 0, 1, 2, 3, 4, 5, 6, 7);
        rdv(0, vf[0]);
#pragma unroll
        for (int b = 0; b < 2 * NB; ++b) {
            if (b + 1 < 2 * NB) rdv(b + 1, vf[(b + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int kk = b / NB, m0 = (b % NB) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < QT; ++nt)
                    o[m0 + i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[b & 1][i], pf[kk][nt], o[m0 + i][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto emit_tile = [&](int kt, const char* tv, f32x4 (&s)[4][QT], const float (&m_run)[QT], const float (&inv_l)[QT]) {
        bf16x4 pb[4][QT];
        make_p(kt, s, m_run, inv_l, pb);
        pv_tile(tv, pb);
    };

    if constexpr (OP > 0) {
        // ============================ single pass (Sk <= 64 * OP): everything resident ================================
#pragma unroll
        for (int t = 0; t < OP; ++t)
            if (t < ntiles) stage(t, t, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        f32x4 sall[OP][4][QT];
        float m_run[QT], inv_l[QT];
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) m_run[nt] = -INFINITY;
#pragma unroll
        for (int t = 0; t < OP; ++t) {
            if (t < ntiles) {
                int kc[4][4];
                load_kcodes(t, kc);
                logits(smem + t * STAGE, sall[t]);
                finish_logits(t, sall[t], kc);
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < QT; ++nt) sall[t][mt][nt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
#pragma unroll
            for (int nt = 0; nt < QT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m_run[nt] = fmaxf(m_run[nt], sall[t][mt][nt][r]);
        }
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) {
            float m = m_run[nt];
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
            if (m > -INFINITY) {
#pragma unroll
                for (int t = 0; t < OP; ++t)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) l += __expf(sall[t][mt][nt][r] - m);
            }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            m_run[nt] = m;
            inv_l[nt] = l > 0.f ? 1.0f / l : 0.f;
            const int r = row0 + nt * 16 + l15;
            if (p.lse != nullptr && g == 0 && r < p.rows) p.lse[(int64_t)z * p.s_lse + r] = l > 0.f ? m + __logf(l) : INFINITY;
        }
#pragma unroll
        for (int t = 0; t < OP; ++t)
            if (t < ntiles) emit_tile(t, smem + t * STAGE + K_BYTES, sall[t], m_run, inv_l);
    } else if constexpr (OP < 0) {
        // ============================ one pass, online softmax: no probabilities leave the chip =======================
        // Per lane: query column l15 of tile nt, keys 16 mt + 4 g + r of every key tile.  The row max is made uniform over the
        // four lane groups of a column every tile (two shuffles), so all of them scale O and l by the same factors and the
        // partial row sums can simply be added at the end.
        float m_run[QT], l_run[QT];
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) { m_run[nt] = -INFINITY; l_run[nt] = 0.f; }
        // Key tiles in which NO key is visible to anybody (all 64 codes INT_MAX: padded prompt slots — 10-40 of pi0.5's 200 slots are
        // filled in kai0's tasks — or the range past Sk) are skipped: the block walks the list of live tiles.  A property of the
        // batch entry's key codes only, so the list is block-uniform; built once from the codes in LDS.
        int* lt = kc_lds + KC_LDS_MAX;  // [0] = number of live tiles, [1 + i] = i-th live tile, [40 + t] = flag of tile t
        const bool use_list = p.kc_lds_keys > 0;
        int nlive = ntiles;
        if (use_list) {
            __syncthreads();  // the codes are in LDS
            for (int t = wave; t < ntiles; t += WAVES) {
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
        }
        auto tile_at = [&](int i) -> int { return use_list ? __builtin_amdgcn_readfirstlane(lt[1 + i]) : i; };
        if (nlive > 0) stage(tile_at(0), 0, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        for (int it = 0; it < nlive; ++it) {
            const int kt = tile_at(it);
            const int buf = NST == 2 ? (it & 1) : 0;
            if (NST == 2 && it + 1 < nlive && !(KAI0_ABL(p) & 8)) stage(tile_at(it + 1), buf ^ 1, true);
            const char* tk = smem + buf * STAGE;
            int kc[4][4];
            load_kcodes(kt, kc);
            f32x4 s[4][QT];
            if (KAI0_ABL(p) & 16) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < QT; ++nt) s[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else
            logits(tk, s);
            finish_logits(kt, s, kc);
            bf16x4 pb[4][QT];
#pragma unroll
            for (int nt = 0; nt < QT; ++nt) {
                float tmax = -INFINITY;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[mt][nt][r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float mn = fmaxf(m_run[nt], tmax);
                const bool live = mn > -INFINITY;                                  // the row has seen a visible key
                const float alpha = live ? __expf(m_run[nt] - mn) : 1.0f;          // m_run = -inf: 0 (O and l are still 0)
                float lsum = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = (KAI0_ABL(p) & 32) ? s[mt][nt][r] : live ? __expf(s[mt][nt][r] - mn) : 0.f;    // masked: exp(-inf) = 0
                        lsum += e;
                        pb[mt][nt][r] = f2bf(e);
                    }
                l_run[nt] = l_run[nt] * alpha + lsum;
                m_run[nt] = mn;
                // O of this column is rescaled BEFORE the tile's P V is added (the whole pending state is at the old max)
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int mt = 0; mt < OMT; ++mt) o[mt][nt] *= alpha;
                }
            }
            pv_tile(tk + K_BYTES, pb);
            if (NST == 1) {
                lds_barrier();
                if (it + 1 < nlive) stage(tile_at(it + 1), 0, true);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) {
            float l = l_run[nt];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
            for (int mt = 0; mt < OMT; ++mt) o[mt][nt] *= inv;
            const int r = row0 + nt * 16 + l15;
            if (p.lse != nullptr && g == 0 && r < p.rows) p.lse[(int64_t)z * p.s_lse + r] = l > 0.f ? m_run[nt] + __logf(l) : INFINITY;
        }
    } else {
    // ================================ pass 1: row max and row sum =====================================================
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int nt = 0; nt < QT; ++nt) { m_run[nt] = -INFINITY; l_run[nt] = 0.f; }
    stage(0, 0, false);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int kt = 0; kt < ((KAI0_ABL(p) & 2) ? 0 : ntiles); ++kt) {
        const int buf = NST == 2 ? (kt & 1) : 0;
        if (NST == 2 && kt + 1 < ntiles && !(KAI0_ABL(p) & 8)) stage(kt + 1, buf ^ 1, false);
        int kc[4][4];
        load_kcodes(kt, kc);
        f32x4 s[4][QT];
        logits(smem + buf * STAGE, s);
        finish_logits(kt, s, kc);
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) {
            float tmax = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[mt][nt][r]);
            const float mn = fmaxf(m_run[nt], tmax);
            if (mn > -INFINITY) {
                float acc = l_run[nt] * __expf(m_run[nt] - mn);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc += __expf(s[mt][nt][r] - mn);
                l_run[nt] = acc;
                m_run[nt] = mn;
            }
        }
        if (NST == 1) {  // one buffer: every wave is done reading tile kt before tile kt + 1 overwrites it
            lds_barrier();
            if (kt + 1 < ntiles && !(KAI0_ABL(p) & 8)) stage(kt + 1, 0, false);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
    // combine the 4 lane groups (each saw the keys 4g..4g+3 of every 16-key block) of a query column
    float inv_l[QT];
#pragma unroll
    for (int nt = 0; nt < QT; ++nt) {
        float m = m_run[nt], l = l_run[nt];
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
            const float mn = fmaxf(m, mo);
            if (mn > -INFINITY) l = l * __expf(m - mn) + lo * __expf(mo - mn);
            m = mn;
        }
        m_run[nt] = m;
        inv_l[nt] = l > 0.f ? 1.0f / l : 0.f;
        const int r = row0 + nt * 16 + l15;
        if (p.lse != nullptr && g == 0 && r < p.rows) p.lse[(int64_t)z * p.s_lse + r] = l > 0.f ? m + __logf(l) : INFINITY;
    }

    // ================================ pass 2: P and O = P V ===========================================================
    stage(0, 0, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = NST == 2 ? (kt & 1) : 0;
        if (NST == 2 && kt + 1 < ntiles && !(KAI0_ABL(p) & 8)) stage(kt + 1, buf ^ 1, true);
        const char* tk = smem + buf * STAGE;
        int kc[4][4];
        load_kcodes(kt, kc);
        f32x4 s[4][QT];
        if (KAI0_ABL(p) & 16) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < QT; ++nt) s[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else
        logits(tk, s);
        finish_logits(kt, s, kc);
        emit_tile(kt, tk + K_BYTES, s, m_run, inv_l);
        if (NST == 1) {
            lds_barrier();
            if (kt + 1 < ntiles && !(KAI0_ABL(p) & 8)) stage(kt + 1, 0, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the P stores are older than these pieces)
        } else {
        // the DMA of tile kt+1 (issued at the top of this iteration) must have landed; the P stores issued after it may fly
        if (p.P != nullptr && !(KAI0_ABL(p) & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();
    }
    }
    // zero the padding columns [64*ntiles, ldp) of P (none when ldp <= 64*ntiles) — the tiles above already wrote
    // zeros for keys in [Sk, 64*ntiles)
    // ---- O: lane (q = l15, g) holds O^T rows d = 16 mt + 4g + r ----------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < QT; ++nt) {
        const int r = row0 + nt * 16 + l15;
        if (r >= p.rows) continue;
#pragma unroll
        for (int mt = 0; mt < OMT; ++mt) {
            const int d = mt * 16 + 4 * g;
            if (d < p.HD) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[mt][nt][e]);
                *reinterpret_cast<bf16x4*>(Ob + (int64_t)r * p.ldo + d) = ov;
            }
        }
    }
}

// Key-split attention, second half: `parts` one-pass launches of attn_fwd_kernel over disjoint key ranges left per range the
// range-normalised output O_s (bf16) and the range's log-sum-exp lse_s (+inf: the row saw no key in the range).  The softmax over all
// keys is the lse-weighted mean: O = sum_s w_s O_s, w_s = exp(lse_s - m) / sum_t exp(lse_t - m), m = max lse.  One thread = 8 head-dim
// columns of one row; f32 arithmetic, one rounding of O.  A row that saw no key in any range gets zeros (as the one-range kernel does).
__global__ __launch_bounds__(256) void attn_combine_kernel(const bf16_t* __restrict__ Op, const float* __restrict__ lse, bf16_t* __restrict__ O,
                                                           int parts, int rows, int HD, int64_t ldo, int64_t part_stride, int64_t lse_stride) {
    const int cpr = HD >> 3;  // 8-column chunks per row
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)rows * cpr) return;
    const int r = (int)(t / cpr), c = (int)(t - (int64_t)r * cpr) * 8;
    float l[8], m = -INFINITY;
    bf16x8 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < parts) {  // (parts <= 8: unrolled, every load of the thread in flight together)
            l[s] = lse[(int64_t)s * lse_stride + r];
            v[s] = *reinterpret_cast<const bf16x8*>(Op + (int64_t)s * part_stride + (int64_t)r * HD + c);
            if (l[s] < INFINITY) m = fmaxf(m, l[s]);
        }
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < parts) {
            const float w = (l[s] < INFINITY) ? __expf(l[s] - m) : 0.f;
            wsum += w;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w * bf2f(v[s][e]);
        }
    }
    const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * inv);
    *reinterpret_cast<bf16x8*>(O + (int64_t)r * ldo + c) = o;
}

}  // namespace

KAI0_API int kai0_attn_combine(const void* o_parts, const float* lse_parts, void* O, int parts, int rows, int HD, int64_t ldo,
                               int64_t part_stride, int64_t lse_stride, kai0_stream_t stream) {
    KAI0_REQUIRE(o_parts && lse_parts && O, "kai0_attn_combine: null operand");
    KAI0_REQUIRE(parts >= 1 && parts <= 8 && HD % 8 == 0 && HD > 0 && ldo % 8 == 0 && part_stride % 8 == 0 && lse_stride >= rows &&
                     ((uintptr_t)o_parts % 16) == 0 && ((uintptr_t)O % 16) == 0,
                 "kai0_attn_combine: 1..8 parts, HD %% 8 == 0, 16-byte aligned rows");
    if (rows <= 0) return 0;
    const int64_t n = (int64_t)rows * (HD / 8);
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o_parts,
                       lse_parts, (bf16_t*)O, parts, rows, HD, ldo, part_stride, lse_stride);
    return kai0_check_launch("kai0_attn_combine");
}

KAI0_API int kai0_attn_desc_size(void) { return (int)sizeof(kai0_attn_desc); }

KAI0_API int kai0_attn_fwd(const kai0_attn_desc* d, kai0_stream_t stream) {
    KAI0_REQUIRE(d != nullptr && d->Q && d->K && d->V && d->O, "kai0_attn_fwd: null operand");
    KAI0_REQUIRE(d->HD % 8 == 0 && d->HD > 0 && d->HD <= 256, "kai0_attn_fwd: HD=%d unsupported", d->HD);
    KAI0_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 4 == 0,
                 "kai0_attn_fwd: leading dims must be multiples of 8");
    KAI0_REQUIRE(d->P == nullptr || (d->ldp % 8 == 0 && d->ldp >= d->Sk && d->ldp <= ((d->Sk + 63) / 64) * 64),
                 "kai0_attn_fwd: ldp=%lld must be a multiple of 8 in [Sk, round_up(Sk, 64)]", (long long)d->ldp);
    KAI0_REQUIRE((d->qcode == nullptr) == (d->kcode == nullptr), "kai0_attn_fwd: qcode/kcode must both be set");
    KAI0_REQUIRE(d->H >= 1, "kai0_attn_fwd: H must be >= 1");
    KAI0_REQUIRE((int64_t)d->Sk * d->ldk * 2 < (int64_t)0x7FFF0000 && (int64_t)d->Sk * d->ldv * 2 < (int64_t)0x7FFF0000,
                 "kai0_attn_fwd: K/V span more than 2 GiB per batch entry");
    KAI0_REQUIRE(d->P == nullptr || (int64_t)d->rows * d->ldp * 2 < (int64_t)0x7FFF0000, "kai0_attn_fwd: P spans more than 2 GiB per batch entry");
    KAI0_REQUIRE((int64_t)d->rows * d->ldq * 2 < (int64_t)0x7FFF0000, "kai0_attn_fwd: Q spans more than 2 GiB per batch entry");
    if (d->rows <= 0 || d->Sk <= 0) return 0;
    AttnArgs p;
    p.Q = (const bf16_t*)d->Q; p.K = (const bf16_t*)d->K; p.V = (const bf16_t*)d->V;
    p.O = (bf16_t*)d->O; p.P = (bf16_t*)d->P;
    p.qcode = d->qcode; p.kcode = d->kcode;
    p.rows = d->rows; p.Sk = d->Sk; p.HD = d->HD; p.H = d->H; p.q0 = d->q0;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo; p.ldp = d->ldp;
    p.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
    p.sQ1 = d->sQ1; p.sQ2 = d->sQ2; p.sK1 = d->sK1; p.sK2 = d->sK2; p.sV1 = d->sV1; p.sV2 = d->sV2;
    p.sO1 = d->sO1; p.sO2 = d->sO2; p.sP = d->sP;
    p.qcode_ld = d->qcode_ld; p.kcode_ld = d->kcode_ld;
    p.scale = d->scale;
    p.lse = d->lse; p.s_lse = d->s_lse;
    KAI0_REQUIRE(d->lse == nullptr || d->s_lse >= d->rows, "kai0_attn_fwd: s_lse=%lld < rows", (long long)d->s_lse);
#ifdef KAI0_ABLATE
    static const int ablate = [] { const char* e = getenv("KAI0_ATTN_ABLATE"); return e ? atoi(e) : 0; }();
    p.ablate = ablate;
#else
    p.ablate = 0;
#endif
    p.nt_p = 1;
    const int kc_keys = ((d->Sk + 63) / 64) * 64;
    p.kc_lds_keys = kc_keys <= KC_LDS_MAX ? kc_keys : 0;  // longer key ranges read their codes from global
    const int batch = d->batch > 0 ? d->batch : 1;
    dim3 grid((d->rows + 127) / 128, batch, 1);
    hipStream_t s = (hipStream_t)stream;
#define KAI0_ATTN_LAUNCH(NKS, VC, QT, OP, RB)                                                                          \
    do {                                                                                                          \
        constexpr int LDS = (OP > 0 ? OP : (RB == 64 ? 1 : 2)) * (NKS * 8192 + 64 * VC * 2) + (RB / 32) * 4096 + KC_LDS_MAX * 4 + 512; \
        static_assert(LDS <= 160 * 1024, "attention LDS budget");                                                    \
        static bool attr_set = false;                                                                             \
        auto kern = attn_fwd_kernel<NKS, VC, QT, OP, RB>;                                                             \
        if (!attr_set) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
            KAI0_REQUIRE(e == hipSuccess, "kai0_attn_fwd: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e)); \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(RB / (16 * QT) * 64), LDS, s, p);                                    \
    } while (0)
    // (Measured and rejected, round 3: the two wave groups of pass 2 one barrier slot apart — 1.095 against 0.965 ms; four-wave
    // blocks with two 16-row tiles per wave — 0.83 against 0.66 ms one-pass; 64-row blocks, two per CU — 0.915 against 0.930 ms two-pass.
    // Round 4: the logits of tile kt + 1 computed ahead of the softmax of tile kt, K staged one tile ahead of V — 0.536 against
    // 0.526 ms; the same with the two waves of a SIMD walking the phases in different orders spilled 62 registers at HD = 256.
    // KAI0_ATTN_ABLATE on the one-pass loop: Q K^T 0.11, P V 0.11, LDS-DMA issue 0.10, everything else — mask / rounding / max / exp
    // VALU, two cross-lane shuffles, barrier, prologue and the row-per-lane O stores — 0.24 of the 0.53 ms, serial per wave.)
    // One pass with an online softmax whenever the caller does not ask for P (d->online: 0 = that rule, 1 = must, 2 = never).  For
    // <= 256 keys at HD <= 128 on small grids the resident single-pass form is exact AND one pass, so it stays.
    KAI0_REQUIRE(d->online != 1 || d->P == nullptr, "kai0_attn_fwd: the one-pass form does not produce P");
    const bool online = d->P == nullptr && d->online != 2;
    if (d->HD <= 128) {
        // 256 keys = 4 resident tiles (144 KiB, one block per CU): wins when the grid is at most a round or two of the chip
        if (d->Sk <= 256 && (int64_t)grid.x * grid.y <= 512) KAI0_ATTN_LAUNCH(2, 128, 1, 4, 128);
        else if (online) KAI0_ATTN_LAUNCH(2, 128, 1, -1, 128);
        else KAI0_ATTN_LAUNCH(2, 128, 1, 0, 128);
    } else {
        if (online) KAI0_ATTN_LAUNCH(4, 256, 1, -1, 128);
        else KAI0_ATTN_LAUNCH(4, 256, 1, 0, 128);
    }
#undef KAI0_ATTN_LAUNCH
    return kai0_check_launch("kai0_attn_fwd");
}
