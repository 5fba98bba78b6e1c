// attn_decode.hip — masked MQA attention of the denoise loop: a few hundred (token, head) query rows against the
// whole static KV cache (prefix + suffix keys), one launch.
//
// At B = 1 there are 50 tokens x 8 heads = 400 query rows and ~1018 keys of one shared KV head (HD = 256): 0.4 GFLOP
// but four dependent launches (logits GEMM, softmax, P V split-K, reduce) on the generic path.  gfx950 design:
//   * two launches, both a short chain of memory latencies with every load of a wave in flight at once:
//       logits : grid = 16-row query tiles x 4 key ranges; writes bf16(bf16(q k) * scale), -inf where hidden;
//       P V    : grid = query tiles x 4 head-dim ranges; f32 softmax of the stored row, P bf16, f32 P V
//     (a one-launch version — every block walking all keys — was measured at 36 us: 1 MB per block through one CU's
//     64 B/clk L2 port; splitting keys without the intermediate would change where P is rounded).
//     Rounding points are the reference's: logits bf16, f32 softmax over the full row, P bf16, f32 accumulate.
//   * no LDS staging and no transposes: operands are read from global directly in MFMA fragment layout.
//       S^T = K Q^T : A operand = K rows (16 B along HD), B operand = Q rows.  The 16 A-rows of the two tiles of a
//                     32-key group are assigned keys so that lane (q, g) ends up holding keys 8g..8g+7 of the group
//                     in its 2 x 4 accumulator registers — exactly the B fragment the second product needs;
//       O^T = V^T P^T : A operand = rows of the TRANSPOSED value cache Vt [HD][keys] (16 B along the keys), B
//                     operand = those 8 probabilities, straight from registers.
//     The value cache is therefore kept transposed for this kernel (the prefix pass transposes its rows once per
//     chunk, kai0_gemm_skinny_bf16 writes the suffix rows transposed).
#include "common.h"
#include "../../include/kai0hip.h"
#include <limits.h>

namespace {

struct DecArgs {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* Vt;
    bf16_t* O;
    const int32_t* qcode;
    const int32_t* kcode;
    int rows, H, Sk, q0, k_rows;
    int64_t q_bs, k_bs, k_ld, vt_bs, vt_ld, qc_ld, kc_ld;
    float scale;
    int range_major;  // 1: blockIdx.x = key range / head-dim slice, blockIdx.y = query tile (see kai0_attn_decode); 0: the former order
};

constexpr int DEC_HD = 256;
constexpr int DEC_KEYS = 1024;  // padded key range (32 groups of 32)
constexpr int DEC_KSPLIT = 4;   // logits kernel: key ranges per query tile (8 groups = 256 keys each, 2 groups per wave)
constexpr int DEC_HSPLIT = 4;   // P V kernel: head-dim ranges per query tile (64 each)

// Staging of a wave's operand rows through wave-private LDS.  The key / value caches are read by every query tile of a launch,
// i.e. out of L2, and the direct MFMA-fragment pattern (16 rows x 64 B per wave-instruction) gets 31-34 GB/s per CU there
// against 80-90 GB/s for contiguous 1-KiB runs (round-2 probe frag_load.hip, profiles/HISTORY.md).  Each wave therefore reads its rows as whole
// 512-B runs (two per instruction), parks them in its own LDS region (row stride 528 B) and takes its fragments from there
// with ds_read_b128; no block barrier is involved.
constexpr int DEC_ROWB = 512 + 16;            // LDS bytes per staged row (256 bf16 + 16 B: consecutive rows on distinct banks)
constexpr int DEC_STAGE = 64 * DEC_ROWB;      // one wave's 64 rows

__device__ __forceinline__ void dec_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- kernel 1: masked, scaled, bf16-rounded logits L[b][row][key] (-inf where hidden) ----------------------------
// NG = 32-key groups per wave: 2 (four key ranges of 256 per query tile) or 1 (eight ranges of 128: twice the blocks, half the rows
// each wave stages)
template <int NG>
__global__ __launch_bounds__(256) void dec_logits_kernel(const DecArgs p, bf16_t* __restrict__ L, int64_t l_bs) {
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, r0 = (p.range_major ? blockIdx.y : blockIdx.x) * 16, kr = p.range_major ? blockIdx.x : blockIdx.y;
    const int rq = r0 + i;
    const bool qok = rq < p.rows;
    char* my = dec_smem + wave * DEC_STAGE;
    // this wave's 64 keys (two 32-key groups) as 32 contiguous 1-KiB reads: instruction q covers key rows 2q, 2q + 1.  Rows 16-31
    // of each 32-row group are stored with their 16-B column index xor 4: the fragment reads below touch rows {0-3, 8-11, 16-19,
    // 24-27} of a group per 16 lanes, and with a 528-B row stride rows r and r + 16 would share their banks.
    const int key0 = (kr * 4 + wave) * NG * 32;
    const bf16_t* Kb = p.K + (int64_t)b * p.k_bs;
    {
        const int sub = lane >> 5, c16 = lane & 31;
        bf16x8 ch[16 * NG];
#pragma unroll
        for (int q = 0; q < 16 * NG; ++q) {
            const int row = 2 * q + sub;
            ch[q] = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)min(key0 + row, p.k_rows - 1) * p.k_ld + c16 * 8);
        }
#pragma unroll
        for (int q = 0; q < 16 * NG; ++q) {
            const int row = 2 * q + sub;
            *reinterpret_cast<bf16x8*>(my + row * DEC_ROWB + ((c16 ^ ((row & 16) >> 2)) << 4)) = ch[q];
        }
    }
    const bf16_t* Qb = p.Q + (int64_t)b * p.q_bs + ((int64_t)p.q0 * p.H + (qok ? rq : 0)) * DEC_HD + 8 * g;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = qok ? *reinterpret_cast<const bf16x8*>(Qb + 32 * c) : zero8;
    // qc: INT_MIN hides everything (row outside the problem); without codes every real key is visible
    const int qc = qok ? (p.qcode ? p.qcode[(int64_t)b * p.qc_ld + p.q0 + rq / p.H] : INT_MAX - 1) : INT_MIN;
    const int arow = 8 * (i >> 2) + (i & 3);  // key (within the group) fed to A-row i of tile 0; tile 1: + 4
    int kc[NG][8];
#pragma unroll
    for (int gl = 0; gl < NG; ++gl) {
        const int base = key0 + gl * 32;
#pragma unroll
        for (int e = 0; e < 8; ++e) {  // unconditional (clamped) loads: all in flight together
            const int key = base + 8 * g + e;
            const int code = p.kcode ? p.kcode[(int64_t)b * p.kc_ld + min(key, p.Sk - 1)] : INT_MIN;
            kc[gl][e] = key < p.Sk ? code : INT_MAX;
        }
    }
    dec_wave_sync();
#pragma unroll
    for (int gl = 0; gl < NG; ++gl) {
        const int base = key0 + gl * 32;
        const int ra = gl * 32 + arow, rb = ra + 4;
        const char* pa = my + ra * DEC_ROWB;
        const char* pb = my + rb * DEC_ROWB;
        const int xa = (ra & 16) >> 2, xb = (rb & 16) >> 2;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(pa + (((4 * c + g) ^ xa) << 4));
            const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(pb + (((4 * c + g) ^ xb) << 4));
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[c], a1, 0, 0, 0);
        }
        // lane (q = i, g) holds keys base + 8g + e: e < 4 in a0, e >= 4 in a1
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = rbf(rbf(e < 4 ? a0[e] : a1[e - 4]) * p.scale);
            o[e] = f2bf(kc[gl][e] <= qc ? v : -INFINITY);
        }
        *reinterpret_cast<bf16x8*>(L + (int64_t)b * l_bs + (int64_t)rq * DEC_KEYS + base + 8 * g) = o;
    }
}

// ---- kernel 2: f32 softmax of the stored logits, P bf16, O = P V for a 64-wide slice of the head dim ------------------
// HT = 16-wide head-dim tiles per block: 4 (four 64-wide slices per query tile) or 2 (eight 32-wide slices)
template <int HT>
__global__ __launch_bounds__(256) void dec_pv_kernel(const DecArgs p, const bf16_t* __restrict__ L, int64_t l_bs) {
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    float (*red)[16][64 + 4] = reinterpret_cast<float (*)[16][64 + 4]>(dec_smem + 4 * DEC_STAGE);
    float (*sstat)[4][16] = reinterpret_cast<float (*)[4][16]>(dec_smem + 4 * DEC_STAGE + sizeof(float) * 4 * 16 * 68);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, r0 = (p.range_major ? blockIdx.y : blockIdx.x) * 16, h0 = (p.range_major ? blockIdx.x : blockIdx.y) * (16 * HT);
    const int ngroups = (p.Sk + 31) >> 5;
    // wave w owns the 256 keys [256 w, 256 w + 256) = key groups 8 w .. 8 w + 7; lane (q, g) keys 8g..8g+7 of each group
    // logits of row r0 + i
    const bf16_t* Lr = L + (int64_t)b * l_bs + (int64_t)(r0 + i) * DEC_KEYS + wave * 256 + 8 * g;
    bf16x8 lf[8];
#pragma unroll
    for (int gl = 0; gl < 8; ++gl) lf[gl] = *reinterpret_cast<const bf16x8*>(Lr + gl * 32);
    // value rows h0 .. h0 + 63 of Vt, this wave's 256 keys: 64 runs of 512 B, two per instruction, parked in the wave's LDS
    // region (see dec_logits_kernel) — in flight during the statistics
    char* my = dec_smem + wave * DEC_STAGE;
    {
        const int sub = lane >> 5, c16 = lane & 31;
        const int kcol = min(wave * 256 + c16 * 8, (int)p.vt_ld - 8);  // (keys past the padded row end: never multiplied, P = 0)
        const bf16_t* Vb = p.Vt + (int64_t)b * p.vt_bs + (int64_t)h0 * p.vt_ld + kcol;
        bf16x8 ch[8 * HT];
#pragma unroll
        for (int q = 0; q < 8 * HT; ++q) ch[q] = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)(2 * q + sub) * p.vt_ld);
#pragma unroll
        for (int q = 0; q < 8 * HT; ++q) *reinterpret_cast<bf16x8*>(my + (2 * q + sub) * DEC_ROWB + (c16 << 4)) = ch[q];
    }
    float s[8][8];
    float m = -INFINITY;
#pragma unroll
    for (int gl = 0; gl < 8; ++gl)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[gl][e] = bf2f(lf[gl][e]);
            m = fmaxf(m, s[gl][e]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (g == 0) sstat[0][wave][i] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sstat[0][0][i], sstat[0][1][i]), fmaxf(sstat[0][2][i], sstat[0][3][i]));
    float sum = 0.f;
    if (m > -INFINITY) {
#pragma unroll
        for (int gl = 0; gl < 8; ++gl)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[gl][e] = expf(s[gl][e] - m);
                sum += s[gl][e];
            }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (g == 0) sstat[1][wave][i] = sum;
    __syncthreads();  // (also: every lane's staged value rows are in LDS — the ds_writes above were waited for by the barrier's lgkmcnt)
    sum = (sstat[1][0][i] + sstat[1][1][i]) + (sstat[1][2][i] + sstat[1][3][i]);
    const float inv = (m > -INFINITY && sum > 0.f) ? 1.0f / sum : 0.f;
    f32x4 acc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gl = 0; gl < 8; ++gl) {
        const bool live = wave * 8 + gl < ngroups;
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = f2bf((live && m > -INFINITY) ? s[gl][e] * inv : 0.f);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(my + (16 * t + i) * DEC_ROWB + ((4 * gl + g) << 4));
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, acc[t], 0, 0, 0);
        }
    }
    // acc[t]: lane (q = i, g) holds hd = h0 + 16t + 4g + reg
#pragma unroll
    for (int t = 0; t < HT; ++t) *reinterpret_cast<f32x4*>(&red[wave][i][16 * t + 4 * g]) = acc[t];
    __syncthreads();
    const int q = tid / (4 * HT), part = tid % (4 * HT);  // 16 rows x 4 HT pieces of 4 columns
    if (tid < 64 * HT && r0 + q < p.rows) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = f2bf((red[0][q][4 * part + e] + red[1][q][4 * part + e]) + (red[2][q][4 * part + e] + red[3][q][4 * part + e]));
        bf16_t* op = p.O + (int64_t)b * p.q_bs + ((int64_t)p.q0 * p.H + r0 + q) * DEC_HD + h0 + 4 * part;
        *reinterpret_cast<bf16x4*>(op) = o;
    }
}

constexpr int DEC_LOGITS_LDS = 4 * DEC_STAGE;
constexpr int DEC_PV_LDS = 4 * DEC_STAGE + (int)sizeof(float) * (4 * 16 * 68 + 2 * 4 * 16);

}  // namespace

KAI0_API int64_t kai0_attn_decode_workspace_bytes(int batch, int rows) {
    return (int64_t)batch * ((rows + 15) / 16 * 16) * DEC_KEYS * 2;
}

KAI0_API int kai0_attn_decode(const void* Q, const void* K, const void* Vt, void* O, const int32_t* qcode,
                              const int32_t* kcode, int batch, int rows, int H, int HD, int Sk, int q0, int64_t q_bs,
                              int64_t k_bs, int64_t k_ld, int k_rows, int64_t vt_bs, int64_t vt_ld, int64_t qcode_ld,
                              int64_t kcode_ld, float scale, void* workspace, int64_t workspace_bytes,
                              kai0_stream_t stream) {
    KAI0_REQUIRE(Q && K && Vt && O, "kai0_attn_decode: null operand");
    KAI0_REQUIRE(HD == DEC_HD, "kai0_attn_decode: HD=%d (only 256)", HD);
    KAI0_REQUIRE(Sk >= 1 && Sk <= DEC_KEYS, "kai0_attn_decode: Sk=%d (1..%d)", Sk, DEC_KEYS);
    KAI0_REQUIRE((qcode == nullptr) == (kcode == nullptr), "kai0_attn_decode: qcode/kcode must both be set");
    KAI0_REQUIRE(H >= 1 && k_ld % 8 == 0 && vt_ld % 8 == 0 && vt_ld >= ((Sk + 31) / 32) * 32 && k_rows >= Sk,
                 "kai0_attn_decode: k_ld/vt_ld must be multiples of 8, vt_ld >= round_up(Sk, 32), k_rows >= Sk");
    if (rows <= 0 || batch <= 0) return 0;
    const int64_t need = kai0_attn_decode_workspace_bytes(batch, rows);
    KAI0_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16) == 0,
                 "kai0_attn_decode: needs %lld workspace bytes (bf16 logits)", (long long)need);
    DecArgs p{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, qcode, kcode, rows, H, Sk, q0, k_rows,
              q_bs, k_bs, k_ld, vt_bs, vt_ld, qcode_ld, kcode_ld, scale};
    const int qt = (rows + 15) / 16;
    const int64_t l_bs = (int64_t)qt * 16 * DEC_KEYS;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dec_logits_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_LOGITS_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_logits_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_LOGITS_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_pv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_PV_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_pv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_PV_LDS);
        KAI0_REQUIRE(e == hipSuccess, "kai0_attn_decode: cannot reserve %d B of LDS: %s", DEC_PV_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    // few query tiles (B = 1: 25): eight key ranges / eight head-dim slices per tile instead of four, so that 200 blocks share
    // the staging of the caches instead of 100
    const int fine = 1;
    // Range-major grids: a workgroup's XCD is its linear id % 8 (observed with a round-2 probe; MI355X_MICROARCH.md says the same: block b runs on XCD b % 8), so with the key range / head-dim slice in
    // blockIdx.x (8 or 4 of them) every block that stages the same rows of the K / V cache runs on the same XCD and the rows cross
    // the fabric once per launch instead of once per XCD (query-tile-major: 5.9 / 10.9 MB fetched per launch for 0.7 / 1.3 MB of
    // operands, profiles/r03_infer_chunk_pmc.txt).
    const int rmaj = 1;
    p.range_major = rmaj;
    auto grid = [&](int ranges) { return rmaj ? dim3(ranges, qt, batch) : dim3(qt, ranges, batch); };
    if (fine && (int64_t)qt * batch <= 32) {
        hipLaunchKernelGGL(dec_logits_kernel<1>, grid(2 * DEC_KSPLIT), dim3(256), DEC_LOGITS_LDS, (hipStream_t)stream, p,
                           (bf16_t*)workspace, l_bs);
        hipLaunchKernelGGL(dec_pv_kernel<2>, grid(2 * DEC_HSPLIT), dim3(256), DEC_PV_LDS, (hipStream_t)stream, p,
                           (const bf16_t*)workspace, l_bs);
    } else {
        hipLaunchKernelGGL(dec_logits_kernel<2>, grid(DEC_KSPLIT), dim3(256), DEC_LOGITS_LDS, (hipStream_t)stream, p,
                           (bf16_t*)workspace, l_bs);
        hipLaunchKernelGGL(dec_pv_kernel<4>, grid(DEC_HSPLIT), dim3(256), DEC_PV_LDS, (hipStream_t)stream, p,
                           (const bf16_t*)workspace, l_bs);
    }
    return kai0_check_launch("kai0_attn_decode");
}
