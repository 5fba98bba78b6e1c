/* kai0hip.h — C ABI of libkai0hip.so: the MI355X (gfx950 / CDNA4) kernels under the pi0.5 hot path.
 *
 * The reference (OpenDriveLab/kai0, a fork of openpi) has NO C/FFI boundary on this path: every op below
 * is a `torch.nn.functional` call inside src/openpi/models_pytorch/{pi0_pytorch,gemma_pytorch}.py and the
 * patched transformers_replace layer library, lowered by torch to cuBLAS/cuDNN/Inductor.  This header is
 * the seam the new framework inserts: each entry cites the reference op it replaces.
 *
 * Conventions
 *   - plain C, `int` return: 0 = ok, <0 = error; message via kai0_last_error() (thread-local).
 *   - the caller (PyTorch-ROCm host code) owns every device buffer and passes raw device pointers,
 *     explicit shapes/strides (in ELEMENTS unless a name says bytes) and the hipStream_t to launch on.
 *   - no allocation, no host synchronisation inside: every entry point is hipGraph-capturable.
 *   - bf16 = IEEE bfloat16 (torch.bfloat16), f32 = float. Rounding to bf16 is round-to-nearest-even,
 *     applied at exactly the points where the reference's bf16-typed torch ops round.
 */
#ifndef KAI0HIP_H
#define KAI0HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* kai0_stream_t; /* hipStream_t */

const char* kai0_last_error(void);
int kai0_abi_version(void);
/* device properties probe: writes CU count, LDS bytes/CU, gcnArchName (<=63 chars) */
int kai0_device_info(int device, int* n_cu, int* lds_bytes, char* arch_name64);

/* ------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogue.  C[M,N] (+)= epilogue(A(M,K) * B(K,N))
 *
 * Replaces: every nn.Linear on the path — SigLIP q/k/v/out/fc1/fc2 (modeling_siglip.py:366-383,411,
 * 420-432), projector (modeling_paligemma.py:91-99), Gemma q/k/v/o/gate/up/down
 * (modeling_gemma.py:113-126,269-280; gemma_pytorch.py:172-174,222), the attention matmuls
 * QK^T and PV of eager_attention_forward (modeling_gemma.py:230-253, modeling_siglip.py:325-345),
 * and their autograd dgrad/wgrad.
 *
 * Operand storage: each operand is a row-major 2-D matrix in memory.
 *   a_kc=1: A stored [M][K] (K contiguous)          a_kc=0: A stored [K][M] (M contiguous)
 *   b_kc=1: B stored [N][K] (K contiguous; nn.Linear weight)   b_kc=0: B stored [K][N]
 * so (a_kc,b_kc) = (1,1) is y = x W^T, (1,0) is dgrad dx = dy W, (0,0) is wgrad dW = dy^T x.
 * All leading dimensions and K-/M-/N-contiguous extents must be multiples of 8 elements (16 B) and the
 * base pointers 16-B aligned; M, N, K themselves are arbitrary multiples of 8 along contiguous dims.
 *
 * Row remap (lets a flattened [B*rows] operand address a padded [B][S_pad] buffer without a copy):
 *   stored_row(r) = rpb ? (r / rpb) * bs + (r % rpb) + off : r
 * applied to the stored-row index of A / B (the M or N index when *_kc=1, the K index when *_kc=0)
 * and to the output row (C, pre_out and residual share c_*).
 *
 * Batching: grid z in [0,batch): z1 = z / batch_inner, z2 = z % batch_inner; operand X is offset by
 * z1*sX1 + z2*sX2 elements.  The strides are SIGNED 64-bit element counts added to the base pointer in 64-bit arithmetic:
 * a stride may be negative, and it may be the distance between two separate allocations (the joint-attention backward runs
 * dV = P^T dO and dK = dS^T Q as one launch whose outer B stride is `&Q - &dO`), as long as every entry z addresses valid,
 * 16-B aligned memory — multiples of 8 elements are required (checked), the range is the caller's responsibility.
 *
 * Epilogue, in this order, on the f32 accumulator v (every step that the reference performs as a bf16
 * torch op rounds to bf16 exactly there):
 *   v += bias[col]                       (bias bf16 or f32)
 *   v = bf16(v)                          (the Linear / matmul output)
 *   if scale != 1: v = bf16(v * scale)   (attention logits: modeling_gemma.py:243)
 *   if act == 1:   pre_out = v (optional); v = bf16(gelu_tanh(v))   (modeling_siglip.py:428-430)
 *   if act == 2:   GeGLU forward in the up-projection GEMM (modeling_gemma.py:122-126): v = u; pre_out = u;
 *                  v = bf16(bf16(gelu_tanh(g)) * u) with g = aux1[row][col]
 *   if act == 3:   GeGLU backward in the down-projection dgrad: v = dh; pre_out = du = bf16(dh * bf16(gelu_tanh(g)));
 *                  v = dg = bf16(bf16(dh * u) * gelu_tanh'(g)) with g = aux1, u = aux2
 *   if act == 5:   GELU backward in the dgrad that produces dh (SigLIP fc2 -> fc1): v = bf16(dh * gelu_tanh'(pre)), pre = aux1
 *   if gate:       v = bf16(v * gate[row / gate_rpb][col])          (_gated_residual, modeling_gemma.py:209-227)
 *   if residual:   v = bf16(v + residual[row][col])
 *   if accumulate: v = v + C_old  (f32 out: exact; bf16 out: rounded once)
 *   C = v (bf16, or f32 when out_f32; with out_f32 none of the bf16 rounding steps above is applied)
 */
typedef struct kai0_gemm_desc {
    const void* A;
    const void* B;
    void* C;
    int32_t M, N, K;
    int32_t a_kc, b_kc;
    int64_t lda, ldb, ldc;
    int32_t batch, batch_inner;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;
    int32_t a_rpb, b_rpb, c_rpb, _pad0;
    int64_t a_bs, a_off, b_bs, b_off, c_bs, c_off;
    const void* bias;
    int32_t bias_f32;
    float scale;
    int32_t act;
    int32_t out_f32;
    void* pre_out;
    const void* gate;
    int32_t gate_rpb;
    int32_t accumulate;
    int64_t gate_ld;
    const void* residual;
    int64_t ldr;
    int64_t sR1, sR2;
    /* split-K (few output tiles: long-contraction wgrads, skinny M = 50 inference GEMMs): split_k > 1 cuts K into
     * split_k chunks, each block writes an f32 partial tile into `workspace` (>= batch*split_k*M*N*4 bytes) and a
     * second kernel sums them and applies the same fused epilogue once.  Needs N % 8 == 0. */
    int32_t split_k;
    /* != 0: C (bf16) is stored with the non-temporal hint — for outputs nothing reads again soon (weight gradients: written during
     * the backward, read by the optimizer at the end of the step), so that they do not displace the operands of the following
     * launches from L2 / Infinity Cache.  Same bits in memory; a hint only. */
    int32_t c_nontemporal;
    void* workspace;
    int64_t workspace_bytes;
    const void* aux1; /* act 2/3: bf16 [rows][ldc], addressed like C */
    const void* aux2; /* act 3 */
    /* nseg > 0 (plain bf16 epilogue, batch 1): output columns [seg[i].n_begin, seg[i+1].n_begin) go to seg[i].dst
     * (leading dimension seg[i].ld, rows through the C row map) instead of C — one GEMM over stacked q|k|v weights
     * writing the padded q buffer and the K / V caches directly. */
    int32_t nseg, _pad2;
    struct {
        void* dst;
        int64_t ld;
        int32_t n_begin, _pad;
    } seg[3];
    /* act 4 — softmax backward fused into dP = dO V^T (modeling_gemma.py:243-248 / modeling_siglip.py:325-345 through
     * autograd): C = bf16( (P * (acc - D[row])) * scale ) with P = aux1 (bf16, addressed like C) and
     * D = rowvec[z1*rv_s1 + z2*rv_s2 + row*rv_ld] = sum_d dO[row][d] * O[row][d] (kai0_rowdot_bf16), which equals the
     * row's <dP, P>.  dP is taken from the f32 accumulator: it is never rounded and never written. */
    const float* rowvec;
    int64_t rv_s1, rv_s2, rv_ld;
    /* act 6 — the Gemma MLP's gate and up projections with GeGLU as ONE GEMM (modeling_gemma.py:113-126: `act_fn(gate_proj(x)) *
     * up_proj(x)`): B = gate_proj.weight, B2 = up_proj.weight (both [N][ldb] bf16, K-contiguous), N = the MLP width.  The kernel
     * stages the two weights interleaved in 32-row groups, so every wave holds gate and up pre-activations of the same output
     * columns and combines them in registers: C = h = bf16(bf16(gelu_tanh(g)) * u) with g = bf16(x Wg^T), u = bf16(x Wu^T) — the
     * rounding points of act 2 — and, when set, pre_out = g and pre_out2 = u (what the backward needs), all addressed like C.
     * Against gate GEMM + up GEMM with act 2: one launch, no read-back of g in the epilogue, and in inference no g / u writes. */
    const void* B2;
    void* pre_out2;
    /* split_k > 1 only: the norm that consumes C, fused into the reduction launch (B = 1 inference, where a launch is worth
     * 4-5 us).  norm_kind 1: RMSNorm norm_out = bf16((x * rstd) * (1 + w)), w f32 [N], rstd = rsqrt(mean(x^2) + eps)
     * (GemmaRMSNorm without cond, modeling_gemma.py:49-104); 2: LayerNorm norm_out = bf16((x - mean) * rstd * w + b), w / b bf16
     * [N] (nn.LayerNorm in modeling_siglip.py's encoder layers); x = the row of C as stored in bf16, statistics in f32;
     * norm_out [M][N] bf16, contiguous.  N % 8 == 0, N <= 2048, one batch entry, bf16 C, act 0, no gate / segments. */
    void* norm_out;
    const void* norm_w;
    const void* norm_b;
    float norm_eps;
    int32_t norm_kind;
    /* act 7 — RoPE fused into the epilogue of a stacked q | k | v projection (round 5; the rotation the reference applies right after
     * q_proj / k_proj: gemma_pytorch.py:185-195, apply_rotary_pos_emb modeling_gemma.py:149-194).  The partners of a rotation — columns
     * j and j + rope_half of a head — must meet in one 128-column tile, so the caller passes B with its rows PERMUTED inside the rotated
     * range: permuted column pc of [0, rope_n_end) is real column (pc / 256) * 256 + ((pc % 128) / 64) * 128 + 64 * ((pc % 256) / 128)
     * + pc % 64 (every tile = 64 first-half columns of a head followed by their 64 partners); columns >= rope_n_end are not rotated
     * and not permuted.  Epilogue: x = bf16(acc) (the Linear's output), then out1 = bf16(bf16(x1 cos) + bf16(-x2 sin)), out2 =
     * bf16(bf16(x2 cos) + bf16(x1 sin)) — bit for bit kai0_gemm_bf16 followed by kai0_rope_inplace — stored at the REAL columns
     * (seg[] boundaries are real columns).  rope_cos / rope_sin: bf16 [M][rope_half] (the values of kai0_rope_table; row = A row),
     * rope_half = 128, rope_n_end % 256 == 0.  K-contiguous operands, one batch entry, no split-K, plain bf16 output; runs on the
     * 128 x 128 configuration (meant for the B = 1 prefix pass, where the rotation's own launch cost more than its arithmetic). */
    const void* rope_cos;
    const void* rope_sin;
    int32_t rope_half, rope_n_end;
    /* A/B and test hooks, all 0 in production.  They are per CALL: the library keeps no process-wide mutable switches (SURVEY.md §8b
     * "no globals beyond a per-device handle" — what remains static is per-device set-up: kernel LDS attributes, the CU count and
     * the persistent kernel's self-cleaning counter slots).
     *   tile_cfg          force a tile / schedule configuration: 0 = automatic; 1 / 2 = 128 x 128 with 2 / 4 stages; 3 = 128 x 128 on eight
     *                     waves (4 stages); 4 = 256 x 256 plain loop; 5 = 256 x 256 two-buffer ping-pong for every layout
     *   persist           persistent NT kernel (one resident block per CU drawing 256 x 256 tiles from an atomic, XCD-grouped ticket queue;
     *                     the next tile's first half-tiles are staged before the current tile's epilogue; bit-identical to one block per
     *                     tile): 0 = the library's rule (K-contiguous one-entry GEMMs of >= 2048 tiles with N >= 8192 or K >= 8192),
     *                     1 = never, 2 = every eligible NT launch of >= 512 tiles
     *   general_epilogue  1 sends the 256 x 256 launches whose epilogue is a store with little else (act 0 / 1, optional bias / residual /
     *                     column routing; no gate, f32 output, scale or row map) through the general per-row epilogue instead of their
     *                     fast path — same bits (tests/test_kernels_gpu.py)
     *   small_w8          the 128 x 128 tile on eight waves (two per SIMD) instead of four: 0 = the library's rule (launches of at most
     *                     one block per CU with K-contiguous operands and act 0 / 1), 1 = never, 2 = every eligible 128 x 128 launch;
     *                     bit-identical to the four-wave tile */
    int32_t tile_cfg, persist, general_epilogue, small_w8;
} kai0_gemm_desc;

int kai0_gemm_bf16(const kai0_gemm_desc* d, kai0_stream_t stream);
/* sizeof(kai0_gemm_desc) as compiled: lets a foreign-language binding verify its struct mirror */
int kai0_gemm_desc_size(void);

/* ------------------------------------------------------------------------------------------------
 * Few-row weight-streaming GEMM for the denoise loop (B*action_horizon <= a few 64-row tiles):
 *   C[M, N] = A[M, K] @ W[N, K]^T, bf16 in, f32 accumulate, with the neighbouring element-wise ops fused.
 * Replaces, per expert layer and denoise step (gemma_pytorch.py:150-279, modeling_gemma.py:149-194,282-329):
 *   mode 1: q_proj | k_proj | v_proj (W = the three weights stacked) + apply_rotary_pos_emb on q and k, written
 *           straight into the padded q buffer / static K and V caches (column segments, each with its own
 *           destination and leading dimension; rows through the output row map);
 *   mode 0: o_proj / down_proj + `x + y * gate` gated residual (same rounding order as kai0_gemm_bf16);
 *   mode 2: gate_proj | up_proj (W = [gate ; up]) + GeGLU: h = bf16( bf16(gelu_tanh(bf16 g)) * bf16 u ).
 * One block computes 64 rows x (16 + 16) weight rows that are `pair_stride` apart, so a RoPE pair (d, d + HD/2) or a
 * GeGLU pair (gate_j, up_j) is combined in registers: pair_stride = HD/2 (mode 1), N/2 (mode 2), 16 (mode 0).
 * K / split_k must be 512 or 1024.  split_k > 1 (mode 0 only, no gate / residual) cuts K over blocks so that the
 * whole chip streams the weight: the kernel then writes the raw f32 partial products, `workspace` = f32
 * [split_k][M][N] (kai0_skinny_workspace_bytes), and kai0_adarms_combine — the gated residual + adaRMS that follows
 * o_proj / down_proj anyway — adds the splits in a fixed order (deterministic, no atomics).
 * rope_cos / rope_sin: f32 [M][rope_half] tables from kai0_rope_table (bf16-rounded cos/sin of pos * inv_freq,
 * exactly the values kai0_rope_inplace computes), indexed by the A row. */
typedef struct kai0_skinny_seg {
    void* dst;
    int64_t ld;
    int32_t n_begin, n_end; /* columns [n_begin, n_end) of the product go to dst[:, 0 : n_end - n_begin) */
    int32_t rope, _pad;     /* 0 plain, 1 rotate (RoPE), 2 plain but stored transposed: dst is [batch][n_end-n_begin][ld]
                             * and element (row, col) goes to dst[b][col][row'] with (b, row') from the output row map */
} kai0_skinny_seg;

typedef struct kai0_skinny_desc {
    const void* A;
    const void* W;
    int64_t lda, ldw;
    int32_t M, N, K;
    int32_t pair_stride, mode, split_k;
    int32_t a_rpb, c_rpb; /* row maps as in kai0_gemm_desc: row r -> (r / rpb) * bs + r % rpb + off; rpb = 0: identity */
    int64_t a_bs, a_off, c_bs, c_off;
    kai0_skinny_seg seg[3]; /* modes 0 and 2 use seg[0] only */
    int32_t nseg, gate_rpb;
    const void* gate; /* bf16 [M / gate_rpb][gate_ld] */
    int64_t gate_ld;
    const void* residual; /* bf16, addressed like the output */
    int64_t ldr;
    const float* rope_cos;
    const float* rope_sin;
    int32_t rope_half;
    /* 1 (split_k == -1 only): W is stored MFMA-fragment-major — for the 16-row tile t of output columns and the 32-wide step s
     * of the contraction one contiguous 1-KiB block at element offset (t * (K / 32) + s) * 512, holding W[16 t + i][32 s + 8 g + e] at
     * (i + 16 g) * 8 + e (i < 16, g < 4, e < 8) — so that every wave-instruction of the weight stream is one contiguous KiB
     * (what kai0_amd.ops.pack_skinny_weight produces; ldw is ignored). */
    int32_t w_packed;
    void* workspace;
    int64_t workspace_bytes;
    /* split_k == -1: the whole contraction inside one block (K / 256 waves, K in {1024, 2048, 4096}): no partial products, the
     * full epilogue in the same launch.  Modes 1 and 2 as above; mode 0 then tiles N in 16-column blocks (pair_stride ignored)
     * and the row tiles of one column tile share the weight slice through one XCD's L2.  With `mod` set (modes 1, 2; K == the
     * row width D) the A operand is adaRMS-normalised on the fly: y = bf16((x * rstd) * (1 + scale) + shift), rstd =
     * rsqrt(mean(x^2) + eps) over the row, scale = mod[b][0:D], shift = mod[b][D:2D], b = row / mod_rpb (modeling_gemma.py:49-104):
     * the norm between the residual stream and the projection needs no launch of its own. */
    const float* mod;
    int64_t mod_ld;
    int32_t mod_rpb;
    float eps;
    /* split_k == -1, the adaRMS norm FOLDED (round 4; inference, where the modulation is a constant of the engine: it depends on the
     * Euler schedule's time values and the weights only, modeling_gemma.py:49-104 with cond = time MLP(t)).  With the caller passing
     *   W' = bf16(W * (1 + scale)[k])  as W   and   cvec[n] = sum_k shift[k] * W[n][k],
     * y = ((x * rstd) * (1 + scale) + shift) W^T = rstd * (x W'^T) + cvec: the kernel multiplies the RAW residual stream and applies
     * the row factor and cvec to the f32 sum (modes 1, 2; K = 1024; `mod` must be NULL) — no row statistics pass over the tile and
     * no per-element normalisation in front of the MFMAs.  rstd[row] = rsqrt(sum_p rowsq_in[p][row] / K + eps), the partial sums of
     * squares written by the launch that produced x:
     *   rowsq_out (mode 0, in-block): this launch's own partials of the bf16 rows it stores, [N / 16][rowsq_out_ld] f32, one per
     *   16-column tile; kai0_denoise_glue writes one partial per row for the step's first layer.
     * The rounding points move (the weights are rounded after the scale instead of the activations after the norm): inside the
     * chunk tolerance of BASELINE.md section 4, not bit-identical to the `mod` form. */
    const float* rowsq_in;
    const float* cvec;
    int32_t rowsq_parts, _pad2;
    int64_t rowsq_ld;
    float* rowsq_out;
    int64_t rowsq_out_ld;
} kai0_skinny_desc;

int kai0_gemm_skinny_bf16(const kai0_skinny_desc* d, kai0_stream_t stream);
int kai0_skinny_desc_size(void);
int64_t kai0_skinny_workspace_bytes(int M, int N, int split_k);
/* Masked MQA attention of the denoise loop in one launch (modeling_gemma.py:224-259 eager_attention_forward with
 * the prefix-LM mask of pi0_pytorch.py:52-81, queries = the suffix tokens, keys = the whole static cache):
 *   Q, O : bf16 [batch][tokens][H][HD] (batch stride q_bs elements), query rows = tokens q0.., `rows` = n_tokens * H
 *   K    : bf16 [batch][k_rows][k_ld] row-major keys (one KV head);  Vt : bf16 [batch][HD][vt_ld] TRANSPOSED values
 *   allowed(b, token s, key j) = j < Sk && kcode[b][j] <= qcode[b][s]   (codes as in kai0_softmax_mask_fwd)
 *   logits bf16(bf16(q.k) * scale) -> f32 softmax -> P bf16 -> O = bf16(P V) : the rounding points of
 *   kai0_gemm_bf16 + kai0_softmax_mask_fwd + kai0_gemm_bf16.  HD = 256, Sk <= 1024, vt_ld >= round_up(Sk, 32). */
int kai0_attn_decode(const void* Q, const void* K, const void* Vt, void* O, const int32_t* qcode, const int32_t* kcode,
                     int batch, int rows, int H, int HD, int Sk, int q0, int64_t q_bs, int64_t k_bs, int64_t k_ld,
                     int k_rows, int64_t vt_bs, int64_t vt_ld, int64_t qcode_ld, int64_t kcode_ld, float scale,
                     void* workspace, int64_t workspace_bytes, kai0_stream_t stream);
/* Query side of the backward of the masked MQA attention with stored probabilities (modeling_gemma.py:230-253 through
 * autograd), one launch per call:
 *   D = rowsum(dO * O);  dP = dO V^T (f32, never written);  dS = bf16((P * (dP - D)) * scale);  dQ = bf16(dS K)
 * dO, O, dQ: bf16 rows of HD elements (row stride ldo; the H query heads of a position folded into the rows, as kai0_attn_fwd
 * takes Q); P, dS: bf16 [batch][rows][ldp] (the forward's probabilities in, dS out: dK = dS^T Q and dV = P^T dO stay GEMMs);
 * K, V: bf16 [batch][Sk][ld]; batch strides sO / sK / sV / sP in elements.  HD % 8 == 0, HD <= 256. */
int kai0_attn_bwd_dq(const void* dO, const void* O, const void* P, const void* K, const void* V, void* dS, void* dQ, int batch,
                     int rows, int Sk, int HD, int64_t ldo, int64_t ldk, int64_t ldv, int64_t ldp, int64_t sO, int64_t sK,
                     int64_t sV, int64_t sP, float scale, kai0_stream_t stream);
/* The same with the probabilities RECOMPUTED instead of read (round 4: the one-pass forward stores lse, not P — kai0_attn_desc.lse):
 *   S = bf16(bf16(Q K^T) * scale) masked with the codes exactly as kai0_attn_fwd does;  P = bf16(exp(S - lse[row]))
 * P is an OUTPUT here ([batch][rows][ldp], written once, columns >= Sk zero) for the dV = P^T dO GEMM that follows; everything
 * else as kai0_attn_bwd_dq.  Q has the layout of dO / O (rows of HD elements, row stride ldo, batch stride sO); folded row r is
 * head r % H of query position r / H (qcode index).  Sk <= 2048.  Replaces the backward of eager_attention_forward
 * (modeling_gemma.py:228-253) without a stored S x S tensor between forward and backward. */
typedef struct kai0_attn_bwd_desc {
    const void* dO; const void* O; const void* Q; const void* K; const void* V;
    const float* lse; const int32_t* qcode; const int32_t* kcode;
    void* P; void* dS; void* dQ;
    int32_t batch, rows, Sk, HD, H, _pad0;
    int64_t ldo, ldk, ldv, ldp, sO, sK, sV, sP, s_lse, qcode_ld, kcode_ld;
    float scale; int32_t _pad1;
} kai0_attn_bwd_desc;
int kai0_attn_bwd_dq2(const kai0_attn_bwd_desc* d, kai0_stream_t stream);
int kai0_attn_bwd_desc_size(void); /* sizeof(kai0_attn_bwd_desc): bindings check their mirror against it at load */
/* bytes of the bf16 logits scratch kai0_attn_decode needs */
int64_t kai0_attn_decode_workspace_bytes(int batch, int rows);
/* batched strided transpose: dst[z][c][r] = src[z][r][c], r < R, c < C (all extents / strides multiples of 8) */
int kai0_transpose_strided_bf16(const void* src, void* dst, int R, int C, int64_t src_ld, int64_t dst_ld, int batch,
                                int64_t src_bs, int64_t dst_bs, kai0_stream_t stream);

/* cos_out/sin_out[r][d] = bf16_round(cos/sin(inv_freq[d] * pos[r])), r < rows, d < half; stored as f32 (out_bf16 = 0) or as bf16
 * (out_bf16 = 1: the tables kai0_gemm_bf16's act 7 reads) — the same values either way */
int kai0_rope_table(const int32_t* pos, const float* inv_freq, void* cos_out, void* sin_out, int64_t rows, int half, int out_bf16,
                    kai0_stream_t stream);
/* Mask codes and position ids of one pi0.5 request in ONE launch (pi0_pytorch.py:52-81 make_att_2d_masks, :186-235 embed_prefix's
 * pad / att masks, :237-314 embed_suffix's, :343 position_ids = cumsum(pad) - 1), for the sequence
 *   [cam 0: n_img image tokens | ... | cam ncam-1 | T prompt tokens | Hs action tokens]:
 * pad = the camera's mask for its image tokens, the prompt mask for prompt tokens, true for action tokens; att = 0 over the prefix
 * and [1, 0, 0, ...] over the suffix, i.e. cumsum(att) = 0 / 1.  Outputs, int32 [B][S] (S = ncam n_img + T + Hs, row stride S):
 *   qcode = pad ? cumsum(att) : -1,  kcode = pad ? cumsum(att) : INT_MAX  (token j visible from i  <=>  kcode[j] <= qcode[i]),
 *   pos = cumsum(pad) - 1.
 * img_masks: ncam device pointers (host array) to bool [B]; lang_mask bool [B][T] contiguous.  Integer logic: bit-exact against the
 * torch restatement (kai0_amd.model.build_mask_codes).  ncam <= 8. */
int kai0_prefix_codes(const void* const* img_masks, int ncam, const void* lang_mask, int B, int n_img, int T, int Hs, int32_t* qcode,
                      int32_t* kcode, int32_t* pos, kai0_stream_t stream);

/* f32 MFMA GEMM, fully strided: C[m,n] = sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n]) (+ C).
 * split_k > 1 slices the contraction over grid.z into raw partial tiles workspace[split_k][M][N] (f32; size from
 * kai0_gemm_f32_workspace_bytes) which a second launch sums in slice order — deterministic, no atomics.
 * Replaces the f32 islands: patch-embed conv as im2col GEMM (modeling_siglip.py:220-226), adaRMS
 * `dense` (modeling_gemma.py:83-104), time MLP and action in/out projections
 * (pi0_pytorch.py:100-105,264-297,364-371) and their backward. */
int kai0_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                  float* C, int64_t ldc, int M, int N, int K, const float* bias, int accumulate, int split_k,
                  void* workspace, int64_t workspace_bytes, kai0_stream_t stream);
int64_t kai0_gemm_f32_workspace_bytes(int M, int N, int split_k);

/* Few-row f32 Linear, out[m][n] = sum_k x[m][k] W[n][k] + bias[n] for 1 <= M <= 16 (weight-streaming GEMV batch):
 * the time-MLP and adaRMS `dense` modulations of all denoise steps (M = steps x batch). */
int kai0_linear_rows_f32(const float* x, const float* W, const float* bias, float* out, int64_t ldo, int M, int N, int K,
                         kai0_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation.
 * RMSNorm (modeling_gemma.py:66-81): y = bf16( f32(x) * rsqrt(mean(x^2) + eps) * (1 + w) ), w f32.
 * adaRMS  (modeling_gemma.py:83-104): mod = [scale|shift|gate] f32 [B][3*D];
 *         y = bf16( xhat * (1 + scale) + shift ); gate_out = bf16(gate)  [B][D]
 * rows = number of token rows, rows_per_batch maps a row to its batch entry of `mod`.
 * rstd (f32 [rows]) is saved for backward.  */
int kai0_rmsnorm_fwd(const void* x, const float* w, void* y, float* rstd, int64_t rows, int D, float eps,
                     kai0_stream_t stream);
/* backward kernels take an optional `dres` (bf16 [rows][D]): the gradient arriving at x through its residual branch
 * is added into dx in the same pass (saves a separate add over the activation).
 * dw_partial: f32 [dw_blocks][D] (LayerNorm: [dwb_blocks][2 D] = dw | db) — one row per block, the block's waves are
 * reduced in LDS; kai0_reduce_partials sums the rows. */
int kai0_rmsnorm_bwd(const void* dy, const void* x, const float* w, const float* rstd, void* dx,
                     float* dw_partial, int dw_blocks, const void* dres, int64_t rows, int D, kai0_stream_t stream);
int kai0_adarms_fwd(const void* x, const float* mod, void* y, void* gate_out, float* rstd, int64_t rows,
                    int rows_per_batch, int D, float eps, kai0_stream_t stream);
int kai0_adarms_bwd(const void* dy, const void* dgate, const void* x, const float* mod, const float* rstd,
                    void* dx, float* dmod, const void* dres, int64_t rows, int rows_per_batch, int D,
                    kai0_stream_t stream);
/* Denoise-loop companion of kai0_gemm_skinny_bf16: finishes a split-K o_proj / down_proj and runs the next adaRMS.
 *   x = bf16( sum_s partials[s][row] ); x = bf16(x * gate_prev[b]) (if gate_prev); x = bf16(x + residual[row]) (if residual)
 *   x_out = x;  y, gate_out = adaRMS(x, mod) exactly as kai0_adarms_fwd.   (gemma_pytorch.py:150-279 `_gated_residual`) */
int kai0_adarms_combine(const float* partials, int splits, int64_t split_stride, const void* gate_prev,
                        const void* residual, void* x_out, const float* mod, void* y, void* gate_out, int64_t rows,
                        int rows_per_batch, int D, float eps, kai0_stream_t stream);
/* LayerNorm over the last dim (modeling_siglip.py:439-441,756): bf16 x, bf16 w/b, f32 statistics. */
int kai0_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                       int64_t rows, int D, float eps, kai0_stream_t stream);
int kai0_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                       void* dx, float* dwb_partial, int dwb_blocks, const void* dres, int64_t rows, int D,
                       kai0_stream_t stream);
/* column sums of per-block partials (f32, `blocks` rows of stride ld, ncols columns) -> out (bf16 or f32) */
int kai0_reduce_partials(const float* partial, int blocks, int ncols, int64_t ld, void* out, int out_f32,
                         kai0_stream_t stream);
/* the same for n (partials, destination) pairs, 32 per launch: the training backward queues the final sums of its norm-weight
 * and bias gradients (254 launches of ~9 us per pi0.5 step) and runs them together before the gradients are read */
typedef struct kai0_reduce_item {
    const float* partial; /* f32 [blocks][ld] */
    void* out;            /* [ncols] bf16 or f32 */
    int64_t ld;
    int32_t blocks, ncols;
    int32_t out_f32, _pad;
} kai0_reduce_item;
int kai0_reduce_partials_batch(const kai0_reduce_item* items, int n, kai0_stream_t stream);
/* first half of kai0_colsum_bf16: per-block partial column sums into scratch [*blocks_used][N] (f32), to be summed by
 * kai0_reduce_partials / kai0_reduce_partials_batch */
int kai0_colsum_partials_bf16(const void* dy, int64_t M, int N, int64_t ld, float* scratch, int scratch_blocks,
                              int* blocks_used, kai0_stream_t stream);
/* bias gradient: out[n] = sum_m dy[m][n]  (dy bf16 [M][ld], out bf16 or f32) */
int kai0_colsum_bf16(const void* dy, int64_t M, int N, int64_t ld, float* scratch, int scratch_blocks,
                     void* out, int out_f32, kai0_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RoPE, half-split layout (modeling_gemma.py:149-194): x is [B][S_ld][H][HD] bf16, in place;
 * rows row0..row0+S of every batch entry are rotated; pos int32 [B][S]; inv_freq f32 [HD/2] is the host's
 * 10000^(-2i/HD) table (ROPE_INIT_FUNCTIONS["default"]); cos/sin are computed in f32, rounded to bf16, then
 * out = bf16( bf16(x*cos) + bf16(rot(x)*sin) ).  inverse=1 applies the transpose (backward). */
int kai0_rope_inplace(void* x, const int32_t* pos, const float* inv_freq, int B, int S, int64_t s_ld_rows,
                      int64_t row0, int H, int HD, int inverse, kai0_stream_t stream);
/* kai0_rope_inplace on two tensors that share the positions — q [B][s_ld_rows][H][HD] and k [B][s_ld_rows][H2][HD] of one
 * attention layer (modeling_gemma.py:172-194 rotates both with the same cos / sin): one launch, the trigonometry once. */
int kai0_rope_inplace2(void* x, int H, void* x2, int H2, const int32_t* pos, const float* inv_freq, int B, int S,
                       int64_t s_ld_rows, int64_t row0, int HD, kai0_stream_t stream);
/* The same rotation out of place and fully strided (elements): dst[b][s] = rope(src[b][s], pos[b * pos_bs + s]) for B x S
 * rows of H heads.  Used to scatter a segment's q / k into the joint [B][S_total] attention buffers (and to gather the
 * gradients back, inverse = 1) with the rotation applied on the way (gemma_pytorch.py:181-195). */
int kai0_rope_copy(const void* src, void* dst, const int32_t* pos, const float* inv_freq, int B, int S, int H, int HD,
                   int64_t src_bs, int64_t src_ld, int64_t dst_bs, int64_t dst_ld, int64_t pos_bs, int inverse,
                   kai0_stream_t stream);

/* Masked row softmax for the prefix-LM mask (pi0_pytorch.py:52-81,156-159; modeling_gemma.py:243-248).
 * scores bf16 [B][Sq*H][ld] already scaled; row r of batch b is query s = q0 + r / H.
 * allowed(b,s,j) = kcode[b][j] <= qcode[b][s] && j < Sk, where kcode = pad ? cumsum(att) : INT_MAX and
 * qcode = pad ? cumsum(att) : -1 (the host builds both: integer logic, bit-exact with make_att_2d_masks).
 * qcode == NULL means "no mask" (SigLIP).  probs = bf16(softmax_f32(scores)); masked columns and the
 * padding columns [Sk, ld) are written as 0.  */
int kai0_softmax_mask_fwd(const void* scores, void* probs, const int32_t* qcode, const int32_t* kcode,
                          int B, int Sq, int H, int Sk, int64_t ld, int64_t batch_stride, int q0,
                          int64_t qcode_ld, int64_t kcode_ld, kai0_stream_t stream);
/* Backward of SigLIP's unmasked multi-head attention for one layer, one block per (image, head)
 * (modeling_siglip.py:325-345 through autograd; S = 256 tokens, head_dim = 72 — the so400m/14 @ 224 tower):
 *   q, k, v, dO, O : bf16 [n_img*S][NH*HD] (head h at column h*HD);  P : bf16 [n_img*NH][S][ldp] from kai0_attn_fwd
 *   dq, dk, dv : bf16 rows of stride ld_grad (0 = NH*HD; 3*NH*HD when the three are the column slices of one [rows][3*NH*HD]
 *                buffer, which lets the fused q|k|v projection backward read them as a single operand)
 *   D = rowsum(dO * O);  dS = bf16((P * (dO V^T - D)) * scale)  (dP in f32, never stored);
 *   dQ = bf16(dS K), dK = bf16(dS^T Q), dV = bf16(P^T dO).  Same rounding points as the GEMM formulation
 *   (kai0_gemm_bf16 act 4 + three batched GEMMs) it replaces. */
int kai0_siglip_attn_bwd(const void* q, const void* k, const void* v, const void* dO, const void* O, const void* P,
                         void* dq, void* dk, void* dv, int n_img, int S, int NH, int HD, int64_t ldp, int64_t ld_grad,
                         float scale, kai0_stream_t stream);
/* The same without stored probabilities (round 4): P = bf16(exp(bf16(bf16(q k^T) * scale) - lse[row])) is recomputed inside the
 * block from the q / k tiles that are in LDS anyway; lse: f32 [n_img * NH][S] from kai0_attn_fwd (kai0_attn_desc.lse). */
/* SigLIP's attention forward at the real tower's shape (S = 256 tokens, head_dim = 72; modeling_siglip.py:325-345), one block per
 * (image, head) with the head's K and V unpadded in LDS: logits = bf16(bf16(q k^T) * scale), EXACT softmax in f32 (the whole row is
 * on chip), P = bf16(softmax), O = bf16(P V) — the reference's rounding order, in one pass, no P written.
 *   q, k, v: bf16, row r of image n at q + n * sq + r * ldq, head h at column h * 72 (the three may be column slices of one stacked
 *   q|k|v buffer); o likewise (so, ldo); lse: optional f32 [n_img * NH][256] for kai0_siglip_attn_bwd2. */
int kai0_siglip_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int n_img, int S, int NH, int HD,
                         int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t sq, int64_t sk, int64_t sv, int64_t so,
                         float scale, kai0_stream_t stream);
int kai0_siglip_attn_bwd2(const void* q, const void* k, const void* v, const void* dO, const void* O, const float* lse,
                          void* dq, void* dk, void* dv, int n_img, int S, int NH, int HD, int64_t ld_grad, float scale,
                          kai0_stream_t stream);
/* out[r] = sum_d a[r][d] * b[r][d] in f32 (rows of D contiguous bf16 elements, D % 8 == 0): the softmax-backward row term */
int kai0_rowdot_bf16(const void* a, const void* b, float* out, int64_t rows, int D, kai0_stream_t stream);
/* dscores = bf16( (probs * (dprobs - sum_j dprobs*probs)) * scale ).  dprobs is bf16 or (dprobs_f32) f32:
 * with near-uniform attention dprobs - <dprobs,probs> cancels catastrophically, so the training path keeps
 * dP = dO V^T in f32 (the reference's autograd rounds it to bf16; this is the more accurate of the two). */
int kai0_softmax_bwd(const void* probs, const void* dprobs, int dprobs_f32, void* dscores, int64_t rows, int Sk,
                     int64_t ld, float scale, kai0_stream_t stream);

/* Fused attention forward (modeling_gemma.py:230-253; modeling_siglip.py:325-345):
 *   logits = bf16(bf16(Q K^T) * scale), masked as in kai0_softmax_mask_fwd, P = bf16(softmax_f32(logits)), O = bf16(P V).
 * Query rows are "folded": row r of a batch entry is head r % H of query position q0 + r / H (H = 1 for plain
 * multi-head attention addressed through the two-level batch strides).  P (optional, bf16 [rows][ldp] per batch entry,
 * ldp in [Sk, round_up(Sk,64)], padding columns zero) is written for the GEMM-based backward.  HD <= 256. */
typedef struct kai0_attn_desc {
    const void* Q; const void* K; const void* V; void* O; void* P;
    const int32_t* qcode; const int32_t* kcode;
    int32_t rows, Sk, HD, H, q0, batch, batch_inner, _pad0;
    int64_t ldq, ldk, ldv, ldo, ldp;
    int64_t sQ1, sQ2, sK1, sK2, sV1, sV2, sO1, sO2, sP;
    int64_t qcode_ld, kcode_ld;
    float scale;
    int32_t online;   /* 0: one pass with an online softmax when P == NULL (training: the backward recomputes P from lse), the exact
                       * two-pass form when P is asked for; 1: one pass required (P must be NULL); 2: never */
    float* lse;       /* optional [batch][s_lse] f32: log-sum-exp of every query row's (scaled, masked) logits, +inf for a row
                       * that sees no key.  exp(logit - lse) is the row's softmax: what kai0_attn_bwd_dq2 /
                       * kai0_siglip_attn_bwd2 recompute instead of reading P */
    int64_t s_lse;    /* batch stride of lse (>= rows) */
} kai0_attn_desc;
int kai0_attn_fwd(const kai0_attn_desc* d, kai0_stream_t stream);
/* Key-split attention (round 5; the B = 1 prefix pass of the action chunk: 61 row blocks cannot fill 256 CUs, 4 key ranges x 61 can).
 * The caller runs kai0_attn_fwd once with `parts` key ranges as batch entries (batch = parts, batch_inner = 1: sQ1 = 0 and qcode_ld = 0
 * share the queries, sK1 / sV1 / kcode_ld = the range length step K / V / the key codes, P = NULL, lse set) into o_parts
 * [parts][rows][HD] bf16 and lse_parts [parts][lse_stride] f32; this call merges them: O[r] = sum_s w_s O_s[r] with
 * w_s = exp(lse_s[r] - max) / sum_t exp(lse_t[r] - max) — the softmax over all keys (eager_attention_forward,
 * modeling_gemma.py:230-253) evaluated range by range; f32 arithmetic, one bf16 rounding of O on top of the parts' own.  A row that
 * saw no key in any range gets zeros.  O: row stride ldo (elements); parts <= 8; HD % 8 == 0. */
int kai0_attn_combine(const void* o_parts, const float* lse_parts, void* O, int parts, int rows, int HD, int64_t ldo,
                      int64_t part_stride, int64_t lse_stride, kai0_stream_t stream);
int kai0_attn_desc_size(void); /* sizeof(kai0_attn_desc) */

/* ------------------------------------------------------------------------------------------------
 * Elementwise pieces.
 * GeGLU (modeling_gemma.py:122-126): h = bf16( bf16(gelu_tanh(g)) * u ) */
int kai0_geglu_fwd(const void* g, const void* u, void* h, int64_t n, kai0_stream_t stream);
int kai0_geglu_bwd(const void* dh, const void* g, const void* u, void* dg, void* du, int64_t n,
                   kai0_stream_t stream);
/* dx = bf16(dy * gelu_tanh'(pre)) (SigLIP fc1 activation backward) */
int kai0_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, kai0_stream_t stream);
/* y = silu(x), f32 (time MLP, pi0_pytorch.py:289-297) */
int kai0_silu_fwd_f32(const float* x, float* y, int64_t n, kai0_stream_t stream);
int kai0_silu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, kai0_stream_t stream);
/* gated residual (modeling_gemma.py:209-227): out = bf16( x + bf16(y * gate[row / rows_per_batch]) ) */
int kai0_gated_fwd(const void* x, const void* y, const void* gate, void* out, int64_t rows, int rows_per_batch,
                   int D, kai0_stream_t stream);
/* gated residual backward: for out = x + y*gate[b]:
 *   dy = bf16(dout * gate[b]);  dgate[b][c] = bf16( sum_rows_in_b dout*y )  (f32 accumulation) */
int kai0_gated_bwd(const void* dout, const void* y, const void* gate, void* dy, void* dgate, int64_t rows,
                   int rows_per_batch, int D, kai0_stream_t stream);
/* token embedding gather * sqrt(D) (gemma_pytorch.py:88-89; pi0_pytorch.py:213-216):
 * out[b][row0 + t] = bf16( table[tok[b][t]] * scale ); out rows have stride out_ld, batch stride out_bs */
int kai0_embed_gather(const void* table, const int64_t* tokens, void* out, int B, int T, int D, float scale,
                      int64_t out_bs, int64_t out_row0, int64_t out_ld, kai0_stream_t stream);
/* backward, deterministic: dtable[tok] = bf16( sum over occurrences of bf16(dout*scale) ) for every token id
 * that occurs; rows of ids that do not occur are left untouched (caller zero-fills the dense gradient). */
int kai0_embed_grad(const void* dout, const int64_t* tokens, void* dtable, int B, int T, int D, float scale,
                    int64_t dout_bs, int64_t dout_row0, int64_t dout_ld, kai0_stream_t stream);
/* casts and adds */
int kai0_cast_f32_to_bf16(const float* x, void* y, int64_t n, kai0_stream_t stream);
int kai0_cast_bf16_to_f32(const void* x, float* y, int64_t n, kai0_stream_t stream);
int kai0_add_bf16(const void* a, const void* b, void* out, int64_t n, kai0_stream_t stream);
int kai0_add_f32(const float* a, const float* b, float* out, int64_t n, kai0_stream_t stream);
/* dst[C][R] = src[R][C]^T (bf16, R and C multiples of 8).  Used to turn dgrad (dx = dy W) into the faster
 * K-contiguous GEMM form: W^T is 0.2 % of the bytes the GEMM streams when M = B*S is large. */
int kai0_transpose_bf16(const void* src, void* dst, int R, int C, kai0_stream_t stream);
/* Sampled content checksum: out_dev[0] += sum over every `stride`-th 64-byte unit (position-mixed) of the n buffers listed in the
 * DEVICE array items_dev (pointers 16-byte aligned, nbytes >= 64).  out_dev is zeroed by the caller.  The inference engine stamps
 * the source tensors of its derived weight copies with it once per action chunk, to notice in-place weight edits that bypass
 * autograd's version counters (the engine-invalidation contract of INTEGRATION.md, now checked). */
typedef struct kai0_ck_item {
    const void* ptr;
    int64_t nbytes;
} kai0_ck_item;
int kai0_sampled_checksum(const kai0_ck_item* items_dev, int n, int stride, unsigned long long* out_dev, kai0_stream_t stream);
/* up to 12 strided row moves in one launch: part i moves B x rows rows of `cols` bf16 elements (strides in elements),
 * mode 0 = copy, 1 = RoPE per head of HD elements with position pos[b * pos_bs + pos_off + r], 2 = the inverse rotation,
 * 3 = zero fill (src ignored).  The joint-attention assembly of a layer (gemma_pytorch.py:181-219: concatenate the experts'
 * q / k / v over the sequence, rotate q and k) and its way back are one launch each. */
typedef struct kai0_pack_part {
    const void* src;
    void* dst;
    int64_t src_bs, src_ld, dst_bs, dst_ld;
    int32_t B, rows, cols, mode;
    int32_t pos_off, _pad;
} kai0_pack_part;
int kai0_pack_rows(const kai0_pack_part* parts, int n, const int32_t* pos, int64_t pos_bs, const float* inv_freq, int HD,
                   kai0_stream_t stream);
/* strided 2-D copy of bf16 rows: dst[b][dst_row0 + r][0:D] = src[b][src_row0 + r][0:D] */
int kai0_copy_rows_bf16(const void* src, void* dst, int B, int rows, int D, int64_t src_bs, int64_t src_row0,
                        int64_t src_ld, int64_t dst_bs, int64_t dst_row0, int64_t dst_ld,
                        kai0_stream_t stream);
/* im2col for the 14x14/stride-14 patch conv (modeling_siglip.py:220-226):
 * img f32 [N][3][224][224] -> cols f32 [N*256][588], k = c*196 + ky*14 + kx (Conv2d weight order) */
int kai0_patch_im2col(const float* img, float* cols, int n_img, int C, int HW, int P, kai0_stream_t stream);
/* out = bf16(x + pos[row % n_pos]) : patch embeddings + position embedding, cast to bf16
 * (modeling_siglip.py:271-281, 777-778) */
int kai0_add_pos_cast(const float* x, const float* pos, void* out, int64_t rows, int n_pos, int D,
                      kai0_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Flow matching glue (pi0_pytorch.py:326-328,373,401-419).
 * x_t = t*noise + (1-t)*a ; u_t = noise - a                         (f32, [B][HA]) */
int kai0_flow_mix(const float* noise, const float* actions, const float* time, float* x_t, float* u_t, int B,
                  int HA, kai0_stream_t stream);
/* time embedding (pi0_pytorch.py:25-42): out[b] = [sin(w_i t_b) | cos(w_i t_b)], f64 math, f32 result */
int kai0_time_sincos(const float* time, float* out, int B, int dim, double min_period, double max_period,
                     kai0_stream_t stream);
/* loss = (u - v)^2 ; dv = -2 (u - v) * dloss                          (f32) */
int kai0_mse_fwd(const float* u, const float* v, float* loss, int64_t n, kai0_stream_t stream);
int kai0_mse_bwd(const float* u, const float* v, const float* dloss, float* dv, int64_t n,
                 kai0_stream_t stream);
/* Euler step x += dt * v (f32) */
int kai0_euler_step(float* x, const float* v, float dt, int64_t n, kai0_stream_t stream);
/* The seam between two Euler steps of `sample_actions` in one launch (pi0_pytorch.py:401-461; denoise_step :421-461 +
 * embed_suffix :237-314): with xs != NULL it CLOSES a step — y = adaRMS(xs, mod) exactly as kai0_adarms_fwd (mod rows
 * [scale | shift | gate] of leading dimension mod_ld, one per rows_per_batch rows), v_t = action_out_proj(f32(y)) (w_out [A][D],
 * b_out [A], f32), x_t[rows][A] += dt * v_t in place — and with xs_next != NULL it OPENS the next one — xs_next[rows][D] =
 * bf16(action_in_proj(x_t)) (w_in [D][A], b_in [D], f32).  Replaces kai0_adarms_fwd + cast + kai0_gemm_f32 + kai0_euler_step +
 * kai0_gemm_f32 + cast; A <= 64, D % 8 == 0, D <= 2048.  rowsq_next (optional, with xs_next): f32 [rows], the sum of squares of every
 * new bf16 row — the statistic the first layer's folded adaRMS projection consumes (kai0_skinny_desc.rowsq_in, one partial). */
int kai0_denoise_glue(const void* xs, const float* mod, int64_t mod_ld, int rows_per_batch, float eps, const float* w_out,
                      const float* b_out, float* x_t, float dt, const float* w_in, const float* b_in, void* xs_next,
                      int64_t rows, int D, int A, float* rowsq_next, kai0_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer (train_pytorch.py:469-475,557-561; optimizer.py:15-85).
 * sumsq: out[0] += sum(g^2) over a bf16 or f32 buffer (caller zeroes out).  scratch: 4096 floats for the per-block partials,
 * which a second launch adds in a fixed order (reproducible gradient norm; no atomics). */
int kai0_sumsq(const void* g, int g_f32, int64_t n, float* out, float* scratch, kai0_stream_t stream);
/* dst[i] = sum_j src[j * chunk_stride + i] (j = 0 .. chunks-1 in that order, f32 accumulation, one rounding to the buffer's
 * dtype: bf16 or f32).  The reduction half of the all-pairs gradient reduce-scatter (kai0_amd.sharded, rs_algo "alltoall"): the
 * all-to-all delivers every peer's copy of this rank's slice over that peer's own xGMI link, this kernel adds them — what the
 * reduction inside DistributedDataParallel's all-reduce does (train_pytorch.py:440-447), without a rounding per ring hop.
 * n and chunk_stride multiples of 8 (bf16) / 4 (f32) elements, buffers 16-byte aligned. */
int kai0_sum_chunks(const void* src, int is_f32, int chunks, int64_t chunk_stride, int64_t n, void* dst, kai0_stream_t stream);
/* Fused AdamW on a flat shard: master/m/v f32, grad bf16 or f32, writes the bf16 (or f32) model copy.
 * clip_coef is read from device memory (coef[0]) so the step stays graph/stream ordered:
 *   g = grad * coef[0]; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p = p - lr * (m/(1-b1^t) / (sqrt(v/(1-b2^t)) + eps) + wd * p)    (torch.optim.AdamW semantics) */
int kai0_adamw(float* master, float* m, float* v, const void* grad, int grad_f32, void* model_param,
               int param_f32, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
               float bias_c1, float bias_c2, const float* clip_coef, kai0_stream_t stream);
/* kai0_adamw for a [n_rows][row_len] parameter with mostly-zero gradient rows (the embedding table, modeling_gemma.py embed_tokens: a
 * step touches <= B x 200 of 257152 rows).  A row with zero moments and a zero gradient is a fixed point of the update when
 * 1 - lr*wd == 1.0f (required), so it is skipped after its gradient has been read; row_active (uint8 [n_rows], persistent: set when
 * a row first sees a nonzero gradient, i.e. its moments may be nonzero; all 1 is always valid) tells which rows must be updated
 * regardless.  Bit-identical to kai0_adamw on the same buffers. */
int kai0_adamw_rows(float* master, float* m, float* v, const void* grad, int grad_f32, void* model_param, int param_f32,
                    int64_t n_rows, int row_len, unsigned char* row_active, float lr, float beta1, float beta2, float eps, float wd,
                    float bias_c1, float bias_c2, const float* clip_coef, kai0_stream_t stream);
/* coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) ; norm_out[0] = sqrt(sumsq[0])
 * (torch.nn.utils.clip_grad_norm_) */
int kai0_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, kai0_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KAI0HIP_H */
