/*
 * dualhyp_hip.h — C ABI of libdualhyp_hip.so: the MI355X (gfx950) implementation of DualHyp's
 * LLM hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 * parameter name starts with `h_`.  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Every function returns 0 on success; on failure it returns non-zero and
 * dh_last_error() describes it (thread-local).  Nothing here allocates unless it says so.
 *
 * The reference has no FFI: its boundary for this path is the Python class API
 * (ger/lora.py GPT.forward, generate/base.py generate).  Each entry point below replaces the
 * tensor program one reference function expands to on CPU/CUDA; the reference file:line is
 * given per function.  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add.
 *
 * dtype: bf16 = uint16_t bit pattern (storage dtype of the reference's bf16-true mode); all
 * accumulation is fp32; results are rounded to bf16 at exactly the points where the
 * reference's eager bf16 tensor program rounds.
 */
#ifndef DUALHYP_HIP_H
#define DUALHYP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t dh_bf16;

#define DH_ABI_VERSION 6

int dh_abi_version(void);
/* Kernel-variant selector for A/B measurements inside one process (bench.py --tune k=v); never needed in production,
 * keys 4, 10 and 16 select a different fp32 summation order (low bits change), no other key changes a result bit.  Keys: 0 decode partial-sum GEMM (0 = K split over the waves of a block,
 * 1 = row-parallel with x staged in LDS); 1 prefill GEMM (0 = always 128x128 tiles, 1..3 = 256x256 loop variants on eight waves, 4 = those with persistent blocks,
 * 5 = the default: 256x256 on four waves with full-line 64-deep stages, round 3);
 * 2 SwiGLU streaming variant; 3 gemm_mid on/off; 4 phase pin of dh_linear_bf16 (0 by shape, 1 tiled, 2 decode);
 * 5 band height of the 256-tile walk; 6 / 7 first row count of the tiled decode kernels (fused epilogues / chain sums);
 * 8 gemm_dt stages; 9 128-tile stages; 10 decode steps from this many rows on the prefill kernels (0 = never);
 * 11 column tiles per wave of the K-sliced kernel; 12 rope + cache append fused into the QKV GEMM (1) or two launches;
 * 13 W through the per-wave LDS ring in gemm_mid (1) or straight to VGPRs; 14 row groups per block of the K-sliced
 * kernel above 128 rows (0 = by grid size, 8, 10); 15 8-wave SwiGLU tile in gemm_dt (0 = from 256 rows, 1 always,
 * -1 never); 16 k-steps per K-slice for K <= 4096 (8 or 16); 17 / 18 / 21 pair-sum decode GEMM (tile width, first row count, sc1 stores);
 * 19 / 20 fp8 tile edge and band height; 22 persistent blocks of the 4-wave prefill GEMM (0 never, 1 where the epilogue loads nothing: default,
 * 2 always); 23 a forward asked for the last position's logits only runs the last block's attention output, projection and MLP on each
 * sequence's last row (1, default) or on every row (0) — same bits either way (dh_engine_forward). */
int dh_set_tuning(int key, int value);
const char* dh_last_error(void);
/* Name of the first visible device's gcnArch ("gfx950") into buf; fails when no GPU. */
int dh_device_info(char* h_buf, int h_buf_len, int* h_num_cu, int64_t* h_hbm_bytes);

/* ------------------------------------------------------------------ elementwise / layout */

/* out[t,:] = wte[ids[t],:]            — nn.Embedding, ger/lora.py:537 */
int dh_embed_bf16(const int64_t* ids, const dh_bf16* wte, dh_bf16* out, int n_tok, int d,
                  int vocab, void* stream);

/* RMSNorm in the storage dtype — ger/rmsnorm.py:17-21:
 *   ms = bf16(mean_fp32(bf16(x*x))) ; r = bf16(rsqrt(bf16(ms+eps))) ; out = bf16(w*bf16(x*r))
 * If `resid` is non-NULL the input is x := bf16(x + resid) and `sum_out` (if non-NULL)
 * receives that sum (the residual stream update of ger/model.py:185-186).
 * row_tail (nullable, uint8 per row): rows flagged non-zero use r = bf16(1 / bf16(sqrt(t)))
 * instead of bf16(1/sqrt(t)).  That is what the reference's CPU path computes for the rows
 * torch's bf16 rsqrt handles in its scalar tail loop (the last n %% 32 rows of a call on AVX-512
 * hosts, hence every single-token decode call) — SURVEY.md quirk list, DESIGN.md Q11. */
int dh_rmsnorm_bf16(const dh_bf16* x, const dh_bf16* resid, const dh_bf16* w, dh_bf16* out,
                    dh_bf16* sum_out, int rows, int d, float eps, const uint8_t* row_tail,
                    void* stream);

/* Split the fused QKV projection, rotate q and k, append k/v to the KV cache —
 * ger/model.py:216-259.  qkv: [n_tok, n_groups*(q_per_kv+2)*hs] in the group-interleaved layout
 * [q0..q(q_per_kv-1) k v] per group (scripts/convert_hf_checkpoint.py:187-202).
 *   q_out  [n_tok, n_head, hs]            rotated queries
 *   k_cache [n_slots, n_groups, s_max, hs]   (compact GQA cache; the reference's 8x expansion
 *   vT_cache[n_slots, n_groups, hs, s_max]    ger/model.py:225-227 is not materialised)
 * tok_slot[t], tok_pos[t]: cache slot (sequence) and position of token t.
 * k_out / v_out (nullable): plain [n_tok, n_groups, hs] copies of the rotated k and of v, kept by
 * the training forward for the attention backward. */
int dh_qkv_rope_cache_bf16(const dh_bf16* qkv, const dh_bf16* cos, const dh_bf16* sin,
                           const int32_t* tok_slot, const int32_t* tok_pos, dh_bf16* q_out,
                           dh_bf16* k_cache, dh_bf16* vT_cache, dh_bf16* k_out, dh_bf16* v_out,
                           int n_tok, int n_head, int n_groups, int hs, int s_max, void* stream);

/* ------------------------------------------------------------------ GEMMs (MFMA) */

/* Epilogue selector for dh_linear_bf16 */
#define DH_EPI_PLAIN   0   /* y = bf16(acc)                                 F.linear            */
#define DH_EPI_LORA    1   /* y = bf16(bf16(acc) + bf16(bf16(lora_acc)*s))  ger/lora.py:159-166,388-402 */
#define DH_EPI_SWIGLU  2   /* y = bf16(bf16(silu(bf16(acc1))) * bf16(acc2)) ger/model.py:313-315 */
#define DH_EPI_ADAPTER 3   /* y = bf16(scale * bf16(bf16(acc) + bias))      ger/lora.py:70-71   */

/* y[M,N] = epilogue(x[M,K] · W[N,K]^T).  K % 64 == 0, N % 8 == 0.
 *  LORA   : xa [M, xa_ld] = bf16(x·A^T) (from dh_linear_bf16 PLAIN with W=A), lora_b [N,16]
 *           (rank zero-padded to 16).  Output column n uses xa columns
 *           [16*seg, 16*seg+16) with seg = (n >= split0) + (n >= split1): the contiguous
 *           [Q|K|V] placement of ger/lora.py:226-234,343-347 (quirk Q2).  split0/1 % 32 == 0;
 *           pass N,N for a single segment.
 *  SWIGLU : w2 [N,K] second weight (fc_2); W is fc_1.
 *  ADAPTER: vec_a = adapter_scale[N], vec_b = adapter_bias[N].
 *  resid  : if non-NULL, y = bf16(resid + y_epilogue)  (residual add of ger/model.py:185-186)
 *  M may be any value >= 1; for M <= 32 a weight-streaming kernel is used, otherwise the tiled MFMA kernels
 *  (the engine additionally pins the kernel family per phase: see DESIGN.md, batch invariance). */
int dh_linear_bf16(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K,
                   int epilogue, const dh_bf16* w2, const dh_bf16* xa, int xa_ld,
                   const dh_bf16* lora_b, float lora_scale, int split0, int split1,
                   const dh_bf16* vec_a, const dh_bf16* vec_b, const dh_bf16* resid,
                   void* stream);

/* The fused-QKV projection of a packed prefill with the work of dh_qkv_rope_cache_bf16 in its epilogue
 * (ger/model.py:216-259 after ger/lora.py:367-402): q/k rotated, q -> q_out [M, n_head, hs], k / v appended to the
 * caches; the [M, (n_head+2g)*hs] qkv tensor is never written.  Same arithmetic and rounding points as
 * dh_linear_bf16(EPI_LORA, splits (d, d+g*hs)) followed by dh_qkv_rope_cache_bf16 — bit-identical results.
 * lora_b NULL = no LoRA.  Only for shapes the 256-tile kernel takes (M >= 256 and >= 128 tiles); smaller calls use the
 * two-step form. */
int dh_linear_qkv_rope_cache_bf16(const dh_bf16* x, const dh_bf16* w, int M, int K, const dh_bf16* xa, int xa_ld,
                                  const dh_bf16* lora_b, float lora_scale, const dh_bf16* cos, const dh_bf16* sin,
                                  const int32_t* tok_slot, const int32_t* tok_pos, dh_bf16* q_out, dh_bf16* k_cache,
                                  dh_bf16* vT_cache, int n_head, int n_groups, int hs, int s_max, void* stream);

/* Round 3 (ABI 4): the same two entry points with the LoRA DOWN-projection computed by the library — ger/lora.py:159-166, 388-402:
 *   after = W x ; after_A = lora_A(x) (bf16) ; after_B = lora_B(after_A) ; result = after + scaling * after_B
 * lora_a: [16 * nseg, K] bf16, segment s in rows 16 s .. 16 s + 15 (rank zero-padded to 16; nseg = 1 + (split0 < N) + (split1 < N),
 * 3 for the fused QKV projection).  Where the launch runs on the 4-wave 256-tile kernel and every segment boundary is a multiple of
 * 256 columns, bf16(x . A^T) rides in the GEMM's own K loop (16 more rows per stage, 4 more MFMAs per wave and k-step on the x
 * fragments already in registers, the result handed to the epilogue through LDS): no separate launch, no [M, 16 nseg] tensor, same
 * bits.  Otherwise the library runs dh_linear_bf16(x, lora_a) into xa_work and then the LORA epilogue.  xa_work: [M, 16 nseg] bf16,
 * always required (which path runs is the library's choice). */
int dh_linear_lora_bf16(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, const dh_bf16* lora_a,
                        const dh_bf16* lora_b, float lora_scale, int split0, int split1, const dh_bf16* resid,
                        dh_bf16* xa_work, void* stream);
int dh_linear_qkv_lora_rope_cache_bf16(const dh_bf16* x, const dh_bf16* w, int M, int K, const dh_bf16* lora_a,
                                       const dh_bf16* lora_b, float lora_scale, const dh_bf16* cos, const dh_bf16* sin,
                                       const int32_t* tok_slot, const int32_t* tok_pos, dh_bf16* q_out, dh_bf16* k_cache,
                                       dh_bf16* vT_cache, int n_head, int n_groups, int hs, int s_max, dh_bf16* xa_work,
                                       void* stream);

/* fp32 partial sums for the fused decode consumers below (M <= 4096 rows, weight streaming):
 *   y32[p][m][n] = sum over K-slice p of x[m,:] . W'[n,:],  W' = [w (n_main rows) ; w_ext (n_ext rows)]
 * y32: [ksplit][M][n_main+n_ext] fp32.  w_ext is the rank-padded LoRA A (so x·A^T comes out of the
 * same pass over x as x·W^T and never needs its own launch).  K %% 32 == 0, rows %% 16 == 0.
 *
 * K-SLICE COMBINE ORDER of the decode family (round 3; one order for every row count, so a row's bits do not depend on
 * how many rows are decoded with it): a slice is one fp32 chain of v_mfma_f32_16x16x32_bf16 from zero, k ascending;
 * ADJACENT slices are added in pairs, (s0 + s1), (s2 + s3), ..., an unpaired last slice stands alone; the pair sums are
 * added in index order.  Producers emit either the slices (dh_linear_partial_bf16: consumers take them with pairs = 1)
 * or the pair sums (dh_linear_partial_pairs_bf16: pairs = 0, half the bytes) or the total (dh_linear_chain_bf16:
 * n_part = 1). */
int dh_linear_partial_bf16(const dh_bf16* x, const dh_bf16* w, const dh_bf16* w_ext, float* y32,
                           int M, int n_main, int n_ext, int K, int ksplit, void* stream);
/* The PAIR SUMS of those ksplit slices, y32: [(ksplit+1)/2][M][n_main+n_ext] fp32, from a tiled split-K kernel (128-row
 * tiles, both operands through LDS, one block per tile and slice pair): y32[j] = slice 2j + slice 2j+1, bit-identical
 * to adding the two partials of dh_linear_partial_bf16.  The decode GEMMs of more than 128 rows (several batches decoded
 * jointly).  Needs K %% 64 == 0, K-slices of 8 or 16 k-steps (ceil(K/32/ksplit)), whole slices. */
int dh_linear_partial_pairs_bf16(const dh_bf16* x, const dh_bf16* w, const dh_bf16* w_ext, float* y32, int M,
                                 int n_main, int n_ext, int K, int ksplit, void* stream);
/* The ksplit slices combined in the family's order (pairs, then pair sums in index order) in one launch (tiled kernel,
 * full K per block): y32 [M][n_main+n_ext] fp32.  Consumers take it with n_part = 1.  Needs K %% 64 == 0 and
 * K-slices of 8 or 16 k-steps (ceil(K/32/ksplit)); other shapes return an error. */
int dh_linear_chain_bf16(const dh_bf16* x, const dh_bf16* w, const dh_bf16* w_ext, float* y32, int M,
                         int n_main, int n_ext, int K, int ksplit, void* stream);

/* LoRA finish + residual add + RMSNorm of one decode row block — ger/lora.py:159-166,
 * ger/model.py:185-186, ger/rmsnorm.py:17-21 in one pass:
 *   h = bf16(bf16(sum_p h32[p]) + bf16(bf16(xa . B^T) * s)) ; x_out = bf16(x_resid + h) ;
 *   xn_out = RMSNorm(x_out; w_norm, eps)      (row_tail: see dh_rmsnorm_bf16)
 * h32: [n_part][rows][d+n_ext] fp32 partials, xa = bf16 of columns [d, d+16); pairs != 0: the partials are SLICES
 * (added in adjacent pairs first), pairs == 0: they are pair sums or the total (added in index order).
 * lora_b == NULL: no LoRA (n_ext = 0).  x_out may alias x_resid. */
int dh_finish_norm_bf16(const float* h32, int n_part, int pairs, int rows, int d, int n_ext,
                        const dh_bf16* lora_b, float lora_scale, const dh_bf16* x_resid,
                        const dh_bf16* w_norm, dh_bf16* x_out, dh_bf16* xn_out, float eps,
                        const uint8_t* row_tail, void* stream);

/* ------------------------------------------------------------------ attention */

/* Causal attention of a packed batch against the KV cache — ger/model.py:261,270-290 with the
 * boolean mask of ger/lora.py:530-531.  Sequence i owns q rows [q_start[i], q_start[i]+q_len[i])
 * of q [n_tok, n_head, hs]; row j of it sits at position kv_pos0[i]+j and attends cache
 * positions 0..kv_pos0[i]+j of slot seq_slot[i].  y: [n_tok, n_head*hs].  scale = 1/sqrt(hs).
 * lse (nullable): [n_tok, n_head] fp32 log-sum-exp of the scaled scores, kept for the backward. */
int dh_attn_prefill_bf16(const dh_bf16* q, const dh_bf16* k_cache, const dh_bf16* vT_cache,
                         const int32_t* seq_slot, const int32_t* q_start, const int32_t* q_len,
                         const int32_t* kv_pos0, dh_bf16* y, float* lse, int n_seq, int max_q_len,
                         int n_head, int n_groups, int hs, int s_max, void* stream);

/* One query token per sequence (decode step): q [n_seq, n_head, hs]; sequence i attends cache
 * positions 0..kv_len[i]-1 of slot seq_slot[i].  work: >= dh_attn_decode_work_bytes(). */
int64_t dh_attn_decode_work_bytes(int n_seq, int n_head, int hs, int s_max);
int dh_attn_decode_bf16(const dh_bf16* q, const dh_bf16* k_cache, const dh_bf16* vT_cache,
                        const int32_t* seq_slot, const int32_t* kv_len, dh_bf16* y, void* work,
                        int n_seq, int n_head, int n_groups, int hs, int s_max, void* stream);

/* The whole attention sub-layer of a decode step in one launch — ger/model.py:216-261 for T = 1:
 * finish the q/k/v LoRA from the fp32 partials of dh_linear_partial_bf16 (columns
 * [qkv_dim, qkv_dim+48) = x.A^T; contiguous [Q|K|V] delta, quirk Q2), rotate q and k at position
 * kv_len-1, append k / v to the caches, attend keys 0..kv_len-1 (split over 8 waves, combined in
 * LDS; the new key is merged from registers), write y [n_seq, n_head*hs].  n_part / pairs: as dh_finish_norm_bf16. */
int dh_attn_decode_fused_bf16(const float* qkv32, int n_part, int pairs, int n_seq, int qkv_dim, int n_ext,
                              const dh_bf16* lora_b, float lora_scale, int split0, int split1,
                              const dh_bf16* cos, const dh_bf16* sin, const int32_t* seq_slot,
                              const int32_t* kv_len, dh_bf16* k_cache, dh_bf16* vT_cache,
                              dh_bf16* y, int n_head, int n_groups, int hs, int s_max,
                              void* stream);

/* ------------------------------------------------------------------ LoRA fine-tune backward
 * finetune/ger.py:278-285 `fabric.backward(loss / accum)` for the frozen-base / LoRA-only case: the dX
 * GEMMs reuse dh_linear_bf16 on transposed copies of the frozen weights; these are the rest. */

/* LoRA-branch dropout, ger/lora.py:96,165,391 (`nn.Dropout(p=lora_dropout)` in front of lora_A; finetune/ger.py:401 trains at
 * 0.05): y = bf16(x * m), mask = m = keep ? bf16(1 / (1 - p)) : 0, keep drawn per element from Philox4x32-10 keyed by
 * (seed, call_id) and counted by (*step_dev, element): `step_dev` is a DEVICE counter the caller bumps once per micro-step, so
 * a captured hipGraph draws new masks on every replay (null = step 0).  n % 8 == 0.  ABI 5. */
int dh_dropout_bf16(const dh_bf16* x, dh_bf16* y, dh_bf16* mask, int64_t n, float p, uint64_t seed, uint32_t call_id,
                    const uint64_t* step_dev, void* stream);
/* act = bf16(bf16(silu(g)) * u) from stored g, u (training forward keeps both; ger/model.py:315) */
int dh_swiglu_fwd_bf16(const dh_bf16* g, const dh_bf16* u, dh_bf16* act, int64_t n, void* stream);
/* y = bf16(bf16(x . W^T) * mul), mul [M, N] bf16 (round 4, ABI 6): a plain GEMM that applies an elementwise multiplier where it
 * rounds — the bits of dh_linear_bf16 followed by a bf16 multiply (the LoRA-branch dropout mask in the fine-tune's backward).
 * Large shapes only: M >= 256, N >= 256, ceil(M/256) * ceil(N/256) >= 128. */
int dh_linear_mul_bf16(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K,
                       const dh_bf16* mul, void* stream);
/* Training forward of fc_1 / fc_2 in one launch (round 4, ABI 6): act = bf16(bf16(silu(g)) * u) with g = bf16(x.W1^T), u = bf16(x.W2^T)
 * also stored ([M, I] each) for the backward — the bits of two dh_linear_bf16 launches + dh_swiglu_fwd_bf16 (ger/model.py:313-315). */
int dh_linear_swiglu_train_bf16(const dh_bf16* x, const dh_bf16* w1, const dh_bf16* w2, dh_bf16* act,
                                dh_bf16* g, dh_bf16* u, int M, int I, int K, void* stream);
/* dgu[rows, 2I] = [dact*u*silu'(g) | dact*silu(g)]   (backward of ger/model.py:315) */
int dh_swiglu_bwd_bf16(const dh_bf16* dact, const dh_bf16* g, const dh_bf16* u, dh_bf16* dgu, int rows,
                       int I, void* stream);
/* dx = d(RMSNorm)/dx . dy (+ dres)   (backward of ger/rmsnorm.py:17-21; fp32 internally) */
int dh_rmsnorm_bwd_bf16(const dh_bf16* dy, const dh_bf16* x, const dh_bf16* w, const dh_bf16* dres,
                        dh_bf16* dx, int rows, int d, float eps, void* stream);
/* conjugate rotation of dq / dk, pass-through of dv, scattered back into the fused-qkv layout
 * (backward of ger/model.py:216-246; cf. ger/fused_rotary_embedding.py:49-90) */
int dh_qkv_rope_bwd_bf16(const dh_bf16* dq, const dh_bf16* dk, const dh_bf16* dv, const dh_bf16* cos,
                         const dh_bf16* sin, const int32_t* tok_pos, dh_bf16* dqkv, int n_tok,
                         int n_head, int n_groups, int hs, void* stream);
/* out[M,N] (+)= scale * a[T,M]^T . b[T,N]   (LoRA dA / dB: contraction over tokens), fp32 out.  work (nullable):
 * >= dh_tn_accum_work_bytes(T, M, N) bytes of scratch; with it a long token loop (packed micro-batches: T > 1024) is split
 * into 512-token chunks over the grid and the chunk sums are added in index order by a second launch (deterministic). */
int64_t dh_tn_accum_work_bytes(int T, int M, int N);
int dh_tn_accum_f32(const dh_bf16* a, int lda, const dh_bf16* b, int ldb, float* out, int ldo, int T,
                    int M, int N, float scale, int accumulate, void* work, void* stream);
/* The same contraction for the three LoRA-B gradients of a fused QKV projection in one launch (round 4, ABI 6):
 * out[m][n] (+)= scale * sum_t a[t][m] * b[t][16 seg(m) + n], n < 16, seg(m) = (m >= seg0) + (m >= seg1); seg0 <= seg1 multiples of 128,
 * b [T, >= 48]; work as dh_tn_accum_work_bytes(T, M, 16). */
int dh_tn_accum_seg_f32(const dh_bf16* a, int lda, const dh_bf16* b, int ldb, float* out, int ldo, int T,
                        int M, int seg0, int seg1, float scale, int accumulate, void* work, void* stream);
/* out[row] = sum_d a[row,d]*b[row,d]   (softmax-backward row term D = rowsum(dO*O)) */
int dh_rowdot_f32(const dh_bf16* a, const dh_bf16* b, float* out, int64_t rows, int hs, void* stream);
/* src [n_tok, heads, hs] -> dst (heads * hs * n_pad elements, zero-initialised by the caller): the token-contiguous copy the
 * backward kernels take as a transposed MFMA operand, in FRAGMENT ORDER (csrc/attention_bwd.hip: tfrag_off — per head and 32-token
 * tile the 2 x hs/32 fragments of v_mfma_f32_32x32x16_bf16, 64 lanes x 8 values each, so a wave's fragment is one contiguous 1-KiB
 * load); sequence i starts at padded token pad_start[i] (multiple of 32). */
int dh_transpose_pad_bf16(const dh_bf16* src, dh_bf16* dst, const int32_t* tok_seq,
                          const int32_t* q_start, const int32_t* pad_start, int n_tok, int heads,
                          int hs, int n_pad, void* stream);
/* The same copy from the INVERSE map (round 4, ABI 6): pad_tok[p] = token of padded position p, or -1 for padding (written as
 * zeros: dst needs no memset).  16-byte accesses both ways; one block per padded 32-token tile and head.  src, dst 16-byte aligned. */
int dh_transpose_frag_bf16(const dh_bf16* src, dh_bf16* dst, const int32_t* pad_tok, int heads, int hs,
                           int n_pad, void* stream);
/* Causal GQA attention backward (no KV cache; ger/model.py:287-289 under autograd): dq [n_tok,H,hs],
 * dk / dv [n_tok,G,hs] from q, k, v (rotated, plain), dout, lse (dh_attn_prefill_bf16) and
 * dsum = rowsum(dout*out); qT / doT / kT from dh_transpose_pad_bf16 / dh_transpose_frag_bf16 (those dh_attn_bwd_transposes names). */
/* Which transposed copies dh_attn_bwd_bf16 reads for a shape (round 4, ABI 6): bit 0 = qT and doT, bit 1 = kT; the others may be null.
 * (hs 64: the dk/dv kernel stages q / dO tiles in LDS and reads them transposed there — ds_read_b64_tr_b16 — so only kT is needed.) */
int dh_attn_bwd_transposes(int n_head, int n_groups, int hs, int n_pad);
int dh_attn_bwd_bf16(const dh_bf16* q, const dh_bf16* k, const dh_bf16* v, const dh_bf16* dout,
                     const dh_bf16* qT, const dh_bf16* doT, const dh_bf16* kT, const float* lse,
                     const float* dsum, const int32_t* q_start, const int32_t* q_len,
                     const int32_t* pad_start, dh_bf16* dq, dh_bf16* dk, dh_bf16* dv, int n_seq,
                     int max_q_len, int n_head, int n_groups, int hs, int n_pad, void* stream);

/* ------------------------------------------------------------------ token sampling */

/* Token-level cross entropy, ignore_index = -1 — ger/utils.py:424-463 (F.cross_entropy on the
 * autocast-upcast logits): loss[r] = logsumexp(logits[r,:]) - logits[r,target[r]] in fp32, 0 for
 * ignored rows (target outside [0, vocab)); lse[r] is kept for the backward.  logits: [rows, vocab]
 * bf16 (is_f32 = 0) or fp32 (is_f32 = 1).  The caller picks the normalisation (quirk Q5: chunked
 * variants divide by ALL rows, chunk_size = 0 by the valid rows). */
int dh_cross_entropy_fwd(const void* logits, int is_f32, const int64_t* targets, float* loss,
                         float* lse, int rows, int vocab, void* stream);
/* dlogits[r,c] = grad_row[r] * (softmax(logits[r,:])[c] - [c == target[r]]), 0 for ignored rows;
 * same dtype as logits. */
int dh_cross_entropy_bwd(const void* logits, int is_f32, const int64_t* targets, const float* lse,
                         const float* grad_row, void* dlogits, int rows, int vocab, void* stream);

/* RelPrompt reliability predictor — ger/relprompt.py:126-147 (NoiseMaskClassifier).  The two k=3 convolutions
 * are dh_linear_bf16 over the matrix built here: out [B*T, ld] with out[b,t, dk*C + c] = x[b, t+dk-1, c]
 * (zero outside the sequence, through ReLU if relu != 0), a column of ones at 3C (the bias rides the fp32
 * accumulation as the weight's column 3C) and zero padding up to ld (multiple of 64 for the GEMM). */
int dh_im2col3_bf16(const dh_bf16* x, dh_bf16* out, int B, int T, int C, int ld, int relu, void* stream);
/* ReLU -> AvgPool1d(pool, stride pool, ceil_mode: a short last window averages its own elements) ->
 * Linear(H -> 3): h [B,T,H] pre-activation, w [3,H], bias [3], out [B, ceil(T/pool), 3]. */
int dh_pool_head_bf16(const dh_bf16* h, const dh_bf16* w, const dh_bf16* bias, dh_bf16* out, int B, int T,
                      int H, int pool, void* stream);
/* Training of the reliability predictors — finetune/relprompt.py:356-387 (the mask cross entropy back-propagates
 * into conv1 / conv2 / classifier; ger/relprompt.py:79-119 keeps them trainable).
 * dh_pool_head_bwd_bf16: backward of ReLU -> AvgPool1d(pool, ceil) -> Linear(H, 3): dlogits fp32 [B, P, 3] ->
 *   dh bf16 [B,T,H] (gradient of the PRE-activation h) and pooled bf16 [B*P, H] (the Linear's input, recomputed
 *   exactly as dh_pool_head_bf16 forms it; the caller contracts it with dlogits for the Linear's weight gradient).
 * dh_col2im3_bf16: backward of dh_im2col3_bf16 fused with what precedes it: dx[b,t,c] = [pre[b,t,c] > 0] *
 *   mask[b,t,c] * sum_dk dcol[b, t+1-dk, dk*C + c]; pre (ReLU input) and mask (scaled dropout mask) may be NULL. */
int dh_pool_head_bwd_bf16(const dh_bf16* h, const dh_bf16* w, const float* dlogits, dh_bf16* dh, dh_bf16* pooled,
                          int B, int T, int H, int pool, void* stream);
int dh_col2im3_bf16(const dh_bf16* dcol, const dh_bf16* pre, const dh_bf16* mask, dh_bf16* dx, int B, int T, int C,
                    int ld, void* stream);

/* One decode-loop tail per sequence — generate/base.py:62-80:
 *   l = logits/temperature (bf16) ; keep l >= k-th largest ; softmax ; multinomial.
 * top_k == 1 is resolved as arg-max with the LOWEST index among equal maxima (the reference
 * breaks bf16 ties with the torch RNG, quirk Q6).  top_k == 0 means no cropping.  For
 * top_k != 1 a counter hash of (seed, step, seq) drives the inverse-CDF draw.
 *   tokens [n_seq, tok_ld] int64 : token buffer per sequence (prompt + generated)
 *   length [n_seq] int32         : tokens currently valid; the new id goes to
 *                                  tokens[i, length[i]] and length[i] is incremented
 *   done   [n_seq] int32         : set to 1 when the new id == eos_id (eos_id < 0: never);
 *                                  finished sequences are left untouched.
 * No host synchronisation: the reference's per-token `if idx_next == eos_id`
 * (generate/base.py:79) becomes the device-side flag. */
int dh_sample_bf16(const dh_bf16* logits, int vocab, int64_t* tokens, int tok_ld, int32_t* length,
                   int32_t* done, int n_seq, float temperature, int top_k, int64_t eos_id,
                   uint64_t seed, int step, void* stream);

/* ------------------------------------------------------------------ fp8 serving path (csrc/fp8.hip)
 * W8A8 with OCP e4m3fn: q = fp8_rne(v * (448 / amax)), scale = amax / 448 per row (amax >= 1e-12, fp32 arithmetic);
 * weights are quantised per output channel ahead of time (dualhyp_amd.quant, after merge_lora_weights), activations
 * per token by the two kernels below.  The CPU restatement is oracle/ger_oracle.py: quantize_rows_fp8 / linear_fp8. */

/* q[r,:] e4m3 [rows, K], scale[r] fp32 from bf16 rows.  K %% 8 == 0. */
int dh_quant_rows_fp8(const dh_bf16* x, uint8_t* q, float* scale, int rows, int K, void* stream);
/* dh_rmsnorm_bf16 (no residual) whose bf16 output row is quantised in registers; xn_out (nullable) receives the bf16 row. */
int dh_rmsnorm_quant_fp8(const dh_bf16* x, const dh_bf16* w, dh_bf16* xn_out, uint8_t* q, float* scale, int rows, int d,
                         float eps, const uint8_t* row_tail, void* stream);
/* y[M,N] bf16 = epilogue( bf16( (xq . wq^T in fp32) * (x_scale[m] * w_scale[n]) ) ) on the block-scaled fp8 MFMA
 * (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales).  K %% 128 == 0, N %% 4 == 0.  Epilogues: DH_EPI_PLAIN
 * (+ resid), DH_EPI_SWIGLU (w2q / w2_scale = fc_2), DH_EPI_ADAPTER (vec_a scale, vec_b bias).  M <= 128 streams the
 * weights once (decode), larger M runs 128 x 128 x 128 tiles. */
int dh_linear_fp8(const uint8_t* xq, const float* x_scale, const uint8_t* wq, const float* w_scale, dh_bf16* y, int M, int N,
                  int K, int epilogue, const uint8_t* w2q, const float* w2_scale, const dh_bf16* vec_a, const dh_bf16* vec_b,
                  const dh_bf16* resid, void* stream);
/* dh_linear_fp8 with the kernel pinned: 0 = by row count (as above), 1 = tiled, 2 = streaming (M <= 128).  The two kernels
 * add the K products in different fp32 orders; the engine pins by PHASE (prefill tiled whatever the packing, single-token
 * steps streaming up to 128 rows) so that a sequence's bits do not depend on what is packed with it. */
int dh_linear_fp8_ex(const uint8_t* xq, const float* x_scale, const uint8_t* wq, const float* w_scale, dh_bf16* y, int M, int N,
                     int K, int epilogue, const uint8_t* w2q, const float* w2_scale, const dh_bf16* vec_a, const dh_bf16* vec_b,
                     const dh_bf16* resid, int kernel, void* stream);

/* dh_linear_fp8 PLAIN (streaming kernel) for 1 <= M <= 128 with the bf16-rounded result written as fp32 [M, N]: the single "partial" the
 * fused decode-attention kernel (dh_attn_decode_fused_bf16, n_part = 1, no LoRA) consumes. */
int dh_linear_fp8_f32(const uint8_t* xq, const float* x_scale, const uint8_t* wq, const float* w_scale, float* y32, int M,
                      int N, int K, void* stream);

/* ------------------------------------------------------------------ decoder engine
 * Native runtime that owns the kernel sequence of ger/lora.py:504-549 for a whole batch:
 * embed -> n_layer x [norm_1, qkv(+LoRA), rope+cache, attention, proj(+LoRA)+residual,
 * norm_2, fc_1/fc_2+SwiGLU, proj+residual] -> ln_f -> lm_head, with the decode step captured
 * in a hipGraph.  Weights are NOT copied: the table holds device pointers owned by the caller
 * (the torch parameters of dualhyp_amd.GPT). */

typedef struct dh_layer_weights {
    const dh_bf16* norm_1;      /* [d]                                         */
    const dh_bf16* norm_2;      /* [d]                                         */
    const dh_bf16* attn_w;      /* [(n_head+2g)*hs, d]  attn.attn.linear.weight */
    const dh_bf16* attn_lora_a; /* [48, d]  rank padded to 16 per q/k/v; NULL = no LoRA */
    const dh_bf16* attn_lora_b; /* [(n_head+2g)*hs, 16]                        */
    const dh_bf16* proj_w;      /* [d, d]               attn.proj.linear.weight */
    const dh_bf16* proj_lora_a; /* [16, d] or NULL                             */
    const dh_bf16* proj_lora_b; /* [d, 16]                                     */
    const dh_bf16* fc_1;        /* [I, d]                                      */
    const dh_bf16* fc_2;        /* [I, d]                                      */
    const dh_bf16* mlp_proj;    /* [d, I]                                      */
    /* fp8 serving (BASELINE config 5; merged-LoRA path of ger/lora.py:152-157,349-365,707-711): when attn_ws is
     * non-NULL every *_ws must be, the five weight pointers above then address OCP e4m3 bytes [N, K] produced by
     * dualhyp_amd.quant (LoRA already merged: attn_lora_* / proj_lora_* NULL) and these are the fp32 scales per
     * output channel, [N] each. */
    const float* attn_ws;
    const float* proj_ws;
    const float* fc_1_ws;
    const float* fc_2_ws;
    const float* mlp_proj_ws;
} dh_layer_weights;

typedef struct dh_model_desc {
    int32_t n_layer, n_head, n_groups, head_size, n_embd, intermediate, vocab, block_size;
    float norm_eps;
    float lora_scale;           /* alpha / r */
    const dh_bf16* wte;         /* [vocab_rows, d] (vocab_rows >= vocab: RelPrompt adds rows) */
    int32_t wte_rows;
    const dh_bf16* ln_f;        /* [d] */
    const dh_bf16* rope_cos;    /* [block_size, head_size] bf16 tables of ger/model.py:319-346 as built by */
    const dh_bf16* rope_sin;    /*   GPT.build_rope_cache (base 10000, quirk Q1); made once by the host     */
    const dh_bf16* lm_head;     /* [vocab, d] */
    const dh_bf16* adapter_scale; /* [vocab] */
    const dh_bf16* adapter_bias;  /* [vocab] */
    const dh_layer_weights* h_layers; /* host array [n_layer] (copied) */
    const float* lm_head_ws;    /* fp8 serving: [vocab] channel scales, lm_head then addresses e4m3 bytes; else NULL */
} dh_model_desc;

typedef struct dh_engine dh_engine;

/* Allocates KV cache [n_layer][max_batch, g, s_max, hs] x2, activations for up to
 * max_tokens packed tokens and the decode-graph state. */
int dh_engine_create(const dh_model_desc* h_desc, int max_batch, int s_max, int max_tokens,
                     dh_engine** h_out);
void dh_engine_destroy(dh_engine* e);
int64_t dh_engine_device_bytes(const dh_engine* e);

/* Forward of a packed batch through the KV cache (ger/lora.py:504-549 with input_pos):
 *   h_seq_len[i] tokens for slot i (i < n_seq) starting at cache position h_pos0[i];
 *   ids: packed [sum(seq_len)] int64 on device.
 * logits_all != NULL : [n_tok, vocab] logits of every position (reference behaviour, Q9)
 * logits_last != NULL: [n_seq, vocab] logits of each sequence's last position only.  With logits_all == NULL (generate's
 *   prompt forward reads `logits[0, -1]`, generate/base.py:57-60) the last block's attention output, projection and MLP run on
 *   those n_seq rows alone — K and V of every token still go to the cache; the logits and the caches are bit-equal to the
 *   all-rows run (dh_set_tuning key 23; tests/test_hip_edges.py).  The hidden rows dh_engine_read(3) returns are then those
 *   of the last block's INPUT. */
int dh_engine_forward(dh_engine* e, const int64_t* ids, const int32_t* h_seq_len,
                      const int32_t* h_pos0, int n_seq, dh_bf16* logits_all,
                      dh_bf16* logits_last, void* stream);
/* Same, with sequence i of the call living in KV-cache slot slot_base + i: several batches are
 * prefilled one after the other into one engine and then decoded together (dh_engine_decode over
 * all occupied slots; rows of a larger decode call equal the same rows decoded alone). */
int dh_engine_forward_at(dh_engine* e, const int64_t* ids, const int32_t* h_seq_len,
                         const int32_t* h_pos0, int n_seq, int slot_base, dh_bf16* logits_all,
                         dh_bf16* logits_last, void* stream);

/* Reproduce the rsqrt rounding of the reference's CPU path (see dh_rmsnorm_bf16 row_tail):
 * vec_width = lanes of torch's bf16 vector loop on the reference host (32 on AVX-512, 16 on
 * AVX2), 0 = off (every row uses bf16(1/sqrt(t)), as a GPU run of the reference would).
 * whole_call = 1: the tail is the last n_tok %% vec_width rows of the packed call (what
 * GPT.forward(idx[B,T]) does); 0: per sequence (B independent batch-1 calls, generate()).
 * Decode steps are single-token calls, i.e. always tail rows.  Default: off. */
int dh_engine_set_cpu_rsqrt_emulation(dh_engine* e, int vec_width, int whole_call);

/* The decode loop of generate/base.py:57-80 for n_seq sequences at once, entirely on device:
 * tokens [n_seq, tok_ld] holds the prompts (length[i] valid ids each, already prefilled
 * through position length[i]-2; the last prompt token's logits are `logits_last` of the
 * prefill and must have been sampled already, i.e. length includes the first new token).
 * Runs `n_steps` x { forward(one token/seq at position length-1) ; sample }.  Sequences with
 * done[i] != 0 keep stepping harmlessly but their tokens/length are frozen. */
int dh_engine_decode(dh_engine* e, int64_t* tokens, int tok_ld, int32_t* length, int32_t* done,
                     int n_seq, int n_steps, float temperature, int top_k, int64_t eos_id,
                     uint64_t seed, int first_step, void* stream);

/* Test hook: copy engine state to `dst` (device memory, n_bytes) on `stream`.
 *   what 0: ln_f(x) of the last dh_engine_forward called with logits_all only, [n_tok, d]
 *   what 1: K   cache of `layer`  [max_batch, g, s_max, hs]
 *   what 2: V^T cache of `layer`  [max_batch, g, hs, s_max]
 *   what 3: residual stream x after the last layer of the last forward, [n_tok, d] */
int dh_engine_read(dh_engine* e, int what, int layer, void* dst, int64_t n_bytes, void* stream);
/* HIP-event timing of the dominant kernels inside the last forward/decode call (bench.py
 * roofline): returns accumulated milliseconds and launch count for kernel class `which`
 * (0 = prefill GEMM, 1 = decode weight-streaming GEMM, 2 = prefill attention,
 *  3 = decode attention); enable with dh_engine_set_timing(e, 1). */
int dh_engine_set_timing(dh_engine* e, int on);
int dh_engine_get_timing(dh_engine* e, int which, double* h_ms, int64_t* h_launches);

#ifdef __cplusplus
}
#endif
#endif /* DUALHYP_HIP_H */
