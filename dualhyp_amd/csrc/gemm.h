// Argument block shared by the tiled (gemm.hip) and weight-streaming (gemm_skinny.hip) kernels.
#pragma once
#include "common.h"

struct GemmArgs {
    const bf16_t* x;
    const bf16_t* w;
    const bf16_t* w2;
    bf16_t* y;
    const bf16_t* xa;
    const bf16_t* lora_a;   // non-null: x.A^T is computed inside the 4-wave 256-tile kernel's K loop ([16 per segment][K]); xa is then unused
    const bf16_t* lora_b;
    const bf16_t* vec_a;
    const bf16_t* vec_b;
    const bf16_t* resid;
    int M, N, K;
    int xa_ld, split0, split1;
    float lora_scale;
    int nb_n, nb_m;
    int gm;            // m-tiles per band of the 256-tile kernel's block order
    int fast_epi;      // 4-wave kernel: bit 0 the v_dot2_f32_bf16 fused-QKV epilogue, bit 1 the v_dot2 LoRA / residual epilogues (dh_set_tuning(24, bits))
    int resid_mul;     // 256-tile kernels, PLAIN + resid: the `resid` operand is an elementwise MULTIPLIER, y = bf16(bf16(acc) * resid) (dh_linear_mul_bf16)
    // DH_EPI_QKV (256-tile kernel only): the fused-QKV projection whose epilogue also rotates q / k, writes q and
    // appends k / v to the KV cache (what dh_qkv_rope_cache_bf16 does in a separate pass over the qkv tensor)
    const bf16_t* rope_cos;
    const bf16_t* rope_sin;
    const int32_t* tok_slot;
    const int32_t* tok_pos;
    bf16_t* q_out;
    bf16_t* k_cache;
    bf16_t* vT_cache;
    int n_head, n_groups, hs, s_max;
};

constexpr int DH_EPI_QKV = 4;   // internal: LoRA (optional) + rope + cache append, see dh_linear_qkv_rope_cache_bf16



// M <= 32: one pass over W straight from HBM to registers (gemm_skinny.hip)
int dh_linear_skinny(const GemmArgs& a, int epilogue, hipStream_t s);
// y[M, N] = bf16(x . W^T), N = 16 .. 64, thousands of rows (gemm_skinny.hip: gemm_skinny_n_kernel); the bits of the tiled kernels
bool dh_linear_skinny_n_ok(int M, int N, int K, const void* x, const void* w, const void* y);
int dh_linear_skinny_n(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, hipStream_t s);
// y[M, N] = bf16(x[M, 64] . W^T) [* mul] for a wide N (gemm_skinny.hip: gemm_k64_kernel); the bits of the tiled kernels
bool dh_linear_k64_ok(int M, int N, int K, const void* x, const void* w, const void* y, const void* mul);
int dh_linear_k64(const dh_bf16* x, const dh_bf16* w, const dh_bf16* mul, dh_bf16* y, int M, int N, hipStream_t s);
// the in-GEMM LoRA down-projection can run for these arguments (4-wave 256-tile kernel, tile-aligned segments): gemm256.hip
bool dh_linear_256_xa_ok(const GemmArgs& a, int epilogue);

// decode phase, M <= 256 rows from several batches in one launch (gemm_mid.hip)
bool dh_linear_mid_ok(const GemmArgs& a, int epilogue);
int dh_linear_mid(const GemmArgs& a, int epilogue, hipStream_t s);
extern int g_mid;            // 1: the decode phase uses the LDS-staged-x streaming kernel for PLAIN/SWIGLU/ADAPTER

// decode phase, > 64 rows: tiled kernel with the streaming kernels' summation order (gemm_dt.hip)
bool dh_linear_dt_ok(const GemmArgs& a, int epilogue, int seg);
int dh_linear_dt(const GemmArgs& a, int epilogue, int seg, hipStream_t s);
bool dh_chain_ok(int M, int n_main, int n_ext, int K, int ksplit);
// decode partial-sum GEMM of M rows: true = the tiled split-K kernel (pair sums, (ksplit+1)/2 partials), false = the
// K-sliced streaming kernel (ksplit slices); the consumers' `pairs` flag is the negation
bool dh_pairs_ok(int M, int n_main, int n_ext, int K, int ksplit);
int dh_pairs_tiled(const bf16_t* x, const bf16_t* w, const bf16_t* w_ext, float* y32, int M, int n_main, int n_ext, int K,
                   int kps, hipStream_t s);
int dh_chain_tiled(const bf16_t* x, const bf16_t* w, const bf16_t* w_ext, float* y32, int M, int n_main, int n_ext, int K,
                   int kps, hipStream_t s);
extern int g_dt_min_rows, g_chain_min_rows;

// true when dh_linear_impl would send this shape to the 256-tile kernel (gemm256.hip)
bool dh_linear_is_big(int M, int N, int epilogue);

// M >= 256: 256 x 256 x 64 tiles, one block per CU (gemm256.hip)
int dh_linear_256(GemmArgs a, int epilogue, hipStream_t s);
extern int g_linear_phase;
extern int g_gemm_gm;
extern int g_gemm_variant;   // 0: always the 128-tile kernel; 1 / 2 / 3: 256-tile kernel, loop variants of gemm256.hip; 4: 1 with persistent blocks for the PLAIN / SwiGLU epilogues; 5 (default): the 4-wave full-line kernel

// dh_linear_bf16 with an explicit kernel choice.  kernel: 0 = by shape (M <= 32 -> streaming),
// 1 = tiled MFMA kernel whatever M, 2 = decode phase (streaming kernels up to 256 rows).  The engine pins the choice per PHASE (prefill = tiled,
// single-token decode = skinny) so a sequence's result never depends on how many other
// sequences were packed into the same call (fp32 summation order differs between the two).
// the LORA epilogue with the down-projection x.A^T computed by the library: inside the 4-wave 256-tile kernel's K loop where possible,
// else one more dh_linear_impl launch into xa_work (gemm.hip)
int dh_linear_lora_impl(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, const dh_bf16* lora_a, const dh_bf16* lora_b,
                        float lora_scale, int split0, int split1, const dh_bf16* resid, dh_bf16* xa_work, int kernel, hipStream_t s);
int dh_linear_impl(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, int epilogue,
                   const dh_bf16* w2, const dh_bf16* xa, int xa_ld, const dh_bf16* lora_b, float lora_scale,
                   int split0, int split1, const dh_bf16* vec_a, const dh_bf16* vec_b, const dh_bf16* resid,
                   int kernel, hipStream_t s);
