// Native decoder runtime: owns the KV cache, the activation workspace and the kernel sequence of
// ger/lora.py:504-549 (GPT.forward) + generate/base.py:57-80 (the decode loop) for a packed,
// ragged batch.  Host side is plain C++; the decode step is captured once into a hipGraph and
// replayed, so a generated token costs one graph launch and no host synchronisation.
#include <vector>

#include "common.h"
#include "gemm.h"

int dh_sample_impl(const dh_bf16* logits, int vocab, int64_t* tokens, int tok_ld, int32_t* length, int32_t* done,
                   int n_seq, float temperature, int top_k, int64_t eos_id, uint64_t seed, int step,
                   const int32_t* step_dev, void* stream);

namespace {

struct Timing {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[4];
    std::vector<hipEvent_t> pool;
};

}  // namespace

constexpr int MAX_DECODE_ROWS = 2048;  // single-token calls up to here take the 7-launch streaming path
// Single-token steps over at least this many rows run the layer on the TILED MFMA GEMMs of the prefill (natural
// k order) instead of the K-sliced streaming kernels: from a few hundred rows on the step is no longer weight-
// bandwidth-bound and the partial-sum traffic of the K slices dominates (dh_set_tuning key 10; 0 = never).
int g_decode_tiled_rows = 0;
int g_short_kps = 8;       // k-steps per K-slice of the partial-sum GEMMs when K <= 4096 (dh_set_tuning key 16: 8 or 16)
int g_fuse_qkv_rope = 1;   // dh_set_tuning key 12: 0 = QKV GEMM, then dh_qkv_rope_cache_bf16 (the two-step form)
// dh_set_tuning key 23.  A prefill that is asked for the LAST position's logits only (generate's prompt forward,
// generate/base.py:57-60: `logits[0, -1]`) needs, of the last layer, the K / V rows of every token (they go to the cache) but the
// attention output, projection and MLP of the last token of each sequence alone: the other rows of the last block feed nothing
// (the reference computes and drops them, as it does the lm_head rows — SURVEY Q9).  1 = run those four products on the
// n_seq last rows (same kernels, same chains: the rows' bits do not change); 0 = on every row.
int g_prune_last_layer = 1;

struct dh_engine {
    dh_model_desc d;
    std::vector<dh_layer_weights> layers;
    int max_batch = 0, s_max = 0, max_tokens = 0;
    int qkv_dim = 0, kv_dim = 0;
    // device memory
    bf16_t *kc = nullptr, *vtc = nullptr;             // [L][B][G][S][HS], [L][B][G][HS][S]
    bf16_t *x = nullptr, *xn = nullptr, *qkv = nullptr, *qrot = nullptr, *att = nullptr, *xa = nullptr,
           *act = nullptr, *xlast = nullptr, *logits = nullptr;
    int32_t *tok_slot = nullptr, *tok_pos = nullptr, *seq_meta = nullptr;   // seq_meta: 4 x [B]
    int32_t *last_row = nullptr, *step_dev = nullptr;
    int32_t* last_meta = nullptr;                       // [ones | position of the last token] x [B]: the pruned last layer's attention call
    bf16_t *att_last = nullptr, *xn_last = nullptr, *act_last = nullptr;   // its n_seq-row operands
    uint8_t *row_tail = nullptr, *last_tail = nullptr, *ones = nullptr;   // Q11 rsqrt emulation flags
    int rsqrt_vec = 0, rsqrt_whole = 0;
    int64_t* dec_ids = nullptr;
    void* dec_work = nullptr;
    float* part32 = nullptr;                            // fp32 partial sums of the decode GEMMs
    bool fp8 = false;                                   // e4m3 weights + channel scales (csrc/fp8.hip)
    uint8_t* xq = nullptr;                              // fp8 mode: quantised activations [max_tokens, max(d, I)]
    float* xscale = nullptr;                            // fp8 mode: their per-token scales [max_tokens]
    int32_t* h_stage = nullptr;                         // pinned staging for the metadata
    size_t cache_layer_elems = 0;
    int64_t dev_bytes = 0;
    // decode graph
    hipStream_t gstream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_stage = nullptr;
    // captured decode steps, keyed by everything baked into the graph (a few batch sizes alternate in practice)
    struct GKey { int64_t* tokens; int tok_ld; int32_t *length, *done; int n_seq, top_k; float temp; int64_t eos; uint64_t seed; int rsqrt_vec, tiled_rows; };
    struct GEntry { GKey key; hipGraphExec_t exec; uint64_t used; };
    std::vector<GEntry> graphs;
    uint64_t graph_clock = 0;
    int last_ntok = 0;
    int slot_base = 0;         // first KV-cache slot of the sequences of the current forward call
    bool capturing = false;   // no event records inside a stream capture
    bool phase_decode = false; // single-token-per-sequence call: weight-streaming GEMMs + split-KV attention
    bool decode_tiled = false; // ... except that this step's row count put it in the tiled class (g_decode_tiled_rows)
    Timing tm;
};

namespace {

template <typename T>
int dmalloc(dh_engine* e, T** p, size_t n) {
    DH_HIP(hipMalloc((void**)p, n * sizeof(T)));
    e->dev_bytes += (int64_t)(n * sizeof(T));
    return 0;
}

// ids of the step, positions and lengths derived on device from (tokens, length)
__global__ void decode_prep_kernel(const int64_t* __restrict__ tokens, int tok_ld, const int32_t* __restrict__ length,
                                   int64_t* __restrict__ ids, int32_t* __restrict__ tok_slot,
                                   int32_t* __restrict__ tok_pos, int32_t* __restrict__ kv_len, int32_t* step_dev,
                                   int n_seq, int s_max) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_seq) {
        int n = length[i];
        n = n < 1 ? 1 : (n > s_max ? s_max : n);     // never index outside the cache
        ids[i] = tokens[(size_t)i * tok_ld + n - 1];
        tok_slot[i] = i;
        tok_pos[i] = n - 1;
        kv_len[i] = n;
    }
    if (i == 0) *step_dev += 1;
}

__global__ void set_i32_kernel(int32_t* p, int32_t v) { *p = v; }
__global__ void iota_i32_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ rows,
                                   bf16_t* __restrict__ dst, int n, int d) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)rows[wave] * d);
    uint4* t = reinterpret_cast<uint4*>(dst + (size_t)wave * d);
    for (int c = lane; c < d / 8; c += 64) t[c] = s[c];
}

struct TimeScope {
    dh_engine* e; int which; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    TimeScope(dh_engine* e_, int w, hipStream_t s_) : e(e_), which(w), s(s_) {
        if (!e->tm.on || e->capturing) return;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
    }
    ~TimeScope() {
        if (!a) return;
        hipEventRecord(b, s);
        e->tm.ev[which].push_back({a, b});
    }
};

int linear(dh_engine* e, const bf16_t* x, const bf16_t* w, bf16_t* y, int M, int N, int K, int epi,
           const bf16_t* w2, const bf16_t* xa, int xa_ld, const bf16_t* lb, int s0, int s1, const bf16_t* va,
           const bf16_t* vb, const bf16_t* resid, hipStream_t s, bool timed) {
    // kernel choice is a property of the phase, never of the packing (batch invariance)
    const int kernel = e->phase_decode && !e->decode_tiled ? 2 : 1;
    if (timed) {
        TimeScope t(e, e->phase_decode ? 1 : 0, s);
        return dh_linear_impl(x, w, y, M, N, K, epi, w2, xa, xa_ld, lb, e->d.lora_scale, s0, s1, va, vb, resid, kernel, s);
    }
    return dh_linear_impl(x, w, y, M, N, K, epi, w2, xa, xa_ld, lb, e->d.lora_scale, s0, s1, va, vb, resid, kernel, s);
}

// The layer stack on n_tok packed tokens whose metadata is already on the device.
// prefill: attention over (seq_slot, q_start, q_len, kv_pos0); decode: one token per sequence.
int run_layers(dh_engine* e, const int64_t* ids, int n_tok, int n_seq, int max_q_len, bool decode,
               const uint8_t* tail_flags, hipStream_t s, bool prune_last = false) {
    const uint8_t* rt = e->rsqrt_vec > 0 ? tail_flags : nullptr;
    e->phase_decode = decode;
    const dh_model_desc& D = e->d;
    const int d = D.n_embd, I = D.intermediate, hs = D.head_size, H = D.n_head, G = D.n_groups;
    int32_t* seq_slot = e->seq_meta + e->slot_base;      // identity: sequence i of the call lives in slot base+i
    int32_t* q_start = e->seq_meta + e->max_batch;
    int32_t* q_len = e->seq_meta + 2 * e->max_batch;
    int32_t* kv_pos0 = e->seq_meta + 3 * e->max_batch;   // decode: kv_len
    int rc;
    if ((rc = dh_embed_bf16(ids, D.wte, e->x, n_tok, d, D.wte_rows, s))) return rc;
    for (int l = 0; l < D.n_layer; ++l) {
        const dh_layer_weights& W = e->layers[l];
        bf16_t* kc = e->kc + (size_t)l * e->cache_layer_elems;
        bf16_t* vtc = e->vtc + (size_t)l * e->cache_layer_elems;
        if ((rc = dh_rmsnorm_bf16(e->x, nullptr, W.norm_1, e->xn, nullptr, n_tok, d, D.norm_eps, rt, s))) return rc;
        // large packed prefills: rope + KV append ride in the QKV GEMM's epilogue (same bits, no pass over the qkv tensor)
        const bool fuse_qkv = !decode && g_fuse_qkv_rope && dh_linear_is_big(n_tok, e->qkv_dim, DH_EPI_LORA);
        // x.A^T: inside the QKV GEMM's K loop on the fused large-prefill path (dh_linear_qkv_lora_rope_cache_bf16 decides), else a launch
        if (W.attn_lora_a && !fuse_qkv)
            if ((rc = linear(e, e->xn, W.attn_lora_a, e->xa, n_tok, 48, d, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 0, 0,
                             nullptr, nullptr, nullptr, s, false))) return rc;
        if (fuse_qkv) {
            TimeScope t(e, 0, s);
            if (W.attn_lora_a) {
                if ((rc = dh_linear_qkv_lora_rope_cache_bf16(e->xn, W.attn_w, n_tok, d, W.attn_lora_a, W.attn_lora_b, D.lora_scale,
                                                             D.rope_cos, D.rope_sin, e->tok_slot, e->tok_pos, e->qrot, kc, vtc, H, G,
                                                             hs, e->s_max, e->xa, s))) return rc;
            } else if ((rc = dh_linear_qkv_rope_cache_bf16(e->xn, W.attn_w, n_tok, d, nullptr, 48, nullptr, D.lora_scale, D.rope_cos,
                                                           D.rope_sin, e->tok_slot, e->tok_pos, e->qrot, kc, vtc, H, G, hs,
                                                           e->s_max, s))) return rc;
        } else {
        if (W.attn_lora_a) {
            if ((rc = linear(e, e->xn, W.attn_w, e->qkv, n_tok, e->qkv_dim, d, DH_EPI_LORA, nullptr, e->xa, 48,
                             W.attn_lora_b, d, d + e->kv_dim, nullptr, nullptr, nullptr, s, true))) return rc;
        } else {
            if ((rc = linear(e, e->xn, W.attn_w, e->qkv, n_tok, e->qkv_dim, d, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr,
                             0, 0, nullptr, nullptr, nullptr, s, true))) return rc;
        }
        if ((rc = dh_qkv_rope_cache_bf16(e->qkv, D.rope_cos, D.rope_sin, e->tok_slot, e->tok_pos, e->qrot, kc, vtc, nullptr,
                                         nullptr, n_tok, H, G, hs, e->s_max, s))) return rc;
        }
        if (prune_last && l == D.n_layer - 1) {
            // last block, last-position logits only: attention of each sequence's LAST query (a one-row tile at position
            // pos0 + len - 1 over the same 64-key steps as in the full call), then projection, norm and MLP on n_seq rows.  The
            // final hidden rows land in e->xlast (where the caller would have gathered them); e->x keeps this block's input.
            const uint8_t* rtl = e->rsqrt_vec > 0 ? e->last_tail : nullptr;
            {
                TimeScope t(e, 2, s);
                if ((rc = dh_attn_prefill_bf16(e->qrot, kc, vtc, seq_slot, e->last_row, e->last_meta, e->last_meta + e->max_batch,
                                               e->att, nullptr, n_seq, 1, H, G, hs, e->s_max, s))) return rc;
            }
            hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_seq, 4)), dim3(256), 0, s, e->att, e->last_row, e->att_last, n_seq, d);
            hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_seq, 4)), dim3(256), 0, s, e->x, e->last_row, e->xlast, n_seq, d);
            DH_LAUNCH_CHECK();
            // (these n_seq-row launches stream the weights; they are not in the timed class of the large prefill GEMMs, and
            // bench.py does not count their FLOPs either)
            if (W.proj_lora_a) {
                if ((rc = dh_linear_lora_impl(e->att_last, W.proj_w, e->xlast, n_seq, d, d, W.proj_lora_a, W.proj_lora_b, D.lora_scale, d, d,
                                              e->xlast, e->xa, 1, s))) return rc;
            } else {
                if ((rc = linear(e, e->att_last, W.proj_w, e->xlast, n_seq, d, d, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 0, 0,
                                 nullptr, nullptr, e->xlast, s, false))) return rc;
            }
            if ((rc = dh_rmsnorm_bf16(e->xlast, nullptr, W.norm_2, e->xn_last, nullptr, n_seq, d, D.norm_eps, rtl, s))) return rc;
            if ((rc = linear(e, e->xn_last, W.fc_1, e->act_last, n_seq, I, d, DH_EPI_SWIGLU, W.fc_2, nullptr, 0, nullptr, 0, 0, nullptr,
                             nullptr, nullptr, s, false))) return rc;
            if ((rc = linear(e, e->act_last, W.mlp_proj, e->xlast, n_seq, d, I, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 0, 0, nullptr,
                             nullptr, e->xlast, s, false))) return rc;
            break;
        }
        if (decode) {
            TimeScope t(e, 3, s);
            if ((rc = dh_attn_decode_bf16(e->qrot, kc, vtc, seq_slot, kv_pos0, e->att, e->dec_work, n_seq, H, G, hs,
                                          e->s_max, s))) return rc;
        } else {
            TimeScope t(e, 2, s);
            if ((rc = dh_attn_prefill_bf16(e->qrot, kc, vtc, seq_slot, q_start, q_len, kv_pos0, e->att, nullptr, n_seq,
                                           max_q_len, H, G, hs, e->s_max, s))) return rc;
        }
        if (W.proj_lora_a) {
            // down-projection in the GEMM's K loop where the launch runs on the 4-wave 256-tile kernel, else its own launch (same bits)
            TimeScope t(e, e->phase_decode ? 1 : 0, s);
            if ((rc = dh_linear_lora_impl(e->att, W.proj_w, e->x, n_tok, d, d, W.proj_lora_a, W.proj_lora_b, D.lora_scale, d, d, e->x,
                                          e->xa, e->phase_decode && !e->decode_tiled ? 2 : 1, s))) return rc;
        } else {
            if ((rc = linear(e, e->att, W.proj_w, e->x, n_tok, d, d, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 0, 0,
                             nullptr, nullptr, e->x, s, true))) return rc;
        }
        if ((rc = dh_rmsnorm_bf16(e->x, nullptr, W.norm_2, e->xn, nullptr, n_tok, d, D.norm_eps, rt, s))) return rc;
        if ((rc = linear(e, e->xn, W.fc_1, e->act, n_tok, I, d, DH_EPI_SWIGLU, W.fc_2, nullptr, 0, nullptr, 0, 0, nullptr,
                         nullptr, nullptr, s, true))) return rc;
        if ((rc = linear(e, e->act, W.mlp_proj, e->x, n_tok, d, I, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 0, 0, nullptr,
                         nullptr, e->x, s, true))) return rc;
    }
    return 0;
}

// fp8 GEMM kernel by PHASE, never by packing (the two kernels sum K in different fp32 orders): a prefill is tiled even
// when a short prompt runs alone; a single-token step streams the weights up to 128 rows (the bench's four batches per
// decode loop) and is tiled above that — the one documented class boundary of the fp8 decode phase.
inline int fp8_kernel(bool decode, int rows) { return decode && rows <= 128 ? 2 : 1; }

// fp8 serving: the same layer sequence with every dense product on the fp8 MFMA (csrc/fp8.hip).  Activations are
// quantised per token right where they are produced (the norm kernels) or by a pass over the attention / SwiGLU
// output; LoRA is merged into the weights before quantisation, so there is no rank-16 side product.  Prefill and
// decode run the same sequence; the GEMM kernel is pinned by phase (fp8_kernel above).
int run_layers_fp8(dh_engine* e, const int64_t* ids, int n_tok, int n_seq, int max_q_len, bool decode,
                   const uint8_t* tail_flags, hipStream_t s, bool prune_last = false) {
    const uint8_t* rt = e->rsqrt_vec > 0 ? tail_flags : nullptr;
    e->phase_decode = decode;
    const dh_model_desc& D = e->d;
    const int d = D.n_embd, I = D.intermediate, hs = D.head_size, H = D.n_head, G = D.n_groups;
    int32_t* seq_slot = e->seq_meta + e->slot_base;
    int32_t* q_start = e->seq_meta + e->max_batch;
    int32_t* q_len = e->seq_meta + 2 * e->max_batch;
    int32_t* kv_pos0 = e->seq_meta + 3 * e->max_batch;   // decode: kv_len
    int rc;
    // the weight pointers of dh_layer_weights address e4m3 bytes in this mode
    auto lin = [&](const bf16_t* w, const float* ws, bf16_t* y, int N, int K, int epi, const bf16_t* w2, const float* w2s,
                   const bf16_t* res) {
        return dh_linear_fp8_ex(e->xq, e->xscale, reinterpret_cast<const uint8_t*>(w), ws, y, n_tok, N, K, epi,
                                reinterpret_cast<const uint8_t*>(w2), w2s, nullptr, nullptr, res, fp8_kernel(decode, n_tok), s);
    };
    if ((rc = dh_embed_bf16(ids, D.wte, e->x, n_tok, d, D.wte_rows, s))) return rc;
    for (int l = 0; l < D.n_layer; ++l) {
        const dh_layer_weights& W = e->layers[l];
        bf16_t* kc = e->kc + (size_t)l * e->cache_layer_elems;
        bf16_t* vtc = e->vtc + (size_t)l * e->cache_layer_elems;
        if ((rc = dh_rmsnorm_quant_fp8(e->x, W.norm_1, nullptr, e->xq, e->xscale, n_tok, d, D.norm_eps, rt, s))) return rc;
        if (decode && n_tok <= 128) {     // every streaming-class step (fp8_kernel above): one family, no 32-row boundary
            // one launch for rope + cache append + split-KV attention + combine (decode_fused.hip): the QKV product
            // is handed over as its single fp32 "partial" (values already rounded to bf16), no LoRA (merged)
            {
                TimeScope t(e, 1, s);
                if ((rc = dh_linear_fp8_f32(e->xq, e->xscale, reinterpret_cast<const uint8_t*>(W.attn_w), W.attn_ws, e->part32,
                                            n_tok, e->qkv_dim, d, s))) return rc;
            }
            TimeScope t(e, 3, s);
            if ((rc = dh_attn_decode_fused_bf16(e->part32, 1, 0, n_seq, e->qkv_dim, 0, nullptr, 0.f, e->qkv_dim, e->qkv_dim,
                                                D.rope_cos, D.rope_sin, seq_slot, kv_pos0, kc, vtc, e->att, H, G, hs,
                                                e->s_max, s))) return rc;
        } else {
        {
            TimeScope t(e, decode ? 1 : 0, s);
            if ((rc = lin(W.attn_w, W.attn_ws, e->qkv, e->qkv_dim, d, DH_EPI_PLAIN, nullptr, nullptr, nullptr))) return rc;
        }
        if ((rc = dh_qkv_rope_cache_bf16(e->qkv, D.rope_cos, D.rope_sin, e->tok_slot, e->tok_pos, e->qrot, kc, vtc, nullptr,
                                         nullptr, n_tok, H, G, hs, e->s_max, s))) return rc;
        if (prune_last && l == D.n_layer - 1) {
            // last block of a prompt forward that wants the last position's logits only: see run_layers.  The products stay on the
            // phase's (tiled) kernel and the activations are quantised per row, so the n_seq rows keep their bits.
            const uint8_t* rtl = e->rsqrt_vec > 0 ? e->last_tail : nullptr;
            auto lin_last = [&](const bf16_t* w, const float* ws, bf16_t* y, int N, int K, int epi, const bf16_t* w2, const float* w2s,
                                const bf16_t* res) {
                return dh_linear_fp8_ex(e->xq, e->xscale, reinterpret_cast<const uint8_t*>(w), ws, y, n_seq, N, K, epi,
                                        reinterpret_cast<const uint8_t*>(w2), w2s, nullptr, nullptr, res, fp8_kernel(false, n_seq), s);
            };
            {
                TimeScope t(e, 2, s);
                if ((rc = dh_attn_prefill_bf16(e->qrot, kc, vtc, seq_slot, e->last_row, e->last_meta, e->last_meta + e->max_batch,
                                               e->att, nullptr, n_seq, 1, H, G, hs, e->s_max, s))) return rc;
            }
            hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_seq, 4)), dim3(256), 0, s, e->att, e->last_row, e->att_last, n_seq, d);
            hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_seq, 4)), dim3(256), 0, s, e->x, e->last_row, e->xlast, n_seq, d);
            DH_LAUNCH_CHECK();
            if ((rc = dh_quant_rows_fp8(e->att_last, e->xq, e->xscale, n_seq, d, s))) return rc;
            if ((rc = lin_last(W.proj_w, W.proj_ws, e->xlast, d, d, DH_EPI_PLAIN, nullptr, nullptr, e->xlast))) return rc;
            if ((rc = dh_rmsnorm_quant_fp8(e->xlast, W.norm_2, nullptr, e->xq, e->xscale, n_seq, d, D.norm_eps, rtl, s))) return rc;
            if ((rc = lin_last(W.fc_1, W.fc_1_ws, e->act_last, I, d, DH_EPI_SWIGLU, W.fc_2, W.fc_2_ws, nullptr))) return rc;
            if ((rc = dh_quant_rows_fp8(e->act_last, e->xq, e->xscale, n_seq, I, s))) return rc;
            if ((rc = lin_last(W.mlp_proj, W.mlp_proj_ws, e->xlast, d, I, DH_EPI_PLAIN, nullptr, nullptr, e->xlast))) return rc;
            break;
        }
        if (decode) {
            TimeScope t(e, 3, s);
            if ((rc = dh_attn_decode_bf16(e->qrot, kc, vtc, seq_slot, kv_pos0, e->att, e->dec_work, n_seq, H, G, hs,
                                          e->s_max, s))) return rc;
        } else {
            TimeScope t(e, 2, s);
            if ((rc = dh_attn_prefill_bf16(e->qrot, kc, vtc, seq_slot, q_start, q_len, kv_pos0, e->att, nullptr, n_seq,
                                           max_q_len, H, G, hs, e->s_max, s))) return rc;
        }
        }
        if ((rc = dh_quant_rows_fp8(e->att, e->xq, e->xscale, n_tok, d, s))) return rc;
        {
            TimeScope t(e, decode ? 1 : 0, s);
            if ((rc = lin(W.proj_w, W.proj_ws, e->x, d, d, DH_EPI_PLAIN, nullptr, nullptr, e->x))) return rc;
        }
        if ((rc = dh_rmsnorm_quant_fp8(e->x, W.norm_2, nullptr, e->xq, e->xscale, n_tok, d, D.norm_eps, rt, s))) return rc;
        {
            TimeScope t(e, decode ? 1 : 0, s);
            if ((rc = lin(W.fc_1, W.fc_1_ws, e->act, I, d, DH_EPI_SWIGLU, W.fc_2, W.fc_2_ws, nullptr))) return rc;
        }
        if ((rc = dh_quant_rows_fp8(e->act, e->xq, e->xscale, n_tok, I, s))) return rc;
        {
            TimeScope t(e, decode ? 1 : 0, s);
            if ((rc = lin(W.mlp_proj, W.mlp_proj_ws, e->x, d, I, DH_EPI_PLAIN, nullptr, nullptr, e->x))) return rc;
        }
    }
    return 0;
}

// ln_f + lm_head of `rows` rows of `xrows` in fp8 mode; the bf16 ln_f output lands in e->xn as in the bf16 path
int head_fp8(dh_engine* e, const bf16_t* xrows, int rows, bf16_t* logits, const uint8_t* rt, hipStream_t s) {
    const dh_model_desc& D = e->d;
    int rc;
    if ((rc = dh_rmsnorm_quant_fp8(xrows, D.ln_f, e->xn, e->xq, e->xscale, rows, D.n_embd, D.norm_eps,
                                   e->rsqrt_vec > 0 ? rt : nullptr, s))) return rc;
    TimeScope t(e, e->phase_decode ? 1 : 0, s);
    return dh_linear_fp8_ex(e->xq, e->xscale, reinterpret_cast<const uint8_t*>(D.lm_head), D.lm_head_ws, logits, rows, D.vocab,
                            D.n_embd, DH_EPI_ADAPTER, nullptr, nullptr, D.adapter_scale, D.adapter_bias, nullptr,
                            fp8_kernel(e->phase_decode, rows), s);
}

// K-slices of 8 (or 16 for long K) k-steps: the row-parallel streaming kernel (gemm_skinny.hip)
int pick_ksplit(int tiles, int nks) {
    (void)tiles;
    if (nks <= 128) return g_short_kps == 16 && nks % 16 == 0 ? nks / 16 : (nks + 7) / 8;
    if (nks <= 256) return (nks + 15) / 16;
    int ks = 1;
    while (ks < 4 && nks / (16 * ks) >= 2) ks *= 2;
    return ks;
}

// Single-token step for n_seq <= MAX_DECODE_ROWS sequences: 7 launches per layer (decode_fused.hip).  Leaves
// ln_f(x) in e->xn.  kv_len lives in seq_meta[3B..], seq_slot in seq_meta[0..].
int run_layers_decode(dh_engine* e, const int64_t* ids, int n_seq, const uint8_t* tail_flags, hipStream_t s) {
    const dh_model_desc& D = e->d;
    const int d = D.n_embd, I = D.intermediate, hs = D.head_size, H = D.n_head, G = D.n_groups;
    const uint8_t* rt = e->rsqrt_vec > 0 ? tail_flags : nullptr;
    const int32_t* seq_slot = e->seq_meta + e->slot_base;
    const int32_t* kv_len = e->seq_meta + 3 * e->max_batch;
    e->phase_decode = true;
    int rc;
    if ((rc = dh_embed_bf16(ids, D.wte, e->x, n_seq, d, D.wte_rows, s))) return rc;
    if ((rc = dh_rmsnorm_bf16(e->x, nullptr, e->layers[0].norm_1, e->xn, nullptr, n_seq, d, D.norm_eps, rt, s))) return rc;
    for (int l = 0; l < D.n_layer; ++l) {
        const dh_layer_weights& W = e->layers[l];
        bf16_t* kc = e->kc + (size_t)l * e->cache_layer_elems;
        bf16_t* vtc = e->vtc + (size_t)l * e->cache_layer_elems;
        // every partial-sum GEMM: the total in one launch (chain), the pair sums from the tiled split-K kernel (more than
        // 128 rows), or the K-slices from the streaming kernel — all in the family's one combine order (dualhyp_hip.h)
        auto partial = [&](const bf16_t* xin, const bf16_t* w, const bf16_t* wext, int N, int ext, int K, int ks, int& np, int& pairs) {
            if (dh_chain_ok(n_seq, N, ext, K, ks)) { np = 1; pairs = 0; return dh_linear_chain_bf16(xin, w, wext, e->part32, n_seq, N, ext, K, ks, s); }
            if (dh_pairs_ok(n_seq, N, ext, K, ks)) { np = (ks + 1) / 2; pairs = 0; return dh_linear_partial_pairs_bf16(xin, w, wext, e->part32, n_seq, N, ext, K, ks, s); }
            np = ks; pairs = 1;
            return dh_linear_partial_bf16(xin, w, wext, e->part32, n_seq, N, ext, K, ks, s);
        };
        int np, pairs;
        const int ext1 = W.attn_lora_a ? 48 : 0;
        if ((rc = partial(e->xn, W.attn_w, W.attn_lora_a, e->qkv_dim, ext1, d, pick_ksplit((e->qkv_dim + ext1) / 16, d / 32), np, pairs))) return rc;
        if ((rc = dh_attn_decode_fused_bf16(e->part32, np, pairs, n_seq, e->qkv_dim, ext1, W.attn_lora_b, D.lora_scale, d,
                                            d + e->kv_dim, D.rope_cos, D.rope_sin, seq_slot, kv_len, kc, vtc, e->att, H, G,
                                            hs, e->s_max, s))) return rc;
        const int ext2 = W.proj_lora_a ? 16 : 0;
        if ((rc = partial(e->att, W.proj_w, W.proj_lora_a, d, ext2, d, pick_ksplit((d + ext2) / 16, d / 32), np, pairs))) return rc;
        if ((rc = dh_finish_norm_bf16(e->part32, np, pairs, n_seq, d, ext2, W.proj_lora_b, D.lora_scale, e->x, W.norm_2, e->x,
                                      e->xn, D.norm_eps, rt, s))) return rc;
        if ((rc = linear(e, e->xn, W.fc_1, e->act, n_seq, I, d, DH_EPI_SWIGLU, W.fc_2, nullptr, 0, nullptr, 0, 0, nullptr,
                         nullptr, nullptr, s, false))) return rc;
        if ((rc = partial(e->act, W.mlp_proj, nullptr, d, 0, I, pick_ksplit(d / 16, I / 32), np, pairs))) return rc;
        const bf16_t* next_norm = l + 1 < D.n_layer ? e->layers[l + 1].norm_1 : D.ln_f;
        if ((rc = dh_finish_norm_bf16(e->part32, np, pairs, n_seq, d, 0, nullptr, 0.f, e->x, next_norm, e->x, e->xn,
                                      D.norm_eps, rt, s))) return rc;
    }
    return 0;
}

// lm_head on rows that are already ln_f-normalised (e->xn)
int head_normed(dh_engine* e, int rows, bf16_t* logits, hipStream_t s) {
    const dh_model_desc& D = e->d;
    return linear(e, e->xn, D.lm_head, logits, rows, D.vocab, D.n_embd, DH_EPI_ADAPTER, nullptr, nullptr, 0, nullptr, 0, 0,
                  D.adapter_scale, D.adapter_bias, nullptr, s, true);
}

int head(dh_engine* e, const bf16_t* xrows, int rows, bf16_t* logits, const uint8_t* rt, hipStream_t s) {
    const dh_model_desc& D = e->d;
    int rc;
    if ((rc = dh_rmsnorm_bf16(xrows, nullptr, D.ln_f, e->xn, nullptr, rows, D.n_embd, D.norm_eps,
                              e->rsqrt_vec > 0 ? rt : nullptr, s))) return rc;
    return linear(e, e->xn, D.lm_head, logits, rows, D.vocab, D.n_embd, DH_EPI_ADAPTER, nullptr, nullptr, 0, nullptr, 0, 0,
                  D.adapter_scale, D.adapter_bias, nullptr, s, true);
}

int decode_step(dh_engine* e, int64_t* tokens, int tok_ld, int32_t* length, int32_t* done, int n_seq, float temperature,
                int top_k, int64_t eos_id, uint64_t seed, hipStream_t s);
int engine_init(dh_engine* e, const dh_model_desc* desc, int max_batch, int s_max, int max_tokens);

}  // namespace

extern "C" int dh_engine_create(const dh_model_desc* desc, int max_batch, int s_max, int max_tokens, dh_engine** out) {
    DH_CHECK(desc && out, "dh_engine_create: null argument");
    DH_CHECK(desc->head_size == 64 || desc->head_size == 128, "dh_engine_create: head_size %d unsupported", desc->head_size);
    DH_CHECK(desc->n_head % desc->n_groups == 0, "dh_engine_create: n_head %% n_groups != 0");
    DH_CHECK(desc->n_embd == desc->n_head * desc->head_size, "dh_engine_create: n_embd != n_head*head_size");
    DH_CHECK(desc->n_embd % 64 == 0 && desc->intermediate % 64 == 0 && desc->vocab % 8 == 0, "dh_engine_create: dims must be multiples of 64");
    DH_CHECK(s_max > 0 && s_max % 64 == 0 && s_max <= (desc->block_size + 63) / 64 * 64, "dh_engine_create: s_max=%d must be a multiple of 64 and <= block_size rounded up", s_max);
    DH_CHECK(max_batch > 0 && max_tokens >= max_batch, "dh_engine_create: bad batch/token capacity");
    DH_CHECK(desc->rope_cos && desc->rope_sin && desc->wte && desc->ln_f && desc->lm_head && desc->h_layers, "dh_engine_create: null weight pointer");
    dh_engine* e = new dh_engine();
    const int rc = engine_init(e, desc, max_batch, s_max, max_tokens);
    if (rc) { dh_engine_destroy(e); return rc; }   // one cleanup path: nothing allocated so far leaks
    *out = e;
    return 0;
}

namespace {

int engine_init(dh_engine* e, const dh_model_desc* desc, int max_batch, int s_max, int max_tokens) {
    e->d = *desc;
    e->layers.assign(desc->h_layers, desc->h_layers + desc->n_layer);
    e->d.h_layers = nullptr;
    e->max_batch = max_batch; e->s_max = s_max; e->max_tokens = max_tokens;
    const int d = desc->n_embd, hs = desc->head_size, G = desc->n_groups, H = desc->n_head;
    e->kv_dim = G * hs;
    e->qkv_dim = (H + 2 * G) * hs;
    e->fp8 = e->layers[0].attn_ws != nullptr;
    if (e->fp8) {
        for (const auto& L : e->layers)
            DH_CHECK(L.attn_ws && L.proj_ws && L.fc_1_ws && L.fc_2_ws && L.mlp_proj_ws && !L.attn_lora_a && !L.proj_lora_a,
                     "dh_engine_create: fp8 mode needs every channel-scale pointer and merged LoRA (no lora_a/lora_b)");
        DH_CHECK(desc->lm_head_ws != nullptr, "dh_engine_create: fp8 mode needs lm_head_ws");
        DH_CHECK(d % 128 == 0 && desc->intermediate % 128 == 0, "dh_engine_create: fp8 mode needs n_embd and intermediate %% 128 == 0");
    }
    e->cache_layer_elems = (size_t)max_batch * G * s_max * hs;
    const size_t T = max_tokens;
    int rc = 0;
    rc |= dmalloc(e, &e->kc, e->cache_layer_elems * desc->n_layer);
    rc |= dmalloc(e, &e->vtc, e->cache_layer_elems * desc->n_layer);
    rc |= dmalloc(e, &e->x, T * d);
    rc |= dmalloc(e, &e->xn, T * d);
    rc |= dmalloc(e, &e->qkv, T * e->qkv_dim);
    rc |= dmalloc(e, &e->qrot, T * d);
    rc |= dmalloc(e, &e->att, T * d);
    rc |= dmalloc(e, &e->xa, T * 48);
    rc |= dmalloc(e, &e->act, T * desc->intermediate);
    rc |= dmalloc(e, &e->xlast, (size_t)max_batch * d);
    rc |= dmalloc(e, &e->logits, (size_t)max_batch * desc->vocab);
    rc |= dmalloc(e, &e->tok_slot, T);
    rc |= dmalloc(e, &e->tok_pos, T);
    rc |= dmalloc(e, &e->seq_meta, (size_t)4 * max_batch);
    rc |= dmalloc(e, &e->last_row, (size_t)max_batch);
    rc |= dmalloc(e, &e->last_meta, (size_t)2 * max_batch);
    rc |= dmalloc(e, &e->att_last, (size_t)max_batch * d);
    rc |= dmalloc(e, &e->xn_last, (size_t)max_batch * d);
    rc |= dmalloc(e, &e->act_last, (size_t)max_batch * desc->intermediate);
    rc |= dmalloc(e, &e->step_dev, 1);
    rc |= dmalloc(e, &e->dec_ids, (size_t)max_batch);
    rc |= dmalloc(e, &e->part32, (size_t)16 * (max_batch < 32 ? 32 : (max_batch < MAX_DECODE_ROWS ? max_batch : MAX_DECODE_ROWS)) * (e->qkv_dim + 48));
    if (e->fp8) {
        rc |= dmalloc(e, &e->xq, T * (size_t)(desc->intermediate > d ? desc->intermediate : d));
        rc |= dmalloc(e, &e->xscale, T);
    }
    rc |= dmalloc(e, &e->row_tail, T);
    rc |= dmalloc(e, &e->last_tail, (size_t)max_batch);
    rc |= dmalloc(e, &e->ones, (size_t)max_batch);
    const int64_t wb = dh_attn_decode_work_bytes(max_batch, H, hs, s_max);
    if (!rc) { hipError_t he = hipMalloc(&e->dec_work, wb); if (he != hipSuccess) rc = 2; e->dev_bytes += wb; }
    if (rc) { dh_set_error("dh_engine_create: device allocation failed (%s)", dh_last_error()); return 2; }
    // the attention kernels rely on finite (zero) cache contents beyond the written positions
    DH_HIP(hipMemset(e->kc, 0, e->cache_layer_elems * desc->n_layer * sizeof(bf16_t)));
    DH_HIP(hipMemset(e->vtc, 0, e->cache_layer_elems * desc->n_layer * sizeof(bf16_t)));
    DH_HIP(hipMemset(e->step_dev, 0, sizeof(int32_t)));
    DH_HIP(hipMemset(e->ones, 1, (size_t)max_batch));
    DH_HIP(hipHostMalloc((void**)&e->h_stage, (3 * T + 8 * (size_t)max_batch) * sizeof(int32_t)));
    DH_HIP(hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking));
    DH_HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    DH_HIP(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
    DH_HIP(hipEventCreateWithFlags(&e->ev_stage, hipEventDisableTiming));
    hipLaunchKernelGGL(iota_i32_kernel, dim3(cdiv(max_batch, 64)), dim3(64), 0, 0, e->seq_meta, max_batch);
    DH_HIP(hipDeviceSynchronize());
    return 0;
}

}  // namespace

extern "C" void dh_engine_destroy(dh_engine* e) {
    if (!e) return;
    for (auto& g : e->graphs) hipGraphExecDestroy(g.exec);
    void* ptrs[] = {e->kc, e->vtc, e->x, e->xn, e->qkv, e->qrot, e->att, e->xa, e->act, e->xlast, e->logits,
                    e->tok_slot, e->tok_pos, e->seq_meta, e->last_row, e->last_meta, e->att_last, e->xn_last, e->act_last, e->step_dev, e->dec_ids, e->dec_work,
                    e->row_tail, e->last_tail, e->ones, e->part32, e->xq, e->xscale};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (e->h_stage) hipHostFree(e->h_stage);
    if (e->gstream) hipStreamDestroy(e->gstream);
    if (e->ev_in) hipEventDestroy(e->ev_in);
    if (e->ev_out) hipEventDestroy(e->ev_out);
    if (e->ev_stage) hipEventDestroy(e->ev_stage);
    for (auto& v : e->tm.ev)
        for (auto& p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    delete e;
}

extern "C" int64_t dh_engine_device_bytes(const dh_engine* e) { return e ? e->dev_bytes : 0; }
extern "C" int dh_engine_read(dh_engine* e, int what, int layer, void* dst, int64_t n_bytes, void* stream) {
    DH_CHECK(e && dst && n_bytes >= 0, "dh_engine_read: bad argument");
    const void* src = nullptr;
    int64_t avail = 0;
    const int64_t cache_bytes = (int64_t)e->cache_layer_elems * sizeof(bf16_t);
    switch (what) {
        case 0: src = e->xn; avail = (int64_t)e->max_tokens * e->d.n_embd * 2; break;
        case 3: src = e->x; avail = (int64_t)e->max_tokens * e->d.n_embd * 2; break;
        case 1: case 2:
            DH_CHECK(layer >= 0 && layer < e->d.n_layer, "dh_engine_read: layer %d out of range", layer);
            src = (what == 1 ? e->kc : e->vtc) + (size_t)layer * e->cache_layer_elems;
            avail = cache_bytes;
            break;
        default: DH_CHECK(false, "dh_engine_read: unknown selector %d", what);
    }
    DH_CHECK(n_bytes <= avail, "dh_engine_read: %lld bytes requested, %lld available", (long long)n_bytes, (long long)avail);
    DH_HIP(hipMemcpyAsync(dst, src, n_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int dh_engine_forward(dh_engine* e, const int64_t* ids, const int32_t* h_seq_len, const int32_t* h_pos0,
                                 int n_seq, dh_bf16* logits_all, dh_bf16* logits_last, void* stream) {
    return dh_engine_forward_at(e, ids, h_seq_len, h_pos0, n_seq, 0, logits_all, logits_last, stream);
}

extern "C" int dh_engine_forward_at(dh_engine* e, const int64_t* ids, const int32_t* h_seq_len, const int32_t* h_pos0,
                                    int n_seq, int slot_base, dh_bf16* logits_all, dh_bf16* logits_last, void* stream) {
    DH_CHECK(e && ids && h_seq_len && h_pos0, "dh_engine_forward: null argument");
    DH_CHECK(n_seq > 0 && slot_base >= 0 && slot_base + n_seq <= e->max_batch,
             "dh_engine_forward: sequences [%d, %d) exceed max_batch=%d", slot_base, slot_base + n_seq, e->max_batch);
    e->slot_base = slot_base;
    hipStream_t s = (hipStream_t)stream;
    int n_tok = 0, max_q = 0;
    for (int i = 0; i < n_seq; ++i) {
        DH_CHECK(h_seq_len[i] > 0 && h_pos0[i] >= 0, "dh_engine_forward: sequence %d has length %d at position %d", i, h_seq_len[i], h_pos0[i]);
        DH_CHECK(h_pos0[i] + h_seq_len[i] <= e->s_max, "Cannot forward sequence %d: %d tokens at position %d exceed the KV cache length %d",
                 i, h_seq_len[i], h_pos0[i], e->s_max);
        n_tok += h_seq_len[i];
        max_q = h_seq_len[i] > max_q ? h_seq_len[i] : max_q;
    }
    DH_CHECK(n_tok <= e->max_tokens, "dh_engine_forward: %d tokens exceed the workspace capacity %d", n_tok, e->max_tokens);
    // metadata: [tok_slot | tok_pos | seq_slot | q_start | q_len | kv_pos0 | last_row]
    const int B = e->max_batch;
    int32_t* hs_ = e->h_stage;
    DH_HIP(hipEventSynchronize(e->ev_stage));   // this engine's previous metadata upload must have left the pinned buffer
    int32_t *h_slot = hs_, *h_pos = hs_ + e->max_tokens, *h_meta = hs_ + 2 * (size_t)e->max_tokens;
    int t = 0;
    for (int i = 0; i < n_seq; ++i) {
        h_meta[i] = i;
        h_meta[B + i] = t;
        h_meta[2 * B + i] = h_seq_len[i];
        h_meta[3 * B + i] = max_q == 1 ? h_pos0[i] + 1 : h_pos0[i];   // single-token call: kv_len
        for (int j = 0; j < h_seq_len[i]; ++j, ++t) { h_slot[t] = slot_base + i; h_pos[t] = h_pos0[i] + j; }
        h_meta[4 * B + i] = t - 1;
    }
    // Q11: rows torch's CPU bf16 rsqrt would process in its scalar tail loop
    uint8_t* h_tail = reinterpret_cast<uint8_t*>(hs_ + 2 * (size_t)e->max_tokens + 5 * (size_t)B);
    uint8_t* h_last_tail = h_tail + e->max_tokens;
    if (e->rsqrt_vec > 0) {
        const int V = e->rsqrt_vec;
        int r = 0;
        for (int i = 0; i < n_seq; ++i) {
            for (int j = 0; j < h_seq_len[i]; ++j, ++r)
                h_tail[r] = e->rsqrt_whole ? (r >= n_tok / V * V) : (j >= h_seq_len[i] / V * V);
            h_last_tail[i] = h_tail[r - 1];   // ln_f runs on all rows in the reference: the last row keeps its flag
        }
        DH_HIP(hipMemcpyAsync(e->row_tail, h_tail, n_tok, hipMemcpyHostToDevice, s));
        DH_HIP(hipMemcpyAsync(e->last_tail, h_last_tail, n_seq, hipMemcpyHostToDevice, s));
    }
    DH_HIP(hipMemcpyAsync(e->tok_slot, h_slot, n_tok * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DH_HIP(hipMemcpyAsync(e->tok_pos, h_pos, n_tok * sizeof(int32_t), hipMemcpyHostToDevice, s));
    // seq_meta[0..B) (slot of sequence i = i) is written once at engine creation and left alone
    DH_HIP(hipMemcpyAsync(e->seq_meta + B, h_meta + B, 3 * B * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DH_HIP(hipMemcpyAsync(e->last_row, h_meta + 4 * B, B * sizeof(int32_t), hipMemcpyHostToDevice, s));
    // last-position logits of a multi-token forward only: the last block runs on the sequences' last rows (g_prune_last_layer)
    const bool prune_last = g_prune_last_layer && max_q > 1 && logits_all == nullptr && logits_last != nullptr;
    if (prune_last) {
        int32_t* h_lm = hs_ + 3 * (size_t)e->max_tokens + 6 * (size_t)B;   // behind the tail flags
        for (int i = 0; i < n_seq; ++i) { h_lm[i] = 1; h_lm[B + i] = h_pos0[i] + h_seq_len[i] - 1; }
        DH_HIP(hipMemcpyAsync(e->last_meta, h_lm, 2 * B * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    DH_HIP(hipEventRecord(e->ev_stage, s));
    int rc;
    // one token per sequence == a decode step (what generate()'s loop issues): same kernels as dh_engine_decode
    e->decode_tiled = max_q == 1 && g_decode_tiled_rows > 0 && n_seq >= g_decode_tiled_rows;
    if (e->fp8) {
        if ((rc = run_layers_fp8(e, ids, n_tok, n_seq, max_q, max_q == 1, e->row_tail, s, prune_last))) return rc;
        e->last_ntok = n_tok;
        if (prune_last) return head_fp8(e, e->xlast, n_seq, logits_last, e->last_tail, s);
        if (logits_all && (rc = head_fp8(e, e->x, n_tok, logits_all, e->row_tail, s))) return rc;
        if (logits_last) {
            hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_seq, 4)), dim3(256), 0, s, e->x, e->last_row, e->xlast, n_seq,
                               e->d.n_embd);
            DH_LAUNCH_CHECK();
            if ((rc = head_fp8(e, e->xlast, n_seq, logits_last, e->last_tail, s))) return rc;
        }
        return 0;
    }
    const bool fast = max_q == 1 && n_seq <= MAX_DECODE_ROWS && e->d.n_embd % 16 == 0 && !e->decode_tiled;
    if (fast) {
        if ((rc = run_layers_decode(e, ids, n_seq, e->row_tail, s))) return rc;
        e->last_ntok = n_tok;
        if (logits_all && (rc = head_normed(e, n_seq, logits_all, s))) return rc;
        if (logits_last && (rc = head_normed(e, n_seq, logits_last, s))) return rc;
        return 0;
    }
    if ((rc = run_layers(e, ids, n_tok, n_seq, max_q, max_q == 1, e->row_tail, s, prune_last))) return rc;
    e->last_ntok = n_tok;
    if (prune_last) return head(e, e->xlast, n_seq, logits_last, e->last_tail, s);   // e->xlast: the last rows' final hidden state
    if (logits_all) {
        // ln_f output lands in e->xn (test hook dh_engine_hidden)
        if ((rc = head(e, e->x, n_tok, logits_all, e->row_tail, s))) return rc;
    }
    if (logits_last) {
        hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_seq, 4)), dim3(256), 0, s, e->x, e->last_row, e->xlast, n_seq,
                           e->d.n_embd);
        DH_LAUNCH_CHECK();
        if (logits_all) {
            // xn currently holds ln_f(x) of all rows; recompute for the gathered rows into a scratch
            // region past them is unnecessary: head() overwrites xn[0:n_seq], which is fine after logits_all.
        }
        if ((rc = head(e, e->xlast, n_seq, logits_last, e->last_tail, s))) return rc;
    }
    return 0;
}

namespace {

int decode_step(dh_engine* e, int64_t* tokens, int tok_ld, int32_t* length, int32_t* done, int n_seq, float temperature,
                int top_k, int64_t eos_id, uint64_t seed, hipStream_t s) {
    int32_t* kv_len = e->seq_meta + 3 * e->max_batch;
    hipLaunchKernelGGL(decode_prep_kernel, dim3(cdiv(n_seq, 64)), dim3(64), 0, s, tokens, tok_ld, length, e->dec_ids,
                       e->tok_slot, e->tok_pos, kv_len, e->step_dev, n_seq, e->s_max);
    DH_LAUNCH_CHECK();
    int rc;
    e->decode_tiled = g_decode_tiled_rows > 0 && n_seq >= g_decode_tiled_rows;
    if (e->fp8) {
        if ((rc = run_layers_fp8(e, e->dec_ids, n_seq, n_seq, 1, true, e->ones, s))) return rc;
        if ((rc = head_fp8(e, e->x, n_seq, e->logits, e->ones, s))) return rc;
    } else if (n_seq <= MAX_DECODE_ROWS && !e->decode_tiled) {
        if ((rc = run_layers_decode(e, e->dec_ids, n_seq, e->ones, s))) return rc;
        if ((rc = head_normed(e, n_seq, e->logits, s))) return rc;
    } else {
        if ((rc = run_layers(e, e->dec_ids, n_seq, n_seq, 1, true, e->ones, s))) return rc;
        if ((rc = head(e, e->x, n_seq, e->logits, e->ones, s))) return rc;
    }
    // the per-step RNG counter lives in step_dev (incremented by decode_prep_kernel)
    return dh_sample_impl(e->logits, e->d.vocab, tokens, tok_ld, length, done, n_seq, temperature, top_k, eos_id,
                          seed, 0, e->step_dev, s);
}

}  // namespace

extern "C" int dh_engine_decode(dh_engine* e, int64_t* tokens, int tok_ld, int32_t* length, int32_t* done, int n_seq,
                                int n_steps, float temperature, int top_k, int64_t eos_id, uint64_t seed,
                                int first_step, void* stream) {
    DH_CHECK(e && tokens && length && done, "dh_engine_decode: null argument");
    DH_CHECK(n_seq > 0 && n_seq <= e->max_batch, "dh_engine_decode: n_seq=%d exceeds max_batch=%d", n_seq, e->max_batch);
    DH_CHECK(temperature > 0.f && top_k >= 0, "dh_engine_decode: bad sampling parameters");
    if (n_steps <= 0) return 0;
    e->slot_base = 0;
    hipStream_t s = (hipStream_t)stream;
    // seq_slot (seq_meta[0..B)) is the identity from engine creation on; nothing here touches the host
    // staging buffer, so the call never waits for the stream (several engines can be driven back to back)
    hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, s, e->step_dev, (int32_t)first_step);
    DH_LAUNCH_CHECK();
    // rsqrt_vec: `rt = rsqrt_vec > 0 ? flags : nullptr` is resolved while capturing, so it is part of the key
    const dh_engine::GKey key{tokens, tok_ld, length, done, n_seq, top_k, temperature, eos_id, seed, e->rsqrt_vec, g_decode_tiled_rows};
    hipGraphExec_t gexec = nullptr;
    for (auto& g : e->graphs) {
        const dh_engine::GKey& k = g.key;
        if (k.tokens == key.tokens && k.tok_ld == key.tok_ld && k.length == key.length && k.done == key.done &&
            k.n_seq == key.n_seq && k.top_k == key.top_k && k.temp == key.temp && k.eos == key.eos && k.seed == key.seed &&
            k.rsqrt_vec == key.rsqrt_vec && k.tiled_rows == key.tiled_rows) {
            gexec = g.exec;
            g.used = ++e->graph_clock;
            break;
        }
    }
    // hand over from the caller's stream to the engine's capture-capable stream
    DH_HIP(hipEventRecord(e->ev_in, s));
    DH_HIP(hipStreamWaitEvent(e->gstream, e->ev_in, 0));
    if (!gexec) {
        hipGraph_t graph = nullptr;
        DH_HIP(hipStreamBeginCapture(e->gstream, hipStreamCaptureModeThreadLocal));
        e->capturing = true;
        int rc = decode_step(e, tokens, tok_ld, length, done, n_seq, temperature, top_k, eos_id, seed, e->gstream);
        e->capturing = false;
        hipError_t ce = hipStreamEndCapture(e->gstream, &graph);
        if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
        DH_CHECK(ce == hipSuccess && graph, "dh_engine_decode: graph capture failed: %s", hipGetErrorString(ce));
        hipError_t ie = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        DH_CHECK(ie == hipSuccess, "dh_engine_decode: hipGraphInstantiate failed: %s", hipGetErrorString(ie));
        if (e->graphs.size() >= 8) {   // evict the least recently used (its launches are stream-ordered before the destroy)
            size_t lru = 0;
            for (size_t i = 1; i < e->graphs.size(); ++i)
                if (e->graphs[i].used < e->graphs[lru].used) lru = i;
            DH_HIP(hipStreamSynchronize(e->gstream));
            hipGraphExecDestroy(e->graphs[lru].exec);
            e->graphs.erase(e->graphs.begin() + lru);
        }
        e->graphs.push_back({key, gexec, ++e->graph_clock});
    }
    for (int i = 0; i < n_steps; ++i) DH_HIP(hipGraphLaunch(gexec, e->gstream));
    DH_HIP(hipEventRecord(e->ev_out, e->gstream));
    DH_HIP(hipStreamWaitEvent(s, e->ev_out, 0));
    return 0;
}

extern "C" int dh_engine_set_cpu_rsqrt_emulation(dh_engine* e, int vec_width, int whole_call) {
    DH_CHECK(e && vec_width >= 0, "dh_engine_set_cpu_rsqrt_emulation: bad argument");
    e->rsqrt_vec = vec_width;
    e->rsqrt_whole = whole_call != 0;
    return 0;
}

extern "C" int dh_engine_set_timing(dh_engine* e, int on) {
    DH_CHECK(e, "dh_engine_set_timing: null engine");
    e->tm.on = on != 0;
    for (auto& v : e->tm.ev) {
        for (auto& p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
        v.clear();
    }
    return 0;
}

extern "C" int dh_engine_get_timing(dh_engine* e, int which, double* h_ms, int64_t* h_launches) {
    DH_CHECK(e && which >= 0 && which < 4 && h_ms && h_launches, "dh_engine_get_timing: bad argument");
    DH_HIP(hipDeviceSynchronize());
    double tot = 0;
    for (auto& p : e->tm.ev[which]) {
        float ms = 0;
        DH_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        tot += ms;
    }
    *h_ms = tot;
    *h_launches = (int64_t)e->tm.ev[which].size();
    return 0;
}
