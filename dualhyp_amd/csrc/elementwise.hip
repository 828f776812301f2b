// Bandwidth-bound kernels of the decoder: embedding gather, RMSNorm (+fused residual add),
// fused QKV split + rotary + KV-cache append.  All 16-byte vectorised, one rounding to bf16 at
// each point where the reference's eager bf16 program rounds.
#include "common.h"

// ------------------------------------------------------------------------------ embedding
// one wave per token; ger/lora.py:537
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids,
                                                    const bf16_t* __restrict__ wte,
                                                    bf16_t* __restrict__ out, int n_tok, int d, int vocab) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n_tok) return;
    int64_t id = ids[wave];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never fault on a bad id
    const uint4* src = reinterpret_cast<const uint4*>(wte + (size_t)id * d);
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)wave * d);
    for (int c = lane; c < d / 8; c += 64) dst[c] = src[c];
}

extern "C" int dh_embed_bf16(const int64_t* ids, const dh_bf16* wte, dh_bf16* out, int n_tok, int d,
                             int vocab, void* stream) {
    DH_CHECK(n_tok >= 0 && d > 0 && d % 8 == 0 && vocab > 0, "dh_embed_bf16: bad shape n_tok=%d d=%d", n_tok, d);
    if (n_tok == 0) return 0;
    hipLaunchKernelGGL(embed_kernel, dim3(cdiv(n_tok, 4)), dim3(256), 0, (hipStream_t)stream, ids, wte, out,
                       n_tok, d, vocab);
    DH_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ RMSNorm
// ger/rmsnorm.py:17-21 in bf16:  ms=mean(x*x) ; xn = x*rsqrt(ms+eps) ; out = w*xn, every
// intermediate tensor rounded to bf16 (the mean is reduced in fp32 and rounded once, as
// torch's CPU sum does).  One wave per row, the row lives in registers (d <= 4096).
template <int MAXC, bool RESID>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x,
                                                      const bf16_t* __restrict__ resid,
                                                      const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out,
                                                      bf16_t* __restrict__ sum_out, int rows, int d, float eps,
                                                      const uint8_t* __restrict__ row_tail) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nchunk = d >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * d);
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            uint4 u = xr[c];
            const bf16_t* p = reinterpret_cast<const bf16_t*>(&u);
            if (RESID) {
                uint4 r = reinterpret_cast<const uint4*>(resid + (size_t)row * d)[c];
                const bf16_t* q = reinterpret_cast<const bf16_t*>(&r);
                uint4 s4;
                bf16_t* sp = reinterpret_cast<bf16_t*>(&s4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sp[j] = f2bf(bf2f(p[j]) + bf2f(q[j]));
                    v[i][j] = bf2f(sp[j]);
                }
                if (sum_out) reinterpret_cast<uint4*>(sum_out + (size_t)row * d)[c] = s4;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = bf2f(p[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += rbf(v[i][j] * v[i][j]);
        }
    }
    ss = wave_sum(ss);
    const float ms = rbf(ss / (float)d);           // fp32 sum / d, ONE rounding (verified vs torch CPU)
    const float t = rbf(ms + eps);
    // torch's CPU bf16 rsqrt: vector loop = one rounding; scalar tail loop rounds sqrt(t) first (Q11)
    const bool tail = row_tail != nullptr && row_tail[row] != 0;
    const float r = tail ? rbf(1.0f / rbf(sqrtf(t))) : rbf(1.0f / sqrtf(t));
    const uint4* wr = reinterpret_cast<const uint4*>(w);
    uint4* orow = reinterpret_cast<uint4*>(out + (size_t)row * d);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            uint4 wu = wr[c];
            const bf16_t* wp = reinterpret_cast<const bf16_t*>(&wu);
            uint4 o;
            bf16_t* op = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
            for (int j = 0; j < 8; ++j) op[j] = f2bf(bf2f(wp[j]) * rbf(v[i][j] * r));
            orow[c] = o;
        }
    }
}

extern "C" int dh_rmsnorm_bf16(const dh_bf16* x, const dh_bf16* resid, const dh_bf16* w, dh_bf16* out,
                               dh_bf16* sum_out, int rows, int d, float eps, const uint8_t* row_tail,
                               void* stream) {
    DH_CHECK(rows >= 0 && d > 0 && d % 8 == 0 && d <= 8192, "dh_rmsnorm_bf16: unsupported d=%d (need d%%8==0, d<=8192)", d);
    if (rows == 0) return 0;
    dim3 grid(cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(MAXC)                                                                                   \
    if (resid)                                                                                         \
        hipLaunchKernelGGL((rmsnorm_kernel<MAXC, true>), grid, block, 0, s, x, resid, w, out, sum_out, rows, d, eps, row_tail); \
    else                                                                                               \
        hipLaunchKernelGGL((rmsnorm_kernel<MAXC, false>), grid, block, 0, s, x, resid, w, out, sum_out, rows, d, eps, row_tail)
    if (d <= 512) { LAUNCH(1); }
    else if (d <= 2048) { LAUNCH(4); }
    else if (d <= 4096) { LAUNCH(8); }
    else { LAUNCH(16); }
#undef LAUNCH
    DH_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------ QKV split + RoPE + cache
// ger/model.py:216-259.  grid (ceil(n_tok/64), n_groups), 256 threads.
//   rope (ger/model.py:349-355): out = bf16( bf16(x*cos) + bf16(rot*sin) ), rot = [-x2, x1]
template <int HS>
__global__ __launch_bounds__(256) void qkv_rope_cache_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ cos, const bf16_t* __restrict__ sin,
    const int32_t* __restrict__ tok_slot, const int32_t* __restrict__ tok_pos, bf16_t* __restrict__ q_out,
    bf16_t* __restrict__ k_cache, bf16_t* __restrict__ vT_cache, bf16_t* __restrict__ k_out, bf16_t* __restrict__ v_out,
    int n_tok, int n_head, int n_groups, int s_max) {
    constexpr int HALF = HS / 2;
    constexpr int CPH = HALF / 8;              // 16-B chunk pairs per head
    const int g = blockIdx.y;
    const int t0 = blockIdx.x * 64;
    const int q_per_kv = n_head / n_groups;
    const int row_elems = n_groups * (q_per_kv + 2) * HS;
    const int grp_off = g * (q_per_kv + 2) * HS;
    __shared__ __attribute__((aligned(16))) bf16_t vt[64][HS + 8];

    // phase 1: rotate q heads and k; write q_out / k_cache
    const int items = 64 * (q_per_kv + 1) * CPH;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int c = it % CPH;
        const int j = (it / CPH) % (q_per_kv + 1);   // 0..q_per_kv-1: q head, q_per_kv: k
        const int tl = it / (CPH * (q_per_kv + 1));
        const int t = t0 + tl;
        if (t >= n_tok) continue;
        const int pos = tok_pos[t];
        const bf16_t* src = qkv + (size_t)t * row_elems + grp_off + j * HS;
        uint4 a = *reinterpret_cast<const uint4*>(src + c * 8);
        uint4 b = *reinterpret_cast<const uint4*>(src + HALF + c * 8);
        uint4 c1 = *reinterpret_cast<const uint4*>(cos + (size_t)pos * HS + c * 8);
        uint4 c2 = *reinterpret_cast<const uint4*>(cos + (size_t)pos * HS + HALF + c * 8);
        uint4 s1 = *reinterpret_cast<const uint4*>(sin + (size_t)pos * HS + c * 8);
        uint4 s2 = *reinterpret_cast<const uint4*>(sin + (size_t)pos * HS + HALF + c * 8);
        const bf16_t *ap = (const bf16_t*)&a, *bp = (const bf16_t*)&b, *c1p = (const bf16_t*)&c1,
                     *c2p = (const bf16_t*)&c2, *s1p = (const bf16_t*)&s1, *s2p = (const bf16_t*)&s2;
        uint4 o1, o2;
        bf16_t *o1p = (bf16_t*)&o1, *o2p = (bf16_t*)&o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x1 = bf2f(ap[e]), x2 = bf2f(bp[e]);
            o1p[e] = f2bf(rbf(x1 * bf2f(c1p[e])) + rbf(-x2 * bf2f(s1p[e])));
            o2p[e] = f2bf(rbf(x2 * bf2f(c2p[e])) + rbf(x1 * bf2f(s2p[e])));
        }
        if (j < q_per_kv) {
            bf16_t* dst = q_out + ((size_t)t * n_head + g * q_per_kv + j) * HS;
            *reinterpret_cast<uint4*>(dst + c * 8) = o1;
            *reinterpret_cast<uint4*>(dst + HALF + c * 8) = o2;
        } else {   // K cache in MFMA-fragment order: 8 aligned channels stay one 16-B store
            bf16_t* kb = k_cache + ((size_t)tok_slot[t] * n_groups + g) * s_max * HS;
            *reinterpret_cast<uint4*>(kb + kfrag_off<HS>(pos, c * 8)) = o1;
            *reinterpret_cast<uint4*>(kb + kfrag_off<HS>(pos, HALF + c * 8)) = o2;
            if (k_out != nullptr) {   // plain [n_tok, g, hs] copy for the training backward
                bf16_t* ko = k_out + ((size_t)t * n_groups + g) * HS;
                *reinterpret_cast<uint4*>(ko + c * 8) = o1;
                *reinterpret_cast<uint4*>(ko + HALF + c * 8) = o2;
            }
        }
    }

    // phase 2: v -> V^T cache through an LDS transpose (adjacent lanes = adjacent tokens)
    for (int it = threadIdx.x; it < 64 * (HS / 8); it += 256) {
        const int c = it % (HS / 8), tl = it / (HS / 8);
        const int t = t0 + tl;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (t < n_tok) u = *reinterpret_cast<const uint4*>(qkv + (size_t)t * row_elems + grp_off + (q_per_kv + 1) * HS + c * 8);
        *reinterpret_cast<uint4*>(&vt[tl][c * 8]) = u;
        if (v_out != nullptr && t < n_tok) *reinterpret_cast<uint4*>(v_out + ((size_t)t * n_groups + g) * HS + c * 8) = u;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < 64 * HS; it += 256) {
        const int tl = it & 63, dd = it >> 6;
        const int t = t0 + tl;
        if (t < n_tok) {
            vT_cache[((size_t)tok_slot[t] * n_groups + g) * HS * s_max + vfrag_off<HS>(tok_pos[t], dd)] = vt[tl][dd];
        }
    }
}

extern "C" int dh_qkv_rope_cache_bf16(const dh_bf16* qkv, const dh_bf16* cos, const dh_bf16* sin,
                                      const int32_t* tok_slot, const int32_t* tok_pos, dh_bf16* q_out,
                                      dh_bf16* k_cache, dh_bf16* vT_cache, dh_bf16* k_out, dh_bf16* v_out, int n_tok,
                                      int n_head, int n_groups, int hs, int s_max, void* stream) {
    DH_CHECK(n_tok >= 0 && n_groups > 0 && n_head % n_groups == 0, "dh_qkv_rope_cache_bf16: bad head counts");
    DH_CHECK(hs == 64 || hs == 128, "dh_qkv_rope_cache_bf16: head_size %d unsupported (64 or 128)", hs);
    if (n_tok == 0) return 0;
    dim3 grid(cdiv(n_tok, 64), n_groups), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (hs == 64)
        hipLaunchKernelGGL((qkv_rope_cache_kernel<64>), grid, block, 0, s, qkv, cos, sin, tok_slot, tok_pos, q_out,
                           k_cache, vT_cache, k_out, v_out, n_tok, n_head, n_groups, s_max);
    else
        hipLaunchKernelGGL((qkv_rope_cache_kernel<128>), grid, block, 0, s, qkv, cos, sin, tok_slot, tok_pos, q_out,
                           k_cache, vT_cache, k_out, v_out, n_tok, n_head, n_groups, s_max);
    DH_LAUNCH_CHECK();
    return 0;
}
