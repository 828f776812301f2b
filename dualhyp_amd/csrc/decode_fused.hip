// Fused kernels of the single-token decode step.  A decode layer is 7 launches:
//   partial GEMM [Wqkv;A]  ->  attn_decode_fused  ->  partial GEMM [Wproj;A]  ->  finish_norm
//   ->  SwiGLU GEMM  ->  partial GEMM Wmlp  ->  finish_norm (next layer's norm)
// The weight-streaming GEMMs (gemm_skinny.hip) emit fp32 partial sums; the LoRA update, the
// residual add, the rounding to bf16 and the RMSNorm are finished HERE, by the consumer, with
// exactly the reference's rounding points (ger/lora.py:159-166,388-402; ger/model.py:185-186,
// 216-259; ger/rmsnorm.py:17-21).  That removes the separate x·A^T launch and its dependent
// latency from the chain, lets small matrices be split over K across blocks, and folds
// rope + KV-cache append + split-KV attention + combine into one kernel.
#include "common.h"

namespace {

constexpr int DCOLS = 16;   // padded head columns of the attention partials
constexpr int MAXP = 16;    // most K-slices a partial-sum GEMM emits

// --------------------------------------------------------------------------- attention (decode)
// grid (n_seq * n_groups), NW waves (hs 64: 16, hs 128: 8).  qkv32: [n_part][n_seq][ldq] fp32, ldq = qkv_dim + n_ext;
// columns [qkv_dim, qkv_dim+48) hold x·A^T of the q/k/v LoRA (when lora_b != null).
// NW waves share the key tiles of one (sequence, group).  Round 3: SIXTEEN waves at hs 64 — at the benchmark's ~544 cached keys
// (17 tiles) every tile is then requested before the LoRA / rope prologue and no wave walks a second or third tile behind an
// exposed HBM round trip (with 8 waves the loop issued a tile's loads only after computing the previous one; the 640-row step
// spent 85 us per layer here, 4.5 TB/s).  The tile -> wave deal and the combine order are a property of the head size, never
// of the row count, so batch invariance is untouched; hs 128 keeps 8 waves (its tile and accumulators need 256 VGPRs).
template <int HS, int PMAX, int NW>
__global__ __launch_bounds__(NW * 64, NW == 16 ? 1 : (NW == 6 ? 3 : (HS == 64 ? 4 : 2))) void attn_decode_fused_kernel(
    const float* __restrict__ qkv32, int n_part, int pairs, int n_seq, int ldq, int qkv_dim,
    const bf16_t* __restrict__ lora_b, float lora_scale, int split0, int split1,
    const bf16_t* __restrict__ cos, const bf16_t* __restrict__ sin, const int32_t* __restrict__ seq_slot,
    const int32_t* __restrict__ kv_len, bf16_t* __restrict__ k_cache, bf16_t* __restrict__ vT_cache,
    bf16_t* __restrict__ y, int n_head, int n_groups, int s_max, float scale) {
    constexpr int KS = HS / 16, DT = HS / 32, HALF = HS / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pair = blockIdx.x, seq = pair / n_groups, g = pair % n_groups;
    const int q_per_kv = n_head / n_groups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int slot = seq_slot[seq];
    const int len = kv_len[seq], pos = len - 1;          // the new token sits at position len-1
    // LDS carve (all offsets multiples of 16 B)
    bf16_t* sQ = reinterpret_cast<bf16_t*>(smem);                       // [16][HS] rotated queries
    float* sKn = reinterpret_cast<float*>(smem + 16 * HS * 2);          // [HS] new key (bf16 values)
    float* sVn = sKn + HS;                                              // [HS] new value
    float* sXa = sVn + HS;                                              // [48] bf16(x·A^T)
    float* sSn = sXa + 48;                                              // [16] score of the new key
    float* sPm = sSn + 16;                                              // [NW][DCOLS] running max
    float* sPl = sPm + NW * DCOLS;                                      // [NW][DCOLS] running sum
    float* sPo = sPl + NW * DCOLS;                                      // [NW][HS][DCOLS] partial O^T
    constexpr int NT_ = NW * 64;                                        // threads

    // ---- request the K / V^T operands of this wave's first PF tiles before anything else: they do
    // not depend on the new token, and their HBM latency hides under the LoRA/rope phase
    // tiles of operands in flight per wave.  8 waves: one (128 VGPRs, two blocks per CU at hs 64); SIX waves (round 3, hs 64): two —
    // three waves per SIMD leave 168 VGPRs, and at ~544 keys (17 tiles = 3 per wave) the third tile's loads are issued right
    // after the first tile is consumed and land under the second: no tile waits for an exposed HBM round trip
    constexpr int PF = NW == 6 ? 2 : 1;
    struct VF { bf16x8 v; };
    struct Tile { bf16x8 kf[KS]; VF vf[DT][2]; };
    Tile tl[PF];
    const bf16_t* kbase = k_cache + ((size_t)slot * n_groups + g) * s_max * HS;
    const bf16_t* vbase = vT_cache + ((size_t)slot * n_groups + g) * HS * s_max;
    const int n_tiles = (pos + 31) / 32;
    // the caches are in MFMA-fragment order (common.h): a tile is 8 coalesced 1-KiB loads
    auto load_tile = [&](Tile& T, int t) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            T.kf[ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kbase + kfrag_blk<HS>(t, ks) + lane * 8));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                T.vf[dt][s2].v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vbase + vfrag_blk<HS>(t, dt, s2) + lane * 8));
    };
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (wave + NW * p < n_tiles) load_tile(tl[p], wave + NW * p);

    const float* row0 = qkv32 + (size_t)seq * ldq;
    const size_t pstride = (size_t)n_seq * ldq;
    // all partial loads are issued before the first add (a runtime-trip-count loop would wait
    // for one L2 round trip per partial)
    auto psum = [&](int c) {
        float v[PMAX];
#pragma unroll
        for (int p = 0; p < PMAX; ++p) v[p] = p < n_part ? row0[p * pstride + c] : 0.f;
        // the decode family's summation order (common.h, "K-slice combine"): leaves are added in adjacent PAIRS, the
        // pair sums in index order; a producer that owns two slices per block has already formed the pairs (pairs == 0)
        float s = 0.f;
        if (pairs) {
#pragma unroll
            for (int p = 0; p < PMAX; p += 2) s += v[p] + v[p + 1];
        } else {
#pragma unroll
            for (int p = 0; p < PMAX; ++p) s += v[p];
        }
        return s;
    };
    if (lora_b != nullptr && tid < 48) sXa[tid] = rbf(psum(qkv_dim + tid));
    for (int i = tid; i < 16 * HS; i += NT_) sQ[i] = 0;               // zero padding rows of Q
    __syncthreads();

    // bf16 value of fused-qkv column c: bf16(bf16(x·W^T) + bf16(bf16(xa·B^T)*s))
    auto finish = [&](int c) -> float {
        float o = rbf(psum(c));
        if (lora_b != nullptr) {
            const int seg = (c >= split0) + (c >= split1);
            const uint4* b4 = reinterpret_cast<const uint4*>(lora_b + (size_t)c * 16);
            float l = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 bv = b4[h];
                const bf16_t* bp = reinterpret_cast<const bf16_t*>(&bv);
#pragma unroll
                for (int e = 0; e < 8; ++e) l = fmaf(sXa[seg * 16 + h * 8 + e], bf2f(bp[e]), l);
            }
            o = rbf(o + rbf(rbf(l) * lora_scale));
        }
        return o;
    };
    const int gbase = g * (q_per_kv + 2) * HS;
    bf16_t* kdst = k_cache + ((size_t)slot * n_groups + g) * s_max * HS;
    bf16_t* vdst = vT_cache + ((size_t)slot * n_groups + g) * HS * s_max;
    const int n_rope = (q_per_kv + 1) * HALF;
    for (int it = tid; it < n_rope + HS; it += NT_) {
        if (it < n_rope) {
            const int j = it / HALF, i = it % HALF;
            const float x1 = finish(gbase + j * HS + i), x2 = finish(gbase + j * HS + HALF + i);
            const bf16_t* cp = cos + (size_t)pos * HS;
            const bf16_t* sp = sin + (size_t)pos * HS;
            const bf16_t o1 = f2bf(rbf(x1 * bf2f(cp[i])) + rbf(-x2 * bf2f(sp[i])));
            const bf16_t o2 = f2bf(rbf(x2 * bf2f(cp[HALF + i])) + rbf(x1 * bf2f(sp[HALF + i])));
            if (j < q_per_kv) {
                sQ[j * HS + i] = o1;
                sQ[j * HS + HALF + i] = o2;
            } else {
                sKn[i] = bf2f(o1);
                sKn[HALF + i] = bf2f(o2);
                kdst[kfrag_off<HS>(pos, i)] = o1;
                kdst[kfrag_off<HS>(pos, HALF + i)] = o2;
            }
        } else {
            const int e = it - n_rope;
            const bf16_t v = f2bf(finish(gbase + (q_per_kv + 1) * HS + e));
            sVn[e] = bf2f(v);
            vdst[vfrag_off<HS>(pos, e)] = v;
        }
    }
    __syncthreads();

    // score of the new key against each head (it is merged at the combine, so nobody has to
    // read this block's own cache write back)
    for (int h = wave; h < q_per_kv; h += NW) {
        float p = 0.f;
        for (int e = lane; e < HS; e += 64) p += bf2f(sQ[h * HS + e]) * sKn[e];
        p = wave_sum(p);
        if (lane == 0) sSn[h] = p * scale;
    }

    // ---- keys 0 .. pos-1 straight from the cache, 32-key tiles dealt over the 8 waves; the
    // operands of two tiles per wave were requested before the LoRA/rope phase (tl[] above)
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(sQ + (lr & 15) * HS + ks * 16 + lh * 8);
    if (lr >= 16) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    auto compute_tile = [&](const Tile& T, int t) {
        const int key0 = t * 32;
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T.kf[ks], qf[ks], st, 0, 0, 0);
        float m_t = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key_abs = key0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float sc = st[r] * scale;
            sc = key_abs < pos ? sc : -INFINITY;
            st[r] = sc;
            m_t = fmaxf(m_t, sc);
        }
        m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
        const float m_new = fmaxf(m_run, m_t);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float psm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(st[r] - m_new);
            st[r] = p;
            psm += p;
        }
        l_run = l_run * alpha + psm;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.u[j] = pack2bf(st[8 * s2 + 2 * j], st[8 * s2 + 2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (s2 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                }
                // keys >= pos carry p == 0 and the cache beyond the written prefix is finite
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T.vf[dt][s2].v, pf.v, o[dt], 0, 0, 0);
            }
        }
    };
    for (int base = wave; base < n_tiles; base += NW * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (base + NW * p < n_tiles) compute_tile(tl[p], base + NW * p);
            if (base + NW * (PF + p) < n_tiles) load_tile(tl[p], base + NW * (PF + p));    // its set is free again
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (lr < q_per_kv) {
        if (lh == 0) {
            sPm[wave * DCOLS + lr] = m_run;
            sPl[wave * DCOLS + lr] = l_tot;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                sPo[(wave * HS + d) * DCOLS + lr] = o[dt][r];
            }
    }
    __syncthreads();
    // ---- combine the NW wave partials and the new key
    for (int it = tid; it < q_per_kv * HS; it += NT_) {
        const int h = it / HS, d = it % HS;
        const float sn = sSn[h];
        float M = sn;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, sPm[w * DCOLS + h]);
        const float pn = __expf(sn - M);
        float L = pn, O = rbf(pn) * sVn[d];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float mw = sPm[w * DCOLS + h];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            L += sPl[w * DCOLS + h] * f;
            O += sPo[(w * HS + d) * DCOLS + h] * f;
        }
        y[(size_t)seq * n_head * HS + (g * q_per_kv + h) * HS + d] = f2bf(O / L);
    }
}

// Round 3, one more shape measured and not kept (bit-identical, 82 tests green): every tile after a wave's first streamed through an 8-KiB LDS slot
// private to the wave, filled by LDS-DMA one tile ahead with the second tile's request issued at kernel start (the wave's partial-O block
// aliased into the slot: 69 KiB per block, two blocks per CU as before) — 4.71-4.73 ms per 640-row decode step against 4.58-4.64: the second
// tile's exposed round trip is not what the kernel waits for.
#ifndef DH_ATTN_WAVES64
#define DH_ATTN_WAVES64 8     // A/B builds: 6 (two tiles in flight per wave) and 16 (one block per CU: measured 5.68 vs 4.74 ms per 640-row step)
#endif
template <int HS>
constexpr int attn_fused_waves() { return HS == 64 ? DH_ATTN_WAVES64 : 8; }
template <int HS>
constexpr size_t attn_fused_lds() {
    constexpr int NW = attn_fused_waves<HS>();
    return 16 * HS * 2 + (HS + HS + 48 + 16 + NW * DCOLS + NW * DCOLS + NW * HS * DCOLS) * sizeof(float);
}

// --------------------------------------------------------------------------- finish + norm
// grid (rows), 256 threads.  h32: [n_part][rows][ldh] fp32 partials of x·[W;A]^T (columns
// [d, d+16) = x·A^T when lora_b != null).  Per row:
//   h  = bf16( bf16(sum_p h32) + bf16( bf16(xa·B^T) * s ) )      (LoRA finish, ger/lora.py:159-166)
//   x' = bf16( x + h )                                           (residual, ger/model.py:185-186)
//   xn = RMSNorm(x') with weight w_norm                          (ger/rmsnorm.py:17-21, Q11 flag)
template <int MAXC>
__global__ __launch_bounds__(256) void finish_norm_kernel(const float* __restrict__ h32, int n_part, int pairs, int rows, int ldh,
                                                          const bf16_t* __restrict__ lora_b, float lora_scale,
                                                          const bf16_t* __restrict__ x_resid,
                                                          const bf16_t* __restrict__ w_norm, bf16_t* __restrict__ x_out,
                                                          bf16_t* __restrict__ xn_out, int d, float eps,
                                                          const uint8_t* __restrict__ row_tail) {
    __shared__ float sXa[16];
    __shared__ float sRed[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* hrow = h32 + (size_t)row * ldh;
    const size_t pstride = (size_t)rows * ldh;
    if (lora_b != nullptr && tid < 16) {
        float pv[MAXP];
#pragma unroll
        for (int p = 0; p < MAXP; ++p) pv[p] = p < n_part ? hrow[p * pstride + d + tid] : 0.f;
        float s = 0.f;
        if (pairs) {
#pragma unroll
            for (int p = 0; p < MAXP; p += 2) s += pv[p] + pv[p + 1];
        } else {
#pragma unroll
            for (int p = 0; p < MAXP; ++p) s += pv[p];
        }
        sXa[tid] = rbf(s);
    }
    __syncthreads();
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c0 = (tid + i * 256) * 8;
        if (c0 < d) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            // partials in groups of 4: 8 independent 16-B loads in flight before the adds
            for (int p0 = 0; p0 < n_part; p0 += 4) {
                float4 a0[4], a1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = p0 + q < n_part ? p0 + q : p0;
                    a0[q] = *reinterpret_cast<const float4*>(hrow + p * pstride + c0);
                    a1[q] = *reinterpret_cast<const float4*>(hrow + p * pstride + c0 + 4);
                }
                if (pairs) {      // leaves: adjacent pairs first (a missing partner counts as 0), pair sums in order
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        if (p0 + q < n_part) {
                            const bool two = p0 + q + 1 < n_part;
                            const float4 b0 = two ? a0[q + 1] : float4{0.f, 0.f, 0.f, 0.f}, b1 = two ? a1[q + 1] : float4{0.f, 0.f, 0.f, 0.f};
                            acc[0] += a0[q].x + b0.x; acc[1] += a0[q].y + b0.y; acc[2] += a0[q].z + b0.z; acc[3] += a0[q].w + b0.w;
                            acc[4] += a1[q].x + b1.x; acc[5] += a1[q].y + b1.y; acc[6] += a1[q].z + b1.z; acc[7] += a1[q].w + b1.w;
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (p0 + q < n_part) {
                        acc[0] += a0[q].x; acc[1] += a0[q].y; acc[2] += a0[q].z; acc[3] += a0[q].w;
                        acc[4] += a1[q].x; acc[5] += a1[q].y; acc[6] += a1[q].z; acc[7] += a1[q].w;
                    }
                }
                }
            }
            const uint4 xr = *reinterpret_cast<const uint4*>(x_resid + (size_t)row * d + c0);
            const bf16_t* xp = reinterpret_cast<const bf16_t*>(&xr);
            uint4 xo;
            bf16_t* xop = reinterpret_cast<bf16_t*>(&xo);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = rbf(acc[e]);
                if (lora_b != nullptr) {
                    const uint4* b4 = reinterpret_cast<const uint4*>(lora_b + (size_t)(c0 + e) * 16);
                    float l = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint4 bv = b4[h];
                        const bf16_t* bp = reinterpret_cast<const bf16_t*>(&bv);
#pragma unroll
                        for (int k = 0; k < 8; ++k) l = fmaf(sXa[h * 8 + k], bf2f(bp[k]), l);
                    }
                    o = rbf(o + rbf(rbf(l) * lora_scale));
                }
                xop[e] = f2bf(bf2f(xp[e]) + o);
                v[i][e] = bf2f(xop[e]);
                ss += rbf(v[i][e] * v[i][e]);
            }
            *reinterpret_cast<uint4*>(x_out + (size_t)row * d + c0) = xo;
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) sRed[wave] = ss;
    __syncthreads();
    ss = sRed[0] + sRed[1] + sRed[2] + sRed[3];
    const float ms = rbf(ss / (float)d);
    const float t = rbf(ms + eps);
    const bool tail = row_tail != nullptr && row_tail[row] != 0;
    const float r = tail ? rbf(1.0f / rbf(sqrtf(t))) : rbf(1.0f / sqrtf(t));
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c0 = (tid + i * 256) * 8;
        if (c0 < d) {
            const uint4 wu = *reinterpret_cast<const uint4*>(w_norm + c0);
            const bf16_t* wp = reinterpret_cast<const bf16_t*>(&wu);
            uint4 o;
            bf16_t* op = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
            for (int e = 0; e < 8; ++e) op[e] = f2bf(bf2f(wp[e]) * rbf(v[i][e] * r));
            *reinterpret_cast<uint4*>(xn_out + (size_t)row * d + c0) = o;
        }
    }
}

}  // namespace

extern "C" int dh_attn_decode_fused_bf16(const float* qkv32, int n_part, int pairs, int n_seq, int qkv_dim, int n_ext,
                                         const dh_bf16* lora_b, float lora_scale, int split0, int split1,
                                         const dh_bf16* cos, const dh_bf16* sin, const int32_t* seq_slot,
                                         const int32_t* kv_len, dh_bf16* k_cache, dh_bf16* vT_cache, dh_bf16* y,
                                         int n_head, int n_groups, int hs, int s_max, void* stream) {
    DH_CHECK(qkv32 && cos && sin && seq_slot && kv_len && k_cache && vT_cache && y, "dh_attn_decode_fused_bf16: null argument");
    DH_CHECK(n_groups > 0 && n_head % n_groups == 0 && n_head / n_groups <= DCOLS, "dh_attn_decode_fused_bf16: bad head counts");
    DH_CHECK(hs == 64 || hs == 128, "dh_attn_decode_fused_bf16: head_size %d unsupported", hs);
    DH_CHECK(s_max % 64 == 0 && n_part >= 1 && n_part <= MAXP, "dh_attn_decode_fused_bf16: bad s_max / n_part");
    DH_CHECK(qkv_dim == (n_head + 2 * n_groups) * hs, "dh_attn_decode_fused_bf16: qkv_dim mismatch");
    DH_CHECK(lora_b == nullptr || n_ext >= 48, "dh_attn_decode_fused_bf16: LoRA needs the 48 x·A^T columns");
    if (n_seq <= 0) return 0;
    const float scale = 1.0f / sqrtf((float)hs);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(n_seq * n_groups);
#define ATT_LAUNCH(HSV, PM)                                                                                           \
    hipLaunchKernelGGL((attn_decode_fused_kernel<HSV, PM, attn_fused_waves<HSV>()>), grid, dim3(64 * attn_fused_waves<HSV>()), attn_fused_lds<HSV>(), s, qkv32, n_part, pairs, n_seq,  \
                       qkv_dim + n_ext, qkv_dim, lora_b, lora_scale, split0, split1, cos, sin, seq_slot, kv_len, k_cache,  \
                       vT_cache, y, n_head, n_groups, s_max, scale)
    if (hs == 64) {
        DH_MAX_LDS_ONCE((attn_decode_fused_kernel<64, 2, attn_fused_waves<64>()>), attn_fused_lds<64>());
        DH_MAX_LDS_ONCE((attn_decode_fused_kernel<64, 8, attn_fused_waves<64>()>), attn_fused_lds<64>());
        DH_MAX_LDS_ONCE((attn_decode_fused_kernel<64, 16, attn_fused_waves<64>()>), attn_fused_lds<64>());
        if (n_part <= 2) ATT_LAUNCH(64, 2);
        else if (n_part <= 8) ATT_LAUNCH(64, 8);
        else ATT_LAUNCH(64, 16);
    } else {
        DH_MAX_LDS_ONCE((attn_decode_fused_kernel<128, 2, 8>), attn_fused_lds<128>());
        DH_MAX_LDS_ONCE((attn_decode_fused_kernel<128, 8, 8>), attn_fused_lds<128>());
        DH_MAX_LDS_ONCE((attn_decode_fused_kernel<128, 16, 8>), attn_fused_lds<128>());
        if (n_part <= 2) ATT_LAUNCH(128, 2);
        else if (n_part <= 8) ATT_LAUNCH(128, 8);
        else ATT_LAUNCH(128, 16);
    }
#undef ATT_LAUNCH
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_finish_norm_bf16(const float* h32, int n_part, int pairs, int rows, int d, int n_ext, const dh_bf16* lora_b,
                                   float lora_scale, const dh_bf16* x_resid, const dh_bf16* w_norm, dh_bf16* x_out,
                                   dh_bf16* xn_out, float eps, const uint8_t* row_tail, void* stream) {
    DH_CHECK(h32 && x_resid && w_norm && x_out && xn_out, "dh_finish_norm_bf16: null argument");
    DH_CHECK(d % 8 == 0 && d <= 8192 && n_part >= 1 && n_part <= MAXP, "dh_finish_norm_bf16: unsupported d=%d", d);
    DH_CHECK(lora_b == nullptr || n_ext >= 16, "dh_finish_norm_bf16: LoRA needs the 16 x·A^T columns");
    DH_CHECK((d + n_ext) % 4 == 0, "dh_finish_norm_bf16: row stride must be a multiple of 4 floats");
    if (rows <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(rows), block(256);
    const int ldh = d + n_ext;
#define LAUNCH(MAXC) hipLaunchKernelGGL((finish_norm_kernel<MAXC>), grid, block, 0, s, h32, n_part, pairs, rows, ldh, lora_b, \
                                        lora_scale, x_resid, w_norm, x_out, xn_out, d, eps, row_tail)
    if (d <= 2048) { LAUNCH(1); }
    else if (d <= 4096) { LAUNCH(2); }
    else { LAUNCH(4); }
#undef LAUNCH
    DH_LAUNCH_CHECK();
    return 0;
}
