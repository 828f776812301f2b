// Backward-pass kernels of the LoRA fine-tune (finetune/ger.py:278-292): everything except the dX
// GEMMs (those reuse gemm.hip / gemm256.hip on transposed copies of the frozen weights) and the
// attention backward (attention_bwd.hip).  Activations and activation-gradients are bf16, sums are
// fp32, LoRA gradients are accumulated in fp32 buffers.
#include "common.h"
#include <climits>

int g_tn_mfma = 1;    // dh_set_tuning(26, 0): the LoRA-gradient contraction on the VALU kernel of rounds 2-3 (A/B)

namespace {

// ---------------------------------------------------------------------------------- SwiGLU backward
// act = bf16(silu(g)) * u  (ger/model.py:313-315)  ->  dg = dact * u * silu'(g), du = dact * silu(g)
// out: dgu [rows, 2*I] = [dg | du]  (feeds ONE dX GEMM against [fc_1 ; fc_2] stacked along K)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ g,
                                                         const bf16_t* __restrict__ u, bf16_t* __restrict__ dgu,
                                                         size_t rows, int I) {
    const size_t n8 = rows * (size_t)(I / 8);
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < n8; c += (size_t)gridDim.x * blockDim.x) {
        const size_t row = c / (I / 8);
        const int col = (int)(c % (I / 8)) * 8;
        const uint4 da = *reinterpret_cast<const uint4*>(dact + row * I + col);
        const uint4 gv = *reinterpret_cast<const uint4*>(g + row * I + col);
        const uint4 uv = *reinterpret_cast<const uint4*>(u + row * I + col);
        const bf16_t *dp = (const bf16_t*)&da, *gp = (const bf16_t*)&gv, *up = (const bf16_t*)&uv;
        uint4 og, ou;
        bf16_t *ogp = (bf16_t*)&og, *oup = (bf16_t*)&ou;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = bf2f(gp[e]), d = bf2f(dp[e]);
            const float sig = 1.0f / (1.0f + expf(-gg));
            const float s = rbf(gg * sig);                      // silu(g) as the forward rounded it
            ogp[e] = f2bf(d * bf2f(up[e]) * (sig * (1.0f + gg * (1.0f - sig))));
            oup[e] = f2bf(d * s);
        }
        *reinterpret_cast<uint4*>(dgu + row * 2 * I + col) = og;
        *reinterpret_cast<uint4*>(dgu + row * 2 * I + I + col) = ou;
    }
}

// act = bf16( bf16(silu(g)) * u ) from separately stored g, u (the training forward keeps both)
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ u,
                                                         bf16_t* __restrict__ act, size_t n8) {
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < n8; c += (size_t)gridDim.x * blockDim.x) {
        const uint4 gv = reinterpret_cast<const uint4*>(g)[c], uv = reinterpret_cast<const uint4*>(u)[c];
        const bf16_t *gp = (const bf16_t*)&gv, *up = (const bf16_t*)&uv;
        uint4 o;
        bf16_t* op = (bf16_t*)&o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = bf2f(gp[e]);
            op[e] = f2bf(rbf(silu_fast(gg)) * bf2f(up[e]));
        }
        reinterpret_cast<uint4*>(act)[c] = o;
    }
}

// ---------------------------------------------------------------------------------- RMSNorm backward
// y = w * x * r, r = rsqrt(mean(x^2)+eps):  dx = r * (w*dy) - x * r^3 * mean(x * w*dy)   (+ dres)
// one wave per row, row in registers.  dx_out = bf16(dx + dres) when dres != null.
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const bf16_t* __restrict__ dres,
                                                          bf16_t* __restrict__ dx, int rows, int d, float eps) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nchunk = d >> 3;
    float xv[MAXC][8], gv[MAXC][8];
    float ss = 0.f, sg = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            const uint4 xu = reinterpret_cast<const uint4*>(x + (size_t)row * d)[c];
            const uint4 du = reinterpret_cast<const uint4*>(dy + (size_t)row * d)[c];
            const uint4 wu = reinterpret_cast<const uint4*>(w)[c];
            const bf16_t *xp = (const bf16_t*)&xu, *dp = (const bf16_t*)&du, *wp = (const bf16_t*)&wu;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xv[i][e] = bf2f(xp[e]);
                gv[i][e] = bf2f(dp[e]) * bf2f(wp[e]);
                ss += xv[i][e] * xv[i][e];
                sg += xv[i][e] * gv[i][e];
            }
        }
    }
    ss = wave_sum(ss);
    sg = wave_sum(sg);
    const float r = 1.0f / sqrtf(ss / (float)d + eps);
    const float k = r * r * r * sg / (float)d;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            uint4 o;
            bf16_t* op = (bf16_t*)&o;
            uint4 ru = make_uint4(0, 0, 0, 0);
            if (dres) ru = reinterpret_cast<const uint4*>(dres + (size_t)row * d)[c];
            const bf16_t* rp = (const bf16_t*)&ru;
#pragma unroll
            for (int e = 0; e < 8; ++e) op[e] = f2bf(r * gv[i][e] - k * xv[i][e] + (dres ? bf2f(rp[e]) : 0.f));
            reinterpret_cast<uint4*>(dx + (size_t)row * d)[c] = o;
        }
    }
}

// ---------------------------------------------------------------------------------- rope backward
// forward: o1 = x1*c1 - x2*s1 ; o2 = x2*c2 + x1*s2  (ger/model.py:349-355)
// backward: dx1 = do1*c1 + do2*s2 ; dx2 = do2*c2 - do1*s1.   v passes through.  Gathers the three
// gradient tensors back into the group-interleaved fused-qkv layout.
template <int HS>
__global__ __launch_bounds__(256) void qkv_rope_bwd_kernel(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk,
                                                           const bf16_t* __restrict__ dv, const bf16_t* __restrict__ cos,
                                                           const bf16_t* __restrict__ sin, const int32_t* __restrict__ tok_pos,
                                                           bf16_t* __restrict__ dqkv, int n_tok, int n_head, int n_groups) {
    // one thread = 8 channels of the first half of a head and the matching 8 of the second half: 16-byte accesses throughout
    // (round 4; the element-per-thread form moved 2 bytes per lane: 129 us per call on the packed micro-step, 1.4 TB/s)
    constexpr int HALF = HS / 2, C8 = HALF / 8;
    const int q_per_kv = n_head / n_groups;
    const int heads = q_per_kv + 2;
    const int row_elems = n_groups * heads * HS;
    const size_t total = (size_t)n_tok * n_groups * heads * C8;
    for (size_t it = blockIdx.x * (size_t)blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(it % C8) * 8;
        const int j = (int)((it / C8) % heads);
        const int g = (int)((it / ((size_t)C8 * heads)) % n_groups);
        const int t = (int)(it / ((size_t)C8 * heads * n_groups));
        bf16_t* dst = dqkv + (size_t)t * row_elems + (g * heads + j) * HS;
        if (j == q_per_kv + 1) {                       // v: straight copy
            const bf16_t* s = dv + ((size_t)t * n_groups + g) * HS;
            *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(s + i);
            *reinterpret_cast<uint4*>(dst + HALF + i) = *reinterpret_cast<const uint4*>(s + HALF + i);
            continue;
        }
        const bf16_t* s = j < q_per_kv ? dq + ((size_t)t * n_head + g * q_per_kv + j) * HS : dk + ((size_t)t * n_groups + g) * HS;
        const int pos = tok_pos[t];
        const uint4 c1v = *reinterpret_cast<const uint4*>(cos + (size_t)pos * HS + i), c2v = *reinterpret_cast<const uint4*>(cos + (size_t)pos * HS + HALF + i);
        const uint4 s1v = *reinterpret_cast<const uint4*>(sin + (size_t)pos * HS + i), s2v = *reinterpret_cast<const uint4*>(sin + (size_t)pos * HS + HALF + i);
        const uint4 d1v = *reinterpret_cast<const uint4*>(s + i), d2v = *reinterpret_cast<const uint4*>(s + HALF + i);
        const bf16_t *c1 = (const bf16_t*)&c1v, *c2 = (const bf16_t*)&c2v, *s1 = (const bf16_t*)&s1v, *s2 = (const bf16_t*)&s2v;
        const bf16_t *d1 = (const bf16_t*)&d1v, *d2 = (const bf16_t*)&d2v;
        uint4 o1, o2;
        bf16_t *o1p = (bf16_t*)&o1, *o2p = (bf16_t*)&o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = bf2f(d1[e]), b = bf2f(d2[e]);
            o1p[e] = f2bf(a * bf2f(c1[e]) + b * bf2f(s2[e]));
            o2p[e] = f2bf(b * bf2f(c2[e]) - a * bf2f(s1[e]));
        }
        *reinterpret_cast<uint4*>(dst + i) = o1;
        *reinterpret_cast<uint4*>(dst + HALF + i) = o2;
    }
}

// ---------------------------------------------------------------------------------- LoRA gradients
// out[m][n] (+)= scale * sum_t a[t][m] * b[t][n]      (contraction over tokens)
//   a: [T, lda] bf16 (columns m0..), b: [T, ldb] bf16.  One 16 x 16 output tile per block of 256 threads:
//   thread (m, n) strides over t in 4 phases ... kept simple: T <= a few thousand, M*N <= 2560*16.
__global__ __launch_bounds__(256) void tn_accum_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b,
                                                       int ldb, float* __restrict__ out, int ldo, int T, int M, int N,
                                                       float scale, int accumulate) {
    // 256 threads = 16 (m) x 16 (n); every thread walks all tokens (a coalesces across m, b broadcasts)
    const int m = blockIdx.x * 16 + (threadIdx.x & 15), n = blockIdx.y * 16 + (threadIdx.x >> 4);
    float acc = 0.f;
    if (m < M && n < N)
        for (int t = 0; t < T; ++t) acc = fmaf(bf2f(a[(size_t)t * lda + m]), bf2f(b[(size_t)t * ldb + n]), acc);
    if (m < M && n < N) {
        float* o = out + (size_t)m * ldo + n;
        *o = (accumulate ? *o : 0.f) + scale * acc;
    }
}

// The LoRA shapes have one LARGE output dimension (d, or the fused qkv width) and one small one (the
// rank, padded to 16 or 3 x 16): lanes run along the large one (coalesced loads of its operand, coalesced
// stores), 16 values of the small one sit in registers (its operand row is a 32-byte broadcast), the four
// waves of a block deal the tokens among them and meet in LDS.  LARGE_IS_M: out[m = large][n = small].
// PACKED micro-batches (round 3: P x 560 tokens per launch) make the token loop the long dimension: the grid's z splits the
// tokens into chunks of `tchunk`; with gridDim.z > 1 a block writes its chunk's plain fp32 sums to part[z][M][N] and
// tn_reduce_kernel adds the chunks in index order (deterministic, no atomics), applies `scale` and accumulates.
// SB: 16-wide groups of the SMALL dimension a block owns (1, or 3 for the 48-row QKV LoRA-A gradient: the large operand — 2048
// columns of the packed tokens — is then read once instead of three times; the sums are unchanged: per output the same 16 wave
// partials in the same order).
template <bool LARGE_IS_M, int SB = 1>
__global__ __launch_bounds__(1024) void tn_accum_wide_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b,
                                                             int ldb, float* __restrict__ out, int ldo, int T, int M, int N,
                                                             float scale, int accumulate, int tchunk) {
    if (gridDim.z > 1) {
        const int t_begin = blockIdx.z * tchunk;
        a += (size_t)t_begin * lda;
        b += (size_t)t_begin * ldb;
        T = min(T - t_begin, tchunk);
        out += (size_t)blockIdx.z * M * N;      // part[z], dense [M][N]
        ldo = N; scale = 1.f; accumulate = 0;
    }
    constexpr int NWV = 16, UN = 4;      // waves per block (tokens dealt over them), tokens in flight per wave
    __shared__ float red[NWV][16][64];   // 64 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* big = LARGE_IS_M ? a : b;
    const bf16_t* small = LARGE_IS_M ? b : a;
    const int ldbig = LARGE_IS_M ? lda : ldb, ldsm = LARGE_IS_M ? ldb : lda;
    const int L = LARGE_IS_M ? M : N;
    const int l = blockIdx.x * 64 + lane, s0 = blockIdx.y * 16 * SB;
    const int lc = l < L ? l : L - 1;
    float acc[SB][16];
#pragma unroll
    for (int sb = 0; sb < SB; ++sb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[sb][j] = 0.f;
    for (int t0 = wave; t0 < T; t0 += NWV * UN) {
        bf16_t xv[UN];
        uint4 lo[UN][SB], hi[UN][SB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {           // all loads of the round first: the loop is latency-bound otherwise
            const int t = min(t0 + u * NWV, T - 1);
            xv[u] = big[(size_t)t * ldbig + lc];
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                lo[u][sb] = *reinterpret_cast<const uint4*>(small + (size_t)t * ldsm + s0 + 16 * sb);
                hi[u][sb] = *reinterpret_cast<const uint4*>(small + (size_t)t * ldsm + s0 + 16 * sb + 8);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float x = t0 + u * NWV < T ? bf2f(xv[u]) : 0.f;
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                const bf16_t* pl = reinterpret_cast<const bf16_t*>(&lo[u][sb]);
                const bf16_t* ph = reinterpret_cast<const bf16_t*>(&hi[u][sb]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[sb][j] = fmaf(x, bf2f(pl[j]), acc[sb][j]);
                    acc[sb][8 + j] = fmaf(x, bf2f(ph[j]), acc[sb][8 + j]);
                }
            }
        }
    }
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        if (sb) __syncthreads();                 // the previous group's sums have been read
#pragma unroll
        for (int j = 0; j < 16; ++j) red[wave][j][lane] = acc[sb][j];
        __syncthreads();
        const int o = threadIdx.x;               // 1024 outputs, one per thread
        const int j = LARGE_IS_M ? (o & 15) : (o >> 6), ll = LARGE_IS_M ? (o >> 4) : (o & 63);
        const int lg = blockIdx.x * 64 + ll;
        if (lg < L) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) v += red[w][j][ll];
            float* p = LARGE_IS_M ? out + (size_t)lg * ldo + s0 + 16 * sb + j : out + (size_t)(s0 + 16 * sb + j) * ldo + lg;
            *p = (accumulate ? *p : 0.f) + scale * v;
        }
    }
}

// Round 4: the same contraction on the matrix pipe.  The kernel above reads its large operand two bytes per lane and spends ~35
// VALU instructions per token and wave (181 / 58 us per call on the packed 17 920-token micro-step, 0.4 / 1.3 TB/s); the FLOPs are
// nothing (16 output rows), the bytes are everything, and both operands have the contraction index (the token) as their ROW index
// while v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane.  So: a block owns 128 columns of the large operand and a chunk
// of tokens; stages of 64 tokens go HBM -> LDS by LDS-DMA in whole 256-byte row pieces (three stages, two in flight), and
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10) hands each lane its 4 + 4 tokens of one column: D[s][l] += small^T . large.
//   large image of a stage: 16 pieces of 1 KiB = 4 tokens x 256 B, chunk-major inside a piece (16-byte chunk c of token q at
//   (4 c + q) * 16: the four rows of a transposed read sit in 64 consecutive bytes -> conflict-free inside a 16-lane group);
//   small image(s): [token][32 B] per 16-wide group.
// The sums are fp32 in the MFMA's order (k ascending inside a chunk), chunks added by tn_reduce_kernel in index order as before:
// deterministic; not the bits of the VALU kernel (tests compare with torch fp32 to 2e-5).
template <bool LARGE_IS_M, int SB>
__global__ __launch_bounds__(256, 2) void tn_accum_mfma_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b,
                                                               int ldb, float* __restrict__ out, int ldo, int T, int M, int N,
                                                               float scale, int accumulate, int tchunk, int seg0, int seg1) {
    if (gridDim.z > 1) {
        const int t_begin = blockIdx.z * tchunk;
        a += (size_t)t_begin * lda;
        b += (size_t)t_begin * ldb;
        T = min(T - t_begin, tchunk);
        out += (size_t)blockIdx.z * M * N;      // part[z], dense [M][N]
        ldo = N; scale = 1.f; accumulate = 0;
    }
    constexpr int LC = 128, TS = 64, NST = 3;
    constexpr int BIG_B = TS * LC * 2, SM_B = TS * 32 * SB, STAGE = BIG_B + SM_B;
    constexpr int NPW = 4 + (SB == 1 ? 1 : 2);          // DMA pieces per wave and stage (every wave the same number: counted waits)
    extern __shared__ __attribute__((aligned(16))) char tn_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bf16_t* big = LARGE_IS_M ? a : b;
    const bf16_t* small = LARGE_IS_M ? b : a;
    const int ldbig = LARGE_IS_M ? lda : ldb, ldsm = LARGE_IS_M ? ldb : lda;
    const int L = LARGE_IS_M ? M : N;
    // seg0 / seg1 (dh_tn_accum_seg_f32; multiples of 128, INT_MAX = unused): the LARGE dimension is cut into up to three segments and
    // segment i contracts with columns 16 i .. of the small operand (the three LoRA-B gradients of the fused QKV projection in one launch)
    const int l0 = blockIdx.x * LC, s0 = blockIdx.y * 16 * SB + 16 * ((l0 >= seg0) + (l0 >= seg1));
    // ---- DMA sources.  large piece pi of a stage: tokens 4 pi + (lane & 3), 16-byte chunk lane >> 2 of the block's 256 bytes (clamped
    // to the row's last chunk: L % 8 == 0); wave w moves pieces 4 w .. 4 w + 3.  small piece sp: image sp >> 1, tokens 32 (sp & 1) + lane / 2.
    const int colc = min(l0 + 8 * (lane >> 2), L - 8);
    auto big_off = [&](int j, int t_stage) __attribute__((always_inline)) -> uint32_t {
        const int t = min(t_stage + 16 * wave + 4 * j + (lane & 3), T - 1);
        return ((uint32_t)t * (uint32_t)ldbig + (uint32_t)colc) * 2u;
    };
    auto small_piece = [&](int k) __attribute__((always_inline)) -> int { return SB == 1 ? (wave & 1) : (2 * (wave % 3) + k); };
    auto small_off = [&](int sp, int t_stage) __attribute__((always_inline)) -> uint32_t {
        const int t = min(t_stage + 32 * (sp & 1) + (lane >> 1), T - 1);
        return ((uint32_t)t * (uint32_t)ldsm + (uint32_t)(s0 + 16 * (sp >> 1) + 8 * (lane & 1))) * 2u;
    };
    auto issue = [&](int st) __attribute__((always_inline)) {
        char* base = tn_smem + (st % NST) * STAGE;
        const int t_stage = st * TS;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_saddr(big, big_off(j, t_stage), base + (4 * wave + j) * 1024);
#pragma unroll
        for (int k = 0; k < NPW - 4; ++k) {
            const int sp = small_piece(k);
            glds16_saddr(small, small_off(sp, t_stage), base + BIG_B + sp * 1024);
        }
    };
    const int nstages = (T + TS - 1) / TS;
    f32x4 acc[2][SB];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) acc[j][sb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- transposed reads: lane 4 q + p of a 16-lane group gives the address of row q, columns 4 p .. 4 p + 3; lane i receives column i
    const int kg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto tr = [&](const char* ptr) __attribute__((always_inline)) -> s16x4 { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr); };
    auto frag = [&](const char* lo, const char* hi) __attribute__((always_inline)) -> bf16x8 {
        const s16x4 x = tr(lo), y = tr(hi);
        return bf16x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    };
#pragma unroll 1
    for (int st = 0; st < NST - 1 && st < nstages; ++st) issue(st);
#pragma unroll 1
    for (int st = 0; st < nstages; ++st) {
        if (st + NST - 2 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // stage st has landed for every wave; every wave is done with stage st - 1
        if (st + NST - 1 < nstages) issue(st + NST - 1);
        const char* base = tn_smem + (st % NST) * STAGE;
        const bool tail = (st + 1) * TS > T;          // block-uniform: tokens past T are clamped copies — zero their products
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[SB], bfr[2];
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                const char* img = base + BIG_B + sb * 2048 + (32 * ks + 8 * kg + q) * 32 + 8 * p;
                af[sb] = frag(img, img + 4 * 32);
                if (tail) {
                    union { bf16x8 v; short e[8]; } u;
                    u.v = af[sb];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (st * TS + 32 * ks + 8 * kg + e >= T) u.e[e] = 0;
                    af[sb] = u.v;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ct = 2 * wave + j;
                const char* blk = base + (8 * ks + 2 * kg) * 1024 + ((2 * ct + (p >> 1)) * 4 + q) * 16 + 8 * (p & 1);
                bfr[j] = frag(blk, blk + 1024);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int sb = 0; sb < SB; ++sb) acc[j][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sb], bfr[j], acc[j][sb], 0, 0, 0);
        }
    }
    // D[s = 4 kg + r][l = 16 ct + (lane & 15)]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int l = l0 + 16 * (2 * wave + j) + (lane & 15);
        if (l >= L) continue;
#pragma unroll
        for (int sb = 0; sb < SB; ++sb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = blockIdx.y * 16 * SB + 16 * sb + 4 * kg + r;      // (output index: without the segment's column offset)
                float* o = LARGE_IS_M ? out + (size_t)l * ldo + s : out + (size_t)s * ldo + l;
                *o = (accumulate ? *o : 0.f) + scale * acc[j][sb][r];
            }
    }
}

// out[m][n] = (accumulate ? out : 0) + scale * (part[0] + part[1] + ... in index order)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int nz, float* __restrict__ out, int ldo,
                                                        int M, int N, float scale, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += part[(size_t)z * M * N + i];
    float* o = out + (size_t)(i / N) * ldo + i % N;
    *o = (accumulate ? *o : 0.f) + scale * s;
}

// D[t][h] = sum_d dO[t][h][d] * O[t][h][d]   (softmax backward row term)
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                     float* __restrict__ out, size_t rows, int hs) {
    const size_t row = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4;   // 16 lanes per row
    const int l = threadIdx.x & 15;
    if (row >= rows) return;
    float s = 0.f;
    for (int e = l; e < hs; e += 16) s += bf2f(a[row * hs + e]) * bf2f(b[row * hs + e]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
    if (l == 0) out[row] = s;
}

}  // namespace

extern "C" int dh_swiglu_bwd_bf16(const dh_bf16* dact, const dh_bf16* g, const dh_bf16* u, dh_bf16* dgu, int rows, int I,
                                  void* stream) {
    DH_CHECK(dact && g && u && dgu && rows >= 0 && I % 8 == 0, "dh_swiglu_bwd_bf16: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, dact, g, u, dgu, (size_t)rows, I);
    DH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- LoRA-branch dropout
// ger/lora.py:96,165,391: nn.Dropout(p) on the input of lora_A only.  y = bf16(x * m), m = keep ? bf16(1 / (1 - p)) : 0 — what
// torch computes for a bf16 tensor (mask.to(bf16) * scalar, then x * mask) — with the keep decisions drawn in the kernel from
// Philox4x32-10 (key = seed, call id; counter = micro-step read from DEVICE memory, element block), so a captured hipGraph of
// the micro-step draws fresh masks on every replay without a host-side launch parameter.  The mask is stored (bf16, one multiply
// in the backward: d/dx = dy * m).  Not reproducible against the reference's CPU generator by construction.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, bf16_t* __restrict__ mask,
                                                      size_t n8, uint32_t thresh, bf16_t keep_scale, uint64_t seed, uint32_t call_id,
                                                      const uint64_t* __restrict__ step_dev) {
    const uint64_t step = step_dev ? *step_dev : 0;
    const float ks = bf2f(keep_scale);
    for (size_t c8 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c8 < n8; c8 += (size_t)gridDim.x * blockDim.x) {
        // eight elements = two Philox calls of four 32-bit draws (counter: element block, half, micro-step)
        const uint4 xv = reinterpret_cast<const uint4*>(x)[c8];
        const bf16_t* xp = (const bf16_t*)&xv;
        uint4 yo, mo;
        bf16_t *yp = (bf16_t*)&yo, *mp = (bf16_t*)&mo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t ctr[4] = {(uint32_t)c8, (uint32_t)(c8 >> 32) * 2u + h, (uint32_t)step, (uint32_t)(step >> 32)};
            philox4x32_10(ctr, (uint32_t)seed ^ (call_id * 0x9E3779B9u), (uint32_t)(seed >> 32) + call_id);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool keep = ctr[e] >= thresh;                  // P(keep) = 1 - p
                mp[4 * h + e] = keep ? keep_scale : (bf16_t)0;
                yp[4 * h + e] = keep ? f2bf(bf2f(xp[4 * h + e]) * ks) : (bf16_t)0;
            }
        }
        reinterpret_cast<uint4*>(y)[c8] = yo;
        reinterpret_cast<uint4*>(mask)[c8] = mo;
    }
}

extern "C" int dh_dropout_bf16(const dh_bf16* x, dh_bf16* y, dh_bf16* mask, int64_t n, float p, uint64_t seed, uint32_t call_id,
                               const uint64_t* step_dev, void* stream) {
    DH_CHECK(x && y && mask && n % 8 == 0 && p >= 0.f && p < 1.f, "dh_dropout_bf16: bad argument (n=%lld, p=%g)", (long long)n, (double)p);
    if (n <= 0) return 0;
    const double t = (double)p * 4294967296.0;
    const uint32_t thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    const bf16_t ks = [](float v) { uint32_t u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }(1.0f / (1.0f - p));
    const size_t n8 = (size_t)(n / 8);
    const int blocks = (int)(n8 / 256 + 1 < 4096 ? n8 / 256 + 1 : 4096);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, mask, n8, thresh, ks, seed, call_id, step_dev);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_swiglu_fwd_bf16(const dh_bf16* g, const dh_bf16* u, dh_bf16* act, int64_t n, void* stream) {
    DH_CHECK(g && u && act && n % 8 == 0, "dh_swiglu_fwd_bf16: bad argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, g, u, act, (size_t)(n / 8));
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_rmsnorm_bwd_bf16(const dh_bf16* dy, const dh_bf16* x, const dh_bf16* w, const dh_bf16* dres, dh_bf16* dx,
                                   int rows, int d, float eps, void* stream) {
    DH_CHECK(dy && x && w && dx && d % 8 == 0 && d <= 8192, "dh_rmsnorm_bwd_bf16: bad argument (d=%d)", d);
    if (rows <= 0) return 0;
    dim3 grid(cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(MAXC) hipLaunchKernelGGL((rmsnorm_bwd_kernel<MAXC>), grid, block, 0, s, dy, x, w, dres, dx, rows, d, eps)
    if (d <= 512) { LAUNCH(1); } else if (d <= 2048) { LAUNCH(4); } else if (d <= 4096) { LAUNCH(8); } else { LAUNCH(16); }
#undef LAUNCH
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_qkv_rope_bwd_bf16(const dh_bf16* dq, const dh_bf16* dk, const dh_bf16* dv, const dh_bf16* cos,
                                    const dh_bf16* sin, const int32_t* tok_pos, dh_bf16* dqkv, int n_tok, int n_head,
                                    int n_groups, int hs, void* stream) {
    DH_CHECK(dq && dk && dv && cos && sin && tok_pos && dqkv, "dh_qkv_rope_bwd_bf16: null argument");
    DH_CHECK(hs == 64 || hs == 128, "dh_qkv_rope_bwd_bf16: head_size %d unsupported", hs);
    if (n_tok <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (hs == 64)
        hipLaunchKernelGGL((qkv_rope_bwd_kernel<64>), dim3(1024), dim3(256), 0, s, dq, dk, dv, cos, sin, tok_pos, dqkv, n_tok, n_head, n_groups);
    else
        hipLaunchKernelGGL((qkv_rope_bwd_kernel<128>), dim3(1024), dim3(256), 0, s, dq, dk, dv, cos, sin, tok_pos, dqkv, n_tok, n_head, n_groups);
    DH_LAUNCH_CHECK();
    return 0;
}

constexpr int TN_CHUNK = 512;     // tokens per block of the split form
extern "C" int64_t dh_tn_accum_work_bytes(int T, int M, int N) {
    return T > 2 * TN_CHUNK ? (int64_t)cdiv(T, TN_CHUNK) * M * N * (int64_t)sizeof(float) : 0;
}

static int tn_accum_impl(const dh_bf16* a, int lda, const dh_bf16* b, int ldb, float* out, int ldo, int T, int M,
                         int N, float scale, int accumulate, void* work, void* stream, int seg0, int seg1) {
    DH_CHECK(a && b && out && T >= 0 && M > 0 && N > 0, "dh_tn_accum_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    // the small operand is read as aligned 16-byte runs
    const bool n_small = N % 16 == 0 && N <= 64 && M >= 64 && ldb % 8 == 0 && ((uintptr_t)b & 15) == 0;
    const bool m_small = M % 16 == 0 && M <= 64 && N >= 64 && lda % 8 == 0 && ((uintptr_t)a & 15) == 0;
    if (n_small || m_small) {
        // long token loops (packed micro-batches) are split over the grid's z when the caller brought scratch
        const int nz = work != nullptr && T > 2 * TN_CHUNK ? cdiv(T, TN_CHUNK) : 1;
        float* dst = nz > 1 ? (float*)work : out;
        // the MFMA form (tn_accum_mfma_kernel): the large operand in aligned 16-byte chunks addressed through 32-bit offsets
        const bf16_t* big = n_small ? a : b;
        const int ldbig = n_small ? lda : ldb, L = n_small ? M : N, S = n_small ? N : M, ldsm = n_small ? ldb : lda;
        const int t_span = nz > 1 ? TN_CHUNK : T;
        const bool segmented = seg0 != INT_MAX || seg1 != INT_MAX;
        DH_CHECK(!segmented || (n_small && N == 16), "dh_tn_accum_seg_f32: segments need a large M and N == 16");
        const bool mfma = (g_tn_mfma || segmented) && T > 0 && L % 8 == 0 && ldbig % 8 == 0 && ((uintptr_t)big & 15) == 0 &&
                          (size_t)t_span * ldbig * 2 < (1ull << 32) && (size_t)t_span * ldsm * 2 < (1ull << 32);
        DH_CHECK(mfma || !segmented, "dh_tn_accum_seg_f32: operands must be 16-byte aligned with row strides that are multiples of 8");
        if (mfma) {
            const int sb = S == 48 ? 3 : 1;
            const dim3 grid(cdiv(L, 128), S / (16 * sb), nz), block(256);
            const int lds1 = 3 * (64 * 128 * 2 + 64 * 32), lds3 = 3 * (64 * 128 * 2 + 64 * 32 * 3);
            if (n_small && sb == 1) {
                hipLaunchKernelGGL((tn_accum_mfma_kernel<true, 1>), grid, block, lds1, st, a, lda, b, ldb, dst, ldo, T, M, N, scale, accumulate, TN_CHUNK, seg0, seg1);
            } else if (n_small) {
                DH_MAX_LDS_ONCE((tn_accum_mfma_kernel<true, 3>), lds3);
                hipLaunchKernelGGL((tn_accum_mfma_kernel<true, 3>), grid, block, lds3, st, a, lda, b, ldb, dst, ldo, T, M, N, scale, accumulate, TN_CHUNK, seg0, seg1);
            } else if (sb == 1) {
                hipLaunchKernelGGL((tn_accum_mfma_kernel<false, 1>), grid, block, lds1, st, a, lda, b, ldb, dst, ldo, T, M, N, scale, accumulate, TN_CHUNK, seg0, seg1);
            } else {
                DH_MAX_LDS_ONCE((tn_accum_mfma_kernel<false, 3>), lds3);
                hipLaunchKernelGGL((tn_accum_mfma_kernel<false, 3>), grid, block, lds3, st, a, lda, b, ldb, dst, ldo, T, M, N, scale, accumulate, TN_CHUNK, seg0, seg1);
            }
        } else if (n_small)
            hipLaunchKernelGGL((tn_accum_wide_kernel<true, 1>), dim3(cdiv(M, 64), N / 16, nz), dim3(1024), 0, st, a, lda, b, ldb, dst, ldo,
                               T, M, N, scale, accumulate, TN_CHUNK);
        else
            hipLaunchKernelGGL((tn_accum_wide_kernel<false, 1>), dim3(cdiv(N, 64), M / 16, nz), dim3(1024), 0, st, a, lda, b, ldb, dst, ldo,
                               T, M, N, scale, accumulate, TN_CHUNK);
        if (nz > 1)
            hipLaunchKernelGGL(tn_reduce_kernel, dim3(cdiv(M * N, 256)), dim3(256), 0, st, (const float*)work, nz, out, ldo, M, N, scale,
                               accumulate);
        DH_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(tn_accum_kernel, dim3(cdiv(M, 16), cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out,
                       ldo, T, M, N, scale, accumulate);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_tn_accum_f32(const dh_bf16* a, int lda, const dh_bf16* b, int ldb, float* out, int ldo, int T, int M,
                               int N, float scale, int accumulate, void* work, void* stream) {
    return tn_accum_impl(a, lda, b, ldb, out, ldo, T, M, N, scale, accumulate, work, stream, INT_MAX, INT_MAX);
}

// out[m][n] (+)= scale * sum_t a[t][m] * b[t][16 seg(m) + n], n < 16, seg(m) = (m >= seg0) + (m >= seg1): the LoRA-B gradients of the
// three segments of a fused QKV projection (ger/lora.py:367-402: q, k and v have their own rank-16 pair) in ONE launch over the
// projection's output gradient instead of one per segment.  seg0 <= seg1, multiples of 128.
extern "C" int dh_tn_accum_seg_f32(const dh_bf16* a, int lda, const dh_bf16* b, int ldb, float* out, int ldo, int T, int M,
                                   int seg0, int seg1, float scale, int accumulate, void* work, void* stream) {
    DH_CHECK(seg0 >= 0 && seg0 <= seg1 && seg0 % 128 == 0 && seg1 % 128 == 0, "dh_tn_accum_seg_f32: segment starts must be ordered multiples of 128");
    DH_CHECK(ldb >= 48, "dh_tn_accum_seg_f32: b holds 3 x 16 columns");
    return tn_accum_impl(a, lda, b, ldb, out, ldo, T, M, 16, scale, accumulate, work, stream, seg0, seg1);
}

extern "C" int dh_rowdot_f32(const dh_bf16* a, const dh_bf16* b, float* out, int64_t rows, int hs, void* stream) {
    DH_CHECK(a && b && out && hs > 0, "dh_rowdot_f32: bad argument");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                       (size_t)rows, hs);
    DH_LAUNCH_CHECK();
    return 0;
}
