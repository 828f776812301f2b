// Causal GQA attention against the compact KV cache (ger/model.py:261,270-290).
//
// Layout chosen for the MFMA operand maps, not inherited from the reference: both caches are stored
// in MFMA-FRAGMENT ORDER (common.h kfrag_off / vfrag_off): per (slot, group) a sequence of 32-key
// tiles, each tile a run of 1-KiB blocks that are, byte for byte, the A-operand fragments of
//   S^T = K . Q^T   (block (ks): lane i holds K[key i&31][16ks + 8(i>>5) .. +8])
//   O^T = V^T . P^T (block (dt,s2): lane i holds V[8 permuted keys][d = 32dt + (i&31)])
// so a decode wave streams a tile as 8 fully coalesced 1-KiB loads straight into MFMA operands,
// and the prefill kernel copies tiles linearly into LDS and reads them back lane-linear
// (conflict-free ds_read_b128, no swizzle).
// Both products keep the QUERY on the accumulator's lane axis:
//   S^T[key][q] = K · Q^T      A = K rows (16 B of a key row),  B = Q rows (16 B of a q row)
//   O^T[d][q]   = V^T · P^T    A = V^T rows (keys contiguous),  B = the S^T accumulator itself,
//                              converted to bf16 in registers (cdna_hip_programming.md §3
//                              "an accumulator tile as the next MFMA's operand": k-slot j of
//                              lane-half h is key 16s + 8(j>>2) + 4h + (j&3), so the V^T
//                              fragment is two 8-byte runs of 4 keys).
// so the online-softmax statistics (running max m, running sum l) are one scalar per lane, the
// rescale of O is a per-lane multiply, and P never touches LDS.  Softmax is fp32; P is rounded
// to bf16 for the second product; O/l is rounded to bf16 once (as the reference's fused CPU/GPU
// SDPA kernels do).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------ prefill
// grid (q tiles of 32, n_groups, n_seq); block = 64 * q_per_kv threads: one wave per query head
// of the group, all waves share the K / V^T tiles of 64 keys staged in LDS.
// PS: scale is a power of two (head size 64: 1/8), so it is applied to the Q fragments once — exact in bf16 and in the
// fp32 products and sums (barring underflow) — instead of to every score: the loop is bound by its softmax VALU work.
template <int HS, bool PS = false>
__global__ __launch_bounds__(HS == 64 ? 1024 : 512) void attn_prefill_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_cache, const bf16_t* __restrict__ vT_cache,
    const int32_t* __restrict__ seq_slot, const int32_t* __restrict__ q_start, const int32_t* __restrict__ q_len,
    const int32_t* __restrict__ kv_pos0, bf16_t* __restrict__ y, float* __restrict__ lse, int n_head, int n_groups,
    int s_max, float scale) {
    constexpr int KS = HS / 16;   // k-steps of the QK product
    constexpr int DT = HS / 32;   // 32-row tiles of O^T
    constexpr int TILE_B = HS * 32 * 2;   // bytes of one 32-key tile of K (and of V^T)
    // two stages of (two 32-key tiles of K, two of V^T): the tiles of step kt+1 are DMA'd (global_load_lds; the
    // cache is already in fragment order, so the copy is linear) while step kt is multiplied
    // (Round 3, measured and not kept: FOUR stages at head size 64, DMA three steps ahead with counted waits — 217 us per launch against
    //  209: the 40 % of wave cycles parked at waits (tools/pmc_attn_prefill.py: SQ_WAIT_ANY 100.6 M of 251 M) are not DMA latency.
    //  Nor are they instruction count or phase lock-step, as far as two more experiments go: the subtraction and the log2 e
    //  multiplication of the softmax on packed pairs (24 fewer VALU instructions per step) 207-210 us; S of step kt+1 issued in front
    //  of the softmax of step kt (software pipelining inside the wave, bit-identical, one more S tile: 163 VGPRs = one block per
    //  CU 284 us, forced to 128 with 17 spills 277 us).  Round 4: all eight K fragments of a step read in one batch in front of the
    //  eight S MFMAs and the eight V^T fragments in front of the PV MFMAs (sched_barrier; 128 VGPRs, 5 spilled): 200-209 us against 200-203,
    //  tools/time_attn_prefill.py — the operand reads' LDS latency is not it either.)
    __shared__ __attribute__((aligned(16))) char sKV[2][4 * TILE_B];

    const int seq = blockIdx.z, g = blockIdx.y;
    const int qt = gridDim.x - 1 - blockIdx.x;          // longest (latest) tiles first
    const int qlen = q_len[seq];
    const int q0 = qt * 32;
    if (q0 >= qlen) return;                             // uniform for the whole block
    const int slot = seq_slot[seq], qs = q_start[seq], p0 = kv_pos0[seq] + q0;
    const int q_per_kv = n_head / n_groups;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int head = g * q_per_kv + wave;

    // Q fragments (B operand): lane holds Q[q0+lr][ks*16 + lh*8 .. +8)
    bf16x8 qf[KS];
    {
        int row = q0 + lr;
        row = row < qlen ? row : qlen - 1;
        const bf16_t* qp = q + ((size_t)(qs + row) * n_head + head) * HS + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
        if (PS) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = (short)f2bf(bf2f((bf16_t)qf[ks][e]) * scale);
        }
    }

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const bf16_t* kbase = k_cache + ((size_t)slot * n_groups + g) * s_max * HS;
    const bf16_t* vbase = vT_cache + ((size_t)slot * n_groups + g) * HS * s_max;
    // keys beyond the tile's last real query position are never attended (and may not exist)
    const int last_key = min(p0 + 31, kv_pos0[seq] + qlen - 1);
    const int n_tiles = last_key / 64 + 1;
    const int q_abs = p0 + lr;

    // one-KiB groups (64 lanes x 16 B) of a stage dealt over the waves: 2*TILE_B/1024 of K then as many of V^T
    constexpr int GRP = 2 * TILE_B / 1024;
    const int nwaves = nthr >> 6;
    auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
        const char* gk = reinterpret_cast<const char*>(kbase + (size_t)(2 * kt) * HS * 32);
        const char* gv = reinterpret_cast<const char*>(vbase + (size_t)(2 * kt) * HS * 32);
        for (int gidx = wave; gidx < 2 * GRP; gidx += nwaves) {
            const char* src = gidx < GRP ? gk + gidx * 1024 : gv + (gidx - GRP) * 1024;
            glds16(src + lane * 16, sKV[buf] + gidx * 1024);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int key0 = kt * 64;
        // buffer (kt+1)&1 was last read in iteration kt-1, which ended with a barrier
        if (kt + 1 < n_tiles) stage((kt + 1) & 1, kt + 1);
        const char* sK = sKV[kt & 1];
        const char* sV = sK + 2 * TILE_B;

        // ---- S^T = K · Q^T for the two 32-key row tiles
        f32x16 st[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[rt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (rt * KS + ks) * 1024 + lane * 16);
                st[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[rt], 0, 0, 0);
            }
        }
        // ---- scale, causal mask (only a tile that reaches past the block's first query can hold masked keys), tile max
        if (!PS) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[rt][r] *= scale;
        }
        if (key0 + 63 > p0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key_abs = key0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    st[rt][r] = key_abs <= q_abs ? st[rt][r] : -INFINITY;
                }
        }
        float m_t = -INFINITY;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_t = fmaxf(m_t, st[rt][r]);
        m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
        const float m_new = fmaxf(m_run, m_t);           // finite: key 0 is always visible
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(st[rt][r] - m_new);
                st[rt][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {   // the running max moved for some query of the wave
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }

        // ---- O^T += V^T · P^T
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) pf.u[j] = pack2bf(st[rt][8 * s + 2 * j], st[rt][8 * s + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + (((rt * DT + dt) * 2 + s) * 1024) + lane * 16);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o[dt], 0, 0, 0);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next stage has landed
        __syncthreads();                                    // ... everybody's, and this stage's reads are done
    }

    // ---- normalise and store y[q][head*HS + d]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (lse != nullptr && lh == 0 && q0 + lr < qlen)     // log-sum-exp of the scaled scores (training backward)
        lse[(size_t)(qs + q0 + lr) * n_head + head] = m_run + logf(l_tot);
    if (q0 + lr < qlen) {
        bf16_t* yp = y + (size_t)(qs + q0 + lr) * n_head * HS + head * HS;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = dt * 32 + 8 * rg + 4 * lh;
                uint2 pk = make_uint2(pack2bf(o[dt][rg * 4 + 0] * inv, o[dt][rg * 4 + 1] * inv),
                                      pack2bf(o[dt][rg * 4 + 2] * inv, o[dt][rg * 4 + 3] * inv));
                *reinterpret_cast<uint2*>(yp + d) = pk;
            }
    }
}

// ------------------------------------------------------------------------------------ decode
// One query per sequence.  The q_per_kv heads of a group sit on the accumulator's lane axis
// (columns >= q_per_kv are zero padding: the step is HBM-bound, the idle MFMA columns are free).
// grid (n_seq*n_groups, NSPLIT); each of the 4*NSPLIT waves of a (sequence, group) pair walks
// 32-key tiles straight from HBM to registers (no LDS: nothing is shared between waves) and
// leaves an (m, l, O^T) partial; attn_decode_combine_kernel merges them.
constexpr int DEC_COLS = 16;   // padded head columns in the partials

template <int HS>
__global__ __launch_bounds__(256) void attn_decode_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_cache, const bf16_t* __restrict__ vT_cache,
    const int32_t* __restrict__ seq_slot, const int32_t* __restrict__ kv_len, float* __restrict__ work,
    int n_head, int n_groups, int s_max, float scale) {
    constexpr int KS = HS / 16, DT = HS / 32;
    const int pair = blockIdx.x, seq = pair / n_groups, g = pair % n_groups;
    const int nw = gridDim.y * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg = blockIdx.y * 4 + wave;
    const int lr = lane & 31, lh = lane >> 5;
    const int q_per_kv = n_head / n_groups;
    const int slot = seq_slot[seq], len = kv_len[seq];
    const int n_tiles = (len + 31) / 32;

    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (lr < q_per_kv)
            z = *reinterpret_cast<const bf16x8*>(q + ((size_t)seq * n_head + g * q_per_kv + lr) * HS + ks * 16 + lh * 8);
        qf[ks] = z;
    }
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const bf16_t* kbase = k_cache + ((size_t)slot * n_groups + g) * s_max * HS;
    const bf16_t* vbase = vT_cache + ((size_t)slot * n_groups + g) * HS * s_max;

    for (int t = wg; t < n_tiles; t += nw) {
        const int key0 = t * 32;
        // issue all loads of the tile first (K: KS x 16 B, V^T: DT x 2 x 2 x 8 B per lane)
        // one 32-key tile = 8 coalesced 1-KiB loads that ARE the MFMA fragments (keys >= len are
        // masked below; their cache bytes are finite)
        bf16x8 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            kf[ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kbase + kfrag_blk<HS>(t, ks) + lane * 8));
        struct VF { bf16x8 v; };
        VF vf[DT][2];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                vf[dt][s].v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vbase + vfrag_blk<HS>(t, dt, s) + lane * 8));
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], st, 0, 0, 0);
        float m_t = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key_abs = key0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float s = st[r] * scale;
            s = key_abs < len ? s : -INFINITY;
            st[r] = s;
            m_t = fmaxf(m_t, s);
        }
        m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
        const float m_new = fmaxf(m_run, m_t);           // finite: key0 < len
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(st[r] - m_new);
            st[r] = p;
            psum += p;
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.u[j] = pack2bf(st[8 * s + 2 * j], st[8 * s + 2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (s == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                }
                // keys >= len carry p == 0; the cache is zero-initialised so 0 * v stays 0
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][s].v, pf.v, o[dt], 0, 0, 0);
            }
        }
    }
    // partial: [pair][wg] -> m[DEC_COLS], l[DEC_COLS], o[HS][DEC_COLS]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float* wp = work + ((size_t)pair * nw + wg) * (2 * DEC_COLS + HS * DEC_COLS);
    if (lr < q_per_kv) {
        if (lh == 0) {
            wp[lr] = m_run;
            wp[DEC_COLS + lr] = l_tot;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                wp[2 * DEC_COLS + d * DEC_COLS + lr] = o[dt][r];
            }
    }
}

template <int HS>
__global__ void attn_decode_combine_kernel(const float* __restrict__ work, bf16_t* __restrict__ y, int n_head,
                                           int n_groups, int nw) {
    const int pair = blockIdx.x, seq = pair / n_groups, g = pair % n_groups;
    const int q_per_kv = n_head / n_groups;
    const int stride = 2 * DEC_COLS + HS * DEC_COLS;
    for (int it = threadIdx.x; it < q_per_kv * HS; it += blockDim.x) {
        const int h = it / HS, d = it % HS;
        const float* wp = work + (size_t)pair * nw * stride;
        float M = -INFINITY;
        for (int w = 0; w < nw; ++w) M = fmaxf(M, wp[w * stride + h]);
        float L = 0.f, O = 0.f;
        for (int w = 0; w < nw; ++w) {
            const float mw = wp[w * stride + h];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            L += wp[w * stride + DEC_COLS + h] * f;
            O += wp[w * stride + 2 * DEC_COLS + d * DEC_COLS + h] * f;
        }
        y[(size_t)seq * n_head * HS + (g * q_per_kv + h) * HS + d] = f2bf(O / L);
    }
}

constexpr int DEC_NSPLIT = 4;

}  // namespace

extern "C" int dh_attn_prefill_bf16(const dh_bf16* q, const dh_bf16* k_cache, const dh_bf16* vT_cache,
                                    const int32_t* seq_slot, const int32_t* q_start, const int32_t* q_len,
                                    const int32_t* kv_pos0, dh_bf16* y, float* lse, int n_seq, int max_q_len,
                                    int n_head, int n_groups, int hs, int s_max, void* stream) {
    DH_CHECK(n_groups > 0 && n_head % n_groups == 0 && n_head / n_groups <= 16, "dh_attn_prefill_bf16: bad head counts");
    DH_CHECK(hs == 64 || hs == 128, "dh_attn_prefill_bf16: head_size %d unsupported", hs);
    DH_CHECK(s_max % 64 == 0, "dh_attn_prefill_bf16: s_max must be a multiple of 64");
    DH_CHECK(64 * (n_head / n_groups) <= (hs == 64 ? 1024 : 512), "dh_attn_prefill_bf16: too many query heads per group");
    if (n_seq <= 0 || max_q_len <= 0) return 0;
    dim3 grid(cdiv(max_q_len, 32), n_groups, n_seq), block(64 * (n_head / n_groups));
    const float scale = 1.0f / sqrtf((float)hs);
    hipStream_t s = (hipStream_t)stream;
    if (hs == 64)
        hipLaunchKernelGGL((attn_prefill_kernel<64, true>), grid, block, 0, s, q, k_cache, vT_cache, seq_slot, q_start, q_len,   // scale = 1/8
                           kv_pos0, y, lse, n_head, n_groups, s_max, scale);
    else
        hipLaunchKernelGGL((attn_prefill_kernel<128>), grid, block, 0, s, q, k_cache, vT_cache, seq_slot, q_start, q_len,
                           kv_pos0, y, lse, n_head, n_groups, s_max, scale);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t dh_attn_decode_work_bytes(int n_seq, int n_head, int hs, int s_max) {
    (void)s_max;
    // upper bound over n_groups: pairs <= n_seq * n_head
    return (int64_t)n_seq * n_head * (DEC_NSPLIT * 4) * (2 * DEC_COLS + hs * DEC_COLS) * sizeof(float);
}

extern "C" int dh_attn_decode_bf16(const dh_bf16* q, const dh_bf16* k_cache, const dh_bf16* vT_cache,
                                   const int32_t* seq_slot, const int32_t* kv_len, dh_bf16* y, void* work, int n_seq,
                                   int n_head, int n_groups, int hs, int s_max, void* stream) {
    DH_CHECK(n_groups > 0 && n_head % n_groups == 0 && n_head / n_groups <= DEC_COLS, "dh_attn_decode_bf16: bad head counts");
    DH_CHECK(hs == 64 || hs == 128, "dh_attn_decode_bf16: head_size %d unsupported", hs);
    DH_CHECK(s_max % 64 == 0, "dh_attn_decode_bf16: s_max must be a multiple of 64");
    DH_CHECK(work != nullptr, "dh_attn_decode_bf16: null workspace");
    if (n_seq <= 0) return 0;
    const int pairs = n_seq * n_groups;
    const float scale = 1.0f / sqrtf((float)hs);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(pairs, DEC_NSPLIT), block(256);
    if (hs == 64) {
        hipLaunchKernelGGL((attn_decode_kernel<64>), grid, block, 0, s, q, k_cache, vT_cache, seq_slot, kv_len,
                           (float*)work, n_head, n_groups, s_max, scale);
        hipLaunchKernelGGL((attn_decode_combine_kernel<64>), dim3(pairs), dim3(256), 0, s, (const float*)work, y, n_head,
                           n_groups, DEC_NSPLIT * 4);
    } else {
        hipLaunchKernelGGL((attn_decode_kernel<128>), grid, block, 0, s, q, k_cache, vT_cache, seq_slot, kv_len,
                           (float*)work, n_head, n_groups, s_max, scale);
        hipLaunchKernelGGL((attn_decode_combine_kernel<128>), dim3(pairs), dim3(256), 0, s, (const float*)work, y, n_head,
                           n_groups, DEC_NSPLIT * 4);
    }
    DH_LAUNCH_CHECK();
    return 0;
}
