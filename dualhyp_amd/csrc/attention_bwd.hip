// Backward of the causal GQA attention (training path, no KV cache: every sequence attends only to
// itself).  Flash-style: P is recomputed from Q, K and the forward's log-sum-exp, nothing S x S is
// stored.  Five MFMA products per (32-query, 32-key) tile pair, arranged so that every product either
// contracts over the head dimension (operands = 16-byte runs of row-major rows) or takes a freshly
// computed accumulator tile as its operand (cdna_hip_programming.md §3 "an accumulator tile as the
// next MFMA's operand"); the other operand of those is then a column fragment, served from
// token-contiguous transposed copies qT / doT / kT ([heads][hs][padded tokens], made by
// transpose_pad_kernel; every sequence starts on a multiple of 32 there, so fragments are aligned).
//
//   dkdv kernel : grid (kv tile, group, seq), one wave per query head of the group
//       S [q][key] = Q.K^T ; dP[q][key] = dO.V^T ; P = exp(S*scale - lse[q]) ; dS = P*(dP - D[q])
//       dV^T[d][key] += dO^T[d][q] . P[q][key]      (A = doT fragment, B = P accumulator)
//       dK^T[d][key] += Q^T [d][q] . dS[q][key]     (A = qT fragment,  B = dS accumulator)
//       per-wave partials are summed over the group's heads through LDS
//   dq kernel   : grid (q tile, head, seq), one wave
//       S^T[key][q] = K.Q^T ; dP^T = V.dO^T ; dS^T = P^T*(dP^T - D[q])
//       dQ^T[d][q] += K^T[d][key] . dS^T[key][q]    (A = kT fragment,  B = dS^T accumulator)
#include "common.h"

int g_attn_bwd_dkdv_img = 1;    // dh_set_tuning(29, 0): the dkdv kernel with every operand fetched from global memory per wave (A/B)
int g_attn_bwd_dq_group = 1;    // dh_set_tuning(27, 0): the single-wave dq kernel of rounds 2-3 (A/B)

namespace {

// Token-contiguous copies of q / dO / k for the products that take them TRANSPOSED, stored in MFMA-FRAGMENT ORDER (round 3; as the
// KV cache of the decode path): the A operand of v_mfma_f32_32x32x16_bf16 that multiplies a P / dS accumulator tile needs, per
// lane (lr = channel % 32, lh), the 8 values of tokens {qo..qo+3, qo+8..qo+11}, qo = 16 s2 + 4 lh, of channel dt*32 + lr.  Padded
// token p (= pad_start[seq] + position, every sequence starting on a multiple of 32) of channel e of head h lives at
//     tfrag_off = ((((h * n_pad/32 + p/32) * 2 + s2) * (hs/32) + e/32) * 64 + lh*32 + e%32) * 8 + j
// with w = p % 32, s2 = w/16, r = w%16, lh = (r%8)/4, j = 4*(r/8) + r%4 — so a wave's fragment is ONE contiguous 1-KiB load
// (16 B per lane) instead of two 8-byte gathers per lane from 64 different cache lines (the round-2 layout [heads][hs][n_pad]:
// the texture-address unit, not the matrix pipe, set the pace of both backward kernels).
template <int HS>
__device__ __forceinline__ size_t tfrag_off(int h, int n_pad, int p, int e) {
    const int w = p & 31, s2 = w >> 4, r = w & 15, lh = (r & 7) >> 2, j = ((r >> 3) << 2) + (r & 3);
    return (((((size_t)h * (n_pad >> 5) + (p >> 5)) * 2 + s2) * (HS / 32) + (e >> 5)) * 64 + lh * 32 + (e & 31)) * 8 + j;
}
// 16-byte fragment of (head h, padded tile `tile`, s2, dt) for this lane
template <int HS>
__device__ __forceinline__ size_t tfrag_lane(int h, int n_pad, int tile, int s2, int dt, int lane) {
    return (((((size_t)h * (n_pad >> 5) + tile) * 2 + s2) * (HS / 32) + dt) * 64 + lane) * 8;
}

// src [n_tok, heads, hs] -> dst in fragment order; token t of sequence i is padded token pad_start[i] + (t - start[i])
__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                            const int32_t* __restrict__ tok_seq,
                                                            const int32_t* __restrict__ q_start,
                                                            const int32_t* __restrict__ pad_start, int n_tok, int heads,
                                                            int hs, int n_pad) {
    __shared__ bf16_t tile[32][130];
    const int t0 = blockIdx.x * 32, h = blockIdx.y;
    for (int it = threadIdx.x; it < 32 * hs; it += 256) {
        const int tl = it / hs, e = it % hs, t = t0 + tl;
        tile[tl][e] = t < n_tok ? src[((size_t)t * heads + h) * hs + e] : (bf16_t)0;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < 32 * hs; it += 256) {
        const int tl = it & 31, e = it >> 5, t = t0 + tl;
        if (t < n_tok) {
            const int s = tok_seq[t];
            const int pp = pad_start[s] + (t - q_start[s]);
            dst[hs == 64 ? tfrag_off<64>(h, n_pad, pp, e) : tfrag_off<128>(h, n_pad, pp, e)] = tile[tl][e];
        }
    }
}

// Round 4: the same copy with 16-byte accesses on both sides.  One block per padded 32-token tile and head: the tile's rows come in
// as 16-byte chunks (token of padded position p = pad_tok[p], -1 = padding -> zeros, so the destination needs no memset), go through a
// [32][hs + 8] LDS image, and ds_read_b64_tr_b16 (cdna_hip_programming.md T10) hands lane (lh, channel) its 4 + 4 tokens: the tile's
// 2 x hs/32 fragments leave as whole 1-KiB wave stores.  (The kernel above moves 2 bytes per lane each way: 80 us per call on the
// packed 17 920-token micro-step, 1.8 TB/s.)
template <int HS>
__global__ __launch_bounds__(256) void transpose_frag_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const int32_t* __restrict__ pad_tok, int heads, int n_pad) {
    constexpr int STR = (HS + 8) * 2, DT = HS / 32;
    __shared__ __attribute__((aligned(16))) char tile[32 * STR];
    const int pt = blockIdx.x, h = blockIdx.y;
#pragma unroll
    for (int it = threadIdx.x; it < 32 * (HS / 8); it += 256) {
        const int tl = it / (HS / 8), c = it % (HS / 8);
        const int tok = pad_tok[pt * 32 + tl];
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tok >= 0) v = *reinterpret_cast<const uint4*>(src + ((size_t)tok * heads + h) * HS + c * 8);
        *reinterpret_cast<uint4*>(tile + tl * STR + c * 16) = v;
    }
    __syncthreads();
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, lh = g >> 1;
#pragma unroll
    for (int f = wave; f < 2 * DT; f += 4) {                      // fragment (s2, dt) of the tile
        const int s2 = f / DT, dt = f % DT;
        const char* rd = tile + (16 * s2 + 4 * lh + q) * STR + (dt * 32 + (g & 1) * 16 + 4 * p) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)rd);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rd + 8 * STR));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        *reinterpret_cast<uint4*>(dst + tfrag_lane<HS>(h, n_pad, pt, s2, dt, lane)) = make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
}

// IMG (round 4, hs 64): a wave's q / dO tile goes HBM/L2 -> LDS ONCE by LDS-DMA (a private double-buffered 2 x 8 KiB image per wave) and
// serves BOTH operand shapes — row fragments by ds_read_b128, transposed fragments by ds_read_b64_tr_b16 (cdna_hip_programming.md T10) —
// instead of 8 KiB of row loads plus 8 KiB of the transposed copies qT / doT per tile pair (the kernel's bound: 2.8 GB per call from
// L2 on the packed micro-step); qT / doT are then not read at all.  Image of a tile: 4 pieces of 1 KiB = 8 rows x 128 B, 16-byte chunk c of
// row r of a piece at (8 c + r) * 16: a DMA piece is one whole-line request per row, a row fragment is 8 consecutive lanes on 128 B,
// the four rows of a transposed read sit in 64 consecutive bytes.  Same products in the same order: the bits of the register form.
template <int HS, int QPKT, bool IMG = false>   // QPKT <= query heads per group (block = 64 x that many threads): sizes the per-thread totals
__global__ __launch_bounds__(512) void attn_bwd_dkdv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qT, const bf16_t* __restrict__ doT,
    const float* __restrict__ lse, const float* __restrict__ dsum, const int32_t* __restrict__ q_start,
    const int32_t* __restrict__ q_len, const int32_t* __restrict__ pad_start, bf16_t* __restrict__ dk,
    bf16_t* __restrict__ dv, int n_head, int n_groups, int n_pad, float scale, int nt, int n_seq) {
    constexpr int KS = HS / 16, DT = HS / 32;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [waves][2][DT][16 regs][64 lanes]
    // 1-D grid, XCD-aware (blocks go round-robin over the 8 XCDs): all key tiles of a (group, sequence) pair run on ONE XCD,
    // so the pair's transposed q / dO copies (64 d-rows x 1.1 KB, strided by the PACKED token count) stay in that XCD's
    // 4 MiB L2.  With 16 packed sequences the row-major walk had every XCD touch every pair (16 MB per L2): each 8-byte
    // fragment piece then fetched its own 128-B line from the fabric and the kernel ran 10x longer than the forward.
    int seq, g, kt;
    {
        const int np = n_groups * n_seq;
        if (np >= 8) {
            const int b = blockIdx.x, j = b >> 3, pair = (b & 7) + 8 * (j / nt);
            if (pair >= np) return;
            kt = j % nt; g = pair % n_groups; seq = pair / n_groups;
        } else {
            const int b = blockIdx.x;
            kt = b % nt; g = (b / nt) % n_groups; seq = b / (nt * n_groups);
            if (seq >= n_seq) return;
        }
    }
    const int len = q_len[seq];
    if (kt * 32 >= len) return;
    const int qs = q_start[seq], ps = pad_start[seq];
    const int q_per_kv = n_head / n_groups;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int head = g * q_per_kv + wave;
    const int key0 = kt * 32;

    // K and V row fragments of this key tile (B operands), kept for the whole block
    bf16x8 kf[KS], vf[KS];
    {
        int key = key0 + lr;
        key = key < len ? key : len - 1;
        const bf16_t* kp = k + ((size_t)(qs + key) * n_groups + g) * HS + lh * 8;
        const bf16_t* vp = v + ((size_t)(qs + key) * n_groups + g) * HS + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 16);
        }
    }
    f32x16 dkT[DT], dvT[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkT[dt][r] = 0.f; dvT[dt][r] = 0.f; }

    const int n_qt = (len + 31) / 32;
    const int tile0 = ps >> 5;           // first padded tile of this sequence
    // Software pipeline (round 3): a wave walks its query tiles alone (nothing is shared between the heads of a block), so
    // every global load it waits for is exposed.  The row fragments of q / dO and the row terms (lse, D) of tile qt+1 are
    // requested at the top of iteration qt, the transposed pieces of tile qt before its first MFMA: by the time they are
    // needed one whole phase of MFMAs and exps has passed.  lse / D travel as ONE value per lane (row q0 + lane % 32) and
    // reach the accumulator layout's rows through ds_bpermute instead of 16 loads per lane each.
    static_assert(!IMG || HS == 64, "the LDS image form is sized for hs 64 (16 KiB per wave)");
    constexpr bool PIPE = HS == 64 && !IMG;       // hs 128 doubles every fragment set: the second set of row operands would spill
    struct RowOps { bf16x8 qf[KS], dof[KS]; float l, d; };
    char* img = reinterpret_cast<char*>(red) + (IMG ? __builtin_amdgcn_readfirstlane(wave) * 16384 : 0);   // IMG: [2][q 4 KiB | dO 4 KiB] of this wave
    auto load_ld = [&](RowOps& R, int qt) __attribute__((always_inline)) {
        int qrow = qt * 32 + lr;
        qrow = qrow < len ? qrow : len - 1;
        R.l = lse[(size_t)(qs + qrow) * n_head + head];
        R.d = dsum[(size_t)(qs + qrow) * n_head + head];
    };
    auto issue_tile = [&](int qt, int b) __attribute__((always_inline)) {      // IMG: the 2 x 4 pieces of tile qt -> buffer b
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            int qrow = qt * 32 + 8 * pc + (lane & 7);
            qrow = qrow < len ? qrow : len - 1;
            const uint32_t off = (uint32_t)((((size_t)(qs + qrow) * n_head + head) * HS + (lane >> 3) * 8) * 2);
            glds16_saddr(q, off, img + b * 8192 + pc * 1024);
            glds16_saddr(dout, off, img + b * 8192 + 4096 + pc * 1024);
        }
    };
    auto load_rows = [&](RowOps& R, int qt) __attribute__((always_inline)) {
        int qrow = qt * 32 + lr;
        qrow = qrow < len ? qrow : len - 1;
        const bf16_t* qp = q + ((size_t)(qs + qrow) * n_head + head) * HS + lh * 8;
        const bf16_t* dop = dout + ((size_t)(qs + qrow) * n_head + head) * HS + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            R.qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
            R.dof[ks] = *reinterpret_cast<const bf16x8*>(dop + ks * 16);
        }
        load_ld(R, qt);
    };
    RowOps cur, nxt;
    if constexpr (IMG) {
        // (the compiler does not count the asm DMA: the K / V fragments are used here once, so that no load of its own is pending when
        // the loop's counted waits begin — see attn_bwd_dq_group_kernel)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
        load_ld(cur, kt);
        issue_tile(kt, 0);
    } else {
        load_rows(cur, kt);
    }
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    for (int qt = kt; qt < n_qt; ++qt) {
        const int q0 = qt * 32;
        const char* tile = img + ((qt - kt) & 1) * 8192;
        if constexpr (IMG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // tile qt (and its lse / D) has landed; nothing else is in flight
            if (qt + 1 < n_qt) {                                         // wave-uniform
                load_ld(nxt, qt + 1);
                issue_tile(qt + 1, (qt + 1 - kt) & 1);                   // (the other buffer's last reads fed iteration qt - 1's MFMAs)
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const char* rp = tile + (lr >> 3) * 1024 + ((2 * ks + lh) * 8 + (lr & 7)) * 16;
                cur.qf[ks] = *reinterpret_cast<const bf16x8*>(rp);
                cur.dof[ks] = *reinterpret_cast<const bf16x8*>(rp + 4096);
            }
        }
        // transposed pieces of THIS tile (A operands of the dV / dK products), requested before the S / dP products
        union TP { bf16x8 v; uint2 h[2]; };
        TP af[2][DT], bfr[2][DT];
        auto load_tr = [&](int s2) __attribute__((always_inline)) {     // one coalesced 16-byte load per lane and fragment
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const size_t fo = tfrag_lane<HS>(head, n_pad, tile0 + qt, s2, dt, lane);
                af[s2][dt].v = *reinterpret_cast<const bf16x8*>(doT + fo);
                bfr[s2][dt].v = *reinterpret_cast<const bf16x8*>(qT + fo);
            }
        };
        auto load_tr_img = [&](int s2) __attribute__((always_inline)) {  // lane 4 qq + p of a 16-lane group: row qq, columns 4 p ..; receives column i
            const int g16 = lane >> 4, qq = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int c = 4 * dt + 2 * (g16 & 1) + (p >> 1), r = 4 * (g16 >> 1) + qq;      // chunk, row inside the 8-row piece
                const char* rp = tile + (2 * s2) * 1024 + (c * 8 + r) * 16 + 8 * (p & 1);
                const s16x4 qa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)rp);
                const s16x4 qb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rp + 1024));
                const s16x4 da = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rp + 4096));
                const s16x4 db = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rp + 4096 + 1024));
                bfr[s2][dt].v = bf16x8{qa[0], qa[1], qa[2], qa[3], qb[0], qb[1], qb[2], qb[3]};
                af[s2][dt].v = bf16x8{da[0], da[1], da[2], da[3], db[0], db[1], db[2], db[3]};
            }
        };
        if (IMG) load_tr_img(0);
        if (PIPE) load_tr(0);
        if (PIPE && qt + 1 < n_qt) load_rows(nxt, qt + 1);       // wave-uniform
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.qf[ks], kf[ks], s, 0, 0, 0);       // rows q, cols key
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.dof[ks], vf[ks], dp, 0, 0, 0);
        }
        if (PIPE) load_tr(1);                            // lands under the exps below
        if (IMG) load_tr_img(1);
        // P and dS in the accumulator layout: row (q) = (r&3) + 8*(r>>2) + 4*lh, col (key) = lr
        const int key_abs = key0 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrel = (r & 3) + 8 * (r >> 2) + 4 * lh;          // row of the tile: lane qrel (either half) holds its lse / D
            const int qa = q0 + qrel;
            const bool ok = qa < len && key_abs < len && key_abs <= qa;
            const float l = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(qrel << 2, __builtin_bit_cast(int, cur.l)));
            const float dd = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(qrel << 2, __builtin_bit_cast(int, cur.d)));
            const float p = ok ? __expf(s[r] * scale - l) : 0.f;
            s[r] = p;
            dp[r] = p * (dp[r] - dd);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            union { bf16x8 v; uint32_t u[4]; } pf, dsf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf.u[j] = pack2bf(s[8 * s2 + 2 * j], s[8 * s2 + 2 * j + 1]);
                dsf.u[j] = pack2bf(dp[8 * s2 + 2 * j], dp[8 * s2 + 2 * j + 1]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (!PIPE && !IMG) {                       // hs 128: one fragment pair at a time (8 registers live)
                    const size_t fo = tfrag_lane<HS>(head, n_pad, tile0 + qt, s2, dt, lane);
                    af[s2][dt].v = *reinterpret_cast<const bf16x8*>(doT + fo);
                    bfr[s2][dt].v = *reinterpret_cast<const bf16x8*>(qT + fo);
                }
                dvT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][dt].v, pf.v, dvT[dt], 0, 0, 0);    // [d][key]
                dkT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[s2][dt].v, dsf.v, dkT[dt], 0, 0, 0);
            }
        }
        if (qt + 1 < n_qt) {
            if (IMG) { cur.l = nxt.l; cur.d = nxt.d; }
            else if (PIPE) cur = nxt;
            else load_rows(cur, qt + 1);
        }
    }
    if constexpr (IMG) __syncthreads();           // the reduction below reuses the waves' images
    // ---- sum the heads of the group (LDS), then write dK (scaled) and dV: lane col = key, rows = d.  The waves pass
    // through LDS FOUR at a time (64 KiB at hs 64 instead of 128: two blocks per CU, twice the waves to hide the loop's
    // global-load latency); the sum still runs over the heads in index order (w0 + w1 + ... from zero).
    constexpr int PER_WAVE = 2 * DT * 16 * 64;                 // floats a wave contributes
    // values per thread (8 at hs 64 with 8 heads per group); QPKT 1 = fewer than four heads per group: a single pass,
    // whose totals need not outlive the loop — NV 1 and a strided walk instead of 128 live registers
    constexpr int NV = QPKT == 1 ? 1 : PER_WAVE / (64 * QPKT);
    float tot[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) tot[i] = 0.f;
    const int nw = q_per_kv;
    for (int w0 = 0; w0 < nw; w0 += 4) {
        if (w0) __syncthreads();                                // the previous four have been read
        if (wave >= w0 && wave < w0 + 4) {
            float* mine = red + (size_t)(wave - w0) * PER_WAVE;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    mine[((0 * DT + dt) * 16 + r) * 64 + lane] = dkT[dt][r];
                    mine[((1 * DT + dt) * 16 + r) * 64 + lane] = dvT[dt][r];
                }
        }
        __syncthreads();
        const int cnt = min(4, nw - w0);
        if (QPKT == 1) break;                                   // nw <= 3: finished below straight from LDS
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int it = threadIdx.x + i * (int)blockDim.x;
            if (it < PER_WAVE)
                for (int w = 0; w < cnt; ++w) tot[i] += red[(size_t)w * PER_WAVE + it];
        }
    }
    const int n_it = QPKT == 1 ? (PER_WAVE + (int)blockDim.x - 1) / (int)blockDim.x : NV;
    for (int i = 0; i < n_it; ++i) {
        const int it = threadIdx.x + i * (int)blockDim.x;
        if (it >= PER_WAVE) continue;
        float sum;
        if (QPKT == 1) {
            sum = 0.f;
            for (int w = 0; w < nw; ++w) sum += red[(size_t)w * PER_WAVE + it];
        } else {
            sum = tot[i < NV ? i : 0];
        }
        const int ln = it & 63, r = (it >> 6) & 15, dt = (it >> 10) % DT, which = it / (DT * 16 * 64);
        const int key = key0 + (ln & 31);
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        if (key < len) {
            bf16_t* dst = (which == 0 ? dk : dv) + ((size_t)(qs + key) * n_groups + g) * HS + d;
            *dst = f2bf(which == 0 ? sum * scale : sum);
        }
    }
}

template <int HS>
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ kT, const float* __restrict__ lse,
    const float* __restrict__ dsum, const int32_t* __restrict__ q_start, const int32_t* __restrict__ q_len,
    const int32_t* __restrict__ pad_start, bf16_t* __restrict__ dq, int n_head, int n_groups, int n_pad, float scale, int nt,
    int n_seq) {
    constexpr int KS = HS / 16, DT = HS / 32;
    // as the dkdv kernel: a (head, sequence) pair's query tiles share one XCD (its K^T copy and K / V rows stay in that L2);
    // the long tiles (large qt: most key tiles) first
    int seq, head, qt;
    {
        const int np = n_head * n_seq, b = blockIdx.x, j = b >> 3, pair = (b & 7) + 8 * (j / nt);
        if (pair >= np) return;
        qt = nt - 1 - j % nt; head = pair % n_head; seq = pair / n_head;
    }
    const int len = q_len[seq];
    if (qt * 32 >= len) return;
    const int qs = q_start[seq], ps = pad_start[seq];
    const int g = head / (n_head / n_groups);
    const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
    const int q0 = qt * 32;
    int qrow = q0 + lr;
    const bool q_ok = qrow < len;
    qrow = q_ok ? qrow : len - 1;
    bf16x8 qf[KS], dof[KS];
    {
        const bf16_t* qp = q + ((size_t)(qs + qrow) * n_head + head) * HS + lh * 8;
        const bf16_t* dop = dout + ((size_t)(qs + qrow) * n_head + head) * HS + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dop + ks * 16);
        }
    }
    const float l = lse[(size_t)(qs + qrow) * n_head + head];
    const float dd = dsum[(size_t)(qs + qrow) * n_head + head];
    const int q_abs = q0 + lr;
    f32x16 dqT[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqT[dt][r] = 0.f;
    const int tile0 = ps >> 5;
    for (int kt = 0; kt <= qt; ++kt) {
        const int key0 = kt * 32;
        int key = key0 + lr;
        key = key < len ? key : len - 1;
        const bf16_t* kp = k + ((size_t)(qs + key) * n_groups + g) * HS + lh * 8;
        const bf16_t* vp = v + ((size_t)(qs + key) * n_groups + g) * HS + lh * 8;
        f32x16 st, dpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vp + ks * 16);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st, 0, 0, 0);        // rows key, cols q
            dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dpt, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ka = key0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool ok = q_ok && ka < len && ka <= q_abs;
            const float p = ok ? __expf(st[r] * scale - l) : 0.f;
            st[r] = p * (dpt[r] - dd);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            union { bf16x8 v; uint32_t u[4]; } dsf;
#pragma unroll
            for (int j = 0; j < 4; ++j) dsf.u[j] = pack2bf(st[8 * s2 + 2 * j], st[8 * s2 + 2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(kT + tfrag_lane<HS>(g, n_pad, tile0 + kt, s2, dt, lane));
                dqT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dsf.v, dqT[dt], 0, 0, 0);   // [d][q]
            }
        }
    }
    if (q_ok) {
        bf16_t* dst = dq + ((size_t)(qs + q0 + lr) * n_head + head) * HS;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = dt * 32 + 8 * rg + 4 * lh;
                uint2 pk = make_uint2(pack2bf(dqT[dt][rg * 4 + 0] * scale, dqT[dt][rg * 4 + 1] * scale),
                                      pack2bf(dqT[dt][rg * 4 + 2] * scale, dqT[dt][rg * 4 + 3] * scale));
                *reinterpret_cast<uint2*>(dst + d) = pk;
            }
    }
}

// Round 4: the dq kernel with one BLOCK per (query tile, group, sequence) and one wave per query head of the group.  The K, V and
// K^T fragments of a key tile are the same for every head of the group: the single-wave kernel above fetches them once per head
// (12 KiB per tile pair from L2: 2.1 GB per call on the packed micro-step, the kernel's bound), here each 1-KiB fragment goes
// HBM/L2 -> LDS once per block by LDS-DMA, in fragment order (lane l's 16 bytes at l * 16: conflict-free ds_read_b128), double
// buffered one key tile ahead.  The arithmetic per (head, query tile) is the single-wave kernel's, product for product.
template <int HS>
__global__ __launch_bounds__(512) void attn_bwd_dq_group_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ kT, const float* __restrict__ lse,
    const float* __restrict__ dsum, const int32_t* __restrict__ q_start, const int32_t* __restrict__ q_len,
    const int32_t* __restrict__ pad_start, bf16_t* __restrict__ dq, int n_head, int n_groups, int n_pad, float scale, int nt,
    int n_seq) {
    constexpr int KS = HS / 16, DT = HS / 32, NF = 2 * KS + 2 * DT;      // 1-KiB fragments per key tile: K, V (KS each), K^T (2 DT)
    extern __shared__ __attribute__((aligned(16))) char dq_smem[];       // [2][NF][1 KiB]
    int seq, g, qt;
    {
        // as the dkdv kernel: a (group, sequence) pair's query tiles share one XCD; the long tiles (large qt) first
        const int np = n_groups * n_seq, b = blockIdx.x, j = b >> 3, pair = (b & 7) + 8 * (j / nt);
        if (pair >= np) return;
        qt = nt - 1 - j % nt; g = pair % n_groups; seq = pair / n_groups;
    }
    const int len = q_len[seq];
    if (qt * 32 >= len) return;                                          // block-uniform, in front of every barrier
    const int qs = q_start[seq], ps = pad_start[seq];
    const int qpk = n_head / n_groups;                                   // = waves of the block
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 31, lh = lane >> 5;
    const int head = g * qpk + wave;
    const int q0 = qt * 32;
    int qrow = q0 + lr;
    const bool q_ok = qrow < len;
    qrow = q_ok ? qrow : len - 1;
    bf16x8 qf[KS], dof[KS];
    {
        const bf16_t* qp = q + ((size_t)(qs + qrow) * n_head + head) * HS + lh * 8;
        const bf16_t* dop = dout + ((size_t)(qs + qrow) * n_head + head) * HS + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dop + ks * 16);
        }
    }
    const float l = lse[(size_t)(qs + qrow) * n_head + head];
    const float dd = dsum[(size_t)(qs + qrow) * n_head + head];
    const int q_abs = q0 + lr;
    f32x16 dqT[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqT[dt][r] = 0.f;
    const int tile0 = ps >> 5;
    // fragment f of key tile kt -> buffer b: K ks (f < KS), V ks (f < 2 KS), K^T (s2, dt); wave w moves f = w, w + qpk, ...
    auto issue = [&](int kt, int b) __attribute__((always_inline)) {
        int key = kt * 32 + lr;
        key = key < len ? key : len - 1;
        // (the asm form: behind the builtin the compiler waits vmcnt(0) in front of every LDS read that may alias the destination,
        // i.e. for the NEXT tile's fragments before this tile's products; 32-bit byte offsets: the host checks the tensor sizes)
        const uint32_t row_off = (uint32_t)((((size_t)(qs + key) * n_groups + g) * HS + lh * 8) * 2);
        for (int f = wave; f < NF; f += qpk) {
            char* dst = dq_smem + (b * NF + f) * 1024;
            if (f < KS) glds16_saddr(k, row_off + (uint32_t)f * 32u, dst);
            else if (f < 2 * KS) glds16_saddr(v, row_off + (uint32_t)(f - KS) * 32u, dst);
            else glds16_saddr(kT, (uint32_t)(tfrag_lane<HS>(g, n_pad, tile0 + kt, (f - 2 * KS) / DT, (f - 2 * KS) % DT, lane) * 2), dst);
        }
    };
    // the compiler cannot see the asm DMA in its vmcnt bookkeeping: left alone it waits for the loads above at their first use INSIDE
    // the loop, with counts that also wait out the next tile's DMA.  Use them here, once: no compiler-visible load is pending in the loop.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));
    asm volatile("" ::"v"(l), "v"(dd));
    issue(0, 0);
    for (int kt = 0; kt <= qt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // key tile kt is in LDS for every wave; every wave is done with the other buffer
        if (kt < qt) issue(kt + 1, (kt + 1) & 1);
        const char* buf = dq_smem + (kt & 1) * NF * 1024 + lane * 16;
        const int key0 = kt * 32;
        f32x16 st, dpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(buf + ks * 1024);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(buf + (KS + ks) * 1024);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st, 0, 0, 0);        // rows key, cols q
            dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dpt, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ka = key0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool ok = q_ok && ka < len && ka <= q_abs;
            const float p = ok ? __expf(st[r] * scale - l) : 0.f;
            st[r] = p * (dpt[r] - dd);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            union { bf16x8 v; uint32_t u[4]; } dsf;
#pragma unroll
            for (int j = 0; j < 4; ++j) dsf.u[j] = pack2bf(st[8 * s2 + 2 * j], st[8 * s2 + 2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(buf + (2 * KS + s2 * DT + dt) * 1024);
                dqT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dsf.v, dqT[dt], 0, 0, 0);   // [d][q]
            }
        }
    }
    if (q_ok) {
        bf16_t* dst = dq + ((size_t)(qs + q0 + lr) * n_head + head) * HS;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = dt * 32 + 8 * rg + 4 * lh;
                uint2 pk = make_uint2(pack2bf(dqT[dt][rg * 4 + 0] * scale, dqT[dt][rg * 4 + 1] * scale),
                                      pack2bf(dqT[dt][rg * 4 + 2] * scale, dqT[dt][rg * 4 + 3] * scale));
                *reinterpret_cast<uint2*>(dst + d) = pk;
            }
    }
}

}  // namespace

extern "C" int dh_transpose_pad_bf16(const dh_bf16* src, dh_bf16* dst, const int32_t* tok_seq, const int32_t* q_start,
                                     const int32_t* pad_start, int n_tok, int heads, int hs, int n_pad, void* stream) {
    DH_CHECK(src && dst && tok_seq && q_start && pad_start && hs <= 128 && n_pad % 32 == 0, "dh_transpose_pad_bf16: bad argument");
    if (n_tok <= 0) return 0;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(cdiv(n_tok, 32), heads), dim3(256), 0, (hipStream_t)stream, src, dst, tok_seq,
                       q_start, pad_start, n_tok, heads, hs, n_pad);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_transpose_frag_bf16(const dh_bf16* src, dh_bf16* dst, const int32_t* pad_tok, int heads, int hs, int n_pad,
                                      void* stream) {
    DH_CHECK(src && dst && pad_tok && (hs == 64 || hs == 128) && heads > 0 && n_pad >= 0 && n_pad % 32 == 0, "dh_transpose_frag_bf16: bad argument");
    DH_CHECK(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "dh_transpose_frag_bf16: src / dst must be 16-byte aligned");
    if (n_pad == 0) return 0;
    if (hs == 64) hipLaunchKernelGGL((transpose_frag_kernel<64>), dim3(n_pad / 32, heads), dim3(256), 0, (hipStream_t)stream, src, dst, pad_tok, heads, n_pad);
    else hipLaunchKernelGGL((transpose_frag_kernel<128>), dim3(n_pad / 32, heads), dim3(256), 0, (hipStream_t)stream, src, dst, pad_tok, heads, n_pad);
    DH_LAUNCH_CHECK();
    return 0;
}

// which transposed copies dh_attn_bwd_bf16 reads for this shape: bit 0 = qT and doT, bit 1 = kT (the rest may be passed as null)
extern "C" int dh_attn_bwd_transposes(int n_head, int n_groups, int hs, int n_pad) {
    (void)n_groups;
    const bool dkdv_img = g_attn_bwd_dkdv_img && hs == 64 && (size_t)n_pad * n_head * hs * 2 < (1ull << 32);
    return (dkdv_img ? 0 : 1) | 2;
}

extern "C" int dh_attn_bwd_bf16(const dh_bf16* q, const dh_bf16* k, const dh_bf16* v, const dh_bf16* dout,
                                const dh_bf16* qT, const dh_bf16* doT, const dh_bf16* kT, const float* lse,
                                const float* dsum, const int32_t* q_start, const int32_t* q_len, const int32_t* pad_start,
                                dh_bf16* dq, dh_bf16* dk, dh_bf16* dv, int n_seq, int max_q_len, int n_head, int n_groups,
                                int hs, int n_pad, void* stream) {
    DH_CHECK(q && k && v && dout && kT && lse && dsum && dq && dk && dv, "dh_attn_bwd_bf16: null argument");
    DH_CHECK((qT && doT) || !(dh_attn_bwd_transposes(n_head, n_groups, hs, n_pad) & 1), "dh_attn_bwd_bf16: this shape reads qT / doT (dh_attn_bwd_transposes)");
    DH_CHECK(hs == 64 || hs == 128, "dh_attn_bwd_bf16: head_size %d unsupported", hs);
    DH_CHECK(n_groups > 0 && n_head % n_groups == 0 && n_head / n_groups <= 8, "dh_attn_bwd_bf16: at most 8 query heads per group");
    DH_CHECK(n_pad % 32 == 0, "dh_attn_bwd_bf16: n_pad must be a multiple of 32");
    if (n_seq <= 0 || max_q_len <= 0) return 0;
    const float scale = 1.0f / sqrtf((float)hs);
    const int nt = cdiv(max_q_len, 32), qpk = n_head / n_groups;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(qpk < 4 ? qpk : 4) * 2 * (hs / 32) * 16 * 64 * sizeof(float);   // four waves at a time pass through LDS
    const int np_kv = n_groups * n_seq, np_q = n_head * n_seq;
    const int grid_kv = np_kv >= 8 ? 8 * cdiv(np_kv, 8) * nt : np_kv * nt, grid_q = 8 * cdiv(np_q, 8) * nt;
    const int grid_qg = 8 * cdiv(np_kv, 8) * nt;        // the group form of the dq kernel: (group, sequence) pairs
    // the image form of the dkdv kernel (hs 64): 16 KiB per wave, q / dO addressed through 32-bit byte offsets
    const size_t lds_img = std::max(lds, (size_t)qpk * 16384);
    const bool dkdv_img = g_attn_bwd_dkdv_img && hs == 64 && (size_t)n_pad * n_head * hs * 2 < (1ull << 32);
    // its LDS-DMA addresses K / V / K^T through 32-bit byte offsets
    const bool dq_group = g_attn_bwd_dq_group && (size_t)n_pad * n_groups * hs * 2 < (1ull << 32);
    if (hs == 64) {
#define DKDV(QT, IM) do { DH_MAX_LDS_ONCE((attn_bwd_dkdv_kernel<64, QT, IM>), 160 * 1024);                                       \
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64, QT, IM>), dim3(grid_kv), dim3(64 * qpk), IM ? lds_img : lds, s, q, k, v, dout, qT, doT, lse, dsum,  \
                           q_start, q_len, pad_start, dk, dv, n_head, n_groups, n_pad, scale, nt, n_seq); } while (0)
        if (dkdv_img) { if (qpk == 8) DKDV(8, true); else if (qpk >= 4) DKDV(4, true); else DKDV(1, true); }
        else { if (qpk == 8) DKDV(8, false); else if (qpk >= 4) DKDV(4, false); else DKDV(1, false); }
#undef DKDV
        if (dq_group)
            hipLaunchKernelGGL((attn_bwd_dq_group_kernel<64>), dim3(grid_qg), dim3(64 * qpk), 2 * (2 * 4 + 2 * 2) * 1024, s, q, k, v, dout, kT, lse, dsum,
                               q_start, q_len, pad_start, dq, n_head, n_groups, n_pad, scale, nt, n_seq);
        else
            hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), dim3(grid_q), dim3(64), 0, s, q, k, v, dout, kT, lse, dsum,
                               q_start, q_len, pad_start, dq, n_head, n_groups, n_pad, scale, nt, n_seq);
    } else {
#define DKDV(QT) do { DH_MAX_LDS_ONCE((attn_bwd_dkdv_kernel<128, QT>), 160 * 1024);                                       \
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, QT>), dim3(grid_kv), dim3(64 * qpk), lds, s, q, k, v, dout, qT, doT, lse, dsum,  \
                           q_start, q_len, pad_start, dk, dv, n_head, n_groups, n_pad, scale, nt, n_seq); } while (0)
        if (qpk == 8) DKDV(8); else if (qpk >= 4) DKDV(4); else DKDV(1);
#undef DKDV
        if (dq_group)
            hipLaunchKernelGGL((attn_bwd_dq_group_kernel<128>), dim3(grid_qg), dim3(64 * qpk), 2 * (2 * 8 + 2 * 4) * 1024, s, q, k, v, dout, kT, lse, dsum,
                               q_start, q_len, pad_start, dq, n_head, n_groups, n_pad, scale, nt, n_seq);
        else
            hipLaunchKernelGGL((attn_bwd_dq_kernel<128>), dim3(grid_q), dim3(64), 0, s, q, k, v, dout, kT, lse, dsum,
                               q_start, q_len, pad_start, dq, n_head, n_groups, n_pad, scale, nt, n_seq);
    }
    DH_LAUNCH_CHECK();
    return 0;
}
