// Decode-phase GEMMs for MORE than 64 rows (several batches decoded jointly): a tiled kernel with both
// operands through LDS, bit-identical to the streaming kernels it replaces.
//
// Above ~64 rows the streaming kernels (gemm_skinny.hip rows kernel, gemm_mid.hip) stop being bound by
// the weight stream: every wave re-reads the whole x slice from LDS for its 16 W rows (one MFMA per
// 1-KiB fragment read — at 256 rows the LDS port is busy 4x longer than the matrix pipe) and the K-sliced
// partial sums cost M*N*ksplit*4 bytes of stores.  Here a block owns a 128(m) x 128(n) tile, 4 waves of
// 64 x 64 (8 fragment reads feed 16 MFMAs), K in 64-wide stages through a 4-deep LDS ring
// (global_load_lds, source-side XOR swizzle as gemm.hip), and the full K is reduced in the block.
//
// Bit-compatibility: the streaming kernels define an output as chains of v_mfma_f32_16x16x32_bf16 over
// CONTIGUOUS K segments (k ascending, fp32 accumulator from zero), the chains added in segment order
// (rows kernel: K-slices of 8 or 16 k-steps, added by the consumer kernels; gemm_mid.hip: KQ K-parts,
// added in its LDS reduction).  This kernel runs the same chains — same MFMA instruction and operand
// roles, `cur` restarted at every segment boundary and folded into `tot` in order — so a row's result is
// the same whichever kernel computed it (tests: test_linear_partial_wide_rows, test_joint_decode_*).
//   MODE 0 : fp32 output [M][N] of x·[w; w_ext]^T (the rows kernel's partials combined in the family's order: adjacent
//            slices added in PAIRS, the pair sums in index order — three accumulator sets)
//   MODE 4 : split-K over blocks, fp32 output [ceil(slices/2)][M][N]: block (tile, y) walks the TWO K-slices 2y, 2y+1 and
//            stores their sum — the pair sums of that same order, half the partial-sum bytes of the rows kernel and whole
//            tiles per block (round 3: the decode GEMMs above 128 rows)
//   MODE 1 : SwiGLU pair, bf16 output; MODE 2 : adapter scale/bias; MODE 3 : plain (+ residual)
//   grid   : m-tiles fastest, then n-tiles (x K-slice pairs in MODE 4)
#include "common.h"
#include "gemm.h"

int g_dt_stages = 0;   // 0: by grid size, else 2 or 4 (dh_set_tuning key 8)
int g_dt_wide = 0;     // SwiGLU tile of 128 pairs x 128 rows on 8 waves: 0 = from 256 rows on, 1 = always, -1 = never (dh_set_tuning key 15)

namespace {

constexpr int TB = 128;                 // rows of x per block tile; rows of W per block: 64 * WN (WN = waves along n)
constexpr int BKD = 64;                 // K per stage (two k-steps of 32)
template <int WN> constexpr int stage_bytes() { return (WN * 64 + TB) * BKD * 2; }   // W tile 8 KiB * WN + x tile 16 KiB

__device__ __forceinline__ int swz7(int row) { return (row >> 1) & 7; }

struct DtArgs {
    const bf16_t* x;
    const bf16_t* w;
    const bf16_t* w2;      // MODE 0: w_ext (rows n_main..), MODE 1: fc_2
    void* y;               // MODE 0: float [M][N], else bf16 [M][N]
    const bf16_t* vec_a;
    const bf16_t* vec_b;
    const bf16_t* resid;
    int M, N, K, n_main, seg;   // seg: k-steps per chain segment (even)
    int split_kt;               // MODE 4: 64-wide stages per block (two segments); else 0
    int wt;                     // MODE 4: write-through (sc1) stores of the pair sums
};

// NSTAGE 4: one block per CU, 3 stages in flight; NSTAGE 2: two blocks per CU.  WN 2: 4 waves, 128 W rows per block;
// WN 4 (SwiGLU at >= 256 rows): 8 waves, 256 W rows (128 fc_1/fc_2 pairs) per block, 3 stages of 48 KiB — half the
// blocks, each moving 1.5x the bytes per stage for 2x the MFMA work (a block's time is set by its stage count, not by
// the bytes: ~1 us per stage of LDS-DMA latency), and 640 rows become one round of 220 blocks instead of 440.
template <int MODE, int NSTAGE, int WN = 2>
__global__ __launch_bounds__(WN * 128) void gemm_dt_kernel(DtArgs a) {
    constexpr int STAGE = stage_bytes<WN>();
    constexpr int ASZ = WN * 64 * BKD * 2;                        // bytes of the W tile of a stage
    constexpr int NB = 16 / (2 * WN);                             // x groups (8 rows x 128 B) staged per wave
    constexpr int PER = 4 + NB;                                   // LDS-DMA requests per wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NSTAGE stages
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, kg = lane >> 4;
    const int m_tiles = (a.M + TB - 1) / TB;
    // XCD-aware order (blocks go round-robin over the 8 XCDs, each with its own L2): an XCD walks a CONTIGUOUS range of
    // the m-fastest tile order, so the m-tiles that share a W tile meet in one L2 instead of fetching it over the fabric
    // once per XCD (at 640 rows that was 5 x 46 MB per SwiGLU launch = 5.2 TB/s: the limit of the kernel).
    int tile, ky = 0;       // ky: MODE 4's K-slice pair (grid.y); the XCD walk runs over the whole 2-D grid, tiles fastest
    {
        const int nwg = gridDim.x * gridDim.y, bid = blockIdx.x + gridDim.x * blockIdx.y, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if (MODE == 4) { ky = tile / (int)gridDim.x; tile -= ky * (int)gridDim.x; }
    }
    const int tm = tile % m_tiles, tn = tile / m_tiles;
    const int m0 = tm * TB;
    const int n0 = MODE == 1 ? tn * (WN * 32) : tn * (WN * 64);     // first output column of the block

    // ---- staging sources: 8 WN + 16 one-KiB groups (8 rows of 128 B) per stage, 4 + NB per wave
    const bf16_t* srcA[4];
    const bf16_t* srcB[NB];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = wave * 4 + j;
        const int row = R * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz7(row);
        const bf16_t* base = a.w;
        int n;
        if (MODE == 1) {
            // a wave's 64 LDS rows: 32 rows of fc_1 then the same 32 rows of fc_2
            const int half = (row >> 5) & 1;
            n = n0 + (row >> 6) * 32 + (row & 31);
            n = n < a.N ? n : a.N - 1;
            base = half ? a.w2 : a.w;
        } else {
            n = n0 + row;
            n = n < a.N ? n : a.N - 1;
            if ((MODE == 0 || MODE == 4) && n >= a.n_main) { base = a.w2; n -= a.n_main; }
        }
        srcA[j] = base + (size_t)n * a.K + chunk * 8 + (MODE == 4 ? (size_t)ky * a.split_kt * BKD : 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int R = wave * NB + j;
        const int row = R * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz7(row);
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        srcB[j] = a.x + (size_t)m * a.K + chunk * 8 + (MODE == 4 ? (size_t)ky * a.split_kt * BKD : 0);
    }
    auto stage = [&](int kt) __attribute__((always_inline)) {
        char* sA = smem + (kt % NSTAGE) * STAGE;
        char* sB = sA + ASZ;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(srcA[j] + kt * BKD, sA + (wave * 4 + j) * 1024);
#pragma unroll
        for (int j = 0; j < NB; ++j) glds16(srcB[j] + kt * BKD, sB + (wave * NB + j) * 1024);
    };
    // stage kt+1 landed (this wave's share) while up to `ahead` younger stages stay in flight
    auto wait_next = [&](int ahead) __attribute__((always_inline)) {
        if (NSTAGE >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        else if (NSTAGE >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x4 cur[4][4], tot[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cur[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int sw = swz7(frow);
    const int offA = (wn * 64 + frow) * 128, offB = (wm * 64 + frow) * 128;
    const int nk = MODE == 4 ? min(a.split_kt, a.K / BKD - ky * a.split_kt) : a.K / BKD;
    const int seg_kt = a.seg / 2;       // stages per chain segment
    f32x4 pairacc[MODE == 0 ? 4 : 1][MODE == 0 ? 4 : 1];   // MODE 0: the open pair's first slice
    int seg_idx = 0;

    auto fold_segment = [&](bool last) __attribute__((always_inline)) {     // end of a chain segment: fold it in, in order
        if (MODE == 0) {
            // the partial-sum family's order: (s0 + s1) + (s2 + s3) + ... — slices in adjacent pairs, pair sums in
            // index order; an unpaired last slice is added on its own
            const bool second = seg_idx & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (second) tot[i][j] += pairacc[i][j] + cur[i][j];
                    else if (last) tot[i][j] += cur[i][j];
                    else pairacc[i][j] = cur[i][j];
                    cur[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            ++seg_idx;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tot[i][j] += cur[i][j];
                    cur[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
    };
    int in_seg = 0;
    if constexpr (WN == 4 && NSTAGE == 3) {
    // ---- 8-wave tile, round 4: fragment reads one PHASE ahead of their MFMAs ("LDS-DMA fill and MFMA add up
    // instead of overlapping" — every iteration began with its 16 ds_read_b128 and their latency in front of the matrix pipe, both
    // waves of a SIMD in step).  An iteration (one 64-deep stage kt) is two phases around ONE barrier:
    //   phase 1: read F1 <- stage kt (second k-step) | x pieces of stage kt+2 | 16 MFMAs on F0 | lgkmcnt(0), vmcnt: stage kt+1 landed | barrier
    //   phase 2: read F0 <- stage kt+1 (first k-step) | W pieces of stage kt+3 -> the slot of stage kt | 16 MFMAs on F1 | segment fold
    // Slot of stage kt is free behind the barrier of iteration kt: its F0 reads were consumed in phase 1, its F1 reads waited for
    // (lgkmcnt(0)) in front of that barrier, by every wave.  Stage kt+1 is read (F0, phase 2) behind the same barrier, in front of
    // which every wave waited for its own pieces of it.  Same MFMAs in the same order per accumulator: same bits.
    //   pieces in flight at the wait of iteration kt: W (4) of stage kt+2 [phase 2 of kt-1] + x (NB) of stage kt+2 [phase 1 of kt] = PER
    auto stage_w = [&](int kt) __attribute__((always_inline)) {
        char* sA = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(srcA[j] + kt * BKD, sA + (wave * 4 + j) * 1024);
    };
    auto stage_x = [&](int kt) __attribute__((always_inline)) {
        char* sB = smem + (kt % NSTAGE) * STAGE + ASZ;
#pragma unroll
        for (int j = 0; j < NB; ++j) glds16(srcB[j] + kt * BKD, sB + (wave * NB + j) * 1024);
    };
    auto read_frags = [&](int kt, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) __attribute__((always_inline)) {
        const char* sA = smem + (kt % NSTAGE) * STAGE;
        const char* sB = sA + ASZ;
        const int co = ((ks * 4 + kg) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *reinterpret_cast<const bf16x8*>(sA + offA + i * 2048 + co);
            fb[i] = *reinterpret_cast<const bf16x8*>(sB + offB + i * 2048 + co);
        }
    };
    auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], cur[i][j], 0, 0, 0);
    };
    // prologue: stages 0 and 1 whole, W of stage 2; stage 0 waited for (stage 1 + the W pieces of 2 stay in flight)
    stage(0);
    if (1 < nk) stage(1);
    if (2 < nk) stage_w(2);
    if (2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER + 4) : "memory");
    else if (1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
    read_frags(0, 0, fa0, fb0);
    for (int kt = 0; kt < nk; ++kt) {
        // ---- phase 1
        read_frags(kt, 1, fa1, fb1);
        if (kt + 2 < nk) stage_x(kt + 2);
        mma(fa0, fb0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 2
        if (kt + 1 < nk) read_frags(kt + 1, 0, fa0, fb0);
        if (kt + 3 < nk) stage_w(kt + 3);
        mma(fa1, fb1);
        if (++in_seg == seg_kt || kt + 1 == nk) {
            in_seg = 0;
            fold_segment(kt + 1 == nk);
        }
    }
    } else {
    // prologue: NSTAGE-1 stages requested, the first one waited for
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
        if (p < nk) stage(p);
    wait_next((nk < NSTAGE - 1 ? nk : NSTAGE - 1) - 1);
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        // ring slot (kt+NSTAGE-1) % NSTAGE was last read in iteration kt-1, which ended with a barrier
        const bool pre = kt + NSTAGE - 1 < nk;
        const char* sA = smem + (kt % NSTAGE) * STAGE;
        const char* sB = sA + ASZ;
        if (pre) stage(kt + NSTAGE - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int co = ((ks * 4 + kg) ^ sw) << 4;
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8*>(sA + offA + i * 2048 + co);
                fb[i] = *reinterpret_cast<const bf16x8*>(sB + offB + i * 2048 + co);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cur[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], cur[i][j], 0, 0, 0);
        }
        if (++in_seg == seg_kt || kt + 1 == nk) {
            in_seg = 0;
            fold_segment(kt + 1 == nk);
        }
        // own share of stage kt+1 landed (later stages may stay in flight), then everybody's
        {
            const int issued = (kt + NSTAGE - 1 < nk ? kt + NSTAGE - 1 : nk - 1);   // youngest stage requested so far
            wait_next(issued - (kt + 1));
        }
        __builtin_amdgcn_s_barrier();
    }
    }

    // ---------------------------------------------------------------- epilogue
    // tot[i][j][r]: LDS row of W = wn*64 + i*16 + 4*kg + r ; m = m0 + wm*64 + j*16 + frow
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        if (m >= a.M) continue;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = n0 + wn * 32 + i * 16 + kg * 4;
                if (n >= a.N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gt = rbf(tot[i][j][r]), up = rbf(tot[i + 2][j][r]);
                    o[r] = rbf(silu_fast(gt)) * up;
                }
                *reinterpret_cast<uint2*>((bf16_t*)a.y + (size_t)m * a.N + n) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + i * 16 + kg * 4;
                if (n >= a.N) continue;
                if (MODE == 0 || MODE == 4) {
                    float* yo = (float*)a.y + (MODE == 4 ? (size_t)ky * a.M * a.N : 0);
                    float* dst = yo + (size_t)m * a.N + n;
                    // the partial sums are read next by OTHER blocks, 7 of 8 of them on another XCD (whose L2 is not coherent
                    // with this one): write them through (sc1) so they leave this L2 while the kernel runs instead of as a
                    // 20-27 MB write-back at the kernel boundary (dh_set_tuning key 21: 0 = plain stores)
                    if (MODE == 4 && a.wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(tot[i][j]) : "memory");
                    else *reinterpret_cast<f32x4*>(dst) = tot[i][j];
                    continue;
                }
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = rbf(tot[i][j][r]);
                    if (MODE == 2) o[r] = rbf(bf2f(a.vec_a[n + r]) * rbf(o[r] + bf2f(a.vec_b[n + r])));
                }
                if (MODE == 3 && a.resid != nullptr) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.N + n);
                    const bf16_t* rp = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = bf2f(rp[r]) + o[r];
                }
                *reinterpret_cast<uint2*>((bf16_t*)a.y + (size_t)m * a.N + n) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
        }
    }
}

template <int MODE, int NSTAGE, int WN = 2>
int launch_dt_n(const DtArgs& a, int blocks, hipStream_t s, int ny = 1) {
    constexpr int lds = NSTAGE * stage_bytes<WN>();     // WN 2: 128 KiB / 64 KiB; WN 4: 144 KiB
    DH_MAX_LDS_ONCE((gemm_dt_kernel<MODE, NSTAGE, WN>), lds);
    hipLaunchKernelGGL((gemm_dt_kernel<MODE, NSTAGE, WN>), dim3(blocks, ny), dim3(WN * 128), lds, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

template <int MODE>
int launch_dt(const DtArgs& a, hipStream_t s) {
    const int m_tiles = cdiv(a.M, TB);
    if ((MODE == 1 || MODE == 2) && (g_dt_wide ? g_dt_wide > 0 : a.M >= 256))
        return launch_dt_n<MODE == 2 ? 2 : 1, 3, 4>(a, m_tiles * cdiv(a.N, MODE == 1 ? 128 : 256), s);
    const int n_tiles = cdiv(a.N, MODE == 1 ? 64 : TB);
    const int blocks = m_tiles * n_tiles;
    // more than one block per CU to go round: two co-resident blocks hide each other's waits better than
    // a deeper ring in one block
    const int stages = g_dt_stages ? g_dt_stages : (blocks > 256 ? 2 : 4);
    return stages == 2 ? launch_dt_n<MODE, 2>(a, blocks, s) : launch_dt_n<MODE, 4>(a, blocks, s);
}

}  // namespace

// Measured (tools/tune_dt.py, TinyLlama shapes, us):         rows  256    512   1024   2048
//   SwiGLU   gemm_mid (two passes from 193 rows on)              46      -      -      -
//            this kernel, 4 stages / 2 stages                   36/41  68/48 102/88 198/134
//   QKV'     K-sliced rows kernel | this kernel (chain mode)    16|33  30|33  43|35  83|45
//   mlp'     K-sliced rows kernel | this kernel (chain mode)    27|81  50|81  76|82 150|83
// A block walks K sequentially (about 1 us per 64-wide stage however many stages are in flight), so the
// chain mode costs a constant ~nk us until the m-tiles fill the chip, while K-slicing across blocks scales
// with the rows: the crossover was near 768 rows in round 1; with round 2's K-sliced kernel it lies between 1024 and 2048.
int g_dt_min_rows = 129;       // fused-epilogue decode GEMMs from this many rows on (dh_set_tuning key 6); the lm_head (N = 32000) already
                               // from 65: measured SwiGLU 24 (streaming) vs 32 us at 128 rows, 40 vs 32 at 130; lm_head 49 vs 39 at 96 rows
// the one-launch total of the partial-sum GEMMs from this many rows on (dh_set_tuning key 7).  Round 3: OFF by default (1 << 30) — in the
// family's pair order the kernel needs a third accumulator set (412 registers: one wave per SIMD, one block per CU) and a 2048-row
// decode step takes 15.25 ms against 13.55 with the pair-sum kernel (round 2's two-set chain in slice order: 13.0).  Kept for the ABI.
int g_chain_min_rows = 1 << 30;   //
                               // (8.34 vs 8.53 ms per decode step), this kernel at 2048 (12.9 vs 14.6)

// x·[w; w_ext]^T summed over the K-slices of `kps` k-steps in slice order: fp32 [M][n_main + n_ext]
int dh_chain_tiled(const bf16_t* x, const bf16_t* w, const bf16_t* w_ext, float* y32, int M, int n_main, int n_ext, int K,
                   int kps, hipStream_t s) {
    DtArgs a{x, w, w_ext ? w_ext : w, y32, nullptr, nullptr, nullptr, M, n_main + n_ext, K, n_main, kps, 0, 0};
    return launch_dt<0>(a, s);
}

// The pair sums of the K-slices (kps k-steps each) of x·[w; w_ext]^T: fp32 [ceil(nslices / 2)][M][n_main + n_ext], one
// block per (tile, slice pair).  Tile shape by grid size: 128 x 256 on 8 waves (3 stages of 48 KiB) when that still gives
// every CU a block, else 128 x 128 on 4 waves (dh_set_tuning key 17: 0 auto, 2 / 4 = waves along n).
// Round 3, measured and not kept: the same 128 x 256 pair-sum tile on FOUR waves in the style of gemm256.hip's 4-wave kernel (one wave
// per SIMD with 128 x 64, 3-stage ring, both k-steps of a stage in registers, one barrier per stage, asm MFMAs with pinned
// accumulators; bit-equal): 16.8 / 16.6 / 24.5 us for qkv' / proj' / mlp' at 640 rows against 16.0 / 15.8 / 24.3 here.  A block
// walks only 8 stages; the launch is bound by its 27 MB of fp32 pair sums, the first stage's latency and the launch itself, not
// by the K walk.
int g_pairs_wn = 0;
int g_pairs_wt = 0;   // write-through (sc1) stores of the pair sums: measured neutral (decode 4.74-4.77 ms per token either way), off
int dh_pairs_tiled(const bf16_t* x, const bf16_t* w, const bf16_t* w_ext, float* y32, int M, int n_main, int n_ext, int K,
                   int kps, hipStream_t s) {
    const int N = n_main + n_ext, nslices = cdiv(K / 32, kps), ny = (nslices + 1) / 2;
    DtArgs a{x, w, w_ext ? w_ext : w, y32, nullptr, nullptr, nullptr, M, N, K, n_main, kps, kps, g_pairs_wt};   // kps k-steps = kps/2 stages per slice, two slices per block
    const int m_tiles = cdiv(M, TB);
    const int wide_blocks = m_tiles * cdiv(N, 256) * ny;
    const bool wide = g_pairs_wn ? g_pairs_wn == 4 : wide_blocks >= 160;   // 640 rows: qkv' 220, proj' 180, mlp' 240 blocks of 128 x 256
    if (wide) return launch_dt_n<4, 3, 4>(a, m_tiles * cdiv(N, 256), s, ny);
    const int blocks = m_tiles * cdiv(N, TB);
    return blocks * ny > 256 ? launch_dt_n<4, 2>(a, blocks, s, ny) : launch_dt_n<4, 4>(a, blocks, s, ny);
}

bool dh_linear_dt_ok(const GemmArgs& a, int epilogue, int seg) {
    return a.M >= (epilogue == DH_EPI_ADAPTER ? (g_dt_min_rows < 65 ? g_dt_min_rows : 65) : g_dt_min_rows) && a.M > 64 && a.K % BKD == 0 && seg > 0 && seg % 2 == 0 && (a.K / 32) % seg == 0 && a.N % 4 == 0 &&
           (epilogue == DH_EPI_PLAIN || epilogue == DH_EPI_SWIGLU || epilogue == DH_EPI_ADAPTER);
}

int dh_linear_dt(const GemmArgs& g, int epilogue, int seg, hipStream_t s) {
    DtArgs a{g.x, g.w, g.w2, g.y, g.vec_a, g.vec_b, g.resid, g.M, g.N, g.K, g.N, seg, 0, 0};
    switch (epilogue) {
        case DH_EPI_SWIGLU: return launch_dt<1>(a, s);
        case DH_EPI_ADAPTER: return launch_dt<2>(a, s);
        case DH_EPI_PLAIN: return launch_dt<3>(a, s);
    }
    dh_set_error("dh_linear_dt: unsupported epilogue %d", epilogue);
    return 1;
}
