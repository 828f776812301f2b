// y[M<=32, N] = epilogue(x · W^T) for the decode step: HBM-bound weight streaming.
//
// Every byte of W is needed once, x (<= 32 rows) is L2-resident and shared by all blocks, so W
// goes global -> VGPR directly (no LDS round trip: nothing is reused; cdna_hip_programming.md
// "GEMV / M <= 16 decode weights") and feeds v_mfma_f32_16x16x32_bf16 as the A operand:
//     C^T[16 n][16 m] += W[16 n][32 k] · x[16 m][32 k]^T      (two MFMAs cover m = 0..31)
//   block : 512 threads = 8 waves, one tile of 16 W rows (SWIGLU: 16 rows of fc_1 + 16 of fc_2)
//   wave  : K is dealt round-robin in 32-wide k-steps over the waves, so the 8 waves of a
//           block read 512 contiguous bytes of each row per round; all loads of a chunk of
//           8 k-steps (8 KiB of W per wave) are issued before the first MFMA waits on them
//   reduce: the per-wave fp32 partials meet in LDS (16 KiB); thread t finishes output
//           (n = t & 15, m = t >> 4) with the same rounding points as the tiled kernel
//   grid  : N / 16 blocks (>= 128 for every matrix of the decoder) x 8 waves of loads in flight
#include "common.h"
#include "gemm.h"

namespace {

constexpr int ROWS = 16;     // W rows per block
constexpr int NW = 8;        // waves per block
constexpr int CH = 8;        // k-steps loaded ahead per wave

template <int EPI, bool RESID>
__global__ __launch_bounds__(512, 2) void gemm_skinny_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float part[NW][32][ROWS];   // [wave][m][n]
    __shared__ __attribute__((aligned(16))) float part2[EPI == DH_EPI_SWIGLU ? NW : 1][32][ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * ROWS;
    const int lrow = lane & 15, kg = lane >> 4;          // MFMA operand row / 8-element k group
    // SWIGLU: every wave streams its k-steps of the SAME 16 rows of fc_1 and fc_2, so one pair of
    // x fragments feeds four MFMAs (x is the larger share of the load instructions otherwise)
    constexpr bool SW = EPI == DH_EPI_SWIGLU;
    constexpr int CHK = SW ? CH / 2 : CH;
    const int kw = wave, kstride = NW;
    int n = n0 + lrow;
    n = n < a.N ? n : a.N - 1;
    const bf16_t* wrow = a.w + (size_t)n * a.K + kg * 8;
    const bf16_t* wrow2 = SW ? a.w2 + (size_t)n * a.K + kg * 8 : nullptr;
    int m_lo = lrow, m_hi = 16 + lrow;
    m_lo = m_lo < a.M ? m_lo : a.M - 1;
    m_hi = m_hi < a.M ? m_hi : a.M - 1;
    const bf16_t* xlo = a.x + (size_t)m_lo * a.K + kg * 8;
    const bf16_t* xhi = a.x + (size_t)m_hi * a.K + kg * 8;

    f32x4 acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc2_lo = {0.f, 0.f, 0.f, 0.f}, acc2_hi = {0.f, 0.f, 0.f, 0.f};
    const int nks = a.K / 32;
    for (int ks0 = kw; ks0 < nks; ks0 += kstride * CHK) {
        bf16x8 wf[CHK], wf2[CHK], xl[CHK], xh[CHK];
#pragma unroll
        for (int c = 0; c < CHK; ++c) {
            const int ks = ks0 + c * kstride;
            if (ks < nks) {
                wf[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow + ks * 32));
                if (SW) wf2[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow2 + ks * 32));
                xl[c] = *reinterpret_cast<const bf16x8*>(xlo + ks * 32);
                xh[c] = *reinterpret_cast<const bf16x8*>(xhi + ks * 32);
            }
        }
#pragma unroll
        for (int c = 0; c < CHK; ++c) {
            const int ks = ks0 + c * kstride;
            if (ks < nks) {
                acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xl[c], acc_lo, 0, 0, 0);
                acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xh[c], acc_hi, 0, 0, 0);
                if (SW) {
                    acc2_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[c], xl[c], acc2_lo, 0, 0, 0);
                    acc2_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[c], xh[c], acc2_hi, 0, 0, 0);
                }
            }
        }
    }
    // C layout: col (m) = lane & 15, rows (n) = 4*(lane>>4) + reg  -> 4 consecutive n per lane
    *reinterpret_cast<f32x4*>(&part[wave][lrow][kg * 4]) = acc_lo;
    *reinterpret_cast<f32x4*>(&part[wave][16 + lrow][kg * 4]) = acc_hi;
    if (SW) {
        *reinterpret_cast<f32x4*>(&part2[wave][lrow][kg * 4]) = acc2_lo;
        *reinterpret_cast<f32x4*>(&part2[wave][16 + lrow][kg * 4]) = acc2_hi;
    }
    __syncthreads();

    const int tn = tid & 15, tm = tid >> 4;
    const int nn = n0 + tn;
    if (tm >= a.M || nn >= a.N) return;
    const float* p = &part[0][0][0] + tm * ROWS + tn;
    float o;
    if (EPI == DH_EPI_SWIGLU) {
        float g = 0.f, u = 0.f;
        const float* p2 = &part2[0][0][0] + tm * ROWS + tn;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            g += p[w * 32 * ROWS];
            u += p2[w * 32 * ROWS];
        }
        g = rbf(g);
        u = rbf(u);
        o = rbf(silu_fast(g)) * u;
    } else {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += p[w * 32 * ROWS];
        o = rbf(s);
        if (EPI == DH_EPI_LORA) {
            const int seg = (nn >= a.split0) + (nn >= a.split1);
            const uint4* xa4 = reinterpret_cast<const uint4*>(a.xa + (size_t)tm * a.xa_ld + seg * 16);
            const uint4* lb4 = reinterpret_cast<const uint4*>(a.lora_b + (size_t)nn * 16);
            float l = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 xv = xa4[h], bv = lb4[h];
                const bf16_t* xp = reinterpret_cast<const bf16_t*>(&xv);
                const bf16_t* bp = reinterpret_cast<const bf16_t*>(&bv);
#pragma unroll
                for (int e = 0; e < 8; ++e) l = fmaf(bf2f(xp[e]), bf2f(bp[e]), l);
            }
            o = rbf(o + rbf(rbf(l) * a.lora_scale));
        }
        if (EPI == DH_EPI_ADAPTER) o = rbf(bf2f(a.vec_a[nn]) * rbf(o + bf2f(a.vec_b[nn])));
        if (RESID) o = bf2f(a.resid[(size_t)tm * a.N + nn]) + o;
    }
    a.y[(size_t)tm * a.N + nn] = f2bf(o);
}

// SwiGLU at decode, 32 row pairs per block: one pair of x fragments now feeds EIGHT MFMAs (two 16-row
// tiles of fc_1 and of fc_2), halving the x share of the vector-load traffic that bounds this GEMM
// (the x loads are L2 hits but ride the same load path as the HBM stream: 20.5 -> 13.6 us with them
// removed entirely, tools/tune_decode.py).  K dealt over the 8 waves, partials meet in LDS (64 KiB).
__global__ __launch_bounds__(512, 1) void swiglu_skinny2_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float part2x[];      // [fc 2][wave 8][m 32][n 32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 32;
    const int lrow = lane & 15, kg = lane >> 4;
    int na = n0 + lrow, nb = n0 + 16 + lrow;
    na = na < a.N ? na : a.N - 1;
    nb = nb < a.N ? nb : a.N - 1;
    const bf16_t* w1a = a.w + (size_t)na * a.K + kg * 8;
    const bf16_t* w1b = a.w + (size_t)nb * a.K + kg * 8;
    const bf16_t* w2a = a.w2 + (size_t)na * a.K + kg * 8;
    const bf16_t* w2b = a.w2 + (size_t)nb * a.K + kg * 8;
    int m_lo = lrow, m_hi = 16 + lrow;
    m_lo = m_lo < a.M ? m_lo : a.M - 1;
    m_hi = m_hi < a.M ? m_hi : a.M - 1;
    const bf16_t* xlo = a.x + (size_t)m_lo * a.K + kg * 8;
    const bf16_t* xhi = a.x + (size_t)m_hi * a.K + kg * 8;
    f32x4 acc[2][2][2];                                    // [fc][row tile][m half]
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[f][t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int C2 = 4;
    const int nks = a.K / 32;
    for (int ks0 = wave; ks0 < nks; ks0 += NW * C2) {
        bf16x8 f1a[C2], f1b[C2], f2a[C2], f2b[C2], xl[C2], xh[C2];
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const int ks = ks0 + c * NW;
            if (ks < nks) {
                f1a[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1a + ks * 32));
                f1b[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1b + ks * 32));
                f2a[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w2a + ks * 32));
                f2b[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w2b + ks * 32));
                xl[c] = *reinterpret_cast<const bf16x8*>(xlo + ks * 32);
                xh[c] = *reinterpret_cast<const bf16x8*>(xhi + ks * 32);
            }
        }
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const int ks = ks0 + c * NW;
            if (ks < nks) {
                acc[0][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1a[c], xl[c], acc[0][0][0], 0, 0, 0);
                acc[0][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1a[c], xh[c], acc[0][0][1], 0, 0, 0);
                acc[0][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1b[c], xl[c], acc[0][1][0], 0, 0, 0);
                acc[0][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1b[c], xh[c], acc[0][1][1], 0, 0, 0);
                acc[1][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2a[c], xl[c], acc[1][0][0], 0, 0, 0);
                acc[1][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2a[c], xh[c], acc[1][0][1], 0, 0, 0);
                acc[1][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2b[c], xl[c], acc[1][1][0], 0, 0, 0);
                acc[1][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2b[c], xh[c], acc[1][1][1], 0, 0, 0);
            }
        }
    }
    // C: col (m) = lrow (+16*h), rows (n) = 4*kg + reg (+16*t)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<f32x4*>(&part2x[(((f * NW + wave) * 32) + 16 * h + lrow) * 32 + 16 * t + kg * 4]) = acc[f][t][h];
    __syncthreads();
    for (int it = tid; it < 32 * 32; it += 512) {
        const int tn = it & 31, tm = it >> 5, nn = n0 + tn;
        if (tm >= a.M || nn >= a.N) continue;
        float g = 0.f, u = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            g += part2x[((0 * NW + w) * 32 + tm) * 32 + tn];
            u += part2x[((1 * NW + w) * 32 + tm) * 32 + tn];
        }
        g = rbf(g);
        u = rbf(u);
        a.y[(size_t)tm * a.N + nn] = f2bf(rbf(silu_fast(g)) * u);
    }
}

int g_swiglu2 = 1;

// fp32 partial sums for consumers that finish the epilogue themselves (decode_fused.hip):
//   part[by][m][n] = sum over the k-steps of K-slice `by` of x[m,:] . W'[n,:],  W' = [w ; w_ext]
// grid (N'/16, ksplit): the 8*ksplit waves that share a row tile interleave over K, so small
// matrices (N = d) still put >= 256 blocks on the chip.
__global__ __launch_bounds__(512, 2) void gemm_skinny_partial_kernel(const bf16_t* __restrict__ x,
                                                                   const bf16_t* __restrict__ w,
                                                                   const bf16_t* __restrict__ w_ext,
                                                                   float* __restrict__ y32, int M, int m_total,
                                                                   int n_main, int N, int K) {
    __shared__ __attribute__((aligned(16))) float part[NW][32][ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * ROWS, ksplit = gridDim.y;
    const int lrow = lane & 15, kg = lane >> 4;
    int n = n0 + lrow;
    n = n < N ? n : N - 1;
    const bf16_t* wrow = (n < n_main ? w + (size_t)n * K : w_ext + (size_t)(n - n_main) * K) + kg * 8;
    int m_lo = lrow, m_hi = 16 + lrow;
    m_lo = m_lo < M ? m_lo : M - 1;
    m_hi = m_hi < M ? m_hi : M - 1;
    const bf16_t* xlo = x + (size_t)m_lo * K + kg * 8;
    const bf16_t* xhi = x + (size_t)m_hi * K + kg * 8;
    f32x4 acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = {0.f, 0.f, 0.f, 0.f};
    const int nks = K / 32, kstride = NW * ksplit;
    for (int ks0 = blockIdx.y * NW + wave; ks0 < nks; ks0 += kstride * CH) {
        bf16x8 wf[CH], xl[CH], xh[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ks = ks0 + c * kstride;
            if (ks < nks) {
                wf[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow + ks * 32));
                xl[c] = *reinterpret_cast<const bf16x8*>(xlo + ks * 32);
                xh[c] = *reinterpret_cast<const bf16x8*>(xhi + ks * 32);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ks = ks0 + c * kstride;
            if (ks < nks) {
                acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xl[c], acc_lo, 0, 0, 0);
                acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xh[c], acc_hi, 0, 0, 0);
            }
        }
    }
    *reinterpret_cast<f32x4*>(&part[wave][lrow][kg * 4]) = acc_lo;
    *reinterpret_cast<f32x4*>(&part[wave][16 + lrow][kg * 4]) = acc_hi;
    __syncthreads();
    const int tn = tid & 15, tm = tid >> 4, nn = n0 + tn;
    if (tm >= M || nn >= N) return;
    const float* p = &part[0][0][0] + tm * ROWS + tn;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) s += p[wv * 32 * ROWS];
    y32[((size_t)blockIdx.y * m_total + tm) * N + nn] = s;   // [ksplit][m_total][N], this launch's rows at y32
}

// Row-parallel variant: the 8 waves of a block own 16*CT W rows EACH (128*CT rows per block) and all
// work on the SAME K-slice, so no cross-wave reduction is needed: each wave stores its fp32 tiles straight from
// the accumulators (as full 128-B lines when it owns two adjacent column tiles).  W is fetched once per block as
// full 128-B lines and held as MFMA fragments in registers; NG 32-row groups of x (several batches decoded in one
// launch) pass through LDS in double-buffered rounds of NGL groups (LDS-DMA) and reuse those fragments.
// CT (column tiles per wave): with one tile every MFMA needs its own ds_read_b128 of x and from ~128 rows on the
// LDS port, not the weight stream, bounds the kernel (rocprof at 640 rows: 275-316 TFLOP/s); with CT tiles one x
// fragment feeds CT MFMAs.  The summation order is untouched: same K-slices, same chain inside a slice.
//   grid (ceil(N/(128*CT)), ksplit, ceil(M / (32*NG))), K-slice = KPS k-steps of 32.
// PSTORE: the partial-sum store; -DDH_ROWS_NT_STORE (non-temporal) was measured and is slower end to end (the consumers
// then read the partials from HBM instead of the caches).
#ifdef DH_ROWS_NT_STORE
#define PSTORE(p, v) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p))
#else
#define PSTORE(p, v) (*reinterpret_cast<f32x4*>(p) = (v))
#endif
#ifdef DH_ROWS_STAMPS   // diagnostic build only (tools/probe_rows.py): 100 MHz timestamps of wave 0 per block
__device__ unsigned long long g_rows_stamps[1024 * 8];
#define ROWS_STAMP(i) do { if (threadIdx.x == 0) { const int _b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (_b < 1024) g_rows_stamps[_b * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int dh_debug_rows_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rows_stamps), sizeof(g_rows_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define ROWS_STAMP(i)
#endif

template <int KPS, int NG, int NGL, int CT>
__global__ __launch_bounds__(512, NG == 1 ? 2 : 1) void gemm_skinny_rows_kernel(const bf16_t* __restrict__ x,
                                                                              const bf16_t* __restrict__ w,
                                                                              const bf16_t* __restrict__ w_ext,
                                                                              float* __restrict__ y32, int M, int n_main,
                                                                              int N, int K) {
    // x passes through LDS in ROUNDS of NGL 32-row groups, two buffers: round r+1 is requested (LDS-DMA, 16 B per lane,
    // rows of KPS*64 B with the 16-B chunk index XOR (row & 15) applied on the SOURCE address so the fragment reads are
    // conflict free) before the MFMAs of round r, and waited for after them — the wait also covers the partial-sum
    // stores of round r-1 (loads and stores share vmcnt and complete out of order with each other), so the stores of
    // round r are issued after it and drain under round r+1.  One barrier per round.  K-slices are whole (host check).
    constexpr int RB = KPS * 64;                           // bytes of an x row of the slice
    constexpr int RND = NGL * 32 * RB;                     // bytes of a round buffer
    constexpr int LPR = RB / 16;                           // lanes (16-B chunks) per row: 32 or 64
    constexpr int XI = RND / 1024 / 8;                     // LDS-DMA requests per wave and round
    constexpr int NR = (NG + NGL - 1) / NGL;               // rounds
    static_assert(KPS % 2 == 0 && RND % 8192 == 0, "a line holds two k-steps; a round is a whole number of requests per wave");
    extern __shared__ __attribute__((aligned(16))) char sx[];   // [2][NGL*32][RB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * (128 * CT) + wave * (16 * CT);
    const int m0 = blockIdx.z * (NG * 32);
    const int ks_begin = blockIdx.y * KPS;
    ROWS_STAMP(0);
    // W: full 128-B lines (request: rows 8h..8h+7 of the tile x two k-steps, lane -> (row lane/8, 16-B piece lane%8)),
    // transposed into MFMA fragments in registers (lines_to_frags, common.h): fragment-shaped requests (16 rows x 64 B)
    // cost the texture-address unit four times the line look-ups per byte.
    i32x4 raw[CT][KPS / 2][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int n = n0 + ct * 16 + h * 8 + (lane >> 3);
            n = n < N ? n : N - 1;
            const bf16_t* wrow = (n < n_main ? w + (size_t)n * K : w_ext + (size_t)(n - n_main) * K) + ks_begin * 32 + (lane & 7) * 8;
#pragma unroll
            for (int c2 = 0; c2 < KPS / 2; ++c2) raw[ct][c2][h] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wrow + c2 * 64));
        }
    // request q of a round covers LDS bytes [q KiB, (q+1) KiB): row q*(64/LPR) + lane/LPR, chunk position lane%LPR
    const int xrow_in = lane / LPR, xpos = lane % LPR;
    auto issue_x = [&](int r) __attribute__((always_inline)) {
        char* buf = sx + (r & 1) * RND;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int q = wave + 8 * i;
            const int row = q * (64 / LPR) + xrow_in;              // row of the round
            int m = m0 + r * (NGL * 32) + row;
            m = m < M ? m : M - 1;
            const int chunk = xpos ^ (row & 15);
            glds16(x + (size_t)m * K + ks_begin * 32 + chunk * 8, buf + q * 1024);
        }
    };
    const int rounds = min(NR, (M - m0 + NGL * 32 - 1) / (NGL * 32));   // >= 1 by construction of the grid
    issue_x(0);
    bf16x8 wf[CT][KPS];
    {
        const int fidx = frag_src_lane(lane) * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int c2 = 0; c2 < KPS / 2; ++c2) lines_to_frags(raw[ct][c2][0], raw[ct][c2][1], fidx, wf[ct][2 * c2], wf[ct][2 * c2 + 1]);
    }
    ROWS_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // round 0 is in LDS
    ROWS_STAMP(2);
    float* out = y32 + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r >= rounds) break;
        if (r + 1 < rounds) issue_x(r + 1);                // its buffer was last read in round r-1, which ended with a barrier
        const char* buf = sx + (r & 1) * RND;
        f32x4 acc_lo[NGL][CT], acc_hi[NGL][CT];
#pragma unroll
        for (int g = 0; g < NGL; ++g) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc_lo[g][ct] = acc_hi[g][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r * NGL + g < NG && m0 + (r * NGL + g) * 32 < M) {
#pragma unroll
                for (int c = 0; c < KPS; ++c) {
                    const int co = ((c * 4 + kg) ^ lrow) << 4;
                    const bf16x8 xl = *reinterpret_cast<const bf16x8*>(buf + (g * 32 + lrow) * RB + co);
                    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(buf + (g * 32 + 16 + lrow) * RB + co);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        acc_lo[g][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][c], xl, acc_lo[g][ct], 0, 0, 0);
                        acc_hi[g][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][c], xh, acc_hi[g][ct], 0, 0, 0);
                    }
                }
            }
        }
        if (r + 1 < rounds) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // round r+1 landed (this wave's share), round r-1's stores done
            __builtin_amdgcn_s_barrier();                      // ... everybody's; and everybody has read round r
        }
        ROWS_STAMP(r == 0 ? 3 : 5);
#pragma unroll
        for (int g = 0; g < NGL; ++g) {
            const int mg = m0 + (r * NGL + g) * 32;
            if (r * NGL + g >= NG || mg >= M) break;
            if (CT >= 2 && n0 + 16 * CT <= N) {
                // full 128-B lines: the wave's adjacent 16-column tiles are stored as rows of 32 columns
                const int lidx = line_src_lane(lane) * 4;
#pragma unroll
                for (int cp = 0; cp + 1 < CT; cp += 2) {
                    float* o = out + n0 + cp * 16 + (lane & 7) * 4;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {              // rows mg..mg+15, then mg+16..mg+31
                        f32x4 r0, r1;
                        tiles_to_lines(h ? acc_hi[g][cp] : acc_lo[g][cp], h ? acc_hi[g][cp + 1] : acc_lo[g][cp + 1], lidx, r0, r1);
                        const int m_a = mg + h * 16 + (lane >> 3), m_b = m_a + 8;
                        if (m_a < M) PSTORE(o + (size_t)m_a * N, r0);
                        if (m_b < M) PSTORE(o + (size_t)m_b * N, r1);
                    }
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int nn = n0 + ct * 16 + kg * 4;
                    if (nn < N) {
                        if (mg + lrow < M) *reinterpret_cast<f32x4*>(out + (size_t)(mg + lrow) * N + nn) = acc_lo[g][ct];
                        if (mg + 16 + lrow < M) *reinterpret_cast<f32x4*>(out + (size_t)(mg + 16 + lrow) * N + nn) = acc_hi[g][ct];
                    }
                }
            }
        }
    }
    ROWS_STAMP(6);
}

int g_rows_ng = 0;   // 0: by grid size; 8 or 10 row groups per block above 128 rows (dh_set_tuning key 14)
int g_rows_ct = 0;   // 0: by row count; else forced column tiles per wave (dh_set_tuning key 11: 1, 2 or 4)

template <int KPS, int NG, int CT>
int launch_rows(const bf16_t* x, const bf16_t* w, const bf16_t* w_ext, float* y32, int M, int n_main, int N, int K,
                int ksplit, hipStream_t s) {
    constexpr int NGL = NG == 1 ? 1 : (KPS == 8 ? (NG >= 4 ? 4 : 2) : 2);   // row groups per LDS round (two round buffers)
    constexpr int lds = 2 * NGL * 32 * KPS * 64;
    static_assert(lds <= 160 * 1024, "x rounds do not fit in LDS");
    if (lds > 48 * 1024) DH_MAX_LDS_ONCE((gemm_skinny_rows_kernel<KPS, NG, NGL, CT>), lds);
    dim3 grid(cdiv(N, 128 * CT), ksplit, cdiv(M, NG * 32)), block(512);
    hipLaunchKernelGGL((gemm_skinny_rows_kernel<KPS, NG, NGL, CT>), grid, block, lds, s, x, w, w_ext, y32, M, n_main, N, K);
    DH_LAUNCH_CHECK();
    return 0;
}

template <int KPS>
int launch_rows_ng(const bf16_t* x, const bf16_t* w, const bf16_t* w_ext, float* y32, int M, int n_main, int N, int K,
                   int ksplit, hipStream_t s) {
    if (M <= 32) return launch_rows<KPS, 1, 1>(x, w, w_ext, y32, M, n_main, N, K, ksplit, s);
    if (M <= 64) return launch_rows<KPS, 2, 1>(x, w, w_ext, y32, M, n_main, N, K, ksplit, s);
    if (M <= 128) return launch_rows<KPS, 4, 1>(x, w, w_ext, y32, M, n_main, N, K, ksplit, s);
    // more than 128 rows: two column tiles per wave — one x fragment read feeds two MFMAs, same bits.  Measured on one
    // box at 640 rows (decode step): CT 1 6.46 ms, CT 2 6.20-6.26 ms, CT 4 6.41 ms (its 180 VGPRs cost more than the
    // saved LDS reads); dh_set_tuning(11, ct) overrides
    constexpr int CTMAX = KPS == 8 ? 4 : 2;               // W fragments: CT * KPS * 4 VGPRs
    // up to 256 rows one column tile per wave (twice the blocks: 168 instead of 88 for QKV', 12.6 vs 16.7 us at 256 rows)
    const int ct = g_rows_ct ? g_rows_ct : (M <= 256 ? 1 : 2);
    if (ct >= 4 && CTMAX >= 4) return launch_rows<KPS, 8, CTMAX>(x, w, w_ext, y32, M, n_main, N, K, ksplit, s);
    if (ct >= 2) {
        // one block per CU: a grid just over 256 blocks (640 rows: 264) runs a second round for a handful of blocks.
        // 10 row groups per block instead of 8 when that saves a round (block time ~ 5 + 2 us per group).
        auto cost = [&](int ng) {
            const int blocks = cdiv(N, 256) * ksplit * cdiv(M, ng * 32);
            return cdiv(blocks, 256) * (5 + 2 * ng);
        };
        const bool ten = g_rows_ng ? g_rows_ng == 10 : cost(10) < cost(8);
        if (ten) return launch_rows<KPS, 10, 2>(x, w, w_ext, y32, M, n_main, N, K, ksplit, s);
        return launch_rows<KPS, 8, 2>(x, w, w_ext, y32, M, n_main, N, K, ksplit, s);
    }
    return launch_rows<KPS, 8, 1>(x, w, w_ext, y32, M, n_main, N, K, ksplit, s);
}

int g_skinny_variant = 1;   // 0: K split over the waves of a block, 1: row-parallel with LDS-staged x

template <int EPI>
int launch(const GemmArgs& a, hipStream_t s) {
    dim3 grid(cdiv(a.N, ROWS)), block(512);
    if (a.resid)
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, false>), grid, block, 0, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- skinny-N GEMM (round 4)
// y[M, N] = bf16(x[M, K] . W[N, K]^T) for N = 16 .. 64 and M in the thousands: the LoRA down-projections of the fine-tune (x . A^T on the
// dropped-out input, dy . B for the backward) read 73-92 MB to produce 16-64 columns.  On the 128-tile kernel they are one column
// tile of mostly idle MFMA columns at 1.9-2.4 TB/s (30 us per call on the packed micro-step); here a block owns 64 rows, x and W go
// HBM/L2 -> LDS in whole 128-byte lines by LDS-DMA (four 64-deep stages, three in flight; source-side XOR swizzle so the fragment
// reads are conflict-free), and each wave multiplies its 16 rows against all of W.  One accumulator per output, k ascending in steps
// of 32 on v_mfma_f32_16x16x32_bf16: the bits of the tiled kernels.
template <int NT>
__global__ __launch_bounds__(256, 2) void gemm_skinny_n_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ y, int M, int K, int ldy) {
    constexpr int RB = 64, NST = 4, XB = RB * 128, WB = NT * 16 * 128, STAGE = XB + WB;
    constexpr int WPW = (2 * NT + 3) / 4, NPW = 2 + WPW;          // W pieces / all pieces per wave and stage (every wave the same number)
    extern __shared__ __attribute__((aligned(16))) char sn_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * RB;
    // ---- DMA sources: a piece = 8 rows x 128 B; LDS position P = lane & 7 of row r holds chunk P ^ ((r >> 1) & 7)
    uint32_t xo[2], wo[WPW];
    int wp[WPW];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (2 * wave + i) + (lane >> 3);
        const int row = min(m0 + r, M - 1);
        xo[i] = ((uint32_t)row * (uint32_t)K + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 8)) * 2u;
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        wp[i] = min(wave + 4 * i, 2 * NT - 1);                    // (a repeated piece writes the same bytes to the same place)
        const int r = 8 * wp[i] + (lane >> 3);
        wo[i] = ((uint32_t)r * (uint32_t)K + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 8)) * 2u;
    }
    auto issue = [&](int st) __attribute__((always_inline)) {
        char* base = sn_smem + (st % NST) * STAGE;
        const bf16_t* xs = x + (size_t)st * 64;
        const bf16_t* ws = w + (size_t)st * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_saddr(xs, xo[i], base + (2 * wave + i) * 1024);
#pragma unroll
        for (int i = 0; i < WPW; ++i) glds16_saddr(ws, wo[i], base + XB + wp[i] * 1024);
    };
    const int nst = K / 64;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, kg = lane >> 4;
    const int xr = 16 * wave + frow;                               // this lane's x row inside the block
#pragma unroll 1
    for (int st = 0; st < NST - 1 && st < nst; ++st) issue(st);
#pragma unroll 1
    for (int st = 0; st < nst; ++st) {
        if (st + NST - 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // stage st has landed for every wave; every wave is done with stage st - 1
        if (st + NST - 1 < nst) issue(st + NST - 1);
        const char* base = sn_smem + (st % NST) * STAGE;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int c = 4 * kh + kg;
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(base + xr * 128 + ((c ^ ((xr >> 1) & 7)) << 4));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int wr = 16 * t + frow;
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(base + XB + wr * 128 + ((c ^ ((wr >> 1) & 7)) << 4));
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[t], 0, 0, 0);
            }
        }
    }
    // D[n = 16 t + 4 kg + r][m = row xr]
    const int m = m0 + xr;
    if (m < M) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            *reinterpret_cast<uint2*>(y + (size_t)m * ldy + 16 * t + 4 * kg) = make_uint2(pack2bf(acc[t][0], acc[t][1]), pack2bf(acc[t][2], acc[t][3]));
    }
}

// ---------------------------------------------------------------------------------- K = 64 GEMM (round 4)
// y[M, N] = bf16(x[M, 64] . W[N, 64]^T) (* mul[M, N]): the up-projection of a rank-padded LoRA product in the fine-tune's backward
// ((dy . B) . A, rank 16 / 48 padded to 64, then the branch's dropout mask).  Two MFMAs per 16 x 16 output tile and 73 MB to write
// (plus the mask to read): on the 256-tile kernels a tile's fixed cost dominates (52 us per call with the mask, 2.8 TB/s).  Here a block
// owns 64 rows x 256 columns, x and W land in LDS once (LDS-DMA, whole 128-byte rows, the skinny-N kernel's swizzle), a wave finishes
// its 16 rows against the 16 column tiles, pairs of tiles regrouped by v_permlane16_swap into 16-byte stores as the 256-tile epilogues do.
// One accumulator, k ascending: the bits of the tiled kernels; the multiply rounds as a separate bf16 multiply would.
template <bool MUL>
__global__ __launch_bounds__(256, 4) void gemm_k64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ mul, bf16_t* __restrict__ y, int M, int N) {
    constexpr int RB = 64, CB = 256, XB = RB * 128, WB = CB * 128;
    __shared__ __attribute__((aligned(16))) char k64_smem[XB + WB];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.y * RB, n0 = blockIdx.x * CB;
    // pieces of 8 rows x 128 B: x 8 (two per wave), W 32 (eight per wave); position P of row r holds chunk P ^ ((r >> 1) & 7)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (2 * wave + i) + (lane >> 3);
        const int row = min(m0 + r, M - 1);
        glds16_saddr(x, ((uint32_t)row * 64u + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 8)) * 2u, k64_smem + (2 * wave + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = 8 * (8 * wave + i) + (lane >> 3);
        const int row = min(n0 + r, N - 1);
        glds16_saddr(w, ((uint32_t)row * 64u + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 8)) * 2u, k64_smem + XB + (8 * wave + i) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 15, kg = lane >> 4;
    const int xr = 16 * wave + frow, m = m0 + xr;
    bf16x8 bf[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) bf[kh] = *reinterpret_cast<const bf16x8*>(k64_smem + xr * 128 + (((4 * kh + kg) ^ ((xr >> 1) & 7)) << 4));
    const bool m_ok = m < M;
    const int cpair = (kg & 1) * 16 + (kg >> 1) * 8;          // this lane's 8 columns inside a 32-column pair after the swap
#pragma unroll
    for (int tp = 0; tp < CB / 32; ++tp) {
        f32x4 acc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int wr = 32 * tp + 16 * h + frow;
            acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(k64_smem + XB + wr * 128 + (((4 * kh + kg) ^ ((wr >> 1) & 7)) << 4));
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[kh], acc[h], 0, 0, 0);
            }
        }
        // lane (row frow, kg): columns 4 kg .. + 3 of tile h -> after the swaps 8 consecutive columns of the pair
        const uint2 ta = make_uint2(pack2bf(acc[0][0], acc[0][1]), pack2bf(acc[0][2], acc[0][3]));
        const uint2 tb = make_uint2(pack2bf(acc[1][0], acc[1][1]), pack2bf(acc[1][2], acc[1][3]));
        const auto rx = __builtin_amdgcn_permlane16_swap(ta.x, tb.x, false, false);
        const auto ry = __builtin_amdgcn_permlane16_swap(ta.y, tb.y, false, false);
        uint4 out = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        const int n = n0 + 32 * tp + cpair;
        if (m_ok && n + 8 <= N) {
            if (MUL) {
                const uint4 mv = *reinterpret_cast<const uint4*>(mul + (size_t)m * N + n);
                auto mul2 = [](uint32_t y2, uint32_t r2) __attribute__((always_inline)) -> uint32_t {
                    return pack2bf(bf2f((bf16_t)(r2 & 0xffffu)) * bf2f((bf16_t)(y2 & 0xffffu)), bf2f((bf16_t)(r2 >> 16)) * bf2f((bf16_t)(y2 >> 16)));
                };
                out = make_uint4(mul2(out.x, mv.x), mul2(out.y, mv.y), mul2(out.z, mv.z), mul2(out.w, mv.w));
            }
            *reinterpret_cast<uint4*>(y + (size_t)m * N + n) = out;
        }
    }
}

}  // namespace

extern "C" int dh_linear_partial_bf16(const dh_bf16* x, const dh_bf16* w, const dh_bf16* w_ext, float* y32, int M,
                                      int n_main, int n_ext, int K, int ksplit, void* stream) {
    DH_CHECK(x && w && y32 && M >= 1 && M <= 4096, "dh_linear_partial_bf16: need 1 <= M <= 4096 (got %d)", M);
    DH_CHECK(K % 32 == 0 && n_main % ROWS == 0 && n_ext % ROWS == 0 && n_main > 0 && n_ext >= 0,
             "dh_linear_partial_bf16: K %% 32 and row counts %% 16 must be 0");
    DH_CHECK(n_ext == 0 || w_ext, "dh_linear_partial_bf16: null w_ext");
    DH_CHECK(ksplit >= 1 && ksplit <= 16, "dh_linear_partial_bf16: ksplit must be 1..16");
    const int N = n_main + n_ext;
    const int nks = K / 32;
    const int kps = (nks + ksplit - 1) / ksplit;
    hipStream_t s = (hipStream_t)stream;
    // the kernel is a function of (K, ksplit) alone, never of M: rows of a larger call equal the same rows alone
    if (g_skinny_variant == 1 && (kps == 8 || kps == 16) && ksplit * kps == nks && N % 4 == 0) {
        if (kps == 8) return launch_rows_ng<8>(x, w, w_ext ? w_ext : w, y32, M, n_main, N, K, ksplit, s);
        return launch_rows_ng<16>(x, w, w_ext ? w_ext : w, y32, M, n_main, N, K, ksplit, s);
    }
    DH_CHECK(ksplit <= 8, "dh_linear_partial_bf16: ksplit must be 1..8 for this shape");
    // K-split-over-waves kernel: 32 rows per launch
    for (int m0 = 0; m0 < M; m0 += 32) {
        const int mm = M - m0 < 32 ? M - m0 : 32;
        hipLaunchKernelGGL(gemm_skinny_partial_kernel, dim3(N / ROWS, ksplit), dim3(512), 0, s, x + (size_t)m0 * K, w,
                           w_ext ? w_ext : w, y32 + (size_t)m0 * N, mm, M, n_main, N, K);
    }
    DH_LAUNCH_CHECK();
    return 0;
}

// the rows kernel's K-slices (kps 8 or 16) are contiguous chains, so their ordered sum can also be
// produced by the tiled kernel in one launch (n_part = 1 for the consumers)
bool dh_chain_ok(int M, int n_main, int n_ext, int K, int ksplit) {
    const int nks = K / 32, kps = (nks + ksplit - 1) / ksplit, N = n_main + n_ext;
    return g_skinny_variant == 1 && M >= g_chain_min_rows && K % 64 == 0 && (kps == 8 || kps == 16) && (ksplit - 1) * kps < nks &&
           N % 4 == 0 && n_main % ROWS == 0 && n_ext % ROWS == 0;
}

// partial-sum GEMMs of at least this many rows take the tiled split-K kernel (dh_set_tuning key 18).  tools/sweep_pairs.py,
// us (slices | pair sums): 160 rows qkv' 11.2 | 12.5, mlp' 15.1 | 19.3; 256 rows 12.9 | 12.8, 18.9 | 20.0; 384 rows 17.0 | 13.5,
// 25.3 | 23.1; 640 rows 20.7 | 16.0, 30.2 | 24.3; 2048 rows 55.6 | 42.2, 81.5 | 74.0 — and the consumers read half the bytes
int g_pairs_min_rows = 288;
bool dh_pairs_ok(int M, int n_main, int n_ext, int K, int ksplit) {
    const int nks = K / 32, kps = (nks + ksplit - 1) / ksplit, N = n_main + n_ext;
    return g_skinny_variant == 1 && M >= g_pairs_min_rows && K % 64 == 0 && (kps == 8 || kps == 16) && ksplit * kps == nks &&
           N % 4 == 0 && n_main % ROWS == 0 && n_ext % ROWS == 0;
}

extern "C" int dh_linear_partial_pairs_bf16(const dh_bf16* x, const dh_bf16* w, const dh_bf16* w_ext, float* y32, int M,
                                            int n_main, int n_ext, int K, int ksplit, void* stream) {
    DH_CHECK(x && w && y32 && M >= 1 && ksplit >= 1 && ksplit <= 16 && (n_ext == 0 || w_ext), "dh_linear_partial_pairs_bf16: bad argument");
    const int nks = K / 32, kps = (nks + ksplit - 1) / ksplit;
    DH_CHECK(K % 64 == 0 && (kps == 8 || kps == 16) && ksplit * kps == nks && (n_main + n_ext) % 4 == 0 && n_main % ROWS == 0 &&
             n_ext % ROWS == 0, "dh_linear_partial_pairs_bf16: unsupported shape N=%d+%d K=%d ksplit=%d (K %% 64 == 0, whole K-slices of 8 or 16 k-steps)",
             n_main, n_ext, K, ksplit);
    return dh_pairs_tiled(x, w, w_ext, y32, M, n_main, n_ext, K, kps, (hipStream_t)stream);
}

extern "C" int dh_linear_chain_bf16(const dh_bf16* x, const dh_bf16* w, const dh_bf16* w_ext, float* y32, int M, int n_main,
                                    int n_ext, int K, int ksplit, void* stream) {
    DH_CHECK(x && w && y32 && M >= 1 && ksplit >= 1 && (n_ext == 0 || w_ext), "dh_linear_chain_bf16: bad argument");
    DH_CHECK(dh_chain_ok(M, n_main, n_ext, K, ksplit),
             "dh_linear_chain_bf16: unsupported shape M=%d N=%d+%d K=%d ksplit=%d (needs >= %d rows, K %% 64 == 0, K-slices of 8 or 16 k-steps)",
             M, n_main, n_ext, K, ksplit, g_chain_min_rows);
    const int kps = (K / 32 + ksplit - 1) / ksplit;
    return dh_chain_tiled(x, w, w_ext, y32, M, n_main, n_ext, K, kps, (hipStream_t)stream);
}

int g_skinny_n = 1;      // dh_set_tuning(31, 0): the 128-tile kernel for N <= 64 (A/B)
bool dh_linear_skinny_n_ok(int M, int N, int K, const void* x, const void* w, const void* y) {
    return g_skinny_n && N >= 16 && N <= 64 && N % 16 == 0 && M >= 512 && K >= 64 && K % 64 == 0 && (size_t)M * K * 2 < (1ull << 32) &&
           (((uintptr_t)x | (uintptr_t)w) & 15) == 0 && ((uintptr_t)y & 7) == 0;
}
int dh_linear_skinny_n(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, hipStream_t s) {
    const dim3 grid(cdiv(M, 64)), block(256);
    const int nt = N / 16, lds = 4 * (64 * 128 + nt * 16 * 128);
    switch (nt) {
        case 1: hipLaunchKernelGGL((gemm_skinny_n_kernel<1>), grid, block, lds, s, x, w, y, M, K, N); break;
        case 2: hipLaunchKernelGGL((gemm_skinny_n_kernel<2>), grid, block, lds, s, x, w, y, M, K, N); break;
        case 3: hipLaunchKernelGGL((gemm_skinny_n_kernel<3>), grid, block, lds, s, x, w, y, M, K, N); break;
        default: hipLaunchKernelGGL((gemm_skinny_n_kernel<4>), grid, block, lds, s, x, w, y, M, K, N); break;
    }
    DH_LAUNCH_CHECK();
    return 0;
}

// y = bf16(x . W^T) [* mul] for K = 64 and a wide N (gemm_k64_kernel); N % 8 == 0 so that a lane's 8 columns are all inside or all outside
bool dh_linear_k64_ok(int M, int N, int K, const void* x, const void* w, const void* y, const void* mul) {
    return g_skinny_n && K == 64 && N >= 256 && N % 8 == 0 && M >= 512 && (size_t)M * 128 < (1ull << 32) && (size_t)N * 128 < (1ull << 32) &&
           (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)mul) & 15) == 0;
}
int dh_linear_k64(const dh_bf16* x, const dh_bf16* w, const dh_bf16* mul, dh_bf16* y, int M, int N, hipStream_t s) {
    const dim3 grid(cdiv(N, 256), cdiv(M, 64)), block(256);
    if (mul) hipLaunchKernelGGL((gemm_k64_kernel<true>), grid, block, 0, s, x, w, mul, y, M, N);
    else hipLaunchKernelGGL((gemm_k64_kernel<false>), grid, block, 0, s, x, w, mul, y, M, N);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_set_tuning(int key, int value) {
    if (key == 0) { g_skinny_variant = value; return 0; }
    if (key == 1) { g_gemm_variant = value; return 0; }
    if (key == 2) { g_swiglu2 = value; return 0; }
    if (key == 3) { g_mid = value; return 0; }
    if (key == 4) { g_linear_phase = value; return 0; }
    if (key == 5 && value >= 0) { g_gemm_gm = value; return 0; }
    if (key == 6 && value >= 1) { g_dt_min_rows = value; return 0; }
    if (key == 7 && value >= 1) { g_chain_min_rows = value; return 0; }
    if (key == 8) { extern int g_dt_stages; g_dt_stages = value; return 0; }
    if (key == 9) { extern int g_gemm128_stages; g_gemm128_stages = value; return 0; }
    if (key == 10 && value >= 0) { extern int g_decode_tiled_rows; g_decode_tiled_rows = value; return 0; }
    if (key == 11 && value >= 0) { g_rows_ct = value; return 0; }
    if (key == 12) { extern int g_fuse_qkv_rope; g_fuse_qkv_rope = value; return 0; }
    if (key == 14 && (value == 0 || value == 8 || value == 10)) { g_rows_ng = value; return 0; }
    if (key == 16 && (value == 8 || value == 16)) { extern int g_short_kps; g_short_kps = value; return 0; }
    if (key == 15) { extern int g_dt_wide; g_dt_wide = value; return 0; }
    if (key == 13) { extern int g_mid_wlds; g_mid_wlds = value; return 0; }
    if (key == 17 && (value == 0 || value == 2 || value == 4)) { extern int g_pairs_wn; g_pairs_wn = value; return 0; }
    if (key == 18 && value >= 1) { g_pairs_min_rows = value; return 0; }
    if (key == 21) { extern int g_pairs_wt; g_pairs_wt = value != 0; return 0; }
    if (key == 19 && (value == 0 || value == 128 || value == 256)) { extern int g_fp8_tile; g_fp8_tile = value; return 0; }
    if (key == 20 && value >= 0 && value <= 64) { extern int g_fp8_gm; g_fp8_gm = value; return 0; }
    if (key == 22 && value >= 0 && value <= 2) { extern int g_w4_persist; g_w4_persist = value; return 0; }
    if (key == 23) { extern int g_prune_last_layer; g_prune_last_layer = value != 0; return 0; }
    if (key == 24) { extern int g_w4_fast_epi; g_w4_fast_epi = value; return 0; }
    if (key == 25) { extern int g_w4_persist_qkv; g_w4_persist_qkv = value != 0; return 0; }
    if (key == 26) { extern int g_tn_mfma; g_tn_mfma = value != 0; return 0; }
    if (key == 31) { g_skinny_n = value != 0; return 0; }
    if (key == 30) { extern int g_w4_persist_lora; g_w4_persist_lora = value != 0; return 0; }
    if (key == 29) { extern int g_attn_bwd_dkdv_img; g_attn_bwd_dkdv_img = value != 0; return 0; }
    if (key == 28) { extern int g_tail_split; g_tail_split = value != 0; return 0; }
    if (key == 27) { extern int g_attn_bwd_dq_group; g_attn_bwd_dq_group = value != 0; return 0; }
    dh_set_error("dh_set_tuning: unknown key %d", key);
    return 1;
}

int dh_linear_skinny(const GemmArgs& a, int epilogue, hipStream_t s) {
    if (epilogue == DH_EPI_SWIGLU && g_swiglu2 && a.N % 32 == 0) {
        DH_MAX_LDS_ONCE(swiglu_skinny2_kernel, 64 * 1024);
        hipLaunchKernelGGL(swiglu_skinny2_kernel, dim3(a.N / 32), dim3(512), 64 * 1024, s, a);
        DH_LAUNCH_CHECK();
        return 0;
    }
    switch (epilogue) {
        case DH_EPI_PLAIN: return launch<DH_EPI_PLAIN>(a, s);
        case DH_EPI_LORA: return launch<DH_EPI_LORA>(a, s);
        case DH_EPI_SWIGLU: return launch<DH_EPI_SWIGLU>(a, s);
        case DH_EPI_ADAPTER: return launch<DH_EPI_ADAPTER>(a, s);
    }
    dh_set_error("dh_linear_bf16: unknown epilogue %d", epilogue);
    return 1;
}
