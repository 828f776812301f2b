// fp8 (OCP e4m3fn) serving path — BASELINE config 5: merged-LoRA weights quantised per output channel, activations
// quantised per token at run time, products on the block-scaled fp8 MFMA of gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4
// with unit block scales: twice the bf16 rate per clock), per-channel x per-token scales applied to the fp32
// accumulator in the epilogue:
//     y[m,n] = bf16( (sum_k xq[m,k] * wq[n,k]) * (x_scale[m] * w_scale[n]) )
//     q = fp8_rne(v * inv),  inv = 448 / amax (IEEE fp32 division),  scale = amax * fp32(1/448),  amax = max|row| (>= 1e-12)
// The reference has no fp8 code (ger/lora.py:152-157,349-365,707-711 are the merge this path starts from); the
// arithmetic is restated on the CPU in oracle/ger_oracle.py (quantize_rows_fp8 / linear_fp8) with torch's e4m3fn type.
//
//   quant_rows_fp8_kernel       bf16 rows -> e4m3 rows + one fp32 scale per row (one wave per row, two passes)
//   rmsnorm_quant_fp8_block_kernel  RMSNorm (ger/rmsnorm.py:17-21, bf16 rounding points) with the quantisation fused
//   gemm_fp8_kernel             M > 32: 128 x 128 x 128 tiles, both operands through LDS (global_load_lds 16 B, rows
//                               of 128 B with the source-side XOR swizzle of gemm.hip), 4 waves x (4 x 4) MFMA tiles
//   gemm_fp8_skinny_kernel      M <= 128 (decode, up to four 32-row batches): W streamed HBM -> VGPR once, K dealt over the 8 waves of a block
// Tried and dropped: a port of gemm256.hip's 256 x 256 ping-pong kernel to v_mfma_scale_f32_32x32x64_f8f6f4 (operand map
// verified with integers: lane l = row l & 31, k 32 (l >> 5) ..+32).  With two fragment sets it spills (128 accumulator
// + 2 x 48 fragment registers: 0.57 PFLOP/s), with one set it reaches 1.15 PFLOP/s against 1.5 for the 128-tile kernel.
// MFMA operand map (checked with integer data, tests/test_hip_fp8.py): lane l holds the 32 consecutive k
// [32 (l >> 4), 32 (l >> 4) + 32) of row (A) / column (B) l & 15 — one MX block per lane, hence one scale per lane.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

int g_fp8_tile = 0;   // dh_set_tuning key 19: 0 = by grid size, 128 / 256 = forced tile edge of the tiled fp8 GEMM
// dh_set_tuning key 20: m-tiles per band of the tiled fp8 GEMM's tile walk (1 = row-major, 0 = by kernel).  tools/sweep_fp8_tiles.py
// at M = 49152, PFLOP/s by band 1 / 2 / 4 / 8 — 256-tile: qkv 1.99 / 2.04 / 2.10 / 1.95, proj 1.71 / 1.64 / 1.80 / 1.56, SwiGLU 2.15 /
// 2.27 / 2.07 / 2.07, mlp 2.08 / 2.06 / 2.16 / 2.01; 128-tile: qkv 1.31 / 1.49 / 1.68 / 1.75, SwiGLU 1.63 / 1.80 / 1.81 / 1.59
int g_fp8_gm = 0;

namespace {

constexpr float FP8_MAX = 448.0f;
constexpr float INV_FP8_MAX = 1.0f / 448.0f;   // the fp32 nearest to 1/448: a constant on both sides, not a division
constexpr int UNIT_SCALE = 0x7F7F7F7F;   // four E8M0 bytes of 2^0

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);   // OCP e4m3fn on gfx950, round to nearest even
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}

__device__ __forceinline__ f32x4 mfma_fp8(const i32x8& a, const i32x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A e4m3 */, 0 /* B e4m3 */, 0, UNIT_SCALE, 0, UNIT_SCALE);
}

// ------------------------------------------------------------------------------ row quantisation
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                             float* __restrict__ scale, int rows, int K) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nchunk = K >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * K);
    float amax = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        const uint4 u = xr[c];
        const bf16_t* p = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf2f(p[j])));
    }
    amax = fmaxf(wave_max(amax), 1e-12f);
    const float inv = __fdiv_rn(FP8_MAX, amax);         // correctly rounded, as the oracle's tensor / tensor division
    if (lane == 0) scale[row] = amax * INV_FP8_MAX;
    uint2* qr = reinterpret_cast<uint2*>(q + (size_t)row * K);
    for (int c = lane; c < nchunk; c += 64) {
        const uint4 u = xr[c];                         // second pass: an L2 hit
        const bf16_t* p = reinterpret_cast<const bf16_t*>(&u);
        qr[c] = make_uint2(pack4_fp8(bf2f(p[0]) * inv, bf2f(p[1]) * inv, bf2f(p[2]) * inv, bf2f(p[3]) * inv),
                           pack4_fp8(bf2f(p[4]) * inv, bf2f(p[5]) * inv, bf2f(p[6]) * inv, bf2f(p[7]) * inv));
    }
}

// Few rows (a decode step): one 256-thread BLOCK per row — a single wave walks K = 14336 in 2 x 28 dependent
// iterations (17 us for 32 rows), a block in 2 x 7.
__global__ __launch_bounds__(256) void quant_rows_fp8_block_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                                   float* __restrict__ scale, int K) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = K >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * K);
    float amax = 0.f;
    for (int c = tid; c < nchunk; c += 256) {
        const uint4 u = xr[c];
        const bf16_t* p = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf2f(p[j])));
    }
    amax = wave_max(amax);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-12f);
    const float inv = __fdiv_rn(FP8_MAX, amax);
    if (tid == 0) scale[row] = amax * INV_FP8_MAX;
    uint2* qr = reinterpret_cast<uint2*>(q + (size_t)row * K);
    for (int c = tid; c < nchunk; c += 256) {
        const uint4 u = xr[c];
        const bf16_t* p = reinterpret_cast<const bf16_t*>(&u);
        qr[c] = make_uint2(pack4_fp8(bf2f(p[0]) * inv, bf2f(p[1]) * inv, bf2f(p[2]) * inv, bf2f(p[3]) * inv),
                           pack4_fp8(bf2f(p[4]) * inv, bf2f(p[5]) * inv, bf2f(p[6]) * inv, bf2f(p[7]) * inv));
    }
}

// RMSNorm with elementwise.hip's rounding points and Q11 flag (ger/rmsnorm.py:17-21) whose bf16 output row is quantised
// before it leaves the registers; xn_out (nullable) still receives the bf16 row.  One 256-thread block per row (16
// elements per thread at d = 4096) for every row count, so a row's sum of squares is added up in the same order in a
// 32-row decode step and in a 49 152-row prefill (a wave-per-row version took 10.7 us for 32 rows: three dependent
// memory phases of a single wave).
template <int EPT>
__global__ __launch_bounds__(256) void rmsnorm_quant_fp8_block_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                      bf16_t* __restrict__ xn_out, uint8_t* __restrict__ q,
                                                                      float* __restrict__ scale, int d, float eps,
                                                                      const uint8_t* __restrict__ row_tail) {
    constexpr int NC = EPT / 8;
    __shared__ float red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = d >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * d);
    const uint4* wr = reinterpret_cast<const uint4*>(w);
    float v[NC][8];
    uint4 wu[NC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        if (c < nchunk) {
            const uint4 u = xr[c];
            wu[i] = wr[c];                              // requested with x: one memory phase, not two
            const bf16_t* p = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = bf2f(p[j]);
                ss += rbf(v[i][j] * v[i][j]);
            }
        }
    }
    // the fp32 sum of the bf16-rounded squares: lanes, then waves, in a fixed order
    ss = wave_sum(ss);
    if (lane == 0) red[0][wave] = ss;
    __syncthreads();
    ss = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float ms = rbf(ss / (float)d);
    const float t = rbf(ms + eps);
    const bool tail = row_tail != nullptr && row_tail[row] != 0;
    const float r = tail ? rbf(1.0f / rbf(sqrtf(t))) : rbf(1.0f / sqrtf(t));
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        if (c < nchunk) {
            const bf16_t* wp = reinterpret_cast<const bf16_t*>(&wu[i]);
            uint4 o;
            bf16_t* op = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                op[j] = f2bf(bf2f(wp[j]) * rbf(v[i][j] * r));
                v[i][j] = bf2f(op[j]);
                amax = fmaxf(amax, fabsf(v[i][j]));
            }
            if (xn_out) reinterpret_cast<uint4*>(xn_out + (size_t)row * d)[c] = o;
        }
    }
    amax = wave_max(amax);
    if (lane == 0) red[1][wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])), 1e-12f);
    const float inv = __fdiv_rn(FP8_MAX, amax);
    if (tid == 0) scale[row] = amax * INV_FP8_MAX;
    uint2* qr = reinterpret_cast<uint2*>(q + (size_t)row * d);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        if (c < nchunk)
            qr[c] = make_uint2(pack4_fp8(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv),
                               pack4_fp8(v[i][4] * inv, v[i][5] * inv, v[i][6] * inv, v[i][7] * inv));
    }
}

// ------------------------------------------------------------------------------ GEMM argument block
struct Fp8Args {
    const uint8_t* x;      // [M, K] e4m3
    const uint8_t* w;      // [N, K] e4m3
    const uint8_t* w2;     // SWIGLU: fc_2
    bf16_t* y;             // [M, N]
    const float* xs;       // [M] activation scales
    const float* ws;       // [N] channel scales of w
    const float* ws2;      // [N] channel scales of w2
    const bf16_t* vec_a;   // ADAPTER scale [N]
    const bf16_t* vec_b;   // ADAPTER bias [N]
    const bf16_t* resid;   // [M, N] or null
    int M, N, K;
    int nb_n, nb_m;
    int gm;                // tiled kernel: m-tiles per band of the tile walk (0 / 1 = row-major)
};

constexpr int BT = 128;                 // block tile edge
constexpr int BKB = 128;                // bytes (= fp8 elements) of K per stage and row
constexpr int TILE_BYTES = BT * BKB;    // 16 KiB

// 16-B piece p of LDS row r sits at piece position p ^ swz(r).  An fp8 fragment lane (row, kg) reads pieces 2kg and 2kg+1,
// so the lanes a ds_read_b128 services together (8 rows with kg = a, 8 with a + 1: MI355X_MICROARCH.md, LDS) ask for
// pieces that differ by 2 — gemm.hip's (row >> 1) & 7, built for pieces that differ by 1, put every pair of them on one
// 16-B slot (2-way conflict on every fragment read: the LDS array was as busy as the matrix pipe).  Moving row bit 2
// onto piece bit 2 separates them (brute-forced over the four lane groups).
__device__ __forceinline__ int swz(int row) {
    const int j = (row >> 1) & 7;
    return j ^ (((j >> 1) & 1) << 2);
}

// finish one accumulator value: scale, round, epilogue (the bf16 rounding points of the bf16 path)
template <int EPI>
__device__ __forceinline__ float fp8_finish(float acc, float sc, const Fp8Args& a, int n) {
    float o = rbf(acc * sc);
    if (EPI == DH_EPI_ADAPTER) o = rbf(bf2f(a.vec_a[n]) * rbf(o + bf2f(a.vec_b[n])));
    return o;
}

// ------------------------------------------------------------------------------ tiled kernel (M > 32)
// Same skeleton as gemm.hip's gemm_nt_kernel: C^T tiles (A operand = W rows, B operand = x rows), XCD-aware
// tile order, NST LDS stages with counted vmcnt.  A stage is one MFMA k-step: 128 k = 128 bytes per row.
// WG (waves per tile edge): 2 = 128 x 128 tile on 4 waves, two blocks per CU; 4 (round 3) = 256 x 256 tile on SIXTEEN waves
// (each still a 64 x 64 wave tile: 64 accumulator registers), one block per CU, two stages of 64 KiB.  The 128-tile stages
// 32 KiB per 4.2 MFLOP: at the ~65 GB/s a CU takes in by LDS-DMA that caps the kernel near 2.1 PFLOP/s; the 256-tile halves
// the bytes per FLOP (and the DMA requests per wave), at four waves per SIMD (128 VGPRs: the x fragments are read one at a
// time instead of four up front).
template <int EPI, bool RESID, int NST, int WG = 2>
__global__ __launch_bounds__(WG * WG * 64, WG == 2 ? 2 : 1) void gemm_fp8_kernel(Fp8Args a) {
    constexpr int BTW = WG * 64;                    // tile edge
    constexpr int TILE = BTW * BKB;                 // bytes of one operand tile of a stage
    constexpr int GPW = 8 / WG;                     // 1-KiB row groups (8 rows x 128 B) per wave and operand
    static_assert(WG == 2 || (WG == 4 && NST == 2), "the 256-tile runs the two-stage loop");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NST x (W tile + x tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WG, wm = wave % WG;
    const int nwg = a.nb_n * a.nb_m;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // tiles are walked in bands of gm m-tiles, m fastest (as gemm256.hip): the ~32 blocks an XCD runs at a time form a
    // gm x (32 / gm) rectangle that shares gm x-tile and 32 / gm W-tile streams through its L2 instead of 1 + 32 — a 256-tile
    // stages 2 MiB per tile, 9.4 GB per QKV launch if nothing is shared
    int tm, tn;
    if (a.gm > 1) {
        const int band = tile / (a.gm * a.nb_n), within = tile - band * (a.gm * a.nb_n);
        const int rows = min(a.gm, a.nb_m - band * a.gm);
        tm = band * a.gm + within % rows; tn = within / rows;
    } else {
        tm = tile / a.nb_n; tn = tile % a.nb_n;
    }
    const int m0 = tm * BTW;
    const int n0 = (EPI == DH_EPI_SWIGLU) ? tn * (WG * 32) : tn * BTW;

    const uint8_t* srcA[GPW];
    const uint8_t* srcB[GPW];
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        const int R = wave * GPW + j;             // 1-KiB row group: LDS rows R*8 .. R*8+7
        const int row = R * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);  // logical 16-B chunk held by this LDS slot
        {
            const uint8_t* base = a.w;
            int n;
            if (EPI == DH_EPI_SWIGLU) {
                const int half = (row >> 5) & 1;  // 32-row halves of a wave's 64 rows: 0 = fc_1, 1 = fc_2
                n = n0 + (row >> 6) * 32 + (row & 31);
                base = half ? a.w2 : a.w;
            } else {
                n = n0 + row;
            }
            n = n < a.N ? n : a.N - 1;
            srcA[j] = base + (size_t)n * a.K + chunk * 16;
        }
        {
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            srcB[j] = a.x + (size_t)m * a.K + chunk * 16;
        }
    }
    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * 2 * TILE;
        char* sB = sA + TILE;
#pragma unroll
        for (int j = 0; j < GPW; ++j) {
            const int R = wave * GPW + j;
            glds16(srcA[j] + (size_t)kt * BKB, sA + R * 1024);
            glds16(srcB[j] + (size_t)kt * BKB, sB + R * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, kg = lane >> 4;
    const int offA = (wn * 64 + frow) * 128, offB = (wm * 64 + frow) * 128;
    const int c0 = ((2 * kg) ^ swz(frow)) << 4, c1 = ((2 * kg + 1) ^ swz(frow)) << 4;
    const int nk = a.K / BKB;
    auto frag = [&](const char* base) __attribute__((always_inline)) -> i32x8 {
        const uint4 lo = *reinterpret_cast<const uint4*>(base + c0);
        const uint4 hi = *reinterpret_cast<const uint4*>(base + c1);
        return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };
    // (Round 3, measured and not kept: the next stage's four DMA pieces of a wave spread between the groups of four MFMAs instead of in
    //  a row at the top of the iteration, as in gemm_dt.hip's 8-wave tile — 0.76 against 2.1 PFLOP/s at the 128-register budget of four
    //  waves per SIMD.)
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* sA = smem + buf * 2 * TILE;
        const char* sB = sA + TILE;
        if constexpr (WG == 2) {
            i32x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = frag(sA + offA + i * 2048);
                fb[i] = frag(sB + offB + i * 2048);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_fp8(fa[i], fb[j], acc[i][j]);
        } else {
            // four waves per SIMD: 128 VGPRs each — the four W fragments up front, the x fragments one (plus one in flight) at a time
            i32x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag(sA + offA + i * 2048);
            i32x8 fb = frag(sB + offB);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                i32x8 nxt = fb;
                if (j + 1 < 4) nxt = frag(sB + offB + (j + 1) * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma_fp8(fa[i], fb, acc[i][j]);
                fb = nxt;
            }
        }
    };
    if constexpr (NST == 2) {
        stage(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            compute(cur);
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int p = 0; p < NST - 1; ++p)
            if (p < nk) stage(p, p);
        if (2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 3 < nk) stage((kt + 3) % NST, kt + 3);
            compute(kt % NST);
            if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---------------------------------------------------------------- epilogue
    // accumulator element r of tile (i,j): n = nt + 4*kg + r ; m = mt + frow
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        if (m >= a.M) continue;
        const float xs = a.xs[m];
        if (EPI == DH_EPI_SWIGLU) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = n0 + wn * 32 + i * 16 + 4 * kg;
                if (n >= a.N) continue;
                const float4 s1 = *reinterpret_cast<const float4*>(a.ws + n), s2 = *reinterpret_cast<const float4*>(a.ws2 + n);
                const float* s1p = reinterpret_cast<const float*>(&s1);
                const float* s2p = reinterpret_cast<const float*>(&s2);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = rbf(acc[i][j][e] * (xs * s1p[e]));
                    const float u = rbf(acc[i + 2][j][e] * (xs * s2p[e]));
                    o[e] = rbf(silu_fast(g)) * u;
                }
                *reinterpret_cast<uint2*>(a.y + (size_t)m * a.N + n) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + i * 16 + 4 * kg;
                if (n >= a.N) continue;
                const float4 s1 = *reinterpret_cast<const float4*>(a.ws + n);
                const float* s1p = reinterpret_cast<const float*>(&s1);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fp8_finish<EPI>(acc[i][j][e], xs * s1p[e], a, n + e);
                if (RESID) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.N + n);
                    const bf16_t* rp = reinterpret_cast<const bf16_t*>(&rr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = bf2f(rp[e]) + o[e];
                }
                *reinterpret_cast<uint2*>(a.y + (size_t)m * a.N + n) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
        }
    }
}

// ------------------------------------------------------------------------------ streaming kernel (M <= 32)
// As gemm_skinny.hip's gemm_skinny_kernel: one block = 16 W rows (SWIGLU: of fc_1 and fc_2), K dealt round-robin in
// 128-wide k-steps over the 8 waves, W global -> VGPR (non-temporal, each byte is needed once), the per-wave fp32
// partials meet in LDS and thread t finishes output (n = t & 15, m = t >> 4).
constexpr int ROWS = 16, NW = 8;
constexpr int FP8_STREAM_MAX_ROWS = 128;

template <int EPI, bool RESID, bool OUT32 = false, int NG = 1>
__global__ __launch_bounds__(512, 2) void gemm_fp8_skinny_kernel(Fp8Args a) {
    // NG 32-row groups (several batches decoded in one step) reuse the W fragments a wave holds: W is still streamed
    // once.  A row's chain (its wave's k-steps ascending, waves added in order) is the same for every NG.
    constexpr bool SW = EPI == DH_EPI_SWIGLU;
    constexpr int CH = SW ? 2 : 4;                        // k-steps (32 B per lane and operand) loaded ahead per wave
    __shared__ __attribute__((aligned(16))) float part[NW][32][ROWS];
    __shared__ __attribute__((aligned(16))) float part2[SW ? NW : 1][32][ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * ROWS;
    const int lrow = lane & 15, kg = lane >> 4;
    // Operands are fetched as FULL 128-B lines — request h covers rows 8h..8h+7 of a 16-row block, lane -> (row lane/8,
    // 16-B piece: the four even pieces on lanes 0-3 of the row's eight, the odd ones on lanes 4-7) — and turned into MFMA fragments (lane (row, kg) holds pieces 2kg, 2kg+1 of its row) in
    // registers: fragment-shaped requests (16 rows x 4 separate pieces per instruction) cost the texture-address unit
    // four times the line look-ups per byte, which bounded this kernel at ~2 TB/s (tools/probe_mid.py has the bf16 case).
    const int lr8 = lane >> 3, pc = (((lane & 3) << 1) | ((lane >> 2) & 1)) * 16;
    auto line_ptr = [&](const uint8_t* base, int row, int rows) __attribute__((always_inline)) {
        row = row < rows ? row : rows - 1;
        return base + (size_t)row * a.K + pc;
    };
    const uint8_t* wl[2] = {line_ptr(a.w, n0 + lr8, a.N), line_ptr(a.w, n0 + 8 + lr8, a.N)};
    const uint8_t* wl2[2] = {SW ? line_ptr(a.w2, n0 + lr8, a.N) : nullptr, SW ? line_ptr(a.w2, n0 + 8 + lr8, a.N) : nullptr};
    const uint8_t* xl4[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) xl4[g][h] = line_ptr(a.x, g * 32 + h * 8 + lr8, a.M);
    // even pieces of both requests in one register (lanes 4-7 of each eight take the second request's even pieces from
    // four lanes below: DPP bank masks cover lanes in fours), odd pieces in another; fragment lane (r, kg) then reads
    // lane 8(r%8) + kg + 4(r/8) of each
    const int fidx = ((lrow & 7) * 8 + kg + 4 * (lrow >> 3)) * 4;
    auto frag = [&](const i32x4& r0, const i32x4& r1) __attribute__((always_inline)) -> i32x8 {
        i32x8 f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int ev = __builtin_amdgcn_update_dpp(r0[d], r1[d], 0x114 /* row_shr:4 */, 0xF, 0xA /* lanes 4-7, 12-15 */, false);
            const int od = __builtin_amdgcn_update_dpp(r1[d], r0[d], 0x104 /* row_shl:4 */, 0xF, 0x5 /* lanes 0-3, 8-11 */, false);
            f[d] = __builtin_amdgcn_ds_bpermute(fidx, ev);
            f[4 + d] = __builtin_amdgcn_ds_bpermute(fidx, od);
        }
        return f;
    };
    auto ldnt = [](const uint8_t* p) __attribute__((always_inline)) { return __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p)); };
    auto ld = [](const uint8_t* p) __attribute__((always_inline)) { return *reinterpret_cast<const i32x4*>(p); };
    f32x4 acc_lo[NG], acc_hi[NG], acc2_lo[SW ? NG : 1], acc2_hi[SW ? NG : 1];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        acc_lo[g] = acc_hi[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (SW) acc2_lo[g] = acc2_hi[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nks = a.K / 128;
    for (int ks0 = wave; ks0 < nks; ks0 += NW * CH) {
        i32x4 wr[CH][2], wr2[SW ? CH : 1][2];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ks = ks0 + c * NW;
            if (ks < nks) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    wr[c][h] = ldnt(wl[h] + (size_t)ks * 128);
                    if (SW) wr2[c][h] = ldnt(wl2[h] + (size_t)ks * 128);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g * 32 >= a.M) break;
            i32x4 xr[CH][4];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int ks = ks0 + c * NW;
                if (ks < nks) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) xr[c][h] = ld(xl4[g][h] + (size_t)ks * 128);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int ks = ks0 + c * NW;
                if (ks < nks) {
                    const i32x8 wf = frag(wr[c][0], wr[c][1]);
                    const i32x8 xl = frag(xr[c][0], xr[c][1]), xh = frag(xr[c][2], xr[c][3]);
                    acc_lo[g] = mfma_fp8(wf, xl, acc_lo[g]);
                    acc_hi[g] = mfma_fp8(wf, xh, acc_hi[g]);
                    if (SW) {
                        const i32x8 wf2 = frag(wr2[c][0], wr2[c][1]);
                        acc2_lo[g] = mfma_fp8(wf2, xl, acc2_lo[g]);
                        acc2_hi[g] = mfma_fp8(wf2, xh, acc2_hi[g]);
                    }
                }
            }
        }
    }
    const int tn = tid & 15, tmi = tid >> 4, nn = n0 + tn;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g * 32 >= a.M) break;                          // block-uniform
        if (g > 0) __syncthreads();                        // the previous group's sums have been read
        // C layout: col (m) = lane & 15, rows (n) = 4*(lane>>4) + reg
        *reinterpret_cast<f32x4*>(&part[wave][lrow][kg * 4]) = acc_lo[g];
        *reinterpret_cast<f32x4*>(&part[wave][16 + lrow][kg * 4]) = acc_hi[g];
        if (SW) {
            *reinterpret_cast<f32x4*>(&part2[wave][lrow][kg * 4]) = acc2_lo[g];
            *reinterpret_cast<f32x4*>(&part2[wave][16 + lrow][kg * 4]) = acc2_hi[g];
        }
        __syncthreads();
        const int tm = g * 32 + tmi;
        if (tm >= a.M || nn >= a.N) continue;
        const float* p = &part[0][0][0] + tmi * ROWS + tn;
        const float xs = a.xs[tm];
        float o;
        if (SW) {
            float gt = 0.f, u = 0.f;
            const float* p2 = &part2[0][0][0] + tmi * ROWS + tn;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                gt += p[w * 32 * ROWS];
                u += p2[w * 32 * ROWS];
            }
            gt = rbf(gt * (xs * a.ws[nn]));
            u = rbf(u * (xs * a.ws2[nn]));
            o = rbf(silu_fast(gt)) * u;
        } else {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += p[w * 32 * ROWS];
            o = fp8_finish<EPI>(s, xs * a.ws[nn], a, nn);
            if (RESID) o = bf2f(a.resid[(size_t)tm * a.N + nn]) + o;
        }
        if (OUT32) reinterpret_cast<float*>(a.y)[(size_t)tm * a.N + nn] = o;     // bf16-exact value as fp32 (fused decode consumer)
        else a.y[(size_t)tm * a.N + nn] = f2bf(o);
    }
}

template <int EPI, bool RESID, int NST, int WG = 2>
int launch_tiled_n(const Fp8Args& a, hipStream_t s) {
    constexpr int lds = NST * 2 * (WG * 64) * BKB;
    DH_MAX_LDS_ONCE((gemm_fp8_kernel<EPI, RESID, NST, WG>), lds);
    hipLaunchKernelGGL((gemm_fp8_kernel<EPI, RESID, NST, WG>), dim3(a.nb_n * a.nb_m), dim3(WG * WG * 64), lds, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch_tiled(Fp8Args a, hipStream_t s) {
    // 256 x 256 tiles on sixteen waves when they still give every CU two tiles to walk (prefill); the same fp32 chain per
    // output as the 128-tile (one accumulator, k ascending in steps of 128), so rows stay bit-identical across the choice
    const int nbm = cdiv(a.M, 256), nbn = EPI == DH_EPI_SWIGLU ? cdiv(a.N, 128) : cdiv(a.N, 256);
    a.gm = g_fp8_gm ? g_fp8_gm : 4;
    if (g_fp8_tile ? g_fp8_tile == 256 : nbm * nbn >= 512) {
        a.nb_m = nbm; a.nb_n = nbn;
        if (!g_fp8_gm && EPI == DH_EPI_SWIGLU) a.gm = 2;
        return a.resid ? launch_tiled_n<EPI, true, 2, 4>(a, s) : launch_tiled_n<EPI, false, 2, 4>(a, s);
    }
    // as gemm.hip: grids smaller than the chip walk K alone -> 4 stages, one block per CU; else 2 stages, two blocks
    if (a.nb_n * a.nb_m < 384) return a.resid ? launch_tiled_n<EPI, true, 4>(a, s) : launch_tiled_n<EPI, false, 4>(a, s);
    return a.resid ? launch_tiled_n<EPI, true, 2>(a, s) : launch_tiled_n<EPI, false, 2>(a, s);
}

template <int EPI, int NG>
int launch_skinny_ng(const Fp8Args& a, hipStream_t s) {
    dim3 grid(cdiv(a.N, ROWS)), block(512);
    if (a.resid) hipLaunchKernelGGL((gemm_fp8_skinny_kernel<EPI, true, false, NG>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_fp8_skinny_kernel<EPI, false, false, NG>), grid, block, 0, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch_skinny(const Fp8Args& a, hipStream_t s) {
    if (a.M <= 32) return launch_skinny_ng<EPI, 1>(a, s);
    if (a.M <= 64) return launch_skinny_ng<EPI, 2>(a, s);
    return launch_skinny_ng<EPI, 4>(a, s);
}

}  // namespace

extern "C" int dh_quant_rows_fp8(const dh_bf16* x, uint8_t* q, float* scale, int rows, int K, void* stream) {
    DH_CHECK(x && q && scale && rows >= 0 && K > 0 && K % 8 == 0, "dh_quant_rows_fp8: bad argument (K %% 8 must be 0)");
    if (rows == 0) return 0;
    if (rows <= 256)
        hipLaunchKernelGGL(quant_rows_fp8_block_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, q, scale, K);
    else
        hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, q, scale, rows, K);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_rmsnorm_quant_fp8(const dh_bf16* x, const dh_bf16* w, dh_bf16* xn_out, uint8_t* q, float* scale, int rows,
                                    int d, float eps, const uint8_t* row_tail, void* stream) {
    DH_CHECK(x && w && q && scale, "dh_rmsnorm_quant_fp8: null argument");
    DH_CHECK(rows >= 0 && d > 0 && d % 8 == 0 && d <= 8192, "dh_rmsnorm_quant_fp8: unsupported d=%d (need d %% 8 == 0, d <= 8192)", d);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (d <= 2048) hipLaunchKernelGGL((rmsnorm_quant_fp8_block_kernel<8>), dim3(rows), dim3(256), 0, s, x, w, xn_out, q, scale, d, eps, row_tail);
    else if (d <= 4096) hipLaunchKernelGGL((rmsnorm_quant_fp8_block_kernel<16>), dim3(rows), dim3(256), 0, s, x, w, xn_out, q, scale, d, eps, row_tail);
    else hipLaunchKernelGGL((rmsnorm_quant_fp8_block_kernel<32>), dim3(rows), dim3(256), 0, s, x, w, xn_out, q, scale, d, eps, row_tail);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_linear_fp8_f32(const uint8_t* xq, const float* x_scale, const uint8_t* wq, const float* w_scale, float* y32,
                                 int M, int N, int K, void* stream) {
    DH_CHECK(xq && x_scale && wq && w_scale && y32, "dh_linear_fp8_f32: null operand");
    DH_CHECK(M >= 1 && M <= FP8_STREAM_MAX_ROWS && N > 0 && K > 0 && K % 128 == 0, "dh_linear_fp8_f32: needs 1 <= M <= %d and K %% 128 == 0 (M=%d K=%d)",
             FP8_STREAM_MAX_ROWS, M, K);
    Fp8Args a{xq, wq, nullptr, reinterpret_cast<bf16_t*>(y32), x_scale, w_scale, nullptr, nullptr, nullptr, nullptr, M, N, K, 0, 0, 1};
    dim3 grid(cdiv(N, ROWS)), block(512);
    hipStream_t s = (hipStream_t)stream;
    // row groups as launch_skinny: the same kernel (and bits) as dh_linear_fp8's streaming form, fp32 store
    if (M <= 32) hipLaunchKernelGGL((gemm_fp8_skinny_kernel<DH_EPI_PLAIN, false, true, 1>), grid, block, 0, s, a);
    else if (M <= 64) hipLaunchKernelGGL((gemm_fp8_skinny_kernel<DH_EPI_PLAIN, false, true, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_fp8_skinny_kernel<DH_EPI_PLAIN, false, true, 4>), grid, block, 0, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_linear_fp8(const uint8_t* xq, const float* x_scale, const uint8_t* wq, const float* w_scale, dh_bf16* y,
                             int M, int N, int K, int epilogue, const uint8_t* w2q, const float* w2_scale,
                             const dh_bf16* vec_a, const dh_bf16* vec_b, const dh_bf16* resid, void* stream) {
    return dh_linear_fp8_ex(xq, x_scale, wq, w_scale, y, M, N, K, epilogue, w2q, w2_scale, vec_a, vec_b, resid, 0, stream);
}

// kernel: 0 = by row count (streaming up to FP8_STREAM_MAX_ROWS rows, tiled above), 1 = tiled, 2 = streaming.  The two
// kernels add the K products in different fp32 orders (tiled: one sequential accumulator; streaming: K dealt round-robin
// over 8 waves, partials summed through LDS), so a caller that wants a row's bits to be independent of what is packed
// with it pins the kernel by PHASE (the engine: prefill tiled, single-token steps streaming up to 128 rows).
extern "C" int dh_linear_fp8_ex(const uint8_t* xq, const float* x_scale, const uint8_t* wq, const float* w_scale, dh_bf16* y,
                                int M, int N, int K, int epilogue, const uint8_t* w2q, const float* w2_scale,
                                const dh_bf16* vec_a, const dh_bf16* vec_b, const dh_bf16* resid, int kernel, void* stream) {
    DH_CHECK(xq && x_scale && wq && w_scale && y, "dh_linear_fp8: null operand");
    DH_CHECK(kernel >= 0 && kernel <= 2, "dh_linear_fp8: kernel %d (0 auto, 1 tiled, 2 streaming)", kernel);
    DH_CHECK(kernel != 2 || M <= FP8_STREAM_MAX_ROWS, "dh_linear_fp8: the streaming kernel takes at most %d rows (M=%d)", FP8_STREAM_MAX_ROWS, M);
    DH_CHECK(M >= 0 && N > 0 && K > 0 && K % 128 == 0 && N % 4 == 0, "dh_linear_fp8: bad shape M=%d N=%d K=%d (K %% 128, N %% 4 must be 0)", M, N, K);
    DH_CHECK(epilogue == DH_EPI_PLAIN || epilogue == DH_EPI_SWIGLU || epilogue == DH_EPI_ADAPTER,
             "dh_linear_fp8: epilogue %d unsupported (LoRA is merged before quantisation)", epilogue);
    DH_CHECK(epilogue != DH_EPI_SWIGLU || (w2q && w2_scale && !resid), "dh_linear_fp8: SWIGLU needs w2/w2_scale and takes no residual");
    DH_CHECK(epilogue != DH_EPI_ADAPTER || (vec_a && vec_b), "dh_linear_fp8: ADAPTER needs scale/bias vectors");
    if (M == 0) return 0;
    Fp8Args a{xq, wq, w2q, y, x_scale, w_scale, w2_scale, vec_a, vec_b, resid, M, N, K, 0, 0, 1};
    hipStream_t s = (hipStream_t)stream;
    if (kernel == 2 || (kernel == 0 && M <= FP8_STREAM_MAX_ROWS)) {      // weight streaming: the decode step of up to four 32-row batches
        switch (epilogue) {
            case DH_EPI_PLAIN: return launch_skinny<DH_EPI_PLAIN>(a, s);
            case DH_EPI_SWIGLU: return launch_skinny<DH_EPI_SWIGLU>(a, s);
            default: return launch_skinny<DH_EPI_ADAPTER>(a, s);
        }
    }
    a.nb_m = cdiv(M, BT);
    a.nb_n = epilogue == DH_EPI_SWIGLU ? cdiv(N, 64) : cdiv(N, BT);
    switch (epilogue) {
        case DH_EPI_PLAIN: return launch_tiled<DH_EPI_PLAIN>(a, s);
        case DH_EPI_SWIGLU: return launch_tiled<DH_EPI_SWIGLU>(a, s);
        default: return launch_tiled<DH_EPI_ADAPTER>(a, s);
    }
}
