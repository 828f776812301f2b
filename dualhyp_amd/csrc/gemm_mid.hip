// y[M, N] = epilogue(x · W^T) for the decode step, M <= 256 rows (one 32-sequence batch, or several
// batches sharing one launch: generate.py generate_gang).  HBM-bound weight streaming; x is staged
// ONCE per block in LDS (double-buffered K-slices) and shared by the 8 waves instead of being fetched
// per wave from L2 as gemm_skinny.hip does, W goes global -> VGPR three slices ahead.
//   block : 512 threads = 8 waves = RS row sets x KQ K-parts; a wave owns 16 W rows (SWIGLU: of fc_1
//           AND fc_2) and KSL/KQ of the 8 k-steps of every slice, so a block covers only 16*RS output
//           columns and even N = 5632 puts 176 blocks on the chip (SWIGLU: RS 2 x KQ 4, else 4 x 2)
//   W     : ring of PD+1 slice fragments per wave, 12 KiB in flight per wave (96 KiB per block)
//   x     : rows [m0, m0 + 32*NG) of the slice in LDS, 528-B padded rows
//   MFMA  : v_mfma_f32_16x16x32_bf16, C^T[16 n][16 m] += W[16 n][32 k] · x[16 m][32 k]^T
//   reduce: the KQ partial accumulators meet in LDS (the x buffers are free by then), summed in
//           K-part order by the first wave of each row set, which also runs the epilogue
//   grid  : (N / (16*RS), ceil(M / (32*NG)))
// The summation order of an output is fixed by K alone (KQ interleaved chains, k ascending in each),
// whatever M and NG: a row's result does not depend on how many other rows ride along (batch
// invariance, same contract as the other decode kernels).
#include "common.h"
#include "gemm.h"

namespace {

constexpr int KSL = 8;             // k-steps per slice
constexpr int XS = KSL * 64 + 16;  // padded LDS row stride (bytes)
constexpr int PD = 3;              // W slices requested ahead of the MFMAs
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (arrays of HIP's uint4 struct are not promoted to registers)

template <int EPI>
struct MidShape {
    static constexpr bool SW = EPI == DH_EPI_SWIGLU;
    static constexpr int NM = SW ? 2 : 1;      // matrices streamed per wave
    static constexpr int KQ = SW ? 4 : 2;      // K-parts per block
    static constexpr int RS = 8 / KQ;          // 16-row sets per block
    static constexpr int KPW = KSL / KQ;       // k-steps per wave per slice
};

template <int EPI, int NG>
__global__ __launch_bounds__(512) void gemm_mid_kernel(GemmArgs a) {
    using S = MidShape<EPI>;
    constexpr bool SW = S::SW;
    constexpr int NM = S::NM, KQ = S::KQ, RS = S::RS, KPW = S::KPW, RING = PD + 1;
    constexpr int CPT = NG * 32 * KSL * 4 / 512;          // 16-B x chunks per thread per slice
    constexpr int NI = NG * NM * 2;                       // accumulator tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][NG*32][XS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int rs = wave % RS, kq = wave / RS;
    const int n0 = blockIdx.x * (16 * RS) + rs * 16;
    const int m0 = blockIdx.y * (NG * 32);
    int n = n0 + lrow;
    n = n < a.N ? n : a.N - 1;
    const bf16_t* w1 = a.w + (size_t)n * a.K + kg * 8;
    const bf16_t* w2 = SW ? a.w2 + (size_t)n * a.K + kg * 8 : nullptr;
    const int nks = a.K / 32, nsl = (nks + KSL - 1) / KSL;

    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wr[RING][NM][KPW];
    u32x4 xs[CPT];
    auto load_w = [&](bf16x8 (&wf)[NM][KPW], int s) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < KPW; ++c) {
            const int ks = s * KSL + kq * KPW + c;
            if (ks < nks) {
                wf[0][c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + ks * 32));
                if (SW) wf[NM - 1][c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w2 + ks * 32));
            }
        }
    };
    auto load_x = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int chunk = tid + i * 512;
            const int row = chunk / (KSL * 4), col = chunk % (KSL * 4);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            // k-steps past the end of K re-read the last valid one (never multiplied)
            const int ks = min(s * KSL + col / 4, nks - 1);
            xs[i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)m * a.K + ks * 32 + (col & 3) * 8);
        }
    };
    auto store_x = [&](int buf) __attribute__((always_inline)) {
        char* xb = smem + buf * (NG * 32 * XS);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int chunk = tid + i * 512;
            const int row = chunk / (KSL * 4), col = chunk % (KSL * 4);
            *reinterpret_cast<u32x4*>(xb + row * XS + col * 16) = xs[i];
        }
    };
    auto body = [&](bf16x8 (&cur)[NM][KPW], bf16x8 (&ahead)[NM][KPW], int s) __attribute__((always_inline)) {
        if (s + PD < nsl) load_w(ahead, s + PD);
        const bool has_next = s + 1 < nsl;
        if (has_next) load_x(s + 1);
        const char* xb = smem + (s & 1) * (NG * 32 * XS);
#pragma unroll
        for (int c = 0; c < KPW; ++c) {
            const int kc = kq * KPW + c;
            if (s * KSL + kc < nks) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (m0 + g * 32 < a.M) {
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xb + (g * 32 + lrow) * XS + kc * 64 + kg * 16);
                        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xb + (g * 32 + 16 + lrow) * XS + kc * 64 + kg * 16);
#pragma unroll
                        for (int q = 0; q < NM; ++q) {
                            acc[(g * NM + q) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[q][c], xl, acc[(g * NM + q) * 2], 0, 0, 0);
                            acc[(g * NM + q) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[q][c], xh, acc[(g * NM + q) * 2 + 1], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // buffer (s+1)&1 was last read in body(s-1), which ended with a barrier
        if (has_next) store_x((s + 1) & 1);
        __syncthreads();
    };

#pragma unroll
    for (int p = 0; p < PD; ++p)
        if (p < nsl) load_w(wr[p], p);
    load_x(0);
    store_x(0);
    __syncthreads();
    for (int s0 = 0; s0 < nsl; s0 += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r)
            if (s0 + r < nsl) body(wr[r], wr[(r + PD) % RING], s0 + r);
    }

    // ---- the KQ partial tiles of a row set meet in LDS (every x buffer read ended at the last barrier)
    f32x4* red = reinterpret_cast<f32x4*>(smem);           // [wave][NI][64 lanes]
    if (kq > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) red[(wave * NI + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (kq > 0) return;
#pragma unroll
    for (int p = 1; p < KQ; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] += red[((p * RS + rs) * NI + i) * 64 + lane];

    // C layout: col (m) = lane & 15, rows (n) = 4*(lane>>4) + reg -> 4 consecutive n per lane
    const int nn = n0 + kg * 4;
    if (nn >= a.N) return;
    float va[4] = {1.f, 1.f, 1.f, 1.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == DH_EPI_ADAPTER) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            va[r] = bf2f(a.vec_a[nn + r]);
            vb[r] = bf2f(a.vec_b[nn + r]);
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + g * 32 + 16 * h + lrow;
            if (m >= a.M) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (SW) {
                    const float gt = rbf(acc[(g * NM) * 2 + h][r]), up = rbf(acc[(g * NM + NM - 1) * 2 + h][r]);
                    o[r] = rbf(gt / (1.0f + expf(-gt))) * up;
                } else {
                    o[r] = rbf(acc[(g * NM) * 2 + h][r]);
                    if (EPI == DH_EPI_ADAPTER) o[r] = rbf(va[r] * rbf(o[r] + vb[r]));
                }
            }
            if (!SW && a.resid != nullptr) {
                const uint2 rv = *reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.N + nn);
                const bf16_t* rp = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = bf2f(rp[r]) + o[r];
            }
            const uint2 pk = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            *reinterpret_cast<uint2*>(a.y + (size_t)m * a.N + nn) = pk;
        }
}

template <int EPI, int NG>
int launch_mid(const GemmArgs& a, hipStream_t s) {
    constexpr int lds = 2 * NG * 32 * XS;
    static bool attr = false;
    if (!attr && lds > 48 * 1024) {
        DH_HIP(hipFuncSetAttribute((const void*)gemm_mid_kernel<EPI, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    dim3 grid(cdiv(a.N, 16 * MidShape<EPI>::RS), cdiv(a.M, NG * 32)), block(512);
    hipLaunchKernelGGL((gemm_mid_kernel<EPI, NG>), grid, block, lds, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch_ng(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 32) return launch_mid<EPI, 1>(a, s);
    if (a.M <= 64) return launch_mid<EPI, 2>(a, s);
    return launch_mid<EPI, 4>(a, s);
}

}  // namespace

int g_mid = 1;

bool dh_linear_mid_ok(const GemmArgs& a, int epilogue) {
    return g_mid != 0 && a.K % 32 == 0 && a.N % 16 == 0 && a.M <= 256 &&
           (epilogue == DH_EPI_PLAIN || epilogue == DH_EPI_SWIGLU || epilogue == DH_EPI_ADAPTER);
}

int dh_linear_mid(const GemmArgs& a, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case DH_EPI_PLAIN: return launch_ng<DH_EPI_PLAIN>(a, s);
        case DH_EPI_SWIGLU: return launch_ng<DH_EPI_SWIGLU>(a, s);
        case DH_EPI_ADAPTER: return launch_ng<DH_EPI_ADAPTER>(a, s);
    }
    dh_set_error("dh_linear_mid: unsupported epilogue %d", epilogue);
    return 1;
}
