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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * ROWS;
    const int lrow = lane & 15, kg = lane >> 4;          // MFMA operand row / 8-element k group
    // SWIGLU: waves 0-3 stream fc_1, waves 4-7 stream fc_2 (4-way K split each)
    const bool second = (EPI == DH_EPI_SWIGLU) && wave >= NW / 2;
    const int kw = (EPI == DH_EPI_SWIGLU) ? (wave & (NW / 2 - 1)) : wave;
    const int kstride = (EPI == DH_EPI_SWIGLU) ? NW / 2 : NW;
    const bf16_t* wbase = second ? a.w2 : a.w;
    int n = n0 + lrow;
    n = n < a.N ? n : a.N - 1;
    const bf16_t* wrow = wbase + (size_t)n * a.K + kg * 8;
    int m_lo = lrow, m_hi = 16 + lrow;
    m_lo = m_lo < a.M ? m_lo : a.M - 1;
    m_hi = m_hi < a.M ? m_hi : a.M - 1;
    const bf16_t* xlo = a.x + (size_t)m_lo * a.K + kg * 8;
    const bf16_t* xhi = a.x + (size_t)m_hi * a.K + kg * 8;

    f32x4 acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = {0.f, 0.f, 0.f, 0.f};
    const int nks = a.K / 32;
    for (int ks0 = kw; ks0 < nks; ks0 += kstride * CH) {
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
    // C layout: col (m) = lane & 15, rows (n) = 4*(lane>>4) + reg  -> 4 consecutive n per lane
    *reinterpret_cast<f32x4*>(&part[wave][lrow][kg * 4]) = acc_lo;
    *reinterpret_cast<f32x4*>(&part[wave][16 + lrow][kg * 4]) = acc_hi;
    __syncthreads();

    const int tn = tid & 15, tm = tid >> 4;
    const int nn = n0 + tn;
    if (tm >= a.M || nn >= a.N) return;
    const float* p = &part[0][0][0] + tm * ROWS + tn;
    float o;
    if (EPI == DH_EPI_SWIGLU) {
        float g = 0.f, u = 0.f;
#pragma unroll
        for (int w = 0; w < NW / 2; ++w) {
            g += p[w * 32 * ROWS];
            u += p[(w + NW / 2) * 32 * ROWS];
        }
        g = rbf(g);
        u = rbf(u);
        o = rbf(g / (1.0f + expf(-g))) * u;
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

}  // namespace

int dh_linear_skinny(const GemmArgs& a, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case DH_EPI_PLAIN: return launch<DH_EPI_PLAIN>(a, s);
        case DH_EPI_LORA: return launch<DH_EPI_LORA>(a, s);
        case DH_EPI_SWIGLU: return launch<DH_EPI_SWIGLU>(a, s);
        case DH_EPI_ADAPTER: return launch<DH_EPI_ADAPTER>(a, s);
    }
    dh_set_error("dh_linear_bf16: unknown epilogue %d", epilogue);
    return 1;
}
