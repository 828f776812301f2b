// RelPrompt reliability predictor (ger/relprompt.py:126-147, NoiseMaskClassifier):
//   Conv1d(C -> H, k 3, pad 1) -> ReLU -> Conv1d(H -> H, k 3, pad 1) -> ReLU -> AvgPool1d(pool, ceil) -> Linear(H -> 3)
// The two convolutions run on the MFMA GEMM (dh_linear_bf16) over an im2col matrix whose extra column of
// ones carries the bias into the fp32 accumulation (one rounding, as the reference's conv); the kernels
// here build that matrix (optionally through the ReLU of the previous layer) and finish the head.
#include "common.h"

namespace {

// out[(b*T + t)][dk*C + c] = f(x[b][t + dk - 1][c]) (0 outside the sequence); out[..][3C] = 1; zero pad to ld
__global__ __launch_bounds__(256) void im2col3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int T,
                                                      int C, int ld, int relu) {
    const int row = blockIdx.x, b = row / T, t = row % T;
    bf16_t* o = out + (size_t)row * ld;
    for (int c8 = threadIdx.x; c8 < ld / 8; c8 += 256) {
        const int col = c8 * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (col < 3 * C) {
            const int dk = col / C, c = col - dk * C, ts = t + dk - 1;     // C % 8 == 0: a chunk never straddles taps
            if (ts >= 0 && ts < T) {
                v = *reinterpret_cast<const uint4*>(x + ((size_t)b * T + ts) * C + c);
                if (relu) {
                    bf16_t* p = reinterpret_cast<bf16_t*>(&v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) p[e] = (p[e] & 0x8000) ? (bf16_t)0 : p[e];   // max(x, 0); -0 -> +0
                }
            }
        } else if (col == 3 * C) {
            reinterpret_cast<bf16_t*>(&v)[0] = 0x3F80;   // 1.0: the bias column
        }
        *reinterpret_cast<uint4*>(o + col) = v;
    }
}

// logits[b][p][j] = bf16( sum_h bf16(mean_{t in window p} relu(h[b][t][h])) * w[j][h] + bias[j] ), ceil-mode windows
__global__ __launch_bounds__(256) void pool_head_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ bias, bf16_t* __restrict__ out, int B, int T,
                                                        int H, int pool, int P) {
    __shared__ float red[3][4];
    const int bp = blockIdx.x, b = bp / P, p = bp % P;
    const int t0 = p * pool, t1 = min(t0 + pool, T);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int c = threadIdx.x; c < H; c += 256) {
        float s = 0.f;
        for (int t = t0; t < t1; ++t) s += fmaxf(bf2f(h[((size_t)b * T + t) * H + c]), 0.f);
        const float m = rbf(s / (float)(t1 - t0));
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = fmaf(m, bf2f(w[(size_t)j * H + c]), acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float v = wave_sum(acc[j]);
        if ((threadIdx.x & 63) == 0) red[j][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int j = threadIdx.x;
        out[(size_t)bp * 3 + j] = f2bf(red[j][0] + red[j][1] + red[j][2] + red[j][3] + bf2f(bias[j]));
    }
}

// ---- training (finetune/relprompt.py:356-387: the mask cross entropy trains conv1 / conv2 / classifier) ----
// backward of ReLU -> AvgPool1d(pool, ceil) -> Linear(H, 3) for one (b, window p) per block:
//   pooled[bp][c] = bf16(mean_t relu(h[b][t][c]))                  (recomputed exactly as the forward does)
//   dh[b][t][c]   = h[b][t][c] > 0 ? bf16((sum_j dl[bp][j] w[j][c]) / len) : 0
__global__ __launch_bounds__(256) void pool_head_bwd_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ dl, bf16_t* __restrict__ dh,
                                                            bf16_t* __restrict__ pooled, int B, int T, int H, int pool, int P) {
    const int bp = blockIdx.x, b = bp / P, p = bp % P;
    const int t0 = p * pool, t1 = min(t0 + pool, T);
    const float inv = 1.f / (float)(t1 - t0);
    const float d0 = dl[(size_t)bp * 3], d1 = dl[(size_t)bp * 3 + 1], d2 = dl[(size_t)bp * 3 + 2];
    for (int c = threadIdx.x; c < H; c += 256) {
        float s = 0.f;
        for (int t = t0; t < t1; ++t) s += fmaxf(bf2f(h[((size_t)b * T + t) * H + c]), 0.f);
        pooled[(size_t)bp * H + c] = f2bf(s / (float)(t1 - t0));
        const bf16_t g = f2bf((d0 * bf2f(w[c]) + d1 * bf2f(w[(size_t)H + c]) + d2 * bf2f(w[(size_t)2 * H + c])) * inv);
        for (int t = t0; t < t1; ++t) {
            const size_t i = ((size_t)b * T + t) * H + c;
            dh[i] = bf2f(h[i]) > 0.f ? g : (bf16_t)0;
        }
    }
}

// backward of the k=3 im2col, fused with the ReLU (and dropout mask) that sits in front of it:
//   dx[b][t][c] = [pre[b][t][c] > 0] * mask[b][t][c] * sum_dk dcol[b][t + 1 - dk][dk*C + c]     (rows outside 0..T-1 dropped)
__global__ __launch_bounds__(256) void col2im3_kernel(const bf16_t* __restrict__ dcol, const bf16_t* __restrict__ pre,
                                                      const bf16_t* __restrict__ mask, bf16_t* __restrict__ dx, int B, int T,
                                                      int C, int ld) {
    const int row = blockIdx.x, b = row / T, t = row % T;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int dk = 0; dk < 3; ++dk) {
            const int tr = t + 1 - dk;
            if (tr >= 0 && tr < T) s += bf2f(dcol[((size_t)b * T + tr) * ld + dk * C + c]);
        }
        const size_t i = (size_t)row * C + c;
        if (pre && !(bf2f(pre[i]) > 0.f)) s = 0.f;
        if (mask) s *= bf2f(mask[i]);
        dx[i] = f2bf(s);
    }
}

}  // namespace

extern "C" int dh_pool_head_bwd_bf16(const dh_bf16* h, const dh_bf16* w, const float* dlogits, dh_bf16* dh, dh_bf16* pooled,
                                     int B, int T, int H, int pool, void* stream) {
    DH_CHECK(h && w && dlogits && dh && pooled && B > 0 && T > 0 && H > 0 && pool > 0, "dh_pool_head_bwd_bf16: bad argument");
    const int P = (T + pool - 1) / pool;
    hipLaunchKernelGGL(pool_head_bwd_kernel, dim3(B * P), dim3(256), 0, (hipStream_t)stream, h, w, dlogits, dh, pooled, B, T, H, pool, P);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_col2im3_bf16(const dh_bf16* dcol, const dh_bf16* pre, const dh_bf16* mask, dh_bf16* dx, int B, int T, int C,
                               int ld, void* stream) {
    DH_CHECK(dcol && dx && B > 0 && T > 0 && C > 0 && ld >= 3 * C, "dh_col2im3_bf16: bad argument");
    hipLaunchKernelGGL(col2im3_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, dcol, pre, mask, dx, B, T, C, ld);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_im2col3_bf16(const dh_bf16* x, dh_bf16* out, int B, int T, int C, int ld, int relu, void* stream) {
    DH_CHECK(x && out && B > 0 && T > 0 && C > 0, "dh_im2col3_bf16: bad argument");
    DH_CHECK(C % 8 == 0 && ld % 8 == 0 && ld >= 3 * C + 8, "dh_im2col3_bf16: C and ld must be multiples of 8 with ld >= 3C + 8");
    hipLaunchKernelGGL(im2col3_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, x, out, B, T, C, ld, relu);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_pool_head_bf16(const dh_bf16* h, const dh_bf16* w, const dh_bf16* bias, dh_bf16* out, int B, int T, int H,
                                 int pool, void* stream) {
    DH_CHECK(h && w && bias && out && B > 0 && T > 0 && H > 0 && pool > 0, "dh_pool_head_bf16: bad argument");
    const int P = (T + pool - 1) / pool;
    hipLaunchKernelGGL(pool_head_kernel, dim3(B * P), dim3(256), 0, (hipStream_t)stream, h, w, bias, out, B, T, H, pool, P);
    DH_LAUNCH_CHECK();
    return 0;
}
