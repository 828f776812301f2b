// Token-level cross entropy with ignore_index = -1 (ger/utils.py:424-463, F.cross_entropy under the
// reference's bf16 autocast: the logits are upcast to fp32 and log-softmax runs in fp32).
//   forward : loss[r] = logsumexp(logits[r,:]) - logits[r, target[r]]   (0 for ignored rows), lse[r]
//   backward: dlogits[r, c] = g * (exp(logits[r,c] - lse[r]) - [c == target[r]])   (0 for ignored rows)
// One block per row; the row (V = 32000: 64 KB in bf16) is read once, online max/sum per thread,
// block reduce through LDS.  HBM-bound: V*2 bytes per row forward, V*4 backward.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ targets,
                                                     float* __restrict__ loss, float* __restrict__ lse, int V) {
    __shared__ float sm[4], ss[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* x = logits + (size_t)row * V;
    float m = -INFINITY, s = 0.f;
    for (int c = tid; c < V; c += 256) {
        const float v = ldf<T>(x + c);
        if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }
        else s += __expf(v - m);
    }
    const float wm = wave_max(m);
    s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));
    if (lane == 0) { sm[wave] = wm; ss[wave] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), S = 0.f;
        for (int w = 0; w < 4; ++w) S += sm[w] == -INFINITY ? 0.f : ss[w] * __expf(sm[w] - M);
        const float l = M + logf(S);
        lse[row] = l;
        const int64_t t = targets[row];
        loss[row] = (t >= 0 && t < V) ? l - ldf<T>(x + t) : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ targets,
                                                     const float* __restrict__ lse, const float* __restrict__ grow,
                                                     T* __restrict__ dlogits, int V) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const T* x = logits + (size_t)row * V;
    T* dx = dlogits + (size_t)row * V;
    const int64_t t = targets[row];
    const bool valid = t >= 0 && t < V;
    const float g = valid ? grow[row] : 0.f, l = lse[row];
    for (int c = tid; c < V; c += 256) {
        float d = 0.f;
        if (valid) d = g * (__expf(ldf<T>(x + c) - l) - (c == t ? 1.f : 0.f));
        stf<T>(dx + c, d);
    }
}

}  // namespace

extern "C" int dh_cross_entropy_fwd(const void* logits, int is_f32, const int64_t* targets, float* loss, float* lse,
                                    int rows, int vocab, void* stream) {
    DH_CHECK(logits && targets && loss && lse && rows >= 0 && vocab > 0, "dh_cross_entropy_fwd: bad argument");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (is_f32)
        hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)logits, targets, loss, lse, vocab);
    else
        hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, targets, loss, lse, vocab);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_cross_entropy_bwd(const void* logits, int is_f32, const int64_t* targets, const float* lse,
                                    const float* grad_row, void* dlogits, int rows, int vocab, void* stream) {
    DH_CHECK(logits && targets && lse && grad_row && dlogits && rows >= 0 && vocab > 0, "dh_cross_entropy_bwd: bad argument");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (is_f32)
        hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)logits, targets, lse, grad_row,
                           (float*)dlogits, vocab);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, targets, lse, grad_row,
                           (bf16_t*)dlogits, vocab);
    DH_LAUNCH_CHECK();
    return 0;
}
