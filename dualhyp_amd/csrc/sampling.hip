// The tail of the decode loop (generate/base.py:62-80) fused into one kernel per step:
// temperature -> top-k crop -> softmax -> draw -> append token -> EOS flag.  One 1024-thread
// block per sequence; nothing returns to the host.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t bf16_key(bf16_t v) {
    // monotone map bf16 bits -> uint16 key (larger value = larger key)
    return (v & 0x8000u) ? (uint32_t)(~v & 0xFFFFu) : (uint32_t)(v | 0x8000u);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr int NT = 1024;

__global__ __launch_bounds__(NT) void sample_kernel(const bf16_t* __restrict__ logits, int vocab,
                                                    int64_t* __restrict__ tokens, int tok_ld,
                                                    int32_t* __restrict__ length, int32_t* __restrict__ done,
                                                    float temperature, int top_k, int64_t eos_id, uint64_t seed,
                                                    int step_arg, const int32_t* __restrict__ step_dev) {
    const int seq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (done[seq]) return;
    const int step = step_dev ? *step_dev : step_arg;   // device counter keeps a captured graph replayable
    const bf16_t* lg = logits + (size_t)seq * vocab;
    __shared__ float s_f[NT / 64];
    __shared__ int s_i[NT / 64];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_sel[2];

    // l = bf16(logit / temperature)   (generate/base.py:62, bf16 tensor / python float)
    auto scaled = [&](int i) -> bf16_t { return f2bf(bf2f(lg[i]) / temperature); };

    int choice = 0;
    if (top_k == 1) {
        // arg-max, lowest index among equal maxima (NaN never wins: it is not > anything)
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if ((vocab & 7) == 0 && (reinterpret_cast<uintptr_t>(lg) & 15) == 0) {
            // 16-byte loads, all of a thread's requests in flight before the first compare (the 2-byte strided loop took
            // 15.8 us for a 32-row step's 2 MB of logits: one dependent compare chain behind 32 small loads per thread);
            // the tie rule carries the index, so the visiting order is free
            const uint4* lg4 = reinterpret_cast<const uint4*>(lg);
            const int n4 = vocab >> 3;
            for (int c0 = tid; c0 < n4; c0 += 4 * NT) {
                uint4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = c0 + u * NT < n4 ? lg4[c0 + u * NT] : uint4{0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c0 + u * NT >= n4) break;
                    const bf16_t* e8 = reinterpret_cast<const bf16_t*>(&q[u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int i = (c0 + u * NT) * 8 + e;
                        const float v = bf2f(f2bf(bf2f(e8[e]) / temperature));
                        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
                    }
                }
            }
        } else {
        for (int i = tid; i < vocab; i += NT) {
            const float v = bf2f(scaled(i));
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { s_f[wave] = best; s_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NT / 64; ++w)
                if (s_f[w] > best || (s_f[w] == best && s_i[w] < bi)) { best = s_f[w]; bi = s_i[w]; }
            s_i[0] = bi == 0x7fffffff ? 0 : bi;
        }
        __syncthreads();
        choice = s_i[0];
    } else {
        // ---- threshold = k-th largest scaled logit: two-pass radix select on the 16-bit keys
        uint32_t thr_key = 0;
        if (top_k > 0 && top_k < vocab) {
            uint32_t prefix = 0;
            int need = top_k;
            for (int pass = 0; pass < 2; ++pass) {
                for (int i = tid; i < 256; i += NT) s_hist[i] = 0;
                __syncthreads();
                for (int i = tid; i < vocab; i += NT) {
                    const uint32_t k = bf16_key(scaled(i));
                    if (pass == 0) atomicAdd(&s_hist[k >> 8], 1u);
                    else if ((k >> 8) == prefix) atomicAdd(&s_hist[k & 255], 1u);
                }
                __syncthreads();
                if (tid == 0) {
                    int b = 255, acc = 0;
                    for (; b > 0; --b) {
                        if (acc + (int)s_hist[b] >= need) break;
                        acc += s_hist[b];
                    }
                    s_sel[0] = b;
                    s_sel[1] = need - acc;
                }
                __syncthreads();
                if (pass == 0) prefix = s_sel[0]; else thr_key = (prefix << 8) | s_sel[0];
                need = s_sel[1];
                __syncthreads();
            }
        }
        // ---- softmax over kept entries (fp32), then inverse-CDF draw in index order
        float mx = -INFINITY;
        for (int i = tid; i < vocab; i += NT) {
            const bf16_t v = scaled(i);
            if (bf16_key(v) >= thr_key) mx = fmaxf(mx, bf2f(v));
        }
        mx = wave_max(mx);
        if (lane == 0) s_f[wave] = mx;
        __syncthreads();
        mx = s_f[0];
        for (int w = 1; w < NT / 64; ++w) mx = fmaxf(mx, s_f[w]);
        __syncthreads();
        // contiguous slab per thread so the CDF is in index order
        const int per = (vocab + NT - 1) / NT;
        const int lo = tid * per, hi = min(vocab, lo + per);
        float mine = 0.f;
        for (int i = lo; i < hi; ++i) {
            const bf16_t v = scaled(i);
            if (bf16_key(v) >= thr_key) mine += __expf(bf2f(v) - mx);
        }
        // block exclusive scan of `mine` (wave scan + wave totals)
        float incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_f[wave] = incl;
        __syncthreads();
        float base = 0.f, total = 0.f;
        for (int w = 0; w < NT / 64; ++w) {
            if (w < wave) base += s_f[w];
            total += s_f[w];
        }
        const float excl = base + incl - mine;
        const uint64_t h = mix64(seed ^ mix64(((uint64_t)step << 32) | (uint32_t)seq));
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f) * total;   // in [0, total)
        if (tid == 0) s_i[0] = -1;
        __syncthreads();
        if (mine > 0.f && u >= excl && u < excl + mine) {
            float c = excl;
            int pick = lo;
            for (int i = lo; i < hi; ++i) {
                const bf16_t v = scaled(i);
                if (bf16_key(v) >= thr_key) {
                    pick = i;
                    c += __expf(bf2f(v) - mx);
                    if (u < c) break;
                }
            }
            atomicMax(&s_i[0], pick);
        }
        __syncthreads();
        if (tid == 0 && s_i[0] < 0) {   // rounding left u past the last bin: take the arg-max
            int bi = 0; float best = -INFINITY;
            for (int i = 0; i < vocab; ++i) { const float v = bf2f(scaled(i)); if (v > best) { best = v; bi = i; } }
            s_i[0] = bi;
        }
        __syncthreads();
        choice = s_i[0];
    }
    if (tid == 0) {
        const int n = length[seq];
        if (n < tok_ld) {
            tokens[(size_t)seq * tok_ld + n] = choice;
            length[seq] = n + 1;
        }
        if (eos_id >= 0 && choice == eos_id) done[seq] = 1;
        else if (n + 1 >= tok_ld) done[seq] = 2;   // buffer full
    }
}

}  // namespace

int dh_sample_impl(const dh_bf16* logits, int vocab, int64_t* tokens, int tok_ld, int32_t* length, int32_t* done,
                   int n_seq, float temperature, int top_k, int64_t eos_id, uint64_t seed, int step,
                   const int32_t* step_dev, void* stream) {
    DH_CHECK(vocab > 0 && tok_ld > 0 && n_seq >= 0, "dh_sample_bf16: bad shape");
    DH_CHECK(temperature > 0.f, "dh_sample_bf16: temperature must be > 0");
    DH_CHECK(top_k >= 0, "dh_sample_bf16: top_k must be >= 0 (0 = no crop)");
    if (n_seq == 0) return 0;
    hipLaunchKernelGGL(sample_kernel, dim3(n_seq), dim3(NT), 0, (hipStream_t)stream, logits, vocab, tokens, tok_ld,
                       length, done, temperature, top_k, eos_id, seed, step, step_dev);
    DH_LAUNCH_CHECK();
    return 0;
}

extern "C" int dh_sample_bf16(const dh_bf16* logits, int vocab, int64_t* tokens, int tok_ld, int32_t* length,
                              int32_t* done, int n_seq, float temperature, int top_k, int64_t eos_id, uint64_t seed,
                              int step, void* stream) {
    return dh_sample_impl(logits, vocab, tokens, tok_ld, length, done, n_seq, temperature, top_k, eos_id, seed, step,
                          nullptr, stream);
}
