// Is v_dot2_f32_bf16 usable as an EXACT bf16 x bf16 -> f32 product and as a correctly rounded bf16 + bf16 -> f32 sum?
// Compares, over all 2^16 x (many) bf16 bit patterns, against v_mul_f32 / v_add_f32 on the expanded operands.
// hipcc --offload-arch=gfx950 -O3 tools/dot2_probe.hip -o tools/bin/dot2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float bf(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    float d;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float dot2_neg(uint32_t a, uint32_t b, float c) {   // the low product negated (neg_lo on src1): a . (-b)
    float d;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3 neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ bool same(float x, float y) { return __float_as_uint(x) == __float_as_uint(y) || (x != x && y != y); }

// counters: [0] product mismatches, [1] sum mismatches, [2] sum-with-C mismatches, [3] product via op_sel_hi + (0, c) table form,
// [4..7] the same restricted to operands that are normal finite numbers with normal finite results
__global__ void probe(unsigned long long* cnt, uint32_t* example, uint32_t salt) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;            // every bf16 pattern
    if (a >= 65536) return;
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t h = a * 2654435761u + salt;
    for (int it = 0; it < 4096; ++it) {
        h = h * 1664525u + 1013904223u;
        uint32_t b = (h >> 8) & 0xffffu;
        if (it & 1) {                                                    // half the time an exponent near a's (sums that cancel / carry)
            const int ea = (a >> 7) & 0xff;
            int eb = ea + (int)((h >> 3) & 31) - 16;
            eb = eb < 0 ? 0 : (eb > 255 ? 255 : eb);
            b = (b & 0x807fu) | ((uint32_t)eb << 7);
        }
        const uint32_t x = (h >> 4) & 0x7f7fu;                           // finite filler for the unused half
        const float fa = bf(a), fb = bf(b);
        const float pr = fa * fb, sr = fa + fb;
        const float p1 = dot2(a | (x << 16), b, 0.f);                    // (a, x) . (b, 0)
        const float s1 = dot2(a | (b << 16), 0x3f803f80u, 0.f);          // (a, b) . (1, 1)
        const float s2 = dot2(a | (x << 16), 0x00003f80u, fb);           // (a, x) . (1, 0) + b
        const float p2 = dot2_neg(a | (x << 16), b, 0.f);                // (a, x) . (-b, 0)
        auto fin = [](float v) { const uint32_t e = (__float_as_uint(v) >> 23) & 0xff; return e != 0 && e != 255; };
        const bool ok_in = fin(fa) && fin(fb);
        // the sums are only ever used ROUNDED TO BF16 (one rounding point of the reference): compare after v_cvt_pk_bf16_f32
        auto rb = [](float v) { uint32_t o; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(o) : "v"(v)); return __uint_as_float(o << 16); };
        const bool r[4] = {same(p1, pr) || (pr == 0.f && p1 == 0.f), same(rb(s1), rb(sr)) || (sr == 0.f && s1 == 0.f),
                           same(rb(s2), rb(sr)) || (sr == 0.f && s2 == 0.f), same(p2, -pr) || (pr == 0.f && p2 == 0.f)};
        if (pr == 0.f && p1 == 0.f) { }   // (the sign of an exact zero is not compared: -0 + 0 = +0 inside the dot)
        const bool okr[4] = {ok_in && fin(pr), ok_in && (fin(sr) || sr == 0.f), ok_in && (fin(sr) || sr == 0.f), ok_in && fin(pr)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!r[k]) {
                c[k]++;
                if (okr[k]) {
                    c[4 + k]++;
                    if (atomicAdd(&example[k * 4], 1u) == 0) { example[k * 4 + 1] = a; example[k * 4 + 2] = b; example[k * 4 + 3] = __float_as_uint(k == 0 ? p1 : k == 1 ? s1 : k == 2 ? s2 : p2); }
                }
            }
        }
    }
    for (int k = 0; k < 8; ++k) if (c[k]) atomicAdd(&cnt[k], c[k]);
}

int main() {
    unsigned long long* d; uint32_t* ex;
    hipMalloc(&d, 64); hipMalloc(&ex, 64);
    hipMemset(d, 0, 64); hipMemset(ex, 0, 64);
    for (uint32_t salt = 0; salt < 8; ++salt) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d, ex, salt * 977u);
    hipDeviceSynchronize();
    unsigned long long h[8]; uint32_t e[16];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost); hipMemcpy(e, ex, 64, hipMemcpyDeviceToHost);
    const char* nm[4] = {"product  (a,x).(b,0)           vs v_mul_f32", "bf16(sum (a,b).(1,1))         vs bf16(v_add_f32)", "bf16(sum (a,x).(1,0) + C=b)   vs bf16(v_add_f32)",
                         "product  (a,x).(-b,0) neg_lo   vs -(v_mul_f32)"};
    const double total = 65536.0 * 4096 * 8;
    for (int k = 0; k < 4; ++k) {
        printf("%s : %llu of %.0f differ (%llu with normal finite operands and results)", nm[k], h[k], total, h[4 + k]);
        if (h[4 + k]) printf("   e.g. a=%04x b=%04x got %08x", e[k * 4 + 1], e[k * 4 + 2], e[k * 4 + 3]);
        printf("\n");
    }
    return 0;
}
