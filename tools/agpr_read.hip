// What does reading an accumulator cost one wave per SIMD?  256 v_accvgpr_read of registers (a) written by v_accvgpr_write,
// (b) written by asm MFMAs as the 4-wave GEMM does, in the read / cvt_pk pair pattern of its epilogue.  s_memtime cycles per read.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink) {
    f32x4 acc[56];
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3f80 + threadIdx.x + i); fb[i] = (short)(0x3f00 + i); }
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = f32x4{0.f, 1.f, 2.f, (float)i};
    if (MODE >= 1) {
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int i = 0; i < 56; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(fa), "v"(fb));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 56; ++i) asm volatile("" : "+a"(acc[i]));
    if (MODE == 2) __builtin_amdgcn_s_sleep(127);                     // let whatever is pending drain
    uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 56; ++i) {
        float t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[i][e]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %0" : "+v"(r[(4 * i + e) & 7]) : "v"(t));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += __uint_as_float(r[i]);
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 256 * 256 * 4);
    const char* nm[3] = {"written by VALU (initialisation)", "written by asm MFMAs, read right behind s_nop 15 x2", "written by asm MFMAs, read after s_sleep 127"};
    auto run = [&](auto kern, int m) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, sink);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, sink);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double a = 0; for (int i = 0; i < 256; ++i) a += h[i];
        printf("%-56s %7.1f cycles per (v_accvgpr_read + v_cvt_pk) pair, 224 pairs\n", nm[m], a / 256 / 224.0);
    };
    run(k<0>, 0); run(k<1>, 1); run(k<2>, 2);
    return 0;
}
