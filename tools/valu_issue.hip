// Issue cost of single VALU instructions for ONE wave per SIMD (the regime of the 4-wave GEMM's epilogue): cycles per instruction of
// an unrolled stream of independent instructions, by s_memtime.  hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/bin/valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, int iters) {
    float v[16], w[16];
    uint32_t u[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16], q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 0.001f + i; w[i] = 1.0f + i * 0.01f; u[i] = threadIdx.x * 77u + i; p[i] = f2{v[i], w[i]}; q[i] = f2{w[i], v[i]}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
                if (OP == 5) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
                if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 7) asm volatile("v_dot2_f32_bf16 %0, %0, %1, 0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 8) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]), "v"(u[(i + 3) & 15]));
                if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 10) asm volatile("v_mul_f32 %0, %0, %0\n\tv_mul_f32 %0, %0, %0" : "+v"(v[i]));      // dependent pair
                if (OP == 11) asm volatile("v_bfe_u32 %0, %1, 16, 1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 12) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]), "v"(u[(i + 3) & 15]));
                if (OP == 13) asm volatile("v_mul_f32 %0, %0, %1\n\tv_lshlrev_b32 %2, 16, %2" : "+v"(v[i]), "+v"(w[i]), "+v"(u[i]));
                if (OP == 14) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(q[i]));
                if (OP == 15) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[i]) : "a"(w[i]));
                if (OP == 16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 8) & 15]));
                if (OP == 17) asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_mul_f32 %2, %2, %2" : "=v"(u[i]) : "a"(w[i]), "v"(v[i]));
                if (OP == 19) asm volatile("v_mul_f32 %0, %0, %1\n\ts_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\tv_mul_f32 %0, %0, %1\n\t1:" : "+v"(v[i]) : "v"(w[i]) : "scc");
                if (OP == 20) asm volatile("v_mul_f32 %0, %0, %1\n\ts_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\tv_mul_f32 %0, %0, %1\n\t1:" : "+v"(v[i]) : "v"(w[i]) : "scc");
                if (OP == 18) asm volatile("v_mul_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %1\n\tv_lshlrev_b32 %3, 16, %3\n\tv_pk_add_f32 %4, %4, %5" : "+v"(v[i]), "+v"(w[i]), "+v"(w[(i+5)&15]), "+v"(u[i]), "+v"(p[i]) : "v"(q[i]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + w[i] + __uint_as_float(u[i]) + p[i].x + p[i].y;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 256 * 256 * 4);
    const char* names[] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_lshlrev_b32", "v_and_b32 lit", "v_add_f32", "v_dot2_f32_bf16",
                           "v_perm_b32", "v_fma_f32", "dependent v_mul pair (per instr)", "v_bfe_u32", "v_add3_u32 lit", "mul + shift pair (per instr)", "v_pk_mul_f32 op_sel_hi", "v_accvgpr_read", "v_permlane16_swap", "accvgpr_read + mul (per instr)", "mul+cvt_pk+shift+pk_add (per instr)", "v_mul + s_cmp + TAKEN s_cbranch over one instruction (per group)", "v_mul + s_cmp + NOT-taken s_cbranch + v_mul (per group)"};
    const int iters = 200;
    auto run = [&](auto kern, int op, int per) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, sink, iters);   // warm
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
        m /= 256;
        printf("%-36s %6.2f cycles per instruction (one wave per SIMD, every CU busy)\n", names[op], m / (iters * 64.0 * per));
    };
    run(k<0>, 0, 1); run(k<1>, 1, 1); run(k<2>, 2, 1); run(k<3>, 3, 1); run(k<4>, 4, 1); run(k<5>, 5, 1); run(k<6>, 6, 1); run(k<7>, 7, 1);
    run(k<8>, 8, 1); run(k<9>, 9, 1); run(k<10>, 10, 2); run(k<11>, 11, 1); run(k<12>, 12, 1); run(k<13>, 13, 2); run(k<14>, 14, 1); run(k<15>, 15, 1); run(k<16>, 16, 1); run(k<17>, 17, 2); run(k<18>, 18, 4); run(k<19>, 19, 1); run(k<20>, 20, 1);
    return 0;
}
