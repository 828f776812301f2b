// Shared device/host helpers for libdualhyp_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>
#include <utility>

#include "../../include/dualhyp_hip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;    // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 MFMA accumulator

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserving: plain casts, MI355X_MICROARCH
// "Correctness boundaries") ---------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(bf16_t, b);
}
// round an fp32 value to the nearest bf16 and return it as fp32: one rounding point of the
// reference's eager bf16 tensor program
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// two values -> one dword of two bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32.  Written as two scalar conversions + shift + or
// the compiler only sometimes finds the packed form (the prefill attention loop had 32 single conversions, 16 shifts and 16
// ors per 32 probabilities); same rounding either way (RNE, the instruction's NaN handling).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// ---- wave64 reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// silu(g) = g / (1 + e^-g) for every SwiGLU epilogue (prefill tiles of both sizes, the decode family, fp8, the training
// forward: one formula, so no row depends on which kernel finished it): hardware
// exp2 and reciprocal (about 1 ulp each; the argument product adds |g| * 1e-7 relative).  libm's expf and the IEEE
// division cost ~45 VALU instructions per element, 64 elements per lane: 5.8 us of a 60-us SwiGLU tile
// (tools/probe_gemm256.py).  The value is rounded to bf16 right after: a relative error of a few 1e-7 moves about one
// result in 10^4 by one bf16 ulp, as libm-vs-Sleef differences between the reference's CPU and GPU runs do.
__device__ __forceinline__ float silu_fast(float g) {
#ifdef DH_SILU_EXACT   // A/B builds only
    return g / (1.0f + expf(-g));
#else
    return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-g * 1.44269504088896341f));
#endif
}

// ---- async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16 ----
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The saddr form of the same request, written out: uniform base in an SGPR pair + one 32-bit byte offset per lane, LDS destination
// (uniform) through M0.  From the builtin the compiler hoists zero-extended lane offsets out of a loop as 64-bit VGPR pairs and
// issues a v_lshl_add_u64 plus the 64-bit-address form per request; the compiler does not count an asm load (wait with
// explicit s_waitcnt vmcnt) and never waits before a ds_read that may alias its destination.
__device__ __forceinline__ void glds16_saddr(const void* base_uniform, uint32_t lane_off, void* lds_wave_base) {
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base_uniform), "s"(lds) : "memory", "m0");
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) (#pragma unroll gives up on big bodies, and a
// loop that stays a loop indexes register arrays dynamically, i.e. puts them in scratch)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// the same with the non-temporal policy (aux = 2): bytes this CU alone reads once (streamed weights)
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

// ---- full-line loads -> MFMA fragments, in registers ------------------------------------------------------
// A 16-row x 128-B operand block fetched as two requests of full lines (request h: rows 8h..8h+7, lane -> row lane/8,
// 16-B piece lane%8 of the row's line) holds, for the 16x16 MFMA forms whose lane (row r = lane%16, kg = lane/16) wants
// 16-B piece 4t + kg of row r (bf16 x32: t = k-step 0/1 of the line), the 64 pieces of fragment t on the lanes with
// (lane & 4) == 4t of both requests.  One DPP move merges them (DPP bank masks cover lanes in fours), one ds_bpermute
// per dword puts them in fragment order: source lane 8 (r%8) + kg + 4 (r/8) for both fragments.
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ int frag_src_lane(int lane) { return ((lane & 7) << 3) + (lane >> 4) + (((lane >> 3) & 1) << 2); }
__device__ __forceinline__ void lines_to_frags(const i32x4& r0, const i32x4& r1, int fidx, bf16x8& f0, bf16x8& f1) {
    union { i32x4 i; bf16x8 b; } a, b;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int lo = __builtin_amdgcn_update_dpp(r0[d], r1[d], 0x114 /* row_shr:4 */, 0xF, 0xA /* lanes 4-7, 12-15 of a row */, false);
        const int hi = __builtin_amdgcn_update_dpp(r1[d], r0[d], 0x104 /* row_shl:4 */, 0xF, 0x5 /* lanes 0-3, 8-11 */, false);
        a.i[d] = __builtin_amdgcn_ds_bpermute(fidx, lo);
        b.i[d] = __builtin_amdgcn_ds_bpermute(fidx, hi);
    }
    f0 = a.b;
    f1 = b.b;
}

// The inverse, for 16x16 accumulator tiles on their way to memory: two C^T tiles of 16 columns (t = 0 / 1) x 16 rows, lane
// (row r = lane%16, kg = lane/16) holding columns 16t + 4kg..+4 of row r, become two requests of full 128-B lines (r0: rows
// 0..7, r1: rows 8..15; lane -> row lane/8, columns 4 (lane%8)..+4 of the 32).  Stores of 16 rows x 64 B per instruction
// pay per 64-B segment like fragment-shaped loads do.
__device__ __forceinline__ int line_src_lane(int lane) { return ((lane & 3) << 4) | (((lane >> 2) & 1) << 3) | (lane >> 3); }
__device__ __forceinline__ void tiles_to_lines(const f32x4& t0, const f32x4& t1, int lidx, f32x4& r0, f32x4& r1) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int m0 = __builtin_amdgcn_ds_bpermute(lidx, __float_as_int(t0[d]));
        const int m1 = __builtin_amdgcn_ds_bpermute(lidx, __float_as_int(t1[d]));
        r0[d] = __int_as_float(__builtin_amdgcn_update_dpp(m0, m1, 0x114 /* row_shr:4 */, 0xF, 0xA, false));
        r1[d] = __int_as_float(__builtin_amdgcn_update_dpp(m1, m0, 0x104 /* row_shl:4 */, 0xF, 0x5, false));
    }
}

// ---- KV cache in MFMA-fragment order --------------------------------------------------------
// Both caches are stored per (slot, group) as 32-key tiles of HS*32 elements, laid out so that a
// wave-wide contiguous 1-KiB load (lane i <- 16 B at i*16) IS an MFMA operand fragment:
//   K  tile: [ks = d/16][lh = (d/8)&1][lr = key&31][8 d]      A operand of S^T = K.Q^T  (k-step ks)
//   V^T tile: [dt = d/32][s2][lh][lr = d&31][8 keys]          A operand of O^T = V^T.P^T (k-step s2),
//             the 8 keys being 16*s2 + 8*(j>>2) + 4*lh + (j&3), j = 0..7 — the k-order in which the
//             S^T accumulator registers come out (cdna_hip_programming.md §3).
// Offsets below are in elements from the start of the (slot, group) block of s_max*HS elements.
template <int HS>
__device__ __forceinline__ size_t kfrag_off(int key, int d) {
    const int t = key >> 5, lr = key & 31, ks = d >> 4, lh = (d >> 3) & 1, e = d & 7;
    return ((size_t)((t * (HS / 16) + ks) * 2 + lh) * 32 + lr) * 8 + e;
}
template <int HS>
__device__ __forceinline__ size_t vfrag_off(int key, int d) {
    const int t = key >> 5, kk = key & 31, s2 = kk >> 4, r = kk & 15;
    const int j = ((r >> 3) << 2) | (r & 3), lh = (r >> 2) & 1, dt = d >> 5, lr = d & 31;
    return ((size_t)(((t * (HS / 32) + dt) * 2 + s2) * 2 + lh) * 32 + lr) * 8 + j;
}
// start (in elements) of the 1-KiB fragment block of tile t: K k-step ks / V^T (dt, s2); add lane*8
template <int HS> __device__ __forceinline__ size_t kfrag_blk(int t, int ks) { return (size_t)(t * (HS / 16) + ks) * 512; }
template <int HS> __device__ __forceinline__ size_t vfrag_blk(int t, int dt, int s2) { return (size_t)((t * (HS / 32) + dt) * 2 + s2) * 512; }

// ---- host-side error plumbing -------------------------------------------------------------
void dh_set_error(const char* fmt, ...);

#define DH_CHECK(cond, ...)                      \
    do {                                         \
        if (!(cond)) {                           \
            dh_set_error(__VA_ARGS__);           \
            return 1;                            \
        }                                        \
    } while (0)

#define DH_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            dh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

#define DH_LAUNCH_CHECK()                                                              \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess) {                                                        \
            dh_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return 3;                                                                  \
        }                                                                              \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute applies to the CURRENT device, and the launchers are entered from several host threads
// (one engine per thread, dualhyp_amd/pipeline.py): remember per device that the attribute is set.  Two threads
// racing on the first launch both set it, which is harmless.  `kernel` must be parenthesised if it contains commas.
#define DH_MAX_LDS_ONCE(kernel, bytes)                                                                       \
    do {                                                                                                     \
        static std::atomic<uint64_t> _done{0};                                                               \
        int _dev = 0;                                                                                        \
        DH_HIP(hipGetDevice(&_dev));                                                                         \
        const uint64_t _bit = 1ull << (_dev & 63);                                                           \
        if (!(_done.load(std::memory_order_acquire) & _bit)) {                                               \
            DH_HIP(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            _done.fetch_or(_bit, std::memory_order_release);                                                 \
        }                                                                                                    \
    } while (0)
