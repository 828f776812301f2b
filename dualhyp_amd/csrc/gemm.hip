// y[M,N] = epilogue(x[M,K] · W[N,K]^T) on the gfx950 matrix cores.
//
// Both operands are K-contiguous, so each MFMA fragment is one 16-byte run of a row.  The kernel
// computes C^T tiles: the A operand is a 16-row slab of W (rows = n), the B operand a 16-row slab
// of x (cols = m); a lane of the 16x16 accumulator then owns 4 consecutive n for one m, which
// makes the epilogue (LoRA rank-16 update, SwiGLU pairing, adapter scale/bias, residual add) a
// per-lane affair with 8-byte row-contiguous loads/stores.
//
// The MFMA shape (16x16x32) and the k order (one accumulator per output, k ascending in steps of 32)
// are those of the 256-tile kernel (gemm256.hip): an output element goes through the SAME fp32 chain
// whichever of the two tile sizes dh_linear_impl picks for the call, so a row's result does not depend
// on how many other rows were packed into the launch (tests/test_hip_ops.py::test_linear_tile_size_invariant).
//
//   block  : 256 threads = 4 waves as 2(n) x 2(m); block tile 128(n) x 128(m), BK = 64
//   wave   : 64 x 64 = 4 x 4 MFMA 16x16x32 tiles, 64 accumulator VGPRs
//   LDS    : 2 stages x (W tile 16 KiB + x tile 16 KiB) = 64 KiB -> 2 blocks / CU
//   staging: global_load_lds 16 B/lane (LDS image is lane-linear, so the bank swizzle
//            chunk ^= (row>>1)&7 is applied to the SOURCE address and again on the ds_read)
//   grid   : 1-D, remapped so each XCD (blocks b, b+8, ...) owns a contiguous band of tiles
//
// LoRA fused as a rank-16 epilogue (ger/lora.py:159-166, 388-402): xa = bf16(x·A^T) is computed
// beforehand ([M,16*segments]); each 16x16 output tile then needs ONE extra MFMA (rank 16 zero-padded
// to the instruction's K = 32) lacc = lora_B[16 n,16] · xa[16 m,16]^T, and
// y = bf16(bf16(acc) + bf16(bf16(lacc)*s)).
#include "common.h"
#include <atomic>
#include "gemm.h"

int g_gemm128_stages = 0;   // 0: by grid size, else 2 or 4 (dh_set_tuning key 9)

namespace {

constexpr int BT = 128;   // block tile edge (both n and m)
constexpr int BK = 64;
constexpr int TILE_BYTES = BT * BK * 2;  // 16 KiB

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// NST: LDS stages.  2: double buffer, two blocks per CU (large grids: a block's waits are covered by its
// neighbour).  4: three K-tiles in flight with counted s_waitcnt, one block per CU — for grids smaller than
// the chip (a training micro-batch: ~100 blocks each walking K alone, where the double buffer spends about
// one memory latency per K-tile).
template <int EPI, bool RESID, int NST>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NST x (W tile 16 KiB + x tile 16 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD -> give them neighbouring tiles
    const int nwg = a.nb_n * a.nb_m;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = tile / a.nb_n, tn = tile % a.nb_n;
    const int m0 = tm * BT;
    // SWIGLU: 64 rows of fc_1 and the same 64 rows of fc_2 per block
    const int n0 = (EPI == DH_EPI_SWIGLU) ? tn * 64 : tn * BT;

    // ---- per-lane source rows for the 4+4 staging instructions of this wave
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = wave * 4 + j;               // 1-KiB row group: LDS rows R*8 .. R*8+7
        const int row = R * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);  // logical 16-B chunk held by this LDS slot
        {
            const bf16_t* base = a.w;
            int n;
            if (EPI == DH_EPI_SWIGLU) {
                const int half = (row >> 5) & 1;  // 32-row halves of a wave's 64 rows: 0 = fc_1, 1 = fc_2
                n = n0 + (row >> 6) * 32 + (row & 31);
                base = half ? a.w2 : a.w;
            } else {
                n = n0 + row;
            }
            n = n < a.N ? n : a.N - 1;
            srcA[j] = base + (size_t)n * a.K + chunk * 8;
        }
        {
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            srcB[j] = a.x + (size_t)m * a.K + chunk * 8;
        }
    }
    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * 2 * TILE_BYTES;
        char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int R = wave * 4 + j;
            glds16(srcA[j] + kt * BK, sA + R * 1024);
            glds16(srcB[j] + kt * BK, sB + R * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside a tile: lane (frow, kg) reads row frow of a 16-row slab, the 16-byte
    // chunk 4*ks + kg of its 128-byte row; (row>>1)&7 only depends on the lane here (slabs start at multiples of 16)
    const int frow = lane & 15, kg = lane >> 4;
    const int offA = (wn * 64 + frow) * 128, offB = (wm * 64 + frow) * 128;
    const int sw = swz(frow);

    const int nk = a.K / BK;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* sA = smem + buf * 2 * TILE_BYTES;
        const char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int co = (((ks * 4 + kg) ^ sw) << 4);
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8*>(sA + offA + i * 2048 + co);
                fb[i] = *reinterpret_cast<const bf16x8*>(sB + offB + i * 2048 + co);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NST == 2) {
        stage(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            compute(cur);
            __syncthreads();  // drains the stage's vmcnt and fences the buffer swap
        }
    } else {
        // 8 DMA requests per wave and stage; K-tile kt+1 must have landed at the end of iteration kt
#pragma unroll
        for (int p = 0; p < NST - 1; ++p)
            if (p < nk) stage(p, p);
        if (2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            // slot (kt+3) % 4 was last read in iteration kt-1, which ended with a barrier
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
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        const bool m_ok = m < a.M;
        if (EPI == DH_EPI_SWIGLU) {
            // the wave's 64 LDS rows are 32 rows of fc_1 (tiles 0,1) and the same 32 rows of fc_2 (tiles 2,3)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = n0 + wn * 32 + i * 16 + 4 * kg;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = rbf(acc[i][j][e]);
                    const float u = rbf(acc[i + 2][j][e]);
                    const float sg = rbf(silu_fast(g));
                    o[e] = sg * u;
                }
                if (m_ok && n < a.N) {
                    uint2 pk = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                    *reinterpret_cast<uint2*>(a.y + (size_t)m * a.N + n) = pk;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nt = n0 + wn * 64 + i * 16;
                const int n = nt + 4 * kg;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rbf(acc[i][j][e]);
                if (EPI == DH_EPI_LORA) {
                    const int seg = (nt >= a.split0) + (nt >= a.split1);
                    int nn = nt + frow;
                    nn = nn < a.N ? nn : a.N - 1;
                    const int mm = m_ok ? m : a.M - 1;
                    bf16x8 lb = zero8, xf = zero8;          // rank 16 zero-padded to the MFMA's K = 32
                    if (kg < 2) {
                        lb = *reinterpret_cast<const bf16x8*>(a.lora_b + (size_t)nn * 16 + kg * 8);
                        xf = *reinterpret_cast<const bf16x8*>(a.xa + (size_t)mm * a.xa_ld + seg * 16 + kg * 8);
                    }
                    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
                    lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lb, xf, lacc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rbf(o[e] + rbf(rbf(lacc[e]) * a.lora_scale));
                }
                if (!(m_ok && n < a.N)) continue;
                if (EPI == DH_EPI_ADAPTER) {
                    const uint2 sc = *reinterpret_cast<const uint2*>(a.vec_a + n);
                    const uint2 bi = *reinterpret_cast<const uint2*>(a.vec_b + n);
                    const bf16_t* sp = reinterpret_cast<const bf16_t*>(&sc);
                    const bf16_t* bp = reinterpret_cast<const bf16_t*>(&bi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rbf(bf2f(sp[e]) * rbf(o[e] + bf2f(bp[e])));
                }
                if (RESID) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.N + n);
                    const bf16_t* rp = reinterpret_cast<const bf16_t*>(&rr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = bf2f(rp[e]) + o[e];
                }
                uint2 pk = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                *reinterpret_cast<uint2*>(a.y + (size_t)m * a.N + n) = pk;
            }
        }
    }
}

template <int EPI, bool RESID, int NST>
int launch_n(const GemmArgs& a, hipStream_t s) {
    constexpr int lds = NST * 2 * TILE_BYTES;
    DH_MAX_LDS_ONCE((gemm_nt_kernel<EPI, RESID, NST>), lds);
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, RESID, NST>), dim3(a.nb_n * a.nb_m), dim3(256), lds, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch(const GemmArgs& a, hipStream_t s) {
    const int blocks = a.nb_n * a.nb_m;
    const int nst = g_gemm128_stages ? g_gemm128_stages : (blocks < 384 ? 4 : 2);
    if (nst == 4) return a.resid ? launch_n<EPI, true, 4>(a, s) : launch_n<EPI, false, 4>(a, s);
    return a.resid ? launch_n<EPI, true, 2>(a, s) : launch_n<EPI, false, 2>(a, s);
}

}  // namespace

int g_gemm_variant = 5;
int g_tail_split = 1;     // dh_set_tuning(28, 0): no row split of grids with a small last round (A/B)
int g_linear_phase = 0;   // kernel choice of the public dh_linear_bf16 (dh_set_tuning key 4; tools and tests)

extern "C" int dh_linear_bf16(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, int epilogue,
                              const dh_bf16* w2, const dh_bf16* xa, int xa_ld, const dh_bf16* lora_b,
                              float lora_scale, int split0, int split1, const dh_bf16* vec_a,
                              const dh_bf16* vec_b, const dh_bf16* resid, void* stream) {
    return dh_linear_impl(x, w, y, M, N, K, epilogue, w2, xa, xa_ld, lora_b, lora_scale, split0, split1, vec_a, vec_b,
                          resid, g_linear_phase, (hipStream_t)stream);
}

bool dh_linear_is_big(int M, int N, int epilogue) {
    const int tiles256 = cdiv(M, 256) * cdiv(N, epilogue == DH_EPI_SWIGLU ? 128 : 256);
    return g_gemm_variant >= 1 && M >= 256 && N >= (epilogue == DH_EPI_SWIGLU ? 128 : 256) && tiles256 >= 128;
}

static int qkv_rope_cache_impl(const dh_bf16* x, const dh_bf16* w, int M, int K, const dh_bf16* xa, int xa_ld, const dh_bf16* lora_a,
                               const dh_bf16* lora_b, float lora_scale, const dh_bf16* cos, const dh_bf16* sin,
                               const int32_t* tok_slot, const int32_t* tok_pos, dh_bf16* q_out, dh_bf16* k_cache,
                               dh_bf16* vT_cache, int n_head, int n_groups, int hs, int s_max, void* stream) {
    DH_CHECK(x && w && cos && sin && tok_slot && tok_pos && q_out && k_cache && vT_cache, "dh_linear_qkv_rope_cache_bf16: null argument");
    DH_CHECK(hs == 64 || hs == 128, "dh_linear_qkv_rope_cache_bf16: head_size %d unsupported", hs);
    DH_CHECK(n_groups > 0 && n_head % n_groups == 0 && K % BK == 0, "dh_linear_qkv_rope_cache_bf16: bad shape");
    const int N = (n_head + 2 * n_groups) * hs, d = n_head * hs, kv = n_groups * hs;
    DH_CHECK(dh_linear_is_big(M, N, DH_EPI_LORA), "dh_linear_qkv_rope_cache_bf16: M=%d is below the 256-tile kernel's range; use dh_linear_bf16 + dh_qkv_rope_cache_bf16", M);
    DH_CHECK(lora_b == nullptr || (xa && xa_ld >= 48 && xa_ld % 8 == 0), "dh_linear_qkv_rope_cache_bf16: LoRA needs xa [M, >=48]");
    GemmArgs a{};
    a.x = x; a.w = w; a.xa = xa; a.lora_b = lora_b; a.M = M; a.N = N; a.K = K; a.xa_ld = xa_ld; a.split0 = d; a.split1 = d + kv;
    a.lora_scale = lora_scale;
    a.rope_cos = cos; a.rope_sin = sin; a.tok_slot = tok_slot; a.tok_pos = tok_pos; a.q_out = q_out; a.k_cache = k_cache;
    a.vT_cache = vT_cache; a.n_head = n_head; a.n_groups = n_groups; a.hs = hs; a.s_max = s_max;
    if (lora_a != nullptr && lora_b != nullptr) {
        // the down-projection inside the GEMM's K loop where the kernel and the segment boundaries allow it, else one launch into xa
        a.lora_a = lora_a;
        if (!dh_linear_256_xa_ok(a, DH_EPI_QKV)) {
            a.lora_a = nullptr;
            const int rc = dh_linear_impl(x, lora_a, const_cast<dh_bf16*>(xa), M, 48, K, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 1.f, 0, 0,
                                          nullptr, nullptr, nullptr, 1, (hipStream_t)stream);
            if (rc) return rc;
        }
    }
    return dh_linear_256(a, DH_EPI_QKV, (hipStream_t)stream);
}

extern "C" int dh_linear_qkv_rope_cache_bf16(const dh_bf16* x, const dh_bf16* w, int M, int K, const dh_bf16* xa, int xa_ld,
                                             const dh_bf16* lora_b, float lora_scale, const dh_bf16* cos, const dh_bf16* sin,
                                             const int32_t* tok_slot, const int32_t* tok_pos, dh_bf16* q_out, dh_bf16* k_cache,
                                             dh_bf16* vT_cache, int n_head, int n_groups, int hs, int s_max, void* stream) {
    return qkv_rope_cache_impl(x, w, M, K, xa, xa_ld, nullptr, lora_b, lora_scale, cos, sin, tok_slot, tok_pos, q_out, k_cache, vT_cache,
                               n_head, n_groups, hs, s_max, stream);
}

extern "C" int dh_linear_qkv_lora_rope_cache_bf16(const dh_bf16* x, const dh_bf16* w, int M, int K, const dh_bf16* lora_a,
                                                  const dh_bf16* lora_b, float lora_scale, const dh_bf16* cos, const dh_bf16* sin,
                                                  const int32_t* tok_slot, const int32_t* tok_pos, dh_bf16* q_out, dh_bf16* k_cache,
                                                  dh_bf16* vT_cache, int n_head, int n_groups, int hs, int s_max, dh_bf16* xa_work,
                                                  void* stream) {
    DH_CHECK(lora_a && lora_b && xa_work, "dh_linear_qkv_lora_rope_cache_bf16: lora_a, lora_b and xa_work ([M, 48] bf16) are required");
    return qkv_rope_cache_impl(x, w, M, K, xa_work, 48, lora_a, lora_b, lora_scale, cos, sin, tok_slot, tok_pos, q_out, k_cache, vT_cache,
                               n_head, n_groups, hs, s_max, stream);
}

// dh_linear_bf16's LORA epilogue with the down-projection computed by the library (include/dualhyp_hip.h)
int dh_linear_lora_impl(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, const dh_bf16* lora_a, const dh_bf16* lora_b,
                        float lora_scale, int split0, int split1, const dh_bf16* resid, dh_bf16* xa_work, int kernel, hipStream_t s) {
    DH_CHECK(x && w && y && lora_a && lora_b && xa_work, "dh_linear_lora_bf16: null argument");
    DH_CHECK(M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0, "dh_linear_lora_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    DH_CHECK(split0 % 32 == 0 && split1 % 32 == 0 && split0 <= split1, "dh_linear_lora_bf16: LoRA splits must be multiples of 32");
    if (M == 0) return 0;
    const int nseg = 1 + (split0 < N) + (split1 < N);
    if (kernel != 2 && M > 32 && dh_linear_is_big(M, N, DH_EPI_LORA)) {
        GemmArgs a{};
        a.x = x; a.w = w; a.y = y; a.lora_a = lora_a; a.lora_b = lora_b; a.resid = resid; a.M = M; a.N = N; a.K = K;
        a.xa_ld = 16 * nseg; a.split0 = split0; a.split1 = split1; a.lora_scale = lora_scale;
        if (dh_linear_256_xa_ok(a, DH_EPI_LORA)) return dh_linear_256(a, DH_EPI_LORA, s);
    }
    const int rc = dh_linear_impl(x, lora_a, xa_work, M, 16 * nseg, K, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 1.f, 0, 0, nullptr, nullptr,
                                  nullptr, kernel, s);
    if (rc) return rc;
    return dh_linear_impl(x, w, y, M, N, K, DH_EPI_LORA, nullptr, xa_work, 16 * nseg, lora_b, lora_scale, split0, split1, nullptr, nullptr,
                          resid, kernel, s);
}

extern "C" int dh_linear_lora_bf16(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, const dh_bf16* lora_a,
                                   const dh_bf16* lora_b, float lora_scale, int split0, int split1, const dh_bf16* resid,
                                   dh_bf16* xa_work, void* stream) {
    return dh_linear_lora_impl(x, w, y, M, N, K, lora_a, lora_b, lora_scale, split0, split1, resid, xa_work, g_linear_phase, (hipStream_t)stream);
}

// Training forward of the MLP's first half in ONE launch (round 4; finetune/ger.py:278-292 keeps fc_1(n2) and fc_2(n2) for the backward):
// the SwiGLU GEMM whose epilogue also stores the rounded pre-activations.  g = bf16(x.W1^T), u = bf16(x.W2^T), act = bf16(bf16(silu(g)) * u)
// — the bits of two plain GEMMs followed by dh_swiglu_fwd_bf16 (the same accumulator chain per output), which is also the fallback
// below the 256-tile kernel's range.
extern "C" int dh_swiglu_fwd_bf16(const dh_bf16* g, const dh_bf16* u, dh_bf16* act, int64_t n, void* stream);
extern "C" int dh_linear_swiglu_train_bf16(const dh_bf16* x, const dh_bf16* w1, const dh_bf16* w2, dh_bf16* act, dh_bf16* g, dh_bf16* u,
                                           int M, int I, int K, void* stream) {
    DH_CHECK(x && w1 && w2 && act && g && u, "dh_linear_swiglu_train_bf16: null argument");
    DH_CHECK(M >= 0 && I > 0 && K > 0 && K % BK == 0 && I % 32 == 0, "dh_linear_swiglu_train_bf16: bad shape M=%d I=%d K=%d", M, I, K);
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (M > 32 && dh_linear_is_big(M, I, DH_EPI_SWIGLU)) {
        GemmArgs a{};
        a.x = x; a.w = w1; a.w2 = w2; a.y = act; a.M = M; a.N = I; a.K = K; a.lora_scale = 1.f;
        a.resid = g;                     // selects the kernels' RESID instantiation (= "store g and u" for this epilogue; never read)
        a.q_out = g; a.k_cache = u;
        return dh_linear_256(a, DH_EPI_SWIGLU, s);
    }
    int rc = dh_linear_impl(x, w1, g, M, I, K, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 1.f, 0, 0, nullptr, nullptr, nullptr, g_linear_phase, s);
    if (rc) return rc;
    rc = dh_linear_impl(x, w2, u, M, I, K, DH_EPI_PLAIN, nullptr, nullptr, 0, nullptr, 1.f, 0, 0, nullptr, nullptr, nullptr, g_linear_phase, s);
    if (rc) return rc;
    return dh_swiglu_fwd_bf16(g, u, act, (int64_t)M * I, stream);
}

// y = bf16(bf16(x . W^T) * mul): a plain GEMM whose epilogue multiplies by an [M, N] bf16 operand where it rounds (round 4, ABI 6) — the
// dropout mask of a LoRA branch in the fine-tune's backward (d drop(x) / dx = m), otherwise one more elementwise pass over
// [tokens, d].  The bits of dh_linear_bf16 followed by a bf16 multiply.  256-tile range only (dh_linear_is_big): the caller multiplies
// small cases itself.
extern "C" int dh_linear_mul_bf16(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, const dh_bf16* mul, void* stream) {
    DH_CHECK(x && w && y && mul, "dh_linear_mul_bf16: null argument");
    DH_CHECK(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0, "dh_linear_mul_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    DH_CHECK(M > 32 && dh_linear_is_big(M, N, DH_EPI_PLAIN), "dh_linear_mul_bf16: M=%d N=%d is below the 256-tile kernels' range", M, N);
    if (dh_linear_k64_ok(M, N, K, x, w, y, mul)) return dh_linear_k64(x, w, mul, y, M, N, (hipStream_t)stream);     // rank-padded LoRA product
    GemmArgs a{};
    a.x = x; a.w = w; a.y = y; a.resid = mul; a.resid_mul = 1; a.M = M; a.N = N; a.K = K; a.lora_scale = 1.f;
    return dh_linear_256(a, DH_EPI_PLAIN, (hipStream_t)stream);
}

int dh_linear_impl(const dh_bf16* x, const dh_bf16* w, dh_bf16* y, int M, int N, int K, int epilogue,
                   const dh_bf16* w2, const dh_bf16* xa, int xa_ld, const dh_bf16* lora_b, float lora_scale,
                   int split0, int split1, const dh_bf16* vec_a, const dh_bf16* vec_b, const dh_bf16* resid,
                   int kernel, hipStream_t s) {
    DH_CHECK(M >= 0 && N > 0 && K > 0, "dh_linear_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    DH_CHECK(K % BK == 0, "dh_linear_bf16: K=%d must be a multiple of %d", K, BK);
    DH_CHECK(N % 8 == 0, "dh_linear_bf16: N=%d must be a multiple of 8", N);
    DH_CHECK(x && w && y, "dh_linear_bf16: null operand");
    if (M == 0) return 0;
    GemmArgs a{};
    a.x = x; a.w = w; a.w2 = w2; a.y = y; a.xa = xa; a.lora_b = lora_b; a.vec_a = vec_a; a.vec_b = vec_b;
    a.resid = resid; a.M = M; a.N = N; a.K = K; a.xa_ld = xa_ld; a.split0 = split0; a.split1 = split1;
    a.lora_scale = lora_scale;
    a.nb_m = cdiv(M, BT);
    a.nb_n = (epilogue == DH_EPI_SWIGLU) ? cdiv(N, 64) : cdiv(N, BT);
    if ((kernel == 2 || (kernel == 0 && M <= 32)) && (epilogue != DH_EPI_SWIGLU || (w2 != nullptr && resid == nullptr)) &&
        (epilogue != DH_EPI_ADAPTER || (vec_a && vec_b)) && dh_linear_mid_ok(a, epilogue))
        return dh_linear_mid(a, epilogue, s);
    if (kernel == 0 && epilogue == DH_EPI_PLAIN && resid == nullptr && dh_linear_k64_ok(M, N, K, x, w, y, nullptr))
        return dh_linear_k64(x, w, nullptr, y, M, N, s);
    if (kernel == 0 && epilogue == DH_EPI_PLAIN && resid == nullptr && dh_linear_skinny_n_ok(M, N, K, x, w, y))
        return dh_linear_skinny_n(x, w, y, M, N, K, s);          // a LoRA down-projection of a long batch: 16-64 output columns
    const bool skinny = (kernel == 0 || kernel == 2) && M <= 32 && K % 32 == 0;   // weight-streaming kernel (gemm_skinny.hip)
    // the 256-tile kernel runs one block per CU: below ~half a chip of tiles (a training micro-batch, M ~ 560)
    // the 128-tile kernel puts four times the blocks in flight and wins
    const bool big = !skinny && dh_linear_is_big(M, N, epilogue);
    // Wave quantisation (round 4): the 256-tile kernels run one tile per CU at a time, so a grid of R full rounds plus a FEW tiles
    // costs R + 1 rounds (the packed fine-tune: 17 920 rows x 2048 columns = 560 tiles = 2.19 rounds -> 3).  When the remainder is
    // under 0.3 of a round, the whole row tiles of the full rounds go to the 256-tile kernel and the remaining rows (a few hundred
    // to a few thousand) to the 128-tile kernel, which puts four times the blocks in flight.  Same accumulator chain per output in
    // both kernels (one accumulator, k ascending): the same bits as the single launch.
    static thread_local bool in_split = false;
    if (big && g_tail_split && !in_split && (epilogue == DH_EPI_PLAIN || epilogue == DH_EPI_LORA)) {
        static std::atomic<int> n_cu{0};                 // (one device model per process: replicas of the same GPU)
        int cu = n_cu.load(std::memory_order_relaxed);
        if (cu == 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
            n_cu.store(cu, std::memory_order_relaxed);
        }
        const int nbn = cdiv(N, 256), tiles = cdiv(M, 256) * nbn, rounds = tiles / cu, rem = tiles % cu;
        const int m1_tiles = rounds * cu / nbn, M1 = m1_tiles * 256, M2 = M - M1;
        if (rounds >= 1 && rem > 0 && rem * 10 <= cu * 3 && M2 > 0 && M2 <= 4096 && dh_linear_is_big(M1, N, epilogue) &&
            !dh_linear_is_big(M2, N, epilogue)) {
            in_split = true;
            int rc = dh_linear_impl(x, w, y, M1, N, K, epilogue, w2, xa, xa_ld, lora_b, lora_scale, split0, split1, vec_a, vec_b, resid, kernel, s);
            if (rc == 0)
                rc = dh_linear_impl(x + (size_t)M1 * K, w, y + (size_t)M1 * N, M2, N, K, epilogue, w2, xa ? xa + (size_t)M1 * xa_ld : nullptr, xa_ld,
                                    lora_b, lora_scale, split0, split1, vec_a, vec_b, resid ? resid + (size_t)M1 * N : nullptr, kernel, s);
            in_split = false;
            return rc;
        }
    }
    if (big) {
        if (epilogue == DH_EPI_LORA) {
            DH_CHECK(xa && lora_b && xa_ld >= 16 && xa_ld % 8 == 0, "dh_linear_bf16: LORA epilogue needs xa/lora_b");
            DH_CHECK(split0 % 32 == 0 && split1 % 32 == 0 && split0 <= split1, "dh_linear_bf16: LoRA splits must be multiples of 32");
            DH_CHECK(xa_ld >= 16 * (1 + (split0 < N) + (split1 < N)), "dh_linear_bf16: xa_ld too small for the segments");
        }
        if (epilogue == DH_EPI_SWIGLU) DH_CHECK(w2 != nullptr && resid == nullptr && N % 32 == 0, "dh_linear_bf16: bad SWIGLU arguments");
        if (epilogue == DH_EPI_ADAPTER) DH_CHECK(vec_a && vec_b, "dh_linear_bf16: ADAPTER epilogue needs scale/bias vectors");
        DH_CHECK(epilogue >= 0 && epilogue <= 3, "dh_linear_bf16: unknown epilogue %d", epilogue);
        return dh_linear_256(a, epilogue, s);
    }
    switch (epilogue) {
        case DH_EPI_PLAIN:
            return skinny ? dh_linear_skinny(a, epilogue, s) : launch<DH_EPI_PLAIN>(a, s);
        case DH_EPI_LORA:
            DH_CHECK(xa && lora_b && xa_ld >= 16 && xa_ld % 8 == 0, "dh_linear_bf16: LORA epilogue needs xa/lora_b");
            DH_CHECK(split0 % 32 == 0 && split1 % 32 == 0 && split0 <= split1, "dh_linear_bf16: LoRA splits must be multiples of 32");
            DH_CHECK(xa_ld >= 16 * (1 + (split0 < N) + (split1 < N)), "dh_linear_bf16: xa_ld too small for the segments");
            return skinny ? dh_linear_skinny(a, epilogue, s) : launch<DH_EPI_LORA>(a, s);
        case DH_EPI_SWIGLU:
            DH_CHECK(w2 != nullptr, "dh_linear_bf16: SWIGLU epilogue needs w2");
            DH_CHECK(resid == nullptr, "dh_linear_bf16: SWIGLU epilogue takes no residual");
            DH_CHECK(N % 32 == 0, "dh_linear_bf16: SWIGLU needs N %% 32 == 0");
            return skinny ? dh_linear_skinny(a, epilogue, s) : launch<DH_EPI_SWIGLU>(a, s);
        case DH_EPI_ADAPTER:
            DH_CHECK(vec_a && vec_b, "dh_linear_bf16: ADAPTER epilogue needs scale/bias vectors");
            return skinny ? dh_linear_skinny(a, epilogue, s) : launch<DH_EPI_ADAPTER>(a, s);
        default:
            DH_CHECK(false, "dh_linear_bf16: unknown epilogue %d", epilogue);
    }
    return 0;
}
