// y[M, N] = epilogue(x · W^T) for the decode step, M <= 192 rows in practice (one 32-sequence batch, or several
// batches decoded jointly: generate.py generate_batch).  HBM-bound weight streaming; x is staged
// ONCE per block in LDS and shared by the 8 waves instead of being fetched per wave from L2 as
// gemm_skinny.hip does.
//   block : 640 threads = 8 compute waves (RS row sets x KQ K-parts) + 2 loader waves; a compute wave owns 16 W rows (SWIGLU: of fc_1
//           AND fc_2) and one contiguous K-part; a slice holds KSL/KQ k-steps of EVERY part, so a block covers only 16*RS output
//           columns and even N = 5632 puts 176 blocks on the chip (SWIGLU: RS 2 x KQ 4, else 4 x 2)
//   W     : up to 64 rows (WL): global -> a PRIVATE per-wave LDS ring by LDS-DMA in full 128-B lines (12-16 KiB per wave,
//           ordered by the wave's own counted vmcnt, no barrier), fragments read back with ds_read_b128;
//           above: global -> VGPR in fragment shape, ring of 8 slices per wave (14 KiB in flight per wave)
//   x     : rows [m0, m0 + 32*NG) x 128 k per slice, global -> LDS directly (global_load_lds, 16 B per
//           lane) by the two loader waves, 4 buffers, requested 3 slices ahead; 256-B rows, 16-B chunk
//           index XOR (row & 15) applied on the SOURCE address so the fragment reads are conflict free
//   sync  : one s_barrier per slice; only the loader waves wait on a counted s_waitcnt (vmcnt is an
//           in-order counter: x requests issued by a compute wave would drain its W queue every slice)
//   MFMA  : v_mfma_f32_16x16x32_bf16, C^T[16 n][16 m] += W[16 n][32 k] · x[16 m][32 k]^T
//   reduce: the KQ partial accumulators meet in LDS (the x buffers are free by then); output tile pair t is
//           summed in K-part order and finished by wave t % KQ of its row set
//   grid  : (N / (16*RS), ceil(M / (32*NG)))
// The summation order of an output is fixed by K alone — KQ chains over CONTIGUOUS K-parts (k ascending in
// each), added in part order; gemm_dt.hip reproduces exactly this order for > 64 rows — whatever M and NG: a row's result does not depend on how many other rows ride along (batch
// invariance, same contract as the other decode kernels).
#include "common.h"
#include "gemm.h"

#ifdef DH_MID_STAMPS   // diagnostic build only (tools/probe_mid.py): 100 MHz timestamps of wave 0 / wave 8 per block
__device__ unsigned long long g_mid_stamps[1024 * 8];
#define MID_STAMP(i) do { if (lane == 0 && (wave == 0 || wave == 8) && blockIdx.y == 0 && blockIdx.x < 1024) \
    g_mid_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int dh_debug_mid_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mid_stamps), sizeof(g_mid_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define MID_STAMP(i)
#endif

int g_mid_wlds = 1;   // 1: W through the per-wave LDS ring (<= 64 rows), 0: W straight to VGPRs (dh_set_tuning 13)

namespace {

constexpr int KSL = 4;             // k-steps (of 32) per slice
constexpr int XROW = KSL * 64;     // LDS row: 256 B = 16 chunks of 16 B
constexpr int NBUF = 4;            // x slices resident in LDS
constexpr int XD = 3;              // x slices requested ahead

template <int EPI>
struct MidShape {
    static constexpr bool SW = EPI == DH_EPI_SWIGLU;
    static constexpr int NM = SW ? 2 : 1;      // matrices streamed per wave
    static constexpr int KQ = SW ? 4 : 2;      // K-parts per block
    static constexpr int RS = 8 / KQ;          // 16-row sets per block
    static constexpr int KPW = KSL / KQ;       // k-steps per wave per slice
};
// W slices held per wave (RING-1 in flight + the one being multiplied); 170 VGPRs per wave at 10 waves
template <int EPI, int NG> constexpr int mid_ring() { return 8; }

// WL: the W stream goes global -> LDS (LDS-DMA, full 128-B lines: 8 rows x 2 k-steps per request) into a ring private
// to the wave, and reaches the MFMA through ds_read_b128; otherwise global -> VGPR in fragment shape (16 rows x 64 B
// per request), which costs the texture-address unit four times the line look-ups per byte and bounds the kernel.
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int NG> constexpr int wl_ring_bytes() { return NG == 1 ? 16384 : 12288; }   // per wave

template <int EPI, int NG, bool WL>
__global__ __launch_bounds__(640) void gemm_mid_kernel(GemmArgs a) {
    using S = MidShape<EPI>;
    constexpr bool SW = S::SW;
    constexpr int NM = S::NM, KQ = S::KQ, RS = S::RS, KPW = S::KPW;
    constexpr int RING = mid_ring<EPI, NG>(), PD = RING - 1;
    constexpr int CI = NM * 2;                            // WL: LDS-DMA requests per chunk (= 2 k-steps): [matrix][row half], 1 KiB each
    constexpr int CB = CI * 1024;                         // bytes per chunk
    constexpr int RC = wl_ring_bytes<NG>() / CB;          // chunks in a wave's ring
    constexpr int NI = NG * NM * 2;                       // accumulator tiles per wave
    constexpr int NXL = NG * 4;                           // x requests per loader wave per slice (4 rows each)
    constexpr int XBUF = NG * 32 * XROW;                  // bytes per x buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][NG*32][XROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * (NG * 32);
    const int nsl = a.K / (KSL * 32);                     // K % 128 == 0 (checked on the host)
    const int kpart = a.K / 32 / KQ;                      // k-steps per K-part: part q owns k-steps [q*kpart, (q+1)*kpart)
    MID_STAMP(wave == 0 ? 0 : 4);

    if (wave >= 8) {
        // ---- loader waves: x slices global -> LDS, XD slices ahead of the MFMAs.  vmcnt is an in-order
        // counter per wave: issued by the compute waves, these requests would force their deep W queue
        // to drain at every slice.  Request j covers rows 4j..4j+3; lane -> (row 4j + lane/16, chunk lane%16).
        const int ldr = wave - 8;
        const bf16_t* xsrc[NXL];
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int row = 4 * (ldr + 2 * i) + (lane >> 4);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            // logical chunk lc of a slice row = k-step (lc / 4) of the slice: K-part (lc/4) / KPW, its step (lc/4) % KPW
            const int lc = (lane & 15) ^ (row & 15);
            xsrc[i] = a.x + (size_t)m * a.K + (((lc >> 2) / KPW) * kpart + (lc >> 2) % KPW) * 32 + (lc & 3) * 8;
        }
        auto load_x = [&](int s) __attribute__((always_inline)) {
            const int ko = s < nsl ? s * (KPW * 32) : 0;       // past K: L2 hits that keep the count uniform
            char* xb = smem + (s % NBUF) * XBUF;
#pragma unroll
            for (int i = 0; i < NXL; ++i) glds16(xsrc[i] + ko, xb + (ldr + 2 * i) * (4 * XROW));
        };
#pragma unroll
        for (int p = 0; p < XD; ++p) load_x(p);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD - 1) * NXL) : "memory");   // slice 0 has landed
        MID_STAMP(5);
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsl; ++s) {
            load_x(s + XD);            // buffer (s+XD) % NBUF was last read in step s-1, which ended with a barrier
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD - 1) * NXL) : "memory");   // slice s+1 has landed
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail requests still target LDS
        __builtin_amdgcn_s_barrier();                       // x buffers free for the reduction
        __builtin_amdgcn_s_barrier();                       // partial tiles written (compute waves)
        return;
    }

    const int lrow = lane & 15, kg = lane >> 4;
    const int rs = wave % RS, kq = wave / RS;
    const int n0 = blockIdx.x * (16 * RS) + rs * 16;
    int n = n0 + lrow;
    n = n < a.N ? n : a.N - 1;
    const bf16_t* w1 = a.w + (size_t)n * a.K + kg * 8;
    const bf16_t* w2 = SW ? a.w2 + (size_t)n * a.K + kg * 8 : nullptr;

    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (WL) {
        // ---- W: chunk j = k-steps 2j, 2j+1 of this wave's K-part, NM matrices x 16 rows x 128 B.  Request (q, h) moves
        // rows 8h..8h+7 of matrix q as full lines: lane -> (row lane/8, 16-B piece (lane%8) ^ row) so that the image
        // [8 rows][128 B] is read back conflict free (fragment of k-step t: piece (4t + kg) ^ row).  The ring is private:
        // only this wave's own counted vmcnt orders it, no barrier.
        char* wring = smem + NBUF * XBUF + wave * (RC * CB);
        const int nch = kpart >> 1;                        // K % (64 * KQ) == 0 (checked on the host)
        const int wr8 = lane >> 3;
        const bf16_t* wsrc[CI];
#pragma unroll
        for (int q = 0; q < NM; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int nr = n0 + h * 8 + wr8;
                nr = nr < a.N ? nr : a.N - 1;
                wsrc[q * 2 + h] = (q == 0 ? a.w : a.w2) + (size_t)nr * a.K + (size_t)kq * kpart * 32 + (((lane & 7) ^ wr8) << 3);
            }
        auto issue = [&](int j) __attribute__((always_inline)) {
            char* dst = wring + (j % RC) * CB;
#pragma unroll
            for (int i = 0; i < CI; ++i) glds16_nt(wsrc[i] + j * 64, dst + i * 1024);
        };
        int foff[2];                                       // this lane's fragment within a matrix image pair, k-step t
#pragma unroll
        for (int t = 0; t < 2; ++t) foff[t] = (lrow >> 3) * 1024 + (lrow & 7) * 128 + (((t * 4 + kg) ^ (lrow & 7)) << 4);
#pragma unroll
        for (int j = 0; j < RC; ++j)
            if (j < nch) issue(j);
        MID_STAMP(1);
        __builtin_amdgcn_s_barrier();      // x slice 0
        MID_STAMP(2);
        for (int s = 0; s < nsl; ++s) {
            const char* xb = smem + (s % NBUF) * XBUF;
#pragma unroll
            for (int c = 0; c < KPW; ++c) {
                const int kk = s * KPW + c, j = kk >> 1, t = kk & 1;
                if (t == 0) {              // chunk j has landed once at most min(nch - j, RC) - 1 younger chunks are outstanding
                    const int rem = (nch - j < RC ? nch - j : RC) - 1;
                    switch (rem) {
                        case 0: wait_vm<0>(); break;
                        case 1: wait_vm<CI>(); break;
                        case 2: wait_vm<(RC > 2 ? 2 : 1) * CI>(); break;
                        case 3: wait_vm<(RC > 3 ? 3 : 1) * CI>(); break;
                        case 4: wait_vm<(RC > 4 ? 4 : 1) * CI>(); break;
                        case 5: wait_vm<(RC > 5 ? 5 : 1) * CI>(); break;
                        case 6: wait_vm<(RC > 6 ? 6 : 1) * CI>(); break;
                        default: wait_vm<(RC - 1) * CI>(); break;
                    }
                }
                const char* wb = wring + (j % RC) * CB + foff[t];
                bf16x8 wf[NM];
#pragma unroll
                for (int q = 0; q < NM; ++q) wf[q] = *reinterpret_cast<const bf16x8*>(wb + q * 2048);
                const int ch = (kq * KPW + c) * 4 + kg;            // logical 16-B chunk of the x row
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (m0 + g * 32 < a.M) {
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xb + (g * 32 + lrow) * XROW + ((ch ^ lrow) << 4));
                        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xb + (g * 32 + 16 + lrow) * XROW + ((ch ^ lrow) << 4));
#pragma unroll
                        for (int q = 0; q < NM; ++q) {
                            acc[(g * NM + q) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], xl, acc[(g * NM + q) * 2], 0, 0, 0);
                            acc[(g * NM + q) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], xh, acc[(g * NM + q) * 2 + 1], 0, 0, 0);
                        }
                    }
                }
                if (t == 1 && j + RC < nch) {              // the chunk's fragments are in registers: refill its slot
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue(j + RC);
                }
            }
            __builtin_amdgcn_s_barrier();      // the loader waves arrive here once slice s+1 is in LDS
        }
        __builtin_amdgcn_s_barrier();          // every x read done, loader tail requests landed
        MID_STAMP(3);
    } else {
    bf16x8 wr[RING][NM][KPW];
        auto load_w = [&](bf16x8 (&wf)[NM][KPW], int s) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < KPW; ++c) {
                const int ko = (kq * kpart + s * KPW + c) * 32;
                wf[0][c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + ko));
                if (SW) wf[NM - 1][c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w2 + ko));
            }
        };
        auto body = [&](bf16x8 (&cur)[NM][KPW], bf16x8 (&ahead)[NM][KPW], int s) __attribute__((always_inline)) {
            if (s + PD < nsl) load_w(ahead, s + PD);
            const char* xb = smem + (s % NBUF) * XBUF;
#pragma unroll
            for (int c = 0; c < KPW; ++c) {
                const int ch = (kq * KPW + c) * 4 + kg;            // logical 16-B chunk of the row
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (m0 + g * 32 < a.M) {
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xb + (g * 32 + lrow) * XROW + ((ch ^ lrow) << 4));
                        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xb + (g * 32 + 16 + lrow) * XROW + ((ch ^ lrow) << 4));
#pragma unroll
                        for (int q = 0; q < NM; ++q) {
                            acc[(g * NM + q) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[q][c], xl, acc[(g * NM + q) * 2], 0, 0, 0);
                            acc[(g * NM + q) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[q][c], xh, acc[(g * NM + q) * 2 + 1], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_s_barrier();      // the loader waves arrive here once slice s+1 is in LDS
        };

#pragma unroll
        for (int p = 0; p < PD; ++p)
            if (p < nsl) load_w(wr[p], p);
        MID_STAMP(1);
        __builtin_amdgcn_s_barrier();          // x slice 0
        MID_STAMP(2);
        for (int s0 = 0; s0 < nsl; s0 += RING) {
#pragma unroll
            for (int r = 0; r < RING; ++r)
                if (s0 + r < nsl) body(wr[r], wr[(r + PD) % RING], s0 + r);
        }
        __builtin_amdgcn_s_barrier();          // every x read done, loader tail requests landed
        MID_STAMP(3);
    }
    // ---- the KQ partial tiles of a row set meet in LDS (every x buffer read ended at the last barrier).
    // Output tile pair t = (g, h) is finished by K-part t % KQ: every wave parks the tiles it does not
    // own, then sums its own in K-part order 0..KQ-1 (its register value taking its place in that order).
    f32x4* red = reinterpret_cast<f32x4*>(smem);           // [wave][NI][64 lanes]
#pragma unroll
    for (int t = 0; t < NG * 2; ++t) {
        if (t % KQ != kq) {
            const int g = t >> 1, h = t & 1;
#pragma unroll
            for (int q = 0; q < NM; ++q) red[(wave * NI + (g * NM + q) * 2 + h) * 64 + lane] = acc[(g * NM + q) * 2 + h];
        }
    }
    __syncthreads();                       // (the loader waves execute the matching barrier and leave)

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
    for (int t = 0; t < NG * 2; ++t) {
        if (t % KQ != kq) continue;
        const int g = t >> 1, h = t & 1;
        const int m = m0 + g * 32 + 16 * h + lrow;
        if (m0 + g * 32 >= a.M) continue;
        f32x4 tot[NM];
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int i = (g * NM + q) * 2 + h;
            f32x4 sum = kq == 0 ? acc[i] : red[((0 * RS + rs) * NI + i) * 64 + lane];
#pragma unroll
            for (int p = 1; p < KQ; ++p) sum += (p == kq) ? acc[i] : red[((p * RS + rs) * NI + i) * 64 + lane];
            tot[q] = sum;
        }
        if (m >= a.M) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (SW) {
                const float gt = rbf(tot[0][r]), up = rbf(tot[NM - 1][r]);
                o[r] = rbf(silu_fast(gt)) * up;
            } else {
                o[r] = rbf(tot[0][r]);
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
    MID_STAMP(6);
}

template <int EPI, int NG>
int launch_mid(const GemmArgs& a, hipStream_t s) {
    constexpr int xlds = NBUF * NG * 32 * XROW;
    dim3 grid(cdiv(a.N, 16 * MidShape<EPI>::RS), cdiv(a.M, NG * 32)), block(640);
    if constexpr (NG <= 2) {
        if (g_mid_wlds && a.K % (64 * MidShape<EPI>::KQ) == 0) {
            constexpr int lds = xlds + 8 * wl_ring_bytes<NG>();
            DH_MAX_LDS_ONCE((gemm_mid_kernel<EPI, NG, true>), lds);
            hipLaunchKernelGGL((gemm_mid_kernel<EPI, NG, true>), grid, block, lds, s, a);
            DH_LAUNCH_CHECK();
            return 0;
        }
    }
    if (xlds > 48 * 1024) DH_MAX_LDS_ONCE((gemm_mid_kernel<EPI, NG, false>), xlds);
    hipLaunchKernelGGL((gemm_mid_kernel<EPI, NG, false>), grid, block, xlds, s, a);
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
    return g_mid != 0 && a.K % (KSL * 32) == 0 && a.N % 16 == 0 && a.M <= 4096 &&
           (epilogue == DH_EPI_PLAIN || epilogue == DH_EPI_SWIGLU || epilogue == DH_EPI_ADAPTER);
}

int dh_linear_mid(const GemmArgs& a, int epilogue, hipStream_t s) {
    // > 64 rows: the tiled kernel runs the same K-part chains (gemm_dt.hip)
    const int seg = a.K / 32 / (epilogue == DH_EPI_SWIGLU ? MidShape<DH_EPI_SWIGLU>::KQ : MidShape<DH_EPI_PLAIN>::KQ);
    if (dh_linear_dt_ok(a, epilogue, seg)) return dh_linear_dt(a, epilogue, seg, s);
    switch (epilogue) {
        case DH_EPI_PLAIN: return launch_ng<DH_EPI_PLAIN>(a, s);
        case DH_EPI_SWIGLU: return launch_ng<DH_EPI_SWIGLU>(a, s);
        case DH_EPI_ADAPTER: return launch_ng<DH_EPI_ADAPTER>(a, s);
    }
    dh_set_error("dh_linear_mid: unsupported epilogue %d", epilogue);
    return 1;
}
