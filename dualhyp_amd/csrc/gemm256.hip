// Large-M prefill GEMM: y[M,N] = epilogue(x[M,K] · W[N,K]^T), 256 x 256 x 64 block tiles.
//
// Two kernels share the tile walk, the chain (one fp32 accumulator per output, k ascending in steps of 32: bit-identical outputs) and
// the epilogue function:
//   gemm_nt256w4_kernel  (round 3, the default, dh_set_tuning(1, 5)): FOUR waves with 128 x 128 each, 64-deep full-line stages,
//                        asm MFMAs with accumulators pinned to AGPRs, optional in-loop LoRA down-projection — described at the kernel;
//   gemm_nt256_kernel    (rounds 1-2; now the fallback for K < 128 / operands >= 4 GiB and the A/B variants 1..4): EIGHT waves,
//                        described in the rest of this header.
//
// Same orientation as gemm.hip (C^T tiles: A operand = W rows, B operand = x rows, a lane of the
// accumulator owns 4 consecutive n of one m), but sized so the LDS read stream stops being the
// limiter: a wave owns 128(n) x 64(m) = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16, i.e. 12 ds_read_b128
// feed 32 MFMAs per 32-deep k-step (24 B/clk/wave of LDS reads against 32 B/clk for a 64 x 64
// wave tile; at two waves per SIMD that is 192 of the CU's 256 B/clk).
//   block : 512 threads = 8 waves as 2(n) x 4(m); one block per CU
//   LDS   : 128 KiB (dynamic): four 32-deep stages of (W 16 KiB + x 16 KiB), global_load_lds 16 B per
//           lane with a source-side XOR swizzle (conflict-free fragment reads), counted s_waitcnt vmcnt
//   loop  : PIPE = 2 (default) PING-PONG: the two waves of a SIMD (the two wn halves) run half a step
//           apart — one owns the matrix pipe for 32 MFMAs while the other issues its DMA requests and
//           the 12 ds_read_b128 of its next fragments; two s_barrier per k-step.  +3..10 % over
//           PIPE = 1 (same stages, both waves in phase) and PIPE = 0 (BK = 64 double buffer).
//   tiles : walked in bands of 4 m-tiles inside each XCD's contiguous share (L2 reuse)
//   MFMA  : 16x16x32 (higher sustained clock than 32x32x16 on gfx950, MI355X_MICROARCH DVFS item 7)
// Epilogues as in gemm.hip; the LoRA rank-16 update is one zero-padded K=32 MFMA per 16 x 16 tile.
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm.h"

int g_w4_fast_epi = 7;  // dh_set_tuning(24, bits): bit 0 the fused-QKV epilogue of full tiles in its v_dot2_f32_bf16 form (g256_epilogue_qkv_fast), bit 1 the same arithmetic in the LoRA / residual epilogues; bit 2: a persistent block's tile start leaves the previous FULL tile's last stores in flight (counted wait); 0 = the round-3 forms (A/B)
int g_w4_persist_lora = 1;  // dh_set_tuning(30, 0 | 1): persistent blocks for the LoRA GEMM with the in-GEMM down-projection (attn proj of the prefill) and for plain + residual (mlp proj): the last two iterations of a tile request the next tile's first stages (CONT)
int g_w4_persist_qkv = 1;   // dh_set_tuning(25, 0 | 1): persistent blocks for the fused-QKV GEMM with the in-GEMM LoRA, the stage stream continuing across
                            // tiles (CONT).  With the next tile's stages requested from inside the epilogue (asm requests: 184 bytes of spills, every
                            // counted wait behind them waiting the DMA out) it measured 374-384 us per launch against 373-374 per-tile; as CONT
                            // (48 bytes of spills) 298.8 against 313.5 (tools/ab_persist_qkv.py), same bits
int g_w4_persist = 1;   // dh_set_tuning(22, 0 | 1 | 2): the 4-wave kernel walks the tiles with one block per CU: never / where the epilogue loads nothing / always

namespace {

constexpr int BT2 = 256;
constexpr int BK = 64;
constexpr int TILE_B = BT2 * BK * 2;   // 32 KiB
constexpr int G256_VIMG_BYTES = 4 * 16 * (128 + 8) * 2;   // g256_epilogue_qkv_fast: [16 keys][HS + 8] bf16 per wave, HS <= 128

#ifdef DH_G256_STAMPS   // diagnostic build only (tools/probe_gemm256.py): 100 MHz timestamps of wave 0 per block
__device__ unsigned long long g_g256_stamps[8192 * 16];
#define G256_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_g256_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// the 4-wave kernel's persistent blocks: stamps by TILE (vb = blockIdx.x + k gridDim.x), so every tile of a block keeps its own
#define G256_STAMP_T(vb, i) do { if (threadIdx.x == 0 && (vb) < 8192) g_g256_stamps[(vb) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int dh_debug_g256_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_g256_stamps), sizeof(g_g256_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define G256_STAMP(i)
#define G256_STAMP_T(vb, i)
#endif

// tiles are walked in bands of `gm` m-tiles, m fastest: the ~32 blocks an XCD runs at a time then form a
// gm x (32/gm) rectangle that shares gm x-tile streams and 32/gm W-tile streams through its L2
// instead of 1 + 32 (row-major); a.gm = 1 is the row-major order
template <int EPI>
__device__ __forceinline__ void g256_tile_origin(const GemmArgs& a, const int bid, int& m0, int& n0) {
    const int nwg = a.nb_n * a.nb_m;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int gm = a.gm, band = tile / (gm * a.nb_n), within = tile - band * (gm * a.nb_n);
    const int rows = min(gm, a.nb_m - band * gm);
    const int tm = band * gm + within % rows, tn = within / rows;
    m0 = tm * BT2;
    n0 = (EPI == DH_EPI_SWIGLU) ? tn * 128 : tn * BT2;
}

// ---- v_dot2_f32_bf16 as the epilogues' bf16 arithmetic (round 4) ---------------------------------------------------------------
// D = S0.lo * S1.lo + S0.hi * S1.hi + S2 on packed bf16 pairs, f32 result: one VALU issue slot (5.3 cycles for a wave alone on its
// SIMD, tools/valu_issue.hip — the price of a v_mul_f32), and what the eager-bf16 reference does between two rounding points is
// exactly such a term:
//   * a PRODUCT of two bf16 values: (x_e, x_e+1) . (c_e, 0) = x_e c_e — exact in f32 like v_mul_f32 on the expanded operands, with
//     no expansion of x (the v_cvt_pk_bf16_f32 result is the operand) and the negation of rotate-half as a free source modifier;
//   * the SUM of two bf16 values that sit in one register, (a, b) . (1, 1): v_cvt_pk_bf16_f32(acc, lora) IS (bf16(acc), bf16(lora)),
//     so "round both, add" is two instructions.  The dot's internal sum is not the IEEE f32 sum when the exponents are > 16 apart,
//     but ROUNDED TO BF16 (the next thing that happens to it at every use here) it is: tools/dot2_probe.hip, 2^31 operand pairs
//     incl. every bf16 pattern against near and far exponents: 0 differences (profiles/r04_dot2_probe.txt).
// Two consequences of the zero-padded second product: a non-finite x_e+1 turns x_e's product into NaN (0 * inf) — a row with an
// inf / NaN in q or k is lost in the attention anyway — and the sign of an exactly-zero product is +0 (numerically equal).
// HAZARD: on gfx90a+ a DOT instruction's result needs 3 wait states before another VALU instruction reads it (4 before one
// overwrites it; LLVM GCNHazardRecognizer DotWriteDifferentVALURead / ...VALUWrite).  The compiler pads this only for instructions it
// knows, and its own selection of the builtin is the accumulate form v_dot2c (D += ., i.e. a v_mov 0 per use), so the dots are
// written as asm BLOCKS whose instruction order keeps every dot result >= 3 instructions away from its reader; each block ends
// >= 4 instructions behind its last dot, and only v_cvt_pk results leave a block.  (First version: single-instruction asms, every
// rotated value wrong on the GPU — v_cvt_pk read its dot operands one instruction behind them.)  Blocks only ever read VALU
// results: MFMA results go through a compiler-visible v_cvt_pk first.
#define DH_ONES_BF16X2 0x3f803f80u
// y0 = cvt_pk(lo + hi of p0, of p1), y1 = cvt_pk(.. p2, .. p3), y2, y3 likewise from p4..p7: eight "round both, add", four packs
__device__ __forceinline__ void dot2_sum8_pack(uint32_t (&p)[8], uint32_t& y0, uint32_t& y1, uint32_t& y2, uint32_t& y3) {
    asm("v_dot2_f32_bf16 %0, %0, %12, 0\n\tv_dot2_f32_bf16 %1, %1, %12, 0\n\tv_dot2_f32_bf16 %2, %2, %12, 0\n\tv_dot2_f32_bf16 %3, %3, %12, 0\n\t"
        "v_dot2_f32_bf16 %4, %4, %12, 0\n\tv_dot2_f32_bf16 %5, %5, %12, 0\n\tv_dot2_f32_bf16 %6, %6, %12, 0\n\tv_dot2_f32_bf16 %7, %7, %12, 0\n\t"
        "v_cvt_pk_bf16_f32 %8, %0, %1\n\tv_cvt_pk_bf16_f32 %9, %2, %3\n\tv_cvt_pk_bf16_f32 %10, %4, %5\n\tv_cvt_pk_bf16_f32 %11, %6, %7"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
        : "s"(DH_ONES_BF16X2));
}
// One half of the rotation of a 16-column tile pair (4 outputs per lane): out_e = bf16(a_e ca_e) + bf16(+-b_e cb_e), e = 0..3, with
// a = (ax: e 0,1 | ay: e 2,3), b likewise, and per e the masked table operand (c, 0) for even e / (0, c) for odd e.  NEG: the second
// product negated (rotate-half's -x2 sin).  8 products, 4 cvt_pk (= both roundings), 4 sums, 2 packs; the s_nop keeps the last sums
// 3 wait states from their pack.
template <bool NEG>
__device__ __forceinline__ uint2 rope_half(uint32_t ax, uint32_t ay, uint32_t bx, uint32_t by, const uint32_t (&ca)[4], const uint32_t (&cb)[4]) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, o0, o1;
    if constexpr (NEG) {
        asm("v_dot2_f32_bf16 %0, %10, %14, 0\n\tv_dot2_f32_bf16 %1, %12, %18, 0 neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
            "v_dot2_f32_bf16 %2, %10, %15, 0\n\tv_dot2_f32_bf16 %3, %12, %19, 0 neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
            "v_dot2_f32_bf16 %4, %11, %16, 0\n\tv_dot2_f32_bf16 %5, %13, %20, 0 neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
            "v_dot2_f32_bf16 %6, %11, %17, 0\n\tv_dot2_f32_bf16 %7, %13, %21, 0 neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
            "v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3\n\tv_cvt_pk_bf16_f32 %4, %4, %5\n\tv_cvt_pk_bf16_f32 %6, %6, %7\n\t"
            "v_dot2_f32_bf16 %1, %0, %22, 0\n\tv_dot2_f32_bf16 %3, %2, %22, 0\n\tv_dot2_f32_bf16 %5, %4, %22, 0\n\tv_dot2_f32_bf16 %7, %6, %22, 0\n\t"
            "s_nop 1\n\tv_cvt_pk_bf16_f32 %8, %1, %3\n\tv_cvt_pk_bf16_f32 %9, %5, %7"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(o0), "=&v"(o1)
            : "v"(ax), "v"(ay), "v"(bx), "v"(by), "v"(ca[0]), "v"(ca[1]), "v"(ca[2]), "v"(ca[3]), "v"(cb[0]), "v"(cb[1]), "v"(cb[2]), "v"(cb[3]),
              "s"(DH_ONES_BF16X2));
    } else {
        asm("v_dot2_f32_bf16 %0, %10, %14, 0\n\tv_dot2_f32_bf16 %1, %12, %18, 0\n\t"
            "v_dot2_f32_bf16 %2, %10, %15, 0\n\tv_dot2_f32_bf16 %3, %12, %19, 0\n\t"
            "v_dot2_f32_bf16 %4, %11, %16, 0\n\tv_dot2_f32_bf16 %5, %13, %20, 0\n\t"
            "v_dot2_f32_bf16 %6, %11, %17, 0\n\tv_dot2_f32_bf16 %7, %13, %21, 0\n\t"
            "v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3\n\tv_cvt_pk_bf16_f32 %4, %4, %5\n\tv_cvt_pk_bf16_f32 %6, %6, %7\n\t"
            "v_dot2_f32_bf16 %1, %0, %22, 0\n\tv_dot2_f32_bf16 %3, %2, %22, 0\n\tv_dot2_f32_bf16 %5, %4, %22, 0\n\tv_dot2_f32_bf16 %7, %6, %22, 0\n\t"
            "s_nop 1\n\tv_cvt_pk_bf16_f32 %8, %1, %3\n\tv_cvt_pk_bf16_f32 %9, %5, %7"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(o0), "=&v"(o1)
            : "v"(ax), "v"(ay), "v"(bx), "v"(by), "v"(ca[0]), "v"(ca[1]), "v"(ca[2]), "v"(ca[3]), "v"(cb[0]), "v"(cb[1]), "v"(cb[2]), "v"(cb[3]),
              "s"(DH_ONES_BF16X2));
    }
    return make_uint2(o0, o1);
}

// The epilogue of a wave that owns 128 (n) x 16 MJ (m) of the block tile: acc[i][j] is the 16 x 16 tile of column tile i, row
// strip j.  MJ = 4: the 8-wave kernel (waves 2 x 4), MJ = 8: the 4-wave kernel (waves 2 x 2).
// FULL: the block tile lies inside the matrix — every bound below is then known at compile time, the epilogue has no exec-mask
// branches, and (what matters) the compiler can COUNT its loads: with predicated loads inside branches it waits vmcnt(0) before
// the first store, i.e. for the operands of every strip requested ahead.
// XS: the wave's x.A^T fragments come from an LDS image [256 rows of the block tile][16 bf16] (the 4-wave kernel's in-GEMM LoRA
// down-projection) instead of from the global xa tensor; the block tile then lies in ONE LoRA segment.
template <int EPI, bool RESID, int MJ, bool FULL, bool XS = false>
__device__ __forceinline__ void g256_epilogue(const GemmArgs& a, f32x4 (&acc)[8][MJ], const int m0, const int n0, const int wn,
                                              const int wm, const int lane, const char* xs = nullptr) {
    // SWIGLU with the RESID flag set is the TRAINING forward (dh_linear_swiglu_train_bf16): no residual is read; the epilogue also
    // stores the rounded pre-activations g = bf16(acc1) -> a.q_out and u = bf16(acc2) -> a.k_cache ([M, N] like y) that the backward needs
    constexpr bool GU = EPI == DH_EPI_SWIGLU && RESID;
    constexpr bool RES = RESID && !GU;
    const int frow = lane & 15, kg = lane >> 4;
    const int mw0 = m0 + wm * (MJ * 16);
    // ---------------------------------------------------------------- epilogue
    // acc[i][j][r]: n = nt + 4*kg + r , m = mt + frow: a lane holds 4 consecutive n (8 bytes of bf16) of one
    // row; the four kg lanes of a row hold one 16-column tile.  Tiles are finished in PAIRS and one
    // v_permlane16_swap per dword regroups them so that every lane owns 8 consecutive columns:
    //   kg0: tile i   cols 0..7     kg2: tile i   cols 8..15     kg1: tile i+1 cols 0..7     kg3: tile i+1 cols 8..15
    // -> one 16-byte store per lane and tile pair instead of two 8-byte ones (the tail of a tile is store-
    // issue-bound, cdna_hip_programming.md T21), 64 contiguous bytes per row and instruction.
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int NT = EPI == DH_EPI_SWIGLU ? 4 : 8;              // output tiles of 16 columns per wave
    const int nw0 = n0 + wn * (NT * 16);
    // Everything the epilogue reads from memory is requested in BATCHES before it is used: the LoRA B rows of the
    // wave's 8 column tiles once, and per row strip the 8 x·A^T fragments and the 8 residual words.  Issued one by one
    // inside the tile loop (load -> MFMA -> store, y may alias resid) they cost one L2 round trip each: ~14 us per
    // 256 x 256 tile of the proj GEMM, a quarter of its fixed per-tile cost.
    constexpr bool LORA = EPI == DH_EPI_LORA || EPI == DH_EPI_QKV;   // QKV: skipped at run time when lora_b is null
    bf16x8 lbv[LORA ? NT : 1];
    // (EPI_LORA always has its B; the fused-QKV epilogue runs with or without LoRA: a uniform run-time branch)
    const bool has_lora = EPI == DH_EPI_LORA || XS || (LORA && a.lora_b != nullptr);
    if (has_lora) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            int nn = nw0 + i * 16 + frow;
            if (!FULL) nn = nn < a.N ? nn : a.N - 1;
            // rank 16 zero-padded to K = 32.  Every lane loads (kg 2, 3 the bytes of kg 0, 1) and the upper half is zeroed by a select: a
            // load under `kg < 2 ? ... :` is an exec-masked branch per load, and loads inside divergent control flow are not counted
            // by the compiler's vmcnt bookkeeping (it waits vmcnt(0))
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.lora_b + (size_t)nn * 16 + (kg & 1) * 8);
            lbv[i] = kg < 2 ? v : zero8;
        }
    }
    // The per-strip operands (x·A^T fragment per tile pair, residual words) of strip j+1 are requested BEFORE strip j is
    // finished and stored: y may alias resid, so the compiler keeps every load behind the previous strip's stores, and a
    // strip then starts with an exposed L2 round trip (four per tile, tools/probe_gemm256.py).  Different strips touch
    // different rows, so the reordering is safe when y == resid.  LoRA segments start at multiples of 32 columns: one
    // fragment per PAIR of column tiles.
    // PD strips are in flight: one ahead hides a round trip behind a strip's ~0.6 us of work when a second wave shares the SIMD
    // (MJ = 4: the 8-wave kernel); the 4-wave kernel has nothing else to run and keeps three ahead (tools/probe_gemm256.py:
    // proj + LoRA + residual 20.8 -> ?? us per tile)
    constexpr int PD = MJ == 8 ? 4 : 2;
    bf16x8 xfv2[PD][LORA ? NT / 2 : 1];
    uint4 rrv2[PD][RES ? NT / 2 : 1];
    auto load_strip = [&](int j, bf16x8 (&xf)[LORA ? NT / 2 : 1], uint4 (&rr)[RES ? NT / 2 : 1]) __attribute__((always_inline)) {
        const int m = mw0 + j * 16 + frow;
        const bool m_ok = FULL || m < a.M;
        if constexpr (XS) {
            // 16 bytes of the row's 16 values (kg 2, 3 re-read kg 0, 1 and are zeroed: rank 16 zero-padded to K = 32)
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(xs + (wm * (MJ * 16) + j * 16 + frow) * 32 + (kg & 1) * 16);
#pragma unroll
            for (int i = 0; i < NT / 2; ++i) xf[i] = kg < 2 ? v : zero8;
        } else if (has_lora) {
            const int mm = m_ok ? m : a.M - 1;
#pragma unroll
            for (int i = 0; i < NT / 2; ++i) {
                const int nt = nw0 + i * 32;
                const int seg = (nt >= a.split0) + (nt >= a.split1);
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.xa + (size_t)mm * a.xa_ld + seg * 16 + (kg & 1) * 8);
                xf[i] = kg < 2 ? v : zero8;
            }
        }
        // residual in the layout of the paired 16-byte stores below (8 consecutive columns per lane): the add is done
        // after the permlane swap, on values that are already bf16-exact, so the rounding is that of bf16(resid + y)
        if (RES) {
#pragma unroll
            for (int ip = 0; ip < NT / 2; ++ip) {
                const int nb = nw0 + ip * 32;
                rr[ip] = (FULL || (m_ok && nb + 32 <= a.N))
                             ? *reinterpret_cast<const uint4*>(a.resid + (size_t)m * a.N + nb + (kg & 1) * 16 + (kg >> 1) * 8)
                             : make_uint4(0, 0, 0, 0);
            }
        }
    };
    // fused-QKV epilogue: the token positions / cache slots of all four row strips up front (the rope tables are
    // indexed by them: one dependent round trip per strip less)
    int posv[EPI == DH_EPI_QKV ? MJ : 1], slotv[EPI == DH_EPI_QKV ? MJ : 1];
    if constexpr (EPI == DH_EPI_QKV) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            int mm = mw0 + j * 16 + frow;
            if (!FULL) mm = mm < a.M ? mm : a.M - 1;
            posv[j] = a.tok_pos[mm];
            slotv[j] = a.tok_slot[mm];
        }
    }
    // ... and the rope rows of a strip (cos / sin of the row's position: c1, c2, s1, s2 per 16-column tile of a half head) one
    // strip ahead, once for both heads of the wave: loaded inside each head they were a dependent round trip per head and strip,
    // 16 of them per tile with nothing to run behind
    uint2 ropev[2][EPI == DH_EPI_QKV ? 16 : 1];
    auto load_rope = [&](int j, uint2 (&r)[EPI == DH_EPI_QKV ? 16 : 1]) __attribute__((always_inline)) {
        if constexpr (EPI == DH_EPI_QKV) {
            const int pos = posv[j];
            auto rows = [&](auto hs_c) __attribute__((always_inline)) {
                constexpr int HS = decltype(hs_c)::value, HALF = HS / 2, H2T = HS / 32;
                const bf16_t* cp = a.rope_cos + (size_t)pos * HS + 4 * kg;
                const bf16_t* sp = a.rope_sin + (size_t)pos * HS + 4 * kg;
#pragma unroll
                for (int t = 0; t < H2T; ++t) {
                    r[4 * t + 0] = *reinterpret_cast<const uint2*>(cp + 16 * t);
                    r[4 * t + 1] = *reinterpret_cast<const uint2*>(cp + HALF + 16 * t);
                    r[4 * t + 2] = *reinterpret_cast<const uint2*>(sp + 16 * t);
                    r[4 * t + 3] = *reinterpret_cast<const uint2*>(sp + HALF + 16 * t);
                }
            };
            if (a.hs == 64) rows(std::integral_constant<int, 64>{});
            else rows(std::integral_constant<int, 128>{});
        }
    };
    static_for<PD - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        load_strip(j, xfv2[j], rrv2[j]);
    });
    load_rope(0, ropev[0]);
    static_for<MJ>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int m = mw0 + j * 16 + frow;
        const bool m_ok = FULL || m < a.M;
        if constexpr (j + PD - 1 < MJ) load_strip(j + PD - 1, xfv2[(j + PD - 1) % PD], rrv2[(j + PD - 1) % PD]);
        bf16x8 (&xfv)[LORA ? NT / 2 : 1] = xfv2[j % PD];
        uint4 (&rrv)[RES ? NT / 2 : 1] = rrv2[j % PD];
        if constexpr (EPI == DH_EPI_QKV && j + 1 < MJ) load_rope(j + 1, ropev[(j + 1) & 1]);
        uint2 (&rp)[EPI == DH_EPI_QKV ? 16 : 1] = ropev[j & 1];
        if constexpr (EPI == DH_EPI_QKV) {
            // ---- fused-QKV epilogue: finish LoRA, then per head of the wave's 128 columns rotate (q, k) and scatter
            // q -> q_out [tok, head, hs], k -> K cache, v -> V^T cache (fragment order, common.h), exactly the arithmetic
            // of qkv_rope_cache_kernel (ger/model.py:216-259, 349-355) on the bf16-rounded projection values
            float ov[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[i][e] = rbf(acc[i][j][e]);
                if (has_lora) {
                    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
                    lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lbv[i], xfv[i >> 1], lacc, 0, 0, 0);
                    // lora_scale == 1 (alpha == r, the reference harnesses' setting): bf16(bf16(l) * 1) is bf16(l) — a wave-uniform
                    // branch drops a multiply and a rounding per element of this VALU-bound epilogue, same bits
                    float lt[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) lt[e] = rbf(lacc[e]);
                    if (a.lora_scale != 1.f) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) lt[e] = rbf(lt[e] * a.lora_scale);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[i][e] = rbf(ov[i][e] + lt[e]);
                }
            }
            {
                // every lane runs the arithmetic and the lane swaps (rows past M use the last row's position); only the
                // stores are predicated
                const int mm = m_ok ? m : a.M - 1;
                const int pos = posv[EPI == DH_EPI_QKV ? j : 0], slot = slotv[EPI == DH_EPI_QKV ? j : 0];
                const int qpk = a.n_head / a.n_groups;
                // two adjacent 16-column tiles -> 8 consecutive columns per lane (as the paired stores of the other epilogues)
                auto pair16 = [&](uint2 ta, uint2 tb) __attribute__((always_inline)) -> uint4 {
                    const auto rx = __builtin_amdgcn_permlane16_swap(ta.x, tb.x, false, false);
                    const auto ry = __builtin_amdgcn_permlane16_swap(ta.y, tb.y, false, false);
                    return make_uint4(rx[0], ry[0], rx[1], ry[1]);
                };
                const int cpair = (kg & 1) * 16 + (kg >> 1) * 8;        // this lane's 8 columns inside a 32-column pair
                auto head = [&](auto hs_c, int hh) __attribute__((always_inline)) {
                    constexpr int HS = decltype(hs_c)::value, HALF = HS / 2, H2T = HS / 32;   // tiles per half head
                    const int col0 = nw0 + hh * HS;
                    if (!FULL && col0 >= a.N) return;                                       // wave-uniform
                    const int hidx = col0 / HS, g = hidx / (qpk + 2), jh = hidx % (qpk + 2);
                    const int t0 = hh * 2 * H2T;
                    if (jh <= qpk) {
                        uint2 p1[H2T], p2[H2T];
#pragma unroll
                        for (int t = 0; t < H2T; ++t) {
                            const uint2 c1 = rp[EPI == DH_EPI_QKV ? 4 * t + 0 : 0], c2 = rp[EPI == DH_EPI_QKV ? 4 * t + 1 : 0];
                            const uint2 s1 = rp[EPI == DH_EPI_QKV ? 4 * t + 2 : 0], s2 = rp[EPI == DH_EPI_QKV ? 4 * t + 3 : 0];
                            const bf16_t *c1p = (const bf16_t*)&c1, *c2p = (const bf16_t*)&c2, *s1p = (const bf16_t*)&s1, *s2p = (const bf16_t*)&s2;
                            float o1[4], o2[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x1 = ov[t0 + t][e], x2 = ov[t0 + H2T + t][e];
                                o1[e] = rbf(x1 * bf2f(c1p[e])) + rbf(-x2 * bf2f(s1p[e]));
                                o2[e] = rbf(x2 * bf2f(c2p[e])) + rbf(x1 * bf2f(s2p[e]));
                            }
                            p1[t] = make_uint2(pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]));
                            p2[t] = make_uint2(pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3]));
                        }
                        bf16_t* qd = a.q_out + ((size_t)mm * a.n_head + g * qpk + jh) * HS;
                        bf16_t* kd = a.k_cache + ((size_t)slot * a.n_groups + g) * a.s_max * HS;
#pragma unroll
                        for (int tp = 0; tp < H2T / 2; ++tp) {
                            const uint4 w1 = pair16(p1[2 * tp], p1[2 * tp + 1]), w2 = pair16(p2[2 * tp], p2[2 * tp + 1]);
                            const int c = 32 * tp + cpair;
                            if (m_ok) {
                                if (jh < qpk) {
                                    *reinterpret_cast<uint4*>(qd + c) = w1;
                                    *reinterpret_cast<uint4*>(qd + HALF + c) = w2;
                                } else {                                   // 8 aligned channels of one key are one 16-byte run
                                    *reinterpret_cast<uint4*>(kd + kfrag_off<HS>(pos, c)) = w1;
                                    *reinterpret_cast<uint4*>(kd + kfrag_off<HS>(pos, HALF + c)) = w2;
                                }
                            }
                        }
                    } else if (m_ok) {
                        bf16_t* vd = a.vT_cache + ((size_t)slot * a.n_groups + g) * HS * a.s_max;
#pragma unroll
                        for (int t = 0; t < 2 * H2T; ++t)
#pragma unroll
                            for (int e = 0; e < 4; ++e) vd[vfrag_off<HS>(pos, 16 * t + 4 * kg + e)] = f2bf(ov[t0 + t][e]);
                    }
                };
                if (a.hs == 64) {
                    head(std::integral_constant<int, 64>{}, 0);
                    head(std::integral_constant<int, 64>{}, 1);
                } else {
                    head(std::integral_constant<int, 128>{}, 0);
                }
            }
            return;
        }
        // packed bf16 x4 of output tile i for this lane (all lanes run it: the swap below is wave-wide)
        auto tile_value = [&](int i) __attribute__((always_inline)) -> uint2 {
            const int nt = nw0 + i * 16, n = nt + kg * 4;
            const bool ok = FULL || (m_ok && n < a.N);
            float o[4];
            if (EPI == DH_EPI_SWIGLU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = rbf(acc[i][j][e]);
                    const float u = rbf(acc[i + 4][j][e]);
                    o[e] = rbf(silu_fast(g)) * u;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rbf(acc[i][j][e]);
                if (EPI == DH_EPI_LORA) {
                    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
                    lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lbv[LORA ? i : 0], xfv[LORA ? i >> 1 : 0], lacc, 0, 0, 0);
                    float lt[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) lt[e] = rbf(lacc[e]);
                    if (a.lora_scale != 1.f) {           // wave-uniform; see the QKV epilogue
#pragma unroll
                        for (int e = 0; e < 4; ++e) lt[e] = rbf(lt[e] * a.lora_scale);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rbf(o[e] + lt[e]);
                }
                if (EPI == DH_EPI_ADAPTER && ok) {
                    const uint2 sc = *reinterpret_cast<const uint2*>(a.vec_a + n);
                    const uint2 bi = *reinterpret_cast<const uint2*>(a.vec_b + n);
                    const bf16_t* sp = reinterpret_cast<const bf16_t*>(&sc);
                    const bf16_t* bp = reinterpret_cast<const bf16_t*>(&bi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rbf(bf2f(sp[e]) * rbf(o[e] + bf2f(bp[e])));
                }
            }
            return make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        };
#pragma unroll
        for (int ip = 0; ip < NT / 2; ++ip) {
            uint2 ta, tb;
#ifndef DH_W4_OLD_EPI
            // round 4, the 4-wave kernel's full tiles: "round both, add" of the LoRA finish as v_cvt_pk + v_dot2 (see dot2_sum8_pack)
            constexpr bool DOT = FULL && MJ == 8 && ((EPI == DH_EPI_LORA && XS) || (EPI == DH_EPI_PLAIN && RES));   // (LoRA with xa from memory: 8 spilled registers)
#else
            constexpr bool DOT = false;
#endif
            if constexpr (DOT && EPI == DH_EPI_LORA) {
                if (a.lora_scale == 1.f) {                       // wave-uniform (alpha == r in both reference harnesses)
                    const f32x4 la = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lbv[LORA ? 2 * ip : 0], xfv[LORA ? ip : 0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4 lb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lbv[LORA ? 2 * ip + 1 : 0], xfv[LORA ? ip : 0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    uint32_t pk[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pk[e] = pack2bf(acc[2 * ip][j][e], la[e]);
                        pk[4 + e] = pack2bf(acc[2 * ip + 1][j][e], lb[e]);
                    }
                    dot2_sum8_pack(pk, ta.x, ta.y, tb.x, tb.y);
                } else {
                    ta = tile_value(2 * ip);
                    tb = tile_value(2 * ip + 1);
                }
            } else if constexpr (DOT) {                          // plain: one v_cvt_pk per pair of values
                ta = make_uint2(pack2bf(acc[2 * ip][j][0], acc[2 * ip][j][1]), pack2bf(acc[2 * ip][j][2], acc[2 * ip][j][3]));
                tb = make_uint2(pack2bf(acc[2 * ip + 1][j][0], acc[2 * ip + 1][j][1]), pack2bf(acc[2 * ip + 1][j][2], acc[2 * ip + 1][j][3]));
            } else {
                ta = tile_value(2 * ip);
                tb = tile_value(2 * ip + 1);
            }
            const int nb = nw0 + ip * 32;                        // first column of the pair (wave-uniform)
            if constexpr (GU) {
                auto pk = [&](int i) __attribute__((always_inline)) -> uint2 {
                    return make_uint2(pack2bf(acc[i][j][0], acc[i][j][1]), pack2bf(acc[i][j][2], acc[i][j][3]));
                };
                auto put = [&](bf16_t* dst, uint2 va, uint2 vb) __attribute__((always_inline)) {
                    if (FULL || nb + 32 <= a.N) {
                        const auto rx = __builtin_amdgcn_permlane16_swap(va.x, vb.x, false, false);
                        const auto ry = __builtin_amdgcn_permlane16_swap(va.y, vb.y, false, false);
                        if (m_ok) *reinterpret_cast<uint4*>(dst + (size_t)m * a.N + nb + (kg & 1) * 16 + (kg >> 1) * 8) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    } else {
                        const int na = nb + kg * 4;
                        if (m_ok && na < a.N) *reinterpret_cast<uint2*>(dst + (size_t)m * a.N + na) = va;
                        if (m_ok && na + 16 < a.N) *reinterpret_cast<uint2*>(dst + (size_t)m * a.N + na + 16) = vb;
                    }
                };
                put(a.q_out, pk(2 * ip), pk(2 * ip + 1));
                put(a.k_cache, pk(2 * ip + 4), pk(2 * ip + 5));
            }
            if (FULL || nb + 32 <= a.N) {
                const auto rx = __builtin_amdgcn_permlane16_swap(ta.x, tb.x, false, false);
                const auto ry = __builtin_amdgcn_permlane16_swap(ta.y, tb.y, false, false);
                uint4 out = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                if constexpr (RES && DOT) {
                    // bf16(resid + y) on 8 values: (y, r) pairs by v_perm, then the same "add the two halves" block
                    const uint4 rr = rrv[RES ? ip : 0];
                    const uint32_t yv[4] = {out.x, out.y, out.z, out.w}, rv[4] = {rr.x, rr.y, rr.z, rr.w};
                    uint32_t pk[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pk[2 * q] = __builtin_amdgcn_perm(rv[q], yv[q], 0x05040100u);       // (y lo, r lo)
                        pk[2 * q + 1] = __builtin_amdgcn_perm(rv[q], yv[q], 0x07060302u);   // (y hi, r hi)
                    }
                    dot2_sum8_pack(pk, out.x, out.y, out.z, out.w);
                } else if (RES) {
                    const uint4 rr = rrv[RES ? ip : 0];
                    // resid_mul (wave-uniform; dh_linear_mul_bf16, 8-wave kernel only): the operand multiplies — the dropout mask of a LoRA branch's
                    // backward applied where the product is rounded, instead of one more pass over [tokens, d]
                    const bool mul = a.resid_mul != 0;
                    auto add2 = [mul](uint32_t y2, uint32_t r2) __attribute__((always_inline)) -> uint32_t {
                        const float rl = bf2f((bf16_t)(r2 & 0xffffu)), rh = bf2f((bf16_t)(r2 >> 16));
                        const float yl = bf2f((bf16_t)(y2 & 0xffffu)), yh = bf2f((bf16_t)(y2 >> 16));
                        return mul ? pack2bf(rl * yl, rh * yh) : pack2bf(rl + yl, rh + yh);
                    };
                    out = make_uint4(add2(out.x, rr.x), add2(out.y, rr.y), add2(out.z, rr.z), add2(out.w, rr.w));
                }
                if (m_ok) *reinterpret_cast<uint4*>(a.y + (size_t)m * a.N + nb + (kg & 1) * 16 + (kg >> 1) * 8) = out;
            } else {                                             // a pair straddling N: the narrow stores
                const int na = nb + kg * 4;
                auto with_resid = [&](uint2 t, int n) __attribute__((always_inline)) -> uint2 {
                    if (!RES) return t;
                    const uint2 rr = *reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.N + n);
                    const bool mul = a.resid_mul != 0;
                    auto op = [mul](uint32_t r2, uint32_t y2) __attribute__((always_inline)) -> uint32_t {
                        const float rl = bf2f((bf16_t)(r2 & 0xffffu)), rh = bf2f((bf16_t)(r2 >> 16));
                        const float yl = bf2f((bf16_t)(y2 & 0xffffu)), yh = bf2f((bf16_t)(y2 >> 16));
                        return mul ? pack2bf(rl * yl, rh * yh) : pack2bf(rl + yl, rh + yh);
                    };
                    return make_uint2(op(rr.x, t.x), op(rr.y, t.y));
                };
                if (m_ok && na < a.N) *reinterpret_cast<uint2*>(a.y + (size_t)m * a.N + na) = with_resid(ta, na);
                if (m_ok && na + 16 < a.N) *reinterpret_cast<uint2*>(a.y + (size_t)m * a.N + na + 16) = with_resid(tb, na + 16);
            }
        }
    });
}

// The fused-QKV epilogue of the 4-wave kernel for a FULL tile with the in-GEMM LoRA down-projection (the prefill of both model
// shapes): LoRA finish, rope on the q / k heads, q -> q_out, k -> K cache, v -> V^T cache — the arithmetic of g256_epilogue's QKV
// branch (= qkv_rope_cache_kernel, ger/model.py:216-259, 349-355; ger/lora.py:367-402), bit for bit, in ~340 instead of ~780 VALU
// instructions per 16-row strip:
//   * LoRA finish: p = cvt_pk(acc_e, lora_e) = (bf16(x W^T), bf16(x A^T B^T)); y_e = dot2_sum(p); packed y = cvt_pk(y_e, y_e+1) — 2.5
//     instructions per element behind the accumulator read (was 5: two round trips f32 -> bf16 -> f32 and an add);
//   * rope: the packed y pairs are the dot operands; cos / sin pairs are split ONCE per strip into (c, 0) / (0, c) masks shared by
//     the wave's heads; per output element two products, one cvt_pk (= both roundings), one dot2_sum, half a final cvt_pk;
//   * heads are classified (q / k / v, group, destination strides) once per tile, not per strip: no integer division in the strips;
//   * V^T: when a strip's 16 rows are 16 consecutive positions of one slot from a multiple of 16 (every strip of a prompt whose
//     packed start and cache position agree mod 16 — the benchmark's prompts), the strip's [16 keys][HS d] values go through a
//     wave-private LDS image and come back transposed by ds_read_b64_tr_b16: a lane holds, for one d, keys {0..3, 8..11} or
//     {4..7, 12..15} — the 8-key run of the cache's fragment order (common.h vfrag_off) — i.e. one 16-byte store instead of eight
//     2-byte ones.  Other strips keep the scatter.
// `vimg`: 4 KiB of LDS per wave (HS 128: [16][128 + 8] bf16).
template <int HS, bool SCALE1, class Hook>
__device__ __forceinline__ void g256_epilogue_qkv_fast(const GemmArgs& a, f32x4 (&acc)[8][8], const int m0, const int n0, const int wn,
                                                       const int wm, const int lane, const char* xs, char* vimg, Hook&& next_tile_dma) {
    constexpr int HALF = HS / 2, H2T = HS / 32, NH = 128 / HS, HT = HS / 16;      // tiles per half head, heads per wave, tiles per head
    constexpr int VSTR = (HS + 8) * 2;                                            // bytes per key row of the V image (8-byte aligned rows)
    const int frow = lane & 15, kg = lane >> 4;
    const int mw0 = m0 + wm * 128, nw0 = n0 + wn * 128;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---- once per tile: LoRA B rows of the wave's 8 column tiles, positions / slots of its 8 row strips, head classes
    bf16x8 lbv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int nn = nw0 + i * 16 + frow;
        nn = nn < a.N ? nn : a.N - 1;                    // (a ragged last column tile: its heads are skipped below)
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.lora_b + (size_t)nn * 16 + (kg & 1) * 8);
        lbv[i] = kg < 2 ? v : zero8;
    }
    const int qpk = a.n_head / a.n_groups;
    int hg[NH], hj[NH];                                  // wave-uniform: group and index inside the group (0..qpk-1 q, qpk k, qpk+1 v)
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        const int hidx = __builtin_amdgcn_readfirstlane((nw0 + hh * HS) / HS);
        hg[hh] = hidx / (qpk + 2);
        hj[hh] = hidx - hg[hh] * (qpk + 2);
    }
    const int cpair = (kg & 1) * 16 + (kg >> 1) * 8;     // this lane's 8 columns inside a 32-column pair after the lane swap
    auto pair16 = [&](uint2 ta, uint2 tb) __attribute__((always_inline)) -> uint4 {
        const auto rx = __builtin_amdgcn_permlane16_swap(ta.x, tb.x, false, false);
        const auto ry = __builtin_amdgcn_permlane16_swap(ta.y, tb.y, false, false);
        return make_uint4(rx[0], ry[0], rx[1], ry[1]);
    };
    // rope rows of a strip: per tile t of a half head c1 (first half), c2 (second half), s1, s2, 4 bf16 each
    auto load_rope = [&](int pos, uint2 (&r)[4 * H2T]) __attribute__((always_inline)) {
        const bf16_t* cp = a.rope_cos + (size_t)pos * HS + 4 * kg;
        const bf16_t* sp = a.rope_sin + (size_t)pos * HS + 4 * kg;
#pragma unroll
        for (int t = 0; t < H2T; ++t) {
            r[4 * t + 0] = *reinterpret_cast<const uint2*>(cp + 16 * t);
            r[4 * t + 1] = *reinterpret_cast<const uint2*>(cp + HALF + 16 * t);
            r[4 * t + 2] = *reinterpret_cast<const uint2*>(sp + 16 * t);
            r[4 * t + 3] = *reinterpret_cast<const uint2*>(sp + HALF + 16 * t);
        }
    };
    auto load_xf = [&](int j) __attribute__((always_inline)) -> bf16x8 {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xs + (wm * 128 + j * 16 + frow) * 32 + (kg & 1) * 16);
        return kg < 2 ? v : zero8;
    };
    auto load_pos = [&](int j, int& pos, int& slot) __attribute__((always_inline)) {
        int mm = mw0 + (j < 8 ? j : 7) * 16 + frow;
        mm = mm < a.M ? mm : a.M - 1;                    // rows past M (ragged last row band): computed like the last row, never stored
        pos = a.tok_pos[mm];
        slot = a.tok_slot[mm];
    };
    char* const vim = vimg + (wn * 2 + wm) * (16 * VSTR);
    // ---- the 8 row strips as a LOOP: one copy of the strip's code (~700 instructions) stays in the instruction cache.  Unrolled
    // (round 3's form, and this function's first version) the tile's epilogue is 50-100 KB of straight-line code executed once
    // per tile — more than the 64 KB instruction cache two CUs share — and ran at ~18 cycles per instruction whatever it did
    // (no MFMA, no loads, no stores: tools/probe_gemm256.py bit experiments, profiles/r04_probe_qkv_epilogue.txt): instruction
    // fetch.  The accumulators need static register numbers, so a strip's 32 values are read behind a compare chain on j;
    // positions / slots / rope rows / x.A^T fragments are software-pipelined through loop-carried registers.
    int pos0, slot0, pos1, slot1;
    uint2 rope0[4 * H2T];
    load_pos(0, pos0, slot0);
    load_pos(1, pos1, slot1);
    load_rope(pos0, rope0);
    bf16x8 xf0 = load_xf(0);
    G256_STAMP(4);
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
        // prefetch: position of strip j + 2, rope rows and x.A^T fragment of strip j + 1 (the last iterations re-load strip 7's)
        int pos2, slot2;
        load_pos(j + 2, pos2, slot2);
        uint2 rope1[4 * H2T];
        load_rope(pos1, rope1);
        const bf16x8 xf1 = load_xf(j + 1 < 8 ? j + 1 : 7);
        // persistent blocks: the NEXT tile's first two stages are requested here — behind the last loads this epilogue will consume
        // (strip 7's rope rows were just requested; loads return in order, so a DMA burst in front of them would delay them by the
        // 64 KiB it moves) and with two strips (~3.4 us) of work left to cover its latency
        if (j == 6) next_tile_dma();
        const int m = mw0 + j * 16 + frow, pos = pos0, slot = slot0;
        const bool m_ok = m < a.M;
        // the strip's eight LoRA MFMAs (rank 16 zero-padded to K = 32) back to back, straight into VGPRs: through the builtin the
        // compiler serialises them on one AGPR quad (MFMA, s_nop, four v_accvgpr_read, eight times).  The accumulator reads
        // below (>= 32 VALU instructions) stand between the last MFMA and the first use of a result.
        f32x4 lora[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // (s_nop 1: a VALU-written operand — the pipeline's xf0 = xf1 copy — in front of an MFMA the compiler cannot see)
            if (i == 0) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(lora[i]) : "v"(lbv[i]), "v"(xf0));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(lora[i]) : "v"(lbv[i]), "v"(xf0));
        }
        // this strip's 32 accumulator values per lane: static register indices behind a uniform compare tree on j
        float av[8][4];
        auto read_acc = [&](auto jc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;
            {
                // volatile: a plain copy is loop-invariant, and hoisted out of the loop all 256 values land in VGPRs at once (204 spilled)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float src = acc[i][J][e];
                        float dst;
                        if (i < 7) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(dst) : "a"(src));
                        else asm volatile("v_mov_b32 %0, %1" : "=v"(dst) : "v"(src));
                        av[i][e] = dst;
                    }
            }
        };
        if (j < 4) {
            if (j < 2) { if (j == 0) read_acc(std::integral_constant<int, 0>{}); else read_acc(std::integral_constant<int, 1>{}); }
            else { if (j == 2) read_acc(std::integral_constant<int, 2>{}); else read_acc(std::integral_constant<int, 3>{}); }
        } else {
            if (j < 6) { if (j == 4) read_acc(std::integral_constant<int, 4>{}); else read_acc(std::integral_constant<int, 5>{}); }
            else { if (j == 6) read_acc(std::integral_constant<int, 6>{}); else read_acc(std::integral_constant<int, 7>{}); }
        }
        asm volatile("s_nop 7" ::: "memory");          // the last MFMA is >= 32 instructions back; this covers a short path whatever the schedule
        // Pin the wait for this strip's rope rows HERE, on every path: a v head never reads them, and loads left pending across
        // that branch make every later reuse of their registers a conservative s_waitcnt vmcnt(N) that, the counter being in
        // order, also waits for the previous strip's stores.
#pragma unroll
        for (int k = 0; k < 4 * H2T; ++k) asm volatile("" : "+v"(rope0[k].x), "+v"(rope0[k].y));
        // ---- LoRA finish: y[i] = the wave's 8 column tiles of this strip as packed bf16 (columns 4 kg .. 4 kg + 3 of tile i)
        uint2 y[8];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            uint32_t pk[8];                          // (bf16(acc), bf16(LoRA term)) of the 2 x 4 elements of tiles i, i + 1
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 l = lora[i + h];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // SCALE1: lora_scale == 1 (alpha == r, both reference harnesses): bf16(bf16(l) * 1) is bf16(l)
                    if constexpr (SCALE1) pk[4 * h + e] = pack2bf(av[i + h][e], l[e]);
                    else pk[4 * h + e] = pack2bf(av[i + h][e], rbf(l[e]) * a.lora_scale);
                }
            }
            dot2_sum8_pack(pk, y[i].x, y[i].y, y[i + 1].x, y[i + 1].y);
        }
        // ---- cos / sin of the row as dot operands: [t][c1 c2 s1 s2][e]: (c, 0) for even e, (0, c) for odd e
        uint32_t cm[H2T][4][4];
#pragma unroll
        for (int t = 0; t < H2T; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint2 r = rope0[4 * t + k];
                cm[t][k][0] = r.x & 0xffffu; cm[t][k][1] = r.x & 0xffff0000u;
                cm[t][k][2] = r.y & 0xffffu; cm[t][k][3] = r.y & 0xffff0000u;
            }
        // is this strip 16 consecutive positions of one slot from a multiple of 16?  (the V^T fast path; wave-uniform)
        const int p0u = __builtin_amdgcn_readfirstlane(pos), s0u = __builtin_amdgcn_readfirstlane(slot);
        const bool vfast = __builtin_amdgcn_ballot_w64(pos == p0u + frow && slot == s0u && m_ok) == ~0ull && (p0u & 15) == 0;
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const int t0 = hh * HT, g = hg[hh], jh = hj[hh];
            if (nw0 + hh * HS >= a.N) continue;          // wave-uniform: a head past the last column
            if (jh <= qpk) {
                // ---- q or k head: rotate, then 16-byte stores of 8 consecutive columns per lane
                uint2 p1[H2T], p2[H2T];
#pragma unroll
                for (int t = 0; t < H2T; ++t) {
                    const uint2 x1 = y[t0 + t], x2 = y[t0 + H2T + t];
                    // o1 = bf16(x1 c1) + bf16(-x2 s1), o2 = bf16(x2 c2) + bf16(x1 s2)   (ger/model.py:349-355, every product and sum rounded)
                    p1[t] = rope_half<true>(x1.x, x1.y, x2.x, x2.y, cm[t][0], cm[t][2]);
                    p2[t] = rope_half<false>(x2.x, x2.y, x1.x, x1.y, cm[t][1], cm[t][3]);
                }
                // destination of the lane's first 16 bytes, stride between the two halves of the head and between 32-column pairs
                bf16_t* d0;
                int d_half, d_pair;
                if (jh < qpk) {
                    d0 = a.q_out + ((size_t)m * a.n_head + g * qpk + jh) * HS + cpair;
                    d_half = HALF;
                    d_pair = 32;
                } else {   // K cache, fragment order (common.h kfrag_off): 8 channels of one key are one 16-byte run
                    d0 = a.k_cache + ((size_t)slot * a.n_groups + g) * a.s_max * HS + kfrag_off<HS>(pos, cpair);
                    d_half = (HALF / 16) * 512;
                    d_pair = 1024;
                }
#pragma unroll
                for (int tp = 0; tp < H2T / 2; ++tp) {
                    const uint4 w1 = pair16(p1[2 * tp], p1[2 * tp + 1]), w2 = pair16(p2[2 * tp], p2[2 * tp + 1]);   // wave-wide swaps: every lane
                    if (m_ok) {
                        *reinterpret_cast<uint4*>(d0 + tp * d_pair) = w1;
                        *reinterpret_cast<uint4*>(d0 + d_half + tp * d_pair) = w2;
                    }
                }
            } else if (vfast) {
                // ---- v head, aligned strip: [16 keys][HS d] through the wave's LDS image, back transposed (4 keys x one d per read)
#pragma unroll
                for (int t = 0; t < HT; ++t) *reinterpret_cast<uint2*>(vim + frow * VSTR + (16 * t + 4 * kg) * 2) = y[t0 + t];
                bf16_t* vd = a.vT_cache + ((size_t)s0u * a.n_groups + g) * HS * a.s_max;
                const int tk = p0u >> 5, s2 = (p0u >> 4) & 1;
                // lane (group kg, i = frow = 4 q + p): block rows 4 lh + q (and + 8), columns 16 ct + 4 p ..; receives column 16 ct + frow
                const char* rd = vim + (frow >> 2) * VSTR + (frow & 3) * 8;
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
                for (int u = 0; u < HT / 4; ++u) {
                    const int ct = kg + 4 * u, dd = 16 * ct + frow;
#pragma unroll
                    for (int lh = 0; lh < 2; ++lh) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rd + (4 * lh) * VSTR + ct * 32));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rd + (4 * lh + 8) * VSTR + ct * 32));
                        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                        const size_t off = ((size_t)(((tk * (HS / 32) + (dd >> 5)) * 2 + s2) * 2 + lh) * 32 + (dd & 31)) * 8;
                        *reinterpret_cast<uint4*>(vd + off) = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    }
                }
            } else if (m_ok) {
                bf16_t* vd = a.vT_cache + ((size_t)slot * a.n_groups + g) * HS * a.s_max;
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    const uint2 v = y[t0 + t];
                    vd[vfrag_off<HS>(pos, 16 * t + 4 * kg + 0)] = (bf16_t)(v.x & 0xffffu);
                    vd[vfrag_off<HS>(pos, 16 * t + 4 * kg + 1)] = (bf16_t)(v.x >> 16);
                    vd[vfrag_off<HS>(pos, 16 * t + 4 * kg + 2)] = (bf16_t)(v.y & 0xffffu);
                    vd[vfrag_off<HS>(pos, 16 * t + 4 * kg + 3)] = (bf16_t)(v.y >> 16);
                }
            }
        }
        // rotate the software pipeline
        pos0 = pos1; slot0 = slot1; pos1 = pos2; slot1 = slot2;
        xf0 = xf1;
#pragma unroll
        for (int k = 0; k < 4 * H2T; ++k) rope0[k] = rope1[k];
    }
}

template <int EPI, bool RESID, int PIPE>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    G256_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 2, wm = wave & 3;
    const int frow = lane & 15, kg = lane >> 4;

    const int nwg = a.nb_n * a.nb_m;
    auto tile_origin = [&](int bid, int& m0, int& n0) __attribute__((always_inline)) { g256_tile_origin<EPI>(a, bid, m0, n0); };
    int m0, n0;
    tile_origin(blockIdx.x, m0, n0);
    // PIPE 3: PERSISTENT ping-pong — the grid is one block per CU and a block walks tiles bid, bid + grid, ...; the first
    // three stages of the NEXT tile are requested before the epilogue of the current one, so the epilogue (2.8-10.8 us),
    // the store drain, the block dispatch and the first-stage latency (2 us) of tools/probe_gemm256.py's timeline overlap
    constexpr bool PERSIST = PIPE == 3;
    constexpr int LOOP = PIPE == 3 ? 2 : PIPE;

    f32x4 acc[8][4];
    auto epilogue = [&](const int m0, const int n0) __attribute__((always_inline)) {
        G256_STAMP(2);
        g256_epilogue<EPI, RESID, 4, false>(a, acc, m0, n0, wn, wm, lane);
        G256_STAMP(3);
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    if constexpr (LOOP == 0) {
    // ---- staging sources: 4 (W) + 4 (x) one-KiB row groups per wave and K-tile
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = wave * 4 + j;
        const int row = R * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        {
            const bf16_t* base = a.w;
            int n;
            if (EPI == DH_EPI_SWIGLU) {
                // wave-row half wn holds 64 rows of fc_1 followed by the same 64 rows of fc_2
                const int within = row & 127;
                n = n0 + (row >> 7) * 64 + (within & 63);
                base = within >= 64 ? a.w2 : a.w;
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
        char* sA = smem + buf * 2 * TILE_B;
        char* sB = sA + TILE_B;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int R = wave * 4 + j;
            glds16(srcA[j] + kt * BK, sA + R * 1024);
            glds16(srcB[j] + kt * BK, sB + R * 1024);
        }
    };

    const int sw = (frow >> 1) & 7;
    const int offA = (wn * 128 + frow) * 128, offB = (wm * 64 + frow) * 128;

    const int nk = a.K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sA = smem + cur * 2 * TILE_B;
        const char* sB = sA + TILE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int co = ((ks * 4 + kg) ^ sw) << 4;
            bf16x8 fb[4], fa[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sB + offB + j * 2048 + co);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sA + offA + i * 2048 + co);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }
    epilogue(m0, n0);

    } else {
        // ---- deep pipeline: a stage is ONE 32-deep k-step (W 16 KiB + x 16 KiB), four stages in LDS.
        // Steady state of iteration k:  ds_read frags(k+1) | global_load_lds stage k+3 | 32 MFMA on
        // frags(k) | s_waitcnt vmcnt(4) (stage k+2 has landed, k+3 stays in flight) | s_barrier.
        // Fragment reads and DMA issue sit in front of the MFMA block they overlap with; nothing
        // drains to vmcnt(0) inside the loop (cdna_hip_programming.md T3+T4).
        constexpr int STG = 2 * 256 * 64;               // bytes per stage (A then B), rows of 64 B
        const bf16_t* srcA[2];
        const bf16_t* srcB[2];
        auto setup_src = [&](const int m0, const int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int R = wave * 2 + j;                   // 1-KiB group = 16 rows of 64 B
            const int row = R * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 1) & 3);
            {
                const bf16_t* base = a.w;
                int n;
                if (EPI == DH_EPI_SWIGLU) {
                    const int within = row & 127;
                    n = n0 + (row >> 7) * 64 + (within & 63);
                    base = within >= 64 ? a.w2 : a.w;
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
        };
        setup_src(m0, n0);
        auto stage = [&](int ks) {
            char* sA = smem + (ks & 3) * STG;
            char* sB = sA + STG / 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int R = wave * 2 + j;
                glds16(srcA[j] + ks * 32, sA + R * 1024);
                glds16(srcB[j] + ks * 32, sB + R * 1024);
            }
        };
        const int co = (kg ^ ((frow >> 1) & 3)) << 4;
        const int offA = (wn * 128 + frow) * 64 + co, offB = (wm * 64 + frow) * 64 + co;
        auto load_frags = [&](int ks, bf16x8 (&fa)[8], bf16x8 (&fb)[4]) {
            const char* sA = smem + (ks & 3) * STG;
            const char* sB = sA + STG / 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sB + offB + j * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sA + offA + i * 1024);
        };
        // first row of MFMAs (needs only the fragments read one step ago: the compiler's wait at this
        // point covers nothing newer), then the NEXT step's 12 ds_read_b128, then the other 28 MFMAs
        auto mma_head = [&](const bf16x8 (&fa)[8], const bf16x8 (&fb)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[j], acc[0][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma_tail = [&](const bf16x8 (&fa)[8], const bf16x8 (&fb)[4]) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 1; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        const int nks = a.K / 32;                         // even: K % 64 == 0
        stage(0);
        stage(1);
        if (2 < nks) stage(2);
        for (int vb = blockIdx.x;;) {
        if (2 < nks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (PERSIST: also the previous tile's stores — loads and
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          //  stores share the counter; at most 4 LOADS stay out)
        __builtin_amdgcn_s_barrier();
        G256_STAMP(1);
        bf16x8 faA[8], fbA[4], faB[8], fbB[4];
        if constexpr (LOOP == 1) {
        load_frags(0, faA, fbA);
        for (int k = 0; k < nks; k += 2) {
            // ---- even step: compute frags A (k), prefetch frags B (k+1)
            if (k + 3 < nks) stage(k + 3);
            mma_head(faA, fbA);
            load_frags(k + 1, faB, fbB);
            mma_tail(faA, fbA);
            if (k + 3 < nks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- odd step: compute frags B (k+1), prefetch frags A (k+2)
            if (k + 4 < nks) stage(k + 4);
            mma_head(faB, fbB);
            if (k + 2 < nks) load_frags(k + 2, faA, fbA);
            mma_tail(faB, fbB);
            if (k + 4 < nks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        } else {
        // ---- ping-pong: the two waves of a SIMD (wave w and w+4, i.e. the two wn halves) run half a
        // step apart, so while one owns the MFMA pipe (compute slot: 32 MFMAs on the fragments it
        // holds) the other is in its memory slot (DMA request for stage k+3, 12 ds_read_b128 of the
        // NEXT step's fragments, wait for its share of stage k+2).  Every slot ends with s_barrier; the
        // wn = 1 waves take one extra barrier up front and the wn = 0 waves one at the end.
        //   slot 2k   : wn0 C(k)   | wn1 M(k-1)
        //   slot 2k+1 : wn0 M(k)   | wn1 C(k)
        // Stage k+3 overwrites the buffer of step k-1, whose fragments were read in M(k-2) (>= 3 slots
        // earlier for either half); fragments of step k+1 are read in M(k), after both halves' shares
        // of stage k+1 were waited for in their M(k-1) (>= 1 barrier earlier).
        auto compute = [&](const bf16x8 (&fa)[8], const bf16x8 (&fb)[4]) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto memory = [&](int k, bf16x8 (&fa)[8], bf16x8 (&fb)[4]) __attribute__((always_inline)) {
            if (k + 3 < nks) stage(k + 3);
            if (k + 1 < nks) load_frags(k + 1, fa, fb);
            if (k + 3 < nks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        load_frags(0, faA, fbA);
        if (wn == 1) __builtin_amdgcn_s_barrier();
        for (int k = 0; k < nks; k += 2) {
            compute(faA, fbA);
            memory(k, faB, fbB);
            compute(faB, fbB);
            memory(k + 1, faA, fbA);
        }
        if (wn == 0) __builtin_amdgcn_s_barrier();
        }
        // every fragment of this tile has been read (the loop ends with a barrier): the ring is free for the next tile
        const int vb_next = vb + (int)gridDim.x;
        const bool more = PERSIST && vb_next < nwg;
        int m0n = 0, n0n = 0;
        if (more) {
            tile_origin(vb_next, m0n, n0n);
            setup_src(m0n, n0n);
            stage(0);
            stage(1);
            if (2 < nks) stage(2);
        }
        epilogue(m0, n0);
        if (!more) break;
        vb = vb_next;
        m0 = m0n;
        n0 = n0n;
        zero_acc();
        }
    }
}

// ---- four waves, one per SIMD, 128 (n) x 128 (m) per wave, operands fetched in FULL 128-byte lines -------------------------
// What the 8-wave kernel above leaves on the table (tools/gemm_yardstick.py, tools/pmc_gemm_ta.py: the vendor library's
// hand-written 256 x 256 x 64 kernel runs the K = 2048 / 5632 shapes of the prefill 1.2-1.3x faster):
//   * its 32-deep stages are rows of 64 B, i.e. every 128-B line of W and x is requested twice, half a line at a time:
//     TCP_TCC_READ_REQ 46.0 M against the library's 23.1 M for the same bytes, TA_ADDR_STALLED_BY_TC 2.3x.  Here a stage is 64
//     deep: one LDS-DMA instruction moves 8 rows x 128 B (whole lines) into a [row][128 B] image, XOR-swizzled on the
//     SOURCE side ((row >> 1) & 7, the ds_read_b128 lane groups of MI355X_MICROARCH.md) so fragment reads are conflict-free;
//   * a wave that owns 128 x 128 feeds 64 MFMAs per 32-deep k-step from 16 ds_read_b128 (128 KiB of LDS reads per 64-deep
//     stage instead of 192), and with one wave per SIMD the accumulators live in the AGPR half of the 512-register file.
// LDS: two stages of (W 32 KiB + x 32 KiB).  Registers hold BOTH k-steps of the current stage (F0, F1: 2 x 16 fragments), so
// a stage's buffer is free again half way through its own iteration:
//   phase A: 64 MFMAs on F0 | 16 reads F1 <- stage s (second k-step) | lgkmcnt(0), barrier B1: buffer s&1 is free
//            | 16 DMA pieces of stage s+2 -> buffer s&1 ...
//   phase B: ... | 64 MFMAs on F1 | vmcnt(16), barrier B2: stage s+1 has landed | 16 reads F0 <- stage s+1 (first k-step)
// A DMA piece has ~1.4 iterations to land.  The MFMAs are asm volatile with the accumulator pinned ("+a" for 56 tiles, "+v"
// for 8): with the builtin the register allocator keeps some accumulators in VGPRs and shuttles them through
// v_accvgpr_write / _read around every use (~230 moves per k-step); with all 64 in AGPRs it has no AGPR left for its own
// copies and does the same.  Memory operations do not cross a volatile asm, so the interleave in the source is the one issued.
// DMA pieces use the saddr form of global_load_lds (uniform base + one 32-bit VGPR offset per piece): no 64-bit VALU address
// arithmetic between the MFMAs.  (buffer_load ... lds through __builtin_amdgcn_raw_ptr_buffer_load_lds is not usable from
// C++: the compiler then puts s_waitcnt vmcnt(0) in front of every ds_read that may alias the DMA's LDS destination; written as
// asm — V# in SGPRs, SGPR byte offset, as the vendor library's kernel does — it measured 0-2 % slower than the saddr form.)
// ---- the 4-wave kernel's schedule: slots 0..127 of an iteration (MFMA t of phase A is slot t, of phase B slot 64 + t)
#ifndef DH_W4_RD1STEP
#define DH_W4_RD1STEP 2      // a read of F1 behind every RD1STEP-th MFMA from slot 0
#endif
#ifndef DH_W4_B1
#define DH_W4_B1 36          // barrier B1 (buffer s&1 free) behind this slot
#endif
#ifndef DH_W4_DMA0
#define DH_W4_DMA0 38        // first DMA piece of stage s+2
#endif
#ifndef DH_W4_DMANUM
#define DH_W4_DMANUM 5       // piece d goes out d * NUM / DEN slots later (a piece per 4-6 MFMAs: +7 % over one per 2-3,
#define DH_W4_DMADEN 1       //  profiles/r03_sweep_w4_*.txt — its issue takes the wave tens of cycles, MI355X_MICROARCH.md)
#endif
#ifndef DH_W4_B2
#define DH_W4_B2 80          // barrier B2 (stage s+1 landed) behind this slot
#endif
#ifndef DH_W4_RD0
#define DH_W4_RD0 81         // first read of the next stage's F0
#endif
#ifndef DH_W4_RD0STEP
#define DH_W4_RD0STEP 2
#endif
constexpr int w4_dma_slot(int d) { return DH_W4_DMA0 + d * DH_W4_DMANUM / DH_W4_DMADEN; }
constexpr int w4_dma_at(int g, int np = 16) {             // the piece issued behind slot g, or -1
    for (int d = 0; d < np; ++d)
        if (w4_dma_slot(d) == g) return d;
    return -1;
}
constexpr int w4_dma_before(int g, int np = 16) {         // pieces issued behind slots <= g
    int n = 0;
    for (int d = 0; d < np; ++d) n += w4_dma_slot(d) <= g;
    return n;
}
static_assert(w4_dma_slot(16) < 128 && 16 * DH_W4_RD1STEP < DH_W4_B1 && DH_W4_RD0 + 16 * DH_W4_RD0STEP < 128, "room for the 17th piece / read of the in-GEMM x.A^T");
static_assert(w4_dma_slot(15) < 128 && DH_W4_B1 < DH_W4_DMA0 && 15 * DH_W4_RD1STEP < DH_W4_B1, "B1 behind the last F1 read, in front of the first piece");
static_assert(DH_W4_RD0 > DH_W4_B2 && DH_W4_RD0 + 15 * DH_W4_RD0STEP < 128, "F0 reads behind B2");

// XA (LoRA epilogues): the LoRA down-projection bf16(x . A^T) of the block's 256 rows rides in the same K loop — the 16 rows of A
// of the tile's segment are a 17th DMA piece per stage (2 KiB behind the x image), one more fragment read per k-step, and four more
// MFMAs per wave and k-step on the x fragments already in registers (each wave of a wn pair takes four of the eight row tiles).
// The same chain per output as the separate x.A^T launch (one accumulator, k ascending in steps of 32): same bits.  After the loop
// the values are rounded to bf16 into an LDS image that the epilogue reads as MFMA fragments: no xa tensor, no x.A^T launch
// (2 x 32 us per layer and prefill launch), no xa loads in the epilogue.
template <int EPI, bool RESID, bool PERSIST, bool XA = false>
__global__ __launch_bounds__(256, 1) void gemm_nt256w4_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, kg = lane >> 4;
    int m0, n0;
    g256_tile_origin<EPI>(a, blockIdx.x, m0, n0);
    G256_STAMP(0);

    f32x4 acc[8][8];
    constexpr int OPB = TILE_B;                       // bytes per operand of a stage (256 rows x 128 B)
    constexpr int BUF = 2 * TILE_B + (XA ? 2048 : 0); // bytes per stage (XA: + 16 rows of A)
    constexpr int NP = XA ? 17 : 16;                  // DMA pieces per wave and stage
    constexpr bool CONT = PERSIST;                    // every persistent kernel streams its stages across tiles: see the tile loop
    const char* xs_img = smem + 2 * BUF;              // XA: [256][16] bf16, written after the K loop
    // ---- DMA sources: wave w moves row groups R = 8 w .. 8 w + 7 (8 rows x 128 B each) of W and of x
    uint32_t voA[8], voB[8], voX = 0;
    const bf16_t* la_seg = nullptr;                   // XA: first of the 16 rows of A of this tile's segment
    auto setup_src = [&](const int m0, const int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int R = wave * 8 + j;
        const int row = R * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int n;
        if (EPI == DH_EPI_SWIGLU) n = n0 + (row >> 7) * 64 + (row & 63);   // wave-row half wn: 64 rows of fc_1, then the same 64 of fc_2
        else n = n0 + row;
        n = n < a.N ? n : a.N - 1;
        voA[j] = ((uint32_t)n * (uint32_t)a.K + chunk * 8) * 2;
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        voB[j] = ((uint32_t)m * (uint32_t)a.K + chunk * 8) * 2;
    }
    if constexpr (XA) {
        // rows 8 (w & 1) .. + 7 of the segment's 16 (waves 2, 3 repeat 0, 1: identical bytes to the same place, and every wave
        // then has the same number of pieces in flight for the counted waits)
        const int row = (wave & 1) * 8 + (lane >> 3);
        voX = ((uint32_t)row * (uint32_t)a.K + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2;
        la_seg = a.lora_a + (size_t)((n0 >= a.split0) + (n0 >= a.split1)) * 16 * a.K;
    }
    };
    setup_src(m0, n0);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma = [&](int d, int st, int b) __attribute__((always_inline)) {      // piece d (0..15: A0 B0 A1 B1 ...) of stage st -> buffer b
        const int j = d >> 1, R = wave_u * 8 + j;
        char* dst = smem + b * BUF + R * 1024;
        auto issue = [&](const char* base, uint32_t vo, char* lds_dst) __attribute__((always_inline)) { glds16_saddr(base, vo, lds_dst); };   // common.h
        if (d == 16) {
            issue(reinterpret_cast<const char*>(la_seg) + (size_t)st * 128, voX, smem + b * BUF + 2 * OPB + (wave_u & 1) * 1024);
        } else if (d & 1) {
            issue(reinterpret_cast<const char*>(a.x) + (size_t)st * 128, voB[j], dst + OPB);
        } else {
            // rows 64..127 of a wave-row half: fc_2 (R & 8 == wave & 1)
            issue(reinterpret_cast<const char*>((EPI == DH_EPI_SWIGLU && (R & 8)) ? a.w2 : a.w) + (size_t)st * 128, voA[j], dst);
        }
    };
    // ---- fragment reads: row r of an operand image at r * 128, its 16-B piece c at position c ^ ((r >> 1) & 7)
    const int sw = (frow >> 1) & 7;
    const int offA = (wn * 128 + frow) * 128, offB = OPB + (wm * 128 + frow) * 128;
    const int co[2] = {(kg ^ sw) << 4, ((4 + kg) ^ sw) << 4};
    bf16x8 fax[2];                                    // XA: the A-of-LoRA fragment of each k-step of the current stage
    f32x4 xacc[4];                                    // XA: x.A^T of row tiles 4 wn .. 4 wn + 3 of this wave's eight
    bf16x8 xsel[4];                                   // XA: their x fragments of the current k-step
    auto rd = [&](int r, int b, int ks, bf16x8 (&fa)[8], bf16x8 (&fb)[8]) __attribute__((always_inline)) {   // read r: 0..15 = b0 a0 b1 a1 ...; 16: A of LoRA
        const char* base = smem + b * BUF + co[ks];
        if (r == 16) fax[ks] = *reinterpret_cast<const bf16x8*>(base + 2 * OPB + frow * 128);
        else if (r & 1) fa[r >> 1] = *reinterpret_cast<const bf16x8*>(base + offA + (r >> 1) * 2048);
        else fb[r >> 1] = *reinterpret_cast<const bf16x8*>(base + offB + (r >> 1) * 2048);
    };
    auto mfma = [&](int t, const bf16x8 (&fa)[8], const bf16x8 (&fb)[8]) __attribute__((always_inline)) {
        const int i = t >> 3, j = t & 7;
        if (i < 7) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[i]), "v"(fb[j]));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fa[i]), "v"(fb[j]));
    };
    const int nst = a.K / 64;                         // >= 2 (dh_linear_256 checks)
    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    // dstage: the stage whose pieces this iteration requests (st + 2; a persistent block's last two iterations: stages 0 and 1 of its
    // NEXT tile, with the DMA offsets already switched to that tile)
    auto iteration = [&](auto pre_c, auto nxt_c, int st, int dstage) __attribute__((always_inline)) {
        constexpr bool PRE = decltype(pre_c)::value, NXT = decltype(nxt_c)::value;
        const int b = st & 1;
        static_for<128>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 64) mfma(g, fa0, fb0);
            else mfma(g - 64, fa1, fb1);
            // XA: the k-step's four x.A^T MFMAs behind its 64.  Their x fragments (row tiles 4 wn .. 4 wn + 3) are SELECTED into `xsel`
            // by v_cndmask early in the phase: a branch on wn around the MFMAs makes the compiler merge the accumulators with
            // v_mov copies right behind them, and it inserts no wait states behind an MFMA it cannot see inside an asm (the copies
            // read half-finished sums: found as wrong values in all but the first tiles)
            if constexpr (XA && (g == 4 || g == 68)) {
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    union { bf16x8 v; uint32_t u[4]; } lo, hi, r;
                    lo.v = g < 64 ? fb0[jx] : fb1[jx];
                    hi.v = g < 64 ? fb0[4 + jx] : fb1[4 + jx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) r.u[e] = wn ? hi.u[e] : lo.u[e];
                    xsel[jx] = r.v;
                }
            }
            if constexpr (XA && (g == 63 || g == 127)) {
                asm volatile("s_nop 1" ::: "memory");          // a VALU-written operand in front of an MFMA the compiler cannot see
#pragma unroll
                for (int jx = 0; jx < 4; ++jx)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(xacc[jx]) : "v"(fax[g >> 6]), "v"(xsel[jx]));
            }
            if constexpr (g % DH_W4_RD1STEP == DH_W4_RD1STEP - 1 && g / DH_W4_RD1STEP < NP) rd(g / DH_W4_RD1STEP, b, 1, fa1, fb1);
            if constexpr (PRE && g == DH_W4_B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (PRE && w4_dma_at(g, NP) >= 0) dma(w4_dma_at(g, NP), dstage, b);
            if constexpr (NXT && g == DH_W4_B2) {
                // the pieces of stage s+2 issued so far stay in flight; everything older (stage s+1) has landed
                if constexpr (PRE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(w4_dma_before(DH_W4_B2, NP)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (NXT && g >= DH_W4_RD0 && (g - DH_W4_RD0) % DH_W4_RD0STEP == 0 && (g - DH_W4_RD0) / DH_W4_RD0STEP < NP)
                rd((g - DH_W4_RD0) / DH_W4_RD0STEP, b ^ 1, 0, fa0, fb0);
        });
    };
    auto first_stages = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < NP; ++d) dma(d, 0, 0);
#pragma unroll
        for (int d = 0; d < NP; ++d) dma(d, 1, 1);
    };
    first_stages();
    const int nwg = a.nb_n * a.nb_m;
    // PERSIST: the grid is one block per CU and a block walks tiles bid, bid + grid, ...; the first two stages of the NEXT tile are
    // requested by the last two iterations of the current one (CONT below), so the epilogue, its store drain, the block dispatch and
    // the first stage's latency overlap (the tile's K loop is only K / 64 = 32 iterations at K = 2048)
    bool prev_full = false;
    for (int vb = blockIdx.x, first = 1;; first = 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (XA) {
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) xacc[jx] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // first tile: stage 1 stays in flight.  Later tiles: the previous epilogue's stores share the counter with the loads, and the
        // ones issued BEHIND this tile's first-stage requests are the youngest operations in flight — a FULL tile's epilogue issues a
        // known number of them (one 16-byte store per column-tile pair and strip, unconditional), so the wait leaves exactly those in
        // flight instead of waiting out the write acknowledgements of the last strips (vmcnt(0)); ragged tiles and the fused-QKV epilogue
        // (stores that depend on the data) wait for everything
        constexpr bool GU_ = EPI == DH_EPI_SWIGLU && RESID;
        constexpr int STORES_PER_STRIP = EPI == DH_EPI_SWIGLU ? (GU_ ? 6 : 2) : 4;
        constexpr int YOUNGER_STORES = EPI == DH_EPI_QKV ? 0 : 8 * STORES_PER_STRIP;      // (every request of the next tile's stages is older than the epilogue)
        static_assert(YOUNGER_STORES <= 63, "vmcnt field");
        if (!PERSIST || first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        else if (prev_full && (a.fast_epi & 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        G256_STAMP_T(vb, 1);
#pragma unroll
        for (int r = 0; r < NP; ++r) rd(r, 0, 0, fa0, fb0);
        // The zeroed accumulators must BE in their registers a few cycles before the first asm MFMA reads them: the compiler sees no MFMA
        // there and may sink a v_accvgpr_write to the instruction in front of it (tools/check_asm_mfma.py found two in the CONT kernels).
        // Empty volatile asms tie every accumulator in front of an s_nop; volatile asms keep their order.
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (i < 7) asm volatile("" : "+a"(acc[i][j]));
                else asm volatile("" : "+v"(acc[i][j]));
            }
        if constexpr (XA) {
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) asm volatile("" : "+v"(xacc[jx]));
        }
        asm volatile("s_nop 3" ::: "memory");
        // CONT (the persistent LoRA / residual tiles): the stream of stage requests never drains — the last two iterations of a tile
        // request stages 0 and 1 of the block's NEXT tile into the buffers they free, behind the same MFMAs that hide every other
        // request (34 requests issued from inside the epilogue cost it ~4 us: tools/probe_w4_persistent.py).  The epilogue's own loads
        // are issued behind those 128 KiB and return behind them (its PD strips of prefetch absorb that).  nst even: the parity of the buffers carries over.
        const int vb_next = vb + (int)gridDim.x;
        const bool more = PERSIST && vb_next < nwg;
        int m0n = 0, n0n = 0;
        if (more) g256_tile_origin<EPI>(a, vb_next, m0n, n0n);
        int st = 0;
        for (; st + 2 < nst; ++st) iteration(T, T, st, st + 2);
        if constexpr (CONT) {
            // no branch around the two iterations (the accumulators would have to merge across it: 1 KiB of spills per lane): a block's LAST
            // tile requests its own first stages again — valid addresses, 128 KiB per block that nobody reads
            setup_src(more ? m0n : m0, more ? n0n : n0);     // (every request of THIS tile has been issued)
            iteration(T, T, st, 0);
            iteration(T, F, st + 1, 1);
        } else {
            iteration(F, T, st, 0);
            iteration(F, F, st + 1, 0);
        }
        // The compiler has no hazard model for an MFMA inside an asm: it would start the epilogue's v_accvgpr_read of a tile right
        // behind that tile's last MFMA.  Wait the matrix pipe out, then tie every accumulator to an (empty) volatile asm behind the
        // wait: volatile asms keep their order, and the epilogue's reads now depend on the later one.
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (i < 7) asm volatile("" : "+a"(acc[i][j]));
                else asm volatile("" : "+v"(acc[i][j]));
            }
        if constexpr (XA) {
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) asm volatile("" : "+v"(xacc[jx]));
            // bf16(x . A^T) of this wave's four row tiles -> the LDS image [row of the block tile][16]: lane (frow, kg) holds values
            // 4 kg .. 4 kg + 3 of row frow.  (The image lies behind the stage buffers: a persistent block's next stages may already
            // be landing; the previous tile's epilogue finished reading it before this tile's K loop — barriers in between.)
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const int row = wm * 128 + (4 * wn + jx) * 16 + frow;
                *reinterpret_cast<uint2*>(const_cast<char*>(xs_img) + row * 32 + kg * 8) =
                    make_uint2(pack2bf(xacc[jx][0], xacc[jx][1]), pack2bf(xacc[jx][2], xacc[jx][3]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // (round 4's first persistent form requested the next tile's stages from inside the epilogue through this hook; CONT replaced it)
        auto next_tile_dma = []() __attribute__((always_inline)) {};
        G256_STAMP_T(vb, 2);
        const bool full = m0 + BT2 <= a.M && n0 + (EPI == DH_EPI_SWIGLU ? 128 : BT2) <= a.N;
        if constexpr (EPI == DH_EPI_QKV && XA) {
            // the fused-QKV kernel with the in-GEMM LoRA down-projection has ONE epilogue, ragged tiles included: a second one in the
            // same kernel costs the register allocator's spill decisions in the code both share (51 scratch accesses per tile behind the K loop)
            char* vimg = smem + 2 * BUF + 256 * 32;
            if (a.hs == 64) {
                if (a.lora_scale == 1.f) g256_epilogue_qkv_fast<64, true>(a, acc, m0, n0, wn, wm, lane, xs_img, vimg, next_tile_dma);
                else g256_epilogue_qkv_fast<64, false>(a, acc, m0, n0, wn, wm, lane, xs_img, vimg, next_tile_dma);
            } else {
                if (a.lora_scale == 1.f) g256_epilogue_qkv_fast<128, true>(a, acc, m0, n0, wn, wm, lane, xs_img, vimg, next_tile_dma);
                else g256_epilogue_qkv_fast<128, false>(a, acc, m0, n0, wn, wm, lane, xs_img, vimg, next_tile_dma);
            }
        } else {
            if (full) g256_epilogue<EPI, RESID, 8, true, XA>(a, acc, m0, n0, wn, wm, lane, xs_img);
            else g256_epilogue<EPI, RESID, 8, false, XA>(a, acc, m0, n0, wn, wm, lane, xs_img);
        }
        G256_STAMP_T(vb, 3);
        if (!more) break;
        prev_full = full;
        vb = vb_next;
        m0 = m0n;
        n0 = n0n;
        if constexpr (CONT && EPI == DH_EPI_QKV) {
            asm volatile("" ::: "memory");
            setup_src(m0, n0);                       // the K loop's own copy of the DMA offsets (kept live across the fused-QKV epilogue they spill)
        }
    }
}

// two 64-deep stages at least; operands addressed through 32-bit buffer offsets
inline bool w4_ok(const GemmArgs& a) {
    return a.K >= 128 && (size_t)a.N * a.K * 2 < (1ull << 32) && (size_t)a.M * a.K * 2 < (1ull << 32);
}

inline int g256_cu_count() {
    static std::atomic<int> n_cu{0};
    int cu = n_cu.load(std::memory_order_relaxed);
    if (cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 256;
        n_cu.store(cu, std::memory_order_relaxed);
    }
    return cu;
}

template <int EPI, bool RESID, bool PERSIST, bool XA = false>
int launch_w4p(const GemmArgs& a, hipStream_t s) {
    auto kfn = gemm_nt256w4_kernel<EPI, RESID, PERSIST, XA>;
    // XA: + 16 rows of A per stage + the [256][16] x.A^T image; fused QKV: + the four waves' V images (g256_epilogue_qkv_fast)
    constexpr int lds = 4 * TILE_B + (XA ? 2 * 2048 + 256 * 32 : 0) + (XA && EPI == DH_EPI_QKV ? G256_VIMG_BYTES : 0);
    DH_MAX_LDS_ONCE(kfn, lds);
    int blocks = a.nb_n * a.nb_m;
    if (PERSIST) blocks = blocks < g256_cu_count() ? blocks : g256_cu_count();   // one block per CU walks the tiles
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

// the in-GEMM x.A^T needs every 256-column tile inside one LoRA segment
inline bool w4_xa_ok(const GemmArgs& a) {
    auto aligned = [&](int sp) { return sp >= a.N || sp % BT2 == 0; };
    return a.lora_a != nullptr && a.lora_b != nullptr && w4_ok(a) && aligned(a.split0) && aligned(a.split1) && g_gemm_variant == 5;
}

template <int EPI, bool RESID>
int launch_w4(const GemmArgs& a, hipStream_t s) {
    // The next tile's first stages are requested in front of the epilogue; loads return in order, so an epilogue that loads
    // (residual, x.A^T fragments, rope rows) would wait for those 64 KiB behind its first operand: persistent blocks only where
    // the epilogue reads nothing (bench: all-persistent 715 utt/s, none 724)
    if constexpr (EPI == DH_EPI_QKV) {
        // round 4 (VERDICT r03 #1a): persistent fused-QKV tiles, the K loop's last two iterations requesting the next tile's stages (CONT)
        if (a.lora_a != nullptr) return g_w4_persist_qkv && (a.K / 64) % 2 == 0 ? launch_w4p<EPI, RESID, true, true>(a, s) : launch_w4p<EPI, RESID, false, true>(a, s);
    }
    if constexpr (EPI == DH_EPI_LORA) {
        // w4_xa_ok checked by the caller.  Persistent form (round 4): the stage stream continues across tiles (CONT in the kernel)
        // (an even number of 64-deep stages: the buffers' parity carries from a tile's last stage to the next tile's first)
        if (a.lora_a != nullptr) return g_w4_persist_lora && (a.K / 64) % 2 == 0 ? launch_w4p<EPI, RESID, true, true>(a, s) : launch_w4p<EPI, RESID, false, true>(a, s);
    }
    const bool cont_ok = (a.K / 64) % 2 == 0;       // persistent blocks stream their stages across tiles (CONT): an even number of stages
    const bool persist = (g_w4_persist == 2 || (g_w4_persist == 1 && ((!RESID && EPI == DH_EPI_PLAIN) || EPI == DH_EPI_SWIGLU)) ||
                          (g_w4_persist_lora && ((RESID && EPI == DH_EPI_PLAIN) || EPI == DH_EPI_LORA))) && cont_ok;      // (LoRA here: xa from memory, the fine-tune)
    return persist ? launch_w4p<EPI, RESID, true>(a, s) : launch_w4p<EPI, RESID, false>(a, s);
}

template <int EPI, bool RESID, int PIPE>
int launch_one(const GemmArgs& a, hipStream_t s) {
    auto kfn = gemm_nt256_kernel<EPI, RESID, PIPE>;
    DH_MAX_LDS_ONCE(kfn, 4 * TILE_B);
    int blocks = a.nb_n * a.nb_m;
    if (PIPE == 3) {                     // persistent: one block per CU walks the tiles
        static std::atomic<int> n_cu{0};
        int cu = n_cu.load(std::memory_order_relaxed);
        if (cu == 0) {
            int dev = 0;
            DH_HIP(hipGetDevice(&dev));
            DH_HIP(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev));
            n_cu.store(cu, std::memory_order_relaxed);
        }
        blocks = blocks < cu ? blocks : cu;
    }
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), 4 * TILE_B, s, a);
    DH_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch(const GemmArgs& a, hipStream_t s) {
    // 5 (default): the 4-wave full-line kernel; 8-wave kernel: 1 ping-pong, 2 4-stage pipeline with both waves of a SIMD in phase,
    // 3 BK = 64 double buffer, 4 persistent ping-pong where its loop-carried state fits the registers (the LoRA epilogues spill 47-116 VGPRs)
    if (g_gemm_variant == 5 && w4_ok(a)) return a.resid ? launch_w4<EPI, true>(a, s) : launch_w4<EPI, false>(a, s);
    if (g_gemm_variant == 4 && (EPI == DH_EPI_PLAIN || EPI == DH_EPI_SWIGLU))
        return a.resid ? launch_one<EPI, true, 3>(a, s) : launch_one<EPI, false, 3>(a, s);
    if (g_gemm_variant == 3) return a.resid ? launch_one<EPI, true, 0>(a, s) : launch_one<EPI, false, 0>(a, s);
    if (g_gemm_variant == 2) return a.resid ? launch_one<EPI, true, 1>(a, s) : launch_one<EPI, false, 1>(a, s);
    return a.resid ? launch_one<EPI, true, 2>(a, s) : launch_one<EPI, false, 2>(a, s);
}

}  // namespace

// argument checks are done by dh_linear_impl (gemm.hip)
int g_gemm_gm = 0;   // 0: default 4 (4 x 8 tile rectangles per XCD: fewest operand streams, tools/tune_gemm.py + PMC), else forced

bool dh_linear_256_xa_ok(const GemmArgs& a, int epilogue) {
    return (epilogue == DH_EPI_LORA || epilogue == DH_EPI_QKV) && w4_xa_ok(a);
}

int dh_linear_256(GemmArgs a, int epilogue, hipStream_t s) {
    a.gm = g_gemm_gm > 0 ? g_gemm_gm : 4;
    a.fast_epi = g_w4_fast_epi;
    a.nb_m = cdiv(a.M, BT2);
    a.nb_n = (epilogue == DH_EPI_SWIGLU) ? cdiv(a.N, 128) : cdiv(a.N, BT2);
    if (a.resid_mul) return launch_one<DH_EPI_PLAIN, true, 2>(a, s);      // (checked by dh_linear_mul_bf16: plain epilogue, multiplier present)
    switch (epilogue) {
        case DH_EPI_PLAIN: return launch<DH_EPI_PLAIN>(a, s);
        case DH_EPI_LORA: return launch<DH_EPI_LORA>(a, s);
        case DH_EPI_SWIGLU: return launch<DH_EPI_SWIGLU>(a, s);
        case DH_EPI_ADAPTER: return launch<DH_EPI_ADAPTER>(a, s);
        case DH_EPI_QKV: return g_gemm_variant == 5 && w4_ok(a) ? launch_w4<DH_EPI_QKV, false>(a, s) : launch_one<DH_EPI_QKV, false, 2>(a, s);
    }
    dh_set_error("dh_linear_bf16: unknown epilogue %d", epilogue);
    return 1;
}
