// Split-precision GEMM, MEDIUM-problem kernel: 128 ... 2047 rows (one utterance - BASELINE config 2 runs 1000 rows per
// evaluation, reference monologue_generation.py:259-304 - the last bin of a ragged directory, the HuBERT / text2semantic
// encoders).  Same contract as cvx_gemm_f16x3 (reference acoustic.py:225-246, :306-310), same interleaved operands
// ([hi 32 | lo 32] fp16 per K-tile of a row = one 128-byte line), same v_mfma_f32_16x16x32_f16 products with swapped operands
// and the transpose-free epilogues of gemm_p8s_epi.h as the 256 x 256 kernel (gemm_f16x3_p8s.hip) - but a problem of 1000 rows
// has 16 such tiles for 256 CUs.  Here:
//
//   * 128 x 128 output tile, 512 threads = two wave GROUPS of four waves (2 x 2 waves of 64 x 64); waves w and w + 4 share a
//     SIMD.  The groups split the K range by PARITY: group g multiplies K-tiles g, g + 2, ... of the tile into its own
//     accumulators (a block-local split-K).  While one group issues the 48 MFMAs of a K-tile, the other reads the fragments of
//     its next K-tile from LDS (16 ds_read_b128) and issues that group's LDS-DMA, so a SIMD's matrix pipe always has a wave
//     feeding it - the ping-pong of the eight-phase kernel with ONE barrier interval per K-tile instead of four quadrant phases
//     (a 64 x 64 wave tile needs all of its A and W fragments at once: nothing to consume progressively).
//   * Ring of FIVE K-tile buffers (A | W, 32 KiB each = all 160 KiB of LDS).  Group g fetches its own K-tiles: K-tile t + 4 is
//     requested in the load segment of K-tile t, into the buffer K-tile t - 1 was read from one interval earlier, and retired by
//     a COUNTED s_waitcnt vmcnt(8) at the end of the group's compute segment + the interval barrier, i.e. three to four
//     intervals (~ 1.2 us) after the request.  RAW: issuing waves wait, barrier, readers read.  WAR: a buffer's last reads are
//     completed (lgkmcnt(0)) before the barrier that ends their interval; the refill is issued after it.
//   * After the K loop the groups exchange HALF of their accumulators through LDS (rows 0-31 of every wave tile end up in
//     group 0, rows 32-63 in group 1; the sum is always group 0 + group 1: deterministic), so all eight waves run the epilogue.
//   * XCD-aware block -> tile map: an XCD (block & 7, observed placement - speed only) owns whole W panels: the (at most 16) row
//     tiles that share a W panel are dispatched back to back to ONE XCD, which fetches the panel from HBM once and serves the
//     other seven reads from its L2; the (small) A operand is what every XCD reads.  tools/archive/dma_probe2.hip: the LDS-DMA stream of
//     the to_qkv / ff1 shapes at 1000 rows runs at 43-45 B/clk/CU with this map against 24-30 with row panels per XCD (the
//     large-problem map), and a 128 x 128 x 32 K-tile at full matrix rate needs 42.
//   * Optional split-K over blocks (ksplit > 1: K-slices ride in the unit index of the map): raw fp32 partial tiles to the
//     caller's workspace [ksplit][M][N], finished by splitk_reduce_kernel (gemm_f16x3.hip) in a fixed order.
#include "gemm_p8s_epi.h"

namespace {

constexpr int MT_B = 128 * 128;                    // bytes per operand tile: 128 rows x one 128-byte line
constexpr int MBUF_B = 2 * MT_B;                   // A | W
constexpr int NBUF = 5;
constexpr int MLDS_B = NBUF * MBUF_B;              // 160 KiB

#ifdef CVX_DEV_FLAGS          // per-block s_memtime stamps of waves 0 and 4 (tools/archive/gemm_small_trace.py; never in the shipped library)
#define CVX_P8M_STAMP(i) do { if (tr) { tr[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define CVX_P8M_STAMP(i) do { } while (0)
#endif
#ifndef CVX_P8M_DMA_FIRST
#define CVX_P8M_DMA_FIRST 0         // dev A/B: the load segment requests K-tile t + 4 before (1) or after (0) its 16 fragment reads
#endif
#ifndef CVX_P8M_PERM
#define CVX_P8M_PERM 1              // dev A/B: 0 = 8-byte pair stores in the to_qkv / ff1 epilogues (rounds 1-4 before the permutation)
#endif
#define CVX_P8M_BARRIER() asm volatile("s_barrier" ::: "memory")
#define CVX_P8M_WAIT_DMA() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define CVX_P8M_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// K loop of one 128 x 128 tile over K-tiles [kt0, kt0 + nk) (32 k each) of [A | A2] . W^T; this wave's group takes the K-tiles
// of its parity.  acc[mi][ni]: 16 x 16 blocks of the wave's 64 x 64 sub-tile (layout: gemm_p8s_epi.h).
// PERM: LDS row rho of the W tile is filled from weight row perm32(rho) (gemm_p8s_epi.h): a lane's tile pair = 8 consecutive columns
template <bool HAS_A2, bool SWAP, bool PERM = false>
__device__ __forceinline__ void tile_mainloop_m(const cvx_gemm_args& p, const PreSplitA& A, const f16* __restrict__ W, char* smem,
                                                int m0, int n0, int kt0, int nk, int lane, int grp, int w4, int wr, int wc,
                                                f32x4 (&acc)[4][4], unsigned long long* tr)
{
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // DMA: wave w4 of a group moves rows [32 w4, 32 w4 + 32) of the A tile and of the W tile, 8 rows x 128 bytes per piece
    uint32_t offA[4], offA2[4], offW[4];
    const int64_t ldaB = A.ld * 2, lda2B = A.ld2 * 2, ldwB = p.ldw * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 32 * w4 + 8 * j + (lane >> 3);
        const uint32_t c = (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
        const uint32_t ga = (uint32_t)min(m0 + r, p.M - 1), gb = (uint32_t)min(n0 + (PERM ? perm32(r) : r), p.N - 1);
        offA[j] = ga * (uint32_t)ldaB + c;             // (every operand spans < 4 GiB: checked by the launcher)
        offA2[j] = HAS_A2 ? ga * (uint32_t)lda2B + c : 0u;
        offW[j] = gb * (uint32_t)ldwB + c;
    }
    const uint32_t dst0 = (uint32_t)(32 * w4 * 128);
    const int t_sw = HAS_A2 ? p.K1 / 32 : 0x7fffffff;
    const char* const a1base = reinterpret_cast<const char*>(A.hi);
    const char* const a2base = reinterpret_cast<const char*>(A.hi2);
    const char* const wbase = reinterpret_cast<const char*>(W);

    // all 8 pieces of K-tile t (local index) -> ring buffer `buf`.  t >= nk: dummy 16-byte re-reads into the (dead) buffer, which
    // keep the vmcnt arithmetic uniform through the tail
    auto issue = [&](int t, int buf) {
        const bool live = t < nk;
        const int u = kt0 + t;
        const uint32_t b = lds0 + (uint32_t)buf * MBUF_B + dst0;
        const char* abase = a1base + (int64_t)u * 128;
        uint32_t a0 = offA[0], a1 = offA[1], a2 = offA[2], a3 = offA[3];
        if constexpr (HAS_A2) {
            if (u >= t_sw) { abase = a2base + (int64_t)(u - t_sw) * 128; a0 = offA2[0]; a1 = offA2[1]; a2 = offA2[2]; a3 = offA2[3]; }
        }
        const char* wb = wbase + (int64_t)u * 128;
        uint32_t w0 = offW[0], w1 = offW[1], w2 = offW[2], w3 = offW[3];
        if (!live) { abase = wbase; wb = wbase; a0 = a1 = a2 = a3 = 0u; w0 = w1 = w2 = w3 = 0u; }
        dma2(a0, a1, b, b + 1024u, abase);
        dma2(w0, w1, b + MT_B, b + MT_B + 1024u, wb);
        dma2(a2, a3, b + 2048u, b + 3072u, abase);
        dma2(w2, w3, b + MT_B + 2048u, b + MT_B + 3072u, wb);
    };

    // fragment addresses inside a buffer: row (lane & 15) of a 16-row MFMA tile, k chunk (lane >> 4) for hi / 4 + (lane >> 4) for lo
    const int lr = lane & 15, kg = lane >> 4, sw8 = lr >> 1;
    const int aoh = (wr * 64 + lr) * 128 + 16 * (kg ^ sw8);
    const int aol = (wr * 64 + lr) * 128 + 16 * ((4 + kg) ^ sw8);
    const int boh = MT_B + (wc * 64 + lr) * 128 + 16 * (kg ^ sw8);
    const int bol = MT_B + (wc * 64 + lr) * 128 + 16 * ((4 + kg) ^ sw8);

    f16x8 fah[4], fal[4], fbh[4], fbl[4];

    // prologue: the group's first two K-tiles; the first one has landed behind the counted wait + barrier
    issue(grp, grp);
    issue(grp + 2, grp + 2);
    CVX_P8M_WAIT_DMA();
    CVX_P8M_BARRIER();
    CVX_P8M_STAMP(1);
    if (grp == 1) CVX_P8M_BARRIER();                    // group 1 runs one interval behind group 0

#define CVX_P8M_MM(x, y, c) (SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0))
    const int nit = (nk + 1) >> 1;
    int t = grp, bi = grp;
    for (int it = 0; it < nit; ++it, t += 2) {
        // ---- load segment (the other group multiplies): fragments of K-tile t, request K-tile t + 4
        const char* sb = smem + bi * MBUF_B;
        const int bn = bi + 4 >= NBUF ? bi + 4 - NBUF : bi + 4;
#if CVX_P8M_DMA_FIRST
        issue(t + 4, bn);
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fbh[i] = *reinterpret_cast<const f16x8*>(sb + i * 16 * 128 + boh);
            fbl[i] = *reinterpret_cast<const f16x8*>(sb + i * 16 * 128 + bol);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fah[i] = *reinterpret_cast<const f16x8*>(sb + i * 16 * 128 + aoh);
            fal[i] = *reinterpret_cast<const f16x8*>(sb + i * 16 * 128 + aol);
        }
#if !CVX_P8M_DMA_FIRST
        issue(t + 4, bn);
#endif
        CVX_P8M_WAIT_LDS();
        CVX_P8M_BARRIER();
        // ---- compute segment: 4 x 4 tiles x three terms = 48 MFMAs, term-major (consecutive MFMAs hit different accumulators)
        if (t < nk) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = CVX_P8M_MM(fal[i], fbh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = CVX_P8M_MM(fah[i], fbl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = CVX_P8M_MM(fah[i], fbh[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        CVX_P8M_WAIT_DMA();                             // this group's K-tile t + 2 has landed (t + 4 may still be in flight)
        CVX_P8M_BARRIER();
        bi = bi + 2 >= NBUF ? bi + 2 - NBUF : bi + 2;
    }
#undef CVX_P8M_MM
    if (grp == 0) CVX_P8M_BARRIER();                    // pairs with group 1's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // drain the tail's dummy pieces before LDS is reused
    CVX_P8M_STAMP(2);
}

// PERM (round 4, the pair-writing epilogues EPI_QKV / EPI_GELU_SPLIT on whole 64-column wave tiles): 16-byte pair stores, see perm32
template <bool HAS_A2, int EPI, bool PERM = false>
__global__ __launch_bounds__(512, 2) void gemm_f16x3_p8m_kernel(
    const cvx_gemm_args p_in, const PreSplitA A, const f16* __restrict__ W, float acc_scale, SplitOut so,
    int tiles_m, int tiles_n, int ksplit, int k_per, float* __restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem_p8m[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, w4 = wid & 3, wr = w4 >> 1, wc = w4 & 1;
    // unit = (column tile, K slice); XCD x owns units x, x + 8, ...: all row tiles of a unit back to back on one XCD
    int unit, tm;
    if (tiles_n * ksplit >= 8) {
        const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
        unit = (q / tiles_m) * 8 + xcd; tm = q % tiles_m;
    } else {                     // fewer W panels than XCDs (to_pred: N = 80): row tiles round-robin over the XCDs, every XCD reads the small W
        unit = (int)blockIdx.x % (tiles_n * ksplit); tm = (int)blockIdx.x / (tiles_n * ksplit);
    }
    if (unit >= tiles_n * ksplit || tm >= tiles_m) return;
    const int tn = unit % tiles_n, ks = unit / tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    cvx_gemm_args p = p_in;
    acc_scale = total_acc_scale(acc_scale, so);
    int kt0 = 0, nk = p.K / 32;
    if (ksplit > 1) {            // this block: K-tiles of slice ks, plain (scaled) fp32 store of the partial tile
        kt0 = ks * (k_per / 32); nk = k_per / 32;
        p.C = partial + (int64_t)ks * p.M * p.N; p.ldc = p.N;
        p.bias = nullptr;
    }

    unsigned long long* tr = nullptr;
#ifdef CVX_DEV_FLAGS
    if ((so.dbg & 4) && so.trace && lane == 0 && w4 == 0) {
        tr = so.trace + ((int64_t)blockIdx.x * 2 + grp) * 8;
        unsigned long long rt; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt));
        tr[5] = rt; tr[0] = __builtin_readcyclecounter();
    }
#endif
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool v_block = false;
    if constexpr (EPI == EPI_QKV) v_block = n0 >= p.rope_cols;          // block-uniform: this tile holds V columns
    // everything the epilogue reads besides the accumulators is requested now (older than every DMA piece, so the counted waits
    // of the K loop cover it)
    const int row0 = m0 + wr * 64 + grp * 32, col0 = n0 + wc * 64;
    EpiPre<2> pre;
    epilogue_prefetch<EPI, 2, PERM>(p, so, row0, col0, lane, v_block, pre);
    if (v_block) tile_mainloop_m<HAS_A2, false, PERM>(p, A, W, smem_p8m, m0, n0, kt0, nk, lane, grp, w4, wr, wc, acc, tr);
    else tile_mainloop_m<HAS_A2, true, PERM>(p, A, W, smem_p8m, m0, n0, kt0, nk, lane, grp, w4, wr, wc, acc, tr);

    // ---- exchange: group 0 keeps rows 0-31 of every wave tile (blocks mi = 0, 1), group 1 rows 32-63 (mi = 2, 3)
    CVX_P8M_BARRIER();                                  // every wave is past its last fragment read and its last DMA piece
    f32x4* const X = reinterpret_cast<f32x4*>(smem_p8m);
    const int mine = ((grp * 4 + w4) * 8) * 64 + lane, theirs = (((1 - grp) * 4 + w4) * 8) * 64 + lane;
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) X[mine + (i * 4 + j) * 64] = acc[2 + i][j];
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) X[mine + (i * 4 + j) * 64] = acc[i][j];
    }
    CVX_P8M_WAIT_LDS();
    CVX_P8M_BARRIER();
    f32x4 e[2][4];
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) e[i][j] = acc[i][j] + X[theirs + (i * 4 + j) * 64];
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) e[i][j] = X[theirs + (i * 4 + j) * 64] + acc[2 + i][j];
    }
    CVX_P8M_STAMP(3);
    if (so.dbg & 1) return;                             // (dev: main loop only, timing)
    if (v_block) epilogue_vt<2, true, false, PERM>(p, e, row0, col0, lane, so, acc_scale, &pre);
    else epilogue_rows<EPI, 2, true, PERM ? 1 : 0>(p, e, row0, col0, lane, so, acc_scale, &pre);
#ifdef CVX_DEV_FLAGS
    if (tr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[4] = __builtin_readcyclecounter();
        unsigned long long rt; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt));
        tr[6] = rt;
    }
#endif
}

}  // namespace

namespace cvxg {

// ksplit > 1: k_per columns per K slice, partial = [ksplit][M][N] floats; the caller then runs splitk_reduce_kernel
bool launch_gemm_f16x3_p8m(const cvx_gemm_args& a, const PreSplitA& A, const f16* w_il, float acc_scale, const SplitOut& so,
                           int ksplit, float* partial, hipStream_t st)
{
    // N % 16 == 0: a trailing partial wave tile is computed on clamped W rows and not stored (to_pred: N = 80); RoPE / V^T epilogues
    // work on whole heads
    if (a.K % 32 != 0 || (A.hi2 && a.K1 % 32 != 0) || a.N % 16 != 0 || (a.N % 64 != 0 && (a.rope_cos || so.vt_hi))) return false;
    if ((int64_t)a.M * A.ld * 2 >= (int64_t)1 << 32 || (A.hi2 && (int64_t)a.M * A.ld2 * 2 >= (int64_t)1 << 32) ||
        (int64_t)a.N * a.ldw * 2 >= (int64_t)1 << 32) return false;
    SplitOut s2 = so;
    cvx_gemm_args a2 = a;
    int k_per = 0;
    if (ksplit > 1) {            // partial tiles: plain fp32 store, everything else happens in the reduction
        if (a.K % (32 * ksplit) != 0 || !partial || a.rope_cos || so.vt_hi) return false;
        k_per = a.K / ksplit;
        a2.act = CVX_ACT_NONE; a2.residual = nullptr; a2.rope_cos = nullptr; a2.rope_sin = nullptr;
        a2.C = partial; a2.ldc = a.N;
        s2.hi = nullptr; s2.lo = nullptr; s2.write_f32 = 1; s2.vt_hi = nullptr; s2.vt_lo = nullptr;
    }
    // 16-byte vector epilogue only: aligned pointers and strides, bias / RoPE tables included
    const bool vec = (!s2.write_f32 || (((uintptr_t)a2.C & 15) == 0 && (a2.ldc & 3) == 0)) &&
                     (!a2.residual || (((uintptr_t)a2.residual & 15) == 0 && (a2.ldr & 3) == 0)) &&
                     (!s2.hi || ((((uintptr_t)s2.hi | (uintptr_t)s2.lo) & 7) == 0 && (s2.ldc_h & 3) == 0)) &&
                     (!a2.bias || ((uintptr_t)a2.bias & 15) == 0) &&
                     (!a2.rope_cos || ((((uintptr_t)a2.rope_cos | (uintptr_t)a2.rope_sin) & 15) == 0 && a2.rope_cols % 128 == 0));
    if (!vec) return false;
    if (s2.vt_hi && !(a2.rope_cos && s2.hi && !s2.write_f32 && !a2.residual && a2.act == CVX_ACT_NONE)) return false;   // V^T only in QKV form
    const int tn = (a.N + 127) / 128, tm = (a.M + 127) / 128;
    const int units = tn * (ksplit > 1 ? ksplit : 1);
    const dim3 grid((unsigned)(units >= 8 ? ((units + 7) / 8) * 8 * tm : units * tm));
    int epi = ksplit > 1 ? (int)EPI_BIAS : classify_epilogue(a2, s2);
    if (epi == EPI_QKV && a2.bias) epi = EPI_GENERIC;
    if (epi == EPI_GENERIC && s2.vt_hi) return false;
    const int ks = ksplit > 1 ? ksplit : 1;
    // the pair-writing epilogues on whole wave tiles: permuted W tile rows, 16-byte pair stores (CVX_P8M_PERM=0 in a dev build: off)
    const bool perm = CVX_P8M_PERM && ksplit <= 1 && !A.hi2 && (epi == EPI_QKV || epi == EPI_GELU_SPLIT) && a.N % 64 == 0 && s2.hi &&
                      (((uintptr_t)s2.hi | (uintptr_t)s2.lo) & 15) == 0 && (s2.ldc_h & 7) == 0;
#define CVX_P8M_LAUNCH_P(E)                                                                                              \
    do {                                                                                                                \
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_f16x3_p8m_kernel<false, E, true>), MLDS_B);            \
        hipLaunchKernelGGL((gemm_f16x3_p8m_kernel<false, E, true>), grid, dim3(512), MLDS_B, st, a2, A, w_il, acc_scale, s2, tm, tn, ks, k_per, partial); \
    } while (0)
    if (perm) {
        if (epi == EPI_QKV) CVX_P8M_LAUNCH_P(EPI_QKV); else CVX_P8M_LAUNCH_P(EPI_GELU_SPLIT);
        return true;
    }
#undef CVX_P8M_LAUNCH_P
#define CVX_P8M_LAUNCH(A2, E)                                                                                           \
    do {                                                                                                                \
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_f16x3_p8m_kernel<A2, E>), MLDS_B);                    \
        hipLaunchKernelGGL((gemm_f16x3_p8m_kernel<A2, E>), grid, dim3(512), MLDS_B, st, a2, A, w_il, acc_scale, s2, tm, tn, ks, k_per, partial); \
    } while (0)
    if (A.hi2) {
        if (epi == EPI_BIAS) CVX_P8M_LAUNCH(true, EPI_BIAS); else CVX_P8M_LAUNCH(true, EPI_GENERIC);
    } else {
        switch (epi) {
            case EPI_QKV: CVX_P8M_LAUNCH(false, EPI_QKV); break;
            case EPI_RES: CVX_P8M_LAUNCH(false, EPI_RES); break;
            case EPI_GELU_SPLIT: CVX_P8M_LAUNCH(false, EPI_GELU_SPLIT); break;
            case EPI_BIAS: CVX_P8M_LAUNCH(false, EPI_BIAS); break;
            default: CVX_P8M_LAUNCH(false, EPI_GENERIC); break;
        }
    }
#undef CVX_P8M_LAUNCH
    return true;
}

}  // namespace cvxg
