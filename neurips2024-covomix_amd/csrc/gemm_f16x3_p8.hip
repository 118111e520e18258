// Split-precision GEMM, large-problem kernel: 256 x 256 block tile, eight waves, EIGHT-PHASE PING-PONG main loop.
//
// Same contract as gemm_f16x3.hip (C = epi([A|A2] * W^T) with three v_mfma_f32_32x32x16_f16 products per tile on
// (fp16 hi, fp16 lo) operand pairs; every nn.Linear of reference acoustic.py:225-246, :306-310) for INTERLEAVED
// operands: one K-tile (32 k) of a row of A or W is one 128-byte line [hi 32 | lo 32].
//
// What is different from the two-stage kernel (one barrier + one vmcnt(0) per K-tile, all eight waves in phase):
//   * the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; waves w and w+4 share a SIMD) run
//     ONE BARRIER INTERVAL APART: while one group issues its 12 MFMAs of a phase the other reads its next
//     fragments from LDS and issues its share of the LDS-DMA - the SIMD's matrix pipe always has a wave feeding it;
//   * a K-tile is four phases, one 64 x 32 quadrant of the wave's 128 x 64 output each, in the order
//     (m0,n0) (m0,n1) (m1,n1) (m1,n0) so that every phase needs at most one new operand block (A m0 + B n0, B n1,
//     A m1, nothing) and the operand QUARTERS of a K-tile are consumed progressively;
//   * the DMA stream runs SIX quarter-tiles (96 KiB) ahead in a two-buffer ring: a quarter (16 KiB: the rows one
//     phase block needs, for all waves) is re-filled two intervals after its last reader, i.e. up to 9 intervals
//     before its first one, and is retired by a COUNTED s_waitcnt vmcnt(8) (never 0) followed by a barrier;
//   * the DMA is issued from inline asm (scalar base + 32-bit lane offset, M0 = LDS slot): hipcc puts a vmcnt(0) in
//     front of every ds_read that follows a global_load_lds it can see.
// LDS: 2 buffers x (A tile [256][128 B] | W tile [256][128 B]) = 128 KiB + 8 KiB dump area for the tail's dummy DMA.
// Swizzle: 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7), applied on the DMA source address and on the
// ds_read_b128 fragment address (conflict-free, as in the two-stage kernel).
#include "gemm_common.h"

namespace {

using namespace cvxg;
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TILE_B = 256 * 128;                  // bytes per operand tile
constexpr int BUF_B = 2 * TILE_B;                  // A | W
constexpr int DUMP_B = 2 * BUF_B;                  // dump area offset (8 KiB)
constexpr int LDS_B = DUMP_B + 8 * 1024;

// two LDS-DMA pieces (1 KiB each: 8 rows x 128 B) of one quarter-tile: LDS destinations m0a / m0b (wave-uniform byte
// addresses; the hardware adds lane * 16), sources = scalar base + per-lane 32-bit byte offsets
__device__ __forceinline__ void dma2(uint32_t voff0, uint32_t voff1, uint32_t m0a, uint32_t m0b, const void* sbase)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5\n\t"
                 "s_mov_b32 m0, %4\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff0), "v"(voff1), "s"(m0a), "s"(m0b), "s"(sbase)
                 : "memory");
}

// raw barrier as an opaque statement with a memory clobber: the builtin does not order LDS reads for the compiler
#define CVX_P8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define CVX_P8_WAIT_DMA() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define CVX_P8_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <bool HAS_A2, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_f16x3_p8_kernel(
    const cvx_gemm_args p, const PreSplitA A, const f16* __restrict__ W, float acc_scale, SplitOut so,
    int tiles_m, int tiles_n, int map_mode)
{
    extern __shared__ __attribute__((aligned(16))) char smem_p8[];
    const unsigned long long ts0 = so.trace ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long rt0 = so.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;                      // wave group (M half), N quarter
    int tile_m, tile_n;
    tile_of_block(tiles_m, tiles_n, map_mode, tile_m, tile_n);
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_p8;

    // ---- DMA plan.  Quarter kinds: A0/A1 = the 64-row block m0/m1 of both wave groups, B0/B1 = the 32-row block
    // n0/n1 of all four N quarters.  A quarter is 128 rows = 16 pieces of 8 rows; wave `wid` moves pieces 2*wid, 2*wid+1.
    uint32_t offA[2][2], offA2[2][2], offW[2][2];                // per-lane source byte offsets
    uint32_t dstA[2][2], dstW[2][2];                             // LDS byte offsets inside a buffer (wave-uniform)
    const int64_t ldaB = A.ld * 2, lda2B = A.ld2 * 2, ldwB = p.ldw * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pr0 = 8 * (2 * wid + j);
            const int ra0 = (pr0 >> 6) * 128 + h * 64 + (pr0 & 63);          // first tile row of the piece
            const int rb0 = (pr0 >> 5) * 64 + h * 32 + (pr0 & 31);
            const int ra = ra0 + (lane >> 3), rb = rb0 + (lane >> 3);
            const uint32_t ca = (uint32_t)(((lane & 7) ^ ((ra >> 1) & 7)) * 16);
            const uint32_t cb = (uint32_t)(((lane & 7) ^ ((rb >> 1) & 7)) * 16);
            const int64_t ga = min(m0 + ra, p.M - 1), gb = min(n0 + rb, p.N - 1);
            offA[h][j] = (uint32_t)(ga * ldaB) + ca;
            offA2[h][j] = HAS_A2 ? (uint32_t)(ga * lda2B) + ca : 0u;
            offW[h][j] = (uint32_t)(gb * ldwB) + cb;
            dstA[h][j] = (uint32_t)(ra0 * 128);
            dstW[h][j] = (uint32_t)(TILE_B + rb0 * 128);
        }
    const int nk = p.K / 32;
    const int t_sw = HAS_A2 ? p.K1 / 32 : 0x7fffffff;
    const char* const a1base = reinterpret_cast<const char*>(A.hi);
    const char* const a2base = reinterpret_cast<const char*>(A.hi2);
    const char* const wbase = reinterpret_cast<const char*>(W);
    const uint32_t dump = lds0 + DUMP_B + (uint32_t)wid * 1024u;

    // quarter A_h of K-tile tt -> buffer tt & 1 (past the end: a dummy 16-byte re-read into the dump area keeps the
    // vmcnt arithmetic uniform)
    auto issue_A = [&](int h, int tt) {
        const bool live = tt < nk;
        const uint32_t b = lds0 + (uint32_t)(tt & 1) * BUF_B;
        const char* base = a1base + (int64_t)tt * 128;
        uint32_t v0 = offA[h][0], v1 = offA[h][1];
        if constexpr (HAS_A2) {
            if (tt >= t_sw) { base = a2base + (int64_t)(tt - t_sw) * 128; v0 = offA2[h][0]; v1 = offA2[h][1]; }
        }
        if (!live) { base = wbase; v0 = 0u; v1 = 0u; }
        dma2(v0, v1, live ? b + dstA[h][0] : dump, live ? b + dstA[h][1] : dump, base);
    };
    auto issue_W = [&](int h, int tt) {
        const bool live = tt < nk;
        const uint32_t b = lds0 + (uint32_t)(tt & 1) * BUF_B;
        const char* base = live ? wbase + (int64_t)tt * 128 : wbase;
        dma2(live ? offW[h][0] : 0u, live ? offW[h][1] : 0u, live ? b + dstW[h][0] : dump, live ? b + dstW[h][1] : dump, base);
    };

    // ---- fragment read addresses (bytes inside a buffer): row i31 of a 32-row MFMA tile, chunk (2s+g | 4+2s+g) ^ swizzle
    const int i31 = lane & 31, g = lane >> 5, sw8 = (i31 >> 1) & 7;
    int aoh[2], aol[2], boh[2], bol[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        aoh[s] = (wr * 128 + i31) * 128 + 16 * ((2 * s + g) ^ sw8);
        aol[s] = (wr * 128 + i31) * 128 + 16 * ((4 + 2 * s + g) ^ sw8);
        boh[s] = TILE_B + (wc * 64 + i31) * 128 + 16 * ((2 * s + g) ^ sw8);
        bol[s] = TILE_B + (wc * 64 + i31) * 128 + 16 * ((4 + 2 * s + g) ^ sw8);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // ---- prologue: quarters #0..#5 = A0(0) B0(0) B1(0) A1(0) A0(1) B0(1); the loop continues B1(1) A1(1) A0(2) B0(2) ...
    issue_A(0, 0); issue_W(0, 0); issue_W(1, 0); issue_A(1, 0); issue_A(0, 1); issue_W(0, 1);
    CVX_P8_WAIT_DMA();                                  // 4 quarters may stay in flight: A0(0), B0(0) have landed
    CVX_P8_BARRIER();
    if (wr == 1) CVX_P8_BARRIER();          // group 1 runs one interval behind group 0
    const unsigned long long ts1 = so.trace ? __builtin_readcyclecounter() : 0ull;

    f16x8 fah[2][2], fal[2][2];                         // A fragments of the current M half: [tile][k slice]
    f16x8 fbh[2][2], fbl[2][2];                         // B fragments: [n half][k slice]

#define CVX_P8_READ_A(buf, mh)                                                                                   \
    _Pragma("unroll") for (int mi2 = 0; mi2 < 2; ++mi2) _Pragma("unroll") for (int s = 0; s < 2; ++s) {          \
        fah[mi2][s] = *reinterpret_cast<const f16x8*>(smem_p8 + (buf) * BUF_B + ((mh) * 2 + mi2) * 32 * 128 + aoh[s]); \
        fal[mi2][s] = *reinterpret_cast<const f16x8*>(smem_p8 + (buf) * BUF_B + ((mh) * 2 + mi2) * 32 * 128 + aol[s]); \
    }
#define CVX_P8_READ_B(buf, nh)                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                              \
        fbh[nh][s] = *reinterpret_cast<const f16x8*>(smem_p8 + (buf) * BUF_B + (nh) * 32 * 128 + boh[s]);        \
        fbl[nh][s] = *reinterpret_cast<const f16x8*>(smem_p8 + (buf) * BUF_B + (nh) * 32 * 128 + bol[s]);        \
    }
    // one phase's matrix work: quadrant (mh, nh), two k slices x three terms x two tiles = 12 MFMAs; consecutive MFMAs
    // alternate between the two accumulators
#define CVX_P8_MFMA(mh, nh)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                              \
        _Pragma("unroll") for (int mi2 = 0; mi2 < 2; ++mi2)                                                      \
            acc[(mh) * 2 + mi2][nh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[mi2][s], fbh[nh][s], acc[(mh) * 2 + mi2][nh], 0, 0, 0); \
        _Pragma("unroll") for (int mi2 = 0; mi2 < 2; ++mi2)                                                      \
            acc[(mh) * 2 + mi2][nh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi2][s], fbl[nh][s], acc[(mh) * 2 + mi2][nh], 0, 0, 0); \
        _Pragma("unroll") for (int mi2 = 0; mi2 < 2; ++mi2)                                                      \
            acc[(mh) * 2 + mi2][nh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi2][s], fbh[nh][s], acc[(mh) * 2 + mi2][nh], 0, 0, 0); \
    }                                                                                                            \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    CVX_P8_BARRIER();
    // end of a load segment: my share of the quarter the NEXT phase reads has landed; rendezvous; my LDS reads returned
#define CVX_P8_SYNC()                                                                                            \
    CVX_P8_WAIT_DMA();                                                                                           \
    CVX_P8_BARRIER();                                                                                \
    CVX_P8_WAIT_LDS();

#define CVX_P8_KTILE(buf, t)                                                                                     \
    {   /* phase 0: (m0, n0) */                                                                                  \
        CVX_P8_READ_B(buf, 0) CVX_P8_READ_A(buf, 0)                                                              \
        issue_W(1, (t) + 1);                                                                                     \
        CVX_P8_SYNC() CVX_P8_MFMA(0, 0)                                                                          \
        /* phase 1: (m0, n1) */                                                                                  \
        CVX_P8_READ_B(buf, 1)                                                                                    \
        issue_A(1, (t) + 1);                                                                                     \
        CVX_P8_SYNC() CVX_P8_MFMA(0, 1)                                                                          \
        /* phase 2: (m1, n1) */                                                                                  \
        CVX_P8_READ_A(buf, 1)                                                                                    \
        issue_A(0, (t) + 2);                                                                                     \
        CVX_P8_SYNC() CVX_P8_MFMA(1, 1)                                                                          \
        /* phase 3: (m1, n0) - B n0 is still in registers */                                                     \
        issue_W(0, (t) + 2);                                                                                     \
        CVX_P8_SYNC() CVX_P8_MFMA(1, 0)                                                                          \
    }

    int t = 0;
    for (; t + 1 < nk; t += 2) {
        CVX_P8_KTILE(0, t)
        CVX_P8_KTILE(1, t + 1)
    }
    if (t < nk) CVX_P8_KTILE(0, t)
#undef CVX_P8_KTILE
#undef CVX_P8_SYNC
#undef CVX_P8_MFMA
#undef CVX_P8_READ_A
#undef CVX_P8_READ_B

    const unsigned long long ts2 = so.trace ? __builtin_readcyclecounter() : 0ull;
    if (wr == 0) CVX_P8_BARRIER();                      // pairs with group 1's last barrier (both epilogues then run together)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the tail's dummy pieces before LDS is released
    acc_scale = total_acc_scale(acc_scale, so);
    if (acc_scale != 1.0f) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= acc_scale;
    }
    if (so.dbg & 1) { if (acc[0][0][0] == 12345.678f) p.C[0] = 1.f; return; }      // timing experiment: main loop only
    gemm_epilogue<4, EPI>(p, acc, m0, n0, wr, wc, lane, so);
    if (so.trace) {       // dev: cycle stamps of wave `wid` of this block: start, main loop start, main loop end, stores issued, stores done
        const unsigned long long ts3 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts4 = __builtin_readcyclecounter();
        if (lane == 0) {
            unsigned long long* t = so.trace + ((size_t)blockIdx.x * 8 + wid) * 8;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            hw = (hw & 0xffffff) | ((xcc & 0xf) << 24);
            t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = ts3; t[4] = ts4; t[5] = __builtin_amdgcn_s_memrealtime(); t[6] = rt0; t[7] = hw;
        }
    }
}

}  // namespace

namespace cvxg {

bool launch_gemm_f16x3_p8(const cvx_gemm_args& a, const PreSplitA& A, const f16* w_il, float acc_scale, const SplitOut& so,
                          int map_mode, hipStream_t st)
{
    // 32-bit lane offsets: every operand must span < 4 GiB; interleaved layouts only
    const int64_t k1 = A.hi2 ? a.K1 : a.K;
    if (a.K % 32 != 0 || (A.hi2 && a.K1 % 32 != 0)) return false;
    if ((int64_t)a.M * A.ld * 2 >= (int64_t)1 << 32 || (A.hi2 && (int64_t)a.M * A.ld2 * 2 >= (int64_t)1 << 32) ||
        (int64_t)a.N * a.ldw * 2 >= (int64_t)1 << 32) return false;
    (void)k1;
    const int tn = (a.N + 255) / 256, tm = (a.M + 255) / 256;
    const int gm = map_mode == 1 ? ((tm + 7) / 8) * 8 : tm;
    const dim3 grid((unsigned)(gm * tn));
    const int epi = (so.dbg & 2) ? EPI_GENERIC : classify_epilogue(a, so);
#define CVX_P8_LAUNCH(A2, E)                                                                                            \
    do {                                                                                                                \
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_f16x3_p8_kernel<A2, E>), LDS_B);                      \
        hipLaunchKernelGGL((gemm_f16x3_p8_kernel<A2, E>), grid, dim3(512), LDS_B, st, a, A, w_il, acc_scale, so, tm, tn, map_mode); \
    } while (0)
    if (A.hi2) {
        if (epi == EPI_BIAS) CVX_P8_LAUNCH(true, EPI_BIAS); else CVX_P8_LAUNCH(true, EPI_GENERIC);
    } else {
        switch (epi) {
            case EPI_QKV: CVX_P8_LAUNCH(false, EPI_QKV); break;
            case EPI_RES: CVX_P8_LAUNCH(false, EPI_RES); break;
            case EPI_GELU_SPLIT: CVX_P8_LAUNCH(false, EPI_GELU_SPLIT); break;
            case EPI_BIAS: CVX_P8_LAUNCH(false, EPI_BIAS); break;
            default: CVX_P8_LAUNCH(false, EPI_GENERIC); break;
        }
    }
#undef CVX_P8_LAUNCH
    return true;
}

int classify_epilogue(const cvx_gemm_args& a, const SplitOut& so)
{
    // the specialised epilogues only have the 16-byte vector path: whole 64-column wave tiles, aligned pointers and strides
    const bool vec = (a.N % 64 == 0) &&
                     (!so.write_f32 || (((uintptr_t)a.C & 15) == 0 && (a.ldc & 3) == 0)) &&
                     (!a.residual || (((uintptr_t)a.residual & 15) == 0 && (a.ldr & 3) == 0)) &&
                     (!so.hi || ((((uintptr_t)so.hi | (uintptr_t)so.lo) & 7) == 0 && (so.ldc_h & 3) == 0));
    if (!vec) return EPI_GENERIC;
    const bool rope = a.rope_cos != nullptr;
    if (rope && so.vt_hi && so.hi && !so.write_f32 && !a.bias && !a.residual && a.act == CVX_ACT_NONE) return EPI_QKV;
    if (rope || so.vt_hi) return EPI_GENERIC;
    if (a.residual && so.write_f32 && a.act == CVX_ACT_NONE) return EPI_RES;
    if (a.bias && a.act == CVX_ACT_GELU && !a.residual && so.hi && !so.write_f32) return EPI_GELU_SPLIT;
    if (a.act == CVX_ACT_NONE && !a.residual && so.write_f32 && !so.hi) return EPI_BIAS;
    return EPI_GENERIC;
}

}  // namespace cvxg
