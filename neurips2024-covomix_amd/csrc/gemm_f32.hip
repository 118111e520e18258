// fp32 GEMM  C[M,N] = epi([A|A2][M,K] * W[N,K]^T)  on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Every nn.Linear of the CoVoMix vector field (reference acoustic.py:225-246, :306-310,
// :361-365, :200, :503-516) goes through this kernel; 86 % of the path's FLOPs.
//
// Design (MI355X): 256 threads = 4 waves (2x2), block tile (TM*64) x 128, K-step 32.
//   * both operands are K-contiguous ("NT"), staged global -> VGPR -> LDS with 16-byte
//     loads, LDS rows padded to 36 floats so the ds_read_b128 fragment reads are
//     bank-conflict free;
//   * one ds_read_b128 feeds FOUR MFMAs: lanes 0-31 hold k = 8q..8q+3, lanes 32-63 hold
//     k = 8q+4..8q+7 of their row, and MFMA t contracts the (8q+t, 8q+4+t) pair - the
//     k order inside a tile is free as long as A and B use the same one;
//   * two LDS buffers; the K loop is ONE branch-free basic block per step (out-of-range
//     rows are clamped, not predicated: their products land in accumulator rows/columns the
//     epilogue never stores), with the next tile's global loads issued first, its LDS
//     writes placed between the 3rd and 4th MFMA group, and one barrier per step - so the
//     matrix pipe always has queued work while loads, address math and LDS writes retire;
//   * XCD-aware block->tile map: an XCD owns whole M row-panels, so each A panel is
//     fetched into exactly one L2.
// Epilogue fuses bias, GELU/SiLU, half-split RoPE (a wave owns one whole 64-wide head, so
// the (j, j+32) partner is the same accumulator register of the neighbouring MFMA tile)
// and the residual add.  K % 32 != 0 (only to_embed's 80 x-columns) takes the predicated
// generic kernel.
#include "gemm_common.h"
#include <stdlib.h>

namespace {

using namespace cvxg;
constexpr int LDS_LD = BK + 4;   // padded row (floats)

template <int TM>
__device__ __forceinline__ void mfma_group(const float* a, const float* b, int q, f32x16 (&acc)[TM][2])
{
    f32x4 af[TM], bf[2];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(a + mi * 32 * LDS_LD + q * 8);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(b + ni * 32 * LDS_LD + q * 8);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][t], bf[ni][t], acc[mi][ni], 0, 0, 0);
}

// ------------------------------------------------------------------ fast path: K % 32 == 0
template <int TM>   // TM = MFMA tiles per wave along M (block M = TM*64)
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const cvx_gemm_args p, int tiles_m, int tiles_n, int map_mode)
{
    constexpr int BM = TM * 64;
    constexpr int NA = TM * 2;                // 16-byte A loads per thread per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tile_m, tile_n;
    tile_of_block(tiles_m, tiles_n, map_mode, tile_m, tile_n);
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int srow = tid >> 3;            // 0..31
    const int skc = (tid & 7) * 4;        // 0..28

    // per-thread streaming pointers (rows clamped into range: no predication in the K loop)
    const float* pa[NA];
    int64_t a_jump[NA];                   // extra element offset applied when the stream switches A -> A2
    const float* pw[4];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = min(m0 + srow + 32 * i, p.M - 1);
        pa[i] = p.A + (int64_t)row * p.lda + skc;
        a_jump[i] = p.A2 ? (p.A2 + (int64_t)row * p.lda2 + skc) - (pa[i] + p.K1) : 0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = min(n0 + srow + 32 * i, p.N - 1);
        pw[i] = p.W + (int64_t)row * p.ldw + skc;
    }
    const int switch_tile = p.A2 ? p.K1 / BK : -1;   // tile index whose loads come first from A2

    f32x4 ra[NA], rb[4];
    f32x16 acc[TM][2];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int a_off = (wm * TM * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int b_off = (wn * 64 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int st_off = srow * LDS_LD + skc;

    const int nk = p.K / BK;
    // prologue: tile 0 -> LDS buffer 0
    {
        const bool sw = (0 == switch_tile);
#pragma unroll
        for (int i = 0; i < NA; ++i) { if (sw) pa[i] += a_jump[i]; ra[i] = gload4(pa[i]); pa[i] += BK; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { rb[i] = gload4(pw[i]); pw[i] += BK; }
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(As + st_off + 32 * i * LDS_LD) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Bs + st_off + 32 * i * LDS_LD) = rb[i];
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float* a = As + cur * BM * LDS_LD + a_off;
        const float* b = Bs + cur * BN * LDS_LD + b_off;
        // next tile's global loads first (clamped to the last tile on the final step: harmless re-read)
        const bool has_next = kt + 1 < nk;
        const int adv = has_next ? BK : 0;
        const bool sw = (kt + 1 == switch_tile);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float* src = pa[i] + (sw ? a_jump[i] : 0) - (has_next ? 0 : BK);
            ra[i] = gload4(src);
            pa[i] = src + adv;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* src = pw[i] - (has_next ? 0 : BK);
            rb[i] = gload4(src);
            pw[i] = src + adv;
        }
        mfma_group<TM>(a, b, 0, acc);
        mfma_group<TM>(a, b, 1, acc);
        mfma_group<TM>(a, b, 2, acc);
        // the other buffer was last read before the previous barrier: safe to refill it now
        float* an = As + (cur ^ 1) * BM * LDS_LD + st_off;
        float* bn = Bs + (cur ^ 1) * BN * LDS_LD + st_off;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(an + 32 * i * LDS_LD) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(bn + 32 * i * LDS_LD) = rb[i];
        mfma_group<TM>(a, b, 3, acc);
        if constexpr (TM == 2) {
            // Issue order of this step (one scheduling region): memory instructions are spread ONE per MFMA
            // instead of in bursts.  Measured on the probe (tools/archive/mfma_probe.hip): a burst of 8 global loads
            // costs 107 vs 135 TFLOP/s when the operands stream from HBM.
            //   q0: 4 frag reads | 8 x (MFMA, global load) | 4 x (2 MFMA, q1 frag read)
            //   q1: 4 x (4 MFMA, q2 frag read)
            //   q2: 8 x (MFMA, LDS write of the next tile) | 4 x (2 MFMA, q3 frag read)
            //   q3: 16 MFMA
#define CVX_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
            CVX_SGB(0x100, 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) { CVX_SGB(0x008, 1); CVX_SGB(0x020, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { CVX_SGB(0x008, 2); CVX_SGB(0x100, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { CVX_SGB(0x008, 4); CVX_SGB(0x100, 1); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { CVX_SGB(0x008, 1); CVX_SGB(0x200, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { CVX_SGB(0x008, 2); CVX_SGB(0x100, 1); }
            CVX_SGB(0x008, 16);
#undef CVX_SGB
        }
        __syncthreads();
    }
    gemm_epilogue<TM>(p, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------ LDS-DMA path: K % 32 == 0, 128 x 128 tile
// Same tiling, but the tiles go global -> LDS directly (global_load_lds_dwordx4): no staging VGPRs and no
// ds_write pass (the probe prices 8 ds_write_b128 per K-step at 8 % of the MFMA rate).  An LDS-DMA wave
// instruction writes 1 KiB contiguously (wave-uniform base + lane*16 B = 8 rows x 128 B), so rows cannot be
// padded; bank conflicts are avoided with an XOR swizzle applied on the per-lane SOURCE address and again on
// the fragment reads:  16-byte chunk c of row r lives at chunk  c ^ ((r >> 1) & 7)  (conflict-free for the
// ds_read_b128 lane groups, whose 16 rows then cover all 16 (row parity, chunk) slots exactly once).
__global__ __launch_bounds__(256, 2) void gemm_f32_glds_kernel(const cvx_gemm_args p, int tiles_m, int tiles_n, int map_mode)
{
    constexpr int TM = 2, BM = 128;
    constexpr int TILE_F = 128 * BK;               // floats per operand tile (unpadded)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [2][128][32]
    float* Bs = smem + 2 * TILE_F;                 // [2][128][32]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tile_m, tile_n;
    tile_of_block(tiles_m, tiles_n, map_mode, tile_m, tile_n);
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // DMA sources: wave `wid` fills tile rows [32*wid, 32*wid+32) of A and of W, 8 rows per instruction
    const float* pa[4];
    int64_t a_jump[4];
    const float* pw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 32 * wid + 8 * j + (lane >> 3);          // row inside the tile
        const int c = (lane & 7) ^ ((r >> 1) & 7);             // source chunk that lands in LDS chunk (lane & 7)
        const int ra = min(m0 + r, p.M - 1), rw = min(n0 + r, p.N - 1);
        pa[j] = p.A + (int64_t)ra * p.lda + 4 * c;
        a_jump[j] = p.A2 ? (p.A2 + (int64_t)ra * p.lda2 + 4 * c) - (pa[j] + p.K1) : 0;
        pw[j] = p.W + (int64_t)rw * p.ldw + 4 * c;
    }
    const int switch_tile = p.A2 ? p.K1 / BK : -1;
    const int dma_off = (32 * wid) * BK;                        // wave-uniform float offset inside a tile

    f32x16 acc[TM][2];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment read offsets: row (lane & 31) of the wave's 32-row sub-tile, chunk (2q + half) ^ swizzle
    const int i31 = lane & 31, half = lane >> 5, swz = (i31 >> 1) & 7;
    int qoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = i31 * BK + 4 * ((2 * q + half) ^ swz);
    const int a_row0 = wm * 64 * BK, b_row0 = wn * 64 * BK;

    const int nk = p.K / BK;
    {   // prologue: tile 0 -> buffer 0
        const bool sw = (0 == switch_tile);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (sw) pa[j] += a_jump[j];
            glds16(pa[j], As + dma_off + 8 * j * BK);
            glds16(pw[j], Bs + dma_off + 8 * j * BK);
            pa[j] += BK; pw[j] += BK;
        }
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool has_next = kt + 1 < nk;
        const int adv = has_next ? BK : 0;
        const bool sw = (kt + 1 == switch_tile);
        float* an = As + (cur ^ 1) * TILE_F + dma_off;
        float* bn = Bs + (cur ^ 1) * TILE_F + dma_off;
        // the NEXT tile's DMA first: it has the whole step (64 MFMAs) to land before the barrier's vmcnt(0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* sa = pa[j] + (sw ? a_jump[j] : 0) - (has_next ? 0 : BK);
            const float* sb = pw[j] - (has_next ? 0 : BK);
            glds16(sa, an + 8 * j * BK);
            glds16(sb, bn + 8 * j * BK);
            pa[j] = sa + adv; pw[j] = sb + adv;
        }
        const float* a = As + cur * TILE_F + a_row0;
        const float* b = Bs + cur * TILE_F + b_row0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 af[TM], bf[2];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(a + mi * 32 * BK + qoff[q]);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(b + ni * 32 * BK + qoff[q]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][t], bf[ni][t], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();          // carries the vmcnt(0) that retires this step's LDS-DMA
    }
    gemm_epilogue<TM>(p, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------ generic path: any K % 4 == 0 (predicated loads)
template <int TM>
__global__ __launch_bounds__(256, 2) void gemm_f32_generic_kernel(const cvx_gemm_args p, int tiles_m, int tiles_n, int map_mode)
{
    constexpr int BM = TM * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tile_m, tile_n;
    tile_of_block(tiles_m, tiles_n, map_mode, tile_m, tile_n);
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int srow = tid >> 3, skc = (tid & 7) * 4;

    f32x4 ra[TM * 2], rb[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tiles = [&](int k0) {
        const float* Ap = p.A;
        int64_t lda = p.lda;
        int kk = k0, klim = p.K;
        if (p.A2 != nullptr) {
            if (k0 >= p.K1) { Ap = p.A2; lda = p.lda2; kk = k0 - p.K1; klim = p.K - p.K1; }
            else            { klim = p.K1; }
        }
#pragma unroll
        for (int i = 0; i < TM * 2; ++i) {
            const int row = m0 + srow + 32 * i;
            ra[i] = (row < p.M && kk + skc < klim)
                        ? *reinterpret_cast<const f32x4*>(Ap + (int64_t)row * lda + kk + skc) : zero4;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = n0 + srow + 32 * i;
            rb[i] = (row < p.N && k0 + skc < p.K)
                        ? *reinterpret_cast<const f32x4*>(p.W + (int64_t)row * p.ldw + k0 + skc) : zero4;
        }
    };
    auto store_tiles = [&](int buf) {
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < TM * 2; ++i) *reinterpret_cast<f32x4*>(a + (srow + 32 * i) * LDS_LD + skc) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(b + (srow + 32 * i) * LDS_LD + skc) = rb[i];
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int a_off = (wm * TM * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int b_off = (wn * 64 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int nk = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const float* a = As + cur * BM * LDS_LD + a_off;
        const float* b = Bs + cur * BN * LDS_LD + b_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) mfma_group<TM>(a, b, q, acc);
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }
    gemm_epilogue<TM>(p, acc, m0, n0, wm, wn, lane);
}

template <int TM>
int launch_gemm(const cvx_gemm_args& a, hipStream_t st)
{
    constexpr int BM = TM * 64;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    const bool fast = (a.K % BK == 0);
    cvx_allow_dynamic_lds(fast ? reinterpret_cast<const void*>(gemm_f32_kernel<TM>)
                               : reinterpret_cast<const void*>(gemm_f32_generic_kernel<TM>), (int)lds);
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int map_mode = 1;             // XCD-aware block -> tile map (gemm_common.h)
    const int grid_m = map_mode == 1 ? ((tiles_m + 7) / 8) * 8 : tiles_m;
    dim3 grid((unsigned)(grid_m * tiles_n));
    if constexpr (TM == 2) {
        if (fast) {
            const size_t lds_dma = (size_t)4 * 128 * BK * sizeof(float);
            cvx_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_f32_glds_kernel), (int)lds_dma);
            hipLaunchKernelGGL(gemm_f32_glds_kernel, grid, dim3(256), lds_dma, st, a, tiles_m, tiles_n, map_mode);
            CVX_CHECK_LAUNCH("cvx_gemm_bias_act_f32");
            return CVX_OK;
        }
    }
    if (fast) hipLaunchKernelGGL(gemm_f32_kernel<TM>, grid, dim3(256), lds, st, a, tiles_m, tiles_n, map_mode);
    else      hipLaunchKernelGGL(gemm_f32_generic_kernel<TM>, grid, dim3(256), lds, st, a, tiles_m, tiles_n, map_mode);
    CVX_CHECK_LAUNCH("cvx_gemm_bias_act_f32");
    return CVX_OK;
}

}  // namespace

int cvxg::validate_gemm_args(const cvx_gemm_args* a)
{
    CVX_REQUIRE(a != nullptr, "gemm: null args");
    CVX_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    CVX_REQUIRE(a->A && a->W && a->C, "gemm: null operand");
    CVX_REQUIRE(a->K % 4 == 0 && a->lda % 4 == 0 && a->ldw % 4 == 0, "gemm: K/lda/ldw must be multiples of 4");
    CVX_REQUIRE((((uintptr_t)a->A | (uintptr_t)a->W) & 15) == 0, "gemm: A/W must be 16-byte aligned");
    if (a->A2) {
        CVX_REQUIRE(a->K1 > 0 && a->K1 < a->K && a->K1 % BK == 0 && a->lda2 % 4 == 0 &&
                    (((uintptr_t)a->A2) & 15) == 0, "gemm: bad split-K operand (K1=%d)", a->K1);
    }
    if (a->rope_cos) {
        CVX_REQUIRE(a->rope_sin && a->rope_T > 0 && a->rope_cols % 64 == 0 && a->rope_cols <= a->N,
                    "gemm: bad RoPE epilogue arguments");
    }
    CVX_REQUIRE(a->act >= CVX_ACT_NONE && a->act <= CVX_ACT_SILU, "gemm: unsupported activation %d", a->act);
    return CVX_OK;
}

extern "C" int cvx_gemm_bias_act_f32(const cvx_gemm_args* a, cvx_stream_t s)
{
    const int rc = cvxg::validate_gemm_args(a);
    if (rc != CVX_OK) return rc;
    if (a->M == 0) return CVX_OK;
    hipStream_t st = cvx_hip_stream(s);
    // Small-M problems (time tables, short utterances) use the 64-row tile to fill more CUs.
    const long blocks128 = (long)((a->M + 127) / 128) * ((a->N + BN - 1) / BN);
    if (a->M <= 64 || blocks128 < 256) return launch_gemm<1>(*a, st);
    return launch_gemm<2>(*a, st);
}
