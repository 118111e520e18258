// Split-precision GEMM on the 2.5 PFLOP/s matrix pipe:  C = epi([A|A2] * W^T) with every fp32 operand
// represented as an (fp16 hi, fp16 lo) pair and three v_mfma_f32_32x32x16_f16 products per tile
//      a*w  ~=  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi          (fp32 accumulate)
// hi = fp16(x), lo = fp16(x - hi): 22 significand bits survive, so the dropped a_lo*w_lo term and the operand
// truncation are ~2^-22 relative - the same class as fp32 rounding (measured in tests/test_kernels_gpu.py),
// unlike a plain bf16/fp16 cast (2.9e-3 end-to-end, SURVEY.md section 7 hard part 1).  Three 32-cycle MFMAs
// replace eight 64-cycle fp32 MFMAs for the same 32x32x16 block: 5.3x less matrix-pipe time.
//
// Same role as gemm_f32.hip (every nn.Linear of reference acoustic.py:225-246, :306-310, :516) and the same
// 128x128x32 block tile / 4 waves / fused epilogue.  Operand paths:
//   W : split ONCE at load time (cvx_split_f16) into two fp16 matrices; tiles arrive by LDS-DMA.
//   A : fp32 activations, 16-byte global loads -> registers -> split on the fly -> ds_write_b64 (hi, lo).
// LDS tiles are [128][32] fp16 (64-byte rows) with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3,
// which makes the ds_read_b128 fragment reads conflict-free (a 256-byte bank row holds 4 rows x 4 chunks).
#include "gemm_common.h"

namespace {

using namespace cvxg;
typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TILE_H = 128 * BK;            // halves per operand tile (8 KiB)
constexpr float F16_MAX = 65504.f;

__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo, float pre, float& amax)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        amax = cvx_amax3_c(amax, v[e] * pre, v[e] * pre);
        const float x = fminf(fmaxf(v[e] * pre, -F16_MAX), F16_MAX);     // saturate instead of inf - inf
        const f16 h = (f16)x;
        hi[e] = h;
        lo[e] = (f16)(x - (float)h);
    }
}

__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(const cvx_gemm_args p, const f16* __restrict__ Whi,
                                                           const f16* __restrict__ Wlo, float acc_scale, SplitOut so,
                                                           int tiles_m, int tiles_n, int map_mode)
{
    constexpr int TM = 2, BM = 128;
    extern __shared__ __attribute__((aligned(16))) f16 smem_h[];
    // stage layout: [Ahi | Alo | Whi | Wlo], two stages
    f16* const S0 = smem_h;
    constexpr int STAGE = 4 * TILE_H;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tile_m, tile_n;
    tile_of_block(tiles_m, tiles_n, map_mode, tile_m, tile_n);
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- A: register staging (fp32 -> split).  thread -> (row = tid/8 + 32*i, 4 floats at k = 4*(tid%8))
    const int srow = tid >> 3, sj = tid & 7;
    const float* pa[4];
    int64_t a_jump[4];
    int a_st[4];                                // LDS half-offset of this thread's 8-byte piece
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = srow + 32 * i;
        const int row = min(m0 + r, p.M - 1);
        pa[i] = p.A + (int64_t)row * p.lda + 4 * sj;
        a_jump[i] = p.A2 ? (p.A2 + (int64_t)row * p.lda2 + 4 * sj) - (pa[i] + p.K1) : 0;
        const int chunk = (sj >> 1) ^ ((r >> 2) & 3);
        a_st[i] = r * BK + chunk * 8 + (sj & 1) * 4;
    }
    // ---- W: LDS-DMA.  wave `wid` fills rows [32*wid, +32) of Whi and Wlo: 2 instructions of 16 rows each
    const f16* pwh[2];
    const f16* pwl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 32 * wid + 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        const int64_t off = (int64_t)min(n0 + r, p.N - 1) * p.ldw + 8 * c;
        pwh[j] = Whi + off;
        pwl[j] = Wlo + off;
    }
    const int dma_off = 32 * wid * BK;
    const int switch_tile = p.A2 ? p.K1 / BK : -1;

    f32x16 acc[TM][2];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment read offsets (halves): row (lane&31), chunk (2s + g) ^ swizzle, g = lane >> 5
    const int i31 = lane & 31, g = lane >> 5, swz = (i31 >> 2) & 3;
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = i31 * BK + 8 * ((2 * s + g) ^ swz);
    const int a_row0 = wm * 64 * BK, b_row0 = wn * 64 * BK;

    f32x4 ra[4];
    const int nk = p.K / BK;
    const float a_pre = so.a_scale ? *so.a_scale : 1.0f;      // activation pre-scale applied while splitting on the fly
    float a_amax = 0.f;
    {   // prologue: tile 0 -> stage 0
        const bool sw = (0 == switch_tile);
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (sw) pa[i] += a_jump[i]; ra[i] = gload4(pa[i]); pa[i] += BK; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            glds16(pwh[j], S0 + 2 * TILE_H + dma_off + 16 * j * BK);
            glds16(pwl[j], S0 + 3 * TILE_H + dma_off + 16 * j * BK);
            pwh[j] += BK; pwl[j] += BK;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 hi, lo;
            split4(ra[i], hi, lo, a_pre, a_amax);
            *reinterpret_cast<f16x4*>(S0 + a_st[i]) = hi;
            *reinterpret_cast<f16x4*>(S0 + TILE_H + a_st[i]) = lo;
        }
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool has_next = kt + 1 < nk;
        const int adv = has_next ? BK : 0;
        const bool sw = (kt + 1 == switch_tile);
        f16* const Sc = S0 + cur * STAGE;
        f16* const Sn = S0 + (cur ^ 1) * STAGE;
        // next tile: W by DMA into the other stage, A into registers
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f16* sh = pwh[j] - (has_next ? 0 : BK);
            const f16* sl = pwl[j] - (has_next ? 0 : BK);
            glds16(sh, Sn + 2 * TILE_H + dma_off + 16 * j * BK);
            glds16(sl, Sn + 3 * TILE_H + dma_off + 16 * j * BK);
            pwh[j] = sh + adv; pwl[j] = sl + adv;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* src = pa[i] + (sw ? a_jump[i] : 0) - (has_next ? 0 : BK);
            ra[i] = gload4(src);
            pa[i] = src + adv;
        }
        // 24 MFMAs on the current stage
        const f16* ah = Sc + a_row0;
        const f16* al = Sc + TILE_H + a_row0;
        const f16* wh = Sc + 2 * TILE_H + b_row0;
        const f16* wl = Sc + 3 * TILE_H + b_row0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 fah[TM], fal[TM], fwh[2], fwl[2];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                fah[mi] = *reinterpret_cast<const f16x8*>(ah + mi * 32 * BK + foff[s]);
                fal[mi] = *reinterpret_cast<const f16x8*>(al + mi * 32 * BK + foff[s]);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                fwh[ni] = *reinterpret_cast<const f16x8*>(wh + ni * 32 * BK + foff[s]);
                fwl[ni] = *reinterpret_cast<const f16x8*>(wl + ni * 32 * BK + foff[s]);
            }
            // term-major order: consecutive MFMAs hit DIFFERENT accumulators (a dependent chain on one accumulator
            // would wait for the previous MFMA's result every time)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[mi], fwh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi], fwl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi], fwh[ni], acc[mi][ni], 0, 0, 0);
        }
        // split + store the next A tile (the other stage was last read before the previous barrier)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 hi, lo;
            split4(ra[i], hi, lo, a_pre, a_amax);
            *reinterpret_cast<f16x4*>(Sn + a_st[i]) = hi;
            *reinterpret_cast<f16x4*>(Sn + TILE_H + a_st[i]) = lo;
        }
        __syncthreads();
    }
    cvx_sat_commit(so.sat, a_amax);
    acc_scale = total_acc_scale(acc_scale, so);
    if (acc_scale != 1.0f) {        // undo the power-of-two weight (and activation) pre-scale (exact)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= acc_scale;
    }
    gemm_epilogue<TM>(p, acc, m0, n0, wm, wn, lane, so);
}

// ------------------------------------------------------------------ all-DMA variant: A arrives pre-split
// Every operand tile (A_hi, A_lo, W_hi, W_lo: 8 KiB each) goes global -> LDS by DMA into a ring of STAGES stages;
// tiles are requested STAGES-1 steps ahead and retired with a COUNTED vmcnt + a raw s_barrier, so loads stay in
// flight across barriers (a 24-MFMA step is only ~770 cycles - far shorter than an L2/HBM round trip, which a
// 2-stage scheme cannot hide).  No staging VGPRs, no split VALU work, no ds_write in the loop.
// (PreSplitA lives in gemm_common.h: shared with gemm_f16x3_p8s.hip / _p8m.hip)

// Split-K (ksplit > 1, small problems only): blockIdx.y selects a K slice of k_per columns; the block writes its raw
// partial sums (fp32, no epilogue) to `partial` [ksplit][M][N] and splitk_reduce_kernel finishes the job.
template <int STAGES, int NT>
__global__ __launch_bounds__(256, (STAGES <= 2 ? 2 : 1)) void gemm_f16x3_dma_kernel(
    const cvx_gemm_args p_in, const PreSplitA A, const f16* __restrict__ Whi, const f16* __restrict__ Wlo,
    float acc_scale, SplitOut so, int tiles_m, int tiles_n, int map_mode, int k_per, float* __restrict__ partial)
{
    cvx_gemm_args p = p_in;
    acc_scale = total_acc_scale(acc_scale, so);
    const int k_begin = k_per > 0 ? (int)blockIdx.y * k_per : 0;
    if (k_per > 0) {            // this block: columns [k_begin, k_begin + k_per) of [A | A2] and of W, plain fp32 store
        p.K = k_per;
        p.C = partial + (int64_t)blockIdx.y * p.M * p.N; p.ldc = p.N;
        p.bias = nullptr; p.residual = nullptr; p.act = CVX_ACT_NONE; p.rope_cos = nullptr;
        so = SplitOut{nullptr, nullptr, 0, 1, nullptr, nullptr, 0};
    }
    constexpr int TM = 2, BM = 128;
    constexpr int STAGE = 4 * TILE_H;
    extern __shared__ __attribute__((aligned(16))) f16 smem_h[];
    f16* const S0 = smem_h;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tile_m, tile_n;
    tile_of_block(tiles_m, tiles_n, map_mode, tile_m, tile_n);
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // DMA sources: wave `wid` fills rows [32*wid, +32) of each of the four tiles, 16 rows per instruction
    const f16* pah[2]; const f16* pal[2]; const f16* pwh[2]; const f16* pwl[2];
    int64_t jmp_h[2], jmp_l[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 32 * wid + 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        const int64_t ra = min(m0 + r, p.M - 1), rw = min(n0 + r, p.N - 1);
        pah[j] = A.hi + ra * A.ld + 8 * c;
        jmp_h[j] = A.hi2 ? (A.hi2 + ra * A.ld2 + 8 * c) - (pah[j] + p.K1) : 0;
        pwh[j] = Whi + rw * p.ldw + 8 * c;
        if constexpr (NT == 1) {      // the "lo" slots carry the NEXT 32 k of the same fp16 operands
            pal[j] = pah[j] + BK; jmp_l[j] = jmp_h[j]; pwl[j] = pwh[j] + BK;
        } else {
            pal[j] = A.lo + ra * A.ld + 8 * c;
            jmp_l[j] = A.lo2 ? (A.lo2 + ra * A.ld2 + 8 * c) - (pal[j] + p.K1) : 0;
            pwl[j] = Wlo + rw * p.ldw + 8 * c;
        }
    }
    constexpr int KSTEP = (NT == 1) ? 2 * BK : BK;      // k consumed per stage
    const int dma_off = 32 * wid * BK;
    if (k_begin > 0) {                                  // K slice: start at column k_begin of [A | A2] and of W
        const bool in2 = A.hi2 && k_begin >= p.K1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pah[j] += k_begin + (in2 ? jmp_h[j] : 0);
            pal[j] += k_begin + (in2 ? jmp_l[j] : 0);
            pwh[j] += k_begin; pwl[j] += k_begin;
        }
    }
    const int switch_tile = (A.hi2 && k_begin < p.K1) ? (p.K1 - k_begin) / KSTEP : -1;
    const int nk = p.K / KSTEP;

    // issue the 8 DMA pieces of tile t (tiles past the end re-read the last tile: keeps the vmcnt arithmetic uniform)
    auto issue = [&](int t) {
        f16* const S = S0 + (t % STAGES) * STAGE + dma_off;
        const bool live = t < nk;
        const bool sw = live && (t == switch_tile);      // (a K slice may END exactly at the A | A2 boundary)
        const int back = live ? 0 : KSTEP, adv = live ? KSTEP : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f16* sh = pah[j] + (sw ? jmp_h[j] : 0) - back;
            const f16* sl = pal[j] + (sw ? jmp_l[j] : 0) - back;
            const f16* wh = pwh[j] - back;
            const f16* wl = pwl[j] - back;
            glds16(sh, S + 16 * j * BK);
            glds16(sl, S + TILE_H + 16 * j * BK);
            glds16(wh, S + 2 * TILE_H + 16 * j * BK);
            glds16(wl, S + 3 * TILE_H + 16 * j * BK);
            if (live) { pah[j] = sh + adv; pal[j] = sl + adv; pwh[j] = wh + adv; pwl[j] = wl + adv; }   // a dummy re-read moves nothing
        }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int i31 = lane & 31, g = lane >> 5, swz = (i31 >> 2) & 3;
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = i31 * BK + 8 * ((2 * s + g) ^ swz);
    const int a_row0 = wm * 64 * BK, b_row0 = wn * 64 * BK;

#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t) issue(t);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most (STAGES-2) younger tiles (8 pieces each) are still outstanding
        if constexpr (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (STAGES == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // everyone's pieces landed; stage (kt-1)%STAGES is free again
        issue(kt + STAGES - 1);
        const f16* Sc = S0 + (kt % STAGES) * STAGE;
        const f16* ah = Sc + a_row0;
        const f16* al = Sc + TILE_H + a_row0;
        const f16* wh = Sc + 2 * TILE_H + b_row0;
        const f16* wl = Sc + 3 * TILE_H + b_row0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 fah[TM], fal[TM], fwh[2], fwl[2];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                fah[mi] = *reinterpret_cast<const f16x8*>(ah + mi * 32 * BK + foff[s]);
                fal[mi] = *reinterpret_cast<const f16x8*>(al + mi * 32 * BK + foff[s]);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                fwh[ni] = *reinterpret_cast<const f16x8*>(wh + ni * 32 * BK + foff[s]);
                fwl[ni] = *reinterpret_cast<const f16x8*>(wl + ni * 32 * BK + foff[s]);
            }
            // term-major order: consecutive MFMAs hit DIFFERENT accumulators (a dependent chain on one accumulator
            // would wait for the previous MFMA's result every time)
            if constexpr (NT == 1) {     // plain fp16: slot "lo" is the next 32 k, one product each
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi], fwh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[mi], fwl[ni], acc[mi][ni], 0, 0, 0);
            } else {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[mi], fwh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi], fwl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi], fwh[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the tail re-reads before LDS is released
    if (acc_scale != 1.0f) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= acc_scale;
    }
    gemm_epilogue<TM>(p, acc, m0, n0, wm, wn, lane, so);
}

__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ w, f16* __restrict__ hi,
                                                       f16* __restrict__ lo, int64_t n, float scale, const float* __restrict__ scale_dev,
                                                       uint32_t* __restrict__ sat)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (scale_dev) scale *= *scale_dev;
    cvx_sat_commit(sat, fabsf(w[i] * scale));
    const float x = fminf(fmaxf(w[i] * scale, -F16_MAX), F16_MAX);
    const f16 h = (f16)x;
    const int64_t o = (lo == hi + 32) ? (((i >> 5) << 6) | (i & 31)) : i;      // interleaved pair: [hi 32 | lo 32] blocks
    hi[o] = h;
    if (lo) lo[o] = (f16)(x - (float)h);
}

// n_sets interleaved split copies of W with scaled columns (deferred norm, gamma on the weight side): thread = 4 consecutive k of
// one row, W read once, one (hi, lo) store pair per set
__global__ __launch_bounds__(256) void split_colscale_il_kernel(const float* __restrict__ W, int64_t ldw, int N, int K,
                                                               const float* __restrict__ colscale, int64_t cs_ld,
                                                               const float* __restrict__ set_scale, int64_t ss_ld, int n_sets, float scale,
                                                               f16* __restrict__ out, uint32_t* __restrict__ sat)
{
    const int k4 = K / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * k4) return;
    const int n = (int)(i / k4), k = 4 * (int)(i - (int64_t)n * k4);
    const f32x4 w = gload4(W + (int64_t)n * ldw + k);
    const int64_t o = (int64_t)n * 2 * K + (((k >> 5) << 6) | (k & 31));
    float amax = 0.f;
    for (int s = 0; s < n_sets; ++s) {
        const f32x4 g = gload4(colscale + (int64_t)s * cs_ld + k);
        const float sc = scale * (set_scale ? set_scale[(int64_t)s * ss_ld] : 1.f);
        cvx_f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = w[e] * g[e] * sc;
            amax = cvx_amax3_c(amax, x, x);
            x = fminf(fmaxf(x, -F16_MAX), F16_MAX);
            h[e] = (f16)x;
            l[e] = (f16)(x - (float)h[e]);
        }
        f16* dst = out + (int64_t)s * N * 2 * K + o;
        *reinterpret_cast<cvx_f16x4*>(dst) = h;
        *reinterpret_cast<cvx_f16x4*>(dst + 32) = l;
    }
    cvx_sat_commit(sat, amax);
}

}  // namespace

extern "C" int cvx_split_f16_colscale_il(const float* W, int64_t ldw, int32_t N, int32_t K, const float* colscale, int64_t cs_ld,
                                         const float* set_scale_dev, int64_t ss_ld, int32_t n_sets, float scale, uint16_t* out, cvx_stream_t s)
{
    CVX_REQUIRE(W && colscale && out && N > 0 && K > 0 && K % 32 == 0 && ldw >= K && ldw % 4 == 0 && cs_ld % 4 == 0 && n_sets >= 0 &&
                (((uintptr_t)W | (uintptr_t)colscale | (uintptr_t)out) & 15) == 0,
                "split_f16_colscale_il: bad arguments (N=%d K=%d ldw=%ld cs_ld=%ld; K %% 32 == 0, 16-byte aligned rows)", N, K, (long)ldw, (long)cs_ld);
    if (n_sets == 0) return CVX_OK;
    CVX_REQUIRE_SAT(s);
    const int64_t n = (int64_t)N * (K / 4);
    hipLaunchKernelGGL(split_colscale_il_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       W, ldw, N, K, colscale, cs_ld, set_scale_dev, ss_ld, n_sets, scale, reinterpret_cast<f16*>(out), cvx_sat_flag_for(s));
    CVX_CHECK_LAUNCH("cvx_split_f16_colscale_il");
    return CVX_OK;
}

extern "C" int cvx_split_f16_dev(const float* w, uint16_t* hi, uint16_t* lo, int64_t n, float scale, const float* scale_dev,
                                 cvx_stream_t s)
{
    CVX_REQUIRE(w && hi && n >= 0, "split_f16: bad arguments");      // lo == NULL: plain fp16 cast (saturating)
    if (n == 0) return CVX_OK;
    CVX_REQUIRE_SAT(s);
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       w, reinterpret_cast<f16*>(hi), reinterpret_cast<f16*>(lo), n, scale, scale_dev, cvx_sat_flag_for(s));
    CVX_CHECK_LAUNCH("cvx_split_f16");
    return CVX_OK;
}

extern "C" int cvx_split_f16(const float* w, uint16_t* hi, uint16_t* lo, int64_t n, float scale, cvx_stream_t s)
{
    return cvx_split_f16_dev(w, hi, lo, n, scale, nullptr, s);
}


// out = epilogue( sum_s partial[s] ) in a fixed order: bias -> act -> residual -> fp32 and / or split store
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int ksplit, const cvx_gemm_args p, SplitOut so)
{
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t total = (int64_t)p.M * p.N;
    if (i4 >= total) return;
    const int row = (int)(i4 / p.N), col = (int)(i4 - (int64_t)row * p.N);          // N % 4 == 0: the 4 columns share a row
    f32x4 v = *reinterpret_cast<const f32x4*>(partial + i4);
    for (int s = 1; s < ksplit; ++s) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(partial + (int64_t)s * total + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] + (p.bias ? p.bias[col + e] : 0.f);
        if (p.act == CVX_ACT_GELU) x = gelu_erf(x);
        else if (p.act == CVX_ACT_SILU) x = silu(x);
        if (p.residual) x += p.residual[(int64_t)row * p.ldr + col + e];
        v[e] = x;
    }
    if (so.write_f32) *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = v;
    if (so.hi) {
        const float cs = so.c_scale ? *so.c_scale : 1.f;
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) store_split(so, (int64_t)row * so.ldc_h + ((so.lo == so.hi + 32) ? il_col(col + e) : col + e), v[e] * cs, amax);
        cvx_sat_commit(so.sat, amax);
    }
}

// The same reduction for whole rows (one wave per row, N <= 1024, N % 256 == 0) FOLLOWED by the row's AdaptiveRMSNorm, whose
// split pair feeds the next GEMM (cvx_gemm_f16x3_norm): the arithmetic of splitk_reduce_kernel (act == NONE) and of
// adarmsnorm_kernel<NV> (elementwise.hip), operation for operation, on the registers that hold the finished row.
typedef _Float16 f16x4_r __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_split4_r(_Float16* hi, _Float16* lo, int64_t row_off, int col, const f32x4 o, CvxSat& amax)
{
    cvx_amax4(amax, o);
    const int64_t off = row_off + ((lo == hi + 32) ? il_col(col) : col);
    f16x4_r h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = fminf(fmaxf(o[e], -65504.f), 65504.f);
        h[e] = (_Float16)x;
        l[e] = (_Float16)(x - (float)h[e]);
    }
    *reinterpret_cast<f16x4_r*>(hi + off) = h;
    if (lo) *reinterpret_cast<f16x4_r*>(lo + off) = l;
}
// One BLOCK per row, one quad of columns per thread (NW = N / 256 waves): a wave per row (the norm kernel's shape) leaves this
// latency-bound pass with one wave per SIMD.  The row's sum of squares is put together in the norm kernel's order - per lane
// the quads lane, lane + 64, ... in sequence, then the wave butterfly - so both paths round identically.
template <int NW>
__global__ __launch_bounds__(64 * NW) void splitk_reduce_norm_kernel(const float* __restrict__ partial, int ksplit, const cvx_gemm_args p, SplitOut so,
                                                                    const cvx_gemm_norm nm)
{
    __shared__ float qs[NW][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = (int)blockIdx.x;
    const int64_t total = (int64_t)p.M * p.N;
    const int col = 4 * (int)threadIdx.x;
    const float* pr = partial + (int64_t)row * p.N + col;
    f32x4 v = gload4(pr);
    const f32x4 gv = gload4(nm.gamma + col);
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, rv = bv, bi = bv;
    if (nm.beta) bv = gload4(nm.beta + col);
    if (p.residual) rv = gload4(p.residual + (int64_t)row * p.ldr + col);
    if (p.bias) bi = gload4(p.bias + col);
    for (int s = 1; s < ksplit; ++s) {
        const f32x4 w = gload4(pr + (int64_t)s * total);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
    const float cs = (so.hi && so.c_scale) ? *so.c_scale : 1.f;
    const float ys = nm.y_scale_dev ? *nm.y_scale_dev : 1.f;
    CvxSat amax;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] + (p.bias ? bi[e] : 0.f);
        if (p.residual) x += rv[e];
        v[e] = x;
    }
    if (so.write_f32) *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = v;
    if (so.hi) store_split4_r(so.hi, so.lo, (int64_t)row * so.ldc_h, col, f32x4{v[0] * cs, v[1] * cs, v[2] * cs, v[3] * cs}, amax);
    float ss = 0.f;
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    if (NW > 1) {
        qs[wv][lane] = ss;
        __syncthreads();
        ss = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) ss += qs[w][lane];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = nm.scale / fmaxf(sqrtf(ss), nm.eps);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] * inv * gv[e];
    if (nm.beta) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += bv[e];
    }
    store_split4_r(reinterpret_cast<_Float16*>(nm.Y_hi), reinterpret_cast<_Float16*>(nm.Y_lo), (int64_t)row * nm.ldy_h, col,
                   f32x4{o[0] * ys, o[1] * ys, o[2] * ys, o[3] * ys}, amax);
    cvx_sat_commit(so.sat, amax);
}

template <int STAGES, int NT>
static void launch_dma(const cvx_gemm_args& a, const PreSplitA& A, const f16* wh, const f16* wl, float acc_scale,
                       const SplitOut& so, dim3 grid, int tiles_m, int tiles_n, int map_mode, hipStream_t st,
                       int k_per = 0, float* partial = nullptr)
{
    const size_t lds = (size_t)STAGES * 4 * TILE_H * sizeof(f16);
    cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_f16x3_dma_kernel<STAGES, NT>), (int)lds);
    hipLaunchKernelGGL((gemm_f16x3_dma_kernel<STAGES, NT>), grid, dim3(256), lds, st, a, A, wh, wl, acc_scale, so,
                       tiles_m, tiles_n, map_mode, k_per, partial);
}

constexpr int SPLITK_MAX_GRID = 128;      // small problems only: at most this many 128 x 128 output tiles

// K slices a small pre-split problem is cut into (1 = never): few output tiles and a long K
static int splitk_factor(int M, int N, int K, int K1, bool has_a2)
{
    const int tiles = ((N + BN - 1) / BN) * ((((M + 127) / 128) + 7) / 8 * 8);
    if (tiles > SPLITK_MAX_GRID || N % 4 != 0) return 1;
    for (int cand = 4; cand >= 2; cand >>= 1) {
        const int kp = K / cand;
        if (K % cand == 0 && kp % 64 == 0 && kp >= 512 && (!has_a2 || K1 % kp == 0)) return cand;
    }
    return 1;
}

// the same for the medium-problem kernel (gemm_f16x3_p8m.hip: interleaved operands, fewer than 2048 rows): slices of at least
// 8 K-tiles, as many as keep the grid within one block per CU
static int splitk_factor_p8m(int M, int N, int K, int K1, bool has_a2)
{
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    if (N % 4 != 0) return 1;
    for (int cand = 8; cand >= 2; cand >>= 1) {
        const int kp = K / cand;
        if (tiles * cand <= (cand == 8 ? 128 : 256) && K % (32 * cand) == 0 && kp >= (cand == 8 ? 128 : 256) && (!has_a2 || K1 % kp == 0)) return cand;
    }
    return 1;
}

extern "C" int64_t cvx_gemm_f16x3_workspace_floats(int32_t M, int32_t N, int32_t K, int32_t K1)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int f = splitk_factor(M, N, K, K1, K1 > 0);
    const int g = M < 2048 ? splitk_factor_p8m(M, N, K, K1, K1 > 0) : 1;
    const int m = f > g ? f : g;
    return m > 1 ? (int64_t)m * M * N : 0;
}

static int gemm_f16x3_impl(const cvx_gemm_args* a, const uint16_t* W_hi, const uint16_t* W_lo, float acc_scale,
                           const cvx_gemm_split_io* io, const cvx_gemm_norm* norm, cvx_stream_t s);

extern "C" int cvx_gemm_f16x3(const cvx_gemm_args* a, const uint16_t* W_hi, const uint16_t* W_lo, float acc_scale,
                              const cvx_gemm_split_io* io, cvx_stream_t s)
{
    return gemm_f16x3_impl(a, W_hi, W_lo, acc_scale, io, nullptr, s);
}

extern "C" int cvx_gemm_f16x3_norm(const cvx_gemm_args* a, const uint16_t* W_hi, const uint16_t* W_lo, float acc_scale,
                                   const cvx_gemm_split_io* io, const cvx_gemm_norm* norm, cvx_stream_t s)
{
    CVX_REQUIRE(a && norm && norm->gamma && norm->Y_hi, "gemm_f16x3_norm: null pointer");
    CVX_REQUIRE(a->act == CVX_ACT_NONE && a->ldc == a->N && a->N % 4 == 0 && (!io || io->write_f32 != 0 || !io->C_hi) && !a->rope_cos,
                "gemm_f16x3_norm: needs act == NONE, a contiguous fp32 C (ldc == N, written) and no RoPE");
    CVX_REQUIRE((norm->Y_lo == norm->Y_hi + 32 && norm->ldy_h == 2 * (int64_t)a->N && a->N % 32 == 0) || norm->ldy_h == a->N,
                "gemm_f16x3_norm: Y must be an interleaved pair with ldy_h == 2N or a plain pair with ldy_h == N");
    return gemm_f16x3_impl(a, W_hi, W_lo, acc_scale, io, norm, s);
}

static int gemm_f16x3_impl(const cvx_gemm_args* a, const uint16_t* W_hi, const uint16_t* W_lo, float acc_scale,
                           const cvx_gemm_split_io* io, const cvx_gemm_norm* norm, cvx_stream_t s)
{
    const int rc = cvxg::validate_gemm_args(a);
    if (rc != CVX_OK) return rc;
    bool norm_done = false;
    CVX_REQUIRE(W_hi, "gemm_f16x3: null split weights");
    const bool single = (W_lo == nullptr);      // plain fp16 operands (hi halves only), one MFMA product
    CVX_REQUIRE(a->K % BK == 0 && a->ldw % 8 == 0, "gemm_f16x3: K must be a multiple of 32 and ldw of 8 (K=%d ldw=%ld)", a->K, (long)a->ldw);
    CVX_REQUIRE((((uintptr_t)W_hi | (uintptr_t)W_lo) & 15) == 0, "gemm_f16x3: split weights must be 16-byte aligned");
    const bool w_il = io && io->w_interleaved != 0;          // [N][K/32][hi 32 | lo 32]: W_lo == W_hi + 32, ldw == 2K
    // interleaved activations: A_lo == A_hi + 32 (and A2_lo == A2_hi + 32), lda_h >= 2K - only together with interleaved weights
    const bool a_il = io && io->A_hi && io->A_lo == io->A_hi + 32;
    if (w_il)
        CVX_REQUIRE(!single && W_lo == W_hi + 32 && a->ldw >= 2 * (int64_t)a->K && a_il,
                    "gemm_f16x3: interleaved weights need W_lo == W_hi + 32, ldw >= 2K and an interleaved pre-split A");
    if (a_il)
        CVX_REQUIRE(w_il && io->lda_h >= 2 * (int64_t)(a->A2 ? a->K1 : a->K) &&
                    (!a->A2 || (io->A2_lo == io->A2_hi + 32 && io->lda2_h >= 2 * (int64_t)(a->K - a->K1))),
                    "gemm_f16x3: interleaved A needs interleaved weights, lda_h >= 2K and an interleaved A2");
    else
        CVX_REQUIRE(!(io && a->A2 && io->A2_hi && io->A2_lo == io->A2_hi + 32),
                    "gemm_f16x3: A2 is an interleaved pair but A is not (both operands must use the same layout)");
    if (single)
        CVX_REQUIRE(io && io->A_hi && a->K % (2 * BK) == 0 && (!a->A2 || a->K1 % (2 * BK) == 0),
                    "gemm_f16x3: the single-term mode (W_lo == NULL) needs a pre-split A and K (K1) a multiple of 64");
    if ((io && (io->C_hi || io->Vt_hi)) || norm) CVX_REQUIRE_SAT(s);      // (the call stores split pairs)
    SplitOut so{nullptr, nullptr, 0, 1, nullptr, nullptr, 0};
    so.sat = cvx_sat_flag_for(s);
    PreSplitA A{nullptr, nullptr, 0, nullptr, nullptr, 0};
    if (io) {
        if (io->C_hi || io->C_lo) {
            CVX_REQUIRE(io->C_hi && (io->C_lo || single) && io->ldc_h >= ((io->Vt_hi && a->rope_cols > 0) ? a->rope_cols : a->N), "gemm_f16x3: bad split output");
            so.hi = reinterpret_cast<f16*>(io->C_hi); so.lo = reinterpret_cast<f16*>(io->C_lo); so.ldc_h = io->ldc_h;
        }
        so.write_f32 = (io->write_f32 != 0 || so.hi == nullptr) ? 1 : 0;
        so.c_scale = io->c_scale_dev; so.vt_scale = io->vt_scale_dev; so.a_scale = io->a_scale_dev;
        so.tw_gamma = io->c_gamma_dev; so.rowsq = io->c_rowsq; so.rowsq_ld = (int)io->c_rowsq_ld; so.row_scale = io->a_row_scale_dev;
        so.res_hi = reinterpret_cast<const f16*>(io->R_hi); so.res_lo = reinterpret_cast<const f16*>(io->R_lo); so.res_ld = io->ldr_h;
        so.res_scale = io->r_scale_dev; so.a2_scale = io->a2_scale_dev;
#ifdef CVX_DEV_FLAGS          // timing experiments (tools/): epilogue skipping, per-block stamps, one tile per block - never in the shipped library
        so.dbg = io->flags >> 8; so.trace = (so.dbg & 4) ? reinterpret_cast<unsigned long long*>(io->workspace) : nullptr;
#else
        so.dbg = 0; so.trace = nullptr;
#endif
        if (io->flags & CVX_GEMM_FLAG_ONE_TILE) so.dbg |= 8;      // scheduling only: same arithmetic, bit-identical results
        if (io->flags & CVX_GEMM_FLAG_TILE192) so.dbg |= 16;      // tile height of the large-problem kernel pinned (same bits either way)
        if (io->flags & CVX_GEMM_FLAG_TILE256) so.dbg |= 32;
        if (io->flags & CVX_GEMM_FLAG_TILE_MIXED) so.dbg |= 64;
        if (io->Vt_hi || io->Vt_lo) {
            CVX_REQUIRE(io->Vt_hi && (io->Vt_lo || single) && so.hi && a->rope_cos && a->rope_cols > 0 && a->rope_cols % 128 == 0 &&
                        (a->N - a->rope_cols) * 2 == a->rope_cols && io->vt_ld >= ((a->rope_T + 15) / 16) * 16 && io->vt_ld % 8 == 0 &&
                        a->M % a->rope_T == 0 && so.write_f32 == 0,
                        "gemm_f16x3: QKV-transpose output needs the RoPE arguments, N = 3*H*64, vt_ld >= T rounded up to 16 and write_f32 = 0");
            so.vt_hi = reinterpret_cast<f16*>(io->Vt_hi); so.vt_lo = reinterpret_cast<f16*>(io->Vt_lo); so.vt_ld = io->vt_ld;
        }
        if (io->A_hi || io->A_lo) {
            CVX_REQUIRE(io->A_hi && (io->A_lo || single) && io->lda_h % 8 == 0 && (((uintptr_t)io->A_hi | (uintptr_t)io->A_lo) & 15) == 0,
                        "gemm_f16x3: bad pre-split A");
            A.hi = reinterpret_cast<const f16*>(io->A_hi); A.lo = reinterpret_cast<const f16*>(io->A_lo); A.ld = io->lda_h;
            if (a->A2) {
                CVX_REQUIRE(io->A2_hi && (io->A2_lo || single) && io->lda2_h % 8 == 0 &&
                            (((uintptr_t)io->A2_hi | (uintptr_t)io->A2_lo) & 15) == 0, "gemm_f16x3: bad pre-split A2");
                A.hi2 = reinterpret_cast<const f16*>(io->A2_hi); A.lo2 = reinterpret_cast<const f16*>(io->A2_lo); A.ld2 = io->lda2_h;
            }
        }
    }
    // deferred norm (producer: gamma on the twin + row sums of squares; consumer: a factor per row): the 16x16x32 epilogues only
    const bool dn = so.tw_gamma || so.rowsq || so.row_scale || so.res_hi || so.res_lo || so.a2_scale;
    if (dn) {
        CVX_REQUIRE(!single && !norm && a->N % 64 == 0 && (!so.tw_gamma || (so.hi && ((uintptr_t)so.tw_gamma & 15) == 0)) &&
                    (!so.rowsq || (io->c_rowsq_ld >= a->N / 64 && io->c_rowsq_ld < (1ll << 20))),
                    "gemm_f16x3: deferred norm needs N %% 64 == 0, a split output for c_gamma_dev (16-byte aligned) and c_rowsq_ld >= N / 64");
        CVX_REQUIRE((so.res_hi == nullptr) == (so.res_lo == nullptr) && (!so.res_hi || (!a->residual && io->ldr_h >= (so.res_lo == so.res_hi + 32 ? 2 : 1) * (int64_t)a->N)),
                    "gemm_f16x3: a pair residual needs R_hi and R_lo, ldr_h >= N (2N interleaved) and residual == NULL");
    }
    if (a->M == 0) return CVX_OK;
    const int tiles_n = (a->N + BN - 1) / BN, tiles_m = (a->M + 127) / 128;
    const int map_mode = 1;             // XCD-aware block -> tile map (gemm_common.h)
    const int grid_m = map_mode == 1 ? ((tiles_m + 7) / 8) * 8 : tiles_m;
    dim3 grid((unsigned)(grid_m * tiles_n));
    hipStream_t st = cvx_hip_stream(s);
    const f16* wh = reinterpret_cast<const f16*>(W_hi);
    const f16* wl = reinterpret_cast<const f16*>(W_lo);
    // Measured on MI355X (tools/bench_kernels.py, M=16000): the 256x256 tile wins on every transformer shape
    // (284-349 vs 263-312 TFLOP/s).  Deeper rings (3-4 stages, K-step 16, 256x128x3) were tried and are slower:
    // the kernel is bound by the per-CU LDS-DMA delivery rate (~35 GB/s/CU), not by DMA latency.
    // 2048 rows and more on 256 x 256 tiles run in ROUNDS of one tile per CU: 4000 rows (two utterances) x N = 1024 are 64 tiles =
    // one round on a quarter of the chip; 9298 rows (the second bin of a directory) 148 tiles = one round at 58 %.  The
    // medium-problem kernel works in 128 x 128 tiles at ~80 % of the large kernel's rate per tile area (measured at 16000 rows:
    // 1.20x the time), so a tile costs 0.31 of a large one: it takes the problem when its rounds come out shorter - out / ff2 /
    // skip at 4000 rows 2.5x faster, at 9298 rows 5 %; to_qkv / ff1 stay on the large kernel from ~4000 rows on.
    // (CVX_GEMM_FLAG_MEDIUM forces it, CVX_GEMM_FLAG_NO_MEDIUM and the A/B kernel flags keep the large kernel.)
    bool medium = a->M < 2048 || a->N < 512;
    if (!medium && A.hi && w_il && a_il && io && !(io->flags & (CVX_GEMM_FLAG_NO_MEDIUM | CVX_GEMM_FLAG_ONE_TILE))) {
        const long ncu = cvx_ctx_cus(s);
        const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256), t192 = (long)((a->M + 191) / 192) * ((a->N + 255) / 256);
        const long t128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128);
        // (the large kernel picks 192-row tiles - 0.8 of a 256-row tile's time - where that gives fewer, shorter rounds: gemm_f16x3_p8s.hip)
        const double r256 = (double)((t256 + ncu - 1) / ncu), r192 = 0.8 * (double)((t192 + ncu - 1) / ncu);
        const double large = r192 < r256 ? r192 : r256, med = 0.31 * (double)((t128 + ncu - 1) / ncu);
        medium = (io->flags & CVX_GEMM_FLAG_MEDIUM) || med < 0.97 * large;
    }
    if (dn) {
        CVX_REQUIRE(A.hi && w_il && a_il && a->M >= 2048 && a->N >= 512,
                    "gemm_f16x3: deferred norm (c_gamma_dev / c_rowsq / a_row_scale_dev) runs on the large-problem 16x16x32 kernel only: "
                    "interleaved pre-split operands, M >= 2048, N >= 512 (M=%d N=%d K=%d)", a->M, a->N, a->K);
        medium = false;
    }
    if (A.hi && w_il && a_il && medium) {
        // fewer than 2048 rows (one utterance, the last bin of a ragged directory, the HuBERT / text2semantic encoders): 128 x 128
        // tiles with two wave groups on alternate K-tiles (gemm_f16x3_p8m.hip); K slices on separate blocks when the output has too
        // few tiles for the chip and the caller provided the scratch
        int ksplit = 1;
        if (io->workspace && !a->rope_cos && !so.vt_hi && a->ldc % 4 == 0) {
            const int f = splitk_factor_p8m(a->M, a->N, a->K, a->K1, a->A2 != nullptr);
            if (f > 1 && io->workspace_floats >= (int64_t)f * a->M * a->N) ksplit = f;
        }
#ifdef CVX_DEV_FLAGS
        if (io->flags & 0x10000) ksplit = 1;               // (dev A/B: no K slices; 0x20000: two)
        if ((io->flags & 0x20000) && ksplit > 2) ksplit = 2;
        if ((so.dbg & 4) && io->workspace)                 // stamps behind the split-K partials (tools/archive/gemm_small_trace.py)
            so.trace = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(io->workspace) + ((size_t)100 << 20));
#endif
        CVX_REQUIRE(cvxg::launch_gemm_f16x3_p8m(*a, A, wh, acc_scale, so, ksplit, ksplit > 1 ? io->workspace : nullptr, st),
                    "gemm_f16x3: interleaved operands below 2048 rows need N %% 16 == 0 (N %% 64 == 0 with RoPE), 16-byte aligned C / residual / "
                    "bias / RoPE tables and rope_cols %% 128 == 0 (M=%d N=%d K=%d)", a->M, a->N, a->K);
        if (ksplit > 1 && norm && a->N <= 1024 && a->N % 256 == 0 && a->ldc % 4 == 0 && (!a->residual || a->ldr % 4 == 0)) {
            const dim3 g((unsigned)a->M);                      // the row's norm rides in the reduction: one block per row
            if (a->N == 1024) hipLaunchKernelGGL(splitk_reduce_norm_kernel<4>, g, dim3(256), 0, st, io->workspace, ksplit, *a, so, *norm);
            else if (a->N == 768) hipLaunchKernelGGL(splitk_reduce_norm_kernel<3>, g, dim3(192), 0, st, io->workspace, ksplit, *a, so, *norm);
            else if (a->N == 512) hipLaunchKernelGGL(splitk_reduce_norm_kernel<2>, g, dim3(128), 0, st, io->workspace, ksplit, *a, so, *norm);
            else hipLaunchKernelGGL(splitk_reduce_norm_kernel<1>, g, dim3(64), 0, st, io->workspace, ksplit, *a, so, *norm);
            norm_done = true;
        } else if (ksplit > 1) {
            const int64_t quads = ((int64_t)a->M * a->N + 3) / 4;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, io->workspace, ksplit, *a, so);
        }
    } else if (A.hi && w_il && a_il) {
        // 2048 rows and more, 512 columns and more, interleaved operands: the eight-phase ping-pong kernel on the 16x16x32 MFMA
        // (gemm_f16x3_p8s.hip).  Shapes it cannot take (N % 64 != 0, RoPE on other than 256-column groups, epilogue operands that are not
        // 16-byte aligned) run on the medium-problem kernel's 128 x 128 tiles, which asks for less (N % 16 == 0, 128-column RoPE groups).
        // (Rounds 2-4 also shipped a two-stage 256 x 256 kernel and an eight-phase form on the 32x32x16 MFMA as fallbacks and A/B
        //  partners: superseded, removed in round 5 - HISTORY.md has their numbers.)
        if (!cvxg::launch_gemm_f16x3_p8s(*a, A, wh, acc_scale, so, st, cvx_ctx_cus(s))) {
            CVX_REQUIRE(!dn, "gemm_f16x3: deferred norm (c_gamma_dev / c_rowsq / a_row_scale_dev) needs N %% 64 == 0 and 16-byte aligned epilogue "
                             "operands (M=%d N=%d K=%d)", a->M, a->N, a->K);
            CVX_REQUIRE(cvxg::launch_gemm_f16x3_p8m(*a, A, wh, acc_scale, so, 1, nullptr, st),
                        "gemm_f16x3: interleaved operands need N %% 16 == 0 (N %% 64 == 0 with RoPE), 16-byte aligned C / residual / bias / RoPE "
                        "tables and rope_cols %% 128 == 0 (M=%d N=%d K=%d)", a->M, a->N, a->K);
        }
    } else if (A.hi) {
        // Small problems (one utterance: M ~ 1000) leave most CUs with at most one block of 4 waves and nothing to hide
        // the DMA / LDS round trips behind.  Two remedies (measured on BASELINE config 2, tools/bench_c2.py):
        //   * split K into up to 4 slices on separate blocks when the caller provides a workspace and the grid is small
        //     (partials in fp32, summed in a fixed order by splitk_reduce_kernel, which also applies the epilogue);
        //   * a 4-stage DMA ring when the whole grid fits the chip at one block per CU.
        // Large grids keep the 2-stage kernel at two blocks per CU.
        int ksplit = 1;
        if (io && io->workspace && !a->rope_cos && !so.vt_hi && a->ldc % 4 == 0) {
            const int f = splitk_factor(a->M, a->N, a->K, a->K1, a->A2 != nullptr);
            if (f > 1 && io->workspace_floats >= (int64_t)f * a->M * a->N) ksplit = f;
        }
        const int k_per = ksplit > 1 ? a->K / ksplit : 0;
        float* part = ksplit > 1 ? io->workspace : nullptr;
        dim3 g2(grid.x, (unsigned)ksplit);
        const bool deep = (int)(grid.x * ksplit) <= 256;
        if (deep) {
            if (single) launch_dma<4, 1>(*a, A, wh, wl, acc_scale, so, g2, tiles_m, tiles_n, map_mode, st, k_per, part);
            else launch_dma<4, 3>(*a, A, wh, wl, acc_scale, so, g2, tiles_m, tiles_n, map_mode, st, k_per, part);
        } else if (single) launch_dma<2, 1>(*a, A, wh, wl, acc_scale, so, g2, tiles_m, tiles_n, map_mode, st, k_per, part);
        else launch_dma<2, 3>(*a, A, wh, wl, acc_scale, so, g2, tiles_m, tiles_n, map_mode, st, k_per, part);   // 2 stages, 2 blocks / CU
        if (ksplit > 1) {
            const int64_t quads = ((int64_t)a->M * a->N + 3) / 4;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, part, ksplit, *a, so);
        }
    } else {
        const size_t lds = (size_t)2 * 4 * TILE_H * sizeof(f16);      // 64 KiB
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x3_kernel), (int)lds);
        hipLaunchKernelGGL(gemm_f16x3_kernel, grid, dim3(256), lds, st, *a, wh, wl, acc_scale, so, tiles_m, tiles_n, map_mode);
    }
    CVX_CHECK_LAUNCH("cvx_gemm_f16x3");
    if (norm && !norm_done)
        return cvx_adarmsnorm_scaled_f32(a->C, norm->gamma, norm->beta, nullptr, norm->Y_hi, norm->Y_lo, a->M, a->N, a->M > 0 ? a->M : 1,
                                         norm->scale, norm->eps, norm->y_scale_dev, s);
    return CVX_OK;
}
