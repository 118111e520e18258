// Shared by the eight-phase split-precision GEMM kernels on v_mfma_f32_16x16x32_f16 (gemm_f16x3_p8s.hip: 256 x 256 tiles,
// gemm_f16x3_p8m.hip: 128 x 128 tiles for problems of 128 ... 2047 rows): the inline-asm LDS-DMA pair, the packed-fp32 split
// helpers and the two transpose-free epilogues, templated on the number MI of 16-row accumulator tiles a wave owns.
#pragma once
#include "gemm_common.h"

namespace {

using namespace cvxg;
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


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

// Column permutation of the pair-moving instances (the deferred-norm forms and to_qkv / ff1 of the large-problem kernel; the medium
// kernel passes its own switch).  LDS row rho = 16 h + 4 g + e of a 32-row group of
// the W tile holds weight row 8 g + 4 h + e, so accumulator tile ni = 2 t + h of lane group g (lane >> 4) holds the output columns
// 32 t + 8 g + 4 h + (0..3): the tile pair (2 t, 2 t + 1) is 8 CONSECUTIVE columns per lane.
#ifndef CVX_P8S_PERM
#define CVX_P8S_PERM 1                         // dev A/B: 0 = the deferred-norm instances without the permutation (8-byte pair accesses)
#endif
// (and the large-problem kernel's plain to_qkv / ff1 instances: the same 16-byte pair stores)
#ifndef CVX_P8S_PERM_PLAIN
#define CVX_P8S_PERM_PLAIN 1                   // dev A/B: 0 = only the deferred-norm instances
#endif
__host__ __device__ constexpr bool epi_perm(int epi) { return CVX_P8S_PERM && (epi >= EPI_RES_TW || (CVX_P8S_PERM_PLAIN && (epi == EPI_QKV || epi == EPI_GELU_SPLIT))); }
__device__ __forceinline__ int perm32(int r) { return (r & ~31) | ((r & 12) << 1) | ((r & 16) >> 2) | (r & 3); }
// first of the 4 columns tile ni of lane group lc / 4 holds, relative to the wave tile's first column
template <bool PERM> __device__ __forceinline__ int tile_col(int ni, int lc) { return PERM ? 32 * (ni >> 1) + 2 * lc + 4 * (ni & 1) : 16 * ni + lc; }

#ifndef CVX_P8S_RES_AHEAD
#define CVX_P8S_RES_AHEAD 2
#endif
#ifndef CVX_P8S_AMAX
#define CVX_P8S_AMAX 0
#endif
#define CVX_P8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define CVX_P8_WAIT_DMA() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define CVX_P8_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ---- packed fp32 helpers.  The epilogues are VALU-issue bound (ff1: ~30 VALU instructions per output element, two waves per
// SIMD in their epilogue at the same time = 19 us of an 83 us tile): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two
// elements per issue slot, and an MFMA accumulator block of four registers is two aligned register pairs, so the whole
// element-wise chain is written on pairs.  Per element the same IEEE operations in the same order as the scalar form.
// a * b with NO licence to be fused into a neighbouring add (the RoPE rotation fixes its rounding points)
#pragma clang fp contract(off)
__device__ __forceinline__ f32x2 mul2_rn(const f32x2 a, const f32x2 b) { return a * b; }
#pragma clang fp contract(fast)


// (hi, lo) fp16 halves of two fp32 values, saturating: v_med3 clamp, packed RNE conversion, exact residual x - hi by
// v_fma_mix_f32 straight from the packed halves, packed conversion of the residuals
__device__ __forceinline__ void split2_pk(const f32x2 v, f16x2& hi, f16x2& lo, CvxSat& amax)
{
#if CVX_P8S_AMAX == 1                          // dev A/B of the saturation bookkeeping: plain C
    amax.m = fmaxf(amax.m, fmaxf(fabsf(v[0]), fabsf(v[1])));
#elif CVX_P8S_AMAX == 2                        // on the clamped values' bit patterns (integer max of the magnitudes)
    amax.m = __builtin_bit_cast(float, max(__builtin_bit_cast(unsigned, amax.m), max(__builtin_bit_cast(unsigned, v[0]) & 0x7fffffffu, __builtin_bit_cast(unsigned, v[1]) & 0x7fffffffu)));
#else
    cvx_amax3(amax, v[0], v[1]);
#endif
    const float x0 = __builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f), x1 = __builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f);
    hi = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    const unsigned int hb = __builtin_bit_cast(unsigned int, hi);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(x1));
    lo = __builtin_convertvector(f32x2{r0, r1}, f16x2);
}
__device__ __forceinline__ void split4_pk(const f32x4 v, f16x4& hi, f16x4& lo, CvxSat& amax)
{
    f16x2 h01, h23, l01, l23;
    split2_pk(f32x2{v[0], v[1]}, h01, l01, amax);
    split2_pk(f32x2{v[2], v[3]}, h23, l23, amax);
    hi = f16x4{h01[0], h01[1], h23[0], h23[1]};
    lo = f16x4{l01[0], l01[1], l23[0], l23[1]};
}

__device__ __forceinline__ f32x2 silu2(const f32x2 v) { return f32x2{silu(v[0]), silu(v[1])}; }

// what an epilogue reads besides the accumulators, requested BEFORE the K loop by the medium-problem kernel (a tile there is one
// short K loop; per-block stamps showed 7-10 k cycles of exposed bias / RoPE-table / residual / scale round trips behind it)
template <int MI> struct EpiPre { f32x4 bias[4]; f32x4 res[MI][4]; f32x4 rc[MI][2], rs[MI][2]; float cs, vs; };

// pins the switches of a specialised epilogue (see EPI_* in gemm_common.h)
template <int EPI>
__device__ __forceinline__ void epi_specialise(cvx_gemm_args& p, SplitOut& so)
{
    if constexpr (EPI == EPI_QKV) {
        p.act = CVX_ACT_NONE; p.bias = nullptr; p.residual = nullptr; so.write_f32 = 0;
    } else if constexpr (EPI == EPI_QKV_RS) {
        p.act = CVX_ACT_NONE; p.residual = nullptr; so.write_f32 = 0;
    } else if constexpr (EPI == EPI_RES) {
        p.act = CVX_ACT_NONE; p.rope_cos = nullptr; so.write_f32 = 1;
    } else if constexpr (EPI == EPI_RES_TW) {
        p.act = CVX_ACT_NONE; p.rope_cos = nullptr;
    } else if constexpr (EPI == EPI_GELU_SPLIT || EPI == EPI_GELU_RS) {
        p.act = CVX_ACT_GELU; p.rope_cos = nullptr; p.residual = nullptr; so.write_f32 = 0;
    } else if constexpr (EPI == EPI_BIAS) {
        p.act = CVX_ACT_NONE; p.rope_cos = nullptr; p.residual = nullptr; so.write_f32 = 1; so.hi = nullptr; so.lo = nullptr;
    } else if constexpr (EPI == EPI_BIAS_TW) {
        p.act = CVX_ACT_NONE; p.rope_cos = nullptr; p.residual = nullptr;
    }
    if constexpr (EPI != EPI_RES_TW) so.res_hi = nullptr;
    if constexpr (EPI != EPI_RES_TW && EPI != EPI_BIAS_TW) { so.tw_gamma = nullptr; so.rowsq = nullptr; }
    if constexpr (EPI != EPI_GELU_RS && EPI != EPI_QKV_RS) so.row_scale = nullptr;
}

template <int EPI, int MI, bool PERM = false>
__device__ __forceinline__ void epilogue_prefetch(const cvx_gemm_args& p_in, const SplitOut& so_in, int row0, int col0, int lane,
                                                  bool v_block, EpiPre<MI>& pre)
{
    cvx_gemm_args p = p_in;
    SplitOut so = so_in;
    epi_specialise<EPI>(p, so);
    pre.cs = (so.hi && so.c_scale) ? *so.c_scale : 1.f;
    pre.vs = so.vt_scale ? *so.vt_scale : 1.f;
    if (col0 >= p.N || v_block) return;
    const int lr = lane & 15, lc = 4 * (lane >> 4);
    const bool do_rope = (p.rope_cos != nullptr) && (col0 < p.rope_cols);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
        if (p.bias && col0 + 16 * ni < p.N) pre.bias[ni] = *reinterpret_cast<const f32x4*>(p.bias + col0 + tile_col<PERM>(ni, lc));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int rr = min(row0 + 16 * mi + lr, p.M - 1);
        if (do_rope) {
            const int pos = rr % p.rope_T;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                pre.rc[mi][ni] = *reinterpret_cast<const f32x4*>(p.rope_cos + (int64_t)pos * 32 + tile_col<PERM>(ni, lc));
                pre.rs[mi][ni] = *reinterpret_cast<const f32x4*>(p.rope_sin + (int64_t)pos * 32 + tile_col<PERM>(ni, lc));
            }
        }
        if (p.residual) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                if (col0 + 16 * ni < p.N) pre.res[mi][ni] = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)rr * p.ldr + col0 + tile_col<PERM>(ni, lc));
        }
    }
}

// ---- epilogue, swapped layout: acc[mi][ni][r] = C[row0 + 16 mi + (lane & 15)][col0 + 16 ni + 4 (lane >> 4) + r]
// PERMSEL: -1 = the large-problem kernel's rule (the deferred-norm instances are permuted), 0 / 1 = the caller's W tile is not / is
template <int EPI, int MI = 8, bool PRE = false, int PERMSEL = -1>
__device__ __forceinline__ void epilogue_rows(const cvx_gemm_args& p_in, f32x4 (&acc)[MI][4], int row0, int col0, int lane,
                                              const SplitOut& so_in, float acc_scale, const EpiPre<MI>* pre = nullptr)
{
    if (col0 >= p_in.N) return;                 // wave tile entirely past the last column (N % 256 != 0; wave-uniform)
    cvx_gemm_args p = p_in;
    SplitOut so = so_in;
    epi_specialise<EPI>(p, so);
    constexpr bool PERM = PERMSEL < 0 ? (!PRE && epi_perm(EPI)) : (PERMSEL != 0);        // a lane's tile pair = 8 consecutive columns
    const int lr = lane & 15, lc = 4 * (lane >> 4);
    const bool do_rope = (p.rope_cos != nullptr) && (col0 < p.rope_cols);          // wave-uniform (64-column wave tile = one head)
    float cs;
    if constexpr (PRE) cs = pre->cs; else cs = (so.hi && so.c_scale) ? *so.c_scale : 1.f;
    const bool il = so.hi && so.lo == so.hi + 32;
    CvxSat amax;
    const f32x2 sc2 = splat2(acc_scale), cs2 = splat2(cs);
    f32x2 bias[4][2];                           // [ni][pair]: columns col0 + 16 ni + lc + 2 pair + {0, 1}
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) { if (p.bias) b4 = pre->bias[ni]; }
        else if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + col0 + tile_col<PERM>(ni, lc));
        bias[ni][0] = f32x2{b4[0], b4[1]}; bias[ni][1] = f32x2{b4[2], b4[3]};
    }
    // residual rows are requested CVX_P8S_RES_AHEAD row groups before they are added (explicitly: where hipcc puts these loads
    // on its own moves with unrelated changes of the epilogue - the saturation bookkeeping cost 0.65 % of the step that way)
    constexpr int RA = CVX_P8S_RES_AHEAD;
    f32x4 rbuf[RA + 1][4];
    // (EPI_RES_TW: the residual may be a split pair - a lane's 4 hi halves travel in r[ni][0..1], its 4 lo halves in r[ni][2..3])
    const bool res_pair = (EPI == EPI_RES_TW) && so.res_hi != nullptr;
    const bool res_il = res_pair && so.res_lo == so.res_hi + 32;
    const bool has_res = p.residual != nullptr || res_pair;
    float res_inv = 1.f;
    if (res_pair && so.res_scale) res_inv = 1.f / *so.res_scale;           // (a power of two: exact)
    auto load_res = [&](int mi_, f32x4 (&r)[4]) {
        const int row_ = row0 + 16 * mi_ + lr;
        const int rr_ = row_ < p.M ? row_ : p.M - 1;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if constexpr (PRE) r[ni] = pre->res[mi_][ni];
            else if (res_pair) {
                const int c_ = col0 + tile_col<PERM>(ni, lc);
                const int64_t o_ = (int64_t)rr_ * so.res_ld + (res_il ? il_col(c_) : c_);
                if constexpr (PERM) {        // the tile pair's 8 columns: one 16-byte load of hi halves (r[even]), one of lo halves (r[odd])
                    if (ni & 1) r[ni] = *reinterpret_cast<const f32x4*>(so.res_lo + o_ - 4);
                    else r[ni] = *reinterpret_cast<const f32x4*>(so.res_hi + o_);
                } else {
                    const f32x2 h_ = *reinterpret_cast<const f32x2*>(so.res_hi + o_), l_ = *reinterpret_cast<const f32x2*>(so.res_lo + o_);
                    r[ni] = f32x4{h_[0], h_[1], l_[0], l_[1]};
                }
            }
            else r[ni] = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)rr_ * p.ldr + col0 + tile_col<PERM>(ni, lc));
        }
    };
    if (RA > 0 && has_res) {
#pragma unroll
        for (int a = 0; a < RA; ++a) load_res(a, rbuf[a]);
    }
    constexpr bool RS = (EPI == EPI_GELU_RS || EPI == EPI_QKV_RS);          // consumer of a deferred norm: one factor per row
    constexpr bool TW = (EPI == EPI_RES_TW || EPI == EPI_BIAS_TW);          // producer: gamma on the twin, row sums of squares
    float rscale[RS ? MI : 1];
    if constexpr (RS) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) rscale[mi] = so.row_scale[min(row0 + 16 * mi + lr, p.M - 1)];
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = row0 + 16 * mi + lr;
        const bool live = row < p.M;
        const int rr = live ? row : p.M - 1;
        if (RA > 0 && has_res && mi + RA < MI) load_res(mi + RA, rbuf[(mi + RA) % (RA + 1)]);
        f32x2 v[4][2];
        f32x2 scr = sc2;
        if constexpr (RS) scr = splat2(acc_scale * rscale[mi]);      // deferred norm: sqrt(D) / ||x_row|| rides on the accumulator scale
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 x = fma2(f32x2{acc[mi][ni][2 * h], acc[mi][ni][2 * h + 1]}, scr, bias[ni][h]);
                if (p.act == CVX_ACT_GELU) x = gelu_fast2(x);
                else if (p.act == CVX_ACT_SILU) x = silu2(x);
                v[ni][h] = x;
            }
        }
        if (do_rope) {      // half-split rotation: column j of the head pairs with j + 32 = tile ni + 2, same lane, same register
            const int pos = rr % p.rope_T;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                f32x4 c, s;
                if constexpr (PRE) { c = pre->rc[mi][ni]; s = pre->rs[mi][ni]; }
                else {
                    c = *reinterpret_cast<const f32x4*>(p.rope_cos + (int64_t)pos * 32 + tile_col<PERM>(ni, lc));
                    s = *reinterpret_cast<const f32x4*>(p.rope_sin + (int64_t)pos * 32 + tile_col<PERM>(ni, lc));
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 c2 = f32x2{c[2 * h], c[2 * h + 1]}, s2 = f32x2{s[2 * h], s[2 * h + 1]};
                    const f32x2 lo = v[ni][h], hi = v[ni + 2][h];
                    v[ni][h] = fma2(lo, c2, -mul2_rn(hi, s2));          // fixed contraction, as in the 32x32 epilogue
                    v[ni + 2][h] = fma2(hi, c2, mul2_rn(lo, s2));
                }
            }
        }
        if (res_pair) {
            static_assert(RA > 0 || EPI != EPI_RES_TW, "the pair residual is read through the look-ahead buffers");
            const f32x2 ri2 = splat2(res_inv);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 r = rbuf[mi % (RA + 1)][ni];
                if constexpr (PERM) {        // r[even] = 8 hi halves, r[odd] = 8 lo halves of the pair's columns: tile ni takes 4 of each
                    const f32x4 rh = rbuf[mi % (RA + 1)][ni & ~1], rl = rbuf[mi % (RA + 1)][ni | 1];
                    r = (ni & 1) ? f32x4{rh[2], rh[3], rl[2], rl[3]} : f32x4{rh[0], rh[1], rl[0], rl[1]};
                }
                const f16x4 h4 = __builtin_bit_cast(f16x4, f32x2{r[0], r[1]}), l4 = __builtin_bit_cast(f16x4, f32x2{r[2], r[3]});
                // hi + lo is exact in fp32 (two 11-bit significands at most 2^11 apart), the un-scaling a power of two
                v[ni][0] = fma2(f32x2{(float)h4[0] + (float)l4[0], (float)h4[1] + (float)l4[1]}, ri2, v[ni][0]);
                v[ni][1] = fma2(f32x2{(float)h4[2] + (float)l4[2], (float)h4[3] + (float)l4[3]}, ri2, v[ni][1]);
            }
        } else if (p.residual) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const f32x4 r = RA > 0 ? rbuf[mi % (RA + 1)][ni]
                                       : *reinterpret_cast<const f32x4*>(p.residual + (int64_t)rr * p.ldr + col0 + tile_col<PERM>(ni, lc));
                v[ni][0] += f32x2{r[0], r[1]};
                v[ni][1] += f32x2{r[2], r[3]};
            }
        }
        if (TW && so.rowsq) {    // sum of squares of the row's 64 columns of this wave tile: 16 in-lane, then the 4 lanes that share the row
            f32x2 q2 = v[0][0] * v[0][0];
            q2 = fma2(v[0][1], v[0][1], q2);
#pragma unroll
            for (int ni = 1; ni < 4; ++ni) { q2 = fma2(v[ni][0], v[ni][0], q2); q2 = fma2(v[ni][1], v[ni][1], q2); }
            float q = q2[0] + q2[1];
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (live && lc == 0) so.rowsq[(int64_t)row * so.rowsq_ld + (col0 >> 6)] = q;
        }
        if (!live) continue;
        if (so.write_f32) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if constexpr (PRE) { if (col0 + 16 * ni >= p.N) continue; }      // (N % 16 == 0 on the medium-problem kernel: partial wave tile)
                *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col0 + tile_col<PERM>(ni, lc)) = f32x4{v[ni][0][0], v[ni][0][1], v[ni][1][0], v[ni][1][1]};
            }
        }
        if (so.hi) {
            f16x4 ph = f16x4{0, 0, 0, 0}, pl = f16x4{0, 0, 0, 0};
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if constexpr (PRE) { if (col0 + 16 * ni >= p.N) continue; }
                const int c = col0 + tile_col<PERM>(ni, lc);
                const int64_t o = (int64_t)row * so.ldc_h + (il ? il_col(c) : c);
                f16x2 h01, h23, l01, l23;
                f32x2 g01 = cs2, g23 = cs2;
                if (TW && so.tw_gamma) {     // (re-read per row group - an L1 hit - rather than 16 registers held across the epilogue)
                    const float* gp = so.tw_gamma + c;
                    asm volatile("" : "+v"(gp));
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp);
                    g01 = f32x2{g4[0], g4[1]} * cs2; g23 = f32x2{g4[2], g4[3]} * cs2;
                }
                split2_pk(v[ni][0] * g01, h01, l01, amax);
                split2_pk(v[ni][1] * g23, h23, l23, amax);
                if constexpr (PERM) {        // tile 2t parks its halves, tile 2t + 1 stores the pair's 8 consecutive columns: 16 bytes of hi, 16 of lo
                    if (ni & 1) {
                        *reinterpret_cast<f16x8*>(so.hi + o - 4) = f16x8{ph[0], ph[1], ph[2], ph[3], h01[0], h01[1], h23[0], h23[1]};
                        if (so.lo) *reinterpret_cast<f16x8*>(so.lo + o - 4) = f16x8{pl[0], pl[1], pl[2], pl[3], l01[0], l01[1], l23[0], l23[1]};
                    } else {
                        ph = f16x4{h01[0], h01[1], h23[0], h23[1]}; pl = f16x4{l01[0], l01[1], l23[0], l23[1]};
                    }
                } else {
                    *reinterpret_cast<f16x4*>(so.hi + o) = f16x4{h01[0], h01[1], h23[0], h23[1]};
                    if (so.lo) *reinterpret_cast<f16x4*>(so.lo + o) = f16x4{l01[0], l01[1], l23[0], l23[1]};
                }
            }
        }
    }
    cvx_sat_commit(so.sat, amax);
}

// ---- epilogue of a V block of a to_qkv projection, UN-swapped layout:
// acc[mi][ni][r] = C[row0 + 16 mi + 4 (lane >> 4) + r][col0 + 16 ni + (lane & 15)]: 4 consecutive frames per lane ->
// vt[((b*H + head)*64 + d) * vt_ld + slot(t)], 8 bytes per store when the four frames are one aligned slot group
template <int MI = 8, bool PRE = false, bool RS = false, bool PERM = false>
__device__ __forceinline__ void epilogue_vt(const cvx_gemm_args& p, f32x4 (&acc)[MI][4], int row0, int col0, int lane,
                                            const SplitOut& so, float acc_scale, const EpiPre<MI>* pre = nullptr)
{
    if (col0 >= p.N) return;                    // wave tile entirely past the last column (H % 4 != 0; wave-uniform)
    const int H = p.rope_cols / 128, T = p.rope_T;
    const int head = (col0 - p.rope_cols) / 64;
    float vs0;
    if constexpr (PRE) vs0 = pre->vs; else vs0 = so.vt_scale ? *so.vt_scale : 1.f;
    const float vs = vs0 * acc_scale;
    CvxSat amax;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r0 = row0 + 16 * mi + 4 * (lane >> 4);
        if (r0 >= p.M) continue;
        const int b = r0 / T, t0 = r0 - b * T;
        const bool vec = (t0 & 3) == 0 && t0 + 3 < T && r0 + 3 < p.M;
        f32x4 rs4 = f32x4{vs, vs, vs, vs};          // deferred norm: the four frames' sqrt(D) / ||x|| on the accumulator scale
        if constexpr (RS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) rs4[e] = vs * so.row_scale[min(r0 + e, p.M - 1)];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int d = PERM ? perm32(16 * ni + (lane & 15)) : 16 * ni + (lane & 15);      // (the W tile's rows may be permuted inside 32)
            const float bv = p.bias ? p.bias[col0 + d] * vs0 : 0.f;
            f16x4 h, l;
            split4_pk(f32x4{fmaf(acc[mi][ni][0], rs4[0], bv), fmaf(acc[mi][ni][1], rs4[1], bv), fmaf(acc[mi][ni][2], rs4[2], bv), fmaf(acc[mi][ni][3], rs4[3], bv)}, h, l, amax);
            if (vec) {
                const int64_t o = ((int64_t)(b * H + head) * 64 + d) * so.vt_ld + vt_slot(t0);
                *reinterpret_cast<f16x4*>(so.vt_hi + o) = h;
                if (so.vt_lo) *reinterpret_cast<f16x4*>(so.vt_lo + o) = l;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = r0 + e;
                    if (row >= p.M) break;
                    const int bb = row / T, tt = row - bb * T;
                    const int64_t o = ((int64_t)(bb * H + head) * 64 + d) * so.vt_ld + vt_slot(tt);
                    so.vt_hi[o] = h[e];
                    if (so.vt_lo) so.vt_lo[o] = l[e];
                }
            }
        }
    }
    cvx_sat_commit(so.sat, amax);
}

}  // namespace
