// Shared pieces of the GEMM kernels: XCD-aware block->tile map and the fused epilogue.
#pragma once
#include "cvx_common.h"

namespace cvxg {

constexpr int BN = 128;
constexpr int BK = 32;

__device__ __forceinline__ void tile_of_block(int tiles_m, int tiles_n, int map_mode, int& tile_m, int& tile_n)
{
    // Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8: observed, used for speed only).
    // mode 1: XCD x owns row-panels x, x+8, ... (all their N tiles): every A panel is read by one L2 only
    // and just the (small) W matrix is read by all eight.
    if (map_mode == 1) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile_m = xcd + 8 * (slot / tiles_n);
        tile_n = slot % tiles_n;
    } else {
        tile_n = blockIdx.x % tiles_n;
        tile_m = blockIdx.x / tiles_n;
    }
}

// acc[mi][ni]: 32x32 MFMA accumulators of a wave that owns rows [m0 + wm*TM*32, +TM*32) and the 64 columns
// [n0 + wn*64, +64).  bias -> act -> half-split RoPE -> residual -> store.
// Optional split copy of the output: (fp16 hi, fp16 lo) of the final value, row stride ldc_h (halves), for a
// consumer GEMM that takes its A operand pre-split; write_f32 == 0 suppresses the fp32 store.  lo == NULL (and
// vt_lo == NULL) writes the hi halves only - the single-term fp16 mode.
// QKV mode (vt_hi != NULL; needs the RoPE arguments: rope_T = frames per sequence, rope_cols = 2*H*64): columns
// [0, rope_cols) (q | k, after RoPE) go to hi/lo [M, ldc_h] as usual, columns >= rope_cols (v) are written TRANSPOSED
// per (sequence, head): vt[((b*H + h)*64 + d) * vt_ld + slot(t)] - the layout the f16x3 attention kernel DMAs its
// V^T tiles from.  slot(t) swaps bits 2 and 3 of t: inside every block of 16 frames the four-frame groups are stored
// in the order 0, 2, 1, 3, so that the eight keys one lane of the attention kernel's O^T += V^T.P^T MFMA needs
// (keys 16s+4g+{0..3} and 16s+8+4g+{0..3}, the order its P registers already have) are ONE 16-byte LDS read.
// Activation scales (DEVICE pointers to one float each, NULL = 1): power-of-two factors that keep the split pairs inside
// fp16's full-precision window whatever the magnitude of the tensor - a_scale is what the PRODUCER of the A (and A2)
// pair multiplied it by (the accumulators are divided by it, exactly), c_scale / vt_scale multiply the values written to
// hi/lo and vt_hi/vt_lo (the fp32 store of C is never scaled).
struct SplitOut { _Float16* hi; _Float16* lo; int64_t ldc_h; int write_f32;
                  _Float16* vt_hi; _Float16* vt_lo; int64_t vt_ld;
                  const float* c_scale; const float* vt_scale; const float* a_scale;
                  int dbg; unsigned long long* trace;
                  uint32_t* sat;
                  // deferred AdaptiveRMSNorm (round 4; the 16x16x32 epilogues of gemm_p8s_epi.h only - the launchers refuse it elsewhere):
                  //   producer: the split twin holds C[m,n] * tw_gamma[n] (* c_scale) instead of C, and every 64-column wave tile leaves
                  //             sum_n C[m,n]^2 over its columns in rowsq[m * rowsq_ld + n0 / 64] (fixed order: deterministic);
                  //   consumer: row m of the accumulators is multiplied by row_scale[m] (= sqrt(D) / ||x_m||) before bias / RoPE.
                  const float* tw_gamma; float* rowsq; int rowsq_ld; const float* row_scale;
                  // residual given as a split pair (EPI_RES_TW only): residual[m,n] = (res_hi + res_lo)[m,n] / *res_scale; the residual
                  // stream then lives in HBM as pairs only (write_f32 = 0 on the producer: as many bytes as the fp32 form moved)
                  const _Float16* res_hi; const _Float16* res_lo; int64_t res_ld; const float* res_scale;
                  // K-split operand pairs with DIFFERENT pre-scales (EPI_BIAS_TW on the large-problem kernel only): A holds x * *a_scale,
                  // A2 holds x2 * *a2_scale; the accumulators are multiplied by *a2_scale / *a_scale (a power of two) where the K loop
                  // switches operands, and divided by *a2_scale at the end
                  const float* a2_scale; };      // sticky saturation flag of the device (cvx_common.h), NULL = no bookkeeping   // trace (dbg bit 2): per-block s_memtime stamps (dev only)          // dbg: timing experiments only (bit 0: skip the epilogue) - cvx_gemm_split_io.flags >> 8

// accumulator factor: 1 / (weight pre-scale) / (activation pre-scale); both powers of two, so the division is exact
__device__ __forceinline__ float total_acc_scale(float acc_scale, const SplitOut& so)
{
    const float* s = so.a2_scale ? so.a2_scale : so.a_scale;
    return s ? acc_scale / *s : acc_scale;
}
typedef _Float16 cvx_f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int vt_slot(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }

// column -> position inside a row of an INTERLEAVED split pair ([hi 32 | lo 32] per block of 32 columns; lo == hi + 32)
__device__ __forceinline__ int il_col(int c) { return ((c >> 5) << 6) | (c & 31); }

// (amax: the caller's running max |v| for the saturation flag - committed once per row block, not per element: an atomic
//  inside the scalar path's 16 x TM unrolled stores made the compiler give up unrolling and index the accumulators from scratch)
__device__ __forceinline__ void store_split(const SplitOut& so, int64_t idx, float v, float& amax)
{
    const float x = fminf(fmaxf(v, -65504.f), 65504.f);
    amax = cvx_amax3_c(amax, v, v);
    const _Float16 h = (_Float16)x;
    so.hi[idx] = h;
    if (so.lo) so.lo[idx] = (_Float16)(x - (float)h);      // lo == NULL: single-term fp16 consumer
}

// 4 x 4 transpose inside every lane quad: lane q (= lane & 3) enters with v[e] = M[e][q] and leaves with v[j] = M[q][j].
// Two butterfly stages (xor 1, xor 2); the shuffles lower to DPP quad permutes.
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3, int lane)
{
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    float x = b0 ? v0 : v1, y = b0 ? v2 : v3;
    float rx = __shfl_xor(x, 1, 64), ry = __shfl_xor(y, 1, 64);
    const float a0 = b0 ? rx : v0, a1 = b0 ? v1 : rx, a2 = b0 ? ry : v2, a3 = b0 ? v3 : ry;
    x = b1 ? a0 : a2; y = b1 ? a1 : a3;
    rx = __shfl_xor(x, 2, 64); ry = __shfl_xor(y, 2, 64);
    v0 = b1 ? rx : a0; v1 = b1 ? ry : a1; v2 = b1 ? a2 : rx; v3 = b1 ? a3 : ry;
}

// EPI: compile-time specialisation of the epilogue for the five call patterns of the transformer block.  The generic
// epilogue carries every combination (activation kinds, RoPE, residual, fp32 / split / transposed stores, vector and
// scalar paths) as straight-line code - a few hundred KiB that every wave walks through once per tile, far more than the
// instruction cache holds; a specialised instance pins the switches to constants (and relies on the launcher having
// checked that the 16-byte vector path applies), so the compiler drops the rest.
enum : int { EPI_GENERIC = 0, EPI_QKV = 1,      // RoPE on q|k, split q|k + transposed split v, no fp32 store, no bias
             EPI_RES = 2,                       // (+ bias) + residual, fp32 store (+ optional split twin): to_out, ff2
             EPI_GELU_SPLIT = 3,                // bias + GELU, split store only: ff1
             EPI_BIAS = 4,                      // bias, fp32 store: skip combiners
             // deferred AdaptiveRMSNorm (SplitOut::tw_gamma / rowsq / row_scale; 16x16x32 epilogues of gemm_p8s_epi.h only):
             EPI_RES_TW = 5,                    // residual (fp32 or a split pair) + optional fp32 store + the split twin (times gamma[n], if given) + row sums of squares: to_out, ff2
             EPI_BIAS_TW = 6,                   // EPI_BIAS + the same twin / sums: skip combiners in front of an attention norm
             EPI_GELU_RS = 7,                   // EPI_GELU_SPLIT with a factor per row on the accumulators: ff1 behind a deferred norm
             EPI_QKV_RS = 8 };                  // EPI_QKV with that factor and a bias (beta . W^T) in front of the rotation

template <int TM, int EPI = EPI_GENERIC>
__device__ __forceinline__ void gemm_epilogue(const cvx_gemm_args& p_in, f32x16 (&acc)[TM][2], int m0, int n0,
                                              int wm, int wn, int lane,
                                              const SplitOut so_in = SplitOut{nullptr, nullptr, 0, 1, nullptr, nullptr, 0})
{
    cvx_gemm_args p = p_in;
    SplitOut so = so_in;
    if constexpr (EPI == EPI_QKV) {
        p.act = CVX_ACT_NONE; p.bias = nullptr; p.residual = nullptr; so.write_f32 = 0;
        __builtin_assume(p.rope_cos != nullptr); __builtin_assume(so.vt_hi != nullptr); __builtin_assume(so.hi != nullptr);
    } else if constexpr (EPI == EPI_RES) {
        p.act = CVX_ACT_NONE; p.rope_cos = nullptr; so.write_f32 = 1; so.vt_hi = nullptr; so.vt_lo = nullptr;
        __builtin_assume(p.residual != nullptr);
    } else if constexpr (EPI == EPI_GELU_SPLIT) {
        p.act = CVX_ACT_GELU; p.rope_cos = nullptr; p.residual = nullptr; so.write_f32 = 0; so.vt_hi = nullptr; so.vt_lo = nullptr;
        __builtin_assume(p.bias != nullptr); __builtin_assume(so.hi != nullptr);
    } else if constexpr (EPI == EPI_BIAS) {
        p.act = CVX_ACT_NONE; p.rope_cos = nullptr; p.residual = nullptr; so.write_f32 = 1; so.vt_hi = nullptr; so.vt_lo = nullptr;
        so.hi = nullptr; so.lo = nullptr;
    }
    const int colw = n0 + wn * 64;                 // first column of this wave (multiple of 64)
    if (colw >= p.N) return;                       // wave entirely past the last column (256-wide tiles, N % 256 != 0; wave-uniform)
    const int c_lo = colw + (lane & 31);
    const int c_hi = c_lo + 32;
    if (so.vt_hi != nullptr && colw >= p.rope_cols) {
        if (colw >= p.N) return;                   // wave entirely past the last column (N not a multiple of 128)
        // V columns of a QKV projection: transposed split store.  Accumulator registers 4*rg .. 4*rg+3 are four
        // consecutive rows (= frames), so they pack into one 8-byte store along t.
        const int H = p.rope_cols / 128, T = p.rope_T;
        const int head = (colw - p.rope_cols) / 64;
        const float b_lo_v = (p.bias && c_lo < p.N) ? p.bias[c_lo] : 0.f;
        const float b_hi_v = (p.bias && c_hi < p.N) ? p.bias[c_hi] : 0.f;
        const float vs = so.vt_scale ? *so.vt_scale : 1.f;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row0 = m0 + wm * TM * 32 + mi * 32 + 8 * rg + 4 * (lane >> 5);
                if (row0 >= p.M) continue;
                const int b = row0 / T, t0 = row0 - b * T;
#pragma unroll
                for (int hsel = 0; hsel < 2; ++hsel) {
                    const int d = (lane & 31) + 32 * hsel;
                    const float bias = hsel ? b_hi_v : b_lo_v;
                    cvx_f16x4 vh, vl;
                    float amax = 0.f;               // committed per store group: no value kept live across the unrolled epilogue
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xr = (acc[mi][hsel][4 * rg + e] + bias) * vs;
                        amax = cvx_amax3_c(amax, xr, xr);
                        const float x = fminf(fmaxf(xr, -65504.f), 65504.f);
                        vh[e] = (_Float16)x;
                        vl[e] = (_Float16)(x - (float)vh[e]);
                    }
                    cvx_sat_commit(so.sat, amax);
                    const int64_t base = ((int64_t)(b * H + head) * 64 + d) * so.vt_ld;
                    if ((t0 & 3) == 0 && t0 + 3 < T && row0 + 3 < p.M) {
                        *reinterpret_cast<cvx_f16x4*>(so.vt_hi + base + vt_slot(t0)) = vh;
                        if (so.vt_lo) *reinterpret_cast<cvx_f16x4*>(so.vt_lo + base + vt_slot(t0)) = vl;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = row0 + e;
                            if (row >= p.M) break;
                            const int bb = row / T, tt = row - bb * T;
                            const int64_t idx = ((int64_t)(bb * H + head) * 64 + d) * so.vt_ld + vt_slot(tt);
                            so.vt_hi[idx] = vh[e];
                            if (so.vt_lo) so.vt_lo[idx] = vl[e];
                        }
                    }
                }
            }
        }
        return;
    }
    const bool do_rope = (p.rope_cos != nullptr) && (colw < p.rope_cols);   // wave-uniform
    const float cs = (so.hi && so.c_scale) ? *so.c_scale : 1.f;
    const float b_lo = (p.bias && c_lo < p.N) ? p.bias[c_lo] : 0.f;
    const float b_hi = (p.bias && c_hi < p.N) ? p.bias[c_hi] : 0.f;

    // ---- vector path (wave-uniform choice): the MFMA layout gives a lane ONE column of 4 consecutive rows per
    // register group; a 4x4 quad transpose turns that into 4 consecutive COLUMNS of one row, so every store /
    // residual load is 16 bytes (8 bytes per fp16 half) instead of 4 (2): 4x fewer memory instructions.
    const bool vec_ok = (EPI != EPI_GENERIC) || ((colw + 64 <= p.N) &&
                        (!so.write_f32 || (((uintptr_t)p.C & 15) == 0 && (p.ldc & 3) == 0)) &&
                        (!p.residual || (((uintptr_t)p.residual & 15) == 0 && (p.ldr & 3) == 0)) &&
                        (!so.hi || ((((uintptr_t)so.hi | (uintptr_t)so.lo) & 7) == 0 && (so.ldc_h & 3) == 0)));
    if (vec_ok) {
        const int q = lane & 3;
        const int c4_lo = colw + 4 * ((lane & 31) >> 2);          // this lane's 4 columns after the transpose
        // residual vectors are requested two store groups ahead, unpredicated (row clamped): a load in front of every group's
        // add was an exposed round trip per group (the same change as in the eight-phase kernel's epilogue)
        constexpr int RA = 2, NG = TM * 4;
        f32x4 rb[RA + 1][2];
        auto load_res = [&](int gi, f32x4 (&r)[2]) {
            const int row_ = min(m0 + wm * TM * 32 + (gi >> 2) * 32 + 8 * (gi & 3) + 4 * (lane >> 5) + q, p.M - 1);
            const float* rp = p.residual + (int64_t)row_ * p.ldr + c4_lo;
            r[0] = *reinterpret_cast<const f32x4*>(rp);
            r[1] = *reinterpret_cast<const f32x4*>(rp + 32);
        };
        if (p.residual) {
#pragma unroll
            for (int a = 0; a < RA && a < NG; ++a) load_res(a, rb[a]);
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int gi = mi * 4 + rg;
                if (p.residual && gi + RA < NG) load_res(gi + RA, rb[(gi + RA) % (RA + 1)]);
                const int row0 = m0 + wm * TM * 32 + mi * 32 + 8 * rg + 4 * (lane >> 5);
                float lo[4], hi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = acc[mi][0][4 * rg + e] + b_lo;
                    hi[e] = acc[mi][1][4 * rg + e] + b_hi;
                    if (p.act == CVX_ACT_GELU) { lo[e] = gelu_erf(lo[e]); hi[e] = gelu_erf(hi[e]); }
                    else if (p.act == CVX_ACT_SILU) { lo[e] = silu(lo[e]); hi[e] = silu(hi[e]); }
                    if (do_rope) {
                        const int pos = min(row0 + e, p.M - 1) % p.rope_T;
                        const float c = p.rope_cos[pos * 32 + (lane & 31)];
                        const float s = p.rope_sin[pos * 32 + (lane & 31)];
                        // fixed contraction (one rounding of the second product, then one fma): with the compiler free to
                        // pick a different fma pairing per unrolled instance, the same row gave 1-ulp different results
                        // depending on its position in the tile
                        const float nlo = __builtin_fmaf(lo[e], c, -__fmul_rn(hi[e], s));
                        const float nhi = __builtin_fmaf(hi[e], c, __fmul_rn(lo[e], s));
                        lo[e] = nlo; hi[e] = nhi;
                    }
                }
                quad_transpose(lo[0], lo[1], lo[2], lo[3], lane);
                quad_transpose(hi[0], hi[1], hi[2], hi[3], lane);
                const int row = row0 + q;
                if (row >= p.M) continue;
                f32x4 vlo = {lo[0], lo[1], lo[2], lo[3]}, vhi = {hi[0], hi[1], hi[2], hi[3]};
                if (p.residual) {
                    const f32x4 r0 = rb[gi % (RA + 1)][0], r1 = rb[gi % (RA + 1)][1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vlo[e] += r0[e]; vhi[e] += r1[e]; }
                }
                if (so.write_f32) {
                    float* cp = p.C + (int64_t)row * p.ldc + c4_lo;
                    *reinterpret_cast<f32x4*>(cp) = vlo;
                    *reinterpret_cast<f32x4*>(cp + 32) = vhi;
                }
                if (so.hi) {
                    const bool il = so.lo == so.hi + 32;
                    const int64_t o = (int64_t)row * so.ldc_h + (il ? il_col(c4_lo) : c4_lo);
                    const int64_t o32 = il ? 64 : 32;              // the partner columns c4_lo + 32 are the next block
                    cvx_f16x4 h0, l0, h1, l1;
                    float amax = 0.f;               // committed per store group: no value kept live across the unrolled epilogue
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s0 = vlo[e] * cs, s1 = vhi[e] * cs;
                        amax = cvx_amax3_c(amax, s0, s1);
                        const float x0 = fminf(fmaxf(s0, -65504.f), 65504.f);
                        const float x1 = fminf(fmaxf(s1, -65504.f), 65504.f);
                        h0[e] = (_Float16)x0; l0[e] = (_Float16)(x0 - (float)h0[e]);
                        h1[e] = (_Float16)x1; l1[e] = (_Float16)(x1 - (float)h1[e]);
                    }
                    cvx_sat_commit(so.sat, amax);
                    *reinterpret_cast<cvx_f16x4*>(so.hi + o) = h0;
                    *reinterpret_cast<cvx_f16x4*>(so.hi + o + o32) = h1;
                    if (so.lo) {
                        *reinterpret_cast<cvx_f16x4*>(so.lo + o) = l0;
                        *reinterpret_cast<cvx_f16x4*>(so.lo + o + o32) = l1;
                    }
                }
            }
        }
        return;
    }
    float amax_s = 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * TM * 32 + mi * 32 + mfma32_row(r, lane);
            if (row >= p.M) continue;
            float lo = acc[mi][0][r] + b_lo;
            float hi = acc[mi][1][r] + b_hi;
            if (p.act == CVX_ACT_GELU) { lo = gelu_erf(lo); hi = gelu_erf(hi); }
            else if (p.act == CVX_ACT_SILU) { lo = silu(lo); hi = silu(hi); }
            if (do_rope) {
                const int pos = row % p.rope_T;
                const float c = p.rope_cos[pos * 32 + (lane & 31)];
                const float s = p.rope_sin[pos * 32 + (lane & 31)];
                const float nlo = __builtin_fmaf(lo, c, -__fmul_rn(hi, s));      // same fixed contraction as the vector path
                const float nhi = __builtin_fmaf(hi, c, __fmul_rn(lo, s));
                lo = nlo; hi = nhi;
            }
            if (p.residual) {
                if (c_lo < p.N) lo += p.residual[(int64_t)row * p.ldr + c_lo];
                if (c_hi < p.N) hi += p.residual[(int64_t)row * p.ldr + c_hi];
            }
            if (so.write_f32) {
                if (c_lo < p.N) p.C[(int64_t)row * p.ldc + c_lo] = lo;
                if (c_hi < p.N) p.C[(int64_t)row * p.ldc + c_hi] = hi;
            }
            if (so.hi) {
                const bool il = so.lo == so.hi + 32;
                if (c_lo < p.N) store_split(so, (int64_t)row * so.ldc_h + (il ? il_col(c_lo) : c_lo), lo * cs, amax_s);
                if (c_hi < p.N) store_split(so, (int64_t)row * so.ldc_h + (il ? il_col(c_hi) : c_hi), hi * cs, amax_s);
            }
        }
    }
    cvx_sat_commit(so.sat, amax_s);
}

// global -> LDS DMA of 16 bytes per lane: LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// pre-split A operand of the all-DMA kernels: (hi, lo) fp16 matrices (and the A2 pair of a K-split), row strides in halves
struct PreSplitA { const _Float16* hi; const _Float16* lo; int64_t ld; const _Float16* hi2; const _Float16* lo2; int64_t ld2; };

// gemm_f16x3_p8s.hip: 256 x 256 tile, eight-phase ping-pong main loop on interleaved operands (A and W as [hi 32 | lo 32] lines),
// 16x16x32 MFMA with swapped operands, transpose-free epilogue, persistent blocks; only problems whose epilogue can run 16-byte
// vectors (N % 64 == 0, aligned pointers); false -> the caller falls back to the 128 x 128 kernel
bool launch_gemm_f16x3_p8s(const cvx_gemm_args& a, const PreSplitA& A, const _Float16* w_il, float acc_scale, const SplitOut& so,
                           hipStream_t st, int n_cu);

// gemm_f16x3_p8m.hip: 128 x 128 tiles, two wave groups on alternate K-tiles, for problems of fewer than 2048 rows (interleaved
// A and W); ksplit > 1: K slices on separate blocks, scaled fp32 partial tiles to `partial` [ksplit][M][N] (the caller reduces)
bool launch_gemm_f16x3_p8m(const cvx_gemm_args& a, const PreSplitA& A, const _Float16* w_il, float acc_scale, const SplitOut& so,
                           int ksplit, float* partial, hipStream_t st);

// which specialised epilogue (EPI_*) covers this call; EPI_GENERIC when none does or the vector-path conditions fail
int classify_epilogue(const cvx_gemm_args& a, const SplitOut& so);

int validate_gemm_args(const cvx_gemm_args* a);   // shared argument checks (gemm_f32.hip)

}  // namespace cvxg
