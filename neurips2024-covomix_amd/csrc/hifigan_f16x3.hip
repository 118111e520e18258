// HiFi-GAN ResBlock convolutions on the fp16 matrix pipe with split-precision operands (gfx950).
//
// Reference: covomix/vocoder/models.py:11-48 (ResBlock1: 3 x [lrelu -> Conv1d(k, dil) -> lrelu -> Conv1d(k, 1) -> +x]),
// :104-110 (xs = sum of the resblocks / num_kernels).  97 % of the vocoder's FLOPs are these 72 stride-1 "same"
// convolutions; hifigan_f32.hip runs them (and everything else) on v_mfma_f32_32x32x2_f32 at 47-62 TFLOP/s.  Here
// they use the scheme of gemm_f16x3.hip: every fp32 value is an (fp16 hi, fp16 lo) pair and each product is three
// v_mfma_f32_32x32x16_f16 (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulate) - fp32-class accuracy.
//
// Layout: activations are CHANNELS-LAST, [B][Lp][Cp] with Cp = C rounded up to 32 and HALO_L zero rows in front of
// position 0 (and zero rows behind position L-1), so "same" padding and ragged tiles need no predication: an
// activation tile is a set of contiguous 64-byte rows that go global -> LDS by DMA.
// Implicit GEMM:  out[l, co] = sum_{c, kk}  z[l + kk*dil - pad, c] * W[co, c, kk]
//   M = positions (A operand = activations), N = output channels (B operand = weights), K = 32 input channels per
//   chunk x taps.  Block = 256 positions x all output channels (Np = 32/64/128/256), 8 waves.
//   A tile: the 256 + (k-1)*dil rows of one channel chunk, staged ONCE per chunk and shared by all taps (a tap is a
//           row offset of the fragment reads) - double buffered across chunks;
//   W stage: TS = 256/Np taps of one chunk ([tap][co][32 ci], 64-byte rows), double buffered; one barrier per stage.
// Both use the GEMM's XOR swizzle (16-byte chunk ^ ((row >> 2) & 3)) on the DMA source and on the ds_read_b128.
// Epilogue (rows = positions, so a lane quad transposes 4x4 blocks and stores 4 consecutive channels): bias, ResBlock
// residual, the running xs accumulate / scale, an fp32 store of the new residual stream and / or the split fp16 store
// of leaky_relu(value) = the next convolution's input - no separate activation pass exists.
#include <algorithm>
#include "gemm_common.h"

namespace {

using namespace cvxg;
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int CK = 32;                 // input channels per chunk (one 64-byte row)
constexpr int TMB = 256;               // positions per block
constexpr int A_ROWS = 320;            // TMB + halo (<= 50) rounded up to the 16-row DMA piece
constexpr int A_TILE = A_ROWS * CK;    // halves of one (hi or lo) activation tile: 20 KiB
constexpr int W_TILE = 256 * CK;       // halves of one (hi or lo) weight stage (TS taps x Np rows = 256 rows): 16 KiB
constexpr int LDS_HALVES = 2 * 2 * A_TILE + 2 * 2 * W_TILE;      // 144 KiB
#ifndef CVX_PAIR_PIPE
#define CVX_PAIR_PIPE 1                 // dev A/B: 0 = the plain loop of the fused pair kernel (lgkmcnt(0) in front of every step)
#endif

struct Conv16Args {
    const f16* z_hi; const f16* z_lo;  // [B][Lp][Cp_in]
    const f16* w_hi; const f16* w_lo;  // [chunk][tap][Np][32]
    const float* bias;                 // [Np]
    const float* res;                  // fp32 [B][Lp][Np] or NULL
    const float* accum;                // fp32 [B][Lp][Np] or NULL
    float* out_x;                      // fp32 [B][Lp][Np] or NULL
    f16* out_zhi; f16* out_zlo;        // [B][Lp][Np] or NULL
    int L, Lp, Cp_in, Np, ksize, dil, pad, halo_l;
    float acc_scale, out_scale, z_slope;
    const float* z_scale;              // device scalar (power of two) carried by z and applied to out_z, or NULL = 1
    uint32_t* sat;                     // sticky saturation flag of the device (cvx_common.h) or NULL
    cvx_item_lengths items;            // ragged batch: valid positions per item (zeros are written behind them)
    // Output-column tiles (blockIdx.z), each with its own taps.  A "same" convolution has ONE (ksize, pad, offset 0).
    // ConvTranspose1d in stride-1 form has stride * Np_out output columns - column r * Np_out + co of row m is channel co of
    // output position stride * m + r - and phase r only sees the taps kk = c_r + stride * j of the kernel, so every tile of
    // NP columns (a group of phases) carries its own tap count, its own "pad" and its own packed weights.
    int zk[8], zpad[8];
    long long zw[8];                   // offset of the tile's packed weights, in halves
    // out row l, tile column n  ->  out_x[b * out_bs + out_base + l * ldo + zt * NP + n]; it is position
    // l * ostride + ((zt * NP + n) >> ph_shift) of the signal, stored when < L_out (and < the item's length)
    long long out_bs, out_base;
    int ldo, ostride, ph_shift, L_out;
    uint32_t* amax_out;                // receives the bit pattern of max |out_x| (atomicMax) or NULL
};

// Kernel argument: ONE convolution, or (GRP) up to three INDEPENDENT convolutions of one shape that share a launch - blockIdx.z picks the
// problem.  The three ResBlocks of a generator stage (kernel sizes 3 / 7 / 11) are independent until their results are summed
// (models.py:104-110): launched one by one each of their convolutions is ~2.4 rounds of tiles (one round of 157 at the 250-channel stage
// of the bench shape), launched together the tiles of the short kernels fill the rounds of the long one (problems ordered by
// descending kernel size: longest tiles first).
template <bool GRP> struct ConvKArgs { Conv16Args a; };
template <> struct ConvKArgs<true> { Conv16Args a[3]; };

// ds_read_b128 with an immediate byte offset, issued from asm: the compiler does not count it, the waits are explicit
__device__ __forceinline__ f16x8 lds_read16(uint32_t addr, const int off)
{
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}

template <int TMI, int TNI, int WN, bool GRP = false>
__global__ __launch_bounds__(512) void conv_f16x3_kernel(const ConvKArgs<GRP> P)
{
    const Conv16Args& p = [&]() -> const Conv16Args& { if constexpr (GRP) return P.a[blockIdx.z]; else return P.a; }();
    constexpr int WM = 8 / WN;
    constexpr int NP = WN * TNI * 32;
    constexpr int TS = 256 / NP;                         // taps per weight stage
    constexpr int TMB = WM * TMI * 32;                   // positions per block: 256, or 192 where that fills the chip better
    constexpr int A_ROWS = TMB + 64;                     // + halo (<= 50) rounded up to the 16-row DMA piece
    constexpr int A_TILE = A_ROWS * CK;
    static_assert(TMB <= ::TMB && TMB % 32 == 0, "block tile: at most 256 positions");
    extern __shared__ __attribute__((aligned(16))) f16 smem_c[];
    f16* const As = smem_c;                              // [2 buffers][hi | lo][A_ROWS][32]
    f16* const Ws = smem_c + 4 * A_TILE;                 // [2 stages][hi | lo][256][32]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
#ifdef CVX_CONV_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const float zs = p.z_scale ? *p.z_scale : 1.f;          // activation pre-scale of this stage's split pairs
    const float a_sc = p.acc_scale / zs;                    // (exact: both are powers of two)
    float amax = 0.f;
    const int l0 = blockIdx.x * TMB;
    const int b = blockIdx.y;
    const int zt = GRP ? 0 : (int)blockIdx.z;           // (grouped problems are plain convolutions: one output-column tile each)
    const int ksize = p.zk[zt], pad = p.zpad[zt];
    const f16* const w_hi = p.w_hi + p.zw[zt];
    const f16* const w_lo = p.w_lo + p.zw[zt];
    const int Lb = cvx_item_len(p.items, b, p.L_out);
    const int n_chunks = p.Cp_in / CK;
    const int n_groups = (ksize + TS - 1) / TS;
    const int steps = n_chunks * n_groups;

    // ---- DMA addressing: a piece = 16 rows x 64 bytes; lane -> (row = lane >> 2, 16-byte chunk = lane & 3, swizzled)
    const int prow = lane >> 2;
    const int64_t a_row0 = (int64_t)b * p.Lp + p.halo_l + l0 - pad;            // global row of tile row 0 (>= 0)
    const int64_t last_row = (int64_t)gridDim.y * p.Lp - 1;
    auto issue_a = [&](int chunk) {
        f16* dst = As + (chunk & 1) * 2 * A_TILE;
        for (int pc = wid; pc < A_ROWS / 16; pc += 8) {
            const int r = 16 * pc + prow;
            const int c4 = (lane & 3) ^ ((r >> 2) & 3);
            const int64_t src = min(a_row0 + r, last_row) * p.Cp_in + chunk * CK + 8 * c4;      // (192-row tiles may reach past roundup(L, 256) + 64)
            glds16(p.z_hi + src, dst + 16 * pc * CK);
            glds16(p.z_lo + src, dst + A_TILE + 16 * pc * CK);
        }
    };
    auto issue_w = [&](int st) {
        const int chunk = st / n_groups, grp = st - chunk * n_groups;
        const int t0 = grp * TS, nt = min(TS, ksize - t0);
        f16* dst = Ws + (st & 1) * 2 * W_TILE;
        const int64_t base = ((int64_t)chunk * ksize + t0) * NP * CK;
        for (int pc = wid; pc < nt * NP / 16; pc += 8) {
            const int r = 16 * pc + prow;
            const int c4 = (lane & 3) ^ ((r >> 2) & 3);
            const int64_t src = base + (int64_t)r * CK + 8 * c4;
            glds16(w_hi + src, dst + 16 * pc * CK);
            glds16(w_lo + src, dst + W_TILE + 16 * pc * CK);
        }
    };

    f32x16 acc[TMI][TNI];
#pragma unroll
    for (int mi = 0; mi < TMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int i31 = lane & 31, g = lane >> 5;
    const int wswz = (i31 >> 2) & 3;
    int woff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) woff[s] = (wn * TNI * 32 + i31) * CK + 8 * ((2 * s + g) ^ wswz);
    const int arow_base = wm * TMI * 32 + i31;

    issue_a(0);
    issue_w(0);
    for (int st = 0; st < steps; ++st) {
        const int chunk = st / n_groups, grp = st - chunk * n_groups;
        const int t0 = grp * TS, nt = min(TS, ksize - t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // stage st (and its activation tile) landed; the other buffers are free
        if (st + 1 < steps) {
            issue_w(st + 1);
            if (grp == 0 && chunk + 1 < n_chunks) issue_a(chunk + 1);
        }
        const f16* Ah = As + (chunk & 1) * 2 * A_TILE;
        const f16* Al = Ah + A_TILE;
        const f16* Wh = Ws + (st & 1) * 2 * W_TILE;
        const f16* Wl = Wh + W_TILE;
        for (int tl = 0; tl < nt; ++tl) {
            const int arow = arow_base + (t0 + tl) * p.dil;            // tile row of output row i31 for this tap
            const int aswz = (arow >> 2) & 3;
            const f16* wth = Wh + tl * NP * CK;
            const f16* wtl = Wl + tl * NP * CK;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int aoff = arow * CK + 8 * ((2 * s + g) ^ aswz);
                f16x8 fah[TMI], fal[TMI], fwh[TNI], fwl[TNI];
#pragma unroll
                for (int ni = 0; ni < TNI; ++ni) {
                    fwh[ni] = *reinterpret_cast<const f16x8*>(wth + ni * 32 * CK + woff[s]);
                    fwl[ni] = *reinterpret_cast<const f16x8*>(wtl + ni * 32 * CK + woff[s]);
                }
#pragma unroll
                for (int mi = 0; mi < TMI; ++mi) {
                    fah[mi] = *reinterpret_cast<const f16x8*>(Ah + mi * 32 * CK + aoff);
                    fal[mi] = *reinterpret_cast<const f16x8*>(Al + mi * 32 * CK + aoff);
                }
#pragma unroll
                for (int mi = 0; mi < TMI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[mi], fwh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < TMI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi], fwl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < TMI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[mi], fwh[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }


#ifdef CVX_CONV_TRACE
    const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- epilogue.  acc register 4*rg + e of lane (i31, g): position 8*rg + 4*g + e, channel i31.  After the quad
    // transpose lane q = lane & 3 holds position 8*rg + 4*g + q and the 4 channels 4*(i31 >> 2) .. +3.
    const int q = lane & 3;
    const int c4 = 4 * (i31 >> 2);
    float omax = 0.f;
    // The residual / accumulate rows of one 32-position slice (TNI x 4 vectors each) are requested TOGETHER, ahead of the
    // slice's arithmetic: with a load in front of every store group the epilogue was a chain of 16-24 exposed HBM round
    // trips per wave (tools/archive/conv_trace.py: 22-28 us per block, as long as the k = 3 main loop).  Rows behind L are inside
    // the allocation (Lp >= halo + roundup(L, 256) + 64), so the loads need no predicate; the stores keep theirs.
#pragma unroll
    for (int mi = 0; mi < TMI; ++mi) {
        const int lrow = l0 + wm * TMI * 32 + mi * 32 + 4 * g + q;              // + 8 * rg
        const int ccol = wn * TNI * 32 + c4;
        const int64_t grow = (int64_t)b * p.Lp + p.halo_l + lrow;
        const int64_t orow = (int64_t)b * p.out_bs + p.out_base + (int64_t)lrow * p.ldo + zt * NP + ccol;       // + 8 * rg * ldo + ni * 32
        auto ldrow = [&](int rg) { return min(grow + 8 * rg, last_row); };      // (192-row tiles of the last item may reach past the buffer)
        f32x4 rres[TNI][4], racc[TNI][4];
        if (p.res) {
#pragma unroll
            for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) rres[ni][rg] = gload4(p.res + ldrow(rg) * NP + ccol + ni * 32);
        }
        if (p.accum) {
#pragma unroll
            for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) racc[ni][rg] = gload4(p.accum + ldrow(rg) * NP + ccol + ni * 32);
        }
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) {
            const int co = wn * TNI * 32 + ni * 32 + c4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float v0 = acc[mi][ni][4 * rg + 0], v1 = acc[mi][ni][4 * rg + 1];
                float v2 = acc[mi][ni][4 * rg + 2], v3 = acc[mi][ni][4 * rg + 3];
                quad_transpose(v0, v1, v2, v3, lane);
                const int l = lrow + 8 * rg;
                const int lpos = l * p.ostride + ((zt * NP + ccol + ni * 32) >> p.ph_shift);       // position of the signal (= l for a plain convolution)
                if (l >= p.L || lpos >= p.L_out) continue;
                const int64_t o = orow + (int64_t)8 * rg * p.ldo + ni * 32;
                f32x4 v = {v0 * a_sc + bv[0], v1 * a_sc + bv[1], v2 * a_sc + bv[2], v3 * a_sc + bv[3]};
                if (p.res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rres[ni][rg][e];
                }
                if (lpos >= Lb) v = f32x4{0.f, 0.f, 0.f, 0.f};   // behind a shorter item's end: the zero padding a B = 1 run sees

                if (p.out_x) {
                    f32x4 w = v;
                    if (p.accum) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] += racc[ni][rg][e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] *= p.out_scale;
                    *reinterpret_cast<f32x4*>(p.out_x + o) = w;
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
                }
                if (p.out_zhi) {
                    cvx_f16x4 zh, zl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float z = (v[e] > 0.f ? v[e] : v[e] * p.z_slope) * zs;
                        amax = cvx_amax3_c(amax, z, z);
                        z = fminf(fmaxf(z, -65504.f), 65504.f);
                        zh[e] = (_Float16)z;
                        zl[e] = (_Float16)(z - (float)zh[e]);
                    }
                    *reinterpret_cast<cvx_f16x4*>(p.out_zhi + o) = zh;
                    *reinterpret_cast<cvx_f16x4*>(p.out_zlo + o) = zl;
                }
            }
        }
    }
    cvx_sat_commit(p.sat, amax);
    if (p.amax_out) {                       // max |out_x| of the launch (non-negative floats order like their bits)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off, 64));
        if (lane == 0 && omax > 0.f && omax < __builtin_inff() && __float_as_uint(omax) > __atomic_load_n(p.amax_out, __ATOMIC_RELAXED))
            atomicMax(p.amax_out, __float_as_uint(omax));        // (most waves cannot raise the maximum: no atomic)
    }
#ifdef CVX_CONV_TRACE
    // trace build (tools/archive/conv_trace.py): with out_scale = 0 the fp32 output is all zero; the block leaves its 100 MHz stamps
    // (start, end of the main loop, end) and its CU in its own first output row
    __syncthreads();
    if (tid == 0 && p.out_x && p.out_scale == 0.f && l0 < p.L) {
        unsigned long long* tb = reinterpret_cast<unsigned long long*>(p.out_x + ((int64_t)b * p.Lp + p.halo_l + l0) * NP);
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tb[0] = tr0; tb[1] = tr1; tb[2] = __builtin_amdgcn_s_memrealtime(); tb[3] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

// ---------------------------------------------------------------- the same convolution on v_mfma_f32_16x16x32_f16 (round 4)
// Same staging, same stage ring, same arguments as conv_f16x3_kernel; what changes is the matrix instruction and everything that
// follows from its fragment layout:
//   * a 16 x 16 x 32 product takes the whole 32-channel chunk of a tap in ONE instruction (a lane's fragment = the 16-byte piece
//     lane >> 4 of row lane & 15: the four lane groups read the four pieces of a 64-byte row), so an accumulator register is read
//     and written once per 32 k instead of once per 16 k - the lever that gave the GEMM +13 % under the power cap;
//   * SWAPPED operands (D = W_frag . Z_frag^T): a lane holds 4 CONSECUTIVE CHANNELS of one position (position = lane & 15,
//     channels 4 * (lane >> 4) + 0..3 of a 16 x 16 tile), so every epilogue access is a 16-byte (8-byte fp16) vector with no
//     quad transposes (5 VALU instructions per value in the 32 x 32 epilogue);
//   * tiles are multiples of 16 positions per wave: 160-position blocks exist (TM16 = 5 on two wave rows), which put the first
//     ResBlock stage of the bench shape (8 x 5,000 positions) on exactly 256 blocks.
// MEASURED (round 4, rocprofv3 per launch, B = 8 x T = 1000, same box): the instruction shape itself buys nothing here - with the
// SAME tiles the Np = 128 stage takes 155.6 us against 146.1 us on the 32x32x16 kernel (+6.5 %; the un-swapped product with quad
// transposes 158.1), the Np = 64 upsampler 138 vs 127 - as in the attention kernel (DESIGN 4.3 iv) and unlike the GEMM.  What
// pays is the tile height: 256 blocks of 160 positions 114.0 us against 120.6 us for 216 blocks of 192 (-5.5 %, 19 launches per
// generator call).  So only the <5, 4, 4> instance is dispatched (dispatch_conv16), and only where it fills the chip better.
// TM16 / TN16 = 16-position / 16-channel tiles per wave; WN waves side by side over the channels (8 / WN over the positions).
template <int TM16, int TN16, int WN, bool GRP = false>
__global__ __launch_bounds__(512) void conv_f16x3_m16_kernel(const ConvKArgs<GRP> P)
{
    const Conv16Args& p = [&]() -> const Conv16Args& { if constexpr (GRP) return P.a[blockIdx.z]; else return P.a; }();
#define CVX_C16_MM(w, z, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(w, z, c, 0, 0, 0)
    constexpr int WM = 8 / WN;
    constexpr int NP = WN * TN16 * 16;
    constexpr int TS = 256 / NP;                         // taps per weight stage
    constexpr int TMB = WM * TM16 * 16;                  // positions per block
    constexpr int A_ROWS = TMB + 64;                     // + halo (<= 50) rounded up to the 16-row DMA piece
    constexpr int A_TILE = A_ROWS * CK;
    static_assert(TMB <= ::TMB && TMB % 16 == 0, "block tile: at most 256 positions");
    extern __shared__ __attribute__((aligned(16))) f16 smem_c[];
    f16* const As = smem_c;                              // [2 buffers][hi | lo][A_ROWS][32]
    f16* const Ws = smem_c + 4 * A_TILE;                 // [2 stages][hi | lo][256][32]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const float zs = p.z_scale ? *p.z_scale : 1.f;          // activation pre-scale of this stage's split pairs
    const float a_sc = p.acc_scale / zs;                    // (exact: both are powers of two)
    float amax = 0.f;
    const int l0 = blockIdx.x * TMB;
    const int b = blockIdx.y;
    const int zt = GRP ? 0 : (int)blockIdx.z;           // (grouped problems are plain convolutions: one output-column tile each)
    const int ksize = p.zk[zt], pad = p.zpad[zt];
    const f16* const w_hi = p.w_hi + p.zw[zt];
    const f16* const w_lo = p.w_lo + p.zw[zt];
    const int Lb = cvx_item_len(p.items, b, p.L_out);
    const int n_chunks = p.Cp_in / CK;
    const int n_groups = (ksize + TS - 1) / TS;
    const int steps = n_chunks * n_groups;

    const int prow = lane >> 2;
    const int64_t a_row0 = (int64_t)b * p.Lp + p.halo_l + l0 - pad;            // global row of tile row 0 (>= 0)
    const int64_t last_row = (int64_t)gridDim.y * p.Lp - 1;
    auto issue_a = [&](int chunk) {
        f16* dst = As + (chunk & 1) * 2 * A_TILE;
        for (int pc = wid; pc < A_ROWS / 16; pc += 8) {
            const int r = 16 * pc + prow;
            const int c4 = (lane & 3) ^ ((r >> 2) & 3);
            const int64_t src = min(a_row0 + r, last_row) * p.Cp_in + chunk * CK + 8 * c4;      // (short tiles may reach past roundup(L, 256) + 64)
            glds16(p.z_hi + src, dst + 16 * pc * CK);
            glds16(p.z_lo + src, dst + A_TILE + 16 * pc * CK);
        }
    };
    auto issue_w = [&](int st) {
        const int chunk = st / n_groups, grp = st - chunk * n_groups;
        const int t0 = grp * TS, nt = min(TS, ksize - t0);
        f16* dst = Ws + (st & 1) * 2 * W_TILE;
        const int64_t base = ((int64_t)chunk * ksize + t0) * NP * CK;
        for (int pc = wid; pc < nt * NP / 16; pc += 8) {
            const int r = 16 * pc + prow;
            const int c4 = (lane & 3) ^ ((r >> 2) & 3);
            const int64_t src = base + (int64_t)r * CK + 8 * c4;
            glds16(w_hi + src, dst + 16 * pc * CK);
            glds16(w_lo + src, dst + W_TILE + 16 * pc * CK);
        }
    };

    f32x4 acc[TM16][TN16];
#pragma unroll
    for (int mi = 0; mi < TM16; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN16; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int i15 = lane & 15, g4 = lane >> 4;
    const int woff = (wn * TN16 * 16 + i15) * CK + 8 * (g4 ^ ((i15 >> 2) & 3));
    const int arow_base = wm * TM16 * 16 + i15;

    issue_a(0);
    issue_w(0);
    for (int st = 0; st < steps; ++st) {
        const int chunk = st / n_groups, grp = st - chunk * n_groups;
        const int t0 = grp * TS, nt = min(TS, ksize - t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // stage st (and its activation tile) landed; the other buffers are free
        if (st + 1 < steps) {
            issue_w(st + 1);
            if (grp == 0 && chunk + 1 < n_chunks) issue_a(chunk + 1);
        }
        const f16* Ah = As + (chunk & 1) * 2 * A_TILE;
        const f16* Al = Ah + A_TILE;
        const f16* Wh = Ws + (st & 1) * 2 * W_TILE;
        const f16* Wl = Wh + W_TILE;
        for (int tl = 0; tl < nt; ++tl) {
            const int arow = arow_base + (t0 + tl) * p.dil;            // tile row of output position i15 for this tap
            const int aoff = arow * CK + 8 * (g4 ^ ((arow >> 2) & 3));
            const f16* wth = Wh + tl * NP * CK + woff;
            const f16* wtl = Wl + tl * NP * CK + woff;
            f16x8 fah[TM16], fal[TM16], fwh[TN16], fwl[TN16];
#pragma unroll
            for (int ni = 0; ni < TN16; ++ni) {
                fwh[ni] = *reinterpret_cast<const f16x8*>(wth + ni * 16 * CK);
                fwl[ni] = *reinterpret_cast<const f16x8*>(wtl + ni * 16 * CK);
            }
#pragma unroll
            for (int mi = 0; mi < TM16; ++mi) {
                fah[mi] = *reinterpret_cast<const f16x8*>(Ah + mi * 16 * CK + aoff);
                fal[mi] = *reinterpret_cast<const f16x8*>(Al + mi * 16 * CK + aoff);
            }
#pragma unroll
            for (int mi = 0; mi < TM16; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN16; ++ni)
                    acc[mi][ni] = CVX_C16_MM(fwh[ni], fal[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < TM16; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN16; ++ni)
                    acc[mi][ni] = CVX_C16_MM(fwl[ni], fah[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < TM16; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN16; ++ni)
                    acc[mi][ni] = CVX_C16_MM(fwh[ni], fah[mi], acc[mi][ni]);
        }
    }

    // ---- epilogue.  acc[mi][ni][e] of lane (i15, g4): position 16 * mi + i15 of the wave's rows, channel 16 * ni + 4 * g4 + e of
    // its columns.  The residual / accumulate vectors of TWO 16-position slices are requested together, ahead of their
    // arithmetic (as many loads in flight as the 32 x 32 epilogue's 32-position slice); rows behind L are inside the
    // allocation (clamped to its last row), so the loads need no predicate; the stores keep theirs.
    float omax = 0.f;
    const int rsel = i15;                                                     // the lane's position inside a 16 x 16 tile
    const int cw = wn * TN16 * 16 + 4 * g4;                                     // its first channel (+ 16 * ni)
#pragma unroll
    for (int mi0 = 0; mi0 < TM16; mi0 += 2) {
        constexpr int MS = 2;
        f32x4 rres[MS][TN16], racc[MS][TN16];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            if (mi0 + ms >= TM16) continue;
            const int lrow = l0 + wm * TM16 * 16 + (mi0 + ms) * 16 + rsel;
            const int64_t grow = min((int64_t)b * p.Lp + p.halo_l + lrow, last_row);
            if (p.res) {
#pragma unroll
                for (int ni = 0; ni < TN16; ++ni) rres[ms][ni] = gload4(p.res + grow * NP + cw + ni * 16);
            }
            if (p.accum) {
#pragma unroll
                for (int ni = 0; ni < TN16; ++ni) racc[ms][ni] = gload4(p.accum + grow * NP + cw + ni * 16);
            }
        }
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            if (mi0 + ms >= TM16) continue;
            const int mi = mi0 + ms;
            const int l = l0 + wm * TM16 * 16 + mi * 16 + rsel;
            const int64_t orow = (int64_t)b * p.out_bs + p.out_base + (int64_t)l * p.ldo + zt * NP + cw;
#pragma unroll
            for (int ni = 0; ni < TN16; ++ni) {
                const int lpos = l * p.ostride + ((zt * NP + cw + ni * 16) >> p.ph_shift);       // position of the signal (= l for a plain convolution)
                const float t0 = acc[mi][ni][0], t1 = acc[mi][ni][1], t2 = acc[mi][ni][2], t3 = acc[mi][ni][3];
                if (l >= p.L || lpos >= p.L_out) continue;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cw + ni * 16);
                const int64_t o = orow + ni * 16;
                f32x4 v = {t0 * a_sc + bv[0], t1 * a_sc + bv[1], t2 * a_sc + bv[2], t3 * a_sc + bv[3]};
                if (p.res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rres[ms][ni][e];
                }
                if (lpos >= Lb) v = f32x4{0.f, 0.f, 0.f, 0.f};   // behind a shorter item's end: the zero padding a B = 1 run sees
                if (p.out_x) {
                    f32x4 w = v;
                    if (p.accum) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] += racc[ms][ni][e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] *= p.out_scale;
                    *reinterpret_cast<f32x4*>(p.out_x + o) = w;
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
                }
                if (p.out_zhi) {
                    cvx_f16x4 zh, zl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float z = (v[e] > 0.f ? v[e] : v[e] * p.z_slope) * zs;
                        amax = cvx_amax3_c(amax, z, z);
                        z = fminf(fmaxf(z, -65504.f), 65504.f);
                        zh[e] = (_Float16)z;
                        zl[e] = (_Float16)(z - (float)zh[e]);
                    }
                    *reinterpret_cast<cvx_f16x4*>(p.out_zhi + o) = zh;
                    *reinterpret_cast<cvx_f16x4*>(p.out_zlo + o) = zl;
                }
            }
        }
    }
    cvx_sat_commit(p.sat, amax);
    if (p.amax_out) {                       // max |out_x| of the launch (non-negative floats order like their bits)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off, 64));
        if (lane == 0 && omax > 0.f && omax < __builtin_inff() && __float_as_uint(omax) > __atomic_load_n(p.amax_out, __ATOMIC_RELAXED))
            atomicMax(p.amax_out, __float_as_uint(omax));        // (most waves cannot raise the maximum: no atomic)
    }
}

// ---------------------------------------------------------------- fused ResBlock pair (narrow stages: Np = 32 / 64)
// x' = c2(leaky_relu(c1(leaky_relu(x)))) + x   (models.py:36-40) in ONE kernel: the intermediate t never leaves the CU
// and neither split pair (z = split(lrelu(x)), t) exists in HBM - a pair moves read x + write x' (+ the xs accumulate)
// instead of z, t (write), t (read), x, x', z'.  The narrow stages are HBM / fixed-cost bound with one launch per
// convolution; the wide ones (Np = 128 / 256) are matrix bound and their tiles do not fit, they keep it.
//
// Persistent blocks walk tiles of TM_out = 256 - (k - 1) output positions.  Per tile:
//   (0) the fp32 x rows [l0 - h2 - pad1, + 320) (h2 = (k-1)/2, pad1 = (k-1)*dil/2; all channels, one contiguous range of
//       the channels-last buffer) were loaded into registers during the previous tile's second pass; leaky_relu, the
//       stage's power-of-two pre-scale and the fp16 split happen on the way into LDS (swizzled chunk tiles, as above);
//   (1) pass 1: t rows [l0 - h2, + 256) = conv1 over the z tile (tap = row offset tap*dil), weights staged as above;
//       epilogue 1 writes split(lrelu(acc*s + b1)) - zero outside [0, L), which is what conv2's padding sees - into
//       LDS OVER the z tile (dead by then);
//   (2) pass 2: conv2 over the t tile (tap = row offset tap): output row r = position l0 + r needs t rows r .. r + k - 1, so
//       rows r < TM_out are complete (the others are computed and dropped: <= 4 % of the MFMAs, no ragged wave);
//       epilogue 2 adds bias and the fp32 residual (re-read from global: an L2 hit, the tile was read a pass earlier)
//       and stores x' (or folds it into xs).
// The weight stages of both passes and of consecutive tiles form one double-buffered stream.
struct PairArgs {
    const float* x;                    // [B][Lp][NP] fp32
    const f16 *w1h, *w1l, *w2h, *w2l;  // [chunk][tap][NP][32]
    const float *b1, *b2;              // [NP]
    const float* accum;                // [B][Lp][NP] or NULL
    float* out;                        // [B][Lp][NP]
    int B, L, Lp, ksize, dil, halo_l, tiles_per_seq, n_tiles;
    float acc1, acc2, out_scale, slope;
    const float* z_scale;
    uint32_t* sat;                     // sticky saturation flag of the device or NULL
    cvx_item_lengths items;            // ragged batch: valid positions per item (zeros are written behind them)
};

// (hi, lo) fp16 halves of four fp32 values, saturating: v_med3 clamp, packed RNE conversions
__device__ __forceinline__ void pair_split4(const f32x4 v, cvx_f16x4& hi, cvx_f16x4& lo, CvxSat& amax)
{
    cvx_amax4(amax, v);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
    const f16x2 h01 = __builtin_convertvector(f32x2{x[0], x[1]}, f16x2), h23 = __builtin_convertvector(f32x2{x[2], x[3]}, f16x2);
    const f16x2 l01 = __builtin_convertvector(f32x2{x[0] - (float)h01[0], x[1] - (float)h01[1]}, f16x2);
    const f16x2 l23 = __builtin_convertvector(f32x2{x[2] - (float)h23[0], x[3] - (float)h23[1]}, f16x2);
    hi = cvx_f16x4{h01[0], h01[1], h23[0], h23[1]};
    lo = cvx_f16x4{l01[0], l01[1], l23[0], l23[1]};
}

// TNI: 32-channel tiles (Np = 32 TNI); NW: waves per block = 32-row slices of the block's tile (ROWS = 32 NW).
// <1, 8>: 256 rows, 72 KiB, two blocks per CU (one block's epilogues and tile conversion run under the other's MFMA
// passes); <2, 8>: 256 rows, 144 KiB, one block per CU.  <2, 4> (128 rows, 2-tap weight stages, 80 KiB, two blocks per
// CU) buys the same overlap at Np = 64 with more halo and shorter stages: measured 0 ... 10 % SLOWER than <2, 8>, kept for
// A/B through cvx_respair16_args.flags.
template <int TNI, int NW> struct PairCfg {
    static constexpr int NP = 32 * TNI;
    static constexpr int ROWS = 32 * NW;                  // t rows per tile (outputs: ROWS - (k - 1))
    static constexpr int ZR = ROWS + 64;                  // rows of the z / t tile (+ halo <= 60, 16-row pieces)
    static constexpr int Z_TILE = ZR * CK;                // halves of one (chunk, hi or lo) tile
    static constexpr int TS = (TNI == 2 && NW == 4) ? 2 : 4;                  // taps per weight stage
    static constexpr int W_ST = TS * NP * CK;             // halves of one (hi or lo) weight stage
    static constexpr int LDS_HALVES = TNI * 2 * Z_TILE + 2 * 2 * W_ST;
};

template <int TNI, int NW>
__global__ __launch_bounds__(64 * NW, (TNI == 2 && NW == 8) ? 2 : NW / 2) void resblock_pair_f16x3_kernel(const PairArgs p)
{
    using Cfg = PairCfg<TNI, NW>;
    constexpr int NP = Cfg::NP, NCH = TNI, TS = Cfg::TS, W_ST = Cfg::W_ST, ZR = Cfg::ZR, ROWS = Cfg::ROWS;
    constexpr int NT = 64 * NW;                           // threads per block
    constexpr int A_TILE = Cfg::Z_TILE;                   // (shadows the convolution kernel's constant: this kernel's tile)
    constexpr int NF = (ZR * NP / 4 + NT - 1) / NT;       // float4 loads per thread for one x tile
    extern __shared__ __attribute__((aligned(16))) f16 smem_c[];
    f16* const Zs = smem_c;                               // [NCH chunks][hi | lo][ZR][32]   (z tile, then t tile)
    f16* const Ws = smem_c + NCH * 2 * A_TILE;            // [2 stages][hi | lo][TS taps * NP][32]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid;                                   // 8 x 1 waves: 32 rows x all NP channels each
    const float zs = p.z_scale ? *p.z_scale : 1.f;
    const float a1 = p.acc1 / zs, a2 = p.acc2 / zs;       // (exact: powers of two)
    CvxSat amax;
    const int k = p.ksize, h2 = (k - 1) / 2, pad1 = (k - 1) * p.dil / 2;
    const int tm_out = ROWS - 2 * h2;
    const int n_groups = (k + TS - 1) / TS;
    const int S = NCH * n_groups;
    const uint32_t last_row = (uint32_t)p.B * (uint32_t)p.Lp - 1u;     // (the launcher checks that a tensor spans < 4 GiB: 32-bit offsets)
    const bool has_accum = p.accum != nullptr;

    const int prow = lane >> 2;
    auto issue_w = [&](const f16* wh, const f16* wl, int st, int par) {
        const int chunk = st / n_groups, grp = st - chunk * n_groups;
        const int t0 = grp * TS, nt = min(TS, k - t0);
        f16* dst = Ws + par * 2 * W_ST;
        const int64_t base = ((int64_t)chunk * k + t0) * NP * CK;
        for (int pc = wid; pc < nt * NP / 16; pc += NW) {
            const int r = 16 * pc + prow;
            const int c4 = (lane & 3) ^ ((r >> 2) & 3);
            const int64_t src = base + (int64_t)r * CK + 8 * c4;
            glds16(wh + src, dst + 16 * pc * CK);
            glds16(wl + src, dst + W_ST + 16 * pc * CK);
        }
    };
    // x tile of tile id `tile`: NF float4 per thread, one contiguous global range (rows clamped to the allocation)
    f32x4 xr[NF];
    auto load_x = [&](int tile) {
        const int b = tile / p.tiles_per_seq, l0 = (tile - b * p.tiles_per_seq) * tm_out;
        const uint32_t g0 = (uint32_t)(b * p.Lp + p.halo_l + l0 - h2 - pad1);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = min(tid + NT * i, ZR * NP / 4 - 1);   // (the last round of a tile may be partial: re-read, not stored)
            const int row = f / (NP / 4), c4 = f % (NP / 4);
            const uint32_t gr = min(g0 + (uint32_t)row, last_row);
            xr[i] = *reinterpret_cast<const f32x4*>(p.x + (gr * (uint32_t)NP + 4u * (uint32_t)c4));
        }
    };
    const float zs_neg = zs * p.slope;
    auto store_z = [&]() {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid + NT * i;
            if (f >= ZR * NP / 4) continue;
            const int row = f / (NP / 4), c4 = f % (NP / 4);
            f32x4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = xr[i][e] * (xr[i][e] > 0.f ? zs : zs_neg);
            cvx_f16x4 zh, zl;
            pair_split4(z, zh, zl, amax);
            const int off = ((c4 >> 3) * 2 * ZR + row) * CK + 8 * (((c4 & 7) >> 1) ^ ((row >> 2) & 3)) + 4 * (c4 & 1);
            *reinterpret_cast<cvx_f16x4*>(Zs + off) = zh;
            *reinterpret_cast<cvx_f16x4*>(Zs + off + A_TILE) = zl;
        }
    };

    const int i31 = lane & 31, g = lane >> 5, q = lane & 3, c4e = 4 * (i31 >> 2);
    const int wswz = (i31 >> 2) & 3;
    int woff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) woff[s] = i31 * CK + 8 * ((2 * s + g) ^ wswz);
    const int arow_base = wm * 32 + i31;

    f32x16 acc[TNI];
    int seq = 0;                                          // weight stages issued so far by the block (buffer = seq & 1)
    // One pass: acc = sum over (chunk, tap) of A[row + tap*dil] . W[tap].  (nwh, nwl): weights of the pass that follows
    // (nullptr: none); prefetch_x: request the next tile's x rows behind the first weight stage.
    auto run_pass_plain = [&](const f16* wh, const f16* wl, const f16* nwh, const f16* nwl, int dil, bool prefetch_x, int next_tile) {
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
        for (int st = 0; st < S; ++st, ++seq) {
            const int chunk = st / n_groups, grp = st - chunk * n_groups;
            const int t0 = grp * TS, nt = min(TS, k - t0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // this stage's weights landed, the tile written before the pass is visible
            if (st + 1 < S) issue_w(wh, wl, st + 1, (seq + 1) & 1);
            else if (nwh) issue_w(nwh, nwl, 0, (seq + 1) & 1);
            if (st == 0 && prefetch_x) load_x(next_tile);
            const f16* Ah = Zs + chunk * 2 * A_TILE;
            const f16* Al = Ah + A_TILE;
            const f16* Wh = Ws + (seq & 1) * 2 * W_ST;
            const f16* Wl = Wh + W_ST;
            for (int tl = 0; tl < nt; ++tl) {
                const int arow = arow_base + (t0 + tl) * dil;
                const int aswz = (arow >> 2) & 3;
                const f16* wth = Wh + tl * NP * CK;
                const f16* wtl = Wl + tl * NP * CK;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int aoff = arow * CK + 8 * ((2 * s + g) ^ aswz);
                    const f16x8 fah = *reinterpret_cast<const f16x8*>(Ah + aoff);
                    const f16x8 fal = *reinterpret_cast<const f16x8*>(Al + aoff);
                    f16x8 fwh[TNI], fwl[TNI];
#pragma unroll
                    for (int ni = 0; ni < TNI; ++ni) {
                        fwh[ni] = *reinterpret_cast<const f16x8*>(wth + ni * 32 * CK + woff[s]);
                        fwl[ni] = *reinterpret_cast<const f16x8*>(wtl + ni * 32 * CK + woff[s]);
                    }
#pragma unroll
                    for (int ni = 0; ni < TNI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, fwh[ni], acc[ni], 0, 0, 0);
#pragma unroll
                    for (int ni = 0; ni < TNI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fwl[ni], acc[ni], 0, 0, 0);
#pragma unroll
                    for (int ni = 0; ni < TNI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fwh[ni], acc[ni], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // every wave is done reading the tile: it may be overwritten
    };

    // Software-pipelined form (see conv_f16x3_kernel; used at Np = 64 - measured 8-10 % faster at k = 7 / 11 - and not at Np = 32,
    // whose 128-register budget it overflows: 10 % slower there): a step = (tap, 16-channel half) = 3 TNI MFMAs on 2 + 2 TNI fragment
    // reads; the reads of step i + 1 are in flight while the MFMAs of step i issue (two register sets, inline-asm reads,
    // counted lgkmcnt waits).  The stage barrier sits between the two steps of a stage's last tap: every wave then holds all
    // of the stage's fragments, so its weight buffer is refilled with stage st + 2 (or stage 0 of the pass that follows).
    struct PFrags { f16x8 ah, al, wh[TNI], wl[TNI]; };
    constexpr int NFR = 2 + 2 * TNI;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_c;
    auto load_frags = [&](PFrags& F, int chunk_, int par, int tap_abs, int tl_, int dil_, const int s_) {
        const int arow = arow_base + tap_abs * dil_;
        const uint32_t aa = lds0 + 2u * (uint32_t)(chunk_ * 2 * A_TILE + arow * CK + 8 * ((2 * s_ + g) ^ ((arow >> 2) & 3)));
        const uint32_t wa = lds0 + 2u * (uint32_t)(NCH * 2 * A_TILE + par * 2 * W_ST + tl_ * NP * CK + woff[s_]);
        F.ah = lds_read16(aa, 0);
        F.al = lds_read16(aa, A_TILE * 2);
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) {
            F.wh[ni] = lds_read16(wa, ni * 32 * CK * 2);
            F.wl[ni] = lds_read16(wa, (W_ST + ni * 32 * CK) * 2);
        }
    };
    auto wait_frags = [&](PFrags& F) {
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NFR));
        asm volatile("" : "+v"(F.ah)); asm volatile("" : "+v"(F.al));
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) { asm volatile("" : "+v"(F.wh[ni])); asm volatile("" : "+v"(F.wl[ni])); }
    };
    auto mma = [&](const PFrags& F) {
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.al, F.wh[ni], acc[ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.ah, F.wl[ni], acc[ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.ah, F.wh[ni], acc[ni], 0, 0, 0);
    };
    auto run_pass_pipe = [&](const f16* wh, const f16* wl, const f16* nwh, const f16* nwl, int dil, bool prefetch_x, int next_tile) {
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // stage 0 of the pass landed, the tile written before the pass is visible
        if (S > 1) issue_w(wh, wl, 1, (seq + 1) & 1);
        else if (nwh) issue_w(nwh, nwl, 0, (seq + 1) & 1);
        if (prefetch_x) load_x(next_tile);
        PFrags F0, F1;
        load_frags(F0, 0, seq & 1, 0, 0, dil, 0);
        int chunk = 0, grp = 0;
        for (int st = 0; st < S; ++st, ++seq) {
            const int t0 = grp * TS, nt = min(TS, k - t0);
            for (int tl = 0; tl < nt; ++tl) {
                load_frags(F1, chunk, seq & 1, t0 + tl, tl, dil, 1);
                wait_frags(F0);
                mma(F0);
                const bool last = tl + 1 == nt;
                if (last && st + 1 < S) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();  // stage st + 1 landed; every wave holds the rest of stage st in registers
                    if (st + 2 < S) issue_w(wh, wl, st + 2, seq & 1);
                    else if (nwh) issue_w(nwh, nwl, 0, seq & 1);
                }
                const bool wrap = last && grp + 1 == n_groups;
                // (behind the pass's last step: a dummy read of valid LDS, never used)
                load_frags(F0, wrap ? (chunk + 1 < NCH ? chunk + 1 : 0) : chunk, last ? (seq + 1) & 1 : seq & 1, wrap ? 0 : t0 + tl + 1, last ? 0 : tl + 1, dil, 0);
                wait_frags(F1);
                mma(F1);
            }
            if (++grp == n_groups) { grp = 0; ++chunk; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // every wave is done reading the tile: it may be overwritten
    };
    auto run_pass = [&](const f16* wh, const f16* wl, const f16* nwh, const f16* nwl, int dil, bool prefetch_x, int next_tile) {
        if constexpr (CVX_PAIR_PIPE && TNI == 2) run_pass_pipe(wh, wl, nwh, nwl, dil, prefetch_x, next_tile);
        else run_pass_plain(wh, wl, nwh, nwl, dil, prefetch_x, next_tile);
    };

    int tile = blockIdx.x;
    if (tile >= p.n_tiles) return;
#ifdef CVX_PAIR_TRACE
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast_ = __builtin_readcyclecounter();
#define PSTAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tr[i] += now_ - tlast_; tlast_ = now_; }
#else
#define PSTAMP(i)
#endif
    load_x(tile);
    issue_w(p.w1h, p.w1l, 0, 0);
    store_z();
    PSTAMP(0)
    for (; tile < p.n_tiles; tile += gridDim.x) {
        const int b = tile / p.tiles_per_seq, l0 = (tile - b * p.tiles_per_seq) * tm_out;
        const int Lb = cvx_item_len(p.items, b, p.L);
        const int next_tile = tile + gridDim.x;
        const bool has_next = next_tile < p.n_tiles;

        run_pass(p.w1h, p.w1l, p.w2h, p.w2l, p.dil, false, 0);            // pass 1: conv1 over the z tile
        PSTAMP(1)
        // ---- epilogue 1: t = split(lrelu(acc*a1 + b1) * zs), zero outside the signal, over the z tile
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b1 + ni * 32 + c4e);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float v0 = acc[ni][4 * rg + 0], v1 = acc[ni][4 * rg + 1], v2 = acc[ni][4 * rg + 2], v3 = acc[ni][4 * rg + 3];
                quad_transpose(v0, v1, v2, v3, lane);
                const int row = wm * 32 + 8 * rg + 4 * g + q;
                const int pos = l0 - h2 + row;
                const bool inside = pos >= 0 && pos < Lb;
                const float sp = inside ? zs : 0.f, sn = inside ? zs_neg : 0.f;
                f32x4 v = {fmaf(v0, a1, bv[0]), fmaf(v1, a1, bv[1]), fmaf(v2, a1, bv[2]), fmaf(v3, a1, bv[3])};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= v[e] > 0.f ? sp : sn;
                cvx_f16x4 zh, zl;
                pair_split4(v, zh, zl, amax);
                const int off = (ni * 2 * ZR + row) * CK + 8 * ((c4e >> 3) ^ ((row >> 2) & 3)) + (c4e & 7);
                *reinterpret_cast<cvx_f16x4*>(Zs + off) = zh;
                *reinterpret_cast<cvx_f16x4*>(Zs + off + A_TILE) = zl;
            }
        }
        PSTAMP(2)
        run_pass(p.w2h, p.w2l, has_next ? p.w1h : nullptr, p.w1l, 1, has_next, next_tile);      // pass 2: conv2 over the t tile
        PSTAMP(3)
        if (has_next) store_z();                   // the next tile's z (its x rows arrived during pass 2)
        PSTAMP(4)
        // ---- epilogue 2: x' = acc*a2 + b2 + x   (rows r < tm_out, positions < L).  The residual (and accumulate) vectors of the
        // wave's whole slice are requested first, unpredicated (rows clamped to the allocation): one exposed round trip to
        // the Infinity Cache per tile instead of one per store group (4 - 8 per tile).
        f32x4 rres[TNI][4], racc[TNI][4];
        const uint32_t grow0 = (uint32_t)(b * p.Lp + p.halo_l + l0 + wm * 32 + 4 * g + q);
        constexpr int RG_AHEAD = TNI == 2 ? 4 : 3;    // (<1, 8> lives under a 128-register cap: 3 of its 4 vectors ahead, no spill)
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
            for (int rg = 0; rg < RG_AHEAD; ++rg)
                rres[ni][rg] = gload4(p.x + (min(grow0 + 8u * rg, last_row) * (uint32_t)NP + (uint32_t)(ni * 32 + c4e)));
        constexpr bool BATCH_ACC = TNI == 2;       // (<1, 8>: the accumulate loads stay in the store loop)
        if (BATCH_ACC && has_accum) {
#pragma unroll
            for (int ni = 0; ni < TNI; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    racc[ni][rg] = gload4(p.accum + (min(grow0 + 8u * rg, last_row) * (uint32_t)NP + (uint32_t)(ni * 32 + c4e)));
        }
#pragma unroll
        for (int ni = 0; ni < TNI; ++ni) {
            const int co = ni * 32 + c4e;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + co);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float v0 = acc[ni][4 * rg + 0], v1 = acc[ni][4 * rg + 1], v2 = acc[ni][4 * rg + 2], v3 = acc[ni][4 * rg + 3];
                quad_transpose(v0, v1, v2, v3, lane);
                const int row = wm * 32 + 8 * rg + 4 * g + q;
                const int l = l0 + row;
                if (row >= tm_out || l >= p.L) continue;
                const uint32_t o = (uint32_t)(b * p.Lp + p.halo_l + l) * (uint32_t)NP + (uint32_t)co;
                const f32x4 r4 = rg < RG_AHEAD ? rres[ni][rg] : gload4(p.x + o);
                f32x4 v = {fmaf(v0, a2, bv[0]) + r4[0], fmaf(v1, a2, bv[1]) + r4[1], fmaf(v2, a2, bv[2]) + r4[2], fmaf(v3, a2, bv[3]) + r4[3]};
                if (has_accum) {
                    const f32x4 a4 = BATCH_ACC ? racc[ni][rg] : gload4(p.accum + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += a4[e];
                }
                if (l >= Lb) v = f32x4{0.f, 0.f, 0.f, 0.f};      // behind a shorter item's end
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
                *reinterpret_cast<f32x4*>(p.out + o) = v;
            }
        }
        PSTAMP(5)
    }
    cvx_sat_commit(p.sat, amax);
#ifdef CVX_PAIR_TRACE
    if (lane == 0 && p.accum == nullptr && p.out_scale == 0.f) {      // (trace build: out_scale 0 makes the output all zero; stamps go on top)
        unsigned long long* tb = reinterpret_cast<unsigned long long*>(p.out) + ((size_t)blockIdx.x * 8 + wid) * 8;
        for (int i = 0; i < 6; ++i) tb[i] = tr[i];
    }
#endif
}

// ---------------------------------------------------------------- layout converters (HBM-bound transposes)
// channel-major fp32 [B][C][L]  ->  channels-last [B][Lp][Cp]: fp32 copy (optional) + split fp16 of leaky_relu(x)
__global__ __launch_bounds__(256) void cm_to_cl_kernel(const float* __restrict__ x, float* __restrict__ x_cl,
                                                      f16* __restrict__ z_hi, f16* __restrict__ z_lo,
                                                      int C, int L, int Lp, int Cp, int halo_l, float slope,
                                                      const float* __restrict__ z_scale, uint32_t* __restrict__ sat)
{
    __shared__ float tile[32][65];
    const float zs = z_scale ? *z_scale : 1.f;
    const int l0 = blockIdx.x * 64, c0 = blockIdx.y * 32, b = blockIdx.z;
    const int tid = threadIdx.x;
    {
        const int j = tid & 63, cs = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + cs + 4 * i, l = l0 + j;
            tile[cs + 4 * i][j] = (c < C && l < L) ? x[((int64_t)b * C + c) * L + l] : 0.f;
        }
    }
    __syncthreads();
    const int c = tid & 31, js = tid >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = js + 8 * i, l = l0 + j;
        if (l >= L) continue;
        const float v = tile[c][j];
        const int64_t o = ((int64_t)b * Lp + halo_l + l) * Cp + c0 + c;
        if (x_cl) x_cl[o] = v;
        if (z_hi) {
            float z = (v > 0.f ? v : v * slope) * zs;
            cvx_sat_commit(sat, fabsf(z));
            z = fminf(fmaxf(z, -65504.f), 65504.f);
            const f16 h = (f16)z;
            z_hi[o] = h;
            z_lo[o] = (f16)(z - (float)h);
        }
    }
}

// channels-last fp32 [B][Lp][Cp]  ->  channel-major fp32 [B][C][L]
__global__ __launch_bounds__(256) void cl_to_cm_kernel(const float* __restrict__ x_cl, float* __restrict__ x, int C, int L,
                                                      int Lp, int Cp, int halo_l)
{
    __shared__ float tile[64][33];
    const int l0 = blockIdx.x * 64, c0 = blockIdx.y * 32, b = blockIdx.z;
    const int tid = threadIdx.x;
    {
        const int c = tid & 31, js = tid >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = js + 8 * i, l = l0 + j;
            tile[j][c] = (l < L) ? x_cl[((int64_t)b * Lp + halo_l + l) * Cp + c0 + c] : 0.f;
        }
    }
    __syncthreads();
    const int j = tid & 63, cs = tid >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c0 + cs + 4 * i, l = l0 + j;
        if (c < C && l < L) x[((int64_t)b * C + c) * L + l] = tile[j][cs + 4 * i];
    }
}

template <int TMI, int TNI, int WN>
void launch_conv16(const Conv16Args* a, int n_grp, int B, int n_ztiles, hipStream_t st)
{
    constexpr int tmb = (8 / WN) * TMI * 32;
    const size_t lds = (size_t)LDS_HALVES * sizeof(f16);
    if (n_grp > 0) {                  // n_grp independent problems of one shape: blockIdx.z = problem
        ConvKArgs<true> P;
        for (int g = 0; g < 3; ++g) P.a[g] = a[g < n_grp ? g : 0];
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_f16x3_kernel<TMI, TNI, WN, true>), (int)lds);
        dim3 grid((unsigned)((a[0].L + tmb - 1) / tmb), (unsigned)B, (unsigned)n_grp);
        hipLaunchKernelGGL((conv_f16x3_kernel<TMI, TNI, WN, true>), grid, dim3(512), lds, st, P);
        return;
    }
    cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_f16x3_kernel<TMI, TNI, WN, false>), (int)lds);
    dim3 grid((unsigned)((a[0].L + tmb - 1) / tmb), (unsigned)B, (unsigned)n_ztiles);
    hipLaunchKernelGGL((conv_f16x3_kernel<TMI, TNI, WN, false>), grid, dim3(512), lds, st, ConvKArgs<false>{a[0]});
}
template <int TM16, int TN16, int WN>
void launch_conv16_m16(const Conv16Args* a, int n_grp, int B, int n_ztiles, hipStream_t st)
{
    constexpr int tmb = (8 / WN) * TM16 * 16;
    const size_t lds = (size_t)LDS_HALVES * sizeof(f16);
    if (n_grp > 0) {
        ConvKArgs<true> P;
        for (int g = 0; g < 3; ++g) P.a[g] = a[g < n_grp ? g : 0];
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_f16x3_m16_kernel<TM16, TN16, WN, true>), (int)lds);
        dim3 grid((unsigned)((a[0].L + tmb - 1) / tmb), (unsigned)B, (unsigned)n_grp);
        hipLaunchKernelGGL((conv_f16x3_m16_kernel<TM16, TN16, WN, true>), grid, dim3(512), lds, st, P);
        return;
    }
    cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_f16x3_m16_kernel<TM16, TN16, WN, false>), (int)lds);
    dim3 grid((unsigned)((a[0].L + tmb - 1) / tmb), (unsigned)B, (unsigned)n_ztiles);
    hipLaunchKernelGGL((conv_f16x3_m16_kernel<TM16, TN16, WN, false>), grid, dim3(512), lds, st, ConvKArgs<false>{a[0]});
}

// kernel instance by output tile width; Np = 256: one block per CU, and the block height is the one whose rounds x height comes
// out smallest on this chip - 256 or 192 positions on the 32x32x16 kernel, or 160 on the 16x16x32 one (measured 13 % slower per
// position, rocprofv3: 114 us for 256 blocks of 160 against 120.6 for 216 blocks of 192 on stage 0 of the bench shape, 8 x 5,000
// positions - it wins where it fills the chip: 160 -> 216 -> 256 blocks there)
// k: the problem (n_grp == 0, n_ztiles output-column tiles) or n_grp problems of one shape (one column tile each)
void dispatch_conv16(const Conv16Args* k, int n_grp, int B, int n_ztiles, hipStream_t st, int cus)
{
    const int nz = n_grp > 0 ? n_grp : n_ztiles;
    if (k[0].Np == 256) {
        auto cost = [&](int rows) { const int64_t n = (int64_t)((k[0].L + rows - 1) / rows) * B * nz; return (double)((n + cus - 1) / cus * rows); };
        const double t256 = cost(256), t192 = cost(192), t160 = 1.13 * cost(160);
        if (t160 < t256 && t160 < t192) launch_conv16_m16<5, 4, 4>(k, n_grp, B, n_ztiles, st);
        else if (t192 < t256) launch_conv16<3, 2, 4>(k, n_grp, B, n_ztiles, st);
        else launch_conv16<4, 2, 4>(k, n_grp, B, n_ztiles, st);
    }
    else if (k[0].Np == 128) launch_conv16<2, 2, 2>(k, n_grp, B, n_ztiles, st);
    else if (k[0].Np == 64) launch_conv16<1, 2, 1>(k, n_grp, B, n_ztiles, st);
    else launch_conv16<1, 1, 1>(k, n_grp, B, n_ztiles, st);
}

// ---------------------------------------------------------------- fp32 channels-last -> split pair; conv_post on channels-last
// z = split(leaky_relu(x, slope) * *z_scale) over a whole channels-last buffer (zero rows / channels stay zero)
__global__ __launch_bounds__(256) void cl_split_kernel(const float* __restrict__ x, f16* __restrict__ z_hi, f16* __restrict__ z_lo,
                                                       int64_t n4, float slope, const float* __restrict__ z_scale, uint32_t* __restrict__ sat)
{
    const float zs = z_scale ? *z_scale : 1.f;
    CvxSat amax;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = gload4(x + 4 * i);
        f32x4 z;
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = (v[e] > 0.f ? v[e] : v[e] * slope) * zs;
        cvx_f16x4 zh, zl;
        pair_split4(z, zh, zl, amax);
        *reinterpret_cast<cvx_f16x4*>(z_hi + 4 * i) = zh;
        *reinterpret_cast<cvx_f16x4*>(z_lo + 4 * i) = zl;
    }
    cvx_sat_commit(sat, amax);
}

// conv_post (Conv1d(C, 1, 7, padding 3) on leaky_relu(x)) + tanh (models.py:112-114) reading the channels-last stage output:
// a block stages its 256 + 6 rows (leaky_relu applied; halo rows are zero in the buffer) in LDS and every thread sums its
// position in the order of the channel-major kernel (bias first, channels outer, taps inner) - the same bits.
constexpr int POST_ROWS = 256 + 6;
__global__ __launch_bounds__(256) void post_cl_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                      float* __restrict__ y, int C, int Np, int L, int Lp, int halo_l, float slope)
{
    extern __shared__ __attribute__((aligned(16))) float post_tile[];          // [POST_ROWS][Np + 1]
    const int l0 = blockIdx.x * 256, b = blockIdx.y, ld = Np + 1;
    const float* xb = x + ((int64_t)b * Lp + halo_l + l0 - 3) * Np;           // halo_l >= 3: row -3 exists
    const int64_t lim = ((int64_t)gridDim.y * Lp - ((int64_t)b * Lp + halo_l + l0 - 3)) * Np;      // floats left in the buffer
    for (int i = threadIdx.x; i < POST_ROWS * Np / 4; i += 256) {
        const int r = (4 * i) / Np, c = (4 * i) % Np;
        f32x4 v = (int64_t)4 * i + 3 < lim ? gload4(xb + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) post_tile[r * ld + c + e] = v[e] > 0.f ? v[e] : v[e] * slope;
    }
    __syncthreads();
    const int l = l0 + threadIdx.x;
    if (l >= L) return;
    float acc = bias;
    for (int ci = 0; ci < C; ++ci) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int pidx = l + k - 3;
            if (pidx >= 0 && pidx < L) acc = fmaf(w[ci * 7 + k], post_tile[(threadIdx.x + k) * ld + ci], acc);
        }
    }
    y[(int64_t)b * L + l] = tanhf(acc);
}

}  // namespace

// validate one convolution and fill the kernel's argument block
static int build_conv16(const cvx_conv16_args* a, cvx_stream_t s, Conv16Args& k)
{
    CVX_REQUIRE(a && a->z_hi && a->z_lo && a->w_hi && a->w_lo && a->bias, "conv1d_f16x3: null pointer");
    CVX_REQUIRE(a->B >= 0 && a->L > 0 && a->Cp_in > 0 && a->Cp_in % 32 == 0 &&
                (a->Np == 32 || a->Np == 64 || a->Np == 128 || a->Np == 256),
                "conv1d_f16x3: bad shape (L=%d Cp_in=%d Np=%d): channels are padded to 32 and Np is 32/64/128/256", a->L, a->Cp_in, a->Np);
    CVX_REQUIRE(a->ksize > 0 && a->dil > 0 && (a->ksize - 1) * a->dil <= A_ROWS - TMB - 14 && (a->ksize - 1) * a->dil % 2 == 0,
                "conv1d_f16x3: (ksize-1)*dil = %d must be even and <= 50", (a->ksize - 1) * a->dil);
    const int pad = (a->ksize - 1) * a->dil / 2;                       // "same" convolution (get_padding, utils.py:34-35)
    CVX_REQUIRE(a->halo_l >= pad && a->Lp >= a->halo_l + ((a->L + TMB - 1) / TMB) * TMB + (A_ROWS - TMB),
                "conv1d_f16x3: buffers need %d zero rows in front and Lp >= halo_l + roundup(L, 256) + 64 (halo_l=%d Lp=%d)", pad, a->halo_l, a->Lp);
    CVX_REQUIRE((a->out_zhi == nullptr) == (a->out_zlo == nullptr) && (a->out_x || a->out_zhi) && (!a->accum || a->out_x),
                "conv1d_f16x3: bad output combination");
    if (a->out_zhi) CVX_REQUIRE_SAT(s);
    k = Conv16Args{reinterpret_cast<const f16*>(a->z_hi), reinterpret_cast<const f16*>(a->z_lo),
                   reinterpret_cast<const f16*>(a->w_hi), reinterpret_cast<const f16*>(a->w_lo), a->bias, a->res, a->accum, a->out_x,
                   reinterpret_cast<f16*>(a->out_zhi), reinterpret_cast<f16*>(a->out_zlo),
                   a->L, a->Lp, a->Cp_in, a->Np, a->ksize, a->dil, pad, a->halo_l, a->acc_scale, a->out_scale, a->z_slope, a->z_scale_dev,
                   a->out_zhi ? cvx_sat_flag_for(s) : nullptr, a->items, {}, {}, {}, 0, 0, 0, 0, 0, 0, nullptr};
    k.zk[0] = a->ksize; k.zpad[0] = pad; k.zw[0] = 0;                  // one output-column tile, the plain layout
    k.out_bs = (long long)a->Lp * a->Np; k.out_base = (long long)a->halo_l * a->Np;
    k.ldo = a->Np; k.ostride = 1; k.ph_shift = 31; k.L_out = a->L;
    return CVX_OK;
}

extern "C" int cvx_hifigan_conv1d_f16x3(const cvx_conv16_args* a, cvx_stream_t s)
{
    Conv16Args k;
    const int rc = build_conv16(a, s, k);
    if (rc != CVX_OK) return rc;
    if (a->B == 0) return CVX_OK;
    dispatch_conv16(&k, 0, a->B, 1, cvx_hip_stream(s), cvx_ctx_cus(s));
    CVX_CHECK_LAUNCH("cvx_hifigan_conv1d_f16x3");
    return CVX_OK;
}

extern "C" int cvx_hifigan_conv1d_group_f16x3(const cvx_conv16_args* a, int32_t n, cvx_stream_t s)
{
    CVX_REQUIRE(a && n >= 1 && n <= 3, "conv1d_group_f16x3: 1..3 convolutions per launch (got %d)", n);
    if (n == 1) return cvx_hifigan_conv1d_f16x3(a, s);
    Conv16Args k[3];
    int order[3] = {0, 1, 2};
    for (int g = 0; g < n; ++g) {
        CVX_REQUIRE(a[g].B == a[0].B && a[g].L == a[0].L && a[g].Lp == a[0].Lp && a[g].Cp_in == a[0].Cp_in && a[g].Np == a[0].Np &&
                    a[g].halo_l == a[0].halo_l && a[g].items.item_len_dev == a[0].items.item_len_dev && a[g].items.mul == a[0].items.mul &&
                    a[g].items.add == a[0].items.add,
                    "conv1d_group_f16x3: the convolutions of a group share B, L, Lp, Cp_in, Np, halo_l and the item lengths");
        for (int h = 0; h < g; ++h)
            CVX_REQUIRE(!(a[g].out_x && a[g].out_x == a[h].out_x) && !(a[g].out_zhi && a[g].out_zhi == a[h].out_zhi),
                        "conv1d_group_f16x3: the convolutions of a group run concurrently: their outputs must not alias");
    }
    // longest tiles first (the blocks of problem 0 are dispatched first): descending kernel size
    for (int x = 0; x < n; ++x)
        for (int y = x + 1; y < n; ++y)
            if (a[order[y]].ksize > a[order[x]].ksize) { const int t = order[x]; order[x] = order[y]; order[y] = t; }
    for (int g = 0; g < n; ++g) {
        const int rc = build_conv16(&a[order[g]], s, k[g]);
        if (rc != CVX_OK) return rc;
    }
    if (a[0].B == 0) return CVX_OK;
    dispatch_conv16(k, n, a[0].B, 1, cvx_hip_stream(s), cvx_ctx_cus(s));
    CVX_CHECK_LAUNCH("cvx_hifigan_conv1d_group_f16x3");
    return CVX_OK;
}

extern "C" int cvx_hifigan_conv_transpose1d_f16x3(const cvx_convt16_args* a, cvx_stream_t s)
{
    CVX_REQUIRE(a && a->z_hi && a->z_lo && a->w_hi && a->w_lo && a->bias && a->out, "conv_transpose1d_f16x3: null pointer");
    CVX_REQUIRE(a->B >= 0 && a->L_in > 0 && a->Cp_in > 0 && a->Cp_in % 32 == 0 && a->stride >= 1 && a->stride <= 8 &&
                (a->Np_out == 32 || a->Np_out == 64 || a->Np_out == 128 || a->Np_out == 256),
                "conv_transpose1d_f16x3: bad shape (L_in=%d Cp_in=%d Np_out=%d stride=%d)", a->L_in, a->Cp_in, a->Np_out, a->stride);
    CVX_REQUIRE((a->tile_np == 32 || a->tile_np == 64 || a->tile_np == 128 || a->tile_np == 256) && a->tile_np % a->Np_out == 0 &&
                a->n_tiles >= 1 && a->n_tiles <= 8 && (int64_t)a->n_tiles * a->tile_np == (int64_t)a->stride * a->Np_out,
                "conv_transpose1d_f16x3: %d tiles of %d columns do not cover stride * Np_out = %d x %d", a->n_tiles, a->tile_np, a->stride, a->Np_out);
    const int M = a->L_out > 0 ? (a->L_out + a->stride - 1) / a->stride : 0;       // rows of `stride` output positions
    CVX_REQUIRE(a->L_out > 0 && M <= a->L_in + 32 && a->halo_out >= 0 && a->Lp_out >= a->halo_out + a->L_out,
                "conv_transpose1d_f16x3: L_out = %d needs ceil(L_out / stride) <= L_in + 32 (the zero rows behind the input) and "
                "Lp_out >= halo_out + L_out (L_in=%d Lp_out=%d)", a->L_out, a->L_in, a->Lp_out);
    int max_up = 0;
    for (int t = 0; t < a->n_tiles; ++t) {
        CVX_REQUIRE(a->tile_taps[t] >= 1 && a->tile_taps[t] <= 8 && a->tile_pad[t] <= a->halo_in && a->tile_pad[t] >= -8 && a->tile_w_off[t] >= 0 &&
                    a->tile_w_off[t] % 8 == 0, "conv_transpose1d_f16x3: bad tap table entry %d", t);
        max_up = std::max(max_up, a->tile_taps[t] - 1 - a->tile_pad[t]);
    }
    CVX_REQUIRE(a->Lp_in >= a->halo_in + ((a->L_in + TMB - 1) / TMB) * TMB + (A_ROWS - TMB) && max_up <= A_ROWS - TMB - 14,
                "conv_transpose1d_f16x3: the input buffer needs Lp_in >= halo_in + roundup(L_in, 256) + 64 (halo_in=%d Lp_in=%d)", a->halo_in, a->Lp_in);
    if (a->B == 0) return CVX_OK;
    Conv16Args k{reinterpret_cast<const f16*>(a->z_hi), reinterpret_cast<const f16*>(a->z_lo),
                 reinterpret_cast<const f16*>(a->w_hi), reinterpret_cast<const f16*>(a->w_lo), a->bias, nullptr, nullptr, a->out,
                 nullptr, nullptr, M, a->Lp_in, a->Cp_in, a->tile_np, 1, 1, 0, a->halo_in, a->acc_scale, 1.f, 0.f, a->z_scale_dev,
                 nullptr, a->items, {}, {}, {}, 0, 0, 0, 0, 0, 0, nullptr};
    for (int t = 0; t < a->n_tiles; ++t) { k.zk[t] = a->tile_taps[t]; k.zpad[t] = a->tile_pad[t]; k.zw[t] = a->tile_w_off[t]; }
    k.out_bs = (long long)a->Lp_out * a->Np_out; k.out_base = (long long)a->halo_out * a->Np_out;
    k.ldo = a->stride * a->Np_out; k.ostride = a->stride; k.ph_shift = __builtin_ctz((unsigned)a->Np_out); k.L_out = a->L_out;
    k.amax_out = a->amax_bits_dev;
    dispatch_conv16(&k, 0, a->B, a->n_tiles, cvx_hip_stream(s), cvx_ctx_cus(s));
    CVX_CHECK_LAUNCH("cvx_hifigan_conv_transpose1d_f16x3");
    return CVX_OK;
}

extern "C" int cvx_hifigan_split_channels_last(const float* x_cl, uint16_t* z_hi, uint16_t* z_lo, int64_t n, float slope,
                                               const float* z_scale_dev, cvx_stream_t s)
{
    CVX_REQUIRE(x_cl && z_hi && z_lo && n >= 0 && n % 4 == 0, "split_channels_last: bad arguments (n must be a multiple of 4)");
    if (n == 0) return CVX_OK;
    CVX_REQUIRE_SAT(s);
    const int64_t n4 = n / 4;
    const unsigned grid = (unsigned)std::min<int64_t>((n4 + 255) / 256, (int64_t)cvx_ctx_cus(s) * 16);
    hipLaunchKernelGGL(cl_split_kernel, dim3(grid), dim3(256), 0, cvx_hip_stream(s), x_cl,
                       reinterpret_cast<f16*>(z_hi), reinterpret_cast<f16*>(z_lo), n4, slope, z_scale_dev, cvx_sat_flag_for(s));
    CVX_CHECK_LAUNCH("cvx_hifigan_split_channels_last");
    return CVX_OK;
}

extern "C" int cvx_hifigan_post_channels_last_f32(const float* x_cl, const float* w, float bias, float* y, int32_t B, int32_t C, int32_t Np,
                                                  int32_t L, int32_t Lp, int32_t halo_l, float slope, cvx_stream_t s)
{
    CVX_REQUIRE(x_cl && w && y && B >= 0 && C > 0 && Np >= C && Np % 4 == 0 && Np <= 64 && L > 0 && halo_l >= 3 && Lp >= halo_l + L + 3,
                "post_channels_last: bad arguments (C=%d Np=%d L=%d Lp=%d halo_l=%d; Np <= 64)", C, Np, L, Lp, halo_l);
    if (B == 0) return CVX_OK;
    const size_t lds = (size_t)POST_ROWS * (Np + 1) * sizeof(float);
    cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&post_cl_kernel), (int)lds);
    hipLaunchKernelGGL(post_cl_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B), dim3(256), lds, cvx_hip_stream(s),
                       x_cl, w, bias, y, C, Np, L, Lp, halo_l, slope);
    CVX_CHECK_LAUNCH("cvx_hifigan_post_channels_last_f32");
    return CVX_OK;
}

extern "C" int cvx_hifigan_resblock_pair_f16x3(const cvx_respair16_args* a, cvx_stream_t s)
{
    CVX_REQUIRE(a && a->x && a->out && a->c1.w_hi && a->c1.w_lo && a->c2.w_hi && a->c2.w_lo && a->c1.bias && a->c2.bias,
                "resblock_pair_f16x3: null pointer");
    CVX_REQUIRE(a->out != a->x, "resblock_pair_f16x3: out must not alias x (tiles read a halo of x)");
    CVX_REQUIRE(a->B >= 0 && a->L > 0 && (a->Np == 32 || a->Np == 64), "resblock_pair_f16x3: Np must be 32 or 64 (got %d)", a->Np);
    CVX_REQUIRE(a->ksize > 0 && a->ksize % 2 == 1 && a->dil > 0 && (a->ksize - 1) * a->dil + (a->ksize - 1) <= 60,
                "resblock_pair_f16x3: (ksize-1)*(dil+1) = %d must be <= 60 and ksize odd", (a->ksize - 1) * (a->dil + 1));
    const int h2 = (a->ksize - 1) / 2, pad1 = (a->ksize - 1) * a->dil / 2;
    CVX_REQUIRE(a->halo_l >= h2 + pad1 && a->Lp >= a->halo_l + a->L + pad1 + h2,
                "resblock_pair_f16x3: buffers need %d zero rows in front of and behind the signal (halo_l=%d Lp=%d L=%d)",
                h2 + pad1, a->halo_l, a->Lp, a->L);
    if (a->B == 0) return CVX_OK;
    CVX_REQUIRE_SAT(s);                                                    // (the intermediate of the pair is a split pair)
    const bool big64 = a->Np == 64 && !(a->flags & 1);                    // default: 256-row tiles, one block per CU
    const int rows = (a->Np == 64 && !big64) ? 128 : 256;
    const int tm_out = rows - 2 * h2;
    const int tps = (a->L + tm_out - 1) / tm_out;
    const int64_t n_tiles = (int64_t)tps * a->B;
    CVX_REQUIRE(n_tiles < (1ll << 30) && (int64_t)a->B * a->Lp * a->Np * 4 < (1ll << 32), "resblock_pair_f16x3: a tensor must span < 4 GiB");
    PairArgs k{a->x, reinterpret_cast<const f16*>(a->c1.w_hi), reinterpret_cast<const f16*>(a->c1.w_lo),
               reinterpret_cast<const f16*>(a->c2.w_hi), reinterpret_cast<const f16*>(a->c2.w_lo), a->c1.bias, a->c2.bias,
               a->accum, a->out, a->B, a->L, a->Lp, a->ksize, a->dil, a->halo_l, tps, (int)n_tiles,
               a->c1.acc_scale, a->c2.acc_scale, a->out_scale, 0.1f, a->z_scale_dev, cvx_sat_flag_for(s), a->items};
    hipStream_t st = cvx_hip_stream(s);
    const int cus = cvx_ctx_cus(s);
    const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, (int64_t)cus * (big64 ? 1 : 2));      // two blocks per CU
#define CVX_LAUNCH_PAIR(TNI_, NW_)                                                                                       \
    {                                                                                                                    \
        const size_t lds = (size_t)PairCfg<TNI_, NW_>::LDS_HALVES * sizeof(f16);                                         \
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&resblock_pair_f16x3_kernel<TNI_, NW_>), (int)lds);          \
        hipLaunchKernelGGL((resblock_pair_f16x3_kernel<TNI_, NW_>), dim3(grid), dim3(64 * NW_), lds, st, k);             \
    }
    if (a->Np == 32) CVX_LAUNCH_PAIR(1, 8)
    else if (big64) CVX_LAUNCH_PAIR(2, 8)
    else CVX_LAUNCH_PAIR(2, 4)
#undef CVX_LAUNCH_PAIR
    CVX_CHECK_LAUNCH("cvx_hifigan_resblock_pair_f16x3");
    return CVX_OK;
}

extern "C" int cvx_hifigan_to_channels_last(const float* x, float* x_cl, uint16_t* z_hi, uint16_t* z_lo, int32_t B, int32_t C,
                                            int32_t L, int32_t Lp, int32_t Cp, int32_t halo_l, float slope, cvx_stream_t s)
{
    return cvx_hifigan_to_channels_last_scaled(x, x_cl, z_hi, z_lo, B, C, L, Lp, Cp, halo_l, slope, nullptr, s);
}

extern "C" int cvx_hifigan_to_channels_last_scaled(const float* x, float* x_cl, uint16_t* z_hi, uint16_t* z_lo, int32_t B, int32_t C,
                                                   int32_t L, int32_t Lp, int32_t Cp, int32_t halo_l, float slope,
                                                   const float* z_scale_dev, cvx_stream_t s)
{
    CVX_REQUIRE(x && (x_cl || z_hi) && ((z_hi == nullptr) == (z_lo == nullptr)) && B >= 0 && C > 0 && L > 0 && Cp >= C && Cp % 32 == 0 &&
                halo_l >= 0 && Lp >= halo_l + L, "to_channels_last: bad arguments");
    if (B == 0) return CVX_OK;
    if (z_hi) CVX_REQUIRE_SAT(s);
    dim3 grid((unsigned)((L + 63) / 64), (unsigned)(Cp / 32), (unsigned)B);
    hipLaunchKernelGGL(cm_to_cl_kernel, grid, dim3(256), 0, cvx_hip_stream(s), x, x_cl,
                       reinterpret_cast<f16*>(z_hi), reinterpret_cast<f16*>(z_lo), C, L, Lp, Cp, halo_l, slope, z_scale_dev,
                       z_hi ? cvx_sat_flag_for(s) : nullptr);
    CVX_CHECK_LAUNCH("cvx_hifigan_to_channels_last");
    return CVX_OK;
}

extern "C" int cvx_hifigan_from_channels_last(const float* x_cl, float* x, int32_t B, int32_t C, int32_t L, int32_t Lp,
                                              int32_t Cp, int32_t halo_l, cvx_stream_t s)
{
    CVX_REQUIRE(x_cl && x && B >= 0 && C > 0 && L > 0 && Cp >= C && Cp % 32 == 0 && halo_l >= 0 && Lp >= halo_l + L,
                "from_channels_last: bad arguments");
    if (B == 0) return CVX_OK;
    dim3 grid((unsigned)((L + 63) / 64), (unsigned)(Cp / 32), (unsigned)B);
    hipLaunchKernelGGL(cl_to_cm_kernel, grid, dim3(256), 0, cvx_hip_stream(s), x_cl, x, C, L, Lp, Cp, halo_l);
    CVX_CHECK_LAUNCH("cvx_hifigan_from_channels_last");
    return CVX_OK;
}
