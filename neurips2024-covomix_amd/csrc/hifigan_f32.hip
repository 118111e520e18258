// HiFi-GAN generator kernels (gfx950): implicit-GEMM Conv1d / ConvTranspose1d on
// v_mfma_f32_32x32x2_f32 with the surrounding elementwise work fused in.
//
// Reference: covomix/vocoder/models.py:75-125 (Generator), :11-48 (ResBlock1),
// covomix/vocoder/utils.py:34-35 (get_padding), hifi-gan/config_covomix.json:11-15.
// The reference issues every conv, leaky_relu, residual add and the resblock average as a
// separate eager op (2.6 MB of fp32 activation traffic per mel frame).
// Here one launch does  lrelu(in) -> conv -> +bias -> +residual -> (+accum)*scale.
//
// GEMM view:  out[co, l] = sum_{ci,kk} W[co,ci,kk] * z[ci, l + kk*dil - pad]
//   M = co (A operand = packed weights), N = l (B operand = activations), K = ci x kk.
// Block = 4 waves; block tile = CO_T (32*MT) output channels x 256 positions; wave tile =
// CO_T x 64 (MT x 2 MFMA tiles).  The input channels are consumed in chunks of 16:
//   Xs[16][256 + halo]  activations with leaky_relu applied once at staging (zero padded,
//                       zero-stuffed when `up` > 1 so ConvTranspose1d runs on the same loop);
//   Ws[kk][CO_T][16(+4)] weights, one contiguous block per (co block, chunk) in the packed
//                       layout so the global read is perfectly coalesced.
// A fragments come from Ws with one ds_read_b128 per 4 MFMAs (same free-k-order trick as the
// GEMM: lanes 0-31 take ci 8q..8q+3, lanes 32-63 take 8q+4..8q+7); B fragments are
// ds_read_b32 of 32 consecutive positions - conflict free, and the tap shift kk*dil is just
// an address offset, so no im2col buffer exists anywhere.
// Channel counts 500/250/125/62/31 are zero-padded to multiples of 16 (ci) / 32 (co).
#include "cvx_common.h"
#include <string.h>

namespace {

constexpr int CK = 16;            // input channels per chunk
constexpr int W_LD = CK + 4;      // padded ci row of Ws (floats)
constexpr int LT = 256;           // output positions per block
constexpr int MAX_HALO = 64;      // (ksize-1)*dil <= 50 for every conv of config_covomix.json
constexpr int X_LD = LT + MAX_HALO;

// POLY: one phase of a ConvTranspose1d (stride s = pin.up) per block - blockIdx.y = phase * co_blocks + co block.
// Output l = s*m + r only sees the taps kk = kk0 + s*j of the flipped kernel (kk0 = (pad' - r) mod s, pad' = the pad of
// the zero-stuffed form) at input positions m + d + j, d = (r - pad' + kk0) / s: a stride-1 convolution with
// nt = ceil((k - kk0) / s) taps over the UN-stuffed input, written with stride s.  1/s of the zero-stuffed form's MFMAs.
// pin.Wp holds the s per-phase kernels Wf[:, :, kk0::s] packed like Conv1d weights, concatenated in phase order.
template <int MT, bool POLY>   // MFMA tiles per wave along co; CO_T = 32*MT
__global__ __launch_bounds__(256, 2) void conv1d_mfma_kernel(const cvx_conv_args pin, int n_chunks, unsigned* __restrict__ amax_bits)
{
    constexpr int CO_T = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [CK][X_LD]
    float* Ws = smem + CK * X_LD;           // [ksize][CO_T][W_LD]

    cvx_conv_args p = pin;
    int co_blk = blockIdx.y, ostride = 1, ooff = 0, Lm = pin.Lout;
    int64_t w_phase_off = 0;
    if constexpr (POLY) {
        const int s = pin.up, co_blocks = (pin.Cout + CO_T - 1) / CO_T;
        const int r = blockIdx.y / co_blocks;
        co_blk = blockIdx.y - r * co_blocks;
        int kk0 = 0, nt = 0;
        for (int rr = 0; rr <= r; ++rr) {                       // weight offset of this phase: the phases before it
            kk0 = ((pin.pad - rr) % s + s) % s;
            nt = kk0 < pin.ksize ? (pin.ksize - kk0 + s - 1) / s : 0;
            if (rr < r) w_phase_off += (int64_t)co_blocks * n_chunks * nt * CO_T * CK;
        }
        p.up = 1; p.dil = 1; p.ksize = nt;
        p.pad = -((r - pin.pad + kk0) / s);                    // (exact division: r - pad' + kk0 = 0 mod s)
        ostride = s; ooff = r;
        Lm = (pin.Lout - r + s - 1) / s;                        // outputs of this phase
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int l0 = blockIdx.x * LT;
    const int b = blockIdx.z;
    if (l0 >= Lm) return;
    const int Lvalid = cvx_item_len(pin.items, b, pin.Lout);    // ragged batch: zeros behind this item's last output
    const int halo = (p.ksize - 1) * p.dil;
    const int xw = LT + halo;                                   // staged positions per channel
    const int64_t Lv = (int64_t)(p.Lin - 1) * p.up + 1;         // virtual (zero-stuffed) length
    const float* xb = p.x + (int64_t)b * p.Cin * p.Lin;
    const int w_chunk_floats = p.ksize * CO_T * CK;
    const float* wblk = p.Wp + w_phase_off + (int64_t)co_blk * n_chunks * w_chunk_floats;

    f32x16 acc[MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    for (int ch = 0; ch < n_chunks; ++ch) {
        // ---- stage activations: Xs[ci][j] = z[ci0+ci][l0 - pad + j]
        for (int idx = tid; idx < CK * xw; idx += 256) {
            const int ci = idx / xw, j = idx - ci * xw;
            const int cg = ch * CK + ci;
            const int64_t pv = (int64_t)l0 - p.pad + j;
            float v = 0.f;
            if (cg < p.Cin && pv >= 0 && pv < Lv) {
                int64_t src = pv;
                bool ok = true;
                if (p.up > 1) { src = pv / p.up; ok = (src * p.up == pv); }
                if (ok) {
                    v = xb[(int64_t)cg * p.Lin + src];
                    v = v > 0.f ? v : v * p.in_slope;
                }
            }
            Xs[ci * X_LD + j] = v;
        }
        // ---- stage weights: contiguous [ksize][CO_T][CK] block -> padded rows
        const float* wsrc = wblk + (int64_t)ch * w_chunk_floats;
        for (int idx = tid; idx < (w_chunk_floats >> 2); idx += 256) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsrc + 4 * idx);
            const int rowi = idx >> 2, c4 = (idx & 3) * 4;        // row = kk*CO_T + co
            *reinterpret_cast<f32x4*>(Ws + rowi * W_LD + c4) = w4;
        }
        __syncthreads();

        // ---- MFMA over taps and the 16 channels of the chunk
        const float* xbase = Xs + wid * 64 + l31;
        for (int kk = 0; kk < p.ksize; ++kk) {
            const float* wk = Ws + (kk * CO_T + l31) * W_LD + 4 * half;
            const float* xk = xbase + kk * p.dil;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 af[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(wk + mi * 32 * W_LD + 8 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ci = 8 * q + 4 * half + t;
                    const float b0 = xk[ci * X_LD];
                    const float b1 = xk[ci * X_LD + 32];
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][t], b0, acc[mi][0], 0, 0, 0);
                        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][t], b1, acc[mi][1], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: row = co, col = position
    float vmax = 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_blk * CO_T + mi * 32 + mfma32_row(r, lane);
            if (co >= p.Cout) continue;
            const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int l = l0 + wid * 64 + ni * 32 + l31;
                if (l >= Lm) continue;
                const int64_t o = ((int64_t)b * p.Cout + co) * p.Lout + (int64_t)l * ostride + ooff;
                float v = acc[mi][ni][r] + bv;
                if (p.res) v += p.res[o];
                if (p.accum) v += p.accum[o];
                v *= p.out_scale;
                if (l * ostride + ooff >= Lvalid) v = 0.f;
                p.out[o] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
        }
    }
    if (amax_bits) {                        // max |out| of the launch (non-negative floats order like their bits)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        if (lane == 0 && vmax > 0.f && vmax < __builtin_inff()) atomicMax(amax_bits, __float_as_uint(vmax));
    }
}

// ---- conv_post: Cout = 1, k = 7: plain VALU kernel, one thread per output sample
__global__ __launch_bounds__(256) void post_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                  float* __restrict__ y, int Cin, int L, float slope)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    const float* xb = x + (int64_t)b * Cin * L;
    float acc = bias;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xr = xb + (int64_t)ci * L;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int pidx = l + k - 3;
            if (pidx >= 0 && pidx < L) {
                float v = xr[pidx];
                v = v > 0.f ? v : v * slope;
                acc = fmaf(w[ci * 7 + k], v, acc);
            }
        }
    }
    y[(int64_t)b * L + l] = tanhf(acc);
}

inline int co_tile(int Cout) { return Cout <= 32 ? 32 : 64; }

}  // namespace

extern "C" int64_t cvx_hifigan_packed_weight_floats(int32_t Cout, int32_t Cin, int32_t ksize)
{
    if (Cout <= 0 || Cin <= 0 || ksize <= 0) return 0;
    const int cot = co_tile(Cout);
    const int64_t co_blocks = (Cout + cot - 1) / cot;
    const int64_t chunks = (Cin + CK - 1) / CK;
    return co_blocks * chunks * ksize * cot * CK;
}

extern "C" int cvx_hifigan_pack_weight_f32(const float* w, int32_t Cout, int32_t Cin, int32_t ksize,
                                           int32_t transposed, float* Wp)
{
    CVX_REQUIRE(w && Wp && Cout > 0 && Cin > 0 && ksize > 0, "pack_weight: bad arguments");
    const int cot = co_tile(Cout);
    const int co_blocks = (Cout + cot - 1) / cot;
    const int chunks = (Cin + CK - 1) / CK;
    memset(Wp, 0, sizeof(float) * (size_t)cvx_hifigan_packed_weight_floats(Cout, Cin, ksize));
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int kk = 0; kk < ksize; ++kk) {
                // Conv1d weight [Cout][Cin][k]; ConvTranspose1d weight [Cin][Cout][k], flipped
                const float v = transposed ? w[((int64_t)ci * Cout + co) * ksize + (ksize - 1 - kk)]
                                           : w[((int64_t)co * Cin + ci) * ksize + kk];
                const int cb = co / cot, cw = co % cot, chn = ci / CK, cc = ci % CK;
                Wp[((((int64_t)cb * chunks + chn) * ksize + kk) * cot + cw) * CK + cc] = v;
            }
    return CVX_OK;
}

extern "C" int cvx_hifigan_conv1d_f32(const cvx_conv_args* a, cvx_stream_t s)
{
    CVX_REQUIRE(a && a->x && a->Wp && a->out, "conv1d: null pointer");
    CVX_REQUIRE(a->B >= 0 && a->Cin > 0 && a->Lin > 0 && a->Cout > 0 && a->Lout > 0, "conv1d: bad shape");
    CVX_REQUIRE(a->ksize > 0 && a->dil > 0 && a->up > 0 && a->pad >= 0, "conv1d: bad conv parameters");
    CVX_REQUIRE((a->ksize - 1) * a->dil <= MAX_HALO, "conv1d: (ksize-1)*dil = %d exceeds %d", (a->ksize - 1) * a->dil, MAX_HALO);
    const int64_t lv = (int64_t)(a->Lin - 1) * a->up + 1;
    CVX_REQUIRE(a->Lout == lv + 2 * a->pad - (int64_t)(a->ksize - 1) * a->dil, "conv1d: Lout %d inconsistent with geometry", a->Lout);
    CVX_REQUIRE(a->x != a->out, "conv1d: in-place on the input is not supported");
    if (a->B == 0) return CVX_OK;
    const int cot = co_tile(a->Cout);
    const int n_chunks = (a->Cin + CK - 1) / CK;
    const size_t lds = sizeof(float) * ((size_t)CK * X_LD + (size_t)a->ksize * cot * W_LD);
    dim3 grid((a->Lout + LT - 1) / LT, (a->Cout + cot - 1) / cot, a->B);
    hipStream_t st = cvx_hip_stream(s);
    if (cot == 32) {
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv1d_mfma_kernel<1, false>), (int)lds);
        hipLaunchKernelGGL((conv1d_mfma_kernel<1, false>), grid, dim3(256), lds, st, *a, n_chunks, (unsigned*)nullptr);
    } else {
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv1d_mfma_kernel<2, false>), (int)lds);
        hipLaunchKernelGGL((conv1d_mfma_kernel<2, false>), grid, dim3(256), lds, st, *a, n_chunks, (unsigned*)nullptr);
    }
    CVX_CHECK_LAUNCH("cvx_hifigan_conv1d_f32");
    return CVX_OK;
}

extern "C" int cvx_hifigan_conv_transpose1d_f32(const cvx_conv_args* a, uint32_t* amax_bits_dev, cvx_stream_t s)
{
    CVX_REQUIRE(a && a->x && a->Wp && a->out, "conv_transpose1d: null pointer");
    CVX_REQUIRE(a->B >= 0 && a->Cin > 0 && a->Lin > 0 && a->Cout > 0 && a->Lout > 0, "conv_transpose1d: bad shape");
    CVX_REQUIRE(a->ksize > 0 && a->dil == 1 && a->up > 1 && a->pad >= 0 && a->pad < a->ksize, "conv_transpose1d: bad conv parameters (dil must be 1, up = stride > 1)");
    CVX_REQUIRE(a->Lout == (int64_t)(a->Lin - 1) * a->up + 1 + 2 * a->pad - (a->ksize - 1), "conv_transpose1d: Lout %d inconsistent with geometry", a->Lout);
    CVX_REQUIRE(!a->res && !a->accum, "conv_transpose1d: no residual / accumulate inputs");
    CVX_REQUIRE(a->x != a->out, "conv_transpose1d: in-place on the input is not supported");
    if (a->B == 0) return CVX_OK;
    const int cot = co_tile(a->Cout);
    const int n_chunks = (a->Cin + CK - 1) / CK;
    const int nt_max = (a->ksize + a->up - 1) / a->up;
    const size_t lds = sizeof(float) * ((size_t)CK * X_LD + (size_t)nt_max * cot * W_LD);
    const int lm = (a->Lout + a->up - 1) / a->up;
    dim3 grid((lm + LT - 1) / LT, ((a->Cout + cot - 1) / cot) * a->up, a->B);
    hipStream_t st = cvx_hip_stream(s);
    if (cot == 32) {
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv1d_mfma_kernel<1, true>), (int)lds);
        hipLaunchKernelGGL((conv1d_mfma_kernel<1, true>), grid, dim3(256), lds, st, *a, n_chunks, amax_bits_dev);
    } else {
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&conv1d_mfma_kernel<2, true>), (int)lds);
        hipLaunchKernelGGL((conv1d_mfma_kernel<2, true>), grid, dim3(256), lds, st, *a, n_chunks, amax_bits_dev);
    }
    CVX_CHECK_LAUNCH("cvx_hifigan_conv_transpose1d_f32");
    return CVX_OK;
}

extern "C" int64_t cvx_hifigan_conv_transpose1d_packed_floats(int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t)
{
    if (Cout <= 0 || Cin <= 0 || ksize <= 0 || stride <= 0 || pad_t < 0 || pad_t >= ksize) return 0;
    int64_t n = 0;
    for (int r = 0; r < stride; ++r) {
        const int kk0 = (((ksize - 1 - pad_t) - r) % stride + stride) % stride;
        const int nt = kk0 < ksize ? (ksize - kk0 + stride - 1) / stride : 0;
        n += cvx_hifigan_packed_weight_floats(Cout, Cin, nt);
    }
    return n;
}

extern "C" int cvx_hifigan_pack_conv_transpose1d_f32(const float* w, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                                                     int32_t pad_t, float* Wp)
{
    CVX_REQUIRE(w && Wp && Cout > 0 && Cin > 0 && ksize > 0 && stride > 0 && pad_t >= 0 && pad_t < ksize, "pack_conv_transpose1d: bad arguments");
    const int cot = co_tile(Cout);
    const int chunks = (Cin + CK - 1) / CK;
    memset(Wp, 0, sizeof(float) * (size_t)cvx_hifigan_conv_transpose1d_packed_floats(Cout, Cin, ksize, stride, pad_t));
    float* dst = Wp;
    for (int r = 0; r < stride; ++r) {
        const int kk0 = (((ksize - 1 - pad_t) - r) % stride + stride) % stride;
        const int nt = kk0 < ksize ? (ksize - kk0 + stride - 1) / stride : 0;
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                for (int j = 0; j < nt; ++j) {
                    const int kk = kk0 + stride * j;                           // tap of the flipped kernel
                    const float v = w[((int64_t)ci * Cout + co) * ksize + (ksize - 1 - kk)];
                    const int cb = co / cot, cw = co % cot, chn = ci / CK, cc = ci % CK;
                    dst[((((int64_t)cb * chunks + chn) * nt + j) * cot + cw) * CK + cc] = v;
                }
        dst += cvx_hifigan_packed_weight_floats(Cout, Cin, nt);
    }
    return CVX_OK;
}

extern "C" int cvx_hifigan_post_f32(const float* x, const float* w, float bias, float* y,
                                    int32_t B, int32_t Cin, int32_t L, float slope, cvx_stream_t s)
{
    CVX_REQUIRE(x && w && y && B >= 0 && Cin > 0 && L > 0, "hifigan_post: bad arguments");
    if (B == 0) return CVX_OK;
    dim3 grid((L + 255) / 256, B);
    hipLaunchKernelGGL(post_kernel, grid, dim3(256), 0, cvx_hip_stream(s), x, w, bias, y, Cin, L, slope);
    CVX_CHECK_LAUNCH("cvx_hifigan_post_f32");
    return CVX_OK;
}
