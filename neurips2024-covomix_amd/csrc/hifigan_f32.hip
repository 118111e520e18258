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

template <int MT>   // MFMA tiles per wave along co; CO_T = 32*MT
__global__ __launch_bounds__(256, 2) void conv1d_mfma_kernel(const cvx_conv_args p, int n_chunks)
{
    constexpr int CO_T = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [CK][X_LD]
    float* Ws = smem + CK * X_LD;           // [ksize][CO_T][W_LD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int l0 = blockIdx.x * LT;
    const int co_blk = blockIdx.y;
    const int b = blockIdx.z;
    const int halo = (p.ksize - 1) * p.dil;
    const int xw = LT + halo;                                   // staged positions per channel
    const int64_t Lv = (int64_t)(p.Lin - 1) * p.up + 1;         // virtual (zero-stuffed) length
    const float* xb = p.x + (int64_t)b * p.Cin * p.Lin;
    const int w_chunk_floats = p.ksize * CO_T * CK;
    const float* wblk = p.Wp + (int64_t)co_blk * n_chunks * w_chunk_floats;

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
                if (l >= p.Lout) continue;
                const int64_t o = ((int64_t)b * p.Cout + co) * p.Lout + l;
                float v = acc[mi][ni][r] + bv;
                if (p.res) v += p.res[o];
                if (p.accum) v += p.accum[o];
                p.out[o] = v * p.out_scale;
            }
        }
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
    hipStream_t st = reinterpret_cast<hipStream_t>(s);
    static size_t lds_set[2] = {0, 0};
    if (cot == 32) {
        if (lds > lds_set[0]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            lds_set[0] = lds;
        }
        hipLaunchKernelGGL(conv1d_mfma_kernel<1>, grid, dim3(256), lds, st, *a, n_chunks);
    } else {
        if (lds > lds_set[1]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            lds_set[1] = lds;
        }
        hipLaunchKernelGGL(conv1d_mfma_kernel<2>, grid, dim3(256), lds, st, *a, n_chunks);
    }
    CVX_CHECK_LAUNCH("cvx_hifigan_conv1d_f32");
    return CVX_OK;
}

extern "C" int cvx_hifigan_post_f32(const float* x, const float* w, float bias, float* y,
                                    int32_t B, int32_t Cin, int32_t L, float slope, cvx_stream_t s)
{
    CVX_REQUIRE(x && w && y && B >= 0 && Cin > 0 && L > 0, "hifigan_post: bad arguments");
    if (B == 0) return CVX_OK;
    dim3 grid((L + 255) / 256, B);
    hipLaunchKernelGGL(post_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(s), x, w, bias, y, Cin, L, slope);
    CVX_CHECK_LAUNCH("cvx_hifigan_post_f32");
    return CVX_OK;
}
