// HuBERT prompt tokeniser: the pieces of the reference's
// HubertFeatureReader.get_feats / ApplyKmeans path that are not plain GEMMs or attention.
//   fairseq-hubert/fairseq/models/wav2vec/wav2vec2.py:844-923   conv feature extractor (first layer + GroupNorm here;
//                                                               layers 1-6 are GEMMs over overlapping channels-last rows)
//   fairseq-hubert/fairseq/models/wav2vec/wav2vec2.py:925-946   grouped positional convolution (operand packing here)
//   fairseq-hubert/fairseq/modules/layer_norm.py                LayerNorm
//   fairseq-hubert/examples/hubert/simple_kmeans/dump_km_label.py:36-43   k-means label = argmin distance
// All HBM-bound, channel axis innermost (coalesced), no LDS needed except the block reductions.
#include "cvx_common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- first conv layer: Conv1d(1 -> C, k, stride), no bias
// out[l][c] = sum_j w[c][j] * wav[l*stride + j];  block = 256 channels x 16 frames, the waveform window in LDS.
constexpr int C0_FR = 16;
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                    float* __restrict__ out, int64_t L, int C, int k, int stride)
{
    __shared__ float win[C0_FR * 8 + 32];          // (C0_FR - 1) * stride + k samples, stride <= 8, k <= 32
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.y * C0_FR;
    const int nfr = (int)min((int64_t)C0_FR, L - l0);
    const int nwin = (nfr - 1) * stride + k;
    for (int i = threadIdx.x; i < nwin; i += 256) win[i] = wav[l0 * stride + i];
    __syncthreads();
    if (c >= C) return;
    float wr[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) wr[j] = j < k ? w[(int64_t)c * k + j] : 0.f;
    for (int f = 0; f < nfr; ++f) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < k) acc = fmaf(wr[j], win[f * stride + j], acc);
        out[(l0 + f) * C + c] = acc;
    }
}

// ---------------------------------------------------------------- per-channel statistics over time (GroupNorm(C, C))
// two passes (mean, then centred second moment), each: per-chunk partial sums (thread = channel, coalesced rows) and a
// fixed-order final reduction in double => deterministic.
constexpr int ST_ROWS = 128;
__global__ __launch_bounds__(256) void colstat_partial_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                              float* __restrict__ partial, int64_t L, int C)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * ST_ROWS, r1 = min(r0 + ST_ROWS, L);
    const float m = mean ? mean[c] : 0.f;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const float d = x[r * C + c] - m;
        s += mean ? d * d : d;
    }
    partial[(int64_t)blockIdx.y * C + c] = s;
}

__global__ __launch_bounds__(256) void colstat_final_kernel(const float* __restrict__ partial, int nchunks, int C, int64_t L,
                                                            float* __restrict__ out, float eps, int rstd)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < nchunks; ++i) s += (double)partial[(int64_t)i * C + c];
    s /= (double)L;
    out[c] = rstd ? (float)(1.0 / sqrt(s + (double)eps)) : (float)s;
}

__global__ __launch_bounds__(256) void groupnorm_gelu_kernel(float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int64_t n4, int C)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    f32x4 v = *reinterpret_cast<f32x4*>(x + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf((v[e] - mean[c + e]) * rstd[c + e] * gamma[c + e] + beta[c + e]);
    *reinterpret_cast<f32x4*>(x + 4 * i) = v;
}

// ---------------------------------------------------------------- LayerNorm over the last axis: wave per row, row in registers
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        int64_t rows, int D, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    const int nvec = D >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nvec) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * j);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum_f(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] -= mean; q = fmaf(v[i][e], v[i][e], q); }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_f(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nvec) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * j);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * j);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[i][e] * rstd * g[e] + b[e];
            *reinterpret_cast<f32x4*>(y + row * D + 4 * j) = o;
        }
    }
}

// ---------------------------------------------------------------- operand of the grouped positional convolution
// out[g][j][c] = x[j - halo][g*cg + c] for halo <= j < halo + T, else 0   (rows j in [0, T + 2*halo)): the im2col row of
// output frame t of group g is then the CONTIGUOUS run out[g][t .. t+k-1][:] (k*cg floats, row stride cg).
__global__ __launch_bounds__(256) void group_pack_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         int T, int D, int G, int halo)
{
    const int cg = D / G, rows = T + 2 * halo;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over rows * D, x-order (coalesced reads)
    if (i >= (int64_t)rows * D) return;
    const int j = (int)(i / D), col = (int)(i % D);
    const int g = col / cg, c = col % cg;
    const int t = j - halo;
    out[((int64_t)g * rows + j) * cg + c] = (t >= 0 && t < T) ? x[(int64_t)t * D + col] : 0.f;
}

// ---------------------------------------------------------------- k-means label: argmin_j (|x|^2 - 2 x.C_j) + |C_j|^2
// one wave per frame; `dots` = x . C^T from the GEMM.  Same operation order as the reference expression
// (x.pow(2).sum(1) - 2 * matmul + Cnorm); ties -> lowest index (torch.argmin).
__global__ __launch_bounds__(256) void kmeans_argmin_kernel(const float* __restrict__ x, const float* __restrict__ dots,
                                                            const float* __restrict__ cnorm, int64_t* __restrict__ labels,
                                                            float* __restrict__ margin, int64_t T, int D, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    float xx = 0.f;
    for (int j = lane; j < D; j += 64) { const float v = x[row * D + j]; xx = fmaf(v, v, xx); }
    xx = wave_sum_f(xx);
    float best = INFINITY, second = INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < K; j += 64) {
        const float d = (xx - 2.0f * dots[row * K + j]) + cnorm[j];
        if (d < best) { second = best; best = d; bi = j; }
        else if (d < second) second = d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob < best || (ob == best && oi < bi)) { second = fminf(best, os); best = ob; bi = oi; }
        else second = fminf(second, ob);
    }
    if (lane == 0) {
        labels[row] = bi;
        if (margin) margin[row] = second - best;
    }
}

}  // namespace

extern "C" int64_t cvx_hubert_conv0_workspace_floats(int64_t L, int32_t C)
{
    return ((L + ST_ROWS - 1) / ST_ROWS + 2) * (int64_t)C;
}

extern "C" int cvx_hubert_conv0_gn_gelu_f32(const float* wav, int64_t n_samples, const float* w, int32_t C, int32_t k, int32_t stride,
                                            const float* gn_gamma, const float* gn_beta, float eps,
                                            float* out, float* workspace, int64_t workspace_floats, cvx_stream_t s)
{
    CVX_REQUIRE(wav && w && out && gn_gamma && gn_beta && workspace, "hubert_conv0: null pointer");
    CVX_REQUIRE(C > 0 && C % 4 == 0 && k > 0 && k <= 32 && stride > 0 && stride <= 8, "hubert_conv0: bad geometry C=%d k=%d stride=%d", C, k, stride);
    CVX_REQUIRE(n_samples >= k, "hubert_conv0: waveform shorter than one kernel (%ld samples)", (long)n_samples);
    const int64_t L = (n_samples - k) / stride + 1;
    CVX_REQUIRE(workspace_floats >= cvx_hubert_conv0_workspace_floats(L, C), "hubert_conv0: workspace too small");
    hipStream_t st = cvx_hip_stream(s);
    const int nchunks = (int)((L + ST_ROWS - 1) / ST_ROWS);
    float* partial = workspace;
    float* mean = workspace + (int64_t)nchunks * C;
    float* rstd = mean + C;
    const unsigned cb = (unsigned)((C + 255) / 256);
    hipLaunchKernelGGL(conv0_kernel, dim3(cb, (unsigned)((L + C0_FR - 1) / C0_FR)), dim3(256), 0, st, wav, w, out, L, C, k, stride);
    hipLaunchKernelGGL(colstat_partial_kernel, dim3(cb, nchunks), dim3(256), 0, st, out, (const float*)nullptr, partial, L, C);
    hipLaunchKernelGGL(colstat_final_kernel, dim3(cb), dim3(256), 0, st, partial, nchunks, C, L, mean, 0.f, 0);
    hipLaunchKernelGGL(colstat_partial_kernel, dim3(cb, nchunks), dim3(256), 0, st, out, (const float*)mean, partial, L, C);
    hipLaunchKernelGGL(colstat_final_kernel, dim3(cb), dim3(256), 0, st, partial, nchunks, C, L, rstd, eps, 1);
    const int64_t n4 = L * C / 4;
    hipLaunchKernelGGL(groupnorm_gelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, out, mean, rstd, gn_gamma, gn_beta, n4, C);
    CVX_CHECK_LAUNCH("cvx_hubert_conv0_gn_gelu_f32");
    return CVX_OK;
}

extern "C" int cvx_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                                 int64_t rows, int32_t D, float eps, cvx_stream_t s)
{
    CVX_REQUIRE(x && gamma && beta && y && rows >= 0, "layernorm: bad arguments");
    CVX_REQUIRE(D > 0 && D % 4 == 0 && D <= 1024, "layernorm: D must be a multiple of 4, at most 1024 (D=%d)", D);
    if (rows == 0) return CVX_OK;
    hipStream_t st = cvx_hip_stream(s);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (D <= 256)      hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, st, x, gamma, beta, y, rows, D, eps);
    else if (D <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, st, x, gamma, beta, y, rows, D, eps);
    else if (D <= 768) hipLaunchKernelGGL(layernorm_kernel<3>, grid, dim3(256), 0, st, x, gamma, beta, y, rows, D, eps);
    else               hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, x, gamma, beta, y, rows, D, eps);
    CVX_CHECK_LAUNCH("cvx_layernorm_f32");
    return CVX_OK;
}

extern "C" int cvx_hubert_group_pack_f32(const float* x, float* out, int32_t T, int32_t D, int32_t groups, int32_t halo, cvx_stream_t s)
{
    CVX_REQUIRE(x && out && T > 0 && D > 0 && groups > 0 && D % groups == 0 && halo >= 0, "hubert_group_pack: bad arguments");
    const int64_t n = (int64_t)(T + 2 * halo) * D;
    hipLaunchKernelGGL(group_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cvx_hip_stream(s), x, out, T, D, groups, halo);
    CVX_CHECK_LAUNCH("cvx_hubert_group_pack_f32");
    return CVX_OK;
}

extern "C" int cvx_kmeans_argmin_f32(const float* x, const float* dots, const float* cnorm, int64_t* labels, float* margin,
                                     int64_t T, int32_t D, int32_t K, cvx_stream_t s)
{
    CVX_REQUIRE(x && dots && cnorm && labels && T >= 0 && D > 0 && K > 0, "kmeans_argmin: bad arguments");
    if (T == 0) return CVX_OK;
    hipLaunchKernelGGL(kmeans_argmin_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, cvx_hip_stream(s),
                       x, dots, cnorm, labels, margin, T, D, K);
    CVX_CHECK_LAUNCH("cvx_kmeans_argmin_f32");
    return CVX_OK;
}

// ---------------------------------------------------------------- polyphase FIR resampler (the reader's sample-rate conversion)
// out[i*up + j] = sum_k kern[j][k] * x[i*down + k - width]   (x = 0 outside [0, n)),  i*up + j < n_out.
// hubert_feature_reader.py:38-41 resamples with torchaudio.transforms.Resample: a bank of `up` windowed-sinc filters of
// kw = 2*width + down taps applied at stride `down` (torchaudio.functional.resample, published algorithm; the filter
// bank itself is computed on the host).
namespace {
__global__ __launch_bounds__(256) void resample_fir_kernel(const float* __restrict__ x, const float* __restrict__ kern,
                                                           float* __restrict__ out, int64_t n, int64_t n_out,
                                                           int up, int down, int width, int kw)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int64_t i = o / up;
    const int j = (int)(o % up);
    const int64_t base = i * down - width;
    const float* kj = kern + (int64_t)j * kw;
    float acc = 0.f;
    for (int k = 0; k < kw; ++k) {
        const int64_t p = base + k;
        if (p >= 0 && p < n) acc = fmaf(kj[k], x[p], acc);
    }
    out[o] = acc;
}
}  // namespace

extern "C" int cvx_resample_fir_f32(const float* x, int64_t n, const float* kern, int32_t up, int32_t down, int32_t width,
                                    float* out, int64_t n_out, cvx_stream_t s)
{
    CVX_REQUIRE(x && kern && out && n >= 0 && n_out >= 0 && up > 0 && down > 0 && width >= 0, "resample_fir: bad arguments");
    if (n_out == 0) return CVX_OK;
    hipLaunchKernelGGL(resample_fir_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       x, kern, out, n, n_out, up, down, width, 2 * width + down);
    CVX_CHECK_LAUNCH("cvx_resample_fir_f32");
    return CVX_OK;
}

// ---------------------------------------------------------------- whole-network entry point
// HubertModel.extract_features(source, mask=False, output_layer) (hubert.py:433-480, 533-549) enqueued from ONE call:
// the ~330 launches of a 10-s utterance cost ~45 us each when stepped from Python (the GPU then idles two thirds of the
// time); from here they cost the bare launch.  Every GEMM runs on cvx_gemm_f16x3 with pre-split A operands (producers
// write the (hi, lo) pair, or cvx_split_f16 makes it), so small problems may split K.
namespace {

struct Bump {
    char* base; int64_t size, off;
    template <typename T> T* take(int64_t n)
    {
        const int64_t bytes = ((n * (int64_t)sizeof(T)) + 255) & ~(int64_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

struct Pair { uint16_t* hi; uint16_t* lo; };

struct HubertPlan {
    int64_t L[9];            // frames after conv layer i
    int T;
    // workspace carve-up (pointers are NULL in the sizing pass)
    float* x0; Pair x0s; float* c0ws; int64_t c0ws_floats;
    Pair cs[8]; float* feats; float* featn; Pair featns; float* h; float* packed; Pair packeds;
    float *x, *y, *qkv, *ff_dummy; Pair xs, atts, ffs; float* skws; int64_t skws_floats;
    int64_t bytes;
};

int plan_hubert(const cvx_hubert_model* m, int64_t n, void* ws, HubertPlan& P)
{
    CVX_REQUIRE(m && m->n_conv >= 2 && m->n_conv <= 8, "hubert: bad model (n_conv)");
    int64_t len = n;
    for (int i = 0; i < m->n_conv; ++i) {
        len = len >= m->conv_k[i] ? (len - m->conv_k[i]) / m->conv_stride[i] + 1 : 0;
        P.L[i] = len;
    }
    P.T = (int)len;
    CVX_REQUIRE(len > 0 && len < (1 << 24), "hubert: the waveform yields %ld frames", (long)len);
    const int C0 = m->conv_c[0], Cl = m->conv_c[m->n_conv - 1], D = m->dim, T = P.T;
    int F = 0, Nmax = 3 * D;
    for (int i = 0; i < m->n_layers; ++i) F = std::max(F, (int)m->layers[i].fc1.N);
    Nmax = std::max(Nmax, F);
    Bump b{reinterpret_cast<char*>(ws), 0, 0};
    P.x0 = b.take<float>(P.L[0] * C0);
    P.x0s = {b.take<uint16_t>(P.L[0] * C0), b.take<uint16_t>(P.L[0] * C0)};
    P.c0ws_floats = cvx_hubert_conv0_workspace_floats(P.L[0], C0);
    P.c0ws = b.take<float>(P.c0ws_floats);
    for (int i = 1; i < m->n_conv - 1; ++i)
        P.cs[i] = {b.take<uint16_t>(P.L[i] * m->conv_c[i]), b.take<uint16_t>(P.L[i] * m->conv_c[i])};
    P.feats = b.take<float>((int64_t)T * Cl);
    P.featn = b.take<float>((int64_t)T * Cl);
    P.featns = {b.take<uint16_t>((int64_t)T * Cl), b.take<uint16_t>((int64_t)T * Cl)};
    P.h = b.take<float>((int64_t)T * D);
    const int64_t prow = (int64_t)(T + 2 * (m->pos_k / 2)) * D;
    P.packed = b.take<float>(prow);
    P.packeds = {b.take<uint16_t>(prow), b.take<uint16_t>(prow)};
    P.x = b.take<float>((int64_t)T * D);
    P.y = b.take<float>((int64_t)T * D);
    P.qkv = b.take<float>((int64_t)T * 3 * D);
    P.xs = {b.take<uint16_t>((int64_t)T * D), b.take<uint16_t>((int64_t)T * D)};
    P.atts = {b.take<uint16_t>((int64_t)T * D), b.take<uint16_t>((int64_t)T * D)};
    P.ffs = {b.take<uint16_t>((int64_t)T * std::max(F, 1)), b.take<uint16_t>((int64_t)T * std::max(F, 1))};
    // split-K scratch: 4 * M * N floats for the largest GEMM with M <= 2048 rows (encoder GEMMs and the late conv layers)
    P.skws_floats = T <= 2048 ? (int64_t)4 * T * Nmax : 0;
    for (int i = 1; i < m->n_conv; ++i)
        if (P.L[i] <= 2048) P.skws_floats = std::max(P.skws_floats, (int64_t)4 * P.L[i] * m->conv_c[i]);
    P.skws = P.skws_floats ? b.take<float>(P.skws_floats) : nullptr;
    P.bytes = b.off;
    return CVX_OK;
}

// C[M,N] = epi(A . W^T): A as a pre-split pair with row stride lda_h; optional fp32 / split outputs
int lin(const cvx_linear& L, Pair a, int64_t lda_h, int M, float* C, int64_t ldc, int act, const float* bias,
        const float* residual, int64_t ldr, Pair c16, int write_f32, const HubertPlan& P, cvx_stream_t s)
{
    cvx_gemm_args g{};
    g.A = reinterpret_cast<const float*>(a.hi); g.lda = 4;          // validated only (A arrives pre-split)
    g.W = L.w; g.ldw = L.K;
    g.C = C; g.ldc = ldc;
    g.bias = bias; g.residual = residual; g.ldr = ldr;
    g.M = M; g.N = L.N; g.K = L.K; g.act = act;
    cvx_gemm_split_io io{};
    io.A_hi = a.hi; io.A_lo = a.lo; io.lda_h = lda_h;
    io.C_hi = c16.hi; io.C_lo = c16.lo; io.ldc_h = L.N;
    io.write_f32 = write_f32;
    if (M <= 2048) { io.workspace = P.skws; io.workspace_floats = P.skws_floats; }
    return cvx_gemm_f16x3(&g, L.w_hi, L.w_lo, L.inv_scale, &io, s);
}

}  // namespace

extern "C" int64_t cvx_hubert_workspace_bytes(const cvx_hubert_model* m, int64_t n_samples)
{
    HubertPlan P{};
    if (plan_hubert(m, n_samples, nullptr, P) != CVX_OK) return -1;
    return P.bytes;
}

extern "C" int32_t cvx_hubert_frames(const cvx_hubert_model* m, int64_t n_samples)
{
    if (!m) return -1;
    int64_t len = n_samples;
    for (int i = 0; i < m->n_conv; ++i) len = len >= m->conv_k[i] ? (len - m->conv_k[i]) / m->conv_stride[i] + 1 : 0;
    return (int32_t)len;
}

#define CVX_TRY(expr) do { const int rc_ = (expr); if (rc_ != CVX_OK) return rc_; } while (0)

extern "C" int cvx_hubert_extract_features(const cvx_hubert_model* m, const float* wav, int64_t n_samples, int32_t output_layer,
                                           float* out, void* workspace, int64_t workspace_bytes, cvx_stream_t s)
{
    CVX_REQUIRE(m && wav && out && workspace, "hubert_extract_features: null pointer");
    CVX_REQUIRE((((uintptr_t)workspace) & 255) == 0, "hubert_extract_features: workspace must be 256-byte aligned");
    CVX_REQUIRE(output_layer >= 0 && output_layer <= m->n_layers, "hubert_extract_features: output_layer %d out of range", output_layer);
    CVX_REQUIRE(m->dim == 64 * m->heads && m->dim % m->pos_groups == 0, "hubert_extract_features: head dim must be 64");
    HubertPlan P{};
    CVX_TRY(plan_hubert(m, n_samples, workspace, P));
    CVX_REQUIRE(workspace_bytes >= P.bytes, "hubert_extract_features: workspace too small (%ld < %ld)", (long)workspace_bytes, (long)P.bytes);
    const int T = P.T, D = m->dim, nc = m->n_conv, Cl = m->conv_c[nc - 1];
    const Pair none{nullptr, nullptr};
    // ---- conv feature extractor: layer 0 + GroupNorm + GELU, then GEMMs over overlapping channels-last rows
    CVX_TRY(cvx_hubert_conv0_gn_gelu_f32(wav, n_samples, m->conv0_w, m->conv_c[0], m->conv_k[0], m->conv_stride[0], m->gn_g, m->gn_b,
                                         1e-5f, P.x0, P.c0ws, P.c0ws_floats, s));
    CVX_TRY(cvx_split_f16(P.x0, P.x0s.hi, P.x0s.lo, P.L[0] * m->conv_c[0], 1.0f, s));
    Pair cur = P.x0s;
    for (int i = 1; i < nc; ++i) {
        const bool last = i == nc - 1;
        const int cin = m->conv_c[i - 1];
        CVX_REQUIRE(m->conv[i].K == m->conv_k[i] * cin && m->conv[i].N == m->conv_c[i], "hubert: conv layer %d weight shape", i);
        CVX_TRY(lin(m->conv[i], cur, (int64_t)m->conv_stride[i] * cin, (int)P.L[i], last ? P.feats : P.x0, m->conv_c[i], CVX_ACT_GELU,
                    nullptr, nullptr, 0, last ? none : P.cs[i], last ? 1 : 0, P, s));
        cur = P.cs[i];
    }
    // ---- LayerNorm, post_extract_proj
    CVX_TRY(cvx_layernorm_f32(P.feats, m->ln_g, m->ln_b, P.featn, T, Cl, 1e-5f, s));
    CVX_TRY(cvx_split_f16(P.featn, P.featns.hi, P.featns.lo, (int64_t)T * Cl, 1.0f, s));
    CVX_TRY(lin(m->proj, P.featns, Cl, T, P.h, D, CVX_ACT_NONE, m->proj.bias, nullptr, 0, none, 1, P, s));
    // ---- positional convolution: x = h + gelu(conv(h) + b), one GEMM per group over the packed operand
    const int G = m->pos_groups, cg = D / G, k = m->pos_k, halo = k / 2, prows = T + 2 * halo;
    CVX_TRY(cvx_hubert_group_pack_f32(P.h, P.packed, T, D, G, halo, s));
    CVX_TRY(cvx_split_f16(P.packed, P.packeds.hi, P.packeds.lo, (int64_t)prows * D, 1.0f, s));
    for (int g = 0; g < G; ++g) {
        const Pair a{P.packeds.hi + (int64_t)g * prows * cg, P.packeds.lo + (int64_t)g * prows * cg};
        CVX_REQUIRE(m->pos[g].K == k * cg && m->pos[g].N == cg, "hubert: positional conv group %d weight shape", g);
        // output / bias / residual are column slices of [T, D] tensors
        CVX_TRY(lin(m->pos[g], a, cg, T, P.x + g * cg, D, CVX_ACT_GELU, m->pos[g].bias, P.h + g * cg, D, none, 1, P, s));
    }
    CVX_TRY(cvx_layernorm_f32(P.x, m->enc_ln_g, m->enc_ln_b, P.x, T, D, 1e-5f, s));
    // ---- post-LN transformer layers
    for (int i = 0; i < output_layer; ++i) {
        const cvx_hubert_layer& ly = m->layers[i];
        CVX_TRY(cvx_split_f16(P.x, P.xs.hi, P.xs.lo, (int64_t)T * D, 1.0f, s));
        CVX_TRY(lin(ly.qkv, P.xs, D, T, P.qkv, 3 * D, CVX_ACT_NONE, ly.qkv.bias, nullptr, 0, none, 1, P, s));
        CVX_TRY(cvx_attention_f32(P.qkv, nullptr, P.atts.hi, P.atts.lo, 1, T, m->heads, 0.125f, s));
        CVX_TRY(lin(ly.out, P.atts, D, T, P.y, D, CVX_ACT_NONE, ly.out.bias, P.x, D, none, 1, P, s));
        CVX_TRY(cvx_layernorm_f32(P.y, ly.ln1_g, ly.ln1_b, P.x, T, D, 1e-5f, s));
        CVX_TRY(cvx_split_f16(P.x, P.xs.hi, P.xs.lo, (int64_t)T * D, 1.0f, s));
        CVX_TRY(lin(ly.fc1, P.xs, D, T, P.y, ly.fc1.N, CVX_ACT_GELU, ly.fc1.bias, nullptr, 0, P.ffs, 0, P, s));
        CVX_TRY(lin(ly.fc2, P.ffs, ly.fc1.N, T, P.y, D, CVX_ACT_NONE, ly.fc2.bias, P.x, D, none, 1, P, s));
        CVX_TRY(cvx_layernorm_f32(P.y, ly.ln2_g, ly.ln2_b, P.x, T, D, 1e-5f, s));
    }
    CVX_REQUIRE(hipMemcpyAsync(out, P.x, (size_t)T * D * sizeof(float), hipMemcpyDeviceToDevice, cvx_hip_stream(s)) == hipSuccess,
                "hubert_extract_features: output copy failed");
    return CVX_OK;
}
