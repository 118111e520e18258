// Memory-bound kernels of the CoVoMix vector field + library-wide error plumbing (gfx950).
//   cvx_adarmsnorm_f32          AdaptiveRMSNorm / RMSNorm       (acoustic.py:165-204)
//   cvx_dwconv31_gelu_res_f32   ConvPositionEmbed + residual    (acoustic.py:141-161, :508)
//   cvx_cfg_combine_axpy_f32    CFG combine + ODE stage update  (acoustic.py:428; torchdiffeq midpoint)
//   cvx_embed_gather_f32        step-invariant to_embed columns (acoustic.py:473-503)
//   cvx_time_fourier_f32        LearnedSinusoidalPosEmb         (acoustic.py:107-111)
//   cvx_wav_to_int16            mel_decode_to_wav tail          (monologue_generation.py:55-57)
// All are HBM-bound: 16-byte coalesced accesses over the channel axis, wavefront (64-lane)
// shuffles for the row reduction, no LDS needed.
#include "cvx_common.h"
#include <algorithm>
#include <stdarg.h>
#include <string.h>

// ---------------------------------------------------------------- error string
static thread_local char g_err[512] = "";
void cvx_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* cvx_last_error_string(void) { return g_err; }

#include <mutex>
#include <map>
#include <set>
#include <utility>
void cvx_allow_dynamic_lds(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> granted;       // largest size asked for so far, per (device, kernel)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    int& g = granted[std::make_pair(dev, kernel)];
    if (bytes > g) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        g = bytes;
    }
}
int cvx_device_cus()
{
    static std::mutex mu;
    static int n_cu[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (n_cu[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev] = v;
    }
    return n_cu[dev];
}
extern "C" int cvx_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, void** out)
{
    CVX_REQUIRE(mask && out && n_words > 0 && n_words <= 64, "stream_create_cu_mask: bad arguments");
    int bits = 0;
    for (int i = 0; i < n_words; ++i) bits += __builtin_popcount(mask[i]);
    CVX_REQUIRE(bits > 0, "stream_create_cu_mask: empty mask");
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask);
    if (e != hipSuccess) { cvx_set_error("stream_create_cu_mask: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e)); return CVX_EHIP; }
    *out = reinterpret_cast<void*>(st);
    return CVX_OK;
}
extern "C" int cvx_stream_destroy(void* stream)
{
    CVX_REQUIRE(stream, "stream_destroy: the NULL stream");
    const hipError_t e = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) { cvx_set_error("stream_destroy: %s", hipGetErrorString(e)); return CVX_EHIP; }
    return CVX_OK;
}
extern "C" int cvx_saturation_flag_reset(cvx_stream_t s)
{
    uint32_t* f = cvx_sat_flag_for(s);
    CVX_REQUIRE(f && (reinterpret_cast<uintptr_t>(f) & 3) == 0, "saturation_flag_reset: the context carries no (4-byte aligned) saturation flag");
    if (hipMemsetAsync(f, 0, sizeof(uint32_t), cvx_hip_stream(s)) != hipSuccess) { cvx_set_error("saturation_flag: memset failed"); return CVX_EHIP; }
    return CVX_OK;
}
extern "C" int cvx_saturation_flag_query(uint32_t* host_out, int32_t reset, cvx_stream_t s)
{
    uint32_t* f = cvx_sat_flag_for(s);
    CVX_REQUIRE(f && host_out, "saturation_flag_query: the context carries no saturation flag / null output");
    hipStream_t st = cvx_hip_stream(s);
    if (hipMemcpyAsync(host_out, f, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { cvx_set_error("saturation_flag: read failed: %s", hipGetErrorString(hipGetLastError())); return CVX_EHIP; }
    if (reset && hipMemsetAsync(f, 0, sizeof(uint32_t), st) != hipSuccess) { cvx_set_error("saturation_flag: memset failed"); return CVX_EHIP; }
    return CVX_OK;
}
// Clock stamps (bench.py): {shader-clock cycle counter, 100 MHz real-time counter} PER COMPUTE UNIT: the cycle counter belongs to the CU
// (stamps of two CUs of one XCD differ by arbitrary offsets - a per-XCD slot gave negative clocks over short regions), so a slot is
// (XCD, the CU / shader-array / shader-engine bits of HW_ID) and the caller pairs the two stamps of the SAME slot.  Two calls around a
// busy region give the shader clock it ran at, per CU; the hwmon file shows one XCD's momentary value, and config 2's time differs
// between boxes that report the same one.  4096 short blocks: the dispatcher spreads them over (nearly) every CU.
namespace {
__global__ __launch_bounds__(64) void clock_stamp_kernel(unsigned long long* __restrict__ out)
{
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    __builtin_amdgcn_s_sleep(64);                                        // (keeps the block resident long enough for the others to land elsewhere)
    const unsigned long long t = __builtin_readcyclecounter(), r = __builtin_amdgcn_s_memrealtime();
    const unsigned slot = ((xcc & 7) << 8) | ((hwid >> 8) & 0xFF);      // HW_ID[15:8] = CU_ID, SH_ID, SE_ID
    if (threadIdx.x == 0) { out[2 * slot] = t; out[2 * slot + 1] = r; }
}
}
extern "C" int cvx_clock_stamps(uint64_t* stamps_dev, cvx_stream_t s)
{
    CVX_REQUIRE(stamps_dev && (reinterpret_cast<uintptr_t>(stamps_dev) & 7) == 0, "clock_stamps: null / unaligned output");
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(4096), dim3(64), 0, cvx_hip_stream(s), reinterpret_cast<unsigned long long*>(stamps_dev));
    CVX_CHECK_LAUNCH("cvx_clock_stamps");
    return CVX_OK;
}
extern "C" int cvx_version(void) { return CVX_ABI_VERSION; }

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
#ifndef CVX_NORM_NT
#define CVX_NORM_NT 0          // dev A/B: bit 0 = non-temporal loads of x, bit 1 = non-temporal stores of the split pair (AdaRMSNorm)
#endif
// split 4 floats into fp16 (hi, lo) and store 8 bytes each (for GEMMs that take their A operand pre-split)
__device__ __forceinline__ void store_split4(_Float16* hi, _Float16* lo, int64_t off, const f32x4 o, CvxSat& amax)
{
    cvx_amax4(amax, o);
    // lo == hi + 32: INTERLEAVED pair, [hi 32 | lo 32] per block of 32 values (one 128-byte line per K-step and row for
    // the consumer GEMM's DMA); the mapping is a function of the flat offset because every row is a multiple of 32 wide
    if (lo == hi + 32) off = ((off >> 5) << 6) | (off & 31);
    f16x4_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = fminf(fmaxf(o[e], -65504.f), 65504.f);
        h[e] = (_Float16)x;
        l[e] = (_Float16)(x - (float)h[e]);
    }
    if (CVX_NORM_NT & 2) {
        __builtin_nontemporal_store(h, reinterpret_cast<f16x4_t*>(hi + off));
        if (lo) __builtin_nontemporal_store(l, reinterpret_cast<f16x4_t*>(lo + off));
        return;
    }
    *reinterpret_cast<f16x4_t*>(hi + off) = h;
    if (lo) *reinterpret_cast<f16x4_t*>(lo + off) = l;      // lo == NULL: hi halves only
}

// ---------------------------------------------------------------- AdaRMSNorm
// one wavefront per row; the row stays in registers when D <= 256*NV.
template <int NV>
__global__ __launch_bounds__(256) void adarmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo,
                                                        int64_t rows, int D, int64_t rows_per_group, float scale, float eps,
                                                        const float* __restrict__ split_scale, uint32_t* __restrict__ sat)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    CvxSat amax;
    const float ssc = split_scale ? *split_scale : 1.f;      // power-of-two pre-scale of the split copy (device scalar)
    const float* xr = x + row * D;
    const int64_t g = row / rows_per_group;
    const float* gr = gamma + g * D;
    const float* br = beta ? beta + g * D : nullptr;
    float* yr = y ? y + row * D : nullptr;
    const int nvec = D >> 2;

    // every load of the row's pass is requested up front - x, and the item's gamma / beta rows (L2 hits) that used to be
    // requested only behind the wave reduction, one exposed round trip per row later
    f32x4 v[NV], gv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nvec) v[i] = (CVX_NORM_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 4 * j)) : gload4(xr + 4 * j);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nvec) {
            gv[i] = gload4(gr + 4 * j);
            if (br) bv[i] = gload4(br + 4 * j);
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nvec) ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    ss = wave_sum(ss);
    const float inv = scale / fmaxf(sqrtf(ss), eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nvec) {
            const f32x4 gg = gv[i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[i][e] * inv * gg[e];
            if (br) {
                const f32x4 bb = bv[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += bb[e];
            }
            if (y) *reinterpret_cast<f32x4*>(yr + 4 * j) = o;
            if (y_hi) { const f32x4 os = {o[0] * ssc, o[1] * ssc, o[2] * ssc, o[3] * ssc}; store_split4(y_hi, y_lo, row * D + 4 * j, os, amax); }
        }
    }
    cvx_sat_commit(sat, amax);
}

// generic-D fallback: two passes over the row (second pass hits L1/L2)
__global__ __launch_bounds__(256) void adarmsnorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo,
                                                                int64_t rows, int D, int64_t rows_per_group, float scale, float eps,
                                                                const float* __restrict__ split_scale, uint32_t* __restrict__ sat)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    CvxSat amax;
    const float ssc = split_scale ? *split_scale : 1.f;
    const float* xr = x + row * D;
    const int64_t g = row / rows_per_group;
    float ss = 0.f;
    for (int j = lane; j < (D >> 2); j += 64) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(xr + 4 * j);
        ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
    }
    ss = wave_sum(ss);
    const float inv = scale / fmaxf(sqrtf(ss), eps);
    for (int j = lane; j < (D >> 2); j += 64) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(xr + 4 * j);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(gamma + g * D + 4 * j);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = t[e] * inv * gg[e];
        if (beta) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + g * D + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += bb[e];
        }
        if (y) *reinterpret_cast<f32x4*>(y + row * D + 4 * j) = o;
        if (y_hi) { const f32x4 os = {o[0] * ssc, o[1] * ssc, o[2] * ssc, o[3] * ssc}; store_split4(y_hi, y_lo, row * D + 4 * j, os, amax); }
    }
    cvx_sat_commit(sat, amax);
}

// ---------------------------------------------------------------- depthwise conv k=31 + GELU + residual
// channels-last [Bt,T,C]: a thread owns one channel and TT consecutive frames; every global
// access is a 256-byte coalesced row segment across the block's 64 channels x 4... (one wave = 64 channels).
#ifndef CVX_DWCONV_ERFF
#define CVX_DWCONV_ERFF 0
#endif
constexpr int DW_K = 31;
#ifndef CVX_DW_TT
#define CVX_DW_TT 16
#endif
constexpr int DW_TT = CVX_DW_TT;
__global__ __launch_bounds__(256) void dwconv31_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int T, int C, const int* __restrict__ cu_seqlens)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int t0 = blockIdx.y * DW_TT;
    int64_t base = (int64_t)blockIdx.z * T * C;
    if (cu_seqlens) {                 // ragged batch: sequence z = rows [cu[z], cu[z+1]) of the packed tensor, zero-padded at ITS ends
        const int r0 = cu_seqlens[blockIdx.z];
        T = cu_seqlens[blockIdx.z + 1] - r0;
        base = (int64_t)r0 * C;
        if (t0 >= T) return;          // block-uniform: the grid is sized for the longest sequence
    }
    float wk[DW_K];
#pragma unroll
    for (int k = 0; k < DW_K; ++k) wk[k] = w[c * DW_K + k];
    // window of DW_TT + 30 frames held in registers; every index below is a compile-time constant
    float xs[DW_TT + DW_K - 1];
#pragma unroll
    for (int i = 0; i < DW_TT + DW_K - 1; ++i) {
        const int t = t0 + i - DW_K / 2;
        xs[i] = (t >= 0 && t < T) ? x[base + (int64_t)t * C + c] : 0.f;
    }
    const float bc = bias[c];
    // two outputs at a time through the branch-free GELU of the GEMM epilogues (1-ulp erf, packed fp32): the library erff was
    // two thirds of this kernel's VALU work (CVX_DWCONV_ERFF=1 in a dev build: the library form)
#pragma unroll
    for (int o = 0; o < DW_TT; o += 2) {
        float a0 = bc, a1 = bc;
#pragma unroll
        for (int k = 0; k < DW_K; ++k) { a0 = fmaf(wk[k], xs[o + k], a0); a1 = fmaf(wk[k], xs[o + 1 + k], a1); }
#if CVX_DWCONV_ERFF
        const f32x2 g2 = f32x2{gelu_erf(a0), gelu_erf(a1)};
#else
        const f32x2 g2 = gelu_fast2(f32x2{a0, a1});
#endif
        const int t = t0 + o;
        if (t < T) y[base + (int64_t)t * C + c] = g2[0] + xs[o + DW_K / 2];
        if (t + 1 < T) y[base + (int64_t)(t + 1) * C + c] = g2[1] + xs[o + 1 + DW_K / 2];
    }
}

// ---------------------------------------------------------------- skinny GEMM: C[M <= 32][N] = A[M][K] . W[N][K]^T (+ bias)
// A product with a handful of rows is weight streaming (the adaptive-norm table of a solve - 32 evaluation times x 32,768
// outputs, K = 4,096 - took 1.35 ms on the tiled fp32 kernel and 0.79 ms on the split-precision one: 537 MB of weights for 8.6
// GFLOP).  Here every wave owns tiles of 32 W rows: lane (r, g) streams 16-byte pieces of ITS row (non-temporal, SK_Q steps
// requested ahead in registers), all M rows of A sit in LDS ([m][K + 4] fp32: conflict-free b128 reads), and the tile is
// 32 x 32 outputs of v_mfma_f32_32x32x2_f32 - exact fp32 products; one 16-byte piece of W and of A feeds four MFMAs, the K
// order being permuted consistently in both operands (lanes g = 0 / 1 take k = 8q + j / 8q + 4 + j in step j).
constexpr int SK_M = 32;
constexpr int SK_KC = 1024;                    // columns of A staged in LDS at a time (longer K: chunk by chunk, accumulators stay)
constexpr int SK_Q = 16;                        // K steps (of 8) requested ahead: 16 x 4 MFMAs = 4096 matrix cycles per round trip
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, float* __restrict__ Cm, int64_t ldc,
                                                         int M, int N, int K, int act)
{
    extern __shared__ __attribute__((aligned(16))) float sk_a[];          // [SK_M][KC + 4] (one K chunk of A), rows >= M zero
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int KC = min(K, SK_KC), ldA = KC + 4;
    const int i31 = lane & 31, g = lane >> 5;
    const float* ar = sk_a + (size_t)i31 * ldA + 4 * g;                   // this lane's A row (m = i31), its half of every 8 k
    const int n_tiles = (N + 31) / 32;
    for (int tile0 = blockIdx.x * 4; tile0 < n_tiles; tile0 += gridDim.x * 4) {       // (block-uniform: the chunk staging has barriers)
        const int tile = tile0 + wid;
        const bool active = tile < n_tiles;
        const int n0 = tile * 32;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        for (int kc = 0; kc < K; kc += KC) {
            const int kn = min(KC, K - kc), k4 = kn >> 2, nq = kn >> 3;  // K % 8 == 0
            __syncthreads();                                              // the previous chunk (or tile round) is consumed
            for (int i = tid; i < SK_M * k4; i += 256) {
                const int m = i / k4, c = i - m * k4;
                *reinterpret_cast<f32x4*>(sk_a + (size_t)m * ldA + 4 * c) = m < M ? gload4(A + (size_t)m * lda + kc + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
            if (!active) continue;
            const float* wr = W + (int64_t)min(n0 + i31, N - 1) * ldw + kc + 4 * g;
            f32x4 wb[2][SK_Q];
#pragma unroll
            for (int u = 0; u < SK_Q; ++u) wb[0][u] = u < nq ? gload4_nt(wr + 8 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
            for (int q0 = 0; q0 < nq; q0 += 2 * SK_Q) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int qb = q0 + half * SK_Q;
                    if (qb < nq) {
#pragma unroll
                        for (int u = 0; u < SK_Q; ++u)                    // the next SK_Q pieces of the row
                            wb[half ^ 1][u] = qb + SK_Q + u < nq ? gload4_nt(wr + 8 * (qb + SK_Q + u)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < SK_Q; ++u) {
                            if (qb + u < nq) {
                                const f32x4 av = *reinterpret_cast<const f32x4*>(ar + 8 * (qb + u));
                                const f32x4 wv = wb[half][u];
                                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[0], av[0], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[1], av[1], acc1, 0, 0, 0);
                                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[2], av[2], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[3], av[3], acc1, 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        if (!active) continue;
        // acc[r]: W row n0 + (r & 3) + 8 (r >> 2) + 4 g, A row m = i31
        if (i31 < M) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (n < N) {
                    float v = acc0[r] + acc1[r];
                    if (bias) v += bias[n];
                    if (act == CVX_ACT_SILU) v = silu(v);
                    else if (act == CVX_ACT_GELU) v = gelu_erf(v);
                    else if (act == CVX_ACT_TANH) v = tanhf(v);
                    Cm[(int64_t)i31 * ldc + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- CFG combine + axpy
__global__ __launch_bounds__(256) void cfg_axpy_kernel(const float* __restrict__ fc, const float* __restrict__ fn,
                                                      const float* y, float s, float coef,   // out may alias y
                                                      float* out, float* __restrict__ out2,
                                                      float* __restrict__ out3, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = fc[i];
    if (fn) v = v * (1.0f + s) - s * fn[i];
    const float o = y[i] + v * coef;
    out[i] = o;
    if (out2) out2[i] = o;
    if (out3) out3[i] = o;
}

// ---------------------------------------------------------------- embedding gather
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ ids, int S,
                                                          const float* __restrict__ table, int E, int n_rows_table,
                                                          const float* __restrict__ cond, const float* __restrict__ cond_row,
                                                          int Cc, int64_t null_id, float* __restrict__ out)
{
    const int64_t m = blockIdx.x;
    const int width = S * E + Cc;
    float* o = out + m * width;
    for (int j = threadIdx.x; j < width; j += 256) {
        float v;
        if (j < S * E) {
            const int s = j / E, e = j - s * E;
            int64_t id = ids ? ids[m * S + s] : null_id;
            id = id < 0 ? 0 : (id >= n_rows_table ? n_rows_table - 1 : id);
            v = table[id * E + e];
        } else {
            const int cidx = j - S * E;
            v = cond_row ? cond_row[cidx] : cond[m * Cc + cidx];
        }
        o[j] = v;
    }
}

// ---------------------------------------------------------------- time Fourier features
__global__ __launch_bounds__(256) void time_fourier_kernel(const float* __restrict__ times, const float* __restrict__ w,
                                                          float* __restrict__ out, int half)
{
    const int i = blockIdx.x;
    const float t = times[i];
    for (int j = threadIdx.x; j < half; j += 256) {
        const float ang = t * w[j] * 2.0f * 3.14159265358979323846f;
        out[(int64_t)i * 2 * half + j] = sinf(ang);
        out[(int64_t)i * 2 * half + half + j] = cosf(ang);
    }
}

// ---------------------------------------------------------------- float wav -> int16 PCM
__global__ __launch_bounds__(256) void wav_to_int16_kernel(const float* __restrict__ wav, int16_t* __restrict__ pcm, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = wav[i] * 32768.0f;
    pcm[i] = (int16_t)(int32_t)v;     // trunc toward zero, wrap like numpy astype on x86
}

// ---------------------------------------------------------------- prompt mel extraction (row N3) epilogues
// spec [T, 2*nb] = (re | im) of the windowed DFT  ->  mag [T, nbp] = sqrt(re^2 + im^2 + 1e-9), zero in the padding
__global__ __launch_bounds__(256) void mel_magnitude_kernel(const float* __restrict__ spec, float* __restrict__ mag,
                                                           int64_t T, int nb, int nbp)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * nbp) return;
    const int64_t t = i / nbp;
    const int k = (int)(i - t * nbp);
    float v = 0.f;
    if (k < nb) {
        const float re = spec[t * 2 * nb + k], im = spec[t * 2 * nb + nb + k];
        v = sqrtf(re * re + im * im + 1e-9f);
    }
    mag[i] = v;
}

// x [T, n_mels] -> y [n_mels, T] = log(max(x, 1e-5))
__global__ __launch_bounds__(256) void mel_log_transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t T, int n_mels)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * n_mels) return;
    const int m = (int)(i / T);
    const int64_t t = i - (int64_t)m * T;
    y[i] = logf(fmaxf(x[t * n_mels + m], 1e-5f));
}

}  // namespace

extern "C" int cvx_mel_magnitude_f32(const float* spec, float* mag, int64_t T, int32_t nb, int32_t nbp, cvx_stream_t s)
{
    CVX_REQUIRE(spec && mag && T >= 0 && nb > 0 && nbp >= nb, "mel_magnitude: bad arguments");
    if (T == 0) return CVX_OK;
    hipLaunchKernelGGL(mel_magnitude_kernel, dim3((unsigned)((T * nbp + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       spec, mag, T, nb, nbp);
    CVX_CHECK_LAUNCH("cvx_mel_magnitude_f32");
    return CVX_OK;
}

extern "C" int cvx_mel_log_transpose_f32(const float* x, float* y, int64_t T, int32_t n_mels, cvx_stream_t s)
{
    CVX_REQUIRE(x && y && T >= 0 && n_mels > 0, "mel_log_transpose: bad arguments");
    if (T == 0) return CVX_OK;
    hipLaunchKernelGGL(mel_log_transpose_kernel, dim3((unsigned)((T * n_mels + 255) / 256)), dim3(256), 0,
                       cvx_hip_stream(s), x, y, T, n_mels);
    CVX_CHECK_LAUNCH("cvx_mel_log_transpose_f32");
    return CVX_OK;
}

extern "C" int cvx_adarmsnorm_scaled_f32(const float* x, const float* gamma, const float* beta, float* y,
                                  uint16_t* y_hi_, uint16_t* y_lo_,
                                  int64_t rows, int32_t D, int64_t rows_per_group, float scale, float eps,
                                  const float* split_scale_dev, cvx_stream_t s)
{
    CVX_REQUIRE(x && gamma && (y || y_hi_) && (y_hi_ || !y_lo_), "adarmsnorm: null pointer");     // y_lo == NULL: hi halves only
    _Float16* y_hi = reinterpret_cast<_Float16*>(y_hi_);
    _Float16* y_lo = reinterpret_cast<_Float16*>(y_lo_);
    CVX_REQUIRE(rows >= 0 && D > 0 && D % 4 == 0 && rows_per_group > 0, "adarmsnorm: bad shape rows=%ld D=%d", (long)rows, D);
    if (rows == 0) return CVX_OK;
    hipStream_t st = cvx_hip_stream(s);
    dim3 grid((unsigned)((rows + 3) / 4));
    if (y_hi) CVX_REQUIRE_SAT(s);
    uint32_t* sat = y_hi ? cvx_sat_flag_for(s) : nullptr;
    if (D <= 256)       hipLaunchKernelGGL(adarmsnorm_kernel<1>, grid, dim3(256), 0, st, x, gamma, beta, y, y_hi, y_lo, rows, D, rows_per_group, scale, eps, split_scale_dev, sat);
    else if (D <= 512)  hipLaunchKernelGGL(adarmsnorm_kernel<2>, grid, dim3(256), 0, st, x, gamma, beta, y, y_hi, y_lo, rows, D, rows_per_group, scale, eps, split_scale_dev, sat);
    else if (D <= 1024) hipLaunchKernelGGL(adarmsnorm_kernel<4>, grid, dim3(256), 0, st, x, gamma, beta, y, y_hi, y_lo, rows, D, rows_per_group, scale, eps, split_scale_dev, sat);
    else                hipLaunchKernelGGL(adarmsnorm_generic_kernel, grid, dim3(256), 0, st, x, gamma, beta, y, y_hi, y_lo, rows, D, rows_per_group, scale, eps, split_scale_dev, sat);
    CVX_CHECK_LAUNCH("cvx_adarmsnorm_scaled_f32");
    return CVX_OK;
}

extern "C" int cvx_adarmsnorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                                  uint16_t* y_hi, uint16_t* y_lo,
                                  int64_t rows, int32_t D, int64_t rows_per_group, float scale, float eps,
                                  cvx_stream_t s)
{
    return cvx_adarmsnorm_scaled_f32(x, gamma, beta, y, y_hi, y_lo, rows, D, rows_per_group, scale, eps, nullptr, s);
}

// ---------------------------------------------------------------- deferred norm: per-row factor from a producer GEMM's partial sums
// (cvx_gemm_split_io.c_rowsq: one sum of squares per 64-column slice of a row; reference acoustic.py:198-204, F.normalize's eps)
namespace {
__global__ __launch_bounds__(256) void rownorm_scale_kernel(const float* __restrict__ rowsq, int64_t rows, int parts, int64_t ld,
                                                            float scale, float eps, float* __restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* p = rowsq + r * ld;
    float q = 0.f;
    if ((parts & 3) == 0 && (ld & 3) == 0) {
        for (int j = 0; j < parts; j += 4) { const f32x4 v = gload4(p + j); q += v[0]; q += v[1]; q += v[2]; q += v[3]; }
    } else {
        for (int j = 0; j < parts; ++j) q += p[j];
    }
    out[r] = scale / fmaxf(sqrtf(q), eps);
}
}  // namespace

extern "C" int cvx_rownorm_scale_f32(const float* rowsq, int64_t rows, int32_t parts, int64_t ld, float scale, float eps, float* out, cvx_stream_t s)
{
    CVX_REQUIRE(rowsq && out && rows >= 0 && parts > 0 && parts <= 64 && ld >= parts && (((uintptr_t)rowsq & 15) == 0 || (parts & 3) || (ld & 3)),
                "rownorm_scale: bad arguments (rows=%ld parts=%d ld=%ld)", (long)rows, parts, (long)ld);
    if (rows == 0) return CVX_OK;
    hipLaunchKernelGGL(rownorm_scale_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       rowsq, rows, parts, ld, scale, eps, out);
    CVX_CHECK_LAUNCH("cvx_rownorm_scale_f32");
    return CVX_OK;
}

// ---------------------------------------------------------------- measured power-of-two pre-scale of a tensor
namespace {
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ amax_bits)
{
    // four independent 16-byte loads per thread and trip; ONE atomic per block, and only from blocks that can still raise the
    // maximum (2048 x 4 same-address atomics used to be most of this kernel: 100 us for 164 MB)
    __shared__ float part[4];
    float m = 0.f;
    const int64_t n4 = n / 4, stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gload4(x + 4 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
    }
    for (; i < n4; i += stride) {
        const f32x4 v = gload4(x + 4 * i);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (m > 0.f && m < __builtin_inff() && __float_as_uint(m) > __atomic_load_n(amax_bits, __ATOMIC_RELAXED))
            atomicMax(amax_bits, __float_as_uint(m));        // non-negative floats order like their bits
    }
}
__global__ void pow2_scale_kernel(const unsigned* __restrict__ amax_bits, float target, float* __restrict__ scale)
{
    const float a = __uint_as_float(*amax_bits);
    float e = 0.f;
    if (a > 0.f) e = fminf(fmaxf(rintf(log2f(target / a)), -40.f), 40.f);
    *scale = exp2f(e);
}
}  // namespace

extern "C" int cvx_amax_pow2_scale_f32(const float* x, int64_t n, float target, float* scale_dev, uint32_t* scratch_dev, cvx_stream_t s)
{
    CVX_REQUIRE(x && scale_dev && scratch_dev && n >= 0 && target > 0.f && ((uintptr_t)x & 15) == 0, "amax_pow2_scale: bad arguments");
    hipStream_t st = cvx_hip_stream(s);
    if (hipMemsetAsync(scratch_dev, 0, sizeof(uint32_t), st) != hipSuccess) { cvx_set_error("amax_pow2_scale: memset failed"); return CVX_EHIP; }
    if (n > 0) {
        const unsigned blocks = (unsigned)((n / 4 + 1023) / 1024 < 2048 ? (n / 4 + 1023) / 1024 + 1 : 2048);
        hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, st, x, n, scratch_dev);
    }
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, st, scratch_dev, target, scale_dev);
    // leave the scratch word ZERO, as cvx_pow2_scale_from_amax_f32 does: a producer kernel that max-accumulates into the same word
    // afterwards (the vocoder's upsampler of the NEXT call) must not see this call's maximum (round 4: it did - the first call of
    // a shape and the later ones differed by an fp32 rounding, and a call's scale depended on the previous call's input)
    if (hipMemsetAsync(scratch_dev, 0, sizeof(uint32_t), st) != hipSuccess) { cvx_set_error("amax_pow2_scale: memset failed"); return CVX_EHIP; }
    CVX_CHECK_LAUNCH("cvx_amax_pow2_scale_f32");
    return CVX_OK;
}

extern "C" int cvx_pow2_scale_from_amax_f32(uint32_t* amax_bits_dev, float target, float* scale_dev, cvx_stream_t s)
{
    CVX_REQUIRE(amax_bits_dev && scale_dev && target > 0.f, "pow2_scale_from_amax: bad arguments");
    hipStream_t st = cvx_hip_stream(s);
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, st, amax_bits_dev, target, scale_dev);
    if (hipMemsetAsync(amax_bits_dev, 0, sizeof(uint32_t), st) != hipSuccess) { cvx_set_error("pow2_scale_from_amax: memset failed"); return CVX_EHIP; }
    CVX_CHECK_LAUNCH("cvx_pow2_scale_from_amax_f32");
    return CVX_OK;
}

extern "C" int cvx_dwconv31_gelu_res_varlen_f32(const float* x, const float* w, const float* bias, float* y,
                                                const int32_t* cu_seqlens_dev, int32_t Bt, int32_t max_T, int32_t C, cvx_stream_t s)
{
    CVX_REQUIRE(x && w && bias && y, "dwconv31: null pointer");
    CVX_REQUIRE(Bt >= 0 && max_T > 0 && C > 0 && Bt <= 65535, "dwconv31: bad shape");
    CVX_REQUIRE(x != y, "dwconv31: in-place operation is not supported");
    if (Bt == 0) return CVX_OK;
    dim3 grid((C + 255) / 256, (max_T + DW_TT - 1) / DW_TT, Bt);
    hipLaunchKernelGGL(dwconv31_kernel, grid, dim3(256), 0, cvx_hip_stream(s), x, w, bias, y, max_T, C, cu_seqlens_dev);
    CVX_CHECK_LAUNCH("cvx_dwconv31_gelu_res_f32");
    return CVX_OK;
}

extern "C" int cvx_gemm_skinny_f32(const float* A, int32_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                                   int32_t M, int32_t N, int32_t K, int32_t act, cvx_stream_t s)
{
    CVX_REQUIRE(A && W && C && M >= 0 && M <= SK_M && N >= 0 && K > 0 && K % 8 == 0 && lda >= K && lda % 4 == 0 && ldw >= K &&
                ldw % 4 == 0 && ldc >= N && act >= CVX_ACT_NONE && act <= CVX_ACT_TANH,
                "gemm_skinny: M <= 32, K a multiple of 8, lda / ldw multiples of 4 (M=%d N=%d K=%d)", M, N, K);
    CVX_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm_skinny: 16-byte aligned operands");
    if (M == 0 || N == 0) return CVX_OK;
    const size_t lds = (size_t)SK_M * (std::min(K, SK_KC) + 4) * sizeof(float);
    cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_skinny_kernel), (int)lds);
    const int64_t groups = ((int64_t)N + 127) / 128;                       // four 32-row tiles per block and trip
    const unsigned grid = (unsigned)std::min<int64_t>(groups, (int64_t)cvx_ctx_cus(s));
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3(grid), dim3(256), lds, cvx_hip_stream(s), A, lda, W, ldw, bias, C, ldc, M, N, K, act);
    CVX_CHECK_LAUNCH("cvx_gemm_skinny_f32");
    return CVX_OK;
}

extern "C" int cvx_dwconv31_gelu_res_f32(const float* x, const float* w, const float* bias, float* y,
                                         int32_t Bt, int32_t T, int32_t C, cvx_stream_t s)
{
    return cvx_dwconv31_gelu_res_varlen_f32(x, w, bias, y, nullptr, Bt, T, C, s);
}

extern "C" int cvx_cfg_combine_axpy_f32(const float* f_c, const float* f_n, const float* y, float cond_scale,
                                        float coef, float* out, float* out2, float* out3, int64_t n, cvx_stream_t s)
{
    CVX_REQUIRE(f_c && y && out && n >= 0, "cfg_axpy: bad arguments");
    if (n == 0) return CVX_OK;
    hipLaunchKernelGGL(cfg_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       f_c, f_n, y, cond_scale, coef, out, out2, out3, n);
    CVX_CHECK_LAUNCH("cvx_cfg_combine_axpy_f32");
    return CVX_OK;
}

extern "C" int cvx_embed_gather_f32(const int64_t* ids, int32_t S, const float* table, int32_t E, int32_t n_rows_table,
                                    const float* cond, const float* cond_row, int32_t Cc, int64_t null_id,
                                    float* out, int64_t M, cvx_stream_t s)
{
    CVX_REQUIRE(table && out && S > 0 && E > 0 && Cc >= 0 && M >= 0, "embed_gather: bad arguments");
    CVX_REQUIRE(cond || cond_row || Cc == 0, "embed_gather: cond and cond_row both null");
    if (M == 0) return CVX_OK;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)M), dim3(256), 0, cvx_hip_stream(s),
                       ids, S, table, E, n_rows_table, cond, cond_row, Cc, null_id, out);
    CVX_CHECK_LAUNCH("cvx_embed_gather_f32");
    return CVX_OK;
}

extern "C" int cvx_time_fourier_f32(const float* times, const float* w, float* out, int32_t n, int32_t half, cvx_stream_t s)
{
    CVX_REQUIRE(times && w && out && n >= 0 && half > 0, "time_fourier: bad arguments");
    if (n == 0) return CVX_OK;
    hipLaunchKernelGGL(time_fourier_kernel, dim3(n), dim3(256), 0, cvx_hip_stream(s), times, w, out, half);
    CVX_CHECK_LAUNCH("cvx_time_fourier_f32");
    return CVX_OK;
}

extern "C" int cvx_wav_to_int16(const float* wav, int16_t* pcm, int64_t n, cvx_stream_t s)
{
    CVX_REQUIRE(wav && pcm && n >= 0, "wav_to_int16: bad arguments");
    if (n == 0) return CVX_OK;
    hipLaunchKernelGGL(wav_to_int16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cvx_hip_stream(s), wav, pcm, n);
    CVX_CHECK_LAUNCH("cvx_wav_to_int16");
    return CVX_OK;
}
