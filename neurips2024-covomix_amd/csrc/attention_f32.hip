// fp32 flash-style attention for the CoVoMix transformer (gfx950, v_mfma_f32_32x32x2_f32).
//
// Replaces Attend.forward's non-flash branch (reference attend.py:108-126: einsum QK^T * scale
// -> softmax -> einsum AV) which materialises a [B,16,T,T] fp32 score tensor (512 MB per layer
// at B=8, T=1000).  Here nothing T x T ever reaches HBM.
//
// Layout: qkv[Bt, T, 3*H*64] as written by the to_qkv GEMM (q | k | v, RoPE already applied
// to q and k by the GEMM epilogue); out[Bt, T, H*64].
//
// Work split: block = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
// K/V tiles of 32 keys are staged global -> VGPR -> LDS (double buffered, one barrier per tile).
// Per tile and wave:
//   S^T[key, q] = K_tile . Q^T      (A = K rows from LDS via ds_read_b128, B = Q kept in VGPRs)
//        -> the "swapped" product puts a whole query column in ONE lane pair (lane, lane^32),
//           so the online-softmax row max / sum is 15 in-lane ops + one cross-half shuffle;
//   O^T[d, q]  += V_tile^T . P^T    (A = V columns from LDS via ds_read_b32, B = P straight from
//           the S^T accumulator registers - no P round trip through LDS, and the per-query
//           rescale factor is lane-local for every O^T register).
// 64 MFMAs (32 for S^T, 32 for O^T) of 64 cycles each per 32-key tile per wave.
#include "cvx_common.h"

namespace {

constexpr int HD = 64;          // head dim
constexpr int QB = 128;         // queries per block
constexpr int KT = 32;          // keys per tile
constexpr int K_LD = HD + 4;    // padded K row in LDS (floats): conflict-free ds_read_b128
constexpr int V_LD = HD;

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
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
    *reinterpret_cast<f16x4_t*>(hi + off) = h;
    if (lo) *reinterpret_cast<f16x4_t*>(lo + off) = l;      // lo == NULL: hi halves only
}

__global__ __launch_bounds__(256, 2) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                              _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo,
                                                              int T, int H, int n_groups, int n_qt, float scale_log2e,
                                                              const int* __restrict__ cu_seqlens, uint32_t* __restrict__ sat)
{
    __shared__ __attribute__((aligned(16))) float Ks[2][KT * K_LD];
    __shared__ __attribute__((aligned(16))) float Vs[2][KT * V_LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // block -> (batch*head group, query tile): all query tiles of one (batch, head) get the same
    // blockIdx % 8, i.e. (observed round-robin dispatch) the same XCD, so its K/V stay in ONE L2.
    const int grp = (blockIdx.x / (8 * n_qt)) * 8 + (blockIdx.x & 7);
    if (grp >= n_groups) return;
    const int head = grp % H, b = grp / H;
    const int q_blk = ((blockIdx.x >> 3) % n_qt) * QB;
    const int64_t row_stride = (int64_t)3 * H * HD;
    // ragged batch (cu_seqlens != NULL): sequence b owns rows [cu[b], cu[b+1]) of the packed [M, 3*H*64] tensor and only
    // attends to its own keys (acoustic.py:313 runs every utterance alone: no padding mask, no cross-utterance keys)
    int64_t row0 = (int64_t)b * T;
    if (cu_seqlens) {
        row0 = cu_seqlens[b];
        T = cu_seqlens[b + 1] - (int)row0;
        if (q_blk >= T) return;                 // block-uniform: the grid is sized for the longest sequence
    }
    const float* qbase = qkv + row0 * row_stride + head * HD;
    const float* kbase = qbase + H * HD;
    const float* vbase = qbase + 2 * H * HD;

    // ---- Q fragment (B operand of S^T): lane (q = l31, half) holds d = 8c + 4*half + e
    int qrow = q_blk + wid * 32 + l31;
    const bool q_valid = qrow < T;
    if (!q_valid) qrow = T - 1;
    f32x4 qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        qf[c] = *reinterpret_cast<const f32x4*>(qbase + (int64_t)qrow * row_stride + 8 * c + 4 * half);

    // ---- staging: a K/V tile is 32 rows x 64 floats = 512 float4 each -> 2 + 2 per thread
    const int s_row = tid >> 4;           // 0..15 (+16 for the second pass)
    const int s_col = (tid & 15) * 4;     // 0..60
    f32x4 rk[2], rv[2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int key0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = key0 + s_row + 16 * i;
            if (key < T) {
                rk[i] = *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * row_stride + s_col);
                rv[i] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)key * row_stride + s_col);
            } else { rk[i] = zero4; rv[i] = zero4; }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&Ks[buf][(s_row + 16 * i) * K_LD + s_col]) = rk[i];
            *reinterpret_cast<f32x4*>(&Vs[buf][(s_row + 16 * i) * V_LD + s_col]) = rv[i];
        }
    };

    f32x16 o0, o1;                 // O^T tiles: d in [0,32) and [32,64); column = this lane's query
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -1e30f;          // running max (in the log2 domain, i.e. of s*scale*log2e)
    float l_run = 0.f;             // this lane's partial of the running denominator

    const int ntiles = (T + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int it = 0; it < ntiles; ++it) {
        const int cur = it & 1;
        const int key0 = it * KT;
        if (it + 1 < ntiles) load_tile(key0 + KT);

        // ---- S^T = K . Q^T
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const float* kp = &Ks[cur][l31 * K_LD + 4 * half];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[c][e], sacc, 0, 0, 0);
        }

        // ---- online softmax over this lane's 16 keys (+ partner lane's 16)
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + mfma32_row(r, lane);
            const float sv = (key < T) ? sacc[r] * scale_log2e : -1e30f;
            sacc[r] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = exp2f(sacc[r] - m_new);
            sacc[r] = pv;
            psum += pv;
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T . P^T   (k pair of MFMA r: keys mfma32_row(r, lane) for the two halves)
        const float* vp = &Vs[cur][l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float v0 = vp[krow * V_LD];
            const float v1 = vp[krow * V_LD + 32];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sacc[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sacc[r], o1, 0, 0, 0);
        }

        if (it + 1 < ntiles) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane holds O[q][d] for d = dt*32 + (r&3) + 8*(r>>2) + 4*half
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    CvxSat amax;
    if (q_valid) {
        const int64_t o_off = (row0 + qrow) * (H * HD) + head * HD + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = o0[4 * g + e] * inv; c[e] = o1[4 * g + e] * inv; }
            if (out) {
                *reinterpret_cast<f32x4*>(out + o_off + 8 * g) = a;
                *reinterpret_cast<f32x4*>(out + o_off + 32 + 8 * g) = c;
            }
            if (out_hi) {     // split copy for the to_out GEMM's pre-split A operand
                store_split4(out_hi, out_lo, o_off + 8 * g, a, amax);
                store_split4(out_hi, out_lo, o_off + 32 + 8 * g, c, amax);
            }
        }
    }
    cvx_sat_commit(sat, amax);
}

}  // namespace

extern "C" int cvx_attention_varlen_f32(const float* qkv, float* out, uint16_t* out_hi, uint16_t* out_lo,
                                        const int32_t* cu_seqlens_dev, int32_t Bt, int32_t max_T, int32_t H, float scale, cvx_stream_t s)
{
    CVX_REQUIRE(qkv && (out || out_hi) && (out_hi || !out_lo), "attention: null pointer");      // out_lo == NULL: hi halves only
    CVX_REQUIRE(Bt >= 0 && max_T > 0 && H > 0, "attention: bad shape Bt=%d T=%d H=%d", Bt, max_T, H);
    CVX_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "attention: pointers must be 16-byte aligned");
    if (Bt == 0) return CVX_OK;
    if (out_hi) CVX_REQUIRE_SAT(s);
    const int n_qt = (max_T + QB - 1) / QB, n_groups = Bt * H;
    dim3 grid((unsigned)(((n_groups + 7) / 8) * 8 * n_qt));
    hipLaunchKernelGGL(attention_f32_kernel, grid, dim3(256), 0, cvx_hip_stream(s),
                       qkv, out, reinterpret_cast<_Float16*>(out_hi), reinterpret_cast<_Float16*>(out_lo),
                       max_T, H, n_groups, n_qt, scale * 1.44269504088896340736f, cu_seqlens_dev,
                       out_hi ? cvx_sat_flag_for(s) : nullptr);
    CVX_CHECK_LAUNCH("cvx_attention_f32");
    return CVX_OK;
}

extern "C" int cvx_attention_f32(const float* qkv, float* out, uint16_t* out_hi, uint16_t* out_lo,
                                 int32_t Bt, int32_t T, int32_t H, float scale, cvx_stream_t s)
{
    return cvx_attention_varlen_f32(qkv, out, out_hi, out_lo, nullptr, Bt, T, H, scale, s);
}
