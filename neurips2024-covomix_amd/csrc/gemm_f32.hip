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
//   * two LDS buffers, next tile's global loads issued before the MFMAs of the current
//     one, one barrier per K-step; 72 KB LDS -> 2 blocks per CU so the other block's
//     waves cover the barrier;
//   * blockIdx.x walks N: with the observed block->XCD round robin every XCD keeps its
//     own W panels L2-resident while A row-panels stream through.
// Epilogue fuses bias, GELU/SiLU, half-split RoPE (a wave owns one whole 64-wide head, so
// the (j, j+32) partner is the same accumulator register of the neighbouring MFMA tile)
// and the residual add.
#include "cvx_common.h"

namespace {

constexpr int BN = 128;
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;   // padded row (floats)

template <int TM>   // TM = MFMA tiles per wave along M (block M = TM*64)
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const cvx_gemm_args p)
{
    constexpr int BM = TM * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int srow = tid >> 3;            // 0..31
    const int skc = (tid & 7) * 4;        // 0..28

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
        for (int i = 0; i < TM * 2; ++i)
            *reinterpret_cast<f32x4*>(a + (srow + 32 * i) * LDS_LD + skc) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(b + (srow + 32 * i) * LDS_LD + skc) = rb[i];
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
        for (int q = 0; q < 4; ++q) {
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
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    const int colw = n0 + wn * 64;                 // first column of this wave (multiple of 64)
    const int c_lo = colw + (lane & 31);
    const int c_hi = c_lo + 32;
    const bool do_rope = (p.rope_cos != nullptr) && (colw < p.rope_cols);   // wave-uniform
    const float b_lo = (p.bias && c_lo < p.N) ? p.bias[c_lo] : 0.f;
    const float b_hi = (p.bias && c_hi < p.N) ? p.bias[c_hi] : 0.f;

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
                const float nlo = lo * c - hi * s;
                const float nhi = hi * c + lo * s;
                lo = nlo; hi = nhi;
            }
            if (p.residual) {
                if (c_lo < p.N) lo += p.residual[(int64_t)row * p.ldr + c_lo];
                if (c_hi < p.N) hi += p.residual[(int64_t)row * p.ldr + c_hi];
            }
            if (c_lo < p.N) p.C[(int64_t)row * p.ldc + c_lo] = lo;
            if (c_hi < p.N) p.C[(int64_t)row * p.ldc + c_hi] = hi;
        }
    }
}

template <int TM>
int launch_gemm(const cvx_gemm_args& a, hipStream_t st)
{
    constexpr int BM = TM * 64;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<TM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { cvx_set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return CVX_EHIP; }
        attr_set = true;
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
    hipLaunchKernelGGL(gemm_f32_kernel<TM>, grid, dim3(256), lds, st, a);
    CVX_CHECK_LAUNCH("cvx_gemm_bias_act_f32");
    return CVX_OK;
}

}  // namespace

extern "C" int cvx_gemm_bias_act_f32(const cvx_gemm_args* a, cvx_stream_t s)
{
    CVX_REQUIRE(a != nullptr, "gemm: null args");
    CVX_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    if (a->M == 0) return CVX_OK;
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
    hipStream_t st = reinterpret_cast<hipStream_t>(s);
    // Small-M problems (time tables, short utterances) use the 64-row tile to fill more CUs.
    const long blocks128 = (long)((a->M + 127) / 128) * ((a->N + BN - 1) / BN);
    if (a->M <= 64 || blocks128 < 256) return launch_gemm<1>(*a, st);
    return launch_gemm<2>(*a, st);
}
