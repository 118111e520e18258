// Split-precision flash attention (gfx950): the same algorithm and register choreography as attention_f32.hip
// (reference attend.py:108-126 without the T x T score tensor), but both contractions run on
// v_mfma_f32_32x32x16_f16 with every operand an (fp16 hi, fp16 lo) pair and three products per tile
//     q.k ~= k_hi*q_hi + k_hi*q_lo + k_lo*q_hi        p.v ~= v_hi*p_hi + v_hi*p_lo + v_lo*p_hi     (fp32 accumulate)
// 24 MFMAs of 32 cycles per 32-key tile and wave instead of 64 MFMAs of 64 cycles.
//
// Inputs come pre-split from the to_qkv GEMM epilogue (cvx_gemm_f16x3, QKV mode):
//   qk_hi/qk_lo [Bt*T, 2*H*64]  q | k after RoPE, row-major;
//   vt_hi/vt_lo [Bt*H*64, Tp]   v transposed per (sequence, head): row = head dim, column = frame slot (inside
//                               every 16 frames the four-frame groups are stored in the order 0, 2, 1, 3)
// so that every tile (K: 32 keys x 64 dims, V^T: 64 dims x 32 keys; hi and lo) is a set of contiguous rows that
// go global -> LDS by DMA - no staging registers, no LDS writes by the waves.  LDS tiles are XOR-swizzled on
// the DMA source address (K: chunk ^ ((row >> 1) & 7) over 128-byte rows; V^T: chunk ^ ((row >> 2) & 3) over
// 64-byte rows) so that all fragment reads are conflict-free ds_read_b128.
//
// S^T = K.Q^T keeps a query's scores in one lane pair; P is split in registers and used directly as the B
// operand of O^T += V^T.P^T: the k-slot order of that MFMA is chosen to be exactly the key order the S^T
// accumulator registers already have (registers 8s..8s+7 of lane half g hold keys 16s+4g+{0..3} and
// 16s+8+4g+{0..3}); the producer stores V^T with exactly those eight keys adjacent (frame-slot order above), so a
// V^T fragment is one 16-byte read and P never moves between lanes.
#include "cvx_common.h"
#include <stdlib.h>

namespace {

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int HD = 64, KT = 32;             // queries per block: 32 per wave, NW waves (128 or 256)
constexpr int TILE = KT * HD;                 // halves per operand tile (4 KiB)
constexpr int STAGE = 4 * TILE;               // Khi | Klo | Vthi | Vtlo

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ f16x8 gload8h(const f16* p)
{
    typedef const f16x8 __attribute__((address_space(1)))* gp;
    return *reinterpret_cast<gp>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ void store_split4(f16* hi, f16* lo, int64_t off, const f32x4 o, CvxSat& amax)
{
    cvx_amax4(amax, o);
    // lo == hi + 32: INTERLEAVED pair, [hi 32 | lo 32] per block of 32 values (one 128-byte line per K-step and row for
    // the consumer GEMM's DMA); the mapping is a function of the flat offset because every row is a multiple of 32 wide
    if (lo == hi + 32) off = ((off >> 5) << 6) | (off & 31);
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = fminf(fmaxf(o[e], -65504.f), 65504.f);
        h[e] = (f16)x;
        l[e] = (f16)(x - (float)h[e]);
    }
    *reinterpret_cast<f16x4*>(hi + off) = h;
    if (lo) *reinterpret_cast<f16x4*>(lo + off) = l;      // lo == NULL: hi halves only
}

// NT = 3: split operands, three products.  NT = 1: hi halves only (plain fp16 operands, fp32 accumulate and softmax).
#ifndef CVX_ATT_WAVES
#define CVX_ATT_WAVES 2
#endif
// ENERGY ABLATIONS (dev builds only, results are WRONG by construction; tools/archive/attn_ablate.py): which part of the loop the
// power-capped launch pays for.  bit 0: no L2 -> LDS DMA after the first tile (tiles stay resident); bit 1: no LDS fragment
// reads after the first tile (K / V^T fragments stay in registers); bit 2: no v_exp_f32 (p = its argument); bit 3: no P.V
// MFMAs; bit 4: no K.Q MFMAs.
#ifndef CVX_ATT_ABLATE
#define CVX_ATT_ABLATE 0
#endif
#ifndef CVX_ATT_SHORT_NW2
#define CVX_ATT_SHORT_NW2 1
#endif
// NW = waves per block (4 or 8), 32 queries each.  The K / V^T tiles a block streams through LDS are shared by its waves:
// with 8 waves (256 queries) the L2 -> LDS DMA bytes per score halve.  Round-3 ablations (tools/archive/attn_ablate.py, Bt = 16,
// T = 1000, H = 16, NW = 4): DMA switched off after the first tile 177 instead of 211 us and 0.233
// instead of 0.286 J (on zero operands, i.e. at full clock, 130 instead of 159 us); no LDS fragment reads -7 %; no
// v_exp_f32 -1 %; no P.V MFMAs 140 us; no K.Q MFMAs 134 us.  Halving the DMA BYTES (NW = 8) does not buy the DMA-off time.
// KS = key-split groups per block (1, 2 or 3; NW = 4): a SHORT launch - one utterance: 2 x 16 (sequence, head) pairs x 4 query
// tiles = 128 blocks of one wave per SIMD, each walking all key tiles in a dependent chain of ~1.45 us per tile (23 us at
// T = 500, while the same wave-tiles take 0.44 us each at three waves per SIMD) - gets KS x 4 waves per block: group s walks
// key tiles s, s + KS, ... through its own two-stage ring with its own running (m, l, O), and group 0 merges the groups' states
// through LDS in the fixed order 0, 1, 2 (the flash-decoding combine: O = sum O_s 2^(m_s - m), l likewise) before it normalises
// and stores.  The chain shortens KS-fold and every SIMD holds KS waves to overlap.
template <int NT, int NW, int KS = 1>
__global__ __launch_bounds__(64 * NW * KS, (KS > 1 ? (NW * KS + 3) / 4 : CVX_ATT_WAVES)) void attention_f16x3_kernel(const f16* __restrict__ qk_hi, const f16* __restrict__ qk_lo,
                                                                const f16* __restrict__ vt_hi, const f16* __restrict__ vt_lo,
                                                                float* __restrict__ out, f16* __restrict__ out_hi, f16* __restrict__ out_lo,
                                                                int T, int Tp, int H, int n_groups, int n_qt, float scale_log2e,
                                                                const float* __restrict__ qk_scale, const float* __restrict__ v_scale,
                                                                const float* __restrict__ out_scale, const int* __restrict__ cu_seqlens,
                                                                uint32_t* __restrict__ sat)
{
    // activation pre-scales (device scalars, powers of two): scores carry qk_scale^2, O carries v_scale
    if (qk_scale) { const float q = *qk_scale; scale_log2e /= q * q; }
    __shared__ __attribute__((aligned(16))) f16 smem_all[2 * STAGE * KS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wall = tid >> 6, ks = KS > 1 ? wall / NW : 0, wid = KS > 1 ? wall % NW : wall;     // key group, query wave
    f16* const smem = smem_all + ks * 2 * STAGE;
    const int g = lane >> 5, l31 = lane & 31;
    const int grp = (blockIdx.x / (8 * n_qt)) * 8 + (blockIdx.x & 7);      // same (batch, head) -> same XCD
    if (grp >= n_groups) return;
    const int head = grp % H, b = grp / H;
    constexpr int QB = 32 * NW;
    const int q_blk = ((blockIdx.x >> 3) % n_qt) * QB;
    const int64_t ldqk = (int64_t)2 * H * HD;
    // Key window [kc0, kc1) in V^T COLUMN coordinates; the q|k row of column c is c + roff.
    //   equal-length batch: every (sequence, head) has its own V^T rows, columns 0 .. T-1 (+ zero padding up to Tp);
    //   ragged batch (cu_seqlens): ONE V^T row set per head over all M packed rows (the to_qkv epilogue ran with
    //     rope_T = M), sequence b = columns [cu[b], cu[b+1]).  Key tiles stay aligned to 32 GLOBAL columns (16-byte DMA
    //     pieces, frame-slot groups of 16), so the first and the last tile of a sequence may contain a neighbour's keys:
    //     they are masked like the keys >= T of the last tile (acoustic.py:313: an utterance only ever sees itself).
    int kc0 = 0, kc1 = T, vt_grp = b * H + head;
    int64_t roff = (int64_t)b * T;
    if (cu_seqlens) {
        kc0 = cu_seqlens[b]; kc1 = cu_seqlens[b + 1]; roff = 0; vt_grp = head;
        if (q_blk >= kc1 - kc0) return;            // block-uniform: the grid is sized for the longest sequence
    }
    const int Tb = kc1 - kc0;

    // ---- Q fragments (B operand of S^T): lane (q = l31, g) holds d = 16s + 8g .. +7 for s = 0..3, hi and lo
    int qrow = q_blk + wid * 32 + l31;
    const bool q_valid = qrow < Tb;
    if (!q_valid) qrow = Tb - 1;
    const int64_t q_grow = roff + kc0 + qrow;      // row of this lane's query in the packed tensors
    f16x8 qh[4], ql[4];
    {
        const int64_t off = q_grow * ldqk + head * HD + 8 * g;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = gload8h(qk_hi + off + 16 * s);
            if constexpr (NT == 3) ql[s] = gload8h(qk_lo + off + 16 * s);
        }
    }

    // ---- DMA sources.  NW = 4: wave w fetches K rows [8w, 8w+8) (hi, lo) and V^T rows [16w, 16w+16) (hi, lo): 4 pieces per
    // tile and wave.  NW = 8: waves 0-3 fetch the K pieces, waves 4-7 the V^T pieces (2 per tile and wave).
    // NW = 2 (64-query blocks of the key-split form): every wave fetches two of the four row groups of K and of V^T.
    constexpr int PW = NW == 2 ? 2 : 1;                                     // row groups per wave
    const int wq = NW == 2 ? 2 * wid : (wid & 3);
    const bool dma_k = NW <= 4 || wid < 4, dma_v = NW <= 4 || wid >= 4;
    int k_r[PW], k_c[PW], v_r[PW], v_c[PW];
    int64_t k_col[PW], v_row[PW];
#pragma unroll
    for (int pp = 0; pp < PW; ++pp) {
        k_r[pp] = 8 * (wq + pp) + (lane >> 3);                              // key row inside the tile
        k_c[pp] = (lane & 7) ^ ((k_r[pp] >> 1) & 7);                        // source chunk for LDS chunk (lane & 7)
        k_col[pp] = (int64_t)H * HD + head * HD + 8 * k_c[pp];
        v_r[pp] = 16 * (wq + pp) + (lane >> 2);                             // head-dim row inside the tile
        v_c[pp] = (lane & 3) ^ ((v_r[pp] >> 2) & 3);
        v_row[pp] = ((int64_t)vt_grp * HD + v_r[pp]) * Tp + 8 * v_c[pp];
    }
    auto issue = [&](int key0, int stage) {
        f16* S = smem + stage * STAGE;
#pragma unroll
        for (int pp = 0; pp < PW; ++pp) {
            if (dma_k) {                                                    // (wave-uniform)
                const int key = min(max(key0 + k_r[pp], kc0), kc1 - 1);
                const int64_t ko = (roff + key) * ldqk + k_col[pp];
                glds16(qk_hi + ko, S + 8 * (wq + pp) * HD);
                if constexpr (NT == 3) glds16(qk_lo + ko, S + TILE + 8 * (wq + pp) * HD);
            }
            if (dma_v) {
                const int64_t vo = v_row[pp] + key0;
                glds16(vt_hi + vo, S + 2 * TILE + 16 * (wq + pp) * KT);
                if constexpr (NT == 3) glds16(vt_lo + vo, S + 3 * TILE + 16 * (wq + pp) * KT);
            }
        }
    };

#ifdef CVX_ATT_REGSTAGE
    // A/B variant (T14 of the guide, "issue early / write late"): the next tile goes global -> VGPR while the current one is
    // computed and VGPR -> LDS (same swizzled layout) behind its last MFMA, instead of by LDS-DMA.  NW = 4 only.
    f16x8 st_kh, st_kl, st_vh, st_vl;
    auto fetch = [&](int key0) {
        const int key = min(max(key0 + k_r[0], kc0), kc1 - 1);
        const int64_t ko = (roff + key) * ldqk + k_col[0];
        st_kh = gload8h(qk_hi + ko);
        if constexpr (NT == 3) st_kl = gload8h(qk_lo + ko);
        const int64_t vo = v_row[0] + key0;
        st_vh = gload8h(vt_hi + vo);
        if constexpr (NT == 3) st_vl = gload8h(vt_lo + vo);
    };
    auto commit = [&](int stage) {
        f16* S = smem + stage * STAGE;
        *reinterpret_cast<f16x8*>(S + 8 * wq * HD + lane * 8) = st_kh;
        if constexpr (NT == 3) *reinterpret_cast<f16x8*>(S + TILE + 8 * wq * HD + lane * 8) = st_kl;
        *reinterpret_cast<f16x8*>(S + 2 * TILE + 16 * wq * KT + lane * 8) = st_vh;
        if constexpr (NT == 3) *reinterpret_cast<f16x8*>(S + 3 * TILE + 16 * wq * KT + lane * 8) = st_vl;
    };
#endif
    // ---- fragment offsets (halves)
    int koff[4];                                   // K rows are 64 halves; chunk (2s+g) ^ ((row>>1)&7)
#pragma unroll
    for (int s = 0; s < 4; ++s) koff[s] = l31 * HD + 8 * ((2 * s + g) ^ ((l31 >> 1) & 7));
    int voff[2];                                   // V^T rows are 32 halves; chunk (2s + g) ^ ((row>>2)&3)
#pragma unroll
    for (int s = 0; s < 2; ++s) voff[s] = l31 * KT + 8 * ((2 * s + g) ^ ((l31 >> 2) & 3));

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    asm volatile("" : "+v"(zero16));          // opaque to the optimiser: stays one register block instead of 16 movs per tile

    // (a stage of two key tiles - one barrier per 64 keys - was measured slower: 64 KiB of LDS drops the kernel
    //  from 3 to 2 blocks per CU)
    // (a 3-stage ring - tiles requested two ahead - measured the same: the loop is not DMA-latency bound)
    const int tile0 = kc0 / KT, ntiles = (kc1 + KT - 1) / KT - tile0;
#ifdef CVX_ATT_REGSTAGE
    fetch(tile0 * KT);
    commit(0);
#else
    if (KS == 1 || ks == 0) issue(tile0 * KT, 0);
#endif
#ifdef CVX_ATT_TRACE
    unsigned long long tr[5] = {0, 0, 0, 0, 0};
#define TSTAMP(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_readcyclecounter(); tr[i] += now_ - tlast_; tlast_ = now_; }
    unsigned long long tlast_ = __builtin_readcyclecounter();
#else
#define TSTAMP(i)
#endif
    f16x8 kfh[4], kfl[4];                         // K fragments of the running tile
    f16x8 vfh[2][2], vfl[2][2];                   // V^T fragments [s][dt]
    if (KS > 1 && ks > 0 && ks < ntiles) issue((tile0 + ks) * KT, 0);      // (group 0's first tile was requested above)
    const int n_it = KS > 1 ? (ntiles + KS - 1) / KS : ntiles;
    for (int jt = 0; jt < n_it; ++jt) {
        const int it = KS > 1 ? jt * KS + ks : jt;
        const int cur = jt & 1, key0 = (tile0 + it) * KT;
        if (KS > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // this group's tile landed (all four waves); its other stage is free
            if (it + KS < ntiles) issue(key0 + KS * KT, cur ^ 1);
            if (it >= ntiles) continue;                     // (group-uniform: a group past its last tile only keeps the barriers company)
        } else {
#ifdef CVX_ATT_REGSTAGE
        __syncthreads();                                    // every wave's ds_writes of tile `it` are visible; stage cur^1 is free
        if (it + 1 < ntiles) fetch(key0 + KT);
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // tile `it` landed everywhere; stage cur^1 is free
        TSTAMP(0)
        if (it + 1 < ntiles && !((CVX_ATT_ABLATE & 1) && it > 0)) issue(key0 + KT, cur ^ 1);
#endif
        }
        const f16* S = smem + cur * STAGE;

        // ---- S^T = K . Q^T  (3 products per 16-wide d slice), ONE accumulator: the matrix pipe forwards the result of an
        // MFMA to a dependent MFMA on the same accumulator, so the chain costs nothing and the adds that would merge
        // per-term accumulators disappear (measured: 1 accumulator 260 us, 3 accumulators 272 us, 2: 285 us).
        // The first MFMA of the chain takes its C operand from a zero register block kept live over the loop.
        f32x16 sacc;
        if (!((CVX_ATT_ABLATE & 2) && it > 0)) {  // all K fragments first (8 reads in flight), then the MFMA chain
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kfh[s] = *reinterpret_cast<const f16x8*>(S + koff[s]);
                if constexpr (NT == 3) kfl[s] = *reinterpret_cast<const f16x8*>(S + TILE + koff[s]);
            }
        }
        if (CVX_ATT_ABLATE & 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = (float)(r + it) * 0.01f;
        } else
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (NT == 3) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[s], qh[s], s == 0 ? zero16 : sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], ql[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], qh[s], sacc, 0, 0, 0);
            } else {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], qh[s], s == 0 ? zero16 : sacc, 0, 0, 0);
            }
        }

#ifdef CVX_ATT_TRACE
        asm volatile("" : "+v"(sacc));
        { float t_ = sacc[0]; asm volatile("s_nop 0" : "+v"(t_)); }
#endif
        TSTAMP(1)
        // ---- online softmax (this lane: 16 keys of query l31; partner lane^32 holds the other 16).
        // The running max m_run is kept in the scaled log2 domain; scores stay raw and the scale is folded into one
        // fma per element: p = exp2(s*c - m).  Only the last tile (ragged batches: and the first) can contain keys outside the sequence (wave-uniform branch), and
        // the 32 accumulator rescales are skipped when no lane's max moved (alpha == 1 exactly - also wave-uniform).
        if (key0 + KT > kc1 || key0 < kc0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = key0 + mfma32_row(r, lane);
                if (kk >= kc1 || kk < kc0) sacc[r] = -1e30f;
            }
        }
        float mx = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;          // scale > 0: max commutes with it
        const float m_new = fmaxf(m_run, mx);
        const bool moved = m_new != m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        f16x8 ph[2], pl[2];
        {
            // two probabilities at a time: one packed RNE conversion for the hi halves, the residuals straight from the
            // packed register with v_fma_mix_f32 (fp16 source, fp32 result: pv - hi, exact), one packed conversion for lo
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 hw[2], lw[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a0 = fmaf(sacc[2 * j], scale_log2e, -m_new), a1 = fmaf(sacc[2 * j + 1], scale_log2e, -m_new);
                const float p0 = (CVX_ATT_ABLATE & 4) ? a0 : __builtin_amdgcn_exp2f(a0);
                const float p1 = (CVX_ATT_ABLATE & 4) ? a1 : __builtin_amdgcn_exp2f(a1);
                psum += p0 + p1;                 // (pairs first: 8 dependent adds instead of 16)
                const f16x2 h2 = __builtin_convertvector(f32x2{p0, p1}, f16x2);
                const unsigned int hb = __builtin_bit_cast(unsigned int, h2);
                hw[j >> 2][j & 3] = hb;
                if constexpr (NT == 3) {
                    float r0, r1;
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(p0));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(p1));
                    const f16x2 l2 = __builtin_convertvector(f32x2{r0, r1}, f16x2);
                    lw[j >> 2][j & 3] = __builtin_bit_cast(unsigned int, l2);
                }
            }
            ph[0] = __builtin_bit_cast(f16x8, hw[0]); ph[1] = __builtin_bit_cast(f16x8, hw[1]);
            if constexpr (NT == 3) { pl[0] = __builtin_bit_cast(f16x8, lw[0]); pl[1] = __builtin_bit_cast(f16x8, lw[1]); }
        }
        l_run = l_run * alpha + psum;
        if (__any(moved)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }

        TSTAMP(2)
        // ---- O^T += V^T . P^T
        const f16* Vh = S + 2 * TILE;
        const f16* Vl = S + 3 * TILE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 (&vhh)[2] = vfh[s];
            f16x8 (&vll)[2] = vfl[s];
            if (!((CVX_ATT_ABLATE & 2) && it > 0)) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    vhh[dt] = *reinterpret_cast<const f16x8*>(Vh + dt * 32 * KT + voff[s]);
                    if constexpr (NT == 3) vll[dt] = *reinterpret_cast<const f16x8*>(Vl + dt * 32 * KT + voff[s]);
                }
            }
            if (CVX_ATT_ABLATE & 8) continue;
            // interleave the two O^T tiles: consecutive MFMAs alternate accumulators
            if constexpr (NT == 3) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vll[0], ph[s], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vll[1], ph[s], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhh[0], pl[s], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhh[1], pl[s], o1, 0, 0, 0);
            }
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhh[0], ph[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhh[1], ph[s], o1, 0, 0, 0);
        }
#ifdef CVX_ATT_REGSTAGE
        if (it + 1 < ntiles) commit(cur ^ 1);
#endif
#ifdef CVX_ATT_TRACE
        asm volatile("" : "+v"(o0), "+v"(o1));
        { float t_ = o0[0] + o1[0]; asm volatile("s_nop 0" : "+v"(t_)); }
#endif
        TSTAMP(3)
    }
#ifdef CVX_ATT_TRACE
    if (out && lane == 0) {
        unsigned long long* tb = reinterpret_cast<unsigned long long*>(out) + ((size_t)blockIdx.x * 4 + wid) * 8;
        tb[0] = tr[0]; tb[1] = tr[1]; tb[2] = tr[2]; tb[3] = tr[3]; tb[4] = ntiles;
    }
    out = nullptr;
#endif

    if constexpr (KS > 1) {
        // merge the key groups' states: groups 1 .. KS-1 park (m, l, O) in LDS (the rings are dead), group 0 folds them in order
        __syncthreads();
        f32x4* const X = reinterpret_cast<f32x4*>(smem_all);
        if (ks > 0) {
            f32x4* const dst = X + ((size_t)((ks - 1) * NW + wid) * 9) * 64 + lane;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                dst[q4 * 64] = f32x4{o0[4 * q4], o0[4 * q4 + 1], o0[4 * q4 + 2], o0[4 * q4 + 3]};
                dst[(4 + q4) * 64] = f32x4{o1[4 * q4], o1[4 * q4 + 1], o1[4 * q4 + 2], o1[4 * q4 + 3]};
            }
            dst[8 * 64] = f32x4{m_run, l_run, 0.f, 0.f};
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int s2 = 1; s2 < KS; ++s2) {
            const f32x4* const src = X + ((size_t)((s2 - 1) * NW + wid) * 9) * 64 + lane;
            const f32x4 ml = src[8 * 64];
            const float m_new = fmaxf(m_run, ml[0]);
            const float fa = __builtin_amdgcn_exp2f(m_run - m_new), fb = __builtin_amdgcn_exp2f(ml[0] - m_new);
            m_run = m_new;
            l_run = l_run * fa + ml[1] * fb;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 a = src[q4 * 64], c = src[(4 + q4) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[4 * q4 + e] = o0[4 * q4 + e] * fa + a[e] * fb;
                    o1[4 * q4 + e] = o1[4 * q4 + e] * fa + c[e] * fb;
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot / (v_scale ? *v_scale : 1.f);        // fp32 output: the true value
    const float osc = out_scale ? *out_scale : 1.f;                        // split output: times the consumer's pre-scale
    CvxSat amax;
    if (q_valid) {
        const int64_t o_off = q_grow * (H * HD) + head * HD + 4 * g;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = o0[4 * gq + e] * inv; c[e] = o1[4 * gq + e] * inv; }
            if (out) {
                *reinterpret_cast<f32x4*>(out + o_off + 8 * gq) = a;
                *reinterpret_cast<f32x4*>(out + o_off + 32 + 8 * gq) = c;
            }
            if (out_hi) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] *= osc; c[e] *= osc; }
                store_split4(out_hi, out_lo, o_off + 8 * gq, a, amax);
                store_split4(out_hi, out_lo, o_off + 32 + 8 * gq, c, amax);
            }
        }
    }
    // a non-finite normaliser (overflowed scores, all-masked row) also means the result cannot be trusted: flag it
    if (!(l_tot > 0.f && l_tot < __builtin_inff())) amax.bad = true;
    cvx_sat_commit(sat, amax);
}

}  // namespace

static int launch_attention_f16x3(const uint16_t* qk_hi, const uint16_t* qk_lo, const uint16_t* vt_hi, const uint16_t* vt_lo,
                                  float* out, uint16_t* out_hi, uint16_t* out_lo, const int32_t* cu_seqlens_dev,
                                  int32_t Bt, int32_t T, int64_t cols, int32_t Tp, int32_t H, float scale,
                                  const float* qk_scale_dev, const float* v_scale_dev, const float* out_scale_dev, cvx_stream_t s)
{
    // T: frames per sequence (equal-length batch) or the LONGEST sequence (ragged batch); cols: V^T columns in use
    const bool single = (qk_lo == nullptr);        // hi halves only: plain fp16 operands, one product
    CVX_REQUIRE(qk_hi && vt_hi && ((qk_lo == nullptr) == (vt_lo == nullptr)) && (out || out_hi) &&
                (out_hi || !out_lo) && (single || (out_hi == nullptr) == (out_lo == nullptr)),
                "attention_f16x3: null pointer");
    CVX_REQUIRE(Bt >= 0 && T > 0 && H > 0 && Tp % 8 == 0 && Tp >= ((cols + KT - 1) / KT) * KT,
                "attention_f16x3: bad shape Bt=%d T=%d Tp=%d H=%d (Tp must be a multiple of 8 and >= the V^T columns in use rounded up to 32)", Bt, T, Tp, H);
    CVX_REQUIRE((((uintptr_t)qk_hi | (uintptr_t)qk_lo | (uintptr_t)vt_hi | (uintptr_t)vt_lo) & 15) == 0,
                "attention_f16x3: inputs must be 16-byte aligned");
    if (Bt == 0) return CVX_OK;
    // 128-query blocks (4 waves, three blocks per CU).  256-query blocks (8 waves: half the L2 -> LDS tile traffic per score,
    // one block per CU) are built and correct but measured 3 % SLOWER (218.6 vs 212.5 us, same joules; on zero operands
    // 182 vs 159 us: the schedule loses what the traffic saves) - opt-in for A/B runs: -DCVX_ATT_QB256.
#ifdef CVX_ATT_QB256
    const int nw = T >= 384 ? 8 : 4;
#else
    const int nw = 4;
#endif
    int qb = 32 * nw;
    int n_qt = (T + qb - 1) / qb;
    const int n_groups = Bt * H;
    dim3 grid((unsigned)(((n_groups + 7) / 8) * 8 * n_qt));
    if (out_hi) CVX_REQUIRE_SAT(s);
    uint32_t* sat = cvx_sat_flag_for(s);
    // key-split groups for short launches (see the kernel): fewer than 2048 query rows = at most one 128-query block per CU (96 KiB of
    // LDS with three groups).
    int ksplit = 1, nwk = 4;
    const int64_t q_rows = cu_seqlens_dev ? cols : (int64_t)Bt * T;             // query rows of the launch
    if (nw == 4 && T >= 4 * KT && q_rows < 2048) {
        ksplit = 3;
#if CVX_ATT_SHORT_NW2
        // half of the chip's SIMDs hold no wave at all when the 128-query blocks number fewer than 128: 64-query blocks (two query waves,
        // four key groups: 8 waves per block) put a wave on every SIMD
        if (grid.x <= 128) {
            nwk = 2; ksplit = 4; qb = 64; n_qt = (T + qb - 1) / qb;
            grid = dim3((unsigned)(((n_groups + 7) / 8) * 8 * n_qt));
        }
#endif
    }
#define CVX_ATT_LAUNCH(NT_, NW_)                                                                                                          \
    hipLaunchKernelGGL((attention_f16x3_kernel<NT_, NW_>), grid, dim3(64 * NW_), 0, cvx_hip_stream(s),                       \
                       reinterpret_cast<const f16*>(qk_hi), reinterpret_cast<const f16*>(qk_lo),                                          \
                       reinterpret_cast<const f16*>(vt_hi), reinterpret_cast<const f16*>(vt_lo),                                          \
                       out, reinterpret_cast<f16*>(out_hi), reinterpret_cast<f16*>(out_lo),                                               \
                       T, Tp, H, n_groups, n_qt, scale * 1.44269504088896340736f, qk_scale_dev, v_scale_dev, out_scale_dev, cu_seqlens_dev, sat)
#define CVX_ATT_LAUNCH_KS(NT_, KS_) CVX_ATT_LAUNCH_KW(NT_, 4, KS_)
#define CVX_ATT_LAUNCH_KW(NT_, NW_, KS_)                                                                                                  \
    hipLaunchKernelGGL((attention_f16x3_kernel<NT_, NW_, KS_>), grid, dim3(64 * NW_ * KS_), 0, cvx_hip_stream(s),              \
                       reinterpret_cast<const f16*>(qk_hi), reinterpret_cast<const f16*>(qk_lo),                                          \
                       reinterpret_cast<const f16*>(vt_hi), reinterpret_cast<const f16*>(vt_lo),                                          \
                       out, reinterpret_cast<f16*>(out_hi), reinterpret_cast<f16*>(out_lo),                                               \
                       T, Tp, H, n_groups, n_qt, scale * 1.44269504088896340736f, qk_scale_dev, v_scale_dev, out_scale_dev, cu_seqlens_dev, sat)
    if (nwk == 2) { if (single) CVX_ATT_LAUNCH_KW(1, 2, 4); else CVX_ATT_LAUNCH_KW(3, 2, 4); }
    else if (ksplit == 3) { if (single) CVX_ATT_LAUNCH_KS(1, 3); else CVX_ATT_LAUNCH_KS(3, 3); }
    else if (single) { if (nw == 8) CVX_ATT_LAUNCH(1, 8); else CVX_ATT_LAUNCH(1, 4); }
    else { if (nw == 8) CVX_ATT_LAUNCH(3, 8); else CVX_ATT_LAUNCH(3, 4); }
#undef CVX_ATT_LAUNCH_KS
#undef CVX_ATT_LAUNCH_KW
#undef CVX_ATT_LAUNCH
    CVX_CHECK_LAUNCH("cvx_attention_f16x3");
    return CVX_OK;
}

extern "C" int cvx_attention_f16x3_scaled(const uint16_t* qk_hi, const uint16_t* qk_lo, const uint16_t* vt_hi, const uint16_t* vt_lo,
                                          float* out, uint16_t* out_hi, uint16_t* out_lo,
                                          int32_t Bt, int32_t T, int32_t Tp, int32_t H, float scale,
                                          const float* qk_scale_dev, const float* v_scale_dev, const float* out_scale_dev, cvx_stream_t s)
{
    return launch_attention_f16x3(qk_hi, qk_lo, vt_hi, vt_lo, out, out_hi, out_lo, nullptr, Bt, T, T, Tp, H, scale,
                                  qk_scale_dev, v_scale_dev, out_scale_dev, s);
}

extern "C" int cvx_attention_f16x3_varlen(const uint16_t* qk_hi, const uint16_t* qk_lo, const uint16_t* vt_hi, const uint16_t* vt_lo,
                                          float* out, uint16_t* out_hi, uint16_t* out_lo, const int32_t* cu_seqlens_dev,
                                          int32_t n_seq, int32_t max_T, int64_t M, int32_t vt_ld, int32_t H, float scale,
                                          const float* qk_scale_dev, const float* v_scale_dev, const float* out_scale_dev, cvx_stream_t s)
{
    CVX_REQUIRE(cu_seqlens_dev && M >= 0 && max_T <= M, "attention_f16x3_varlen: needs cu_seqlens and max_T <= M");
    return launch_attention_f16x3(qk_hi, qk_lo, vt_hi, vt_lo, out, out_hi, out_lo, cu_seqlens_dev, n_seq, max_T, M, vt_ld, H, scale,
                                  qk_scale_dev, v_scale_dev, out_scale_dev, s);
}

extern "C" int cvx_attention_f16x3(const uint16_t* qk_hi, const uint16_t* qk_lo, const uint16_t* vt_hi, const uint16_t* vt_lo,
                                   float* out, uint16_t* out_hi, uint16_t* out_lo,
                                   int32_t Bt, int32_t T, int32_t Tp, int32_t H, float scale, cvx_stream_t s)
{
    return cvx_attention_f16x3_scaled(qk_hi, qk_lo, vt_hi, vt_lo, out, out_hi, out_lo, Bt, T, Tp, H, scale, nullptr, nullptr, nullptr, s);
}
