// text2semantic autoregressive decode: one token step of the reference's
// TextToSemantic.generate sampling loop (covomix/covomix_model/text2semantic.py:748-820) with a KV cache.
//
// One new position per step: every projection is a matrix-VECTOR product, so the step is bound by streaming the
// decoder weights (CoSingle 60 MB, CoMix 186 MB of fp32 per token step) - an HBM/MALL-bound path, not an MFMA one.
// BATCH: up to 64 decode SLOTS advance together (each at its OWN position, with its own context / cache / eos): a weight row
// is read once per group of 8 slots and multiplied with every slot's vector.  The chain of 34 dependent launches is latency-bound,
// so a step at batch 32 costs little more than at batch 8.  The arithmetic per utterance (summation order included) does not
// depend on the batch size, the slot or the position of the other slots: results are bit-identical to batch 1.
// CONTINUOUS BATCHING (round 6): with a dialogue queue (cvx_t2s_decoder.queue) a slot whose dialogue has sampled its eos (or
// reached its step limit) takes the next pending dialogue INSIDE sample_kernel - no host round trip, no idle slot-steps; the
// per-dialogue buffers (context k/v, uniforms, tokens) are indexed by the dialogue number the slot record carries.
// Kernels (all fp32, fp32 accumulate):
//   gemv_kernel<MODE>   block = 4 waves, every wave owns TWO output rows (the pairs are chosen so that the epilogue
//                       has both members of a RoPE pair / a GEGLU (value, gate) pair in one wave); the input vector is
//                       staged in LDS once per block, optionally RMS-normalised (F.normalize * sqrt(D) * gamma,
//                       text2semantic.py:143-151) on the way; rows stream with 16-byte loads + a wave reduction.
//   attn_kernel         one block per head over the cached keys (self: roped keys [0, pos]; cross: learned null k/v +
//                       the encoder context, text2semantic.py:253-262); 16 lanes per key for coalesced 256-byte rows.
//   sample_kernel       top-k (k = ceil(0.1 * V), :126-132) + Gumbel argmax (:105-113) from caller-supplied U(0,1)
//                       draws, eos bookkeeping (:803-818), and the embedding of the sampled ids = next step's input.
// The reference rotates ALL cached keys again every step with rotary_embedding_torch's interleaved pairs
// (rotary_embedding_torch.py:146-157); rotating a key once at its own position when it enters the cache is the same
// arithmetic.  Interleaved pairs (2i, 2i+1) become half-split pairs (i, i+32) by permuting the rows of to_q / to_k
// inside every head at load time (q.k is invariant under a common permutation) - done by the host packer.
// Positions: the device-side slot records (state[slot][0]); every kernel reads them, so a captured HIP graph of N steps replays as is.
#include "cvx_common.h"

// No implicit a * b + c -> fma contraction in this file: the batch-1 / 2 / 4 / 8 instances of a kernel are REQUIRED to agree
// bitwise per utterance (tests/test_t2s_gpu.py: batched == one-by-one tokens); every fused multiply-add below is an explicit fmaf.
#pragma clang fp contract(off)

namespace {

constexpr int T2S_MAX_KEYS = 4096;
constexpr int T2S_MAX_DIM = 4096;     // floats of the staged input vector (16 KiB of LDS)
constexpr int SR = 8;                 // int32 per slot record / dialogue record (cvx_t2s_decoder.state / .dialogues)
constexpr int T2S_MAX_BATCH = 64;
#ifndef CVX_T2S_STAGE_HALF
#define CVX_T2S_STAGE_HALF 0
#endif
constexpr bool STAGE_HALF = CVX_T2S_STAGE_HALF != 0;   // (dev A/B) eight slots staged as two passes of four

enum { MODE_QKV = 0, MODE_PLAIN = 1, MODE_RES = 2, MODE_GEGLU = 3, MODE_LOGITS = 4 };

struct GemvArgs {
    const float* W;          // [N, ldw]
    int64_t ldw;
    const float* x;          // input vectors [batch][x_stride], K used
    const float* gamma;      // RMSNorm weight over x (NULL: x is used as is)
    const float* bias;       // [N] or NULL
    float* y;                // output [batch][y_stride]
    int N, K;
    int x_stride, y_stride;
    int64_t cache_stride;    // floats between the k/v caches of two utterances
    // MODE_QKV: rows [0, inner) q, [inner, 2 inner) k, [2 inner, 3 inner) v; RoPE on q/k at position *pos
    int inner;
    const float* rope_cos;   // [max_len, 32]
    const float* rope_sin;
    float* k_cache;          // [max_len, inner]
    float* v_cache;
    const int* state;        // slot records: state[SR * slot + 0] = pos
    int max_len;             // positions >= max_len are clamped (the host never asks for them; keeps a stray call in bounds)
    // slot groups (batch > 8): the batch is ceil(batch / 8) groups of BQ = 8 slots; a block works on `gl` consecutive groups (the
    // weight rows of its pairs stay in registers) and `gy` blocks share a row block (dispatched back to back on one XCD: L2 hits)
    int gy, gl, n_blocks;
    // MODE_GEGLU: rows j (value) and j + F (gate), F = N / 2; y[j] for j < F, zero fill up to y_pad
    int y_pad;
    // MODE_LOGITS: `streams` independent slices of the normalised vector: y[s*N + n] = W[n,:] . xn[s*K .. (s+1)*K)
    int streams;
};

// x[l] + x[l ^ o] for o = 8 / 4 / 2 / 1 without an LDS permute: DPP row rotation (xor 8 inside a 16-lane row), the LDS crossbar's swizzle
// (xor 4) and DPP quad permutes fused into the add - the pairings of the xor butterfly exactly, so the same bits as __shfl_xor
__device__ __forceinline__ float xor_add8(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false)); }   // row_ror:8
__device__ __forceinline__ float xor_add4(float v) { return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)); }
__device__ __forceinline__ float xor_add2(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)); }
__device__ __forceinline__ float xor_add1(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)); }

__device__ __forceinline__ float wave_sum(float v)
{
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    return xor_add1(xor_add2(xor_add4(xor_add8(v))));
}

constexpr int PF = 4;                                      // 4 x 256 floats per row in flight (K <= 1024 entirely)
#ifndef CVX_T2S_NT
#define CVX_T2S_NT 1
#endif
// weight rows: streamed once per token step by one wave each -> non-temporal loads (A/B: -DCVX_T2S_NT=0)
__device__ __forceinline__ f32x4 wload4(const float* p) { return CVX_T2S_NT ? gload4_nt(p) : gload4(p); }
// ACTIVATION reads (x, q, att, h, logits, the state record, cache rows) are plain loads (round 4 measured L1-bypassing loads at
// -36 % on this per-launch path: 147.8 -> 201.2 us per CoSingle step)
__device__ __forceinline__ float aload(const float* p) { return *p; }
__device__ __forceinline__ int aloadi(const int* p) { return *p; }
__device__ __forceinline__ f32x4 aload4(const float* p) { return gload4(p); }

// the two rows of row-pair `pair` (the pairs are chosen so that the epilogue has both members of a RoPE pair / a GEGLU
// (value, gate) pair in one wave)
template <int MODE>
__device__ __forceinline__ bool pair_rows(const GemvArgs& a, int pair, int& r0, int& r1, int& sidx)
{
    sidx = 0;
    if (MODE == MODE_QKV) {
        // pair p -> head-local (h, i): rows base + h*64 + i and + 32 for i in [0, 32); 3*inner/2 pairs in total
        const int per = a.inner / 2;
        const int sec = pair / per, q = pair - sec * per;             // 0 q, 1 k, 2 v
        r0 = sec * a.inner + (q >> 5) * 64 + (q & 31);
        r1 = r0 + 32;
        return pair < 3 * per;
    } else if (MODE == MODE_GEGLU) {
        const int F = a.N / 2;
        r0 = pair; r1 = pair + F;
        return pair < F;
    } else if (MODE == MODE_LOGITS) {
        const int per = (a.N + 1) / 2;
        sidx = pair / per;
        r0 = 2 * (pair - sidx * per); r1 = r0 + 1;
        return sidx < a.streams;
    }
    r0 = 2 * pair; r1 = r0 + 1;
    return r0 < a.N;
}

// One CHUNK = 1024 consecutive floats of the input vector(s) = PF x 256-float strips; lane l of a wave works on the 4-vectors
// 4 l + 256 i of every row (fixed summation order: lane-strided 4-vectors in ascending k, then the lane tree below).
struct RowChunk { f32x4 pa[PF], pb[PF]; };
struct PairInfo { const float *w0, *w1; int r0, r1, sidx; bool valid, has1; };

template <int MODE>
__device__ __forceinline__ PairInfo pair_info(const GemvArgs& a, int pair)
{
    PairInfo p;
    p.valid = pair_rows<MODE>(a, pair, p.r0, p.r1, p.sidx);
    p.has1 = p.valid && p.r1 < a.N;
    p.w0 = a.W + (int64_t)(p.valid ? p.r0 : 0) * a.ldw;
    p.w1 = a.W + (int64_t)(p.has1 ? p.r1 : (p.valid ? p.r0 : 0)) * a.ldw;
    return p;
}

// chunk c of the two weight rows of a pair: independent of the input vectors, so the HBM / MALL round trip overlaps whatever
// precedes the dot product (the staging of x).  Strips are classified with WAVE-UNIFORM conditions (whole / partial / absent): the
// common whole strip carries no lane masks (a lane-wise guard on every strip cost 290 selects and 220 spilled mask registers).
__device__ __forceinline__ void load_chunk(const GemvArgs& a, const PairInfo& p, int c, int lane, RowChunk& w)
{
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int kb = 1024 * c + 256 * i, k = kb + 4 * lane;
        if (kb + 256 <= a.K) { w.pa[i] = wload4(p.w0 + k); w.pb[i] = wload4(p.w1 + k); }
        else if (kb < a.K) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            w.pa[i] = z; w.pb[i] = z;
            if (k < a.K) { w.pa[i] = wload4(p.w0 + k); w.pb[i] = wload4(p.w1 + k); }
        }
    }
}

// Lane tree: the xor butterfly 32, 16, 8, 4, 2, 1 ("v += shfl_xor(v, o)").  For BQ values per lane it runs as a reduce-scatter - at
// the first log2(BQ) steps a lane keeps half of its values and hands the other half to its partner - which computes, for every value,
// exactly the sums of the plain butterfly (own + partner's, fp addition commutes) with 2 BQ + ... instead of 6 BQ exchanges: the same
// bits for every BQ.  Afterwards value b sits in the lanes with (lane >> (6 - log2 BQ)) == b.
// One reduce-scatter step of the lane tree over H value pairs (v[j], v[j + H]): afterwards lanes with (lane & o) == 0 hold
// v[j][l] + v[j][l ^ o] in v[j], the others v[j + H][l] + v[j + H][l ^ o].  o = 32 / 16: gfx950's v_permlane32_swap / v_permlane16_swap
// exchange the upper half (odd rows) of one register with the lower half (even rows) of the other - one swap and one add per pair, no
// selects, no LDS permute (round-6 first form: two selects + ds_bpermute + add).  o = 8: both sums by DPP row rotation, one select.
template <int H, int N>
__device__ __forceinline__ void tree_step(float (&v)[N], int lane, int o)
{
#pragma unroll
    for (int j = 0; j < H; ++j) {                     // (H is a template constant: v is indexed statically and stays in registers)
        if (o == 32) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + H]), false, false);
            v[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else if (o == 16) {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[j]), __float_as_uint(v[j + H]), false, false);
            v[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else {
            const float t0 = xor_add8(v[j]), t1 = xor_add8(v[j + H]);
            v[j] = (lane & 8) ? t1 : t0;
        }
    }
}
template <int BQ>
__device__ __forceinline__ void lane_tree(float (&v)[BQ], int lane)
{
    if constexpr (BQ == 8) { tree_step<4>(v, lane, 32); tree_step<2>(v, lane, 16); tree_step<1>(v, lane, 8); }
    if constexpr (BQ == 4) { tree_step<2>(v, lane, 32); tree_step<1>(v, lane, 16); }
    if constexpr (BQ == 2) { tree_step<1>(v, lane, 32); }
    // the plain steps that are left (all lanes of a slot's group end up with the total)
    if constexpr (BQ == 1) v[0] += __shfl_xor(v[0], 32, 64);
    if constexpr (BQ <= 2) v[0] += __shfl_xor(v[0], 16, 64);
    if constexpr (BQ <= 4) v[0] = xor_add8(v[0]);
    v[0] = xor_add4(v[0]);
    v[0] = xor_add2(v[0]);
    v[0] = xor_add1(v[0]);
}
template <int BQ> struct SlotShift { static constexpr int value = BQ == 8 ? 3 : BQ == 4 ? 4 : BQ == 2 ? 5 : 6; };

// LDS image of one chunk of the BQ input vectors.  BQ >= 2: slots interleaved in pairs, xs[(b >> 1)][k][b & 1], so that the two
// slots of a pair sit in one 64-bit register pair and a v_pk_fma_f32 advances both (two independent fp32 FMAs: the bits of fmaf).
template <int BQ>
__device__ __forceinline__ void stage_chunk(const GemvArgs& a, int bofs, int c, int Kin, float* xs, float (&ss)[BQ])
{
    const int k = 1024 * c + 4 * (int)threadIdx.x;
    const float* const xg = a.x + (int64_t)bofs * a.x_stride + k;
    const bool in = k < Kin;                          // (Kin is a multiple of 4; wave-uniform whenever it is a multiple of 256)
    f32x4 gk = {1.f, 1.f, 1.f, 1.f};
    if (in && a.gamma) gk = aload4(a.gamma + k);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};             // zero padding of the last chunk: a partial strip multiplies it with zero weights
    const int t4 = 4 * (int)threadIdx.x;
    if (BQ == 1) {
        const f32x4 v = in ? aload4(xg) : z;
#pragma unroll
        for (int e = 0; e < 4; ++e) ss[0] = fmaf(v[e], v[e], ss[0]);
        *reinterpret_cast<f32x4*>(xs + t4) = v * gk;
    } else {
        // HB slots per pass: their 16-byte loads are independent (one L2 round trip per pass), then slot pair by slot pair with a
        // scheduling barrier in between - left alone the scheduler keeps the raw values, the products and the interleaved copies of all
        // eight slots live at once (96 registers on top of the weight strips: two blocks per CU).  BQ = 8 stages in two passes of four
        // slots under -DCVX_T2S_STAGE_HALF=1 (dev): 152 -> 144 registers only - still three blocks per CU - for a second round trip: off.
        constexpr int HB = (BQ == 8 && STAGE_HALF) ? 4 : BQ;
#pragma unroll
        for (int h = 0; h < BQ / HB; ++h) {
            f32x4 v[HB];
#pragma unroll
            for (int b = 0; b < HB; ++b) v[b] = in ? aload4(xg + (int64_t)(h * HB + b) * a.x_stride) : z;
#pragma unroll
            for (int q = 0; q < HB / 2; ++q) {
                const int bp = h * (HB / 2) + q;
                const f32x4 va = v[2 * q], vb = v[2 * q + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) { ss[2 * bp] = fmaf(va[e], va[e], ss[2 * bp]); ss[2 * bp + 1] = fmaf(vb[e], vb[e], ss[2 * bp + 1]); }
                const f32x4 pa = va * gk, pb = vb * gk;
                const f32x4 lo = {pa[0], pb[0], pa[1], pb[1]};
                const f32x4 hi = {pa[2], pb[2], pa[3], pb[3]};
                float* const d = xs + ((size_t)bp * 1024 + t4) * 2;
                *reinterpret_cast<f32x4*>(d) = lo;
                *reinterpret_cast<f32x4*>(d + 4) = hi;
                if (BQ >= 4) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// accumulators of one output row over the BQ slots of a group: slot pairs as 64-bit register pairs (v_pk_fma_f32 operands)
template <int BQ> struct Acc {
    f32x2 p[(BQ + 1) / 2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < (BQ + 1) / 2; ++i) p[i] = f32x2{0.f, 0.f};
    }
    __device__ __forceinline__ void unpack(float (&v)[BQ]) const {
#pragma unroll
        for (int b = 0; b < BQ; ++b) v[b] = p[b >> 1][b & 1];
    }
};

// one 256-float strip of a row pair against the staged vectors of the BQ slots
template <int BQ>
__device__ __forceinline__ void dot_strip(const float* xs, int xk, const f32x4 a0, const f32x4 a1, Acc<BQ>& acc0, Acc<BQ>& acc1)
{
    if (BQ == 1) {
        const f32x4 xw = *reinterpret_cast<const f32x4*>(xs + xk);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc0.p[0][0] = fmaf(a0[e], xw[e], acc0.p[0][0]); acc1.p[0][0] = fmaf(a1[e], xw[e], acc1.p[0][0]); }
    } else {
#pragma unroll
        for (int bp = 0; bp < BQ / 2; ++bp) {
            const float* const sp = xs + ((size_t)bp * 1024 + xk) * 2;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 4);
            const f32x2 xe[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc0.p[bp] = fma2(splat2(a0[e]), xe[e], acc0.p[bp]); acc1.p[bp] = fma2(splat2(a1[e]), xe[e], acc1.p[bp]); }
        }
    }
}

// dot products of one row pair with one staged chunk: acc0 row 0, acc1 row 1 (strips past K are skipped wave-uniformly; inside a
// partial strip the lanes past K multiply zero weights with the zero padding stage_chunk wrote)
template <int BQ>
__device__ __forceinline__ void dot_chunk(const GemvArgs& a, int c, int xoff, const float* xs, const RowChunk& w, int lane,
                                          Acc<BQ>& acc0, Acc<BQ>& acc1)
{
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        if (1024 * c + 256 * i < a.K) dot_strip<BQ>(xs, xoff + 4 * lane + 256 * i, w.pa[i], w.pb[i], acc0, acc1);
    }
}

// the MODE epilogue of one (row pair, slot): s0 / s1 = the two rows' finished dot products
template <int MODE>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, const PairInfo& p, int slot, float s0, float s1)
{
    float* const yb = a.y + (int64_t)slot * a.y_stride;
    const int r0 = p.r0, r1 = p.r1;
    if (a.bias) { s0 += a.bias[r0]; if (p.has1) s1 += a.bias[r1]; }
    if (MODE == MODE_QKV) {
        // every slot decodes at its own position (continuous batching: slots are refilled at different steps)
        const int pos = min(aloadi(a.state + SR * slot), a.max_len - 1);
        const int sec = r0 / a.inner, c0 = r0 - sec * a.inner;        // column inside q / k / v
        if (sec < 2) {                                                // half-split RoPE on the (i, i+32) pair
            const float c = a.rope_cos[pos * 32 + (c0 & 31)], s = a.rope_sin[pos * 32 + (c0 & 31)];
            const float n0 = s0 * c - s1 * s, n1 = s1 * c + s0 * s;
            s0 = n0; s1 = n1;
        }
        float* dst = sec == 0 ? yb : (sec == 1 ? a.k_cache : a.v_cache) + slot * a.cache_stride + (int64_t)pos * a.inner;
        dst[c0] = s0;
        dst[c0 + 32] = s1;
    } else if (MODE == MODE_RES) {
        yb[r0] = aload(yb + r0) + s0;
        if (p.has1) yb[r1] = aload(yb + r1) + s1;
    } else if (MODE == MODE_GEGLU) {
        yb[r0] = s0 * gelu_erf(s1);                                   // F.gelu(gate) * x, text2semantic.py:154-157
    } else if (MODE == MODE_LOGITS) {
        yb[p.sidx * a.N + r0] = s0;
        if (p.has1) yb[p.sidx * a.N + r1] = s1;
    } else {
        yb[r0] = s0;
        if (p.has1) yb[r1] = s1;
    }
}

// y[slot] = epilogue(W . norm(x[slot])) for the slots of the block's groups.  PPW = row pairs per wave (the wave's weight strips stay in
// registers for every group of slots it walks); BQ = slots per group.  The input vectors go through LDS in chunks of 1024 floats (32
// KiB at BQ = 8: four blocks per CU whatever K is), staged with ONE 16-byte load per thread and slot - the launch is one link of a
// chain of 34 dependent launches per token, and its critical path is [weight strip | x chunk] -> FMAs -> lane tree -> store.
// (Round 5 staged scalar-wise, four dependent L2 round trips per 1024 floats, ran the lane tree once per value - 96 exchanges per
// pair at BQ = 8 against 20 now - and left the epilogue of all slots to lane 0.)  Per-(row, slot) arithmetic does not depend on BQ, PPW
// or the grouping: same bits.
template <int MODE, int BQ, int PPW, bool LOOP = false>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a)
{
    __shared__ __attribute__((aligned(16))) float xs[BQ * 1024];
    __shared__ float red[BQ][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int Kin = (MODE == MODE_LOGITS) ? a.K * a.streams : a.K;     // staged length per slot
    const int nchunk = (Kin + 1023) >> 10;
    // block -> (row block, first slot group).  More than one block per row block (gy > 1): the gy blocks of a row block are
    // consecutive ON ONE XCD (blocks are dealt to the 8 XCDs round robin), so the first one brings the rows into that XCD's L2
    int rb = blockIdx.x, g0 = 0;
    if (a.gy > 1) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        rb = (i / a.gy) * 8 + xcd;
        g0 = (i % a.gy) * a.gl;
        if (rb >= a.n_blocks) return;                                   // (block-uniform)
    }
    PairInfo pi[PPW];
    RowChunk w[PPW];                                                    // the current chunk of the wave's weight rows (K <= 1024: all of them,
#pragma unroll                                                          // loaded once for every group of slots the block walks)
    for (int p = 0; p < PPW; ++p) pi[p] = pair_info<MODE>(a, (rb * 4 + wid) * PPW + p);
    load_chunk(a, pi[0], 0, lane, w[0]);                                // weights first: in flight under the staging of x
    // (the strips of a second row pair are requested BEHIND the staging, in flight under the first pair's dot products: 32 registers
    //  fewer across the staging - 204 -> 170 VGPRs at two pairs per wave; with the group loop a template constant, 152 -> 118 at one)
    constexpr int SH = SlotShift<BQ>::value;
    const int myb = BQ == 1 ? 0 : lane >> SH;                          // the slot (of a group) whose results the lane tree leaves here
    const bool writer = (lane & ((1 << SH) - 1)) == 0;
    // LOOP (dev hint cvx_t2s_decoder.group_loop > 1): the block walks a.gl groups of slots with its weight strips in registers; the default
    // is ONE group per block - no loop, so nothing (the second pair's strips in particular) has to stay live across a back edge
    const int n_g = LOOP ? a.gl : 1;
#pragma unroll 1
    for (int g = 0; g < n_g; ++g) {
        const int bofs = (g0 + g) * BQ;
        Acc<BQ> acc0[PPW], acc1[PPW];
        float ss[BQ];
#pragma unroll
        for (int b = 0; b < BQ; ++b) ss[b] = 0.f;
#pragma unroll
        for (int p = 0; p < PPW; ++p) { acc0[p].zero(); acc1[p].zero(); }
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            const bool reload = c > 0 || (g > 0 && nchunk > 1);
            if (reload) load_chunk(a, pi[0], c, lane, w[0]);
            if (g | c) __syncthreads();                                 // (xs / red of the previous chunk / group are free)
            stage_chunk<BQ>(a, bofs, c, Kin, xs, ss);
            __syncthreads();
            if (PPW > 1 && (reload || (g | c) == 0)) {
#pragma unroll
                for (int p = 1; p < PPW; ++p) load_chunk(a, pi[p], c, lane, w[p]);
            }
#pragma unroll
            for (int p = 0; p < PPW; ++p) {
                // LOGITS: the staged vector holds `streams` slices; the pair's slice starts at sidx * K (one chunk: Kin <= 1024)
                const int xoff = (MODE == MODE_LOGITS) ? pi[p].sidx * a.K : 0;
                dot_chunk<BQ>(a, c, xoff, xs, w[p], lane, acc0[p], acc1[p]);
            }
        }
        float inv = 1.f;
        if (a.gamma) {                                                  // F.normalize(eps = 1e-12) * sqrt(dim)
            lane_tree<BQ>(ss, lane);
            if (writer) red[myb][wid] = ss[0];
            __syncthreads();
            const float tot = red[myb][0] + red[myb][1] + red[myb][2] + red[myb][3];
            inv = sqrtf((float)Kin) / fmaxf(sqrtf(tot), 1e-12f);
        }
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            if (!pi[p].valid) {
                if (MODE == MODE_GEGLU) {      // zero the K padding of the consumer GEMV
                    const int pair = (rb * 4 + wid) * PPW + p;
                    if (pair >= a.N / 2 && pair < a.y_pad && lane < BQ) a.y[(int64_t)(bofs + lane) * a.y_stride + pair] = 0.f;
                }
                continue;
            }
            float v0[BQ], v1[BQ];
            acc0[p].unpack(v0);
            acc1[p].unpack(v1);
            lane_tree<BQ>(v0, lane);
            lane_tree<BQ>(v1, lane);
            if (writer) gemv_epilogue<MODE>(a, pi[p], bofs + myb, v0[0] * inv, v1[0] * inv);
        }
    }
}

// ---------------------------------------------------------------- attention of ONE query over n cached keys
struct AttnArgs {
    const float* q;          // [batch][heads*64]
    const float* k;          // key j of head h at k + B*batch_stride + j*stride + h*64; B = the slot (self-attention cache) or the
    const float* v;          // dialogue the slot is decoding (cross-attention context, n_fixed == -2 / by_dialogue)
    int64_t stride, batch_stride;
    float* out;              // [batch][heads*64]
    const int* state;        // [batch][SR]
    int n_fixed;             // >= 0: that many keys; -1: state[slot][0] + 1 (self-attention); -2: state[slot][3] (context length)
    int by_dialogue;         // 1: k / v are per DIALOGUE (state[slot][4]), not per slot
    float scale;
    int max_len;
};

// sc: T2S_MAX_KEYS floats, red: 4 floats, part: 16 x 64 floats (16-byte aligned) of LDS; all 256 threads of the block
__device__ __forceinline__ void attn_body(const AttnArgs& a, int h, int b, int heads, float* sc, float* red, float (*part)[64])
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int HD64 = heads * 64;
    // an idle slot (position max_len: a drained dialogue queue, or a slot past its last step) has nothing to attend to - without
    // this exit it would walk max_len stale cache rows every step (block-uniform)
    if (a.n_fixed < 0 && aloadi(a.state + SR * b) >= a.max_len) return;
    const int n = a.n_fixed >= 0 ? a.n_fixed
                                 : (a.n_fixed == -1 ? min(aloadi(a.state + SR * b) + 1, a.max_len) : min(aloadi(a.state + SR * b + 3), T2S_MAX_KEYS));
    const int64_t kvb = a.by_dialogue ? aloadi(a.state + SR * b + 4) : b;
    const float* const kb = a.k + kvb * a.batch_stride;
    const float* const vb = a.v + kvb * a.batch_stride;
    const int sub = tid & 15, grp = tid >> 4;              // 16 lanes per key, 16 keys per pass
    const f32x4 q4 = aload4(a.q + (int64_t)b * HD64 + h * 64 + 4 * sub);
    float mx = -3.0e38f;
    // four key rows per thread in flight (the loop is a chain of cache round trips otherwise: one 16-byte load, four shuffles and a
    // compare per trip - 38 dependent trips at 600 keys); the arithmetic per key is unchanged
    for (int j0 = 0; j0 < n; j0 += 64) {
        f32x4 k4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 16 * u + grp;
            if (j < n) k4[u] = aload4(kb + (int64_t)j * a.stride + h * 64 + 4 * sub);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 16 * u + grp;
            float d = 0.f;
            if (j < n) d = k4[u][0] * q4[0] + k4[u][1] * q4[1] + k4[u][2] * q4[2] + k4[u][3] * q4[3];
            d = xor_add1(xor_add2(xor_add4(xor_add8(d))));              // the 16 lanes of a key (xor 8, 4, 2, 1: no LDS permute)
            d *= a.scale;
            if (j < n) { if (sub == 0) sc[j] = d; mx = fmaxf(mx, d); }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x128, 0xF, 0xF, false)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(mx), 0x101F)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xF, 0xF, false)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xF, 0xF, false)));
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < n; j += 256) { const float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = grp; j0 < n; j0 += 64) {              // (value rows four at a time; accumulated in ascending key order as before)
        f32x4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (j0 + 16 * u < n) v4[u] = aload4(vb + (int64_t)(j0 + 16 * u) * a.stride + h * 64 + 4 * sub);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + 16 * u < n) {
                const float p = sc[j0 + 16 * u];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(p, v4[u][e], acc[e]);
            }
        }
    }
    *reinterpret_cast<f32x4*>(&part[grp][4 * sub]) = acc;
    __syncthreads();
    if (tid < 64) {
        float o = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) o += part[g][tid];
        a.out[(int64_t)b * HD64 + h * 64 + tid] = o / sum;
    }
}

__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a)
{
    __shared__ float sc[T2S_MAX_KEYS];
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float part[16][64];
    attn_body(a, blockIdx.x, blockIdx.y, gridDim.x, sc, red, part);
}

// ---------------------------------------------------------------- top-k + Gumbel argmax, eos bookkeeping, next input
struct SampleArgs {
    const float* logits;     // [batch][streams, V]
    const float* uniforms;   // [dialogue][uniform_steps][streams, V]
    const float* emb;        // [V, dim_emb]
    float* x;                // [batch][streams * dim_emb]  next step's input (residual stream)
    int64_t* tokens;         // [dialogue][streams, max_len]
    int* state;              // [batch][SR]: [0] pos  [1] done  [2] length at the first eos  [3] context rows  [4] dialogue  [5] step limit
                             //              [6] flags (bit 0: the eos does not end the dialogue)
    int* queue;              // NULL, or {next pending dialogue, number of dialogues}: continuous batching
    int* dialogues;          // [n][SR] (queue != NULL): in [0] context rows [1] step limit [2] flags; out [3] status (0 pending, 1 running,
                             //   2 ended by its eos, 3 by its limit) [4] steps decoded [5] the slot it ran in
    const float* start;      // [streams * dim_emb] start token (queue != NULL): the input of a refilled slot
    int batch, uniform_steps;
    int V, dim_emb, streams, max_len, top_k, eos_id;
    float inv_temp;
    float cfg_scale;         // > 1: classifier-free guidance (text2semantic.py:780-792) - slots 2u (text context) and 2u + 1 (context
                             // masked out: the learned null key / value only) decode the SAME tokens: slot 2u samples from
                             // null + (cond - null) * cfg_scale and feeds both; one-output models
};

// NT threads (a multiple of 64, <= 1024); every thread owns the vocabulary entries tid, tid + NT, ...  The selection is an
// exact function of the logits and the uniforms (rank counting, then argmax with the lowest index on ties), so it does not
// depend on NT.  lg: 1024 floats, bv / bi: 16 entries, chosen: one int of LDS.
template <int NT>
__device__ __forceinline__ void sample_body(const SampleArgs& a, int b, float* lg, float* bv, int* bi, int* chosen)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool cfg = a.cfg_scale > 1.0f;
    if (cfg && (b & 1)) return;                     // the null-context slot follows its partner (block-uniform)
    int* const state = a.state + SR * b;
    const int pos = aloadi(state);
    if (pos >= a.max_len) return;                   // (block-uniform; also: an idle slot of a drained queue)
    const int64_t dlg = aloadi(state + 4);          // the dialogue this slot decodes (== b without a queue)
    bool eos = false;
    for (int s = 0; s < a.streams; ++s) {
        for (int i = tid; i < a.V; i += NT) {
            const float c = aload(a.logits + ((int64_t)b * a.streams + s) * a.V + i);
            if (cfg) {          // null_logits + (logits - null_logits) * cond_scale, the reference's operation order (no contraction here)
                const float n = aload(a.logits + ((int64_t)(b + 1) * a.streams + s) * a.V + i);
                lg[i] = n + (c - n) * a.cfg_scale;
            } else lg[i] = c;
        }
        if (tid < 4 && a.V + tid < ((a.V + 3) & ~3)) lg[a.V + tid] = -INFINITY;      // (the rank count below reads whole 4-vectors)
        __syncthreads();
        float val = -INFINITY;
        int idx = tid;
        for (int i = tid; i < a.V; i += NT) {
            const float me = lg[i];
            int cnt = 0;
            for (int j = 0; j < a.V; j += 4) {       // rank of entry i = number of larger logits (exact: any order gives the same count)
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lg + j);
                cnt += (l4[0] > me ? 1 : 0) + (l4[1] > me ? 1 : 0) + (l4[2] > me ? 1 : 0) + (l4[3] > me ? 1 : 0);
            }
            if (cnt < a.top_k) {
                const float u = a.uniforms[((dlg * a.uniform_steps + pos) * a.streams + s) * a.V + i];
                const float g = -logf(fmaxf(-logf(fmaxf(u, 1e-20f)), 1e-20f));
                const float v = me * a.inv_temp + g;
                if (v > val) { val = v; idx = i; }   // (ascending i: the lowest index wins ties)
            }
        }
        // argmax, lowest index on ties (torch.argmax)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(val, o, 64);
            const int oi = __shfl_xor(idx, o, 64);
            if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        if (lane == 0) { bv[wid] = val; bi[wid] = idx; }
        __syncthreads();
        if (tid == 0) {
            float best = bv[0]; int bt = bi[0];
            for (int w = 1; w < NT / 64; ++w)
                if (bv[w] > best || (bv[w] == best && bi[w] < bt)) { best = bv[w]; bt = bi[w]; }
            *chosen = bt;
            a.tokens[(dlg * a.streams + s) * a.max_len + pos] = bt;
        }
        __syncthreads();
        const int tok = *chosen;
        eos = eos || (tok == a.eos_id);
        for (int d = tid; d < a.dim_emb; d += NT) {
            const float e = a.emb[(int64_t)tok * a.dim_emb + d];
            a.x[((int64_t)b * a.streams + s) * a.dim_emb + d] = e;
            if (cfg) a.x[((int64_t)(b + 1) * a.streams + s) * a.dim_emb + d] = e;
        }
        if (cfg && tid == 0) a.tokens[((dlg + 1) * a.streams + s) * a.max_len + pos] = tok;
        __syncthreads();
    }
    if (a.queue == nullptr) {
        if (tid == 0) {
            int done = aloadi(state + 1), len = aloadi(state + 2);
            if (eos && done == 0) { done = 1; len = pos + 1; state[1] = 1; state[2] = len; }
            state[0] = pos + 1;
            if (cfg) { int* const sn = state + SR; sn[1] = done; sn[2] = len; sn[0] = pos + 1; }
        }
        return;
    }
    // continuous batching: the dialogue ends with its first eos (text2semantic.py:803-818) or at its step limit; the slot then
    // takes the next pending dialogue: position 0, the start token as input, that dialogue's context / uniforms / token rows
    const bool ends_eos = eos && !(aloadi(state + 6) & 1);
    const bool ends = ends_eos || pos + 1 >= aloadi(state + 5);        // (block-uniform: `eos` comes from LDS, the record is read by all)
    if (!ends) {
        if (tid == 0) state[0] = pos + 1;
        return;
    }
    if (tid == 0) {
        int* const dr = a.dialogues + SR * dlg;
        dr[4] = pos + 1;
        dr[5] = b;
        __threadfence();
        dr[3] = ends_eos ? 2 : 3;
        const int nxt = atomicAdd(a.queue, 1);
        *chosen = nxt < aloadi(a.queue + 1) ? nxt : -1;
    }
    __syncthreads();
    const int nxt = *chosen;
    if (nxt < 0) {                                  // nothing pending: the slot idles (every kernel clamps / skips at max_len)
        if (tid == 0) { state[0] = a.max_len; state[1] = 1; }
        return;
    }
    for (int d = tid; d < a.dim_emb * a.streams; d += NT) a.x[(int64_t)b * a.streams * a.dim_emb + d] = a.start[d];
    if (tid == 0) {
        int* const dn = a.dialogues + SR * nxt;
        state[0] = 0; state[1] = 0; state[2] = 0; state[3] = dn[0]; state[4] = nxt; state[5] = dn[1]; state[6] = dn[2];
        dn[5] = b;
        dn[3] = 1;
    }
}

__global__ __launch_bounds__(1024) void sample_kernel(const SampleArgs a)
{
    __shared__ __attribute__((aligned(16))) float lg[1024];
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int chosen;
    sample_body<1024>(a, blockIdx.x, lg, bv, bi, &chosen);
}

__global__ __launch_bounds__(256) void geglu_kernel(const float* __restrict__ h, float* __restrict__ out, int64_t rows,
                                                   int F, int64_t ld_out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld_out) return;
    const int64_t r = i / ld_out;
    const int c = (int)(i - r * ld_out);
    out[i] = c < F ? h[r * 2 * F + c] * gelu_erf(h[r * 2 * F + F + c]) : 0.f;
}


constexpr int T2S_GROUP_LOOP = 1;   // default of cvx_t2s_decoder.group_loop: slot groups one block walks with its weight rows in registers
template <int MODE, int BQ, int PPW>
void launch_gemv_p(GemvArgs g, int pairs, int groups, hipStream_t st)
{
    g.n_blocks = (pairs + 4 * PPW - 1) / (4 * PPW);
    const int want = g.gl > 0 ? g.gl : T2S_GROUP_LOOP;
    g.gl = 1;
    while (g.gl < want && groups % (2 * g.gl) == 0) g.gl *= 2;
    g.gy = groups / g.gl;
    const unsigned grid = g.gy > 1 ? (unsigned)((g.n_blocks + 7) / 8 * 8 * g.gy) : (unsigned)g.n_blocks;
    if constexpr (PPW == 1 && BQ == 8) {
        if (g.gl > 1) { hipLaunchKernelGGL((gemv_kernel<MODE, BQ, PPW, true>), dim3(grid), dim3(256), 0, st, g); return; }
    }
    if (g.gl > 1) { g.gy *= g.gl; g.gl = 1; }          // (the group loop exists at one row pair per wave, eight slots per group only)
    const unsigned grid1 = g.gy > 1 ? (unsigned)((g.n_blocks + 7) / 8 * 8 * g.gy) : (unsigned)g.n_blocks;
    hipLaunchKernelGGL((gemv_kernel<MODE, BQ, PPW, false>), dim3(grid1), dim3(256), 0, st, g);
}
template <int MODE, int BQ>
void launch_gemv_b(const GemvArgs& g, int pairs, int groups, bool two, hipStream_t st)
{
    if constexpr (BQ >= 8) {
        if (two) { launch_gemv_p<MODE, BQ, 2>(g, pairs, groups, st); return; }
    }
    launch_gemv_p<MODE, BQ, 1>(g, pairs, groups, st);
}

// batch 1 / 2 / 4 / 8: one group of that many slots; above: ceil(batch / 8) groups of 8 (the per-slot buffers of the caller hold
// whole groups; the slots past `batch` compute on whatever they hold and nothing reads them)
template <int MODE>
void launch_gemv(const GemvArgs& g, int pairs, int batch, bool few_cus, hipStream_t st)
{
    if (batch <= 1) launch_gemv_b<MODE, 1>(g, pairs, 1, few_cus, st);
    else if (batch <= 2) launch_gemv_b<MODE, 2>(g, pairs, 1, few_cus, st);
    else if (batch <= 4) launch_gemv_b<MODE, 4>(g, pairs, 1, few_cus, st);
    else launch_gemv_b<MODE, 8>(g, pairs, (batch + 7) / 8, few_cus, st);
}

}  // namespace

extern "C" int cvx_geglu_f32(const float* h, float* out, int64_t rows, int32_t F, int64_t ld_out, cvx_stream_t s)
{
    CVX_REQUIRE(h && out && rows >= 0 && F > 0 && ld_out >= F, "geglu: bad arguments");
    if (rows == 0) return CVX_OK;
    const int64_t n = rows * ld_out;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       h, out, rows, F, ld_out);
    CVX_CHECK_LAUNCH("cvx_geglu_f32");
    return CVX_OK;
}

static int t2s_validate(const cvx_t2s_decoder* d, int32_t n_steps)
{
    CVX_REQUIRE(d && d->layers && n_steps >= 0, "t2s_decode: null decoder");
    CVX_REQUIRE(d->dim > 0 && d->dim % 4 == 0 && d->dim <= T2S_MAX_DIM && d->inner == d->heads * 64 && d->depth > 0 &&
                d->streams >= 1 && d->streams <= 2 && d->dim_emb * d->streams == d->dim && d->dim_emb % 4 == 0 &&
                (d->streams == 1 || d->dim <= 1024) && d->vocab > 0 && d->vocab <= 1024 &&
                d->ff_inner > 0 && d->ff_inner_pad >= d->ff_inner && d->ff_inner_pad % 4 == 0 && d->ff_inner_pad <= T2S_MAX_DIM &&
                d->n_ctx >= 0 && d->n_ctx <= T2S_MAX_KEYS && d->max_len > 0 && d->max_len <= T2S_MAX_KEYS &&
                d->top_k > 0 && d->top_k <= d->vocab && d->temperature >= 0.f && d->batch >= 1 && d->batch <= T2S_MAX_BATCH &&
                d->uniform_steps > 0 &&
                d->ctx_rows > 0 && d->ctx_rows <= T2S_MAX_KEYS && d->n_ctx <= d->ctx_rows,
                "t2s_decode: bad dimensions (dim=%d inner=%d heads=%d streams=%d dim_emb=%d vocab=%d ff=%d/%d n_ctx=%d/%d max_len=%d batch=%d)",
                d->dim, d->inner, d->heads, d->streams, d->dim_emb, d->vocab, d->ff_inner, d->ff_inner_pad, d->n_ctx, d->ctx_rows,
                d->max_len, d->batch);
    CVX_REQUIRE(!(d->cfg_scale > 1.f) || (d->streams == 1 && d->batch % 2 == 0 && d->n_ctx == 0),
                "t2s_decode: guidance (cfg_scale > 1) needs a one-output model, an even batch (context / null-context slot pairs) and "
                "per-slot context rows (n_ctx == 0)");
    CVX_REQUIRE(!d->queue || (d->dialogues && d->start && !(d->cfg_scale > 1.f) && d->n_ctx == 0),
                "t2s_decode: a dialogue queue needs the dialogue records and the start token, per-dialogue context rows (n_ctx == 0) and no guidance");
    CVX_REQUIRE(d->final_gamma && d->emb && d->rope_cos && d->rope_sin && d->uniforms && d->x && d->q && d->att && d->h &&
                d->logits && d->tokens && d->state, "t2s_decode: null buffer");
    for (int l = 0; l < d->depth; ++l) {
        const cvx_t2s_layer& L = d->layers[l];
        CVX_REQUIRE(L.gamma_s && L.wqkv_s && L.wo_s && L.gamma_c && L.wq_c && L.wo_c && L.kv_c && L.gamma_f && L.w1 && L.b1 &&
                    L.w2 && L.b2 && L.k_cache && L.v_cache, "t2s_decode: null pointer in layer %d", l);
    }
    return CVX_OK;
}

extern "C" int cvx_t2s_decode_steps(const cvx_t2s_decoder* d, int32_t n_steps, cvx_stream_t s)
{
    const int rc = t2s_validate(d, n_steps);
    if (rc != CVX_OK) return rc;
    hipStream_t st = cvx_hip_stream(s);
    const float scale = 0.125f;        // dim_head ** -0.5
    const int nb = d->batch;
    // ONE row pair per wave everywhere since the kernel holds 118 VGPRs (four blocks per CU): 64 slots 462 vs 480 us per CoMix step with two
    // pairs (170 VGPRs, two blocks), 32-CU side stream at 8 slots 434 vs 439 (round 5, at two blocks per CU either way, two pairs won there:
    // 760 vs 813).  cvx_t2s_decoder.pairs_per_wave = 2 still selects the other form (same bits).
    const bool few = d->pairs_per_wave >= 2;
    const int64_t cache_stride = (int64_t)d->max_len * d->inner;
    for (int step = 0; step < n_steps; ++step) {
        for (int l = 0; l < d->depth; ++l) {
            const cvx_t2s_layer& L = d->layers[l];
            GemvArgs g{};
            g.gl = d->group_loop;
            // self-attention: q | k | v with RoPE; k, v appended to the cache at position pos
            g.W = L.wqkv_s; g.ldw = d->dim; g.x = d->x; g.x_stride = d->dim; g.gamma = L.gamma_s; g.y = d->q; g.y_stride = d->inner;
            g.N = 3 * d->inner; g.K = d->dim;
            g.inner = d->inner; g.rope_cos = d->rope_cos; g.rope_sin = d->rope_sin; g.k_cache = L.k_cache; g.v_cache = L.v_cache;
            g.cache_stride = cache_stride; g.state = d->state; g.max_len = d->max_len;
            launch_gemv<MODE_QKV>(g, 3 * d->inner / 2, nb, few, st);
            AttnArgs at{d->q, L.k_cache, L.v_cache, d->inner, cache_stride, d->att, d->state, -1, 0, scale, d->max_len};
            hipLaunchKernelGGL(attn_kernel, dim3((unsigned)d->heads, (unsigned)nb), dim3(256), 0, st, at);
            g = GemvArgs{};
            g.gl = d->group_loop;
            g.W = L.wo_s; g.ldw = d->inner; g.x = d->att; g.x_stride = d->inner; g.y = d->x; g.y_stride = d->dim; g.N = d->dim; g.K = d->inner;
            launch_gemv<MODE_RES>(g, (d->dim + 1) / 2, nb, few, st);
            // cross-attention over [null kv | encoder context]
            g = GemvArgs{};
            g.gl = d->group_loop;
            g.W = L.wq_c; g.ldw = d->dim; g.x = d->x; g.x_stride = d->dim; g.gamma = L.gamma_c; g.y = d->q; g.y_stride = d->inner;
            g.N = d->inner; g.K = d->dim;
            launch_gemv<MODE_PLAIN>(g, d->inner / 2, nb, few, st);
            AttnArgs ac{d->q, L.kv_c, L.kv_c + d->inner, 2 * (int64_t)d->inner, (int64_t)d->ctx_rows * 2 * d->inner, d->att, d->state,
                        d->n_ctx > 0 ? d->n_ctx : -2, 1, scale, d->max_len};
            hipLaunchKernelGGL(attn_kernel, dim3((unsigned)d->heads, (unsigned)nb), dim3(256), 0, st, ac);
            g = GemvArgs{};
            g.gl = d->group_loop;
            g.W = L.wo_c; g.ldw = d->inner; g.x = d->att; g.x_stride = d->inner; g.y = d->x; g.y_stride = d->dim; g.N = d->dim; g.K = d->inner;
            launch_gemv<MODE_RES>(g, (d->dim + 1) / 2, nb, few, st);
            // GEGLU feed-forward
            g = GemvArgs{};
            g.gl = d->group_loop;
            g.W = L.w1; g.ldw = d->dim; g.x = d->x; g.x_stride = d->dim; g.gamma = L.gamma_f; g.bias = L.b1; g.y = d->h;
            g.y_stride = d->ff_inner_pad; g.N = 2 * d->ff_inner; g.K = d->dim; g.y_pad = d->ff_inner_pad;
            launch_gemv<MODE_GEGLU>(g, d->ff_inner_pad, nb, few, st);
            g = GemvArgs{};
            g.gl = d->group_loop;
            g.W = L.w2; g.ldw = d->ff_inner_pad; g.x = d->h; g.x_stride = d->ff_inner_pad; g.bias = L.b2; g.y = d->x; g.y_stride = d->dim;
            g.N = d->dim; g.K = d->ff_inner_pad;
            launch_gemv<MODE_RES>(g, (d->dim + 1) / 2, nb, few, st);
        }
        GemvArgs g{};
        g.gl = d->group_loop;
        g.W = d->emb; g.ldw = d->dim_emb; g.x = d->x; g.x_stride = d->dim; g.gamma = d->final_gamma; g.y = d->logits;
        g.y_stride = d->streams * d->vocab; g.N = d->vocab; g.K = d->dim_emb; g.streams = d->streams;
        launch_gemv<MODE_LOGITS>(g, d->streams * ((d->vocab + 1) / 2), nb, few, st);
        SampleArgs sa{d->logits, d->uniforms, d->emb, d->x, d->tokens, d->state, d->queue, d->dialogues, d->start, nb, d->uniform_steps,
                      d->vocab, d->dim_emb, d->streams, d->max_len, d->top_k, d->vocab - 1, 1.0f / fmaxf(d->temperature, 1e-10f),
                      d->cfg_scale};
        hipLaunchKernelGGL(sample_kernel, dim3((unsigned)nb), dim3(1024), 0, st, sa);
    }
    CVX_CHECK_LAUNCH("cvx_t2s_decode_steps");
    return CVX_OK;
}
