// Split-precision GEMM, LARGE-problem kernel (2048 rows and more, 512 columns and more): 256 x 256 block tile, eight waves, EIGHT-PHASE
// PING-PONG main loop on v_mfma_f32_16x16x32_f16 with SWAPPED operands, persistent blocks, transpose-free epilogues.
//
// Contract of cvx_gemm_f16x3 (C = epi([A|A2] * W^T) with three MFMA products per tile on (fp16 hi, fp16 lo) operand pairs; every nn.Linear
// of reference acoustic.py:225-246, :306-310) for INTERLEAVED operands: one K-tile (32 k) of a row of A or W is one 128-byte line
// [hi 32 | lo 32].
//   * the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; waves w and w+4 share a SIMD) run ONE BARRIER INTERVAL
//     APART: while one group issues the MFMAs of a phase the other reads its next fragments from LDS and issues its share of the
//     LDS-DMA - the SIMD's matrix pipe always has a wave feeding it;
//   * a K-tile is four phases, one 64 x 32 quadrant of the wave's 128 x 64 output each, in the order (m0,n0) (m0,n1) (m1,n1) (m1,n0), so
//     that every phase needs at most one new operand block (A m0 + B n0, B n1, A m1, nothing) and the operand QUARTERS of a K-tile
//     are consumed progressively;
//   * the DMA stream runs SIX quarter-tiles (96 KiB) ahead in a two-buffer ring: a quarter (16 KiB: the rows one phase block needs,
//     for all waves) is re-filled two intervals after its last reader, i.e. up to 9 intervals before its first one, and is retired by
//     a COUNTED s_waitcnt vmcnt(8) (never 0) followed by a barrier;
//   * the DMA is issued from inline asm (scalar base + 32-bit lane offset, M0 = LDS slot; gemm_p8s_epi.h dma2): hipcc puts a vmcnt(0)
//     in front of every ds_read that follows a global_load_lds it can see;
//   * 16x16x32 instead of 32x32x16: an accumulator register is read and written once per 32 k instead of once per 16 k; what limits
//     the loop is the clock the chip sustains under that load (power), and this shape moves fewer register bytes per MAC;
//   * swapped operands, D = W_frag . A_frag^T: a lane then holds 4 CONSECUTIVE COLUMNS of one output row (row = lane & 15, columns
//     4 * (lane >> 4) + 0..3 of a 16 x 16 tile), so every epilogue access (fp32 store, residual load, split store, bias / RoPE table
//     loads) is a 16-byte (8-byte fp16) vector without register transposes.  Blocks that own V columns of a to_qkv projection run
//     the UN-swapped product instead: a lane then holds 4 consecutive FRAMES of one head-dim column - what the transposed V^T store
//     wants;
//   * PERSISTENT blocks (one per CU of the stream, cvx_ctx.n_cus) walk tile slots of an XCD-aware map; the tail of a tile fetches the
//     first six quarters of the next one.
// LDS: 2 buffers x (A tile [256][128 B] | W tile [256][128 B]) = 128 KiB + 8 KiB dump area for the tail's dummy DMA.  Swizzle: 16-byte
// chunk c of row r sits at chunk c ^ ((r >> 1) & 7), applied on the DMA source address and on the ds_read_b128 fragment address
// (conflict-free).  Only full 64-column wave tiles with 16-byte aligned epilogue operands are accepted: the launcher returns false
// otherwise and the 128 x 128 kernel (gemm_f16x3_p8m.hip) takes the problem.  (Rounds 2-4 shipped two more large-problem forms - a
// two-stage 256 x 256 kernel and this main loop on the 32x32x16 MFMA: ff1 at 16,000 rows 447 / 399 us against 355 us here - removed in round 5, HISTORY.md.)
#include "gemm_p8s_epi.h"

namespace {

constexpr int TILE_B = 256 * 128;                  // bytes per operand tile
constexpr int BUF_B = 2 * TILE_B;                  // A | W
constexpr int DUMP_B = 2 * BUF_B;                  // dump area offset (8 KiB)
constexpr int LDS_B = DUMP_B + 8 * 1024;

// the main loop for one output tile; SWAP selects the operand order of every MFMA (see the header)
// first: this is the block's first tile (issue the six-quarter prologue); (m0n, n0n): the block's NEXT tile, whose first
// six quarters take the place of the tail's dummy pieces (m0n < 0: no next tile), so that it starts without a prologue.
// PERM (the deferred-norm instances): LDS row rho of the W tile holds weight row perm32(rho) (a permutation inside every 32 rows, free:
// the DMA's source address is per lane) so that a lane's accumulators of a tile PAIR are 8 consecutive output columns - 16-byte
// split-pair loads / stores in the epilogue instead of 8-byte ones (gemm_p8s_epi.h)
// MI (8 or 6): 16-row accumulator tiles per wave = tile HEIGHT 32 * MI (256 or 192 rows; round 5).  The 192-row form runs the same
// loop with three instead of four A tiles per M half: 72 instead of 96 MFMAs per K-tile and wave, 24 instead of 32 A pieces (the
// waves whose A slots fall past row 192 issue dummy pieces so that the counted vmcnt holds) - a tile costs 0.8 of a 256-row one (0.75 of the MFMAs, the same W stream), and
// the launcher takes it where fewer, shorter rounds of tiles come out (9,298 rows x N = 1024: 196 tiles of 192 rows = one round at 0.75
// against 148 tiles of 256 rows = one round at 1.0).  Every output element is the same sum in the same order: same bits.
template <bool HAS_A2, bool SWAP, bool PERM = false, int MI = 8>
__device__ __forceinline__ void tile_mainloop(const cvx_gemm_args& p, const PreSplitA& A, const f16* __restrict__ W, char* smem,
                                              int m0, int n0, int m0n, int n0n, bool first, int lane, int wid, int wr, int wc,
                                              f32x4 (&acc)[MI][4], const float a2_ratio = 1.f)
{
    constexpr int MH = MI / 2;                                   // 16-row tiles per M half of a wave
    constexpr int RG = MI * 16, RH = RG / 2;                     // rows per wave group / per M half of a group
    constexpr int A_PIECES = 2 * RH / 8;                         // 8-row DMA pieces per A quarter (16 issue slots: 2 per wave)
    const bool a_slot = 2 * wid < A_PIECES;                      // this wave's two A slots hold rows of the tile (wave-uniform)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t offA[2][2], offA2[2][2], offW[2][2];                // per-lane source byte offsets of the tile being FETCHED
    uint32_t dstA[2][2], dstW[2][2];                             // LDS byte offsets inside a buffer (wave-uniform)
    const int64_t ldaB = A.ld * 2, lda2B = A.ld2 * 2, ldwB = p.ldw * 2;
    const bool has_next = m0n >= 0;
    auto set_offsets = [&](int h, int mm, int nn) {               // quarters A_h / W_h of the tile at (mm, nn)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int slot = 2 * wid + j, pr0 = 8 * slot;
            const int ra0 = a_slot ? (slot / (RH / 8)) * RG + h * RH + (slot % (RH / 8)) * 8 : 0;
            const int rb0 = (pr0 >> 5) * 64 + h * 32 + (pr0 & 31);
            const int ra = ra0 + (lane >> 3), rb = rb0 + (lane >> 3);
            const uint32_t ca = (uint32_t)(((lane & 7) ^ ((ra >> 1) & 7)) * 16);
            const uint32_t cb = (uint32_t)(((lane & 7) ^ ((rb >> 1) & 7)) * 16);
            const uint32_t ga = (uint32_t)min(mm + ra, p.M - 1), gb = (uint32_t)min(nn + (PERM ? perm32(rb) : rb), p.N - 1);
            offA[h][j] = ga * (uint32_t)ldaB + ca;                 // (every operand spans < 4 GiB: checked by the launcher)
            offA2[h][j] = HAS_A2 ? ga * (uint32_t)lda2B + ca : 0u;
            offW[h][j] = gb * (uint32_t)ldwB + cb;
            dstA[h][j] = (uint32_t)(ra0 * 128);
            dstW[h][j] = (uint32_t)(TILE_B + rb0 * 128);
        }
    };
    set_offsets(0, m0, n0);
    set_offsets(1, m0, n0);
    const int nk = p.K / 32;
    const int t_sw = HAS_A2 ? p.K1 / 32 : 0x7fffffff;
    const char* const a1base = reinterpret_cast<const char*>(A.hi);
    const char* const a2base = reinterpret_cast<const char*>(A.hi2);
    const char* const wbase = reinterpret_cast<const char*>(W);
    const uint32_t dump = lds0 + DUMP_B + (uint32_t)wid * 1024u;

    // quarter A_h / W_h of K-tile tt -> buffer tt & 1.  tt >= nk: K-tile tt - nk of the NEXT tile (nk is even whenever a
    // block has a next tile, so the buffer parity continues; the offsets were switched to that tile before the first such
    // quarter, see the K loop), or, without a next tile, a dummy 16-byte re-read into the dump area that keeps the vmcnt
    // arithmetic uniform
    auto issue_A = [&](int h, int tt) {
        const bool cur = tt < nk;
        const bool live = cur || has_next;
        const int u = cur ? tt : tt - nk;
        const uint32_t b = lds0 + (uint32_t)(tt & 1) * BUF_B;
        const char* base = a1base + (int64_t)u * 128;
        uint32_t v0 = offA[h][0], v1 = offA[h][1];
        if constexpr (HAS_A2) {
            if (u >= t_sw) { base = a2base + (int64_t)(u - t_sw) * 128; v0 = offA2[h][0]; v1 = offA2[h][1]; }
        }
        const bool put = live && a_slot;
        if (!put) { base = wbase; v0 = 0u; v1 = 0u; }
        dma2(v0, v1, put ? b + dstA[h][0] : dump, put ? b + dstA[h][1] : dump, base);
    };
    auto issue_W = [&](int h, int tt) {
        const bool cur = tt < nk;
        const bool live = cur || has_next;
        const int u = cur ? tt : tt - nk;
        const uint32_t b = lds0 + (uint32_t)(tt & 1) * BUF_B;
        const char* base = live ? wbase + (int64_t)u * 128 : wbase;
        dma2(live ? offW[h][0] : 0u, live ? offW[h][1] : 0u, live ? b + dstW[h][0] : dump, live ? b + dstW[h][1] : dump, base);
    };

    // fragment addresses: row (lane & 15) of a 16-row MFMA tile, k chunk (lane >> 4) for hi / 4 + (lane >> 4) for lo
    const int lr = lane & 15, kg = lane >> 4, sw8 = lr >> 1;
    const int aoh = (wr * RG + lr) * 128 + 16 * (kg ^ sw8);
    const int aol = (wr * RG + lr) * 128 + 16 * ((4 + kg) ^ sw8);
    const int boh = TILE_B + (wc * 64 + lr) * 128 + 16 * (kg ^ sw8);
    const int bol = TILE_B + (wc * 64 + lr) * 128 + 16 * ((4 + kg) ^ sw8);

    if (first) {          // (the quarters of a later tile were issued by the previous tile's tail and waited for by its last phases)
        issue_A(0, 0); issue_W(0, 0); issue_W(1, 0); issue_A(1, 0); issue_A(0, 1); issue_W(0, 1);
        CVX_P8_WAIT_DMA();
        CVX_P8_BARRIER();
    }
    if (wr == 1) CVX_P8_BARRIER();                      // group 1 runs one interval behind group 0

    f16x8 fah[MH], fal[MH];                             // A fragments of the current M half (MH tiles of 16 rows)
    f16x8 fbh[2][2], fbl[2][2];                         // W fragments: [n half][tile]

#define CVX_P8S_READ_A(buf, mh)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MH; ++i) {                                                             \
        fah[i] = *reinterpret_cast<const f16x8*>(smem + (buf) * BUF_B + ((mh) * MH + i) * 16 * 128 + aoh);       \
        fal[i] = *reinterpret_cast<const f16x8*>(smem + (buf) * BUF_B + ((mh) * MH + i) * 16 * 128 + aol);       \
    }
#define CVX_P8S_READ_B(buf, nh)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
        fbh[nh][j] = *reinterpret_cast<const f16x8*>(smem + (buf) * BUF_B + ((nh) * 2 + j) * 16 * 128 + boh);    \
        fbl[nh][j] = *reinterpret_cast<const f16x8*>(smem + (buf) * BUF_B + ((nh) * 2 + j) * 16 * 128 + bol);    \
    }
#define CVX_P8S_MM(x, y, c) (SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0))
    // quadrant (mh, nh): MH x 2 tiles x three terms = 24 (18) MFMAs, term-major (consecutive MFMAs hit different accumulators)
#define CVX_P8S_MFMA(mh, nh)                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
        acc[(mh) * MH + i][(nh) * 2 + j] = CVX_P8S_MM(fal[i], fbh[nh][j], acc[(mh) * MH + i][(nh) * 2 + j]);     \
    _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
        acc[(mh) * MH + i][(nh) * 2 + j] = CVX_P8S_MM(fah[i], fbl[nh][j], acc[(mh) * MH + i][(nh) * 2 + j]);     \
    _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
        acc[(mh) * MH + i][(nh) * 2 + j] = CVX_P8S_MM(fah[i], fbh[nh][j], acc[(mh) * MH + i][(nh) * 2 + j]);     \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    CVX_P8_BARRIER();
#define CVX_P8S_SYNC()                                                                                           \
    CVX_P8_WAIT_DMA();                                                                                           \
    CVX_P8_BARRIER();                                                                                            \
    CVX_P8_WAIT_LDS();
    // Offsets switch to the NEXT tile at the top of a K-tile, where no fragment register is live: the A0 / W0 quarters
    // issued from K-tile nk-2 on and the W1 / A1 quarters issued from K-tile nk-1 on belong to it (nk is even then, so
    // nk-2 is a buffer-0 instance and nk-1 a buffer-1 instance).
#define CVX_P8S_KTILE(buf, t)                                                                                    \
    {                                                                                                            \
        if (has_next && (t) == nk - 2 + (buf)) set_offsets((buf), m0n, n0n);                                     \
        if (HAS_A2 && a2_ratio != 1.f && (t) == t_sw) {          /* the A2 pairs carry another pre-scale: bring the sums so far to it */ \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] *= a2_ratio;   \
        }                                                                                                        \
        CVX_P8S_READ_B(buf, 0) CVX_P8S_READ_A(buf, 0)                                                            \
        issue_W(1, (t) + 1);                                                                                     \
        CVX_P8S_SYNC() CVX_P8S_MFMA(0, 0)                                                                        \
        CVX_P8S_READ_B(buf, 1)                                                                                   \
        issue_A(1, (t) + 1);                                                                                     \
        CVX_P8S_SYNC() CVX_P8S_MFMA(0, 1)                                                                        \
        CVX_P8S_READ_A(buf, 1)                                                                                   \
        issue_A(0, (t) + 2);                                                                                     \
        CVX_P8S_SYNC() CVX_P8S_MFMA(1, 1)                                                                        \
        issue_W(0, (t) + 2);                                                                                     \
        CVX_P8S_SYNC() CVX_P8S_MFMA(1, 0)                                                                        \
    }
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        CVX_P8S_KTILE(0, t)
        CVX_P8S_KTILE(1, t + 1)
    }
    if (t < nk) CVX_P8S_KTILE(0, t)
#undef CVX_P8S_KTILE
#undef CVX_P8S_SYNC
#undef CVX_P8S_MFMA
#undef CVX_P8S_MM
#undef CVX_P8S_READ_A
#undef CVX_P8S_READ_B
    if (wr == 0) CVX_P8_BARRIER();                      // pairs with group 1's last barrier: both epilogues then run together
    if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the tail's dummy pieces before LDS is released
}

template <bool HAS_A2, int EPI, int MI = 8>
__global__ __launch_bounds__(512, 2) void gemm_f16x3_p8s_kernel(
    const cvx_gemm_args p, const PreSplitA A, const f16* __restrict__ W, float acc_scale, SplitOut so,
    int tiles_m, int tiles_n, int n_slots, int m_base)
{
    // (m_base: first row of this launch - a problem may run as whole rounds of 256-row tiles plus a tail launch of 192-row tiles)
    // PERSISTENT over output tiles: block b walks tile slots b, b + gridDim.x, ... (gridDim.x is a multiple of 8, so all of
    // them sit on the XCD that owns their row panels: slot s -> XCD s & 7, row panel (s & 7) + 8 * ((s >> 3) / tiles_n)).
    // The LDS-DMA stream never stops at a tile boundary: the tail of a tile fetches the first six quarters of the next one,
    // which therefore starts without a prologue, behind the epilogue of its predecessor.
    extern __shared__ __attribute__((aligned(16))) char smem_p8s[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    acc_scale = total_acc_scale(acc_scale, so);
    const float a2_ratio = (HAS_A2 && so.a2_scale && so.a_scale) ? *so.a2_scale / *so.a_scale : 1.f;

    auto tile_of_slot = [&](int s, int& m0, int& n0) {          // -> false for the padding slots of the XCD map
        const int xcd = s & 7, q = s >> 3;
        const int tm = xcd + 8 * (q / tiles_n), tn = q % tiles_n;
        m0 = m_base + tm * (32 * MI); n0 = tn * 256;
        return tm < tiles_m;
    };
    auto next_slot = [&](int s) {                                // next slot of this block that holds a real tile, or -1
        int m, n;
        for (s += (int)gridDim.x; s < n_slots; s += (int)gridDim.x)
            if (tile_of_slot(s, m, n)) return s;
        return -1;
    };
    int slot = (int)blockIdx.x, m0, n0;
    if (!tile_of_slot(slot, m0, n0)) {
        slot = next_slot(slot);
        if (slot < 0) return;
        tile_of_slot(slot, m0, n0);
    }
    bool first = true;
    while (true) {
        const int nslot = next_slot(slot);
        int m0n = -1, n0n = -1;
        if (nslot >= 0) tile_of_slot(nslot, m0n, n0n);

        f32x4 acc[MI][4];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row0 = m0 + wr * (16 * MI), col0 = n0 + wc * 64;
        bool v_block = false;
        if constexpr (EPI == EPI_QKV || EPI == EPI_QKV_RS) v_block = n0 >= p.rope_cols;          // block-uniform: this tile holds V columns
        constexpr bool PERM = epi_perm(EPI);
        if (v_block) {
            tile_mainloop<HAS_A2, false, PERM, MI>(p, A, W, smem_p8s, m0, n0, m0n, n0n, first, lane, wid, wr, wc, acc, a2_ratio);
            if (!(so.dbg & 1)) epilogue_vt<MI, false, EPI == EPI_QKV_RS, PERM>(p, acc, row0, col0, lane, so, acc_scale);
        } else {
            tile_mainloop<HAS_A2, true, PERM, MI>(p, A, W, smem_p8s, m0, n0, m0n, n0n, first, lane, wid, wr, wc, acc, a2_ratio);
            if (!(so.dbg & 1)) epilogue_rows<EPI, MI>(p, acc, row0, col0, lane, so, acc_scale);  // (dbg bit 0: main loop only, timing)
        }
        if (nslot < 0) break;
        // the epilogue's own loads / stores share the vmcnt counter with the DMA: start the next tile from a clean count
        // (its first quarters landed long ago: the last phases of this tile already waited for them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        slot = nslot; m0 = m0n; n0 = n0n;
        first = false;
    }
}

}  // namespace

namespace cvxg {

int classify_epilogue(const cvx_gemm_args& a, const SplitOut& so)
{
    // the specialised epilogues only have the 16-byte vector path: whole 64-column wave tiles, aligned pointers and strides
    const bool vec = (a.N % 64 == 0) &&
                     (!so.write_f32 || (((uintptr_t)a.C & 15) == 0 && (a.ldc & 3) == 0)) &&
                     (!a.residual || (((uintptr_t)a.residual & 15) == 0 && (a.ldr & 3) == 0)) &&
                     (!so.hi || ((((uintptr_t)so.hi | (uintptr_t)so.lo) & 7) == 0 && (so.ldc_h & 3) == 0));
    if (!vec) return EPI_GENERIC;
    const bool rope = a.rope_cos != nullptr;
    if (rope && so.vt_hi && so.hi && !so.write_f32 && !a.bias && !a.residual && a.act == CVX_ACT_NONE) return EPI_QKV;
    if (rope || so.vt_hi) return EPI_GENERIC;
    if (a.residual && so.write_f32 && a.act == CVX_ACT_NONE) return EPI_RES;
    if (a.bias && a.act == CVX_ACT_GELU && !a.residual && so.hi && !so.write_f32) return EPI_GELU_SPLIT;
    if (a.act == CVX_ACT_NONE && !a.residual && so.write_f32 && !so.hi) return EPI_BIAS;
    return EPI_GENERIC;
}

bool launch_gemm_f16x3_p8s(const cvx_gemm_args& a, const PreSplitA& A, const f16* w_il, float acc_scale, const SplitOut& so,
                           hipStream_t st, int n_cu)
{
    if (a.K % 32 != 0 || (A.hi2 && a.K1 % 32 != 0) || a.N % 64 != 0) return false;
    if ((int64_t)a.M * A.ld * 2 >= (int64_t)1 << 32 || (A.hi2 && (int64_t)a.M * A.ld2 * 2 >= (int64_t)1 << 32) ||
        (int64_t)a.N * a.ldw * 2 >= (int64_t)1 << 32) return false;
    // 16-byte vector epilogue only: aligned pointers and strides, bias / RoPE tables included
    const bool vec = (!so.write_f32 || (((uintptr_t)a.C & 15) == 0 && (a.ldc & 3) == 0)) &&
                     (!a.residual || (((uintptr_t)a.residual & 15) == 0 && (a.ldr & 3) == 0)) &&
                     (!so.hi || ((((uintptr_t)so.hi | (uintptr_t)so.lo) & 7) == 0 && (so.ldc_h & 3) == 0)) &&
                     (!a.bias || ((uintptr_t)a.bias & 15) == 0) &&
                     (!a.rope_cos || ((((uintptr_t)a.rope_cos | (uintptr_t)a.rope_sin) & 15) == 0 && a.rope_cols % 256 == 0));
    if (!vec) return false;
    if (so.vt_hi && !(a.rope_cos && so.hi && !so.write_f32 && !a.residual && a.act == CVX_ACT_NONE)) return false;   // V^T only in QKV form
    const int tn = (a.N + 255) / 256;
    // n_cu: CUs the launch context's stream owns (cvx_ctx.n_cus; default: the device's)
    // tile height: 256 rows, or 192 where rounds x height comes out smaller (a launch runs in rounds of one tile per CU and a tile's
    // time goes with its height).  so.dbg bits 16 / 32 pin 192 / 256 (CVX_GEMM_FLAG_TILE192 / _TILE256: A/B, bit-identity tests).
    const int cus8 = (n_cu / 8) * 8 > 0 ? (n_cu / 8) * 8 : 8;
    // (a 192-row tile against a 256-row one, per round, measured at 18,596 rows in round 6 - profiles/r06_gemm_tile_forms.txt: 0.71-0.75 at
    //  K = 1024 (to_qkv 63 vs 89 us, ff1 61 vs 83, to_out 61 vs 81), 0.79 at K = 2048 / 4096 (the W stream per flop is 4/3): priced at 0.75 /
    //  0.8.  Round 5 priced every shape at 0.8 from one N = 1024 measurement and kept to_qkv of a 9,298-frame launch on 4 rounds of 256-row
    //  tiles: 356 us against 315 for 5 rounds of 192.)
    const long u192 = a.K <= 1024 ? 192 : 205;
    auto cost = [&](int h) { const long t = (long)((a.M + h - 1) / h) * tn; return ((t + cus8 - 1) / cus8) * (long)(h == 192 ? u192 : 256); };
    // MIXED (round 6): whole rounds of 256-row tiles, then ONE launch of 192-row tiles over the rows that are left - a ragged packed
    // launch (9,298 utterance rows x 2 CFG branches = 18,596 rows, N = 4096) runs 4 full rounds + one short round instead of 5 rounds.
    // r256 = row tiles of the first launch: the largest count whose tiles are whole rounds on this stream's CUs (and whole groups of 8
    // row panels, the XCD map).  The launch boundary is NOT free: no prefetch across it, a drain and a ramp - ff1 at 18,596 rows
    // measures 413.3 us mixed against 415.2 (256-row) where 4 x 83 + 61 = 393 was the model: ~20 us = 60 units.  So the form only wins
    // where a whole round is saved at small round counts; it is bit-identical and stays selectable (CVX_GEMM_FLAG_TILE_MIXED).
    long r256 = 0, cost_mixed = -1;
    {
        long step = cus8;                                           // smallest r with r * tn % cus8 == 0 ...
        for (long r = 8; r <= cus8; r += 8) if ((r * tn) % cus8 == 0) { step = r; break; }
        r256 = ((long)(a.M / 256) / step) * step;
        const long rest = a.M - r256 * 256;
        if (r256 > 0 && rest > 0) {
            const long t = ((rest + 191) / 192) * tn;
            cost_mixed = (r256 * tn / cus8) * 256 + ((t + cus8 - 1) / cus8) * u192 + 60;
        }
    }
    const long c256 = cost(256), c192 = cost(192);
    const bool pinned = (so.dbg & (16 | 32 | 64)) != 0;
    const bool mixed = (so.dbg & 64) ? cost_mixed >= 0 : (!pinned && cost_mixed >= 0 && cost_mixed < c256 && cost_mixed < c192);
    const bool h192 = (so.dbg & 16) ? true : (so.dbg & 32) ? false : c192 < c256;
    // persistent grid: one block per CU (136 KiB of LDS each), a multiple of 8; a next tile needs an even number of K-tiles
    // (the two LDS buffers alternate across the tile boundary), otherwise every tile gets its own block
    struct Part { int th, tm, n_slots, g, m_base; };
    auto part = [&](int th, long rows, int m_base) {
        Part q;
        q.th = th; q.m_base = m_base;
        q.tm = (int)((rows + th - 1) / th);
        q.n_slots = ((q.tm + 7) / 8) * 8 * tn;                     // XCD map: an XCD owns whole row panels (padding slots are skipped)
        q.g = q.n_slots;
        if ((a.K / 32) % 2 == 0 && !(so.dbg & 8)) q.g = q.n_slots < cus8 ? q.n_slots : cus8;
        return q;
    };
    Part parts[2];
    int n_parts = 1;
    if (mixed) { parts[0] = part(256, r256 * 256, 0); parts[1] = part(192, a.M - r256 * 256, (int)(r256 * 256)); n_parts = 2; }
    else parts[0] = part(h192 ? 192 : 256, a.M, 0);
    int epi = classify_epilogue(a, so);
    if (epi == EPI_QKV && a.bias) epi = EPI_GENERIC;
    if (so.tw_gamma || so.rowsq || so.row_scale || so.res_hi) {         // deferred norm: its own instances, nothing else carries it
        const bool prod = so.hi && a.act == CVX_ACT_NONE && !a.rope_cos && !so.vt_hi && !so.row_scale && !(a.residual && so.res_hi) &&
                          (!so.res_hi || (so.res_lo && (((uintptr_t)so.res_hi | (uintptr_t)so.res_lo) & 7) == 0 && (so.res_ld & 3) == 0));
        const bool cons = so.row_scale && !so.tw_gamma && !so.rowsq && !so.res_hi && so.hi && !so.write_f32 && !a.residual;
        if (prod && (a.residual || so.res_hi) && !A.hi2) epi = EPI_RES_TW;
        else if (prod && !a.residual && !so.res_hi && A.hi2) epi = EPI_BIAS_TW;
        else if (cons && a.bias && a.act == CVX_ACT_GELU && !a.rope_cos && !so.vt_hi && !A.hi2) epi = EPI_GELU_RS;
        else if (cons && a.act == CVX_ACT_NONE && a.rope_cos && so.vt_hi && !A.hi2) epi = EPI_QKV_RS;
        else return false;
    }
    // (the permuted instances move split pairs 16 bytes at a time: 8 consecutive columns per lane, see perm32; anything less aligned
    //  goes to the 32x32x16 kernel)
    if (epi_perm(epi) && (((((uintptr_t)so.hi | (uintptr_t)so.lo | (uintptr_t)so.res_hi | (uintptr_t)so.res_lo) & 15) != 0) || (so.ldc_h & 7) != 0 ||
                          (so.res_hi && (so.res_ld & 7) != 0))) return false;
    if (epi == EPI_GENERIC && so.vt_hi) return false;
    if (so.a2_scale && !(epi == EPI_BIAS_TW && so.a_scale)) return false;
#define CVX_P8S_LAUNCH_MI(A2, E, MI_, Q)                                                                                \
    do {                                                                                                                \
        cvx_allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_f16x3_p8s_kernel<A2, E, MI_>), LDS_B);                \
        hipLaunchKernelGGL((gemm_f16x3_p8s_kernel<A2, E, MI_>), dim3((unsigned)(Q).g), dim3(512), LDS_B, st, a, A, w_il, acc_scale, so,    \
                           (Q).tm, tn, (Q).n_slots, (Q).m_base);                                                        \
    } while (0)
#define CVX_P8S_LAUNCH(A2, E)                                                                                           \
    do {                                                                                                                \
        for (int pi = 0; pi < n_parts; ++pi) {                                                                          \
            if (parts[pi].th == 192) CVX_P8S_LAUNCH_MI(A2, E, 6, parts[pi]); else CVX_P8S_LAUNCH_MI(A2, E, 8, parts[pi]); \
        }                                                                                                               \
    } while (0)
    if (A.hi2) {
        if (epi == EPI_BIAS) CVX_P8S_LAUNCH(true, EPI_BIAS); else if (epi == EPI_BIAS_TW) CVX_P8S_LAUNCH(true, EPI_BIAS_TW); else CVX_P8S_LAUNCH(true, EPI_GENERIC);
    } else {
        switch (epi) {
            case EPI_RES_TW: CVX_P8S_LAUNCH(false, EPI_RES_TW); break;
            case EPI_GELU_RS: CVX_P8S_LAUNCH(false, EPI_GELU_RS); break;
            case EPI_QKV_RS: CVX_P8S_LAUNCH(false, EPI_QKV_RS); break;
            case EPI_QKV: CVX_P8S_LAUNCH(false, EPI_QKV); break;
            case EPI_RES: CVX_P8S_LAUNCH(false, EPI_RES); break;
            case EPI_GELU_SPLIT: CVX_P8S_LAUNCH(false, EPI_GELU_SPLIT); break;
            case EPI_BIAS: CVX_P8S_LAUNCH(false, EPI_BIAS); break;
            default: CVX_P8S_LAUNCH(false, EPI_GENERIC); break;
        }
    }
#undef CVX_P8S_LAUNCH
#undef CVX_P8S_LAUNCH_MI
    return true;
}

}  // namespace cvxg
