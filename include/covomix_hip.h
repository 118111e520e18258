/* covomix_hip.h - C ABI of libcovomix_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the build.  The reference has no native
 * code on this path - its contract is the PyTorch semantics of the modules cited
 * below (paths relative to /root/reference) - so every entry point here names the
 * reference computation it replaces.  Conventions:
 *   - plain C symbols; device pointers + explicit shapes/strides; a hipStream_t
 *     (passed as void*) on which the work is enqueued; no hidden allocation, no
 *     global mutable state besides a thread-local last-error string;
 *   - return 0 on success, CVX_EINVAL on a shape/alignment error, CVX_EHIP when
 *     the HIP runtime reported a launch error (message via cvx_last_error_string);
 *   - all floating-point data is fp32 row-major; token ids are int64.
 * Weights are borrowed (never copied or freed by the library).
 */
#ifndef COVOMIX_HIP_H
#define COVOMIX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVX_OK      0
#define CVX_EINVAL (-22)
#define CVX_EHIP   (-5)

#define CVX_ACT_NONE 0
#define CVX_ACT_GELU 1   /* exact erf GELU  (nn.GELU default, acoustic.py:159,244) */
#define CVX_ACT_SILU 2   /* SiLU            (time MLP, acoustic.py:364)            */
#define CVX_ACT_TANH 3

/* LAUNCH CONTEXT (version 107): what every entry point takes where it used to take a bare hipStream_t.  A plain host struct the
 * CALLER owns and may keep for the life of the stream; the library only reads it during the call and keeps NOTHING about a stream
 * between calls (no table, no lock: re-entrant per context; SURVEY.md section 8(b) "no global mutable state").  NULL = the null
 * stream, no flag, the whole device. */
typedef struct cvx_ctx {
    void*     stream;     /* hipStream_t the call launches on */
    uint32_t* sat_flag;   /* the sticky saturation flag of this stream: one 4-byte aligned uint32 of DEVICE memory the caller zeroed and
                           * keeps alive (below), or NULL */
    int32_t   n_cus;      /* compute units the stream owns (a CU-masked stream, cvx_stream_create_cu_mask); 0 = the device's: what the
                           * persistent grids and the large / medium GEMM choice are sized from */
    int32_t   flags;      /* CVX_CTX_* */
} cvx_ctx;
#define CVX_CTX_NO_SATURATION_FLAG 1   /* run the calls that write split (fp16 hi, lo) pairs WITHOUT the saturation bookkeeping: without
                                        * this bit such a call refuses a context whose sat_flag is NULL (CVX_EINVAL) - a C caller cannot
                                        * lose the safety net by forgetting it */
typedef const cvx_ctx* cvx_stream_t;

/* ABI version of the library: CVX_ABI_VERSION of the header it was built from.  The argument structs carry no size field, so a
 * caller built against another header version must not call in: compare once after loading (the ctypes binding does,
 * covomix_amd/_lib.py).  104: round 4 (cvx_gemm_f16x3_norm, caller-owned saturation flags, cvx_t2s_decoder.cfg_scale,
 * cvx_t2s_decode_xcd, CVX_GEMM_FLAG_MEDIUM / _NO_MEDIUM).  105: round 4, the deferred AdaptiveRMSNorm - cvx_gemm_split_io grew
 * c_gamma_dev ... a2_scale_dev at its end; cvx_rownorm_scale_f32, cvx_split_f16_colscale_il.  106: round 5 - CU-partitioned
 * streams (cvx_stream_create_cu_mask / _destroy / cvx_stream_set_cus / cvx_stream_cus); the library reads no environment
 * variable in any build; REMOVED (measured slower, numbers in
 * HISTORY.md): cvx_t2s_decode_persistent, cvx_t2s_decode_xcd, cvx_embed_conv31_f32, CVX_GEMM_FLAG_TWO_STAGE / _MFMA32 (the superseded
 * large-problem GEMM forms: shapes the eight-phase kernel cannot take run on the 128 x 128 kernel).  107: round 6 - text2semantic
 * decode of up to 64 slots at independent positions with an on-device dialogue queue (continuous batching): cvx_t2s_decoder grew
 * uniform_steps / queue / dialogues / start, slot records are int32[8], uniforms and tokens are indexed by dialogue; every entry point
 * takes a LAUNCH CONTEXT (cvx_ctx: stream + caller-owned saturation flag + CU count) instead of a bare stream - REMOVED: the library's
 * (device, stream) table and cvx_saturation_flag_bind / cvx_stream_set_cus / cvx_stream_cus. */
#define CVX_ABI_VERSION 107
int         cvx_version(void);
const char* cvx_last_error_string(void);

/* ------------------------------------------------------------------------
 * Sticky saturation flag: CALLER-OWNED, one uint32 of device memory per stream, carried by the launch context (cvx_ctx.sat_flag;
 * versions 104-106 kept a (device, stream) -> flag table inside the library, until 102 the library allocated one per device).
 * The split-precision path stores activations as (fp16 hi, fp16 lo) pairs times a power-of-two pre-scale chosen so that
 * the tensor's expected RMS sits at 2^4 (transformer: a weights-only gain model) or its measured max at 2^10 (vocoder):
 * 2^12 resp. 2^6 of headroom before `hi` saturates at 65504.  A checkpoint with one outlier row / channel can leave that
 * window; the pair is then CLAMPED and no longer represents the fp32 value.  Every kernel that writes split pairs (GEMM
 * epilogues incl. the transposed V^T store, AdaRMSNorm, attention, cvx_split_f16*, the HiFi-GAN convolutions and layout
 * converter) ORs bit 0 into the flag OF THE CONTEXT IT IS LAUNCHED WITH when a value it stored exceeded 65504 in magnitude
 * (the attention kernel also when a softmax normaliser is not a positive finite number).  The flag is never cleared by a kernel:
 * reset it, run any number of calls, query once.  The host path (CoVoMixModel.synthesis_sample, Generator.__call__) does exactly
 * that and re-runs the call on the exact-fp32 kernels when the flag is set - a saturated result is never returned silently.
 * Several contexts may share one flag (the side stream of a two-stream schedule, a graph-capture stream).  A call that writes
 * pairs with a context that has no flag returns CVX_EINVAL unless the context says CVX_CTX_NO_SATURATION_FLAG.
 *   cvx_saturation_flag_reset: enqueue a clear of the context's flag on its stream (no host synchronisation); CVX_EINVAL without one.
 *   cvx_saturation_flag_query: copy the flag to *host_out behind everything enqueued on the stream (SYNCHRONISES it), then
 *                              optionally enqueue a clear. */
int cvx_saturation_flag_reset(cvx_stream_t ctx);
int cvx_saturation_flag_query(uint32_t* host_out, int32_t reset, cvx_stream_t ctx);

/* ------------------------------------------------------------------------
 * CU-partitioned streams (round 5): the text2semantic decode of the NEXT batch of dialogues (a latency chain of 34 dependent
 * launches per token that needs a handful of CUs; reference loop dialogue_generation.py:272-329, decode
 * covomix/covomix_model/text2semantic.py:749-848) can run UNDER the acoustic solve of the current one.  The large-problem GEMM is a
 * persistent kernel that owns every CU it gets for a whole launch, so a plain side stream would only be served at launch
 * boundaries: the two stages run on streams restricted to DISJOINT CU sets instead.
 *   cvx_stream_create_cu_mask: hipExtStreamCreateWithCUMask -> *out_stream (a hipStream_t the caller owns; put it into a cvx_ctx
 *                              together with n_cus = the number of mask bits).  Bit k of the mask names CU (k / 8) of XCD (k % 8);
 *                              inside an XCD consecutive indices go round the four shader engines (measured on MI355X,
 *                              tools/archive/cu_mask_probe.hip).  One-block-per-CU kernels are only co-resident when every shader engine keeps
 *                              the same number of CUs, i.e. when the CUs per XCD are a multiple of 4 (tools/archive/cu_mask_probe2.hip: 30 CUs
 *                              per XCD -> 12 of 240 blocks wait for a second round; 28 -> none).
 *   cvx_stream_destroy:        hipStreamDestroy of such a stream. */
/* Diagnostics (bench.py): stamps_dev is 2048 x 2 uint64 of ZEROED device memory; slot = xcd * 256 + HW_ID[15:8] (the CU's id bits) receives
 * {that CU's shader-clock cycle counter, the 100 MHz real-time counter} (slots of CUs no block landed on stay zero).  Two calls around a
 * busy region, paired slot by slot (the cycle counters of different CUs are not comparable) -> the shader clock the region ran at, per CU. */
int cvx_clock_stamps(uint64_t* stamps_dev, cvx_stream_t ctx);
int cvx_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, void** out_stream);
int cvx_stream_destroy(void* stream);

/* ------------------------------------------------------------------------
 * C[M,N] = epilogue( [A | A2][M,K] * W[N,K]^T )      fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *
 * Replaces every nn.Linear on the path: to_embed (acoustic.py:503-505), to_qkv /
 * to_out (:225-237), FeedForward (:241-246), skip combiner on cat(x, skip)
 * (:306-310, expressed as a K-split over two inputs: columns [0,K1) come from A,
 * [K1,K) from A2), to_pred (:516), the time MLP (:361-365) and the to_gamma /
 * to_beta projections (:200).
 * epilogue order:  v = acc + bias[n];  v = act(v);
 *                  RoPE (half-split, acoustic.py:132-137) on columns [0,rope_cols):
 *                       64-wide heads, position = row % rope_T, tables cos/sin[rope_T][32];
 *                  v += residual[m,n];  C[m,n] = v.
 * Requirements: K % 4 == 0, lda/lda2/ldw % 4 == 0, 16-byte aligned A/A2/W,
 *               K1 % 32 == 0 when A2 != NULL, rope_cols % 64 == 0.
 */
typedef struct {
    const float* A;   int64_t lda;
    const float* A2;  int64_t lda2;  int32_t K1;
    const float* W;   int64_t ldw;
    float*       C;   int64_t ldc;
    const float* bias;
    const float* residual; int64_t ldr;
    int32_t M, N, K;
    int32_t act;
    const float* rope_cos; const float* rope_sin; int32_t rope_T; int32_t rope_cols;
} cvx_gemm_args;
int cvx_gemm_bias_act_f32(const cvx_gemm_args* a, cvx_stream_t s);

/* Split-precision variant of the same contract (same nn.Linear call sites): every fp32 operand is an
 * (fp16 hi, fp16 lo) pair, hi = fp16(x), lo = fp16(x - hi), and  a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi
 * on v_mfma_f32_32x32x16_f16 with fp32 accumulation (~2^-22 relative operand error, inputs saturated to
 * +-65504).  W_hi / W_lo are [N, a->ldw] fp16 matrices produced once by cvx_split_f16 from the fp32 weight
 * (a->W is only validated, not read).  Extra requirements: K % 32 == 0, ldw % 8 == 0.
 * cvx_split_f16 splits w*scale (scale = a power of two that lifts small weights out of the fp16 subnormal
 * range); cvx_gemm_f16x3 multiplies the accumulators by acc_scale = 1/scale (exact) before the epilogue.
 * `io` (may be NULL) lets activations stay in split form between kernels: A_hi/A_lo (and A2_*) give the A operand
 * already split (then a->A / a->A2 are only validated and every tile arrives by LDS-DMA); C_hi/C_lo receive a split
 * copy of the output for the next GEMM, and write_f32 == 0 drops the fp32 store of C altogether.
 *
 * Interleaved pairs.  Wherever this header takes an (fp16 hi, fp16 lo) pair of a row-major activation - A_*, A2_*,
 * C_* here, y_hi/y_lo of cvx_adarmsnorm_f32, out_hi/out_lo of cvx_attention_f32 / cvx_attention_f16x3, hi/lo of
 * cvx_split_f16 - passing lo == hi + 32 (halves) selects ONE buffer of row length 2*cols in which every block of 32
 * values is stored as [hi 32 | lo 32]: flat offset o of the two-tensor form lives at ((o >> 5) << 6) | (o & 31) for
 * hi and 32 halves later for lo; the leading dimension (lda_h, ldc_h) is then the row length of that buffer in
 * halves (>= 2*cols).  Values are bit-identical to the two-tensor form; the layout makes one K-step of a row one
 * 128-byte cache line.  cvx_gemm_f16x3 reads interleaved A/A2 only in the large-problem kernel (M >= 2048,
 * N >= 512) and only together with w_interleaved; A and A2 must then both be interleaved.  lo == NULL always means
 * "hi only" (single-term fp16). */
typedef struct {
    const uint16_t* A_hi;  const uint16_t* A_lo;  int64_t lda_h;
    const uint16_t* A2_hi; const uint16_t* A2_lo; int64_t lda2_h;
    uint16_t* C_hi; uint16_t* C_lo; int64_t ldc_h;
    int32_t write_f32;
    /* QKV mode (optional; needs the RoPE arguments, N = 3*H*64, write_f32 = 0; T % 4 != 0 is accepted but stores the v columns 2 bytes at a time): the v columns are not
     * written to C_hi/C_lo but TRANSPOSED per (sequence, head) to Vt_*[((b*H + h)*64 + d) * vt_ld + slot(t)] (slot swaps bits 2 and 3
     * of t: four-frame groups in the order 0, 2, 1, 3 inside every 16 frames), the layout
     * cvx_attention_f16x3 reads its V^T tiles from. */
    uint16_t* Vt_hi; uint16_t* Vt_lo; int64_t vt_ld;
    /* Optional caller-owned scratch (fp32, at least 4*M*N floats) for small problems: with few output tiles and a long
     * K the K range is cut into up to 4 slices computed by different blocks, the partial sums go to `workspace` and a
     * second kernel adds them in a fixed order (deterministic) and applies the epilogue.  NULL = never split K.
     * Not available together with the RoPE / QKV-transpose epilogues. */
    float* workspace; int64_t workspace_floats;
    /* Non-zero: the split weight is ONE interleaved matrix [N][K/32][hi 32 | lo 32] (W_lo == W_hi + 32 halves,
     * a->ldw >= 2K halves), so that a K-step of a row is a whole 128-byte cache line.  Large-problem kernel only
     * (pre-split A, M >= 2048, N >= 512). */
    int32_t w_interleaved;
    /* Kernel selection for A/B measurements (0 = default): CVX_GEMM_FLAG_* below.  Results of the kernels agree to fp32
     * rounding.  (Bits 8.. are timing experiments that exist only in -DCVX_DEV_FLAGS builds of the library; the shipped
     * library ignores them.) */
    int32_t flags;
    /* Activation scales: DEVICE pointers to one float each (NULL = 1.0), powers of two.  a_scale_dev = the factor the
     * producer of A_hi/A_lo (and of A2_*: both operands must share it) multiplied the values by before splitting - the
     * accumulators are divided by it (exact); c_scale_dev / vt_scale_dev = the factor applied to the values written to
     * C_hi/C_lo and Vt_hi/Vt_lo (the fp32 store of C is never scaled).  They keep split pairs inside fp16's
     * full-precision window (|x| in [2^-3, 2^16)) whatever the magnitude of the tensor; living in device memory they can
     * be computed by an earlier kernel without a host round trip (and replayed from a captured graph). */
    const float* a_scale_dev; const float* c_scale_dev; const float* vt_scale_dev;
    /* Deferred AdaptiveRMSNorm (version 105; all NULL = off).  The norm between a to_out / ff2 / skip-combiner product and the
     * to_qkv / ff1 product that follows it (acoustic.py:306-318, :198-204) is one gamma / beta row for every frame of an
     * evaluation, so  norm(x) . W^T = (sqrt(D) / ||x_row||) * ((x * gamma) . W^T) + beta . W^T :
     *   the PRODUCER call passes c_gamma_dev ([N] floats: C_hi/C_lo then hold C[m,n] * gamma[n] * *c_scale_dev - the fp32 store of C
     *   is unchanged) and c_rowsq ([M][c_rowsq_ld >= N/64] floats: every 64-column slice of a row leaves its sum of squares there, in
     *   a fixed order); cvx_rownorm_scale_f32 turns those into one factor per row; the CONSUMER call passes them as a_row_scale_dev
     *   ([M] floats, multiplied into row m of the accumulators before bias / activation / RoPE) and beta . W^T as (part of) its bias.
     * No normalised tensor exists in HBM and the norm kernel does not run.  Only the large-problem kernel takes these (pre-split
     * interleaved A and W, M >= 2048, N >= 512, N % 64 == 0, 16-byte aligned epilogue operands) in four forms - producer:
     * residual + fp32 store (to_out, ff2) or A2 + bias + fp32 store (skip combiner); consumer: bias + GELU + split store (ff1) or
     * the QKV mode with an optional bias - and every other combination is refused (CVX_ERR_INVALID). */
    const float* c_gamma_dev; float* c_rowsq; int64_t c_rowsq_ld; const float* a_row_scale_dev;
    /* The residual as a split pair (producer forms with a residual only; a->residual must be NULL): residual[m,n] =
     * (R_hi + R_lo)[m,n] / *r_scale_dev (R_lo == R_hi + 32: interleaved, ldr_h >= 2N).  With write_f32 == 0 the residual stream
     * of the transformer lives in HBM as pairs only - the to_out / ff2 epilogue then moves the bytes the fp32 form moved (read
     * 4 B, write 4 B per element) and the norm kernel's 8 B per element are gone.  c_gamma_dev may be NULL (gamma folded into the
     * consumer's weights instead: cvx_split_f16_colscale_il).  R may alias C_hi / C_lo (same element, same lane). */
    const uint16_t* R_hi; const uint16_t* R_lo; int64_t ldr_h; const float* r_scale_dev;
    /* A | A2 pairs with DIFFERENT pre-scales (skip-combiner producer form only): A2_hi/A2_lo hold a2 * *a2_scale_dev while A holds
     * a * *a_scale_dev (NULL: both operands share a_scale_dev).  Every stage of the pair-only residual stream carries its own
     * power-of-two pre-scale, so a skip saved at the input of layer i and the stream at layer depth-1-i need not share one. */
    const float* a2_scale_dev;
} cvx_gemm_split_io;
/* (bits 1 and 2 selected the two superseded large-problem kernels until version 105: ignored now) */
#define CVX_GEMM_FLAG_MEDIUM 8      /* interleaved operands, 2048 rows and more: take the medium-problem kernel (128 x 128 tiles) whatever the tile count */
#define CVX_GEMM_FLAG_NO_MEDIUM 16  /* ... never take it there (the large-problem kernel's rounds of 256 x 256 tiles): A/B measurements */
#define CVX_GEMM_FLAG_ONE_TILE 4    /* eight-phase 16x16x32 kernel: one output tile per block instead of persistent blocks (bit-identical results) */
#define CVX_GEMM_FLAG_TILE192 64    /* large-problem kernel: 192-row tiles whatever the tile count (default: 192 where rounds x height come out smaller than with 256) */
#define CVX_GEMM_FLAG_TILE256 128   /* ... 256-row tiles always (bit-identical results either way: A/B measurements) */
#define CVX_GEMM_FLAG_TILE_MIXED 32 /* ... whole rounds of 256-row tiles + one launch of 192-row tiles over the rest, wherever such a split exists
                                     * (default: where it is the cheapest of the three; bit-identical results) */
int cvx_split_f16(const float* w, uint16_t* hi, uint16_t* lo, int64_t n, float scale, cvx_stream_t s);
/* n_sets interleaved split copies of ONE weight matrix with its COLUMNS scaled: out[s][n][il(k)] = split( W[n,k] * colscale[s*cs_ld + k] *
 * set_scale_dev[s*ss_ld] * scale ), each [N][K/32][hi 32 | lo 32] (the w_interleaved layout, row length 2K halves), K % 32 == 0.
 * Deferred AdaptiveRMSNorm with gamma on the weight side: set s = evaluation time s of a solve, colscale = that time's gamma row,
 * set_scale_dev = a power of two that keeps |gamma| <= 1 (the consumer divides it out through a_scale_dev).  W is read once. */
int cvx_split_f16_colscale_il(const float* W, int64_t ldw, int32_t N, int32_t K, const float* colscale, int64_t cs_ld,
                              const float* set_scale_dev, int64_t ss_ld, int32_t n_sets, float scale, uint16_t* out, cvx_stream_t s);
/* the same with an additional DEVICE-resident factor (the pair holds w * scale * *scale_dev; scale_dev may be NULL) */
int cvx_split_f16_dev(const float* w, uint16_t* hi, uint16_t* lo, int64_t n, float scale, const float* scale_dev, cvx_stream_t s);
int cvx_gemm_f16x3(const cvx_gemm_args* a, const uint16_t* W_hi, const uint16_t* W_lo, float acc_scale,
                   const cvx_gemm_split_io* io, cvx_stream_t s);
/* cvx_gemm_f16x3 followed by the AdaptiveRMSNorm / RMSNorm of its fp32 output, as ONE call (reference acoustic.py:306-318: every
 * to_out, ff2 and skip-combiner product of the transformer is followed by the next norm; :198-204, :175):
 *     C = epilogue([A | A2] . W^T)  (fp32, written: io->write_f32 must not be 0, ldc == N, act == NONE);
 *     Y = split( C[r,:] / max(||C[r,:]||_2, eps) * scale * gamma + beta ) * *y_scale_dev     (one gamma / beta row for all rows)
 * Problems that take the split-K path (fewer than 2048 rows, N <= 1024, N % 256 == 0) run the norm inside the split-K reduction
 * (the row is in registers there: no extra launch, no re-read of C); every other problem runs cvx_adarmsnorm_scaled_f32 behind
 * the product.  Both give the same bits. */
typedef struct {
    const float* gamma; const float* beta;          /* [N]; beta may be NULL (RMSNorm) */
    uint16_t* Y_hi; uint16_t* Y_lo; int64_t ldy_h;   /* split pair of the normalised rows, row stride in halves (Y_lo == Y_hi + 32,
                                                      * ldy_h == 2N: interleaved [hi 32 | lo 32]; otherwise ldy_h == N) */
    const float* y_scale_dev;                       /* power-of-two pre-scale of Y in DEVICE memory, NULL = 1 */
    float scale; float eps;                         /* sqrt(N) and 1e-12 in the reference */
} cvx_gemm_norm;
int cvx_gemm_f16x3_norm(const cvx_gemm_args* a, const uint16_t* W_hi, const uint16_t* W_lo, float acc_scale,
                        const cvx_gemm_split_io* io, const cvx_gemm_norm* norm, cvx_stream_t s);
/* Size (in floats) of cvx_gemm_split_io.workspace that lets a problem of this shape split K (0: this shape never does).
 * K1 = the A | A2 boundary of a K-split operand, 0 without A2. */
int64_t cvx_gemm_f16x3_workspace_floats(int32_t M, int32_t N, int32_t K, int32_t K1);

/* y[r,:] = x[r,:] / max(||x[r,:]||_2, eps) * scale * gamma[g,:] + beta[g,:],  g = r / rows_per_group
 * AdaptiveRMSNorm.forward (acoustic.py:198-204) with gamma/beta = the already projected
 * to_gamma/to_beta(time_emb) rows; RMSNorm.forward (:175) when beta == NULL and one group.
 * D % 4 == 0. */
int cvx_adarmsnorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                       uint16_t* y_hi, uint16_t* y_lo,   /* optional fp16 (hi, lo) split copy of y; y may then be NULL */
                       int64_t rows, int32_t D, int64_t rows_per_group, float scale, float eps,
                       cvx_stream_t s);
/* the same; the split copy (y_hi, y_lo) holds y * *split_scale_dev (a power of two in DEVICE memory, NULL = 1: the
 * activation pre-scale of cvx_gemm_split_io.a_scale_dev).  The fp32 output y is never scaled. */
/* out[r] = scale / max(sqrt(sum_{j < parts} rowsq[r * ld + j]), eps): the per-row factor of a deferred norm from the partial sums a
 * producer GEMM left (cvx_gemm_split_io.c_rowsq; F.normalize's eps clamp, acoustic.py:198-204).  parts <= 64, summed in index order. */
int cvx_rownorm_scale_f32(const float* rowsq, int64_t rows, int32_t parts, int64_t ld, float scale, float eps, float* out, cvx_stream_t s);
int cvx_adarmsnorm_scaled_f32(const float* x, const float* gamma, const float* beta, float* y,
                              uint16_t* y_hi, uint16_t* y_lo,
                              int64_t rows, int32_t D, int64_t rows_per_group, float scale, float eps,
                              const float* split_scale_dev, cvx_stream_t s);

/* out[b,t,h*64+d] = softmax_j( q[b,h,t,:] . k[b,h,j,:] * scale ) @ v[b,h,j,d]
 * Attend.forward non-flash branch (attend.py:108-126) without materialising the T x T
 * scores.  qkv is the to_qkv output [Bt, T, 3*H*64] (q | k | v, heads contiguous
 * 64-blocks, acoustic.py:227-229) with RoPE already applied to q and k (GEMM epilogue).
 * Head dim is fixed at 64 (every shipped config, running_command/Acous_*.sh). */
int cvx_attention_f32(const float* qkv, float* out,
                      uint16_t* out_hi, uint16_t* out_lo,  /* optional fp16 (hi, lo) split copy; out may then be NULL */
                      int32_t Bt, int32_t T, int32_t H, float scale, cvx_stream_t s);

/* ------------------------------------------------------------------------
 * RAGGED BATCHES (utterances of different length in one launch; the reference runs them one at a time,
 * monologue_generation.py:259-304, and its network has no key-padding mask, acoustic.py:313, so an utterance must
 * never see another one).  The n sequences are PACKED: sequence i owns rows [cu_seqlens[i], cu_seqlens[i+1]) of every
 * [M, width] tensor (M = cu_seqlens[n]; cu_seqlens: n+1 int32 in DEVICE memory; max_T = the longest sequence, which
 * sizes the grid).  Everything row-wise (GEMMs, norms, CFG combine, embedding gather) needs nothing; the three
 * operators that look along the time axis take the table:
 *   - attention: keys restricted to the own sequence (the *_varlen entry points below);
 *   - RoPE: run the to_qkv GEMM with rope_T = M and per-ROW tables cos/sin [M][32] (row r: its position inside its
 *     sequence), which also makes its V^T output ONE row set per head over all M columns: vt [H*64, vt_ld],
 *     column = slot(row) - the layout cvx_attention_f16x3_varlen reads (vt_ld >= M rounded up to 32);
 *   - ConvPositionEmbed: zero padding at the ends of every sequence (cvx_dwconv31_gelu_res_varlen_f32).
 * Results per utterance equal its B = 1 run up to fp32 summation order (key tiles stay aligned to 32 packed rows). */
int cvx_attention_varlen_f32(const float* qkv, float* out, uint16_t* out_hi, uint16_t* out_lo,
                             const int32_t* cu_seqlens_dev, int32_t n_seq, int32_t max_T, int32_t H, float scale, cvx_stream_t s);

/* Split-precision variant of cvx_attention_f32 (same reference computation, attend.py:108-126) on
 * v_mfma_f32_32x32x16_f16: inputs are the (fp16 hi, fp16 lo) pairs written by cvx_gemm_f16x3 in QKV mode -
 * qk_* [Bt*T, 2*H*64] (q | k after RoPE) and vt_* [Bt*H*64, Tp] (v transposed per (sequence, head), Tp >= T
 * rounded up to 32, columns >= T finite) - and q.k, p.v are each computed as three fp16 products with fp32
 * accumulation.  Output as for cvx_attention_f32 (fp32 and/or split). */
int cvx_attention_f16x3(const uint16_t* qk_hi, const uint16_t* qk_lo, const uint16_t* vt_hi, const uint16_t* vt_lo,
                        float* out, uint16_t* out_hi, uint16_t* out_lo,
                        int32_t Bt, int32_t T, int32_t Tp, int32_t H, float scale, cvx_stream_t s);
/* the same with activation pre-scales (DEVICE scalars, powers of two, NULL = 1): qk_* hold (q | k) * *qk_scale_dev and
 * vt_* hold v * *v_scale_dev (cvx_gemm_split_io.c_scale_dev / vt_scale_dev of the to_qkv GEMM); both are divided out
 * (the fp32 `out` is the true value) and the split output holds out * *out_scale_dev for the to_out GEMM. */
int cvx_attention_f16x3_scaled(const uint16_t* qk_hi, const uint16_t* qk_lo, const uint16_t* vt_hi, const uint16_t* vt_lo,
                               float* out, uint16_t* out_hi, uint16_t* out_lo,
                               int32_t Bt, int32_t T, int32_t Tp, int32_t H, float scale,
                               const float* qk_scale_dev, const float* v_scale_dev, const float* out_scale_dev, cvx_stream_t s);
/* ragged batch (see RAGGED BATCHES above): qk_* [M, 2*H*64], vt_* [H*64, vt_ld] over all M packed rows */
int cvx_attention_f16x3_varlen(const uint16_t* qk_hi, const uint16_t* qk_lo, const uint16_t* vt_hi, const uint16_t* vt_lo,
                               float* out, uint16_t* out_hi, uint16_t* out_lo, const int32_t* cu_seqlens_dev,
                               int32_t n_seq, int32_t max_T, int64_t M, int32_t vt_ld, int32_t H, float scale,
                               const float* qk_scale_dev, const float* v_scale_dev, const float* out_scale_dev, cvx_stream_t s);

/* y[b,t,c] = GELU( bias[c] + sum_k w[c,k] * x[b,t+k-K/2,c] ) + x[b,t,c]
 * ConvPositionEmbed + residual (acoustic.py:141-161, :508), channels-last, K == 31. */
int cvx_dwconv31_gelu_res_f32(const float* x, const float* w, const float* bias, float* y,
                              int32_t Bt, int32_t T, int32_t C, cvx_stream_t s);
/* ragged batch (see RAGGED BATCHES above): x, y [M, C] packed */
int cvx_dwconv31_gelu_res_varlen_f32(const float* x, const float* w, const float* bias, float* y,
                                     const int32_t* cu_seqlens_dev, int32_t n_seq, int32_t max_T, int32_t C, cvx_stream_t s);
/* C[M][N] = act(A[M][K] . W[N][K]^T + bias) for a HANDFUL of rows (M <= 32, K % 8 == 0; round 3): weight streaming - every W row
 * is read once (non-temporal) by one wave and multiplied with all rows of A from LDS; exact fp32 FMAs.  The adaptive-norm
 * table of a solve (acoustic.py:360-370 evaluated for all n evaluation times at once: 32 x 32,768 x 1,024) and the time MLP. */
int cvx_gemm_skinny_f32(const float* A, int32_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                        int32_t M, int32_t N, int32_t K, int32_t act, cvx_stream_t s);
/* v = f_c*(1+s) - s*f_n   (f_n == NULL: v = f_c)        CFG combine, acoustic.py:428
 * out = y + coef*v ; out2, out3 = optional extra copies  ODE stage update (torchdiffeq midpoint:
 * y_mid = y + f0*dt/2, y1 = y + dt*f_mid); `out` may alias `y`. */
int cvx_cfg_combine_axpy_f32(const float* f_c, const float* f_n, const float* y, float cond_scale,
                             float coef, float* out, float* out2, float* out3, int64_t n, cvx_stream_t s);

/* rows of [ emb(ids[m,0]) | emb(ids[m,1]) ... | cond[m,:] ]  -> out[M, S*E + Cc]
 * (the step-invariant columns of the to_embed input, acoustic.py:496-503).
 * ids == NULL: every id is `null_id`; cond_row != NULL: cond is that single broadcast row
 * (CFG null substitution, acoustic.py:473-494). */
int cvx_embed_gather_f32(const int64_t* ids, int32_t S, const float* table, int32_t E, int32_t n_rows_table,
                         const float* cond, const float* cond_row, int32_t Cc, int64_t null_id,
                         float* out, int64_t M, cvx_stream_t s);

/* out[i, :] = cat( sin(t_i * w * 2pi), cos(t_i * w * 2pi) )   LearnedSinusoidalPosEmb (acoustic.py:107-111) */
int cvx_time_fourier_f32(const float* times, const float* w, float* out, int32_t n, int32_t half,
                         cvx_stream_t s);

/* ------------------------------------------------------------------------
 * HiFi-GAN generator (covomix/vocoder/models.py:75-125, hifi-gan/config_covomix.json)
 *
 * One implicit-GEMM fp32-MFMA kernel covers Conv1d (dilated, "same" padding) and
 * ConvTranspose1d (as a stride-1 conv over the zero-stuffed input with the flipped
 * kernel):  out[b,co,l] = ((bias[co] + sum_{ci,kk} Wp[co,ci,kk] * z[b,ci,l + kk*dil - pad]
 *                          + res[b,co,l]) + accum[b,co,l]) * out_scale
 *   z = leaky_relu(x, in_slope) zero-stuffed by `up` (up == 1: plain conv), zero outside.
 * Wp is the packed weight produced by cvx_hifigan_pack_weight_f32 (host-side layout
 * [co_blk][ci_chunk][kk][CO_T][16]).  res / accum may be NULL; accum may alias out.
 * Covers conv_pre (:81), ups[i] preceded by leaky_relu (:102-103), every ResBlock1
 * conv with its preceding leaky_relu and residual add (:35-42) and the
 * xs accumulate / divide by num_kernels (:104-110). */
/* Ragged batches in the vocoder (items of different length in one launch; the reference vocodes utterances one by
 * one, monologue_generation.py:52-59, :299-304): all tensors are sized for the LONGEST item and item b is valid on its
 * first  item_len_dev[b] * mul + add  OUTPUT positions (the lengths of a stage are an affine function of the item's
 * mel frames: item_len_dev holds the frames, the caller supplies the stage's mul / add).  A kernel given this table
 * writes ZEROS behind an item's last valid position (up to the common length), which is exactly the zero padding the
 * next convolution of a B = 1 run sees there; inputs must obey the same rule (zero-padded mel).  item_len_dev == NULL:
 * every item has the common length. */
typedef struct { const int32_t* item_len_dev; int32_t mul, add; } cvx_item_lengths;
typedef struct {
    const float* x;  int32_t B, Cin, Lin;
    const float* Wp; const float* bias;
    float* out;      int32_t Cout, Lout;
    int32_t ksize, dil, pad, up;
    float in_slope;
    const float* res; const float* accum; float out_scale;
    cvx_item_lengths items;                      /* ragged batch (valid OUTPUT positions per item) or {NULL, 0, 0} */
} cvx_conv_args;
int     cvx_hifigan_conv1d_f32(const cvx_conv_args* a, cvx_stream_t s);
/* number of floats of the packed weight for (Cout, Cin, ksize) */
int64_t cvx_hifigan_packed_weight_floats(int32_t Cout, int32_t Cin, int32_t ksize);
/* host-side (CPU) packing of a Conv1d weight [Cout,Cin,k] (transposed == 0) or a
 * ConvTranspose1d weight [Cin,Cout,k] (transposed != 0, kernel flipped) into Wp. */
int     cvx_hifigan_pack_weight_f32(const float* w, int32_t Cout, int32_t Cin, int32_t ksize,
                                    int32_t transposed, float* Wp);

/* ConvTranspose1d (models.py:85-88, :102-103: leaky_relu then ups[i]) in POLYPHASE form: output l = stride*m + r only sees the
 * taps kk = kk0 + stride*j of the flipped kernel, so every phase r is a stride-1 convolution with ceil((k - kk0)/stride) taps
 * over the input as it is - 1/stride of the matrix work of the zero-stuffed form (cvx_hifigan_conv1d_f32 with up > 1, which
 * stays as the cross-check).  `a` is read exactly like cvx_hifigan_conv1d_f32 reads it for that form (up = stride, dil = 1,
 * pad = ksize - 1 - padding, Lout = (Lin-1)*up + 1 + 2*pad - (ksize-1); no res / accum) except that a->Wp is the phase-major
 * packed weight written by cvx_hifigan_pack_conv_transpose1d_f32 (w = the module's [Cin, Cout, k] weight, pad_t = its padding;
 * cvx_hifigan_conv_transpose1d_packed_floats floats).  amax_bits_dev (optional, DEVICE uint32, zero before the call): the
 * launch leaves the bit pattern of max|out| there - cvx_pow2_scale_from_amax_f32 turns it into the next stage's
 * activation pre-scale (as cvx_amax_pow2_scale_f32 would from a separate pass over out) and zeroes it again. */
int     cvx_hifigan_conv_transpose1d_f32(const cvx_conv_args* a, uint32_t* amax_bits_dev, cvx_stream_t s);
int64_t cvx_hifigan_conv_transpose1d_packed_floats(int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t);
int     cvx_hifigan_pack_conv_transpose1d_f32(const float* w, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                                              int32_t pad_t, float* Wp);
int     cvx_pow2_scale_from_amax_f32(uint32_t* amax_bits_dev, float target, float* scale_dev, cvx_stream_t s);

/* y[b,0,l] = tanh( bias + sum_{ci,k} w[ci,k] * leaky_relu(x[b,ci,l+k-3], slope) )
 * final leaky_relu (default slope 0.01, models.py:112) + conv_post + tanh (:113-114). ksize == 7. */
int cvx_hifigan_post_f32(const float* x, const float* w, float bias, float* y,
                         int32_t B, int32_t Cin, int32_t L, float slope, cvx_stream_t s);

/* ------------------------------------------------------------------------
 * ResBlock1 convolutions (covomix/vocoder/models.py:11-48, 97 % of the vocoder's FLOPs) on the fp16 matrix pipe
 * with split-precision operands (three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate - same scheme and
 * accuracy class as cvx_gemm_f16x3).  Stride-1 "same" convolution, pad = (ksize-1)*dil/2 (utils.py:34-35).
 *
 * All tensors here are CHANNELS-LAST with zero halos: [B][Lp][Cp], position l of batch b at row b*Lp + halo_l + l,
 * rows outside [halo_l, halo_l + L) and channels >= C must hold zeros (the kernels keep them zero);
 * halo_l >= pad, Lp >= halo_l + roundup(L, 256) + 64, Cp = channels rounded up to a multiple of 32.
 *   z_hi/z_lo : input = leaky_relu of the previous value, as (fp16 hi, fp16 lo) pairs, [B][Lp][Cp_in]
 *   w_hi/w_lo : weights pre-scaled by 1/acc_scale and split, packed [Cp_in/32][ksize][Np][32 ci]
 *   v = acc*acc_scale + bias[co] (+ res);   out_x = (v (+ accum)) * out_scale   [fp32, optional]
 *   out_zhi/out_zlo = split(leaky_relu(v, z_slope))                             [optional: the next conv's input]
 * Np (padded output channels) must be 32, 64, 128 or 256; (ksize-1)*dil even and <= 50.
 */
typedef struct {
    const uint16_t *z_hi, *z_lo;
    int32_t B, L, Lp, Cp_in, halo_l;
    const uint16_t *w_hi, *w_lo;
    float acc_scale;
    const float* bias;
    int32_t Np, ksize, dil;
    const float* res;
    const float* accum;
    float* out_x;
    float out_scale;
    uint16_t *out_zhi, *out_zlo;
    float z_slope;
    /* Activation pre-scale of this ResBlock stage (DEVICE scalar, power of two, NULL = 1): z_hi/z_lo hold
     * leaky_relu(x) * *z_scale_dev (the accumulators are divided by it, exactly) and out_zhi/out_zlo are written times
     * it; res / accum / out_x are true fp32 values.  Keeps the split pairs inside fp16's full-precision window whatever
     * the magnitude of the stage's activations (cvx_amax_pow2_scale_f32 measures it from the stage input). */
    const float* z_scale_dev;
    cvx_item_lengths items;                      /* ragged batch (valid positions per item, of L) or {NULL, 0, 0} */
} cvx_conv16_args;
int cvx_hifigan_conv1d_f16x3(const cvx_conv16_args* a, cvx_stream_t s);
/* n (1..3) INDEPENDENT convolutions of one shape (B, L, Lp, Cp_in, Np, halo_l, items; kernel size, dilation, weights, inputs and outputs their
 * own; outputs must not alias) as ONE launch: the same results as n calls above, bit for bit - the tiles of the short kernels fill the rounds
 * of the long one (round 6: the three ResBlocks of a generator stage, kernel sizes 3 / 7 / 11, models.py:104-110). */
int cvx_hifigan_conv1d_group_f16x3(const cvx_conv16_args* a, int32_t n, cvx_stream_t s);

/* ResBlock1.forward (covomix/vocoder/models.py:35-42) on the split-precision convolution above - the operator-level
 * form of section 8(b)'s cvx_hifigan_resblock_* for the path the host actually runs: three times
 *     x = c2(leaky_relu(c1(leaky_relu(x, .1)), .1)) + x        (c1: dilation dil[m], c2: dilation 1)
 * on channels-last buffers [B][Lp][Np] (layout rules of cvx_conv16_args; Np = the stage's padded channel count for
 * inputs and outputs).  x (fp32) and z = split(leaky_relu(x) * *z_scale_dev) are the block's input and are not
 * modified; t, xa/za, xb/zb are caller-owned scratch of the same shapes (xa/za, xb/zb alternate as the intermediate
 * x / z of the pairs).  The last pair writes out = (x_final (+ accum)) * out_scale (accum may alias out): the
 * generator's xs (+)= resblock(x), / num_kernels on the last block (models.py:104-110).  Six launches. */
typedef struct {
    const uint16_t *w_hi, *w_lo; float acc_scale; const float* bias;     /* one convolution: cvx_conv16_args fields */
} cvx_conv16_weights;
typedef struct {
    const float* x; const uint16_t *z_hi, *z_lo;
    int32_t B, L, Lp, Np, halo_l;
    cvx_conv16_weights c1[3], c2[3];
    int32_t ksize; int32_t dil[3];
    uint16_t *t_hi, *t_lo;
    float* xa; uint16_t *za_hi, *za_lo;
    float* xb; uint16_t *zb_hi, *zb_lo;
    const float* accum; float* out; float out_scale;
    const float* z_scale_dev;
    cvx_item_lengths items;                      /* ragged batch (valid positions per item, of L) or {NULL, 0, 0} */
} cvx_resblock16_args;
int cvx_hifigan_resblock_f16x3(const cvx_resblock16_args* a, cvx_stream_t s);
/* The n (1..3) ResBlocks of ONE generator stage (models.py:104-110: xs = sum_j resblocks[j](x), / num_kernels): the results of
 * cvx_hifigan_resblock_f16x3(&blocks[0]) ... (&blocks[n-1]), bit for bit, with the convolutions of different blocks that do not depend on
 * each other sharing launches (cvx_hifigan_conv1d_group_f16x3; 18 -> 8 launches on a wide stage).  The blocks name the same x / z / shape, their
 * OWN scratch (t, xa/za, xb/zb) and may share `out`, which their last convolutions accumulate into in block order (blocks[j].accum = out for
 * j > 0).  Blocks that do not qualify (narrow stages: fused pair kernels; shared scratch) run one after the other. */
int cvx_hifigan_resblock_stage_f16x3(const cvx_resblock16_args* blocks, int32_t n, cvx_stream_t s);

/* One ResBlock1 pair  out = (c2(leaky_relu(c1(leaky_relu(x, .1)), .1)) + x (+ accum)) * out_scale  (models.py:36-40; c1:
 * dilation dil, c2: dilation 1, both kernel size ksize) as ONE kernel for the narrow stages (Np = 32 or 64): the
 * intermediate activation stays in LDS and no split pair exists in HBM - the launch reads x (with a halo) and writes
 * out.  x / accum / out: fp32 channels-last [B][Lp][Np] with the layout rules of cvx_conv16_args, and at least
 * (ksize-1)*(dil+1)/2 zero rows on either side of the signal; out must not alias x (accum may alias out).
 * *z_scale_dev (power of two, NULL = 1) is the pre-scale the in-kernel split pairs of leaky_relu(x) and of the
 * intermediate carry.  cvx_hifigan_resblock_f16x3 uses it for Np <= 64 (its z / t / za / zb buffers may then be NULL). */
typedef struct {
    const float* x;
    int32_t B, L, Lp, Np, halo_l;
    cvx_conv16_weights c1, c2;
    int32_t ksize, dil;
    const float* accum; float* out; float out_scale;
    const float* z_scale_dev;
    int32_t flags;      /* 0; bit 0 (dev A/B): Np = 64 on 128-row tiles, two blocks per CU, instead of 256-row tiles, one per CU */
    cvx_item_lengths items;                      /* ragged batch (valid positions per item, of L) or {NULL, 0, 0} */
} cvx_respair16_args;
int cvx_hifigan_resblock_pair_f16x3(const cvx_respair16_args* a, cvx_stream_t s);

/* Layout converters between the channel-major fp32 tensors of cvx_hifigan_conv1d_f32 ([B][C][L]) and the
 * channels-last buffers above: to_channels_last writes the fp32 copy (x_cl, optional) and / or the split pair of
 * leaky_relu(x, slope) (z_hi/z_lo, optional); from_channels_last the reverse of the fp32 copy. */
int cvx_hifigan_to_channels_last(const float* x, float* x_cl, uint16_t* z_hi, uint16_t* z_lo, int32_t B, int32_t C,
                                 int32_t L, int32_t Lp, int32_t Cp, int32_t halo_l, float slope, cvx_stream_t s);
int cvx_hifigan_to_channels_last_scaled(const float* x, float* x_cl, uint16_t* z_hi, uint16_t* z_lo, int32_t B, int32_t C,
                                        int32_t L, int32_t Lp, int32_t Cp, int32_t halo_l, float slope,
                                        const float* z_scale_dev /* the pair holds leaky_relu(x) * *z_scale_dev; NULL = 1 */,
                                        cvx_stream_t s);
int cvx_hifigan_from_channels_last(const float* x_cl, float* x, int32_t B, int32_t C, int32_t L, int32_t Lp,
                                   int32_t Cp, int32_t halo_l, cvx_stream_t s);
/* leaky_relu + ConvTranspose1d (the upsamplers, covomix/vocoder/models.py:85-88, :102-103) on the split-precision pipe,
 * channels-last in and out (round 3; the fp32 form is cvx_hifigan_conv_transpose1d_f32, channel-major).
 * Stride-1 form: output position stride * m + r only sees the kernel taps c_r + stride * j (c_r = (r + padding) mod stride) at the inputs m + a_r - j
 * (a_r = (r + padding) div stride): the layer is ONE stride-1 convolution over the input with stride * Np_out output
 * columns - column r * Np_out + co of row m is channel co of position stride * m + r, which is exactly the channels-last
 * output buffer read as rows of `stride` positions.  The columns are cut into n_tiles tiles of tile_np (a whole number of
 * phases); tile t runs tile_taps[t] taps, tap kk reading input row m + kk - tile_pad[t], from the packed weights at
 * w_hi/w_lo + tile_w_off[t] (layout [Cp_in/32][tile_taps[t]][tile_np][32 ci], pre-scaled by 1/acc_scale and split, taps a
 * phase does not see are zero).  covomix_amd.ops.hifigan_pack_conv_transpose1d_f16x3 builds weights, bias and table.
 *   z_hi/z_lo : split(leaky_relu(x) * *z_scale_dev) of the input, [B][Lp_in][Cp_in], layout rules of cvx_conv16_args
 *   bias      : [stride * Np_out], bias[co] repeated per phase (zero in the padded channels)
 *   out       : fp32 [B][Lp_out][Np_out], rows halo_out + l for l < L_out = (L_in - 1) * stride + kernel size - 2 * padding are
 *               written (zeros behind an item's end when `items` - lengths in OUTPUT positions - is set); rows m >= L_in of
 *               the stride-1 form (config_covomix's first upsampler: L_out = 5 L_in + 1) read the zero rows behind the input:
 *               ceil(L_out / stride) <= L_in + 32; Lp_out >= halo_out + L_out
 *   amax_bits_dev : optional; receives the bit pattern of max |out| (atomicMax; feed cvx_pow2_scale_from_amax_f32) */
typedef struct {
    const uint16_t *z_hi, *z_lo;
    int32_t B, L_in, Lp_in, Cp_in, halo_in;
    const uint16_t *w_hi, *w_lo;
    float acc_scale;
    const float* bias;
    int32_t Np_out, stride, n_tiles, tile_np;
    int32_t tile_taps[8], tile_pad[8];
    int64_t tile_w_off[8];                       /* in halves */
    float* out;
    int32_t L_out, Lp_out, halo_out;
    const float* z_scale_dev;
    uint32_t* amax_bits_dev;
    cvx_item_lengths items;
} cvx_convt16_args;
int cvx_hifigan_conv_transpose1d_f16x3(const cvx_convt16_args* a, cvx_stream_t s);
/* z = split(leaky_relu(x, slope) * *z_scale_dev) over n floats of a channels-last fp32 buffer (n % 4 == 0; the whole
 * buffer: zero rows and channels stay zero) - the input pair of the call above / of cvx_hifigan_conv1d_f16x3. */
int cvx_hifigan_split_channels_last(const float* x_cl, uint16_t* z_hi, uint16_t* z_lo, int64_t n, float slope,
                                    const float* z_scale_dev, cvx_stream_t s);
/* cvx_hifigan_post_f32 (leaky_relu + conv_post + tanh, models.py:112-114) reading the channels-last stage output
 * [B][Lp][Np] (Np <= 64, halo_l >= 3, zero rows around the signal): same summation order, same bits. y: [B][L]. */
int cvx_hifigan_post_channels_last_f32(const float* x_cl, const float* w, float bias, float* y, int32_t B, int32_t C, int32_t Np,
                                       int32_t L, int32_t Lp, int32_t halo_l, float slope, cvx_stream_t s);
/* *scale_dev = 2^round(log2(target / max|x|)) (1 when x is all zero; exponent clamped to +-40): the power-of-two factor that
 * brings the largest magnitude of x to about `target`.  Everything stays on the device (scratch_dev: one uint32 of
 * caller-owned scratch, any value before the call, ZERO after it - so the same word can serve as a producer's amax_bits_dev
 * next), so a consumer kernel can use the scale without a host round trip. */
int cvx_amax_pow2_scale_f32(const float* x, int64_t n, float target, float* scale_dev, uint32_t* scratch_dev, cvx_stream_t s);

/* ------------------------------------------------------------------------
 * Prompt mel extraction - SURVEY.md section 8f row N3 (data_preparation/generate_mel.py:49-72 as called by
 * monologue_generation.py:62-74): the 480-point windowed DFT of every frame and the mel projection are two calls of
 * cvx_gemm_bias_act_f32 (A = the reflect-padded signal viewed as overlapping rows, lda = hop = 160; W = the
 * hann-windowed cos | sin basis [482, 480]; then the [80, 244] mel basis); these two are the steps in between:
 *   mag[t, k] = sqrt(re[t,k]^2 + im[t,k]^2 + 1e-9)  for k < nb (= 241), 0 for nb <= k < nbp      spec = [T, 2*nb] (re | im)
 *   y[m, t]   = log(max(x[t, m], 1e-5))                                                         x = [T, n_mels]
 */
int cvx_mel_magnitude_f32(const float* spec, float* mag, int64_t T, int32_t nb, int32_t nbp, cvx_stream_t s);
int cvx_mel_log_transpose_f32(const float* x, float* y, int64_t T, int32_t n_mels, cvx_stream_t s);

/* pcm[i] = (int16) trunc( wav[i] * 32768 )   mel_decode_to_wav tail (monologue_generation.py:55-57),
 * numpy astype('int16') semantics for in-range values (C truncation toward zero). */
int cvx_wav_to_int16(const float* wav, int16_t* pcm, int64_t n, cvx_stream_t s);

/* ------------------------------------------------------------------------
 * text2semantic autoregressive decode - SURVEY.md section 8f row N1 ("next" row, built after the hot path).
 *
 * cvx_t2s_decode_steps enqueues n_steps token steps of the sampling loop of TextToSemantic.generate
 * (covomix/covomix_model/text2semantic.py:748-820) at batch 1 with a KV cache: per decoder layer
 * Attention.forward for the causal self-attention (:225-270, rotary_embedding_torch.py:146-157) and the
 * cross-attention over [learned null k/v | encoder context] (:253-262), the GEGLU FeedForward (:154-167), then
 * final RMSNorm (:143-151), tied logits (:545), top_k (:126-132) + gumbel_sample (:105-113) from the caller's
 * U(0,1) draws, eos bookkeeping (:803-818) and the embedding of the sampled ids as the next input.
 * No host synchronisation: positions live in the slot records on the device, so a captured graph replays.
 *
 * Host packing contract (neurips2024-covomix_amd/t2s.py):
 *   wqkv_s [3*inner, dim] = to_q | to_k | to_v rows; inside every 64-row head of to_q and to_k the rows are
 *       permuted (0,2,..,62,1,3,..,63) so the reference's interleaved rotary pairs become half-split pairs;
 *   kv_c   [dialogues, ctx_rows, 2, inner]: row 0 = null_kv, rows 1.. = to_kv(encoder output), computed once per dialogue
 *       (dialogue = utterance; without a queue dialogue b is decoded by slot b);
 *   k_cache / v_cache [slots, max_len, inner]; every per-slot buffer below is [slots, ...] with slots = batch for batch in
 *       {1, 2, 4} and batch rounded up to a multiple of 8 otherwise (the kernels work on groups of 8 slots);
 *   w2     [dim, ff_inner_pad]: the K dimension zero-padded to a multiple of 4;
 *   rope_cos / rope_sin [max_len, 32]: cos / sin(position * freqs[i]);
 *   uniforms [dialogues, uniform_steps, streams, vocab]; tokens [dialogues, streams, max_len] int64;
 *   batch (1..64) slots advance together, EACH AT ITS OWN POSITION; a weight row is read once per step and group of 8
 *   slots, and the arithmetic per utterance does not depend on the batch size, the slot or the other slots' positions
 *   (results are bit-identical to batch 1);
 *   state int32[slots][8] (slot records): [0] position (= tokens produced so far), [1] 1 once an eos was sampled in any
 *       stream, [2] number of steps at that moment, [3] context rows (null row included) used when n_ctx == 0, so that one
 *       captured graph serves utterances of different text length, [4] the dialogue the slot decodes (indexes kv_c, uniforms,
 *       tokens; the caller writes the slot number there when there is no queue), [5] step limit and [6] flags of that
 *       dialogue (queue only; bit 0: an eos does not end it), [7] reserved; x must hold start_token and state[0..2] zeros
 *       before a slot's first step.
 *   Continuous batching (queue != NULL; reference loop per dialogue: text2semantic.py:749-848, its exit :803-818): queue
 *       int32[2] = {next pending dialogue, number of dialogues}; dialogues int32[n][8]: the caller writes [0] context rows,
 *       [1] step limit (<= uniform_steps, <= max_len), [2] flags; the device writes [3] status (0 pending, 1 running, 2 ended
 *       by its eos, 3 by its limit), [4] steps decoded, [5] the slot it ran in.  A slot whose dialogue ends takes the next
 *       pending one inside the sampling kernel of the same step (position 0, `start` as input) or idles when none is left;
 *       the caller fills the first `batch` slots itself (slot b <- dialogue b, queue[0] = batch).  Not with guidance.
 * The caller must not ask for more than max_len steps per slot without a queue (extra steps are ignored on the device).
 */
typedef struct {
    const float *gamma_s, *wqkv_s, *wo_s;        /* self-attention: norm.gamma, packed to_q|to_kv, to_out [dim, inner] */
    const float *gamma_c, *wq_c, *wo_c;          /* cross-attention */
    const float *kv_c;
    const float *gamma_f, *w1, *b1, *w2, *b2;    /* FeedForward: RMSNorm gamma, Linear(dim, 2F)+b, Linear(F, dim)+b */
    float *k_cache, *v_cache;                    /* [max_len, inner] */
} cvx_t2s_layer;

typedef struct {
    int32_t dim, inner, heads, ff_inner, ff_inner_pad, depth, streams, vocab, dim_emb, n_ctx, max_len, top_k;
    int32_t batch, ctx_rows;                     /* slots decoded together (1..64); kv_c rows allocated per dialogue */
    float temperature;
    const cvx_t2s_layer* layers;                 /* HOST array of `depth` entries */
    const float *final_gamma, *emb, *rope_cos, *rope_sin, *uniforms;
    float *x, *q, *att, *h, *logits;             /* per utterance: [dim], [inner], [inner], [ff_inner_pad], [streams, vocab] */
    int64_t* tokens;
    int32_t* state;
    float cfg_scale;                             /* <= 1: off.  > 1: classifier-free guidance (text2semantic.py:780-792), streams == 1, batch
                                                  * even: slot 2u decodes with the text context, slot 2u + 1 with the context masked out
                                                  * (state[3] = 1: the null key / value row only); every step samples slot 2u's token from
                                                  * null + (cond - null) * cfg_scale (uniforms of slot 2u) and feeds it to both slots */
    int32_t uniform_steps;                       /* steps of uniforms allocated per dialogue */
    int32_t* queue;                              /* NULL: slot b decodes dialogue b until the caller stops.  Else continuous batching (above) */
    int32_t* dialogues;
    const float* start;                          /* [dim] start token: the input of a slot that takes a new dialogue (queue != NULL) */
    int32_t group_loop;                          /* tuning hint, 0 = the library's choice: groups of 8 slots one thread block walks with its
                                                  * weight rows in registers (1, 2, 4, 8); the other groups run as blocks of their own */
    int32_t pairs_per_wave;                      /* tuning hint, 0 = the library's choice: output row pairs per wavefront (1 or 2) */
} cvx_t2s_decoder;

int cvx_t2s_decode_steps(const cvx_t2s_decoder* dec, int32_t n_steps, cvx_stream_t stream);

/* out[r, c] = h[r, c] * gelu(h[r, F + c]) for c < F, 0 for F <= c < ld_out   (GEGLU, text2semantic.py:154-157;
 * the encoder's feed-forward; ld_out >= F pads the K dimension of the following GEMM). */
int cvx_geglu_f32(const float* h, float* out, int64_t rows, int32_t F, int64_t ld_out, cvx_stream_t stream);

/* ------------------------------------------------------------------------
 * Operator-level entry points (the names SURVEY.md section 8(b) lists): compositions of the calls above for a C
 * caller that works operator by operator.  The Python host uses the finer-grained entry points because it fuses
 * further (RoPE into the to_qkv GEMM epilogue, the xs accumulation into the last ResBlock convolution).
 *
 * cvx_rope_attention_f32: Attention.forward between to_qkv and to_out (acoustic.py:227-235): qkv [Bt, T, 3*H*64]
 *   WITHOUT rotary embedding; half-split RoPE (:132-137; tables cos/sin [T][32]) on q and k, then the attention of
 *   cvx_attention_f32.  workspace: Bt*T*3*H*64 floats (the rotated copy).
 * cvx_hifigan_convt_f32: the ConvTranspose1d upsamplers (models.py:85-88, :103) = cvx_hifigan_conv_transpose1d_f32
 *   (the polyphase kernel the host path runs) without the max|out| side output: up = stride (> 1),
 *   pad = ksize - 1 - (ksize - up)/2, a->Wp packed by cvx_hifigan_pack_conv_transpose1d_f32.
 * cvx_hifigan_resblock_f32: ResBlock1.forward (models.py:35-42), x [B, C, L] -> out [B, C, L]:
 *   three times  x = c2(leaky_relu(c1(leaky_relu(x, .1)), .1)) + x  with dilations dil[m] (c1) and 1 (c2);
 *   the last add can also apply the generator's  xs (+)= ... / num_kernels  (accum, out_scale; models.py:104-110).
 *   x, tmp, out are three distinct buffers (x is not modified).
 * cvx_hifigan_pre_post_f32: conv_pre (models.py:81, :100; `pre`, may be NULL) and / or the output stage
 *   leaky_relu + conv_post + tanh (:112-114; post_x may be NULL) - see cvx_hifigan_post_f32. */
int cvx_rope_attention_f32(const float* qkv, const float* rope_cos, const float* rope_sin, float* out,
                           int32_t Bt, int32_t T, int32_t H, float scale, float* workspace, cvx_stream_t s);
int64_t cvx_rope_attention_workspace_floats(int32_t Bt, int32_t T, int32_t H);
int cvx_hifigan_convt_f32(const cvx_conv_args* a, cvx_stream_t s);
typedef struct {
    const float* x; int32_t B, C, L;
    const float* Wp1[3]; const float* b1[3];     /* convs1[m]: dilation dil[m]   (packed by cvx_hifigan_pack_weight_f32) */
    const float* Wp2[3]; const float* b2[3];     /* convs2[m]: dilation 1 */
    int32_t ksize; int32_t dil[3];
    float* tmp; float* out;
    const float* accum; float out_scale;         /* out = (resblock(x) (+ accum)) * out_scale; accum may alias out */
} cvx_resblock_args;
int cvx_hifigan_resblock_f32(const cvx_resblock_args* a, cvx_stream_t s);
int cvx_hifigan_pre_post_f32(const cvx_conv_args* pre, const float* post_x, const float* post_w, float post_bias,
                             float* post_y, int32_t B, int32_t C, int32_t L, float slope, cvx_stream_t s);

/* ------------------------------------------------------------------------
 * HuBERT layer-L + k-means prompt tokeniser - SURVEY.md section 8f row N4
 * (fairseq-hubert/examples/textless_nlp/gslm/speech2unit/pretrained/hubert_feature_reader.py:58-78 ->
 *  fairseq-hubert/fairseq/models/hubert/hubert.py:433-480 -> fairseq/models/wav2vec/wav2vec2.py:844-946,1078-1163,
 *  1343-1370; labels: fairseq-hubert/examples/hubert/simple_kmeans/dump_km_label.py:25-43).
 * Activations are channels-last [frames, channels] fp32 throughout, so that
 *   - conv layers 1..6 (Conv1d(C, C, k, stride), no bias, GELU) are cvx_gemm_bias_act_f32 calls over OVERLAPPING rows:
 *     A = previous layer's output, lda = stride*C, K = k*C, W repacked [C_out][k][C_in];
 *   - post_extract_proj, q|k|v, out_proj, fc1 (+GELU), fc2 (+residual) are the same GEMM, attention is
 *     cvx_attention_f32 (H = 12 heads of 64, no RoPE);
 *   - the grouped positional convolution (k = 128, 16 groups) is one GEMM per group over the packed operand below
 *     (K = k*C/groups, lda = C/groups), bias + GELU + residual in its epilogue, written back at column g*C/groups;
 *   - x.C^T of the k-means distance is a GEMM against cluster_centers_ [n_clusters, D].
 * The entry points here are the steps in between. */

/* First conv layer Conv1d(1, C, k, stride) (no bias) + GroupNorm(C, C) (per channel over all frames, biased variance)
 * + exact GELU:  out[l, c], l < L = (n_samples - k) / stride + 1.   wav2vec2.py:856-910 ("default" extractor mode).
 * workspace: cvx_hubert_conv0_workspace_floats(L, C) floats. */
int64_t cvx_hubert_conv0_workspace_floats(int64_t L, int32_t C);
int cvx_hubert_conv0_gn_gelu_f32(const float* wav, int64_t n_samples, const float* w, int32_t C, int32_t k, int32_t stride,
                                 const float* gn_gamma, const float* gn_beta, float eps,
                                 float* out, float* workspace, int64_t workspace_floats, cvx_stream_t s);

/* y[r, :] = (x[r, :] - mean) / sqrt(var + eps) * gamma + beta  (biased variance; fairseq/modules/layer_norm.py).
 * D % 4 == 0, D <= 1024; y may alias x. */
int cvx_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                      int64_t rows, int32_t D, float eps, cvx_stream_t s);

/* out[g][j][c] = x[j - halo][g*D/groups + c] for halo <= j < halo + T, else 0;  out = [groups, T + 2*halo, D/groups].
 * Operand of make_conv_pos's grouped Conv1d (wav2vec2.py:925-946): output frame t of group g reads the contiguous
 * run out[g][t .. t + k - 1][:] (halo = k / 2; SamePad drops the extra last frame of an even kernel). */
int cvx_hubert_group_pack_f32(const float* x, float* out, int32_t T, int32_t D, int32_t groups, int32_t halo, cvx_stream_t s);

/* labels[t] = argmin_j ( (|x[t]|^2 - 2 * dots[t, j]) + cnorm[j] ),  dots = x . C^T  [T, K]   (dump_km_label.py:36-43;
 * ties -> lowest index).  margin (optional) receives second-best minus best distance per frame. */
int cvx_kmeans_argmin_f32(const float* x, const float* dots, const float* cnorm, int64_t* labels, float* margin,
                          int64_t T, int32_t D, int32_t K, cvx_stream_t s);

/* The whole feature extractor behind ONE call (HubertModel.extract_features(source, mask=False, output_layer),
 * hubert.py:433-480, 533-549): all launches are enqueued from C (stepping them from Python costs more host time than the
 * GPU needs).  HuBERT-Base style models: extractor_mode "default" (GroupNorm after the first conv only, no conv bias),
 * post-LN encoder layers, head dim 64.  Weights are borrowed; every cvx_linear carries the fp32 weight [N, K] (validated
 * only) and its cvx_split_f16 halves (w_hi, w_lo = halves of w / inv_scale).  conv[i] (1 <= i < n_conv) is the Conv1d
 * weight repacked [C_out][k][C_in] (K = k * C_in); pos[g] the weight-norm-folded positional conv of group g repacked
 * [C/groups][k][C/groups] with bias = the group's slice; qkv = q_proj | k_proj | v_proj stacked.  `layers` and `pos` are
 * HOST arrays.  out = [cvx_hubert_frames(m, n), dim] fp32, output of encoder layer `output_layer` (1-based, 0 = before
 * the first layer).  workspace: 256-byte aligned device memory of cvx_hubert_workspace_bytes(m, n) bytes. */
typedef struct {
    const float* w; const uint16_t* w_hi; const uint16_t* w_lo; float inv_scale;
    const float* bias;
    int32_t N, K;
} cvx_linear;
typedef struct {
    cvx_linear qkv, out, fc1, fc2;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
} cvx_hubert_layer;
typedef struct {
    int32_t n_conv; int32_t conv_k[8], conv_stride[8], conv_c[8];
    const float *conv0_w, *gn_g, *gn_b;
    cvx_linear conv[8];
    const float *ln_g, *ln_b;
    cvx_linear proj;
    int32_t dim, heads, pos_k, pos_groups;
    const cvx_linear* pos;
    const float *enc_ln_g, *enc_ln_b;
    int32_t n_layers; const cvx_hubert_layer* layers;
} cvx_hubert_model;
int32_t cvx_hubert_frames(const cvx_hubert_model* m, int64_t n_samples);
int64_t cvx_hubert_workspace_bytes(const cvx_hubert_model* m, int64_t n_samples);
int cvx_hubert_extract_features(const cvx_hubert_model* m, const float* wav, int64_t n_samples, int32_t output_layer,
                                float* out, void* workspace, int64_t workspace_bytes, cvx_stream_t s);

/* Sample-rate conversion in front of the tokeniser (hubert_feature_reader.py:38-41: torchaudio.transforms.Resample,
 * third-party; its published polyphase windowed-sinc algorithm): out[i*up + j] = sum_k kern[j][k] * x[i*down + k - width]
 * with x = 0 outside [0, n), kern = [up][2*width + down] computed by the host (covomix_amd.hubert.sinc_resample_kernel),
 * n_out = ceil(up * n / down). */
int cvx_resample_fir_f32(const float* x, int64_t n, const float* kern, int32_t up, int32_t down, int32_t width,
                         float* out, int64_t n_out, cvx_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* COVOMIX_HIP_H */
