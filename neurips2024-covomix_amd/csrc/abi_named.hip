// Operator-level entry points under the names SURVEY.md section 8(b) lists as the minimum symbol set of a replacement
// library.  Each is a short composition of the kernels behind the finer-grained entry points (which the Python host
// uses directly, because it fuses further: RoPE into the to_qkv GEMM epilogue, the xs accumulation into the last
// ResBlock convolution, ...); a C caller that works operator by operator can use these.
#include "cvx_common.h"

namespace {

// q | k | v rows [M, 3*H*64]: copy with the half-split rotation applied to q and k (acoustic.py:132-137):
// x'[j] = x[j]*cos - x[j+32]*sin, x'[j+32] = x[j+32]*cos + x[j]*sin, position = row % T, tables [T][32].
__global__ __launch_bounds__(256) void rope_copy_kernel(const float* __restrict__ qkv, const float* __restrict__ rcos,
                                                        const float* __restrict__ rsin, float* __restrict__ out,
                                                        int64_t M, int T, int H)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // over M * 3*H*32 column pairs
    const int pairs = 3 * H * 32;
    if (i >= M * pairs) return;
    const int64_t row = i / pairs;
    const int pj = (int)(i % pairs), head = pj >> 5, j = pj & 31;     // head over q heads | k heads | v heads
    const int64_t base = row * (3 * H * 64) + head * 64 + j;
    const float lo = qkv[base], hi = qkv[base + 32];
    if (head < 2 * H) {
        const int pos = (int)(row % T);
        const float c = rcos[pos * 32 + j], s = rsin[pos * 32 + j];
        out[base] = __builtin_fmaf(lo, c, -__fmul_rn(hi, s));
        out[base + 32] = __builtin_fmaf(hi, c, __fmul_rn(lo, s));
    } else {
        out[base] = lo; out[base + 32] = hi;
    }
}

}  // namespace

extern "C" int64_t cvx_rope_attention_workspace_floats(int32_t Bt, int32_t T, int32_t H)
{
    return (Bt > 0 && T > 0 && H > 0) ? (int64_t)Bt * T * 3 * H * 64 : 0;
}

extern "C" int cvx_rope_attention_f32(const float* qkv, const float* rope_cos, const float* rope_sin, float* out,
                                      int32_t Bt, int32_t T, int32_t H, float scale, float* workspace, cvx_stream_t s)
{
    CVX_REQUIRE(qkv && rope_cos && rope_sin && out && workspace, "rope_attention: null pointer");
    CVX_REQUIRE(Bt >= 0 && T > 0 && H > 0, "rope_attention: bad shape");
    if (Bt == 0) return CVX_OK;
    const int64_t M = (int64_t)Bt * T, n = M * 3 * H * 32;
    hipLaunchKernelGGL(rope_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cvx_hip_stream(s),
                       qkv, rope_cos, rope_sin, workspace, M, T, H);
    CVX_CHECK_LAUNCH("cvx_rope_attention_f32");
    return cvx_attention_f32(workspace, out, nullptr, nullptr, Bt, T, H, scale, s);
}

extern "C" int cvx_hifigan_convt_f32(const cvx_conv_args* a, cvx_stream_t s)
{
    CVX_REQUIRE(a && a->up > 1, "hifigan_convt: a ConvTranspose1d needs up (its stride) > 1");
    return cvx_hifigan_conv_transpose1d_f32(a, nullptr, s);          // the polyphase kernel the host path runs
}

extern "C" int cvx_hifigan_resblock_f32(const cvx_resblock_args* a, cvx_stream_t s)
{
    CVX_REQUIRE(a && a->x && a->tmp && a->out && a->tmp != a->x && a->tmp != a->out && a->out != a->x, "hifigan_resblock: bad buffers");
    CVX_REQUIRE(a->ksize % 2 == 1 && a->B >= 0 && a->C > 0 && a->L > 0, "hifigan_resblock: bad shape");
    const float* cur = a->x;
    for (int m = 0; m < 3; ++m) {
        CVX_REQUIRE(a->Wp1[m] && a->Wp2[m] && a->dil[m] > 0, "hifigan_resblock: missing weights / dilation of pair %d", m);
        cvx_conv_args c{};
        c.B = a->B; c.Cin = a->C; c.Lin = a->L; c.Cout = a->C; c.Lout = a->L; c.ksize = a->ksize; c.up = 1; c.in_slope = 0.1f;
        c.out_scale = 1.0f;
        // xt = c1(leaky_relu(x)), dilated                      models.py:36-37
        c.x = cur; c.Wp = a->Wp1[m]; c.bias = a->b1[m]; c.out = a->tmp; c.dil = a->dil[m]; c.pad = (a->ksize - 1) * a->dil[m] / 2;
        int rc = cvx_hifigan_conv1d_f32(&c, s);
        if (rc != CVX_OK) return rc;
        // x = c2(leaky_relu(xt)) + x                           models.py:38-40   (+ the caller's xs accumulate on the last pair)
        c.x = a->tmp; c.Wp = a->Wp2[m]; c.bias = a->b2[m]; c.out = a->out; c.dil = 1; c.pad = (a->ksize - 1) / 2; c.res = cur;
        if (m == 2) { c.accum = a->accum; c.out_scale = a->out_scale; }
        rc = cvx_hifigan_conv1d_f32(&c, s);
        if (rc != CVX_OK) return rc;
        cur = a->out;
    }
    return CVX_OK;
}

extern "C" int cvx_hifigan_resblock_f16x3(const cvx_resblock16_args* a, cvx_stream_t s)
{
    CVX_REQUIRE(a && a->x && a->out && a->xa && a->xb, "hifigan_resblock_f16x3: null pointer");
    if (a->Np <= 64) {                                                      // narrow stages: one kernel per pair, no split pairs in HBM
        const float* cur = a->x;
        for (int m = 0; m < 3; ++m) {
            cvx_respair16_args r{};
            r.x = cur; r.B = a->B; r.L = a->L; r.Lp = a->Lp; r.Np = a->Np; r.halo_l = a->halo_l;
            r.c1 = a->c1[m]; r.c2 = a->c2[m]; r.ksize = a->ksize; r.dil = a->dil[m]; r.z_scale_dev = a->z_scale_dev; r.items = a->items;
            r.out_scale = 1.0f;
            if (m < 2) r.out = (m == 0) ? a->xa : a->xb;
            else { r.out = a->out; r.accum = a->accum; r.out_scale = a->out_scale; }
            const int rc = cvx_hifigan_resblock_pair_f16x3(&r, s);
            if (rc != CVX_OK) return rc;
            cur = r.out;
        }
        return CVX_OK;
    }
    CVX_REQUIRE(a->z_hi && a->z_lo && a->t_hi && a->t_lo, "hifigan_resblock_f16x3: null pointer");
    CVX_REQUIRE(a->za_hi && a->za_lo && a->zb_hi && a->zb_lo, "hifigan_resblock_f16x3: missing scratch buffers");
    const float* cur_x = a->x;
    const uint16_t *cur_zh = a->z_hi, *cur_zl = a->z_lo;
    for (int m = 0; m < 3; ++m) {
        CVX_REQUIRE(a->c1[m].w_hi && a->c2[m].w_hi && a->dil[m] > 0, "hifigan_resblock_f16x3: missing weights / dilation of pair %d", m);
        cvx_conv16_args c{};
        c.B = a->B; c.L = a->L; c.Lp = a->Lp; c.Cp_in = a->Np; c.halo_l = a->halo_l; c.Np = a->Np; c.ksize = a->ksize;
        c.z_slope = 0.1f; c.out_scale = 1.0f; c.z_scale_dev = a->z_scale_dev; c.items = a->items;
        // t = split(leaky_relu(c1(z)))                                     models.py:36-38
        c.z_hi = cur_zh; c.z_lo = cur_zl; c.dil = a->dil[m];
        c.w_hi = a->c1[m].w_hi; c.w_lo = a->c1[m].w_lo; c.acc_scale = a->c1[m].acc_scale; c.bias = a->c1[m].bias;
        c.out_zhi = a->t_hi; c.out_zlo = a->t_lo;
        int rc = cvx_hifigan_conv1d_f16x3(&c, s);
        if (rc != CVX_OK) return rc;
        // x' = c2(t) + x ; z' = split(leaky_relu(x'))                       models.py:38-40
        c.z_hi = a->t_hi; c.z_lo = a->t_lo; c.dil = 1;
        c.w_hi = a->c2[m].w_hi; c.w_lo = a->c2[m].w_lo; c.acc_scale = a->c2[m].acc_scale; c.bias = a->c2[m].bias;
        c.res = cur_x;
        if (m < 2) {
            float* ox = (m == 0) ? a->xa : a->xb;
            c.out_x = ox; c.out_zhi = (m == 0) ? a->za_hi : a->zb_hi; c.out_zlo = (m == 0) ? a->za_lo : a->zb_lo;
            rc = cvx_hifigan_conv1d_f16x3(&c, s);
            if (rc != CVX_OK) return rc;
            cur_x = ox; cur_zh = c.out_zhi; cur_zl = c.out_zlo;
        } else {                                                            // last pair: fold into the generator's xs
            c.out_x = a->out; c.accum = a->accum; c.out_scale = a->out_scale; c.out_zhi = nullptr; c.out_zlo = nullptr;
            rc = cvx_hifigan_conv1d_f16x3(&c, s);
            if (rc != CVX_OK) return rc;
        }
    }
    return CVX_OK;
}

// The ResBlocks of one generator stage (models.py:104-110: xs = sum_j resblocks[j](x), / num_kernels) - the calls
// cvx_hifigan_resblock_f16x3(&blocks[0]), ..., (&blocks[n-1]) with the SAME results, bit for bit, but with the convolutions that do not
// depend on each other sharing a launch: the blocks read the same x / z, keep their own scratch, and only meet in `out`, which their last
// convolutions accumulate into IN BLOCK ORDER (those stay one launch each).  Narrow stages (fused pair kernels) run block by block.
extern "C" int cvx_hifigan_resblock_stage_f16x3(const cvx_resblock16_args* blocks, int32_t n, cvx_stream_t s)
{
    CVX_REQUIRE(blocks && n >= 1 && n <= 3, "hifigan_resblock_stage_f16x3: 1..3 ResBlocks per stage call (got %d)", n);
    bool group = n > 1 && blocks[0].Np > 64;
    for (int j = 1; j < n && group; ++j) {
        const cvx_resblock16_args &a = blocks[j], &b = blocks[0];
        group = a.x == b.x && a.z_hi == b.z_hi && a.z_lo == b.z_lo && a.B == b.B && a.L == b.L && a.Lp == b.Lp && a.Np == b.Np &&
                a.halo_l == b.halo_l && a.z_scale_dev == b.z_scale_dev && a.items.item_len_dev == b.items.item_len_dev &&
                a.items.mul == b.items.mul && a.items.add == b.items.add;
        for (int h = 0; h < j && group; ++h)          // own scratch each (the grouped convolutions run concurrently)
            group = a.t_hi != blocks[h].t_hi && a.xa != blocks[h].xa && a.xb != blocks[h].xb && a.za_hi != blocks[h].za_hi &&
                    a.zb_hi != blocks[h].zb_hi;
    }
    if (!group) {
        for (int j = 0; j < n; ++j) {
            const int rc = cvx_hifigan_resblock_f16x3(&blocks[j], s);
            if (rc != CVX_OK) return rc;
        }
        return CVX_OK;
    }
    for (int j = 0; j < n; ++j) {
        const cvx_resblock16_args& a = blocks[j];
        CVX_REQUIRE(a.x && a.out && a.xa && a.xb && a.z_hi && a.z_lo && a.t_hi && a.t_lo && a.za_hi && a.za_lo && a.zb_hi && a.zb_lo,
                    "hifigan_resblock_stage_f16x3: null pointer in block %d", j);
    }
    cvx_conv16_args c[3];
    for (int m = 0; m < 3; ++m) {
        // t_j = split(leaky_relu(c1_j(z_j)))                                models.py:36-38 - all blocks in one launch
        for (int j = 0; j < n; ++j) {
            const cvx_resblock16_args& a = blocks[j];
            CVX_REQUIRE(a.c1[m].w_hi && a.c2[m].w_hi && a.dil[m] > 0, "hifigan_resblock_stage_f16x3: missing weights / dilation of pair %d", m);
            cvx_conv16_args& q = c[j];
            q = cvx_conv16_args{};
            q.B = a.B; q.L = a.L; q.Lp = a.Lp; q.Cp_in = a.Np; q.halo_l = a.halo_l; q.Np = a.Np; q.ksize = a.ksize;
            q.z_slope = 0.1f; q.out_scale = 1.0f; q.z_scale_dev = a.z_scale_dev; q.items = a.items;
            q.z_hi = m == 0 ? a.z_hi : (m == 1 ? a.za_hi : a.zb_hi); q.z_lo = m == 0 ? a.z_lo : (m == 1 ? a.za_lo : a.zb_lo);
            q.dil = a.dil[m];
            q.w_hi = a.c1[m].w_hi; q.w_lo = a.c1[m].w_lo; q.acc_scale = a.c1[m].acc_scale; q.bias = a.c1[m].bias;
            q.out_zhi = a.t_hi; q.out_zlo = a.t_lo;
        }
        int rc = cvx_hifigan_conv1d_group_f16x3(c, n, s);
        if (rc != CVX_OK) return rc;
        // x'_j = c2_j(t_j) + x_j ; z'_j = split(leaky_relu(x'_j))           models.py:38-40
        for (int j = 0; j < n; ++j) {
            const cvx_resblock16_args& a = blocks[j];
            cvx_conv16_args& q = c[j];
            q.z_hi = a.t_hi; q.z_lo = a.t_lo; q.dil = 1;
            q.w_hi = a.c2[m].w_hi; q.w_lo = a.c2[m].w_lo; q.acc_scale = a.c2[m].acc_scale; q.bias = a.c2[m].bias;
            q.res = m == 0 ? a.x : (m == 1 ? a.xa : a.xb);
            if (m < 2) {
                q.out_x = m == 0 ? a.xa : a.xb;
                q.out_zhi = m == 0 ? a.za_hi : a.zb_hi; q.out_zlo = m == 0 ? a.za_lo : a.zb_lo;
            } else {                                                        // last pair: folds into the generator's xs, block after block
                q.out_x = a.out; q.accum = a.accum; q.out_scale = a.out_scale; q.out_zhi = nullptr; q.out_zlo = nullptr;
            }
        }
        if (m < 2) rc = cvx_hifigan_conv1d_group_f16x3(c, n, s);
        else
            for (int j = 0; j < n && rc == CVX_OK; ++j) rc = cvx_hifigan_conv1d_f16x3(&c[j], s);
        if (rc != CVX_OK) return rc;
    }
    return CVX_OK;
}

extern "C" int cvx_hifigan_pre_post_f32(const cvx_conv_args* pre, const float* post_x, const float* post_w, float post_bias,
                                        float* post_y, int32_t B, int32_t C, int32_t L, float slope, cvx_stream_t s)
{
    CVX_REQUIRE(pre || post_x, "hifigan_pre_post: nothing to do");
    if (pre) {
        CVX_REQUIRE(pre->up == 1 && pre->in_slope == 1.0f, "hifigan_pre_post: conv_pre is a plain Conv1d (up = 1, no input leaky_relu)");
        const int rc = cvx_hifigan_conv1d_f32(pre, s);
        if (rc != CVX_OK) return rc;
    }
    if (post_x) return cvx_hifigan_post_f32(post_x, post_w, post_bias, post_y, B, C, L, slope, s);
    return CVX_OK;
}
