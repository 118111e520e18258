"""Per-kernel parity: every C-ABI entry point against a plain PyTorch fp32 (fp64 where cheap)
reference of the same op, on the GPU, called through the C ABI (ctypes)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = 2e-6   # fp32 MFMA == fmaf chain; differences are summation order only


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import covomix_amd.ops as o
    return o


def dev():
    return torch.device("cuda:0")


def randn(*s, seed=0):
    g = torch.Generator().manual_seed(seed + sum(s))
    return torch.randn(*s, generator=g).to(dev())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 1024, 1024), (1000, 3072, 1024), (37, 80, 1024),
                                   (999, 1024, 80), (130, 4096, 1024), (64, 1024, 4096), (5, 128, 64)])
def test_gemm_plain(ops, M, N, K):
    a, w = randn(M, K, seed=1), randn(N, K, seed=2) / math.sqrt(K)
    out = torch.full((M, N), float("nan"), device=dev())
    ops.gemm(a, w, out)
    ref = (a.double() @ w.double().T)
    assert rel_l2(out, ref) < TOL
    # asymmetric-identity check (transpose detector)
    eye = torch.eye(K, device=dev())[:M] if M <= K else None
    if eye is not None:
        o2 = torch.empty(eye.shape[0], N, device=dev())
        ops.gemm(eye.contiguous(), w, o2)
        assert torch.allclose(o2, w.T[: eye.shape[0]], atol=1e-6)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_epilogues(ops, act):
    M, N, K = 300, 256, 512
    a, w, b, r = randn(M, K, seed=3), randn(N, K, seed=4) / math.sqrt(K), randn(N, seed=5), randn(M, N, seed=6)
    out = torch.empty(M, N, device=dev())
    ops.gemm(a, w, out, bias=b, act=act, residual=r)
    z = a.double() @ w.double().T + b.double()
    z = [z, F.gelu(z), F.silu(z)][act] + r.double()
    assert rel_l2(out, z) < TOL
    # in-place residual (C aliases residual)
    c = r.clone()
    ops.gemm(a, w, c, bias=b, act=act, residual=c)
    assert rel_l2(c, z) < TOL


def test_gemm_split_k_concat_and_strided_w(ops):
    M, N = 257, 384
    x, s = randn(M, 256, seed=7), randn(M, 256, seed=8)
    w = randn(N, 512, seed=9) / math.sqrt(512)
    b = randn(N, seed=10)
    out = torch.empty(M, N, device=dev())
    ops.gemm(x, w, out, bias=b, a2=s)
    ref = torch.cat((x, s), -1).double() @ w.double().T + b.double()
    assert rel_l2(out, ref) < TOL
    # W as a column slice of a wider matrix (to_embed x / rest split) and K=80 tail
    wide = randn(N, 2288, seed=11) / 40
    a80 = randn(M, 80, seed=12)
    ops.gemm(a80, wide[:, :80], out)
    assert rel_l2(out, a80.double() @ wide[:, :80].double().T) < TOL
    arest = randn(M, 2208, seed=13)
    ops.gemm(arest, wide[:, 80:], out, bias=b)
    assert rel_l2(out, arest.double() @ wide[:, 80:].double().T + b.double()) < TOL


def test_gemm_rope_epilogue(ops):
    Bt, T, H = 3, 77, 2
    dim = 128
    x = randn(Bt * T, dim, seed=14)
    w = randn(3 * H * 64, dim, seed=15) / math.sqrt(dim)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(T).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().to(dev()).contiguous(), ang.sin().to(dev()).contiguous()
    out = torch.empty(Bt * T, 3 * H * 64, device=dev())
    ops.gemm(x, w, out, rope=(cos, sin), rope_cols=2 * H * 64)
    qkv = (x.double() @ w.double().T).reshape(Bt, T, 3, H, 64).cpu()
    full = torch.cat((ang, ang), -1).double()[None, :, None, :]

    def rot(t):
        r = torch.cat((-t[..., 32:], t[..., :32]), -1)
        return t * full.cos() + r * full.sin()
    ref = torch.stack((rot(qkv[:, :, 0]), rot(qkv[:, :, 1]), qkv[:, :, 2]), dim=2).reshape(Bt * T, -1)
    assert rel_l2(out, ref) < TOL


def test_gemm_rejects_bad_args(ops):
    import covomix_amd._lib as L
    a, w = randn(8, 30), randn(8, 30)
    with pytest.raises(L.CovomixHipError):
        ops.gemm(a, w, torch.empty(8, 8, device=dev()))          # K % 4 != 0


@pytest.mark.parametrize("D", [1024, 128, 512, 2048])
def test_adarmsnorm(ops, D):
    rows = 333
    x, g, b = randn(rows, D, seed=20), randn(D, seed=21), randn(D, seed=22)
    y = torch.empty_like(x)
    ops.adarmsnorm(x, g, b, y)
    ref = F.normalize(x.double(), dim=-1) * D ** 0.5 * g.double() + b.double()
    assert rel_l2(y, ref) < TOL
    ops.adarmsnorm(x, g, None, y)
    assert rel_l2(y, F.normalize(x.double(), dim=-1) * D ** 0.5 * g.double()) < TOL
    # per-group gamma/beta (per-batch times) and an all-zero row (eps clamp)
    x2 = x[:300].clone().contiguous(); x2[7] = 0
    g2, b2 = randn(3, D, seed=23), randn(3, D, seed=24)
    y2 = torch.empty_like(x2)
    ops.adarmsnorm(x2, g2, b2, y2, rows_per_group=100)
    ref2 = F.normalize(x2.double().reshape(3, 100, D), dim=-1) * D ** 0.5 * g2.double()[:, None] + b2.double()[:, None]
    assert rel_l2(y2, ref2.reshape(300, D)) < TOL
    assert torch.isfinite(y2).all()


@pytest.mark.parametrize("Bt,T,H", [(2, 128, 2), (1, 37, 1), (3, 200, 2), (2, 1000, 1), (1, 129, 3)])
def test_attention(ops, Bt, T, H):
    qkv = randn(Bt, T, 3 * H * 64, seed=30)
    qkv[..., : 2 * H * 64] *= 1.5
    out = torch.full((Bt, T, H * 64), float("nan"), device=dev())
    ops.attention(qkv.contiguous(), out, Bt, T, H, 0.125)
    q, k, v = qkv.double().reshape(Bt, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(Bt, T, H * 64)
    assert rel_l2(out, ref) < 5e-6


def test_attention_forced_rescale(ops):
    """A late key that dominates one query forces the online-softmax rescale branch."""
    Bt, T, H = 1, 160, 1
    qkv = randn(Bt, T, 192, seed=31)
    qkv[0, 5, :64] = 3.0
    qkv[0, 140, 64:128] = 4.0        # key 140 spikes against query 5 in a late tile
    out = torch.empty(Bt, T, 64, device=dev())
    ops.attention(qkv.contiguous(), out, Bt, T, H, 0.125)
    q, k, v = qkv.double().reshape(Bt, T, 3, 1, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(Bt, T, 64)
    assert rel_l2(out, ref) < 5e-6


@pytest.mark.parametrize("Bt,T,C", [(2, 100, 1024), (1, 31, 128), (3, 65, 256)])
def test_dwconv31(ops, Bt, T, C):
    x, w, b = randn(Bt, T, C, seed=40), randn(C, 31, seed=41) / 5, randn(C, seed=42)
    y = torch.empty_like(x)
    ops.dwconv31_gelu_res(x, w, b, y, Bt, T)
    xd = x.double().transpose(1, 2)
    ref = F.gelu(F.conv1d(xd, w.double()[:, None, :], b.double(), padding=15, groups=C)).transpose(1, 2) + x.double()
    assert rel_l2(y, ref) < TOL


@pytest.mark.parametrize("M,N,K,act", [(32, 32768, 1024, 0), (7, 1000, 512, 2), (1, 9, 64, 0), (32, 130, 1024, 1), (20, 4096, 264, 0), (32, 2048, 4096, 0), (5, 333, 1032, 2)])
def test_gemm_skinny(ops, M, N, K, act):
    """The weight-streaming GEMM for a handful of rows vs fp64 (odd N: the last wave owns one row; K not a multiple of 128)."""
    a, w, b = randn(M, K, seed=55), randn(N, K, seed=56) / math.sqrt(K), randn(N, seed=57)
    out = torch.full((M, N), float("nan"), device=dev())
    ops.gemm_skinny(a, w, out, bias=b, act=act)
    ref = a.double() @ w.double().t() + b.double()
    ref = F.gelu(ref) if act == 1 else F.silu(ref) if act == 2 else ref
    assert rel_l2(out, ref) < TOL
    tiled = torch.empty(M, N, device=dev())
    ops.gemm(a, w, tiled, bias=b, act=act)
    assert rel_l2(out, tiled.double()) < 1e-6


def test_cfg_axpy_gather_fourier_int16(ops):
    n = 12345
    fc, fn, y = randn(n, seed=50), randn(n, seed=51), randn(n, seed=52)
    o1, o2, o3 = (torch.empty(n, device=dev()) for _ in range(3))
    ops.cfg_combine_axpy(fc, fn, y, 0.7, 0.03125, o1, o2, o3)
    ref = y + (fc * 1.7 - 0.7 * fn) * 0.03125
    assert rel_l2(o1, ref) < 1e-6 and torch.equal(o1, o2) and torch.equal(o1, o3)
    yy = y.clone()
    ops.cfg_combine_axpy(fc, None, yy, 1.0, 0.5, yy)              # in place, no null branch
    assert rel_l2(yy, y + fc * 0.5) < 1e-6
    # gather
    M, S, E, Cc = 50, 2, 64, 160
    table = randn(503, E, seed=53)
    ids = torch.randint(0, 502, (M, S), generator=torch.Generator().manual_seed(1)).to(dev())
    cond = randn(M, Cc, seed=54)
    out = torch.empty(M, S * E + Cc, device=dev())
    ops.embed_gather(ids, S, table, cond, None, Cc, 502, out, M)
    ref = torch.cat((table[ids].reshape(M, -1), cond), -1)
    assert torch.equal(out, ref)
    row = randn(Cc, seed=55)
    ops.embed_gather(None, S, table, None, row, Cc, 502, out, M)
    ref = torch.cat((table[502].repeat(M, S), row.expand(M, Cc)), -1)
    assert torch.equal(out, ref)
    # fourier
    t = torch.tensor([0.0, 0.03125, 0.5, 0.96875], device=dev())
    w = randn(512, seed=56)
    f = torch.empty(4, 1024, device=dev())
    ops.time_fourier(t, w, f)
    ang = t[:, None] * w[None, :] * 2 * math.pi
    assert torch.allclose(f, torch.cat((ang.sin(), ang.cos()), -1), atol=2e-6)
    # int16 (numpy astype semantics)
    wav = torch.tensor([0.0, 0.5, -0.5, 0.99999, -1.0, 1e-6, -3.0517578e-05 * 1.5], device=dev())
    pcm = ops.wav_to_int16(wav)
    assert (pcm.cpu().numpy() == (wav.cpu() * 32768.0).numpy().astype("int16")).all()


@pytest.mark.parametrize("B,Cin,Cout,L,k,d", [(2, 80, 500, 50, 7, 1), (1, 250, 250, 251, 11, 5), (2, 125, 125, 300, 3, 3),
                                               (1, 62, 62, 1000, 7, 1), (2, 31, 31, 700, 11, 3), (1, 16, 64, 33, 3, 1)])
def test_hifigan_conv1d(ops, B, Cin, Cout, L, k, d):
    x = randn(B, Cin, L, seed=60)
    w = randn(Cout, Cin, k, seed=61) / math.sqrt(Cin * k)
    b = randn(Cout, seed=62)
    pad = (k * d - d) // 2
    wp = ops.hifigan_pack_weight(w, False).to(dev())
    out = torch.full((B, Cout, L), float("nan"), device=dev())
    ops.hifigan_conv1d(x, wp, b, out, cout=Cout, ksize=k, dil=d, pad=pad, in_slope=0.1)
    ref = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=d, padding=pad)
    assert rel_l2(out, ref) < TOL
    res, acc = randn(B, Cout, L, seed=63), randn(B, Cout, L, seed=64)
    o2 = acc.clone()
    ops.hifigan_conv1d(x, wp, b, o2, cout=Cout, ksize=k, dil=d, pad=pad, in_slope=0.1, res=res, accum=o2, out_scale=1 / 3)
    assert rel_l2(o2, (ref + res.double() + acc.double()) / 3) < TOL


@pytest.mark.parametrize("Cin,Cout,L,k,u", [(500, 250, 50, 8, 5), (250, 125, 101, 8, 4), (125, 62, 64, 4, 4), (62, 31, 200, 4, 2)])
def test_hifigan_conv_transpose(ops, Cin, Cout, L, k, u):
    B = 2
    x = randn(B, Cin, L, seed=70)
    w = randn(Cin, Cout, k, seed=71) / math.sqrt(Cin * k / u)
    b = randn(Cout, seed=72)
    p = (k - u) // 2
    lout = (L - 1) * u - 2 * p + k
    wp = ops.hifigan_pack_weight(w, True).to(dev())
    out = torch.full((B, Cout, lout), float("nan"), device=dev())
    ops.hifigan_conv1d(x, wp, b, out, cout=Cout, ksize=k, dil=1, pad=k - 1 - p, up=u, in_slope=0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=p)
    assert rel_l2(out, ref) < TOL


def _to_cl(ops, x, cp, scale=1.0, slope=None):
    """[B, C, L] (CPU/GPU fp32) -> zero-haloed channels-last fp32 [B, Lp, cp] (leaky_relu * scale applied when slope is given)."""
    B, C_, L = x.shape
    buf = torch.zeros(B, ops.hifigan_cl_rows(L), cp, device=dev())
    v = x if slope is None else F.leaky_relu(x, slope) * scale
    buf[:, ops.HIFI_HALO_L:ops.HIFI_HALO_L + L, :C_] = v.transpose(1, 2)
    return buf


@pytest.mark.parametrize("Cin,Cout,L,k,u", [(500, 250, 50, 8, 5), (250, 125, 301, 8, 4), (125, 62, 256, 4, 4), (62, 31, 700, 4, 2),
                                            (500, 250, 512, 11, 5), (64, 40, 130, 3, 3), (33, 31, 77, 2, 2), (96, 128, 260, 16, 8)])
def test_conv_transpose1d_f16x3(ops, Cin, Cout, L, k, u):
    """leaky_relu + ConvTranspose1d on the split pipe (stride-1 form, channels-last in and out) vs fp64 conv_transpose1d: the four
    upsamplers of config_covomix (the first has kernel - 2 padding = stride + 1: one output more than stride * L), a kernel as
    long as its stride (one tap per phase), stride 8, L a multiple of the tile; with a measured pre-scale on
    inputs 2^12 times larger, the fused max|out|, and a ragged batch (zeros behind the shorter item)."""
    B = 2
    p = (k - u) // 2
    x = randn(B, Cin, L, seed=70) * 4096.0
    w = randn(Cin, Cout, k, seed=71) / math.sqrt(Cin * k / u)
    b = randn(Cout, seed=72)
    pk = ops.hifigan_pack_conv_transpose1d_f16x3(w, b, u, p)
    zs = torch.empty(1, device=dev()); scr = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.amax_pow2_scale(x, 1024.0, zs, scr)
    xin = _to_cl(ops, x, pk["cp_in"])
    z = (torch.empty_like(xin, dtype=torch.float16), torch.empty_like(xin, dtype=torch.float16))
    ops.hifigan_split_channels_last(xin, z, 0.1, z_scale=zs)
    lout = (L - 1) * u + k - 2 * p
    out = torch.zeros(B, ops.hifigan_cl_rows(lout), pk["np_out"], device=dev())
    amax = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.hifigan_conv_transpose1d_f16x3(z, pk, B, L, out, lout, z_scale=zs, amax_bits=amax)
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=p)
    assert ref.shape[2] == lout
    got = out[:, ops.HIFI_HALO_L:ops.HIFI_HALO_L + lout, :Cout].transpose(1, 2)
    assert rel_l2(got, ref) < 2e-6
    assert float((got.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    assert float(out[:, :ops.HIFI_HALO_L].abs().max()) == 0.0 and float(out[:, ops.HIFI_HALO_L + lout:].abs().max()) == 0.0
    assert Cout == pk["np_out"] or float(out[:, :, Cout:].abs().max()) == 0.0              # padded channels stay zero
    assert abs(float(amax.view(torch.float32)) - float(got.abs().max())) == 0.0
    # ragged: item 1 is valid on L - 7 input positions -> u * (L - 7) outputs, zeros behind them, item 0 untouched
    lens = torch.tensor([L, L - 7], dtype=torch.int32, device=dev())
    x2 = x.clone(); x2[1, :, L - 7:] = 0.0
    xin2 = _to_cl(ops, x2, pk["cp_in"])
    ops.hifigan_split_channels_last(xin2, z, 0.1, z_scale=zs)
    out2 = torch.zeros_like(out)
    ops.hifigan_conv_transpose1d_f16x3(z, pk, B, L, out2, lout, z_scale=zs, items=(lens, u, k - 2 * p - u))
    l1 = (L - 7 - 1) * u + k - 2 * p
    assert torch.equal(out2[0], out[0])
    assert float(out2[1, ops.HIFI_HALO_L + l1:].abs().max()) == 0.0
    one = F.conv_transpose1d(F.leaky_relu(x2[1:, :, :L - 7].double(), 0.1), w.double(), b.double(), stride=u, padding=p)
    assert one.shape[2] == l1
    assert rel_l2(out2[1, ops.HIFI_HALO_L:ops.HIFI_HALO_L + l1, :Cout].t(), one[0]) < 2e-6


def test_hifigan_post_channels_last_same_bits(ops):
    B, C, L = 3, 31, 1000
    x, w = randn(B, C, L, seed=80), randn(1, C, 7, seed=81) / 10
    y0, y1 = torch.empty(B, 1, L, device=dev()), torch.empty(B, 1, L, device=dev())
    ops.hifigan_post(x, w.reshape(C, 7).contiguous(), 0.05, y0)
    ops.hifigan_post_channels_last(_to_cl(ops, x, 32), C, L, w.reshape(C, 7).contiguous(), 0.05, y1)
    assert torch.equal(y0, y1)


def test_hifigan_post(ops):
    B, C, L = 2, 31, 999
    x, w = randn(B, C, L, seed=80), randn(1, C, 7, seed=81) / 10
    y = torch.empty(B, 1, L, device=dev())
    ops.hifigan_post(x, w.reshape(C, 7).contiguous(), 0.05, y)
    ref = torch.tanh(F.conv1d(F.leaky_relu(x.double(), 0.01), w.double(), torch.tensor([0.05], dtype=torch.float64, device=dev()), padding=3))
    assert rel_l2(y, ref) < TOL


@pytest.mark.parametrize("M,N,K", [(256, 1024, 1024), (1000, 3072, 1024), (130, 4096, 1024), (999, 1024, 4096), (300, 80, 1024)])
def test_gemm_f16x3_accuracy(ops, M, N, K):
    """Split-precision (fp16 hi/lo, 3 MFMA products) GEMM: error class of fp32, far below a plain fp16/bf16 cast."""
    a, w = randn(M, K, seed=101) * 3.0, randn(N, K, seed=102) / math.sqrt(K)
    b, r = randn(N, seed=103), randn(M, N, seed=104)
    ws = ops.split_f16(w)
    assert torch.isfinite(ws[0].float()).all() and torch.isfinite(ws[1].float()).all()
    out = torch.full((M, N), float("nan"), device=dev())
    ops.gemm(a, w, out, bias=b, act=1, residual=r, w_split=ws)
    ref = F.gelu(a.double() @ w.double().T + b.double()) + r.double()
    e3 = rel_l2(out, ref)
    o32 = torch.empty_like(out)
    ops.gemm(a, w, o32, bias=b, act=1, residual=r)
    e32 = rel_l2(o32, ref)
    half = F.gelu((a.half().double() @ w.half().double().T) + b.double()) + r.double()
    e16 = rel_l2(half, ref)
    print(f"f16x3 {e3:.2e}  fp32 {e32:.2e}  plain-fp16 {e16:.2e}")
    assert e3 < 3e-6 and e3 < e16 / 50


def test_gemm_f16x3_split_k_rope_and_range(ops):
    M, N = 257, 384
    x, s = randn(M, 256, seed=110), randn(M, 256, seed=111)
    w = randn(N, 512, seed=112) / math.sqrt(512)
    out = torch.empty(M, N, device=dev())
    ops.gemm(x, w, out, a2=s, w_split=ops.split_f16(w))
    assert rel_l2(out, torch.cat((x, s), -1).double() @ w.double().T) < 3e-6
    # tiny and huge magnitudes: fp16 subnormal lo parts and saturation
    xs = randn(M, 256, seed=113) * 1e-3
    ws_ = randn(N, 256, seed=114) * 1e-2
    ops.gemm(xs, ws_, out, w_split=ops.split_f16(ws_))
    e_small = rel_l2(out, xs.double() @ ws_.double().T)
    print("small-magnitude f16x3", e_small)
    assert e_small < 2e-4
    # RoPE epilogue on the split kernel
    Bt, T, H = 2, 130, 2
    xq = randn(Bt * T, 128, seed=115)
    wq = randn(3 * H * 64, 128, seed=116) / math.sqrt(128)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(T).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().to(dev()).contiguous(), ang.sin().to(dev()).contiguous()
    o1 = torch.empty(Bt * T, 3 * H * 64, device=dev()); o2 = torch.empty_like(o1)
    ops.gemm(xq, wq, o1, rope=(cos, sin), rope_cols=2 * H * 64, w_split=ops.split_f16(wq))
    ops.gemm(xq, wq, o2, rope=(cos, sin), rope_cols=2 * H * 64)
    assert rel_l2(o1, o2) < 3e-6


@pytest.mark.parametrize("M,N,K", [(256, 1024, 1024), (1000, 3072, 1024), (999, 1024, 4096), (130, 256, 64)])
def test_gemm_f16x3_presplit_dma_and_split_output(ops, M, N, K):
    """A pre-split (all-DMA multi-stage kernel) and split outputs must agree with the on-the-fly split kernel."""
    a, w = randn(M, K, seed=120) * 2.0, randn(N, K, seed=121) / math.sqrt(K)
    b, r = randn(N, seed=122), randn(M, N, seed=123)
    ws = ops.split_f16(w)
    ref = F.gelu(a.double() @ w.double().T + b.double()) + r.double()
    o1 = torch.empty(M, N, device=dev())
    ops.gemm(a, w, o1, bias=b, act=1, residual=r, w_split=ws)
    asp = ops.split_act_f16(a)
    o2 = torch.full((M, N), float("nan"), device=dev())
    oh = torch.empty(M, N, dtype=torch.float16, device=dev()); ol = torch.empty_like(oh)
    ops.gemm(a, w, o2, bias=b, act=1, residual=r, w_split=ws, a_split=asp, out_split=(oh, ol))
    assert rel_l2(o2, ref) < 3e-6 and rel_l2(o2, o1) < 1e-6
    assert rel_l2(oh.float() + ol.float(), o2) < 1e-6
    o3 = torch.full((M, N), 7.0, device=dev())
    ops.gemm(a, w, o3, bias=b, act=1, residual=r, w_split=ws, a_split=asp, out_split=(oh, ol), write_f32=False)
    assert bool((o3 == 7.0).all())                      # fp32 store suppressed
    assert rel_l2(oh.float() + ol.float(), ref) < 3e-6


def test_gemm_f16x3_presplit_concat(ops):
    M, N = 300, 256
    x, s = randn(M, 256, seed=130), randn(M, 256, seed=131)
    w = randn(N, 512, seed=132) / math.sqrt(512)
    out = torch.empty(M, N, device=dev())
    ops.gemm(x, w, out, a2=s, w_split=ops.split_f16(w), a_split=ops.split_act_f16(x), a2_split=ops.split_act_f16(s))
    assert rel_l2(out, torch.cat((x, s), -1).double() @ w.double().T) < 3e-6


@pytest.mark.parametrize("Bt,T,H", [(2, 128, 2), (1, 36, 1), (3, 200, 2), (2, 1000, 1), (1, 132, 3)])
def test_attention_f16x3_with_qkv_transposed_epilogue(ops, Bt, T, H):
    """to_qkv GEMM in QKV mode (q|k split, v split + transposed) -> split-precision attention, vs fp64."""
    dim = 128
    x = randn(Bt * T, dim, seed=140)
    w = randn(3 * H * 64, dim, seed=141) / math.sqrt(dim) * 1.5
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(T).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().to(dev()).contiguous(), ang.sin().to(dev()).contiguous()
    M = Bt * T
    qk = (torch.empty(M, 2 * H * 64, dtype=torch.float16, device=dev()), torch.empty(M, 2 * H * 64, dtype=torch.float16, device=dev()))
    Tp = ((T + 31) // 32) * 32
    vt = (torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev()), torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev()))
    dummy = torch.empty(M, 3 * H * 64, device=dev())
    ops.gemm(x, w, dummy, rope=(cos, sin), rope_cols=2 * H * 64, w_split=ops.split_f16(w), a_split=ops.split_act_f16(x),
             out_split=qk, vt_split=vt, write_f32=False)
    ref_qkv = torch.empty(M, 3 * H * 64, device=dev())
    ops.gemm(x, w, ref_qkv, rope=(cos, sin), rope_cols=2 * H * 64)
    assert rel_l2(qk[0].float() + qk[1].float(), ref_qkv[:, : 2 * H * 64]) < 2e-6
    v_ref = ref_qkv[:, 2 * H * 64:].reshape(Bt, T, H, 64).permute(0, 2, 3, 1).reshape(Bt * H * 64, T)
    slots = ops.vt_frame_slots(T, dev())                      # frame t lives in column slot(t)
    free = torch.ones(Tp, dtype=torch.bool, device=dev()); free[slots] = False
    assert rel_l2((vt[0].float() + vt[1].float())[:, slots], v_ref) < 2e-6
    assert bool(((vt[0].float() + vt[1].float())[:, free] == 0).all())
    out = torch.full((Bt, T, H * 64), float("nan"), device=dev())
    oh = torch.empty(Bt, T, H * 64, dtype=torch.float16, device=dev()); ol = torch.empty_like(oh)
    ops.attention_f16x3(qk, vt, out, Bt, T, H, 0.125, out_split=(oh, ol))
    q, k, v = ref_qkv.double().reshape(Bt, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(Bt, T, H * 64)
    e = rel_l2(out, ref)
    print("attention f16x3", Bt, T, H, e)
    assert e < 5e-6 and rel_l2(oh.float() + ol.float(), out) < 1e-6


def test_attention_f16x3_forced_rescale_and_tail(ops):
    """Late dominant key (forces the rescale branch after tiles where the max did not move) + a ragged last tile."""
    Bt, T, H = 1, 164, 1
    qkv = randn(Bt * T, 192, seed=150) * 0.5
    qkv[5, :64] = 3.0
    qkv[140, 64:128] = 4.0
    qkv[163, 64:128] = -4.0
    M = Bt * T
    ah, al = ops.split_act_f16(qkv.contiguous())
    qk = (ah[:, :128].contiguous(), al[:, :128].contiguous())
    Tp = 192
    vt = (torch.zeros(64, Tp, dtype=torch.float16, device=dev()), torch.zeros(64, Tp, dtype=torch.float16, device=dev()))
    slots = ops.vt_frame_slots(T, dev())
    vt[0][:, slots] = ah[:, 128:].T
    vt[1][:, slots] = al[:, 128:].T
    out = torch.empty(Bt, T, 64, device=dev())
    ops.attention_f16x3(qk, vt, out, Bt, T, H, 0.125)
    q, k, v = qkv.double().reshape(Bt, T, 3, 1, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(Bt, T, 64)
    assert rel_l2(out, ref) < 5e-6


# ---------------------------------------------------------------- single-term fp16 mode (precision="f16", opt-in)
F16_TOL = 1e-3        # plain fp16 operands (11-bit significand), fp32 accumulate: measured 2-4e-4


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 1024), (2500, 512, 4096), (300, 256, 512), (999, 80, 1024)])
def test_gemm_f16_single_term(ops, M, N, K):
    """W_lo == NULL: hi halves only, K consumed 64 per stage (256x256 tile for big shapes, 128x128 otherwise).
    The result must equal the product of the fp16-ROUNDED operands to fp32 accuracy, and the unrounded product
    to fp16 accuracy."""
    a, w = randn(M, K, seed=200) * 2.0, randn(N, K, seed=201) / math.sqrt(K)
    b, r = randn(N, seed=202), randn(M, N, seed=203)
    wh, wl, inv = ops.split_f16(w, with_lo=False)
    assert wl is None
    ah = torch.empty(M, K, dtype=torch.float16, device=dev())
    ops.split_act_f16(a, ah, None)
    assert torch.equal(ah, a.half())
    out = torch.full((M, N), float("nan"), device=dev())
    oh = torch.empty(M, N, dtype=torch.float16, device=dev())
    ops.gemm(a, w, out, bias=b, act=1, residual=r, w_split=(wh, None, inv), a_split=(ah, None), out_split=(oh, None))
    rounded = F.gelu(ah.double() @ (wh.double() * inv).T + b.double()) + r.double()
    exact = F.gelu(a.double() @ w.double().T + b.double()) + r.double()
    assert rel_l2(out, rounded) < 3e-6
    e = rel_l2(out, exact)
    print("f16 single-term gemm", (M, N, K), e)
    assert e < F16_TOL
    assert torch.equal(oh, out.half())


def test_gemm_f16_single_term_concat(ops):
    M, N = 2304, 512
    x, s = randn(M, 256, seed=210), randn(M, 192, seed=211)
    w = randn(N, 448, seed=212) / math.sqrt(448)
    wh, _, inv = ops.split_f16(w, with_lo=False)
    out = torch.empty(M, N, device=dev())
    ops.gemm(x, w, out, a2=s, w_split=(wh, None, inv), a_split=(x.half(), None), a2_split=(s.half(), None))
    ref = torch.cat((x.half(), s.half()), -1).double() @ (wh.double() * inv).T
    assert rel_l2(out, ref) < 3e-6


def test_gemm_f16_single_term_rejects_bad_k(ops):
    import covomix_amd._lib as L
    a, w = randn(128, 96), randn(128, 96)
    wh, _, inv = ops.split_f16(w, with_lo=False)
    with pytest.raises(L.CovomixHipError):              # K = 96 is not a multiple of 64
        ops.gemm(a, w, torch.empty(128, 128, device=dev()), w_split=(wh, None, inv), a_split=(a.half(), None))


@pytest.mark.parametrize("Bt,T,H", [(2, 128, 2), (1, 36, 1), (2, 1000, 1)])
def test_attention_f16_single_term(ops, Bt, T, H):
    """QKV GEMM (single-term, q|k hi + v^T hi) -> single-term attention, against fp64 attention."""
    dim = 128
    x = randn(Bt * T, dim, seed=220)
    w = randn(3 * H * 64, dim, seed=221) / math.sqrt(dim)
    inv_f = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(T).float()[:, None] * inv_f[None, :]
    cos, sin = ang.cos().to(dev()).contiguous(), ang.sin().to(dev()).contiguous()
    Tp = ((T + 31) // 32) * 32
    qk = (torch.empty(Bt * T, 2 * H * 64, dtype=torch.float16, device=dev()), None)
    vt = (torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev()), None)
    wh, _, inv = ops.split_f16(w, with_lo=False)
    dummy = torch.empty(Bt * T, 3 * H * 64, device=dev())
    ops.gemm(x, w, dummy, rope=(cos, sin), rope_cols=2 * H * 64, w_split=(wh, None, inv), a_split=(x.half(), None),
             out_split=qk, vt_split=vt, write_f32=False)
    ref_qkv = torch.empty(Bt * T, 3 * H * 64, device=dev())
    ops.gemm(x, w, ref_qkv, rope=(cos, sin), rope_cols=2 * H * 64)
    q, k, v = [t.reshape(Bt, T, H, 64).permute(0, 2, 1, 3).double() for t in ref_qkv.reshape(Bt, T, 3, H * 64).unbind(2)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(Bt * T, H * 64)
    out = torch.empty(Bt * T, H * 64, device=dev())
    oh = torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev())
    ops.attention_f16x3(qk, vt, out, Bt, T, H, 0.125, out_split=(oh, None))
    e = rel_l2(out, ref)
    print("f16 single-term attention", (Bt, T, H), e)
    assert e < 2e-3
    assert torch.equal(oh, out.half())


# ---------------------------------------------------------------- HiFi-GAN ResBlock convolution on the split-fp16 pipe
@pytest.mark.parametrize("C,k,dil,L,B", [(250, 11, 5, 700, 2), (125, 7, 3, 1030, 1), (62, 3, 1, 2049, 2), (31, 11, 1, 515, 3),
                                         (16, 7, 5, 256, 1), (250, 3, 1, 255, 1)])
def test_hifigan_conv1d_f16x3_vs_torch(ops, C, k, dil, L, B):
    """cvx_hifigan_conv1d_f16x3 (channels-last, zero halos) against torch conv1d in fp64 for every tile configuration
    (Np = 256 / 128 / 64 / 32), ragged L, all epilogue outputs: new residual stream, running accumulate and scale, and the
    split leaky_relu copy for the next convolution; padding rows / channels must stay zero."""
    g = torch.Generator().manual_seed(300 + C + k)
    x = torch.randn(B, C, L, generator=g).to(dev())
    w = (torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).to(dev())
    b = (torch.randn(C, generator=g) * 0.1).to(dev())
    res = torch.randn(B, C, L, generator=g).to(dev())
    acc = torch.randn(B, C, L, generator=g).to(dev())
    wpk = ops.hifigan_pack_weight_f16x3(w)
    Np, Lp = wpk[3], ops.hifigan_cl_rows(L)
    bias = torch.zeros(Np, device=dev()); bias[:C] = b
    z = (torch.zeros(B, Lp, Np, dtype=torch.float16, device=dev()), torch.zeros(B, Lp, Np, dtype=torch.float16, device=dev()))
    f32 = lambda: torch.zeros(B, Lp, Np, device=dev())
    res_cl, acc_cl, out_x = f32(), f32(), f32()
    out_z = (torch.zeros_like(z[0]), torch.zeros_like(z[0]))
    ops.hifigan_to_channels_last(x, None, z, 0.1)                       # z = split(leaky_relu(x))
    ops.hifigan_to_channels_last(res, res_cl, None, 1.0)
    ops.hifigan_to_channels_last(acc, acc_cl, None, 1.0)
    ops.hifigan_conv1d_f16x3(z, wpk, bias, B, L, ksize=k, dil=dil, res=res_cl, accum=acc_cl, out_x=out_x, out_scale=1.0 / 3,
                             out_z=out_z, z_slope=0.1)
    pad = (k * dil - dil) // 2
    v = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=dil, padding=pad) + res.double()
    want_x = (v + acc.double()) / 3
    want_z = F.leaky_relu(v, 0.1)
    got = torch.empty(B, C, L, device=dev())
    ops.hifigan_from_channels_last(out_x, got)
    assert rel_l2(got, want_x) < 2e-6
    H = ops.HIFI_HALO_L
    zsum = (out_z[0].float() + out_z[1].float())
    assert rel_l2(zsum[:, H:H + L, :C].transpose(1, 2), want_z) < 2e-6
    for t in (out_x, zsum):                                             # halos and padded channels stay zero
        assert float(t[:, :H].abs().max()) == 0 and float(t[:, H + L:].abs().max()) == 0
        if Np > C:
            assert float(t[:, :, C:].abs().max()) == 0


def test_hifigan_conv1d_f16x3_rejects_bad_args(ops):
    import covomix_amd._lib as L
    w = torch.randn(32, 32, 4, device=dev())                            # (k-1)*dil odd: not a "same" convolution
    wpk = ops.hifigan_pack_weight_f16x3(w)
    z = (torch.zeros(1, ops.hifigan_cl_rows(64), 32, dtype=torch.float16, device=dev()),) * 2
    with pytest.raises(L.CovomixHipError):
        ops.hifigan_conv1d_f16x3(z, wpk, torch.zeros(32, device=dev()), 1, 64, ksize=4, dil=1, out_x=torch.zeros(1, z[0].shape[1], 32, device=dev()))


# ---------------------------------------------------------------- split-K path of the small-problem GEMM
@pytest.mark.parametrize("terms", [3, 1])
@pytest.mark.parametrize("M,N,K,K1", [(1000, 1024, 4096, 0), (1000, 1024, 2048, 1024), (500, 1024, 1024, 0), (77, 256, 2048, 0)])
def test_gemm_split_k_small_problem(ops, M, N, K, K1, terms):
    """Few output tiles + long K: K is cut into 2-4 slices on separate blocks, partials are added in a fixed order and the
    epilogue (bias, GELU, residual in place, split output) runs in the reduce kernel.  Against fp64, deterministic."""
    g = torch.Generator().manual_seed(400 + M + K)
    a = torch.randn(M, K, generator=g).to(dev())
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    r = torch.randn(M, N, generator=g).to(dev())
    ws = ops.split_f16(w, with_lo=(terms == 3))
    ah, al = ops.split_act_f16(a)
    if terms == 1:
        al = None
    kw = {}
    if K1:
        kw = dict(a2=a[:, K1:].contiguous(), a2_split=(ah[:, K1:].contiguous(), None if al is None else al[:, K1:].contiguous()))
        a_in, asp = a[:, :K1].contiguous(), (ah[:, :K1].contiguous(), None if al is None else al[:, :K1].contiguous())
    else:
        a_in, asp = a, (ah, al)
    src = ah.double() + (al.double() if al is not None else 0)
    wd = (ws[0].double() + (ws[1].double() if ws[1] is not None else 0)) * ws[2]
    ref = F.gelu(src @ wd.T + b.double()) + r.double()
    outs = []
    for _ in range(2):
        c = r.clone()                                                   # residual in place
        oh = torch.empty(M, N, dtype=torch.float16, device=dev())
        ol = torch.empty_like(oh) if terms == 3 else None
        ops.gemm(a_in, w, c, bias=b, act=1, residual=c, w_split=ws, a_split=asp, out_split=(oh, ol), **kw)
        outs.append(c)
        assert rel_l2(c, ref) < 3e-6
        got = oh.float() + (ol.float() if ol is not None else 0)
        assert rel_l2(got, c) < (1e-6 if terms == 3 else 1e-3)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,K1", [(2048, 512, 1024, 0), (4100, 1024, 4096, 0), (2300, 3072, 1024, 0), (2048, 1024, 2048, 1024)])
def test_gemm_f16x3_interleaved_weights(ops, M, N, K, K1):
    """Large-problem kernel with the weight stored interleaved ([hi 32 | lo 32] per K-step): must equal the separate
    hi / lo layout bit for bit (same products in the same order) and fp64 to fp32 accuracy."""
    g = torch.Generator().manual_seed(500 + N + K)
    a = torch.randn(M, K, generator=g).to(dev())
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    ws = ops.split_f16(w)
    il = ops.split_f16_interleaved(ws)
    assert il[0].shape == (N, 2 * K)
    assert torch.equal(il[0][:, 64:96], ws[0][:, 32:64]) and torch.equal(il[0][:, 32:64], ws[1][:, :32])
    ah, al = ops.split_act_f16(a)
    kw = {}
    if K1:
        kw = dict(a2=a[:, K1:].contiguous(), a2_split=(ah[:, K1:].contiguous(), al[:, K1:].contiguous()))
        a_in, asp = a[:, :K1].contiguous(), (ah[:, :K1].contiguous(), al[:, :K1].contiguous())
    else:
        a_in, asp = a, (ah, al)
    c0, c1 = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev())
    ops.gemm(a_in, w, c0, bias=b, w_split=ws, a_split=asp, **kw)
    ops.gemm(a_in, w, c1, bias=b, w_split=ws, a_split=asp, w_il=il, **kw)
    assert torch.equal(c0, c1)
    assert rel_l2(c1, a.double() @ w.double().T + b.double()) < 3e-6


def test_interleaved_activation_pairs_end_to_end(ops):
    """SplitIL (one buffer, [hi 32 | lo 32] per 32 columns) through every producer and the large-problem GEMM: each
    producer must write exactly the values of the two-tensor form, and the GEMM must give bit-identical results."""
    M, D, H = 2100, 1024, 16
    g = torch.Generator().manual_seed(77)
    x = torch.randn(M, D, generator=g).to(dev())
    gam, bet = torch.randn(D, generator=g).to(dev()), torch.randn(D, generator=g).to(dev())
    # norm
    sep = (torch.empty(M, D, dtype=torch.float16, device=dev()), torch.empty(M, D, dtype=torch.float16, device=dev()))
    il = ops.SplitIL(M, D, dev())
    ops.adarmsnorm(x, gam, bet, None, out_split=sep)
    ops.adarmsnorm(x, gam, bet, None, out_split=il)
    assert all(torch.equal(a, b) for a, b in zip(il.dense(), sep))
    # plain split
    il2 = ops.SplitIL(M, D, dev())
    ops.split_act_f16(x, il2)
    assert all(torch.equal(a, b) for a, b in zip(il2.dense(), ops.split_act_f16(x)))
    # GEMM: interleaved A (and A2) + interleaved W + interleaved split output, against the separate form
    w = (torch.randn(1024, 2 * D, generator=g) / 45).to(dev())
    ws = ops.split_f16(w)
    wil = ops.split_f16_interleaved(ws)
    b = torch.randn(1024, generator=g).to(dev())
    c0, c1 = torch.empty(M, 1024, device=dev()), torch.empty(M, 1024, device=dev())
    o_sep = (torch.empty(M, 1024, dtype=torch.float16, device=dev()), torch.empty(M, 1024, dtype=torch.float16, device=dev()))
    o_il = ops.SplitIL(M, 1024, dev())
    ops.gemm(x, w, c0, bias=b, act=1, a2=x, w_split=ws, w_il=wil, a_split=sep, a2_split=ops.split_act_f16(x), out_split=o_sep)
    ops.gemm(x, w, c1, bias=b, act=1, a2=x, w_split=ws, w_il=wil, a_split=il, a2_split=il2, out_split=o_il)
    # (interleaved A runs the eight-phase kernel: another summation order and erf evaluation - fp32-rounding agreement)
    assert rel_l2(c0, c1) < 1e-6
    assert rel_l2(o_il.dense()[0].float() + o_il.dense()[1].float(), c1) < 1e-6
    assert rel_l2(o_sep[0].float() + o_sep[1].float(), c0) < 1e-6
    # attention output (both attention kernels)
    Bt, T = 2, 1050
    qkv = torch.randn(Bt, T, 3 * H * 64, generator=g).to(dev()) * 0.3
    a_sep = (torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev()), torch.empty(Bt * T, H * 64, dtype=torch.float16, device=dev()))
    a_il = ops.SplitIL(Bt * T, H * 64, dev())
    ops.attention(qkv, None, Bt, T, H, 0.125, out_split=a_sep)
    ops.attention(qkv, None, Bt, T, H, 0.125, out_split=a_il)
    assert all(torch.equal(a, b_) for a, b_ in zip(a_il.dense(), a_sep))
    qh, ql = ops.split_act_f16(qkv.reshape(Bt * T, -1)[:, : 2 * H * 64].contiguous())
    Tp = (T + 31) // 32 * 32
    vt = (torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev()), torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev()))
    vt[0][:, :T] = torch.randn(Bt * H * 64, T, generator=g).half().to(dev())
    ops.attention_f16x3((qh, ql), vt, None, Bt, T, H, 0.125, out_split=a_sep)
    ops.attention_f16x3((qh, ql), vt, None, Bt, T, H, 0.125, out_split=a_il)
    assert all(torch.equal(a, b_) for a, b_ in zip(a_il.dense(), a_sep))
    # only the large-problem kernel can read an interleaved A
    with pytest.raises(AssertionError):
        ops.gemm(x[:100], w[:, :D], torch.empty(100, 1024, device=dev()), w_split=ops.split_f16(w[:, :D].contiguous()), a_split=ops.SplitIL(100, D, dev()))


@pytest.mark.parametrize("M", [2100, 2304])
def test_gemm_large_problem_kernels_every_epilogue(ops, M):
    """The large-problem kernel behind cvx_gemm_f16x3 for interleaved operands (eight-phase, 16x16x32 MFMA; pinned with flag 16)
    and the library's own choice at this size (the medium-problem kernel) on every epilogue class of the transformer block,
    against fp64: QKV (RoPE + split q|k + transposed split v, T % 4 == 0 and != 0), residual (+ bias, + split twin),
    bias + GELU + split, K-split (skip combiner) + bias, plain; M = 2100 leaves a ragged last row panel."""
    dev_ = dev()
    g = torch.Generator().manual_seed(M)
    K, H = 1024, 4
    x = torch.randn(M, K, generator=g).to(dev_)
    il = ops.SplitIL(M, K, dev_); ops.split_act_f16(x, il)
    xs = il.dense()[0].double() + il.dense()[1].double()

    def weights(N, Kw=K):
        w = (torch.randn(N, Kw, generator=g) / math.sqrt(Kw)).to(dev_)
        ws = ops.split_f16(w)
        return w, ws, ops.split_f16_interleaved(ws)
    saved = ops._GEMM_FLAGS
    try:
        for flags in (16, 0):                    # large-problem kernel pinned / the library's choice (medium kernel here)
            ops._GEMM_FLAGS = flags
            # plain + bias / residual / twin
            w, ws, wil = weights(1024)
            b, r = torch.randn(1024, generator=g).to(dev_), torch.randn(M, 1024, generator=g).to(dev_)
            ref = xs @ w.double().T
            c = torch.full((M, 1024), float("nan"), device=dev_)
            ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il)
            assert rel_l2(c, ref) < 1e-6, flags
            ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il, residual=r)
            assert rel_l2(c, ref + r.double()) < 1e-6, flags
            tw = ops.SplitIL(M, 1024, dev_)
            ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il, bias=b, residual=r, out_split=tw)
            want = ref + b.double() + r.double()
            assert rel_l2(c, want) < 1e-6 and rel_l2(tw.dense()[0].double() + tw.dense()[1].double(), want) < 1e-6, flags
            # bias + GELU + split only
            w, ws, wil = weights(2048)
            b = torch.randn(2048, generator=g).to(dev_)
            o = ops.SplitIL(M, 2048, dev_)
            guard = torch.full((M, 2048), 7.0, device=dev_)
            ops.gemm(x, w, guard, w_split=ws, w_il=wil, a_split=il, bias=b, act=1, out_split=o, write_f32=False)
            want = F.gelu(xs @ w.double().T + b.double())
            assert rel_l2(o.dense()[0].double() + o.dense()[1].double(), want) < 1e-6 and bool((guard == 7.0).all()), flags
            # K-split (A | A2) + bias
            w, ws, wil = weights(1024, 2 * K)
            b = torch.randn(1024, generator=g).to(dev_)
            ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il, a2=x, a2_split=il, bias=b)
            assert rel_l2(c, torch.cat((xs, xs), 1) @ w.double().T + b.double()) < 1e-6, flags
            # QKV: rope on q|k, split q|k, transposed split v
            for T in (M // 4, M // 3):
                Bt = M // T
                Mq = Bt * T
                w, ws, wil = weights(3 * H * 64)
                inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
                ang = torch.arange(T).float()[:, None] * inv[None, :]
                cos, sin = ang.cos().to(dev_).contiguous(), ang.sin().to(dev_).contiguous()
                xq = x[:Mq].contiguous()
                ilq = ops.SplitIL(Mq, K, dev_); ops.split_act_f16(xq, ilq)
                qk = (torch.empty(Mq, 2 * H * 64, dtype=torch.float16, device=dev_), torch.empty(Mq, 2 * H * 64, dtype=torch.float16, device=dev_))
                Tp = (T + 31) // 32 * 32
                vt = (torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev_), torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev_))
                dummy = torch.empty(Mq, 3 * H * 64, device=dev_)
                ops.gemm(xq, w, dummy, w_split=ws, w_il=wil, a_split=ilq, rope=(cos, sin), rope_cols=2 * H * 64, out_split=qk,
                         vt_split=vt, write_f32=False)
                z = (ilq.dense()[0].double() + ilq.dense()[1].double()) @ w.double().T
                zq = z[:, : 2 * H * 64].reshape(Bt, T, 2 * H, 64)
                c_, s_ = torch.cat((ang.cos(), ang.cos()), -1).double().to(dev_), torch.cat((ang.sin(), ang.sin()), -1).double().to(dev_)
                rot = torch.cat((-zq[..., 32:], zq[..., :32]), -1)
                want_qk = (zq * c_[None, :, None, :] + rot * s_[None, :, None, :]).reshape(Mq, -1)
                assert rel_l2(qk[0].double() + qk[1].double(), want_qk) < 1e-6, (flags, T)
                v = z[:, 2 * H * 64:].reshape(Bt, T, H, 64).permute(0, 2, 3, 1).reshape(Bt * H * 64, T)
                slots = ops.vt_frame_slots(T, dev_)
                got_v = (vt[0].double() + vt[1].double())[:, slots]
                assert rel_l2(got_v, v) < 1e-6, (flags, T)
                free = torch.ones(Tp, dtype=torch.bool, device=dev_); free[slots] = False      # columns no frame maps to stay zero
                assert float(vt[0][:, free].abs().max() if bool(free.any()) else 0) == 0
    finally:
        ops._GEMM_FLAGS = saved


def test_gemm_large_problem_kernels_n_not_multiple_of_256(ops):
    """N % 64 == 0 but N % 256 != 0 on the 256-column-tile kernels (N = 576, 640; QKV with H = 5 heads): the trailing wave
    tiles lie past column N and must neither read bias / residual / RoPE tables nor write C, the split pair or V^T there.
    Outputs live inside wider NaN-filled buffers (ldc > N), so a stray store shows up as a finite value in the guard."""
    dev_ = dev()
    g = torch.Generator().manual_seed(99)
    M, K = 2100, 1024
    x = torch.randn(M, K, generator=g).to(dev_)
    il = ops.SplitIL(M, K, dev_); ops.split_act_f16(x, il)
    xs = il.dense()[0].double() + il.dense()[1].double()
    saved = ops._GEMM_FLAGS
    try:
        for flags in (16, 0):
            ops._GEMM_FLAGS = flags
            for N in (576, 640):
                w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev_)
                ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
                b = torch.randn(N, generator=g).to(dev_)
                rwide = torch.randn(M, N + 256, generator=g).to(dev_)
                ref = xs @ w.double().T
                cw = torch.full((M + 1, N + 256), float("nan"), device=dev_)         # guard columns and one guard row
                ops.gemm(x, w, cw[:M, :N], w_split=ws, w_il=wil, a_split=il, bias=b, residual=rwide[:, :N])
                assert rel_l2(cw[:M, :N], ref + b.double() + rwide[:, :N].double()) < 1e-6, (flags, N)
                assert bool(torch.isnan(cw[:, N:]).all()) and bool(torch.isnan(cw[M]).all()), (flags, N)
                # bias + GELU + split only (no fp32 store): pair inside a wider NaN-filled pair
                oh = torch.full((M + 1, N + 256), float("nan"), dtype=torch.float16, device=dev_)
                ol = torch.full((M + 1, N + 256), float("nan"), dtype=torch.float16, device=dev_)
                guard = torch.full((M, N), 7.0, device=dev_)
                ops.gemm(x, w, guard, w_split=ws, w_il=wil, a_split=il, bias=b, act=1, out_split=(oh[:M, :N], ol[:M, :N]), write_f32=False)
                assert rel_l2(oh[:M, :N].double() + ol[:M, :N].double(), F.gelu(ref + b.double())) < 1e-6, (flags, N)
                assert bool(torch.isnan(oh[:, N:]).all()) and bool(torch.isnan(ol[:, N:]).all()) and bool(torch.isnan(oh[M]).all()), (flags, N)
                assert bool((guard == 7.0).all())
            # QKV with 5 heads: N = 960, rope_cols = 640 (the large kernel declines - RoPE groups of 256 columns; the medium kernel takes it)
            H, T = 5, 700
            Bt = M // T
            Mq = Bt * T
            wq = (torch.randn(3 * H * 64, K, generator=g) / math.sqrt(K)).to(dev_)
            wqs = ops.split_f16(wq); wqil = ops.split_f16_interleaved(wqs)
            inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
            ang = torch.arange(T).float()[:, None] * inv[None, :]
            cos, sin = ang.cos().to(dev_).contiguous(), ang.sin().to(dev_).contiguous()
            xq = x[:Mq].contiguous()
            ilq = ops.SplitIL(Mq, K, dev_); ops.split_act_f16(xq, ilq)
            qh = torch.full((Mq + 1, 2 * H * 64 + 128), float("nan"), dtype=torch.float16, device=dev_)
            ql = torch.full((Mq + 1, 2 * H * 64 + 128), float("nan"), dtype=torch.float16, device=dev_)
            Tp = (T + 31) // 32 * 32
            rows = Bt * H * 64
            vh = torch.zeros(rows + 64, Tp, dtype=torch.float16, device=dev_); vl = torch.zeros(rows + 64, Tp, dtype=torch.float16, device=dev_)
            vh[rows:] = float("nan"); vl[rows:] = float("nan")                       # guard: the rows a sixth head of the last sequence would own
            ops.gemm(xq, wq, torch.empty(Mq, 3 * H * 64, device=dev_), w_split=wqs, w_il=wqil, a_split=ilq, rope=(cos, sin),
                     rope_cols=2 * H * 64, out_split=(qh[:Mq, : 2 * H * 64], ql[:Mq, : 2 * H * 64]), vt_split=(vh[:rows], vl[:rows]), write_f32=False)
            z = (ilq.dense()[0].double() + ilq.dense()[1].double()) @ wq.double().T
            zq = z[:, : 2 * H * 64].reshape(Bt, T, 2 * H, 64)
            c_, s_ = torch.cat((ang.cos(), ang.cos()), -1).double().to(dev_), torch.cat((ang.sin(), ang.sin()), -1).double().to(dev_)
            rot = torch.cat((-zq[..., 32:], zq[..., :32]), -1)
            want_qk = (zq * c_[None, :, None, :] + rot * s_[None, :, None, :]).reshape(Mq, -1)
            assert rel_l2(qh[:Mq, : 2 * H * 64].double() + ql[:Mq, : 2 * H * 64].double(), want_qk) < 1e-6, flags
            assert bool(torch.isnan(qh[:, 2 * H * 64:]).all()) and bool(torch.isnan(qh[Mq]).all()), flags
            v = z[:, 2 * H * 64:].reshape(Bt, T, H, 64).permute(0, 2, 3, 1).reshape(rows, T)
            slots = ops.vt_frame_slots(T, dev_)
            assert rel_l2((vh[:rows].double() + vl[:rows].double())[:, slots], v) < 1e-6, flags
            assert bool(torch.isnan(vh[rows:]).all()) and bool(torch.isnan(vl[rows:]).all()), flags
    finally:
        ops._GEMM_FLAGS = saved


def test_gemm_persistent_blocks_walk_several_tiles(ops):
    """More output tiles than CUs: a block of the eight-phase kernel then walks several tiles and fetches the first quarters
    of the next tile during the tail of the current one.  M = 5000 rows (20 row panels -> 24 slots per tile column on the XCD
    map, padding slots included) x N = 4096 = 384 slots: QKV-style (V blocks included), GELU-split and residual epilogues
    against fp64, and bit-identity with one tile per block (CVX_GEMM_FLAG_ONE_TILE)."""
    dev_ = dev()
    g = torch.Generator().manual_seed(4)
    M, K = 5000, 1024
    x = torch.randn(M, K, generator=g).to(dev_)
    il = ops.SplitIL(M, K, dev_); ops.split_act_f16(x, il)
    xs = il.dense()[0].double() + il.dense()[1].double()
    w = (torch.randn(4096, K, generator=g) / math.sqrt(K)).to(dev_)
    ws = ops.split_f16(w); wil = ops.split_f16_interleaved(ws)
    b = torch.randn(4096, generator=g).to(dev_)
    saved = ops._GEMM_FLAGS
    try:
        outs = []
        for flags in (16, 16 | 4):               # (16 = CVX_GEMM_FLAG_NO_MEDIUM: 5000 rows would otherwise go to the medium-problem kernel)
            ops._GEMM_FLAGS = flags
            o = ops.SplitIL(M, 4096, dev_)
            ops.gemm(x, w, torch.empty(M, 4096, device=dev_), w_split=ws, w_il=wil, a_split=il, bias=b, act=1, out_split=o, write_f32=False)
            outs.append(o.buf.clone())
        want = F.gelu(xs @ w.double().T + b.double())
        got = outs[0].view(M, 128, 2, 32)
        assert rel_l2(got[:, :, 0].reshape(M, 4096).double() + got[:, :, 1].reshape(M, 4096).double(), want) < 1e-6
        assert torch.equal(outs[0], outs[1])
        # residual + fp32 on a narrower N (4 tile columns x 24 slots = 96 slots: one tile per block) and a wide one
        r = torch.randn(M, 4096, generator=g).to(dev_)
        c = torch.full((M, 4096), float("nan"), device=dev_)
        ops._GEMM_FLAGS = 16
        ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il, residual=r)
        assert rel_l2(c, xs @ w.double().T + r.double()) < 1e-6
        # QKV form with 16 heads: 12 tile columns x 24 slots = 288 slots, V blocks interleaved with q | k blocks in a block's walk
        H, T = 16, 1000
        wq = (torch.randn(3 * H * 64, K, generator=g) / math.sqrt(K)).to(dev_)
        wqs = ops.split_f16(wq); wqil = ops.split_f16_interleaved(wqs)
        inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
        ang = torch.arange(T).float()[:, None] * inv[None, :]
        cos, sin = ang.cos().to(dev_).contiguous(), ang.sin().to(dev_).contiguous()
        Bt = M // T
        qk = (torch.empty(M, 2 * H * 64, dtype=torch.float16, device=dev_), torch.empty(M, 2 * H * 64, dtype=torch.float16, device=dev_))
        Tp = (T + 31) // 32 * 32
        vt = (torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev_), torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev_))
        ops.gemm(x, wq, torch.empty(M, 3 * H * 64, device=dev_), w_split=wqs, w_il=wqil, a_split=il, rope=(cos, sin), rope_cols=2 * H * 64,
                 out_split=qk, vt_split=vt, write_f32=False)
        z = xs @ wq.double().T
        zq = z[:, : 2 * H * 64].reshape(Bt, T, 2 * H, 64)
        c_, s_ = torch.cat((ang.cos(), ang.cos()), -1).double().to(dev_), torch.cat((ang.sin(), ang.sin()), -1).double().to(dev_)
        rot = torch.cat((-zq[..., 32:], zq[..., :32]), -1)
        assert rel_l2(qk[0].double() + qk[1].double(), (zq * c_[None, :, None, :] + rot * s_[None, :, None, :]).reshape(M, -1)) < 1e-6
        v = z[:, 2 * H * 64:].reshape(Bt, T, H, 64).permute(0, 2, 3, 1).reshape(Bt * H * 64, T)
        assert rel_l2((vt[0].double() + vt[1].double())[:, ops.vt_frame_slots(T, dev_)], v) < 1e-6
    finally:
        ops._GEMM_FLAGS = saved


@pytest.mark.parametrize("M", [2100, 4096])
def test_gemm_deferred_norm_producer_and_consumers(ops, M):
    """Deferred AdaptiveRMSNorm (cvx_gemm_split_io version 105; reference acoustic.py:198-204 between :306-318's products):
    producer forms (residual + fp32 + twin * gamma + row sums of squares; A | A2 + bias + the same), cvx_rownorm_scale_f32, and the
    consumer forms (bias + GELU + split with a factor per row; QKV with a bias and that factor on q | k | v) - each against fp64, and
    the chain producer -> factor -> consumer against  norm(x) @ W.T  computed the reference's way."""
    dev_ = dev()
    g = torch.Generator().manual_seed(77 + M)
    K, D, H = 1024, 1024, 4
    x = torch.randn(M, K, generator=g).to(dev_)
    il = ops.SplitIL(M, K, dev_); ops.split_act_f16(x, il)
    xs = il.dense()[0].double() + il.dense()[1].double()

    def weights(N, Kw=K):
        w = (torch.randn(N, Kw, generator=g) / math.sqrt(Kw)).to(dev_)
        ws = ops.split_f16(w)
        return w, ws, ops.split_f16_interleaved(ws)
    pair = lambda t: t.dense()[0].double() + t.dense()[1].double()
    gamma = (1.0 + 0.3 * torch.randn(D, generator=g)).to(dev_)
    beta = (0.2 * torch.randn(D, generator=g)).to(dev_)
    cs = torch.tensor([4.0], device=dev_)
    # ---- producer 1: to_out / ff2 form
    w, ws, wil = weights(D)
    b, r = torch.randn(D, generator=g).to(dev_), torch.randn(M, D, generator=g).to(dev_)
    c = torch.full((M, D), float("nan"), device=dev_)
    tw = ops.SplitIL(M, D, dev_)
    rowsq = torch.full((M, D // 64), float("nan"), device=dev_)
    ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il, bias=b, residual=r, out_split=tw, c_scale=cs, c_gamma=gamma, c_rowsq=rowsq)
    want = xs @ w.double().T + b.double() + r.double()
    assert rel_l2(c, want) < 1e-6
    assert rel_l2(pair(tw), want * gamma.double() * 4.0) < 1e-6
    assert rel_l2(rowsq.double(), c.double().square().reshape(M, D // 64, 64).sum(-1)) < 1e-6
    rs = torch.empty(M, device=dev_)
    ops.rownorm_scale(rowsq, M, D // 64, rs, float(D) ** 0.5)
    assert rel_l2(rs.double(), math.sqrt(D) / c.double().norm(dim=-1)) < 1e-6
    # ---- producer 2: skip-combiner form (A | A2, bias)
    w2, ws2, wil2 = weights(D, 2 * K)
    c2 = torch.full((M, D), float("nan"), device=dev_)
    tw2 = ops.SplitIL(M, D, dev_)
    rowsq2 = torch.full((M, D // 64), float("nan"), device=dev_)
    ops.gemm(x, w2, c2, w_split=ws2, w_il=wil2, a_split=il, a2=x, a2_split=il, bias=b, out_split=tw2, c_gamma=gamma, c_rowsq=rowsq2)
    want2 = torch.cat((xs, xs), 1) @ w2.double().T + b.double()
    assert rel_l2(c2, want2) < 1e-6 and rel_l2(pair(tw2), want2 * gamma.double()) < 1e-6
    assert rel_l2(rowsq2.double(), c2.double().square().reshape(M, D // 64, 64).sum(-1)) < 1e-6
    # the two K halves with DIFFERENT pre-scales (every stage of the pair-only residual stream carries its own)
    sa, sb = torch.tensor([2.0], device=dev_), torch.tensor([32.0], device=dev_)
    ila, ilb = ops.SplitIL(M, K, dev_), ops.SplitIL(M, K, dev_)
    ops.split_act_f16(x, ila, scale=sa); ops.split_act_f16(x, ilb, scale=sb)
    xa, xb = pair(ila) / 2.0, pair(ilb) / 32.0
    c2b = torch.full((M, D), float("nan"), device=dev_)
    tw2b = ops.SplitIL(M, D, dev_)
    ops.gemm(x, w2, c2b, w_split=ws2, w_il=wil2, a_split=ila, a2=x, a2_split=ilb, a_scale=sa, a2_scale=sb, bias=b, out_split=tw2b, c_rowsq=rowsq2)
    want2b = torch.cat((xa, xb), 1) @ w2.double().T + b.double()
    assert rel_l2(c2b, want2b) < 1e-6 and rel_l2(pair(tw2b), want2b) < 1e-6
    # ---- the reference's norm of the producer's fp32 output, in fp64
    cd = c.double()
    normed = cd / cd.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(D) * gamma.double() + beta.double()
    a_tw = pair(tw) / 4.0                                   # what the consumer's A operand holds (x * gamma, pre-scale divided out)
    # ---- consumer 1: ff1 form  (bias' = b1 + beta @ W1.T)
    w1, ws1, wil1 = weights(2048, D)
    b1 = torch.randn(2048, generator=g).to(dev_)
    b1p = (b1.double() + beta.double() @ w1.double().T).float()
    o = ops.SplitIL(M, 2048, dev_)
    guard = torch.full((M, 2048), 7.0, device=dev_)
    ops.gemm(c, w1, guard, w_split=ws1, w_il=wil1, a_split=tw, a_scale=cs, bias=b1p, act=1, out_split=o, write_f32=False, a_row_scale=rs)
    assert bool((guard == 7.0).all())
    got = pair(o)
    assert rel_l2(got, F.gelu(a_tw * rs.double()[:, None] @ w1.double().T + b1p.double())) < 1e-6       # the kernel's own arithmetic
    assert rel_l2(got, F.gelu(normed @ w1.double().T + b1.double())) < 2e-6                            # = the reference's norm -> ff1
    # ---- consumer 2: to_qkv form (bias = beta @ Wqkv.T in front of the rotation; q | k split, v transposed)
    T = M // 4
    Bt, Mq = 4, 4 * T
    wq, wsq, wilq = weights(3 * H * 64, D)
    bq = (beta.double() @ wq.double().T).float()
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(T).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().to(dev_).contiguous(), ang.sin().to(dev_).contiguous()
    twq = tw.rows_view(0, Mq)
    qk = (torch.empty(Mq, 2 * H * 64, dtype=torch.float16, device=dev_), torch.empty(Mq, 2 * H * 64, dtype=torch.float16, device=dev_))
    Tp = (T + 31) // 32 * 32
    vt = (torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev_), torch.zeros(Bt * H * 64, Tp, dtype=torch.float16, device=dev_))
    dummy = torch.empty(Mq, 3 * H * 64, device=dev_)
    ops.gemm(c[:Mq], wq, dummy, w_split=wsq, w_il=wilq, a_split=twq, a_scale=cs, bias=bq, rope=(cos, sin), rope_cols=2 * H * 64, out_split=qk,
             vt_split=vt, write_f32=False, a_row_scale=rs)
    z = normed[:Mq] @ wq.double().T
    zq = z[:, : 2 * H * 64].reshape(Bt, T, 2 * H, 64)
    c_, s_ = torch.cat((ang.cos(), ang.cos()), -1).double().to(dev_), torch.cat((ang.sin(), ang.sin()), -1).double().to(dev_)
    rot = torch.cat((-zq[..., 32:], zq[..., :32]), -1)
    want_qk = (zq * c_[None, :, None, :] + rot * s_[None, :, None, :]).reshape(Mq, -1)
    assert rel_l2(qk[0].double() + qk[1].double(), want_qk) < 2e-6
    v = z[:, 2 * H * 64:].reshape(Bt, T, H, 64).permute(0, 2, 3, 1).reshape(Bt * H * 64, T)
    slots = ops.vt_frame_slots(T, dev_)
    assert rel_l2((vt[0].double() + vt[1].double())[:, slots], v) < 2e-6
    # ---- the residual stream as pairs only: residual read from a pair, no fp32 store, raw twin (gamma on the weight side), in place
    hs = torch.tensor([8.0], device=dev_)
    rp = ops.SplitIL(M, D, dev_); ops.split_act_f16(r, rp, scale=hs)
    r_held = pair(rp) / 8.0
    c3 = torch.full((M, D), 7.0, device=dev_)
    rowsq3 = torch.full((M, D // 64), float("nan"), device=dev_)
    ops.gemm(x, w, c3, w_split=ws, w_il=wil, a_split=il, bias=b, res_split=rp, res_scale=hs, out_split=rp, c_scale=hs, c_rowsq=rowsq3, write_f32=False)
    want3 = xs @ w.double().T + b.double() + r_held
    assert bool((c3 == 7.0).all()) and rel_l2(pair(rp) / 8.0, want3) < 1e-6
    assert rel_l2(rowsq3.double(), want3.square().reshape(M, D // 64, 64).sum(-1)) < 1e-6
    ops.rownorm_scale(rowsq3, M, D // 64, rs, float(D) ** 0.5)
    # the same with the fp32 store as well (last layer: the final norm reads fp32), no row sums
    rp2 = ops.SplitIL(M, D, dev_); ops.split_act_f16(r, rp2, scale=hs)
    tw3 = ops.SplitIL(M, D, dev_)
    ops.gemm(x, w, c3, w_split=ws, w_il=wil, a_split=il, bias=b, res_split=rp2, res_scale=hs, out_split=tw3, c_scale=hs, write_f32=True)
    assert rel_l2(c3, want3) < 1e-6 and rel_l2(pair(tw3) / 8.0, want3) < 1e-6
    # ---- W diag(gamma(t)) pairs for several evaluation times at once, and the consumer on them
    n_sets = 3
    gam = (1.0 + 0.5 * torch.randn(n_sets, D, generator=g)).to(dev_)
    gsc = torch.exp2(-torch.ceil(torch.log2(gam.abs().amax(-1)))).contiguous()
    wg = torch.empty(n_sets, 2048, 2 * D, dtype=torch.float16, device=dev_)
    ops.split_f16_colscale_il(w1, gam, gsc, 1.0 / ws1[2], wg)
    for s_ in range(n_sets):
        dense = wg[s_].view(2048, D // 32, 2, 32)
        got_w = (dense[:, :, 0, :].double() + dense[:, :, 1, :].double()).reshape(2048, D)
        assert rel_l2(got_w, w1.double() * gam[s_].double() * float(gsc[s_]) / ws1[2]) < 1e-6
    h3 = pair(rp) / 8.0                                     # the pair-only stream from above is the consumer's A operand
    normed3 = h3 / h3.norm(dim=-1, keepdim=True) * math.sqrt(D) * gam[1].double() + beta.double()
    a_sc = (hs * gsc[1]).reshape(1).contiguous()
    ops.gemm(c, w1, guard, w_split=ws1, w_il=(wg[1], ws1[2]), a_split=rp, a_scale=a_sc, bias=b1p, act=1, out_split=o, write_f32=False, a_row_scale=rs)
    assert rel_l2(pair(o), F.gelu(normed3 @ w1.double().T + b1.double())) < 2e-6
    # ---- refused outside the large-problem kernel's four forms
    with pytest.raises(ops._lib.CovomixHipError):
        ops.gemm(x[:1000], w, c[:1000], w_split=ws, w_il=wil, a_split=il.rows_view(0, 1000), residual=r[:1000], out_split=tw.rows_view(0, 1000),
                 c_gamma=gamma, c_rowsq=rowsq)
    with pytest.raises(ops._lib.CovomixHipError):
        ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il, a_row_scale=rs)            # fp32 store with a row factor: not a consumer form


@pytest.mark.parametrize("M", [2100, 9298, 18596])
def test_gemm_192_row_tiles_are_bit_identical_to_256_row_tiles(ops, M):
    """Round 5: the large-problem kernel in its 192-row form (three instead of four A tiles per M half; taken where rounds x height
    come out smaller, e.g. 9,298 rows x N = 1024: 196 tiles of 192 rows instead of 148 of 256) - every output element is the same
    sum in the same order, so EVERY epilogue class must give the same bits as the 256-row form (flags 64 / 128 pin the height): plain,
    residual + twin, bias + GELU + split, A | A2 + bias with two pre-scales, QKV (RoPE, V^T; T % 4 != 0), the deferred-norm producer
    (pair residual in place, row sums) and consumer (factor per row); also against fp64, and the library's own choice at these sizes."""
    dev_ = dev()
    g = torch.Generator().manual_seed(M)
    K, H = 1024, 4
    x = torch.randn(M, K, generator=g).to(dev_)
    il = ops.SplitIL(M, K, dev_); ops.split_act_f16(x, il)
    xs = il.dense()[0].double() + il.dense()[1].double()
    pair = lambda t: t.dense()[0].double() + t.dense()[1].double()

    def weights(N, Kw=K):
        w = (torch.randn(N, Kw, generator=g) / math.sqrt(Kw)).to(dev_)
        ws = ops.split_f16(w)
        return w, ws, ops.split_f16_interleaved(ws)
    W1, W2, WA2, WQ = weights(1024), weights(2048), weights(1024, 2 * K), weights(3 * H * 64)
    b1, b2 = torch.randn(1024, generator=g).to(dev_), torch.randn(2048, generator=g).to(dev_)
    r = torch.randn(M, 1024, generator=g).to(dev_)
    rs = (0.5 + torch.rand(M, generator=g)).to(dev_)
    sa, sb, hs = (torch.tensor([v], device=dev_) for v in (2.0, 32.0, 8.0))
    ila, ilb = ops.SplitIL(M, K, dev_), ops.SplitIL(M, K, dev_)
    ops.split_act_f16(x, ila, scale=sa); ops.split_act_f16(x, ilb, scale=sb)
    T = M // 3
    Mq = 3 * T
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(T).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().to(dev_).contiguous(), ang.sin().to(dev_).contiguous()
    ilq = ops.SplitIL(Mq, K, dev_); ops.split_act_f16(x[:Mq].contiguous(), ilq)

    def run(flags):
        out = []
        with ops.gemm_flags(flags):
            c = torch.full((M, 1024), float("nan"), device=dev_)
            ops.gemm(x, W1[0], c, w_split=W1[1], w_il=W1[2], a_split=il); out.append(c.clone())
            tw = ops.SplitIL(M, 1024, dev_)
            ops.gemm(x, W1[0], c, w_split=W1[1], w_il=W1[2], a_split=il, bias=b1, residual=r, out_split=tw); out += [c.clone(), tw.buf.clone()]
            o = ops.SplitIL(M, 2048, dev_)
            ops.gemm(x, W2[0], torch.empty(M, 2048, device=dev_), w_split=W2[1], w_il=W2[2], a_split=il, bias=b2, act=1, out_split=o, write_f32=False)
            out.append(o.buf.clone())
            ops.gemm(x, WA2[0], c, w_split=WA2[1], w_il=WA2[2], a_split=ila, a2=x, a2_split=ilb, a_scale=sa, a2_scale=sb, bias=b1,
                     out_split=tw, c_rowsq=torch.empty(M, 16, device=dev_)); out += [c.clone(), tw.buf.clone()]
            qk = (torch.empty(Mq, 2 * H * 64, dtype=torch.float16, device=dev_), torch.empty(Mq, 2 * H * 64, dtype=torch.float16, device=dev_))
            Tp = (T + 31) // 32 * 32
            vt = (torch.zeros(3 * H * 64, Tp, dtype=torch.float16, device=dev_), torch.zeros(3 * H * 64, Tp, dtype=torch.float16, device=dev_))
            ops.gemm(x[:Mq], WQ[0], torch.empty(Mq, 3 * H * 64, device=dev_), w_split=WQ[1], w_il=WQ[2], a_split=ilq, rope=(cos, sin),
                     rope_cols=2 * H * 64, out_split=qk, vt_split=vt, write_f32=False)
            out += [qk[0].clone(), qk[1].clone(), vt[0].clone(), vt[1].clone()]
            rp = ops.SplitIL(M, 1024, dev_); ops.split_act_f16(r, rp, scale=hs)
            rowsq = torch.full((M, 16), float("nan"), device=dev_)
            ops.gemm(x, W1[0], c, w_split=W1[1], w_il=W1[2], a_split=il, bias=b1, res_split=rp, res_scale=hs, out_split=rp, c_scale=hs,
                     c_rowsq=rowsq, write_f32=False); out += [rp.buf.clone(), rowsq.clone()]
            o2 = ops.SplitIL(M, 2048, dev_)
            ops.gemm(x, W2[0], torch.empty(M, 2048, device=dev_), w_split=W2[1], w_il=W2[2], a_split=il, bias=b2, act=1, out_split=o2,
                     write_f32=False, a_row_scale=rs); out.append(o2.buf.clone())
        torch.cuda.synchronize()
        return out
    t256, t192, auto, mixed = run(16 | 128), run(16 | 64), run(16), run(16 | 32)
    assert len(t256) == len(t192) == 13
    for k, (a, b, c_, d_) in enumerate(zip(t256, t192, auto, mixed)):
        assert torch.equal(a, b), k
        assert torch.equal(a, c_), k
        assert torch.equal(a, d_), k          # round 6: whole rounds of 256-row tiles + a tail launch of 192-row tiles (flag 32)
    assert rel_l2(t192[0], xs @ W1[0].double().T) < 1e-6
    assert rel_l2(t192[1], xs @ W1[0].double().T + b1.double() + r.double()) < 1e-6
    o_pair = t192[3].view(M, 64, 2, 32)
    assert rel_l2(o_pair[:, :, 0].reshape(M, 2048).double() + o_pair[:, :, 1].reshape(M, 2048).double(), F.gelu(xs @ W2[0].double().T + b2.double())) < 1e-6
