"""GPU: the operator-level entry points named in SURVEY.md section 8(b) (cvx_rope_attention_f32, cvx_hifigan_convt_f32,
cvx_hifigan_resblock_f32, cvx_hifigan_pre_post_f32) against plain fp64 torch restatements of the reference operators
(acoustic.py:132-137, 227-235; attend.py:108-126; covomix/vocoder/models.py:35-42, 81, 85-88, 100-114)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from covomix_amd import ops  # noqa: E402  (ops._stream(): the launch context the entry points take)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("Bt,T,H", [(2, 100, 16), (1, 37, 2), (3, 256, 12)])
def test_rope_attention(Bt, T, H):
    from covomix_amd import _lib
    g = torch.Generator().manual_seed(Bt * 1000 + T)
    qkv = torch.randn(Bt, T, 3 * H * 64, generator=g)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).double() / 64))
    ang = torch.arange(T).double()[:, None] * inv[None, :]
    cos, sin = ang.cos(), ang.sin()
    q, k, v = (t.view(Bt, T, H, 64).transpose(1, 2).double() for t in qkv.chunk(3, dim=-1))

    def rot(x):                                            # rotate_half / apply_rotary_pos_emb, acoustic.py:132-137
        c, s = torch.cat((cos, cos), -1), torch.cat((sin, sin), -1)
        x1, x2 = x[..., :32], x[..., 32:]
        return x * c + torch.cat((-x2, x1), -1) * s
    sim = rot(q) @ rot(k).transpose(-1, -2) * 0.125
    want = (sim.softmax(-1) @ v).transpose(1, 2).reshape(Bt, T, H * 64)
    qd = qkv.cuda()
    out = torch.empty(Bt, T, H * 64, device="cuda")
    ws = torch.empty_like(qd)
    cs, sn = cos.float().cuda().contiguous(), sin.float().cuda().contiguous()
    _lib.check(_lib.load().cvx_rope_attention_f32(qd.data_ptr(), cs.data_ptr(), sn.data_ptr(), out.data_ptr(), Bt, T, H, 0.125,
                                                  ws.data_ptr(), ops._stream()), "cvx_rope_attention_f32")
    assert rel(out, want) < 5e-6
    assert torch.equal(qd.cpu(), qkv)                     # the input is not modified


def _packed(w, transposed=False):
    return ops.hifigan_pack_weight(w, transposed).cuda()


@pytest.mark.parametrize("C_,k,dils,L", [(62, 3, (1, 3, 5), 700), (31, 11, (1, 3, 5), 1000), (125, 7, (1, 3, 5), 333)])
def test_resblock_entry_point(C_, k, dils, L):
    from covomix_amd import _lib
    g = torch.Generator().manual_seed(C_ * 100 + k)
    B = 2
    x = torch.randn(B, C_, L, generator=g)
    w1 = [torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5 for _ in range(3)]
    w2 = [torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5 for _ in range(3)]
    b1 = [torch.randn(C_, generator=g) * 0.1 for _ in range(3)]
    b2 = [torch.randn(C_, generator=g) * 0.1 for _ in range(3)]
    xs = torch.randn(B, C_, L, generator=g)
    y = x.double()
    for m in range(3):                                     # ResBlock1.forward, models.py:35-42
        xt = F.conv1d(F.leaky_relu(y, 0.1), w1[m].double(), b1[m].double(), dilation=dils[m], padding=(k - 1) * dils[m] // 2)
        xt = F.conv1d(F.leaky_relu(xt, 0.1), w2[m].double(), b2[m].double(), padding=(k - 1) // 2)
        y = xt + y
    want = (y + xs.double()) / 3.0
    a = _lib.ResblockArgs()
    xd, tmp, out, acc = x.cuda(), torch.empty(B, C_, L, device="cuda"), torch.empty(B, C_, L, device="cuda"), xs.cuda()
    keep = []
    for m in range(3):
        p1, p2, c1, c2 = _packed(w1[m]), _packed(w2[m]), b1[m].cuda(), b2[m].cuda()
        keep += [p1, p2, c1, c2]
        a.Wp1[m], a.Wp2[m], a.b1[m], a.b2[m], a.dil[m] = p1.data_ptr(), p2.data_ptr(), c1.data_ptr(), c2.data_ptr(), dils[m]
    a.x, a.B, a.C, a.L, a.ksize = xd.data_ptr(), B, C_, L, k
    a.tmp, a.out, a.accum, a.out_scale = tmp.data_ptr(), out.data_ptr(), acc.data_ptr(), 1.0 / 3.0
    _lib.check(_lib.load().cvx_hifigan_resblock_f32(C.byref(a), ops._stream()), "cvx_hifigan_resblock_f32")
    assert rel(out, want) < 5e-6
    assert torch.equal(xd.cpu(), x)
    a.out = a.x                                            # aliasing x is rejected
    assert _lib.load().cvx_hifigan_resblock_f32(C.byref(a), ops._stream()) != 0


def test_convt_and_pre_post_entry_points():
    from covomix_amd import _lib
    from covomix_amd._lib import ConvArgs
    g = torch.Generator().manual_seed(5)
    lib, st = _lib.load(), ops._stream()
    B, Cin, Cout, L, k, u = 2, 62, 31, 300, 4, 2          # ups[3] of config_covomix: ConvTranspose1d(62, 31, 4, 2, padding=1)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cin, Cout, k, generator=g) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=(k - u) // 2)
    a = ConvArgs()
    xd, wp, bd = x.cuda(), ops.hifigan_pack_conv_transpose1d(w, u, (k - u) // 2).cuda(), b.cuda()      # polyphase packing
    out = torch.empty(B, Cout, want.shape[2], device="cuda")
    a.x, a.B, a.Cin, a.Lin, a.Wp, a.bias = xd.data_ptr(), B, Cin, L, wp.data_ptr(), bd.data_ptr()
    a.out, a.Cout, a.Lout, a.ksize, a.dil, a.pad, a.up = out.data_ptr(), Cout, out.shape[2], k, 1, k - 1 - (k - u) // 2, u
    a.in_slope, a.res, a.accum, a.out_scale = 0.1, None, None, 1.0
    _lib.check(lib.cvx_hifigan_convt_f32(C.byref(a), st), "cvx_hifigan_convt_f32")
    assert rel(out, want) < 5e-6
    a.up = 1
    assert lib.cvx_hifigan_convt_f32(C.byref(a), st) != 0           # a plain conv is not a ConvTranspose1d
    # conv_pre (Conv1d(80, C0, 7, padding=3)) and the output stage, each alone and both in one call
    C0, T = 64, 50
    mel = torch.randn(B, 80, T, generator=g)
    wpre, bpre = torch.randn(C0, 80, 7, generator=g) / 24, torch.randn(C0, generator=g) * 0.1
    want_pre = F.conv1d(mel.double(), wpre.double(), bpre.double(), padding=3)
    pre = ConvArgs()
    md, wpp, bpp = mel.cuda(), _packed(wpre), bpre.cuda()
    o_pre = torch.empty(B, C0, T, device="cuda")
    pre.x, pre.B, pre.Cin, pre.Lin, pre.Wp, pre.bias = md.data_ptr(), B, 80, T, wpp.data_ptr(), bpp.data_ptr()
    pre.out, pre.Cout, pre.Lout, pre.ksize, pre.dil, pre.pad, pre.up = o_pre.data_ptr(), C0, T, 7, 1, 3, 1
    pre.in_slope, pre.res, pre.accum, pre.out_scale = 1.0, None, None, 1.0
    hx = torch.randn(B, 31, 400, generator=g)
    wpost, bpost = torch.randn(1, 31, 7, generator=g) / 15, 0.05
    want_post = torch.tanh(F.conv1d(F.leaky_relu(hx.double(), 0.01), wpost.double(), torch.tensor([bpost]).double(), padding=3))
    hd, wd = hx.cuda(), wpost.reshape(31, 7).contiguous().cuda()
    o_post = torch.empty(B, 1, 400, device="cuda")
    _lib.check(lib.cvx_hifigan_pre_post_f32(C.byref(pre), hd.data_ptr(), wd.data_ptr(), bpost, o_post.data_ptr(), B, 31, 400, 0.01, st),
               "cvx_hifigan_pre_post_f32")
    assert rel(o_pre, want_pre) < 5e-6 and rel(o_post, want_post) < 5e-6
    o_pre.zero_()
    _lib.check(lib.cvx_hifigan_pre_post_f32(C.byref(pre), None, None, 0.0, None, 0, 0, 0, 0.0, st), "cvx_hifigan_pre_post_f32")
    assert rel(o_pre, want_pre) < 5e-6
    assert lib.cvx_hifigan_pre_post_f32(None, None, None, 0.0, None, 0, 0, 0, 0.0, st) != 0


@pytest.mark.parametrize("C_,k,dils,L", [(62, 3, (1, 3, 5), 700), (31, 11, (1, 3, 5), 1000), (125, 7, (1, 3, 5), 333), (250, 3, (1, 3, 5), 300)])
def test_resblock_f16x3_entry_point(C_, k, dils, L):
    """cvx_hifigan_resblock_f16x3: the operator-level ResBlock1 on the split-precision convolutions - the form the host
    runs (channels-last buffers, device-resident activation pre-scale) - against the fp64 torch restatement."""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(C_ * 100 + k + 1)
    B = 2
    x = torch.randn(B, C_, L, generator=g) * 3.0
    w1 = [torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5 for _ in range(3)]
    w2 = [torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5 for _ in range(3)]
    b1 = [torch.randn(C_, generator=g) * 0.1 for _ in range(3)]
    b2 = [torch.randn(C_, generator=g) * 0.1 for _ in range(3)]
    xs = torch.randn(B, C_, L, generator=g)
    y = x.double()
    for m in range(3):                                     # ResBlock1.forward, models.py:35-42
        xt = F.conv1d(F.leaky_relu(y, 0.1), w1[m].double(), b1[m].double(), dilation=dils[m], padding=(k - 1) * dils[m] // 2)
        xt = F.conv1d(F.leaky_relu(xt, 0.1), w2[m].double(), b2[m].double(), padding=(k - 1) // 2)
        y = xt + y
    want = (y + xs.double()) / 3.0
    dev = torch.device("cuda:0")
    np_ = 32 if C_ <= 32 else 64 if C_ <= 64 else 128 if C_ <= 128 else 256
    Lp = ops.hifigan_cl_rows(L)
    f32 = lambda: torch.zeros(B, Lp, np_, dtype=torch.float32, device=dev)
    f16 = lambda: (torch.zeros(B, Lp, np_, dtype=torch.float16, device=dev), torch.zeros(B, Lp, np_, dtype=torch.float16, device=dev))
    buf = dict(x0=f32(), z0=f16(), t=f16(), r0=f32(), r1=f32(), rz0=f16(), rz1=f16(), xs=f32(), acc=f32())
    xd = x.to(dev)
    scale = torch.ones(1, device=dev)
    ops.amax_pow2_scale(xd, 1024.0, scale, torch.zeros(1, dtype=torch.int32, device=dev))
    assert float(scale) == 2.0 ** round(float(torch.log2(1024.0 / x.abs().max())))
    ops.hifigan_to_channels_last(xd, buf["x0"], buf["z0"], 0.1, z_scale=scale)
    ops.hifigan_to_channels_last(xs.to(dev), buf["acc"], None, 0.1)

    def conv(w, b):
        c = SimpleNamespace(k=k, dil=1)
        c.w16 = ops.hifigan_pack_weight_f16x3(w.to(dev))
        c.bias16 = torch.zeros(np_, device=dev)
        c.bias16[:C_] = b.to(dev)
        return c
    block = []
    for m in range(3):
        c1, c2 = conv(w1[m], b1[m]), conv(w2[m], b2[m])
        c1.dil = dils[m]
        block.append((c1, c2))
    x0_before = buf["x0"].clone()
    ops.hifigan_resblock_f16x3(buf["x0"], buf["z0"], block, B, L, buf, accum=buf["acc"], out=buf["xs"], out_scale=1.0 / 3.0, z_scale=scale)
    out = torch.empty(B, C_, L, device=dev)
    ops.hifigan_from_channels_last(buf["xs"], out)
    assert rel(out, want) < 5e-6
    assert torch.equal(buf["x0"], x0_before)               # the block input is not modified
    assert float(buf["xs"][:, :, C_:].abs().max() if np_ > C_ else 0.0) == 0.0      # padded channels stay zero


@pytest.mark.parametrize("C_,k,dil,L,B", [(31, 11, 5, 40000, 2), (62, 7, 3, 36000, 2), (31, 3, 1, 70000, 1), (62, 11, 5, 517, 3), (20, 7, 5, 100, 1)])
def test_resblock_pair_fused_kernel(C_, k, dil, L, B):
    """cvx_hifigan_resblock_pair_f16x3 (one kernel per conv pair, intermediate in LDS, persistent blocks walking several
    tiles when B * ceil(L / (256 - (k-1))) exceeds the CU count) against the fp64 torch restatement of models.py:36-40,
    with the xs accumulate / scale of Generator.forward; the zero halos and padded channels stay zero."""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(C_ * 1000 + k * 10 + dil)
    x = torch.randn(B, C_, L, generator=g) * 0.05                 # (small: the measured pre-scale does the work)
    w1 = torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5
    w2 = torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5
    b1, b2 = torch.randn(C_, generator=g) * 0.01, torch.randn(C_, generator=g) * 0.01
    xs = torch.randn(B, C_, L, generator=g) * 0.05
    xt = F.conv1d(F.leaky_relu(x.double(), 0.1), w1.double(), b1.double(), dilation=dil, padding=(k - 1) * dil // 2)
    xt = F.conv1d(F.leaky_relu(xt, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2)
    want_x = xt + x.double()
    want_xs = (want_x + xs.double()) * 0.5
    dev = torch.device("cuda:0")
    np_ = 32 if C_ <= 32 else 64
    Lp = ops.hifigan_cl_rows(L)
    f32 = lambda: torch.zeros(B, Lp, np_, dtype=torch.float32, device=dev)
    x0, o1, o2, acc = f32(), f32(), f32(), f32()
    xd = x.to(dev)
    scale = torch.ones(1, device=dev)
    ops.amax_pow2_scale(xd, 1024.0, scale, torch.zeros(1, dtype=torch.int32, device=dev))
    ops.hifigan_to_channels_last(xd, x0, None, 0.1)
    ops.hifigan_to_channels_last(xs.to(dev), acc, None, 0.1)

    def conv(w, b, d):
        c = SimpleNamespace(k=k, dil=d)
        c.w16 = ops.hifigan_pack_weight_f16x3(w.to(dev))
        c.bias16 = torch.zeros(np_, device=dev)
        c.bias16[:C_] = b.to(dev)
        return c
    c1, c2 = conv(w1, b1, dil), conv(w2, b2, 1)
    ops.hifigan_resblock_pair_f16x3(x0, c1, c2, B, L, o1, z_scale=scale)
    ops.hifigan_resblock_pair_f16x3(x0, c1, c2, B, L, o2, accum=acc, out_scale=0.5, z_scale=scale)
    for buf, want in ((o1, want_x), (o2, want_xs)):
        out = torch.empty(B, C_, L, device=dev)
        ops.hifigan_from_channels_last(buf, out)
        assert rel(out, want) < 5e-6
        assert float(buf[:, : ops.HIFI_HALO_L].abs().max()) == 0.0 and float(buf[:, ops.HIFI_HALO_L + L:].abs().max()) == 0.0
        if np_ > C_:
            assert float(buf[:, :, C_:].abs().max()) == 0.0
    if np_ == 64:                                           # the 128-row / two-blocks-per-CU variant of the same kernel (A/B flag)
        o3 = f32()
        ops.hifigan_resblock_pair_f16x3(x0, c1, c2, B, L, o3, z_scale=scale, flags=1)
        out = torch.empty(B, C_, L, device=dev)
        ops.hifigan_from_channels_last(o3, out)
        assert rel(out, want_x) < 5e-6
    # accum may alias out (the generator's running xs)
    ops.hifigan_resblock_pair_f16x3(x0, c1, c2, B, L, acc, accum=acc, out_scale=0.5, z_scale=scale)
    assert torch.equal(acc, o2)


@pytest.mark.parametrize("Cin,Cout,k,u,pad,L,B", [(500, 250, 8, 5, 1, 130, 2), (250, 125, 8, 4, 2, 300, 2), (125, 62, 4, 4, 0, 777, 1),
                                                  (62, 31, 4, 2, 1, 1000, 3), (20, 10, 3, 5, 0, 50, 1), (16, 40, 7, 3, 2, 64, 2)])
def test_conv_transpose1d_polyphase(Cin, Cout, k, u, pad, L, B):
    """cvx_hifigan_conv_transpose1d_f32 (one stride-1 convolution per output phase) against torch's fp64 conv_transpose1d of
    leaky_relu(x) - the four upsamplers of config_covomix.json (models.py:85-88), a kernel shorter than the stride (phases
    that see no tap at all: bias only) and an odd one - and the fused max|out| against the tensor's own maximum."""
    g = torch.Generator().manual_seed(Cin * 7 + k * 3 + u)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cin, Cout, k, generator=g) / (Cin * k / u) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=pad)
    dev = torch.device("cuda:0")
    wp = ops.hifigan_pack_conv_transpose1d(w, u, pad).to(dev)
    out = torch.full((B, Cout, want.shape[2]), float("nan"), device=dev)
    bits = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.hifigan_conv_transpose1d(x.to(dev), wp, b.to(dev), out, cout=Cout, ksize=k, stride=u, padding=pad, in_slope=0.1, amax_bits=bits)
    assert rel(out, want) < 5e-6
    assert float(bits.view(torch.float32)) == float(out.abs().max())
    scale = torch.zeros(1, device=dev)
    ops.pow2_scale_from_amax(bits, 1024.0, scale)
    assert float(scale) == 2.0 ** round(float(torch.log2(1024.0 / out.abs().max()))) and int(bits) == 0
    # the zero-stuffed form of the same operator (cvx_hifigan_conv1d_f32 with up > 1) agrees to rounding
    if k - 1 - pad >= 0 and (k - 1) <= 50:
        ref = torch.empty_like(out)
        ops.hifigan_conv1d(x.to(dev), ops.hifigan_pack_weight(w, True).to(dev), b.to(dev), ref, cout=Cout, ksize=k, pad=k - 1 - pad, up=u, in_slope=0.1)
        assert rel(out, ref.double()) < 2e-6


def test_clock_stamps_give_a_plausible_shader_clock():
    """cvx_clock_stamps (bench.py's effective clock): two calls around a BUSY region, paired CU by CU (the cycle counter belongs to the CU:
    stamps of different CUs are not comparable) -> cycles / real time must be a shader clock this chip can run at on every XCD, most CUs
    must have been paired, and the real-time counter must advance at 100 MHz (checked against the host's clock over the region)."""
    import time
    x = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        x = torch.tanh(x) * 1.01                                   # (clocks up before the first stamp)
    torch.cuda.synchronize()
    c0 = ops.clock_stamps(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        x = torch.tanh(x) * 1.01
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    c1 = ops.clock_stamps(); torch.cuda.synchronize()
    r = ops.clock_from_stamps(c0, c1)
    assert r and r["cus"] >= 128 and len(r["xcd_mhz"]) == 8, r
    assert 300.0 < min(r["xcd_mhz"]) and max(r["xcd_mhz"]) < 2500.0, r
    a, b = c0.cpu(), c1.cpu()
    ok = (a[:, 1] != 0) & (b[:, 1] != 0)
    real_s = float((b[ok, 1] - a[ok, 1]).double().mean()) / 100e6
    assert abs(real_s / (t1 - t0) - 1.0) < 0.25, (real_s, t1 - t0)
