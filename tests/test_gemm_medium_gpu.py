"""The medium-problem split-precision GEMM (csrc/gemm_f16x3_p8m.hip: interleaved operands, fewer than 2048 rows - one utterance,
reference monologue_generation.py:259-304) and the GEMM + norm entry point (cvx_gemm_f16x3_norm, reference acoustic.py:306-318),
through the C ABI, against fp64 products of the split operands."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import covomix_amd.ops as o
    return o


def dev():
    return torch.device("cuda:0")


def _il(ops, x):
    il = ops.SplitIL(x.shape[0], x.shape[1], dev())
    ops.split_act_f16(x, il)
    return il, il.dense()[0].double() + il.dense()[1].double()


def _weights(ops, g, N, K):
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev())
    ws = ops.split_f16(w)
    return w, ws, ops.split_f16_interleaved(ws)


@pytest.mark.parametrize("M,K", [(1000, 1024), (130, 1024), (1999, 512), (64, 1024), (333, 96), (257, 32), (1000, 4096)])
def test_every_epilogue(ops, M, K):
    """plain / residual / bias + residual + split twin / bias + GELU + split only / K-split A | A2 + bias, K with an odd and a
    single K-tile, M with a ragged last row tile; K = 4096 and N = 1024 take the split-K path (4 slices)."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(dev())
    il, xs = _il(ops, x)
    w, ws, wil = _weights(ops, g, 1024, K)
    b, r = torch.randn(1024, generator=g).to(dev()), torch.randn(M, 1024, generator=g).to(dev())
    ref = xs @ w.double().T
    c = torch.full((M + 1, 1024), float("nan"), device=dev())
    ops.gemm(x, w, c[:M], w_split=ws, w_il=wil, a_split=il)
    assert rel_l2(c[:M], ref) < 1e-6 and bool(torch.isnan(c[M]).all())
    ops.gemm(x, w, c[:M], w_split=ws, w_il=wil, a_split=il, residual=r)
    assert rel_l2(c[:M], ref + r.double()) < 1e-6
    tw = ops.SplitIL(M, 1024, dev())
    ops.gemm(x, w, c[:M], w_split=ws, w_il=wil, a_split=il, bias=b, residual=r, out_split=tw)
    want = ref + b.double() + r.double()
    assert rel_l2(c[:M], want) < 1e-6 and rel_l2(tw.dense()[0].double() + tw.dense()[1].double(), want) < 1e-6
    assert bool(torch.isnan(c[M]).all())
    w, ws, wil = _weights(ops, g, 2048, K)
    b = torch.randn(2048, generator=g).to(dev())
    o = ops.SplitIL(M, 2048, dev())
    guard = torch.full((M, 2048), 7.0, device=dev())
    ops.gemm(x, w, guard, w_split=ws, w_il=wil, a_split=il, bias=b, act=1, out_split=o, write_f32=False)
    assert rel_l2(o.dense()[0].double() + o.dense()[1].double(), F.gelu(xs @ w.double().T + b.double())) < 1e-6 and bool((guard == 7.0).all())
    w, ws, wil = _weights(ops, g, 1024, 2 * K)
    b = torch.randn(1024, generator=g).to(dev())
    x2 = torch.randn(M, K, generator=g).to(dev())
    il2, xs2 = _il(ops, x2)
    ops.gemm(x, w, c[:M], w_split=ws, w_il=wil, a_split=il, a2=x2, a2_split=il2, bias=b)
    assert rel_l2(c[:M], torch.cat((xs, xs2), 1) @ w.double().T + b.double()) < 1e-6


@pytest.mark.parametrize("Bt,T,H", [(2, 500, 16), (3, 333, 4), (1, 203, 2), (4, 64, 4)])
def test_qkv_epilogue(ops, Bt, T, H):
    """RoPE on q | k, split q | k, transposed split v (T % 4 == 0 and != 0), V blocks on the un-swapped product."""
    M, K = Bt * T, 1024
    g = torch.Generator().manual_seed(M + H)
    x = torch.randn(M, K, generator=g).to(dev())
    il, xs = _il(ops, x)
    w, ws, wil = _weights(ops, g, 3 * H * 64, K)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(T).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().to(dev()).contiguous(), ang.sin().to(dev()).contiguous()
    qk = (torch.full((M + 1, 2 * H * 64), float("nan"), dtype=torch.float16, device=dev()),
          torch.full((M + 1, 2 * H * 64), float("nan"), dtype=torch.float16, device=dev()))
    Tp = (T + 31) // 32 * 32
    rows = Bt * H * 64
    vt = (torch.zeros(rows + 64, Tp, dtype=torch.float16, device=dev()), torch.zeros(rows + 64, Tp, dtype=torch.float16, device=dev()))
    vt[0][rows:] = float("nan"); vt[1][rows:] = float("nan")
    ops.gemm(x, w, torch.empty(M, 3 * H * 64, device=dev()), w_split=ws, w_il=wil, a_split=il, rope=(cos, sin), rope_cols=2 * H * 64,
             out_split=(qk[0][:M], qk[1][:M]), vt_split=(vt[0][:rows], vt[1][:rows]), write_f32=False)
    z = xs @ w.double().T
    zq = z[:, : 2 * H * 64].reshape(Bt, T, 2 * H, 64)
    c_, s_ = torch.cat((ang.cos(), ang.cos()), -1).double().to(dev()), torch.cat((ang.sin(), ang.sin()), -1).double().to(dev())
    rot = torch.cat((-zq[..., 32:], zq[..., :32]), -1)
    want_qk = (zq * c_[None, :, None, :] + rot * s_[None, :, None, :]).reshape(M, -1)
    assert rel_l2(qk[0][:M].double() + qk[1][:M].double(), want_qk) < 1e-6
    assert bool(torch.isnan(qk[0][M]).all()) and bool(torch.isnan(qk[1][M]).all())
    v = z[:, 2 * H * 64:].reshape(Bt, T, H, 64).permute(0, 2, 3, 1).reshape(rows, T)
    slots = ops.vt_frame_slots(T, dev())
    assert rel_l2((vt[0][:rows].double() + vt[1][:rows].double())[:, slots], v) < 1e-6
    free = torch.ones(Tp, dtype=torch.bool, device=dev()); free[slots] = False
    assert float(vt[0][:rows][:, free].abs().max() if bool(free.any()) else 0) == 0
    assert bool(torch.isnan(vt[0][rows:]).all()) and bool(torch.isnan(vt[1][rows:]).all())


def test_n_not_multiple_of_128(ops):
    """N % 64 == 0, N % 128 != 0: the trailing wave tiles lie past column N (no table reads, no stores there)."""
    M, K = 700, 1024
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g).to(dev())
    il, xs = _il(ops, x)
    for N in (576, 64, 192):
        w, ws, wil = _weights(ops, g, N, K)
        b = torch.randn(N, generator=g).to(dev())
        rwide = torch.randn(M, N + 128, generator=g).to(dev())
        cw = torch.full((M + 1, N + 128), float("nan"), device=dev())
        ops.gemm(x, w, cw[:M, :N], w_split=ws, w_il=wil, a_split=il, bias=b, residual=rwide[:, :N])
        assert rel_l2(cw[:M, :N], xs @ w.double().T + b.double() + rwide[:, :N].double()) < 1e-6, N
        assert bool(torch.isnan(cw[:, N:]).all()) and bool(torch.isnan(cw[M]).all()), N
        oh = torch.full((M + 1, N + 128), float("nan"), dtype=torch.float16, device=dev())
        ol = torch.full((M + 1, N + 128), float("nan"), dtype=torch.float16, device=dev())
        ops.gemm(x, w, torch.empty(M, N, device=dev()), w_split=ws, w_il=wil, a_split=il, bias=b, act=1, out_split=(oh[:M, :N], ol[:M, :N]), write_f32=False)
        assert rel_l2(oh[:M, :N].double() + ol[:M, :N].double(), F.gelu(xs @ w.double().T + b.double())) < 1e-6, N
        assert bool(torch.isnan(oh[:, N:]).all()) and bool(torch.isnan(oh[M]).all()), N


def test_results_do_not_depend_on_the_run(ops):
    """Block-local K split + exchange, split-K over blocks: fixed summation orders - the same bits every time."""
    M, K = 1000, 2048
    g = torch.Generator().manual_seed(6)
    x = torch.randn(M, K, generator=g).to(dev())
    il, _ = _il(ops, x)
    w, ws, wil = _weights(ops, g, 1024, K)
    outs = []
    for _ in range(4):
        c = torch.empty(M, 1024, device=dev())
        ops.gemm(x, w, c, w_split=ws, w_il=wil, a_split=il)
        outs.append(c.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("M,N,K,il_out", [(1000, 1024, 1024, True), (1000, 1024, 4096, True), (700, 512, 2048, False), (4096, 1024, 1024, True),
                                           (1000, 2048, 1024, True)])
def test_gemm_norm_equals_gemm_then_norm(ops, M, N, K, il_out):
    """cvx_gemm_f16x3_norm: the AdaptiveRMSNorm of the product's rows, inside the split-K reduction (M < 2048, N <= 1024) or as the
    separate kernel behind the product (every other shape) - the same bits as ops.gemm followed by ops.adarmsnorm, and the
    reference arithmetic (acoustic.py:198-204) against fp64."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev())
    il, xs = _il(ops, x)
    w, ws, wil = _weights(ops, g, N, K)
    b, r = torch.randn(N, generator=g).to(dev()), torch.randn(M, N, generator=g).to(dev())
    gam, bet = torch.randn(N, generator=g).to(dev()), torch.randn(N, generator=g).to(dev())
    sc = torch.tensor([4.0], device=dev())
    mk = (lambda: ops.SplitIL(M, N, dev())) if il_out else (lambda: (torch.empty(M, N, dtype=torch.float16, device=dev()), torch.empty(M, N, dtype=torch.float16, device=dev())))
    dense = lambda p: p.dense() if isinstance(p, ops.SplitIL) else p
    for beta in (bet, None):
        c0, c1 = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev())
        y0, y1, t0, t1 = mk(), mk(), ops.SplitIL(M, N, dev()), ops.SplitIL(M, N, dev())
        ops.gemm(x, w, c0, w_split=ws, w_il=wil, a_split=il, bias=b, residual=r, out_split=t0)
        ops.adarmsnorm(c0, gam, beta, None, out_split=y0, split_scale=sc)
        ops.gemm(x, w, c1, w_split=ws, w_il=wil, a_split=il, bias=b, residual=r, out_split=t1,
                 norm=dict(gamma=gam, beta=beta, out_split=y1, scale=sc))
        assert torch.equal(c0, c1) and torch.equal(t0.buf, t1.buf)
        assert all(torch.equal(p, q) for p, q in zip(dense(y0), dense(y1)))
        h = xs @ w.double().T + b.double() + r.double()
        want = h / h.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(N) * gam.double() + (beta.double() if beta is not None else 0.0)
        got = (dense(y1)[0].double() + dense(y1)[1].double()) / 4.0
        assert rel_l2(got, want) < 2e-6


@pytest.mark.parametrize("M,N,K", [(1000, 80, 1024), (300, 16, 256), (700, 208, 512)])
def test_partial_wave_tile(ops, M, N, K):
    """N % 16 == 0 only (to_pred: N = 80, eight K slices): the trailing partial wave tile multiplies clamped W rows and stores
    nothing; fp32 output inside a NaN-filled wider buffer is not required to be contiguous."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev())
    il, xs = _il(ops, x)
    w, ws, wil = _weights(ops, g, N, K)
    b = torch.randn(N, generator=g).to(dev())
    c = torch.full((M + 1, N), float("nan"), device=dev())
    ops.gemm(x, w, c[:M], w_split=ws, w_il=wil, a_split=il, bias=b)
    assert rel_l2(c[:M], xs @ w.double().T + b.double()) < 1e-6 and bool(torch.isnan(c[M]).all())
    cw = torch.full((M, N + 16), float("nan"), device=dev())
    ops.gemm(x, w, cw[:, :N], w_split=ws, w_il=wil, a_split=il)
    assert rel_l2(cw[:, :N], xs @ w.double().T) < 1e-6 and bool(torch.isnan(cw[:, N:]).all())
