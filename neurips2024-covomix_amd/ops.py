"""Torch-tensor front ends of the C-ABI kernels.  Tensors are only device memory +
a stream here; every FLOP of the hot path happens in libcovomix_hip.so."""
from __future__ import annotations

import ctypes as C
import os as _os
import functools
import threading
from contextlib import contextmanager
from typing import Optional

import torch

from . import _lib
from ._lib import ConvArgs, GemmArgs, GemmNorm, GemmSplitIO

ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2


# Launch contexts (include/covomix_hip.h: cvx_ctx): one caller-owned struct per (device, stream) - the stream handle, its sticky
# saturation flag and the CUs it owns.  The library keeps nothing about a stream; everything an entry point needs travels in the
# struct whose ADDRESS is passed where the C prototypes say cvx_stream_t.  Never freed (captured graphs do not hold them, but a
# raw address in flight must stay valid; there are a handful per process).
_CTX: dict = {}


def ctx_of(stream: Optional["torch.cuda.Stream"] = None) -> "_lib.Ctx":
    st = stream if stream is not None else torch.cuda.current_stream()
    key = (st.device.index, st.cuda_stream)
    c = _CTX.get(key)
    if c is None:
        c = _lib.Ctx(st.cuda_stream, None, torch.cuda.get_device_properties(st.device).multi_processor_count, 0)
        _CTX[key] = c
    return c


def _stream() -> int:
    """What the entry points take as `cvx_stream_t`: the address of the current stream's launch context."""
    return C.addressof(ctx_of())


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.CovomixHipError("covomix_amd ops need tensors on the GPU (no CPU fallback exists)")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")


class SplitIL:
    """INTERLEAVED split pair of a [rows, cols] fp32 tensor: one fp16 buffer [rows, 2*cols] laid out [hi 32 | lo 32] per
    block of 32 columns, so that a K-step of one row is one 128-byte cache line for the large-problem GEMM's DMA.  The C ABI
    recognises it by lo == hi + 32 halves.  Accepted wherever a (hi, lo) tuple of split ACTIVATIONS is (GEMM A / A2 operands
    and split outputs, norm and attention outputs); only the large-problem kernel (M >= 2048, N >= 512, interleaved weights)
    can consume it."""

    def __init__(self, rows: int, cols: int, device):
        assert cols % 32 == 0
        self.rows, self.cols = rows, cols
        self.buf = torch.empty(rows, 2 * cols, dtype=torch.float16, device=device)

    def rows_view(self, r0: int, r1: int) -> "SplitIL":
        """The same storage restricted to rows [r0, r1) (a contiguous slice: rows are whole interleaved lines)."""
        v = SplitIL.__new__(SplitIL)
        v.rows, v.cols, v.buf = r1 - r0, self.cols, self.buf[r0:r1]
        return v

    def dense(self):
        """(hi, lo) as ordinary [rows, cols] tensors (copies; tests)."""
        b = self.buf.view(self.rows, self.cols // 32, 2, 32)
        return b[:, :, 0].reshape(self.rows, self.cols), b[:, :, 1].reshape(self.rows, self.cols)


def _pair(x, rows=None, cols=None):
    """(hi_ptr, lo_ptr, ld) of a split pair: a (hi, lo) tuple (lo may be None) or a SplitIL."""
    if isinstance(x, SplitIL):
        assert (rows is None or x.rows == rows) and (cols is None or x.cols == cols), "SplitIL shape mismatch"
        return x.buf.data_ptr(), x.buf.data_ptr() + 64, 2 * x.cols
    hi, lo = x
    assert hi.dtype == torch.float16 and hi.stride(-1) == 1 and hi.is_cuda
    assert (rows is None or hi.shape[0] == rows) and (cols is None or hi.shape[-1] == cols)
    assert lo is None or (lo.dtype == torch.float16 and lo.shape == hi.shape and lo.stride() == hi.stride())
    return hi.data_ptr(), _p(lo), hi.stride(0) if hi.ndim == 2 else hi.shape[-1]


# cvx_gemm_split_io.flags of every interleaved-operand GEMM (dev A/B, read here, never inside the library):
# CVX_GEMM_MEDIUM_AUTO=0: 2048 rows and more always on the large-problem kernel (flag 16), =force: always on the medium one (8);
# default: the library picks by how full the large kernel's last round of tiles would be.  Flag 4: one tile per block (large kernel).
_GEMM_FLAGS = {"0": 16, "force": 8}.get(_os.environ.get("CVX_GEMM_MEDIUM_AUTO", "1"), 0)
_GEMM_FLAGS |= int(_os.environ.get("CVX_GEMM_FLAGS_EXTRA", "0"), 0)      # dev builds (-DCVX_DEV_FLAGS): 0x10000 no K slices, 0x20000 at most two
_TL = threading.local()          # per-thread additions to the flags (gemm_flags): a schedule that pins a kernel must not leak into other threads


@contextmanager
def gemm_flags(extra: int):
    """with ops.gemm_flags(16): every GEMM THIS THREAD launches inside the block carries the extra cvx_gemm_split_io.flags bits
    (16 = CVX_GEMM_FLAG_NO_MEDIUM pins the large-problem kernel, 64 / 128 its tile height: A/B measurements and the bit-identity
    tests).  Thread-local: another host thread's solve is not affected (round-4 advice)."""
    old = getattr(_TL, "flags", 0)
    _TL.flags = old | int(extra)
    try:
        yield
    finally:
        _TL.flags = old

_SPLITK_WS: dict = {}


def il_min_rows() -> int:
    """Row count from which GEMM A operands are kept as INTERLEAVED pairs (SplitIL: the medium- and large-problem kernels)."""
    return 128


def _splitk_workspace(device) -> torch.Tensor:
    """Caller-owned scratch of cvx_gemm_f16x3's split-K path: 4 x 2048 x 4096 floats per (device, stream) - two host threads on
    one device run their GEMMs on different streams and must not share partial sums - allocated once and never re-allocated
    (its address is baked into captured HIP graphs; launches inside a capture use the scratch of the stream that captures)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = torch.cuda.current_stream(idx)
    key = (idx, _CAPTURE_OWNER.get((idx, st.cuda_stream), st.cuda_stream))
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.empty(4 * 2048 * 4096, dtype=torch.float32, device=dev)
        _SPLITK_WS[key] = ws
    return ws


class _CaptureGate:
    """Host threads vs stream capture.  HIP (ROCm 7.2) refuses device-wide synchronisation and synchronous copies from ANY host
    thread while one thread captures a stream: thread-local capture mode does not shield the other threads the way CUDA's does
    (measured: torch.cuda.synchronize() / .item() in thread B fail with hipErrorStreamCaptureUnsupported, thread A's capture
    ends with hipErrorStreamCaptureInvalidated, sometimes the process aborts).  So the package serialises its captures against
    its own entry points: an entry point (model upload, a solve, a vocoder call, a text2semantic decode) holds the gate SHARED
    for its whole host side, a capture (once per shape and stream) holds it EXCLUSIVE.  Steady state - graph replays from many
    threads - is unaffected (shared holders do not exclude each other).  A thread that holds the gate shared and reaches a
    capture gives its share back while it waits, so two such threads cannot deadlock; new shared entries queue behind a
    waiting capture so that it cannot starve.  Torch calls a caller makes from its OWN other threads are outside this gate."""

    def __init__(self):
        self._c = threading.Condition()
        self._readers = 0
        self._writer = False
        self._waiting = 0
        self._tl = threading.local()

    @contextmanager
    def shared(self):
        tl = self._tl
        d = getattr(tl, "depth", 0)
        outer = d == 0 and not getattr(tl, "excl", False)
        if outer:
            with self._c:
                while self._writer or self._waiting:
                    self._c.wait()
                self._readers += 1
        tl.depth = d + 1
        try:
            yield
        finally:
            tl.depth = d
            if outer:
                with self._c:
                    self._readers -= 1
                    self._c.notify_all()

    @contextmanager
    def exclusive(self):
        tl = self._tl
        if getattr(tl, "excl", False):
            yield
            return
        held = getattr(tl, "depth", 0) > 0
        with self._c:
            if held:
                self._readers -= 1
            self._waiting += 1
            self._c.notify_all()
            while self._writer or self._readers > 0:
                self._c.wait()
            self._waiting -= 1
            self._writer = True
        tl.excl = True
        try:
            yield
        finally:
            tl.excl = False
            with self._c:
                self._writer = False
                if held:
                    self._readers += 1
                self._c.notify_all()


CAPTURE_GATE = _CaptureGate()


def gated(fn):
    """Decorator: run the entry point with the capture gate held shared (see _CaptureGate)."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with CAPTURE_GATE.shared():
            return fn(*args, **kwargs)
    return wrapper


_CAPTURE_OWNER: dict = {}        # (device, side / capture stream) -> the stream whose call it serves (saturation_share)


def _sp(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of an activation pre-scale: a one-element fp32 CUDA tensor (a view into a scale table), or None = 1."""
    if t is None or isinstance(t, int):          # (an int is a raw device address from a pre-computed pointer table)
        return t
    assert t.is_cuda and t.dtype == torch.float32 and t.numel() == 1
    return t.data_ptr()


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias=None, act=ACT_NONE, residual=None,
         a2: Optional[torch.Tensor] = None, rope=None, rope_cols: int = 0, w_split=None,
         a_split=None, a2_split=None, out_split=None, write_f32: bool = True, vt_split=None, w_il=None,
         a_scale=None, c_scale=None, vt_scale=None, norm=None, c_gamma=None, c_rowsq=None, a_row_scale=None,
         res_split=None, res_scale=None, a2_scale=None) -> torch.Tensor:
    """out[M,N] = epilogue([a | a2] @ w[:, :K].T).  `a`, `a2`, `out`, `residual` are 2-D with unit
    inner stride (row stride may exceed the width); `w` may be a column-slice view of a wider matrix.
    w_split = (hi, lo) fp16 halves from split_f16(w): run the split-precision f16x3 MFMA kernel instead
    of the fp32 one (needs K % 32 == 0; small-M problems stay on the fp32 kernel).
    a_split / a2_split = (hi, lo) fp16 copies of a / a2 (then every tile arrives by LDS-DMA; `a` is only used for
    its shape); out_split = (hi, lo) receives a split copy of the result; write_f32=False skips the fp32 store.
    Any `lo` may be None: w_split = (hi, None, 1/scale) selects the single-term fp16 kernel (plain fp16 operands,
    fp32 accumulate; needs a_split and K % 64 == 0), whose inputs / outputs only carry the hi halves.
    a_scale / c_scale / vt_scale: one-element fp32 CUDA tensors (powers of two) - the pre-scale the producer of a_split (and
    a2_split) applied, and the pre-scales to apply to out_split / vt_split (see cvx_gemm_split_io in the header).
    norm = dict(gamma, beta (or None), out_split, scale (device pre-scale or None)[, eps]): the AdaptiveRMSNorm / RMSNorm of the
    rows of `out` as part of the same call (cvx_gemm_f16x3_norm): its split pair feeds the next GEMM; on the split-K path the
    norm rides in the reduction (no launch, no re-read).
    Deferred norm (cvx_gemm_split_io, version 105; interleaved operands only): c_gamma [N] multiplies the columns of out_split,
    c_rowsq [M, >= N/64] receives the rows' sums of squares per 64-column slice (rownorm_scale turns them into one factor per
    row); a_row_scale [M] multiplies the rows of the accumulators of the GEMM that consumes that pair; res_split (+ res_scale) gives
    the residual as a split pair (then residual must be None; may be the very pair out_split names)."""
    _chk_f32(a, w, out, bias, residual, a2)
    M = a.shape[0]
    N = w.shape[0]
    k1 = a.shape[1]
    K = k1 + (a2.shape[1] if a2 is not None else 0)
    assert w.shape[1] == K, f"gemm: K mismatch ({w.shape[1]} vs {K})"
    assert a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1 and out.shape == (M, N)
    g = GemmArgs()
    g.A, g.lda = a.data_ptr(), a.stride(0)
    if a2 is not None:
        assert a2.shape[0] == M and a2.stride(1) == 1
        g.A2, g.lda2, g.K1 = a2.data_ptr(), a2.stride(0), k1
    else:
        g.A2, g.lda2, g.K1 = None, 0, 0
    g.W, g.ldw = w.data_ptr(), w.stride(0)
    g.C, g.ldc = out.data_ptr(), out.stride(0)
    g.bias = _p(bias)
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
        g.residual, g.ldr = residual.data_ptr(), residual.stride(0)
    else:
        g.residual, g.ldr = None, 0
    g.M, g.N, g.K, g.act = M, N, K, act
    if rope is not None:
        cos, sin = rope
        _chk_f32(cos, sin)
        g.rope_cos, g.rope_sin, g.rope_T, g.rope_cols = cos.data_ptr(), sin.data_ptr(), cos.shape[0], rope_cols
    else:
        g.rope_cos, g.rope_sin, g.rope_T, g.rope_cols = None, None, 0, 0
    if isinstance(a_split, SplitIL):
        assert w_il is not None, "an interleaved A operand needs interleaved weights"
    # interleaved weights go with an interleaved A: the large- (M >= 2048, N >= 512) and medium-problem kernels
    if w_il is not None and w_split is not None and w_split[1] is not None and isinstance(a_split, SplitIL):
        il, inv_il = w_il                      # interleaved [N, 2K] copy of the same split weight (split_f16_interleaved)
        assert il.dtype == torch.float16 and il.shape == (N, 2 * K) and il.is_contiguous() and il.is_cuda
        use_il = True
    else:
        use_il = False
    if w_split is not None and K % 32 == 0 and (M > 64 or a_split is not None or out_split is not None):
        hi, lo, inv_scale = w_split
        ensure_saturation_bound()
        assert hi.dtype == torch.float16 and hi.shape == (N, K) and hi.stride(1) == 1 and hi.is_cuda
        assert lo is None or (lo.dtype == torch.float16 and lo.shape == (N, K) and lo.stride() == hi.stride() and lo.is_cuda)
        g.ldw = hi.stride(0)                       # rows may be padded (row stride > K)
        io = GemmSplitIO()
        io.write_f32 = 1 if write_f32 else 0
        io.a_scale_dev, io.c_scale_dev, io.vt_scale_dev = _sp(a_scale), _sp(c_scale), _sp(vt_scale)
        if a_split is not None and M <= 2048 and rope is None:          # small problems may split K (see the header)
            ws = _splitk_workspace(a.device)
            io.workspace, io.workspace_floats = ws.data_ptr(), ws.numel()
        if a_split is not None:
            io.A_hi, io.A_lo, io.lda_h = _pair(a_split, M, k1)
            if a2 is not None:
                assert a2_split is not None, "pre-split A needs a pre-split A2 as well"
                assert isinstance(a2_split, SplitIL) == isinstance(a_split, SplitIL)
                io.A2_hi, io.A2_lo, io.lda2_h = _pair(a2_split, M, K - k1)
        if out_split is not None:
            io.C_hi, io.C_lo, io.ldc_h = _pair(out_split, M, rope_cols if vt_split is not None else N)
        if a2_scale is not None:
            assert a2 is not None and a_scale is not None
            io.a2_scale_dev = _sp(a2_scale)
        if res_split is not None:
            assert residual is None
            io.R_hi, io.R_lo, io.ldr_h = _pair(res_split, M, N)
            io.r_scale_dev = _sp(res_scale)
        if c_gamma is not None or c_rowsq is not None or a_row_scale is not None:
            _chk_f32(c_gamma, c_rowsq, a_row_scale)
            if c_gamma is not None:
                assert c_gamma.numel() == N and c_gamma.stride(-1) == 1 and out_split is not None
                io.c_gamma_dev = c_gamma.data_ptr()
            if c_rowsq is not None:
                assert c_rowsq.ndim == 2 and c_rowsq.shape[0] >= M and c_rowsq.stride(1) == 1 and c_rowsq.shape[1] * 64 >= N
                io.c_rowsq, io.c_rowsq_ld = c_rowsq.data_ptr(), c_rowsq.stride(0)
            if a_row_scale is not None:
                assert a_row_scale.numel() >= M and a_row_scale.stride(-1) == 1
                io.a_row_scale_dev = a_row_scale.data_ptr()
        if vt_split is not None:          # QKV mode: v columns transposed per (sequence, head) for the f16x3 attention
            vh, vl = vt_split
            assert vh.dtype == torch.float16 and vh.is_contiguous() and (vl is None or (vl.is_contiguous() and vh.shape == vl.shape))
            io.Vt_hi, io.Vt_lo, io.vt_ld = vh.data_ptr(), _p(vl), vh.shape[-1]
        if _lib._DEBUG_SYNC:
            import sys
            print(f"[cvx] gemm_f16x3 M={M} N={N} K={K} k1={g.K1} lo={lo is not None} a_split={a_split is not None} rope={rope is not None} "
                  f"out_split={out_split is not None} vt={vt_split is not None} ws={bool(io.workspace)}", file=sys.stderr, flush=True)
        w_hi_ptr, w_lo_ptr = hi.data_ptr(), _p(lo)
        if use_il:
            io.flags = _GEMM_FLAGS | getattr(_TL, "flags", 0)
            io.w_interleaved, g.ldw = 1, 2 * K
            w_hi_ptr, w_lo_ptr, inv_scale = il.data_ptr(), il.data_ptr() + 64, inv_il
        if norm is not None:
            nm = GemmNorm()
            _chk_f32(norm["gamma"], norm.get("beta"))
            assert norm["gamma"].stride(-1) == 1 and norm["gamma"].numel() == N and out.is_contiguous() and write_f32
            nm.gamma, nm.beta = norm["gamma"].data_ptr(), _p(norm.get("beta"))
            nm.Y_hi, nm.Y_lo, nm.ldy_h = _pair(norm["out_split"], M, N)
            nm.y_scale_dev = _sp(norm.get("scale"))
            nm.scale, nm.eps = float(N) ** 0.5, float(norm.get("eps", 1e-12))
            _lib.check(_lib.load().cvx_gemm_f16x3_norm(C.byref(g), w_hi_ptr, w_lo_ptr, inv_scale, C.byref(io), C.byref(nm), _stream()),
                       "cvx_gemm_f16x3_norm")
            return out
        _lib.check(_lib.load().cvx_gemm_f16x3(C.byref(g), w_hi_ptr, w_lo_ptr, inv_scale, C.byref(io), _stream()),
                   "cvx_gemm_f16x3")
        return out
    assert norm is None, "a fused norm needs the f16x3 kernel"
    assert a_split is None and out_split is None and write_f32, "split I/O needs the f16x3 kernel (w_split, K % 32 == 0, M > 64)"
    assert a_scale is None and c_scale is None and vt_scale is None
    assert c_gamma is None and c_rowsq is None and a_row_scale is None and res_split is None and a2_scale is None, "a deferred norm needs the f16x3 kernel"
    _lib.check(_lib.load().cvx_gemm_bias_act_f32(C.byref(g), _stream()), "cvx_gemm_bias_act_f32")
    return out


def split_act_f16(x: torch.Tensor, hi=None, lo: Optional[torch.Tensor] = None, scale=None):
    """(hi, lo) fp16 halves of an fp32 activation tensor - for GEMMs that take A pre-split.
    With `hi` given and lo=None only the (saturating) fp16 cast is written; hi may be a SplitIL (interleaved pair).
    scale: one-element fp32 CUDA tensor, the power-of-two pre-scale (the consumer GEMM's a_scale); None = 1."""
    _chk_f32(x)
    assert x.is_contiguous()
    ensure_saturation_bound()
    if isinstance(hi, SplitIL):
        h, l, _ = _pair(hi, x.shape[0], x.shape[1])
        _lib.check(_lib.load().cvx_split_f16_dev(x.data_ptr(), h, l, x.numel(), 1.0, _sp(scale), _stream()), "cvx_split_f16")
        return hi
    if hi is None:
        hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
        lo = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().cvx_split_f16_dev(x.data_ptr(), hi.data_ptr(), _p(lo), x.numel(), 1.0, _sp(scale), _stream()),
               "cvx_split_f16")
    return hi, lo


def split_f16(w: torch.Tensor, with_lo: bool = True):
    """(hi, lo, 1/scale): fp16 halves of w*scale, hi = fp16(w*scale), lo = fp16(w*scale - hi), with scale the power
    of two that brings max|w| to [2^13, 2^14) so the lo halves stay out of the fp16 subnormal range.
    Load-time weight packing.  with_lo=False: (hi, None, 1/scale) for the single-term fp16 mode."""
    import math
    _chk_f32(w)
    w = w.contiguous()
    amax = float(w.abs().max())
    scale = 2.0 ** (13 - math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
    hi = torch.empty(w.shape, dtype=torch.float16, device=w.device)
    lo = torch.empty(w.shape, dtype=torch.float16, device=w.device) if with_lo else None
    ensure_saturation_bound()
    _lib.check(_lib.load().cvx_split_f16(w.data_ptr(), hi.data_ptr(), _p(lo), w.numel(), scale, _stream()),
               "cvx_split_f16")
    return hi, lo, 1.0 / scale


def split_f16_interleaved(w_split):
    """(hi, lo, 1/scale) from split_f16 -> (il [N, 2K], 1/scale) with il[n] = [hi 0:32 | lo 0:32 | hi 32:64 | lo 32:64 | ...]:
    a K-step of one row becomes one 128-byte cache line for the large-problem GEMM's DMA.  Load-time packing."""
    hi, lo, inv = w_split
    N, K = hi.shape
    assert K % 32 == 0 and lo is not None
    il = torch.stack((hi.view(N, K // 32, 32), lo.view(N, K // 32, 32)), dim=2).reshape(N, 2 * K).contiguous()
    return il, inv


def adarmsnorm(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], out: Optional[torch.Tensor],
               rows_per_group: Optional[int] = None, eps: float = 1e-12, out_split=None, split_scale=None):
    """out_split = (hi, lo) fp16 tensors: also (or, with out=None, only) write the result as a split pair (times the
    one-element device tensor split_scale, if given)."""
    _chk_f32(x, gamma, beta, out)
    assert x.is_contiguous() and (out is None or out.is_contiguous()) and gamma.stride(-1) == 1
    D = x.shape[-1]
    rows = x.numel() // D
    oh, ol = (None, None)
    if out_split is not None:
        ensure_saturation_bound()
        oh, ol, ld = _pair(out_split, rows if isinstance(out_split, SplitIL) else None, D)
        assert isinstance(out_split, SplitIL) or (out_split[0].is_contiguous() and out_split[0].numel() == x.numel())
    rpg = rows if rows_per_group is None else rows_per_group
    _lib.check(_lib.load().cvx_adarmsnorm_scaled_f32(x.data_ptr(), gamma.data_ptr(), _p(beta), _p(out), oh, ol, rows, D, rpg,
                                                     float(D) ** 0.5, eps, _sp(split_scale), _stream()), "cvx_adarmsnorm_f32")
    return out if out is not None else out_split


def split_f16_colscale_il(w: torch.Tensor, colscale: torch.Tensor, set_scale: Optional[torch.Tensor], scale: float, out: torch.Tensor) -> torch.Tensor:
    """out[s] = interleaved split pair of w * colscale[s][None, :] * set_scale[s] * scale for every set s (cvx_split_f16_colscale_il):
    w [N, K] fp32, colscale [n_sets, K] (row stride free), set_scale [n_sets] (stride free) or None, out [n_sets, N, 2K] fp16."""
    ensure_saturation_bound()
    _chk_f32(w, colscale, set_scale)
    N, K = w.shape
    n_sets = colscale.shape[0]
    assert w.stride(1) == 1 and colscale.shape[1] == K and colscale.stride(1) == 1 and out.dtype == torch.float16 and out.is_contiguous()
    assert tuple(out.shape) == (n_sets, N, 2 * K) and (set_scale is None or set_scale.shape[0] == n_sets)
    _lib.check(_lib.load().cvx_split_f16_colscale_il(w.data_ptr(), w.stride(0), N, K, colscale.data_ptr(), colscale.stride(0), _p(set_scale),
                                                     set_scale.stride(0) if set_scale is not None else 0, n_sets, scale, out.data_ptr(), _stream()),
               "cvx_split_f16_colscale_il")
    return out


def rownorm_scale(rowsq: torch.Tensor, rows: int, parts: int, out: torch.Tensor, scale: float, eps: float = 1e-12) -> torch.Tensor:
    """out[r] = scale / max(sqrt(sum(rowsq[r, :parts])), eps) (cvx_rownorm_scale_f32): the per-row factor of a deferred norm."""
    _chk_f32(rowsq, out)
    assert rowsq.ndim == 2 and rowsq.stride(1) == 1 and rowsq.shape[0] >= rows and rowsq.shape[1] >= parts and out.numel() >= rows and out.stride(-1) == 1
    _lib.check(_lib.load().cvx_rownorm_scale_f32(rowsq.data_ptr(), rows, parts, rowsq.stride(0), scale, eps, out.data_ptr(), _stream()),
               "cvx_rownorm_scale_f32")
    return out


# Saturation flags are CALLER-OWNED: one int32 of device memory per (device, stream), bound to the stream with
# The kernels a stream runs OR bit 0 into the flag of the stream's launch context (two threads / streams on one device never see or
# clear each other's).  Allocated once and never freed (their addresses are baked into captured HIP graphs).
_SAT_FLAGS: dict = {}


def saturation_flag(stream: Optional["torch.cuda.Stream"] = None) -> torch.Tensor:
    """The OWN flag of `stream` (default: the current stream), allocated and put into its launch context on first use.  A flag that
    was only lent to the stream (saturation_share: torch hands stream handles out of a small pool, so a stream a caller just created
    may be the very side / capture stream some earlier call borrowed) is replaced: whoever resets or queries a stream owns its flag."""
    st = stream if stream is not None else torch.cuda.current_stream()
    key = (st.device.index, st.cuda_stream)
    ent = _SAT_FLAGS.get(key)
    if ent is None or ent[1] != key:
        f = torch.zeros(1, dtype=torch.int32, device=st.device)
        torch.cuda.current_stream(st.device).synchronize()        # (the zero fill is on the current stream; `stream` may be another)
        _SAT_FLAGS[key] = (f, key)
        _CAPTURE_OWNER.pop(key, None)
        ctx_of(st).sat_flag = f.data_ptr()
        return f
    return ent[0]


def saturation_share(src: "torch.cuda.Stream", dst: "torch.cuda.Stream") -> None:
    """Kernels launched on `dst` report into the flag of `src` until further notice, and their persistent grids are sized for the
    CUs `src` owns: the capture stream of a HIP graph - and any side stream a call opens - belongs to the call that runs on `src`
    (call it before every such use)."""
    f = saturation_flag(src)
    skey = (src.device.index, src.cuda_stream)
    key = (dst.device.index, dst.cuda_stream)
    if key == skey:
        return
    _CAPTURE_OWNER[key] = src.cuda_stream
    ctx_of(dst).n_cus = ctx_of(src).n_cus
    ent = _SAT_FLAGS.get(key)
    if ent is None or ent[0] is not f:
        _SAT_FLAGS[key] = (f, skey)
        ctx_of(dst).sat_flag = f.data_ptr()


def ensure_saturation_bound() -> None:
    """A call that writes split pairs refuses a launch context without a saturation flag (CVX_EINVAL; a C caller can waive the
    bookkeeping with CVX_CTX_NO_SATURATION_FLAG).  The Python front ends never waive it: every one of them that launches a split-pair
    kernel makes sure the current stream's context has a flag - allocated here on first use outside a capture; inside a capture a
    stream without one is an error (the stream that captures must have been shared with the calling stream: saturation_share)."""
    st = torch.cuda.current_stream()
    if (st.device.index, st.cuda_stream) in _SAT_FLAGS:
        return
    if torch.cuda.is_current_stream_capturing():
        raise _lib.CovomixHipError("a split-precision kernel is being captured on a stream without a saturation flag: call "
                                   "ops.saturation_share(calling_stream, capture_stream) before the capture")
    saturation_flag(st)


# ---------------------------------------------------------------- CU-partitioned streams (include/covomix_hip.h, round 5)
class CUPartition:
    """Two streams of one device on DISJOINT sets of compute units: `main` (the acoustic solve and the vocoder) and `side` (the
    text2semantic decode of the next batch).  Bit k of a HIP CU mask names CU k // 8 of XCD k % 8 and consecutive indices of an XCD go
    round its four shader engines (tools/archive/cu_mask_probe.hip), so `side` takes the top `side_per_xcd` indices of every XCD
    (side_per_xcd a multiple of 4: every shader engine keeps the same number of CUs, which one-block-per-CU kernels need to be
    co-resident - tools/archive/cu_mask_probe2.hip) and `main` the rest.  The streams live as long as the process."""

    def __init__(self, device, side_per_xcd: int = 4):
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        n = torch.cuda.get_device_properties(idx).multi_processor_count
        if side_per_xcd <= 0 or side_per_xcd % 4 or n % 8 or 8 * side_per_xcd >= n:
            raise ValueError(f"side_per_xcd must be a positive multiple of 4 below {n // 8} (CUs per XCD), got {side_per_xcd}")
        self.device = torch.device("cuda", idx)
        self.n_side, self.n_main = 8 * side_per_xcd, n - 8 * side_per_xcd
        words = (n + 31) // 32
        lib = _lib.load()

        def make(bits):
            m = (C.c_uint32 * words)()
            for b in bits:
                m[b // 32] |= 1 << (b % 32)
            out = C.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(lib.cvx_stream_create_cu_mask(m, words, C.byref(out)), "cvx_stream_create_cu_mask")
            st = torch.cuda.ExternalStream(out.value, device=self.device)
            ctx_of(st).n_cus = len(bits)               # (what the persistent grids and the GEMM kernel choice are sized from)
            return st
        self.main = make(range(0, self.n_main))
        self.side = make(range(self.n_main, n))


_CU_PARTITIONS: dict = {}


def cu_partition(device=None, side_per_xcd: int = 4) -> CUPartition:
    """The (cached) CUPartition of a device."""
    dev = torch.device(device if device is not None else "cuda")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, side_per_xcd)
    if key not in _CU_PARTITIONS:
        if not _CU_PARTITIONS:
            import atexit
            atexit.register(_destroy_partitions)      # while the HIP runtime is still up (a masked stream that outlives it crashed rocprofv3's finaliser)
        _CU_PARTITIONS[key] = CUPartition(torch.device("cuda", idx), side_per_xcd)
    return _CU_PARTITIONS[key]


def _destroy_partitions() -> None:
    lib = _lib.load()
    for part in list(_CU_PARTITIONS.values()):
        try:
            with torch.cuda.device(part.device):
                torch.cuda.synchronize()
                # torch's pinned-memory allocator keeps events of the streams its blocks were last copied on (the decode's state
                # record travels by non_blocking copies on the side stream): give the blocks back BEFORE their stream goes, or the
                # allocator touches a dead stream at process exit (SIGSEGV, tools/archive/cu_mask_exit_probe.py)
                # (without that call the streams are left to process teardown: destroying them under live pinned blocks is the crash)
                if hasattr(torch._C, "_host_emptyCache"):
                    torch._C._host_emptyCache()
                    for st in (part.main, part.side):
                        st.synchronize()
                        _CTX.pop((st.device.index, st.cuda_stream), None)
                        _SAT_FLAGS.pop((st.device.index, st.cuda_stream), None)
                        lib.cvx_stream_destroy(st.cuda_stream)
        except Exception:               # noqa: BLE001 - interpreter shutdown: nothing left to report to
            pass
    _CU_PARTITIONS.clear()


def is_partition_stream(stream: Optional["torch.cuda.Stream"] = None) -> bool:
    """Is `stream` (default: the current one) one of the CU-masked streams of a CUPartition?"""
    st = stream if stream is not None else torch.cuda.current_stream()
    return any(st.cuda_stream in (p.main.cuda_stream, p.side.cuda_stream) for p in _CU_PARTITIONS.values())


def stream_cus(stream: Optional["torch.cuda.Stream"] = None) -> int:
    """CUs the library sizes persistent grids for on `stream` (default: the current stream)."""
    st = stream if stream is not None else torch.cuda.current_stream()
    n = ctx_of(st).n_cus
    return int(n) if n > 0 else torch.cuda.get_device_properties(st.device).multi_processor_count


def clock_stamps() -> torch.Tensor:
    """[2048, 2] int64 (enqueued on the current stream): per (XCD, CU) slot that CU's shader-clock cycle counter and the 100 MHz real-time
    counter (zero where no block landed).  Pair two calls slot by slot with clock_from_stamps."""
    out = torch.zeros(2048, 2, dtype=torch.int64, device=torch.cuda.current_device())
    _lib.check(_lib.load().cvx_clock_stamps(out.data_ptr(), _stream()), "cvx_clock_stamps")
    return out


def clock_from_stamps(c0: torch.Tensor, c1: torch.Tensor) -> dict:
    """Shader clock of the (busy) region between two clock_stamps() calls: cycles counted / real time on every CU that holds both stamps,
    averaged per XCD -> {mhz (mean over XCDs), xcd_mhz (8 values), cus (CUs paired)}; {} when nothing could be paired."""
    a, b = c0.cpu(), c1.cpu()
    ok = (a[:, 1] != 0) & (b[:, 1] != 0)
    d = (b - a).double()
    ok &= (d[:, 1] > 0) & (d[:, 0] > 0)
    if not bool(ok.any()):
        return {}
    mhz = torch.where(ok, d[:, 0] / d[:, 1].clamp_min(1.0) * 100.0, torch.zeros_like(d[:, 0])).view(8, 256)      # the real-time counter: 100 MHz
    n = ok.view(8, 256).sum(dim=1)
    per_xcd = [float(mhz[x].sum() / n[x]) for x in range(8) if int(n[x]) > 0]
    return {"mhz": sum(per_xcd) / len(per_xcd), "xcd_mhz": per_xcd, "cus": int(ok.sum())}


def saturation_reset() -> None:
    """Clear the current stream's sticky saturation flag (enqueued on that stream; no host synchronisation)."""
    saturation_flag()
    _lib.check(_lib.load().cvx_saturation_flag_reset(_stream()), "cvx_saturation_flag_reset")


def saturation_query(reset: bool = True) -> int:
    """The current stream's sticky saturation flag after everything enqueued so far on it (synchronises that stream):
    non-zero = some split-pair store since the last reset had to clamp (or a softmax normaliser was not finite) - the
    split-precision result must not be trusted."""
    saturation_flag()
    v = C.c_uint32(0)
    _lib.check(_lib.load().cvx_saturation_flag_query(C.byref(v), 1 if reset else 0, _stream()), "cvx_saturation_flag_query")
    return int(v.value)


_SAT_DEFER = threading.local()


def saturation_checked() -> bool:
    """Do the entry points check the saturation flag themselves (reset before, blocking read after)?  False under
    CVX_SAT_CHECK=0 and inside `saturation_deferred()`."""
    return _os.environ.get("CVX_SAT_CHECK", "1") == "1" and getattr(_SAT_DEFER, "depth", 0) == 0


class saturation_deferred:
    """with ops.saturation_deferred() as guard: ...several entry-point calls on the current stream...
    ONE reset before and ONE flag read after the whole block instead of a blocking read per call (round-3 advice: the per-call
    reads are host synchronisations between acoustic model and vocoder, and they keep the entry points out of a caller's own
    stream capture).  Inside the block the entry points return their split-precision results unchecked; afterwards
    `guard.flagged` says whether ANY of them saturated - the caller must then discard the block's results and repeat the calls
    outside the block (where each call checks itself and a flagged one re-runs in fp32 or raises).  `read=False`: no read at
    exit either (a caller that captures the block into its own graph reads the flag with ops.saturation_query() after the
    replay)."""

    def __init__(self, read: bool = True):
        self.read, self.flagged = read, None

    def __enter__(self):
        if getattr(_SAT_DEFER, "depth", 0) == 0 and _os.environ.get("CVX_SAT_CHECK", "1") == "1":
            saturation_reset()
        _SAT_DEFER.depth = getattr(_SAT_DEFER, "depth", 0) + 1
        return self

    def __exit__(self, et, ev, tb):
        _SAT_DEFER.depth -= 1
        if et is None and self.read and _SAT_DEFER.depth == 0 and _os.environ.get("CVX_SAT_CHECK", "1") == "1":
            self.flagged = bool(saturation_query())
        return False


def h2d(t: torch.Tensor, device, dtype=None) -> torch.Tensor:
    """Host tensor -> device through pinned memory, asynchronously on the current stream.  A copy from PAGEABLE memory makes the host
    wait for everything the stream still has to do before it (the runtime stages it behind the queue and waits): one small tensor
    made with torch.tensor(..., device=...) in the middle of a call is enough to keep the host from running ahead of the device."""
    if t.device.type != "cpu":
        return t.to(device=device, dtype=dtype) if dtype is not None else t.to(device)
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def saturation_snapshot():
    """The current stream's saturation flag as it stands after everything enqueued so far, WITHOUT waiting: an asynchronous copy into a
    pinned int32 tensor (read element 0 once an event recorded after this call has completed).  For callers that run the host ahead
    of the device - `with saturation_deferred(read=False): ...calls...; snap = saturation_snapshot()` - and look at the batch later.
    None under CVX_SAT_CHECK=0."""
    if _os.environ.get("CVX_SAT_CHECK", "1") != "1":
        return None
    snap = torch.zeros(1, dtype=torch.int32).pin_memory()
    snap.copy_(saturation_flag(), non_blocking=True)
    return snap


class Ragged:
    """A packed batch of sequences of different length: sequence i owns rows [cu[i], cu[i+1]) of every [M, width] tensor.
    cu: int32 CUDA tensor [n + 1]; lengths: the python list (host side: grid sizes, slicing)."""

    def __init__(self, lengths, device, repeat: int = 1):
        """repeat = 2: the same sequences twice (conditional rows, then the null-branch rows of CFG)."""
        self.lengths = [int(t) for t in lengths] * repeat
        assert self.lengths and min(self.lengths) > 0
        self.n, self.max_T, self.M = len(self.lengths), max(self.lengths), sum(self.lengths)
        cu = [0]
        for t in self.lengths:
            cu.append(cu[-1] + t)
        self.cu_host = cu
        self.cu = h2d(torch.tensor(cu, dtype=torch.int32), device)

    def positions(self) -> torch.Tensor:
        """Position of every packed row inside its own sequence (fp32, on the device)."""
        cu = self.cu.to(torch.int64)
        rows = torch.arange(self.M, device=self.cu.device)
        seq = torch.searchsorted(cu[1:], rows, right=True)
        return (rows - cu[seq]).to(torch.float32)


def attention(qkv: torch.Tensor, out: Optional[torch.Tensor], Bt: int, T: int, H: int, scale: float, out_split=None,
              ragged: Optional[Ragged] = None):
    """ragged: packed batch (Bt, T are then ignored: rows = ragged.M, keys restricted to the own sequence)."""
    _chk_f32(qkv, out)
    rows = ragged.M if ragged is not None else Bt * T
    assert qkv.is_contiguous() and (out is None or out.is_contiguous())
    assert qkv.numel() == rows * 3 * H * 64 and (out is None or out.numel() == rows * H * 64)
    oh, ol = (None, None)
    if out_split is not None:
        ensure_saturation_bound()
        oh, ol, _ = _pair(out_split, rows if isinstance(out_split, SplitIL) else None, H * 64)
        assert isinstance(out_split, SplitIL) or (out_split[0].is_contiguous() and out_split[0].numel() == rows * H * 64)
    if ragged is not None:
        _lib.check(_lib.load().cvx_attention_varlen_f32(qkv.data_ptr(), _p(out), oh, ol, ragged.cu.data_ptr(), ragged.n, ragged.max_T,
                                                        H, scale, _stream()), "cvx_attention_varlen_f32")
    else:
        _lib.check(_lib.load().cvx_attention_f32(qkv.data_ptr(), _p(out), oh, ol, Bt, T, H, scale, _stream()),
                   "cvx_attention_f32")
    return out if out is not None else out_split


def vt_frame_slots(T: int, device=None) -> torch.Tensor:
    """Column of frame t in the V^T buffers of the f16x3 attention: bits 2 and 3 of t swapped (four-frame groups in the
    order 0, 2, 1, 3 inside every 16 frames) - what the to_qkv epilogue writes and the attention kernel reads."""
    t = torch.arange(T, device=device)
    return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)


def attention_f16x3(qk_split, vt_split, out: Optional[torch.Tensor], Bt: int, T: int, H: int, scale: float, out_split=None,
                    qk_scale=None, v_scale=None, out_scale=None, ragged: Optional[Ragged] = None):
    """Split-precision attention on the pairs written by gemm(..., out_split=qk_split, vt_split=vt_split).
    (hi, None) pairs select the single-term fp16 kernel.  qk_scale / v_scale: the pre-scales the producer applied to the
    pairs (c_scale / vt_scale of the to_qkv GEMM); out_scale: pre-scale of out_split (a_scale of the to_out GEMM).
    ragged: packed batch - qk pairs [M, 2*H*64], vt pairs [H*64, >= M rounded up to 32] written by a to_qkv GEMM that ran
    with per-row RoPE tables [M, 32] (rope_T = M); Bt, T are ignored."""
    qh, ql = qk_split
    vh, vl = vt_split
    ensure_saturation_bound()
    assert (ql is None) == (vl is None)
    for t in (qh, ql, vh, vl):
        assert t is None or (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous())
    rows = ragged.M if ragged is not None else Bt * T
    assert qh.shape == (rows, 2 * H * 64) and vh.shape[0] == (H * 64 if ragged is not None else Bt * H * 64)
    Tp = vh.shape[1]
    _chk_f32(out)
    oh, ol = (None, None)
    if out_split is not None:
        oh, ol, _ = _pair(out_split, rows if isinstance(out_split, SplitIL) else None, H * 64)
        assert isinstance(out_split, SplitIL) or (out_split[0].is_contiguous() and out_split[0].numel() == rows * H * 64)
    if ragged is not None:
        _lib.check(_lib.load().cvx_attention_f16x3_varlen(qh.data_ptr(), _p(ql), vh.data_ptr(), _p(vl), _p(out), oh, ol, ragged.cu.data_ptr(),
                                                          ragged.n, ragged.max_T, ragged.M, Tp, H, scale, _sp(qk_scale), _sp(v_scale),
                                                          _sp(out_scale), _stream()), "cvx_attention_f16x3_varlen")
    else:
        _lib.check(_lib.load().cvx_attention_f16x3_scaled(qh.data_ptr(), _p(ql), vh.data_ptr(), _p(vl), _p(out), oh, ol,
                                                          Bt, T, Tp, H, scale, _sp(qk_scale), _sp(v_scale), _sp(out_scale), _stream()),
                   "cvx_attention_f16x3")
    return out if out is not None else out_split


def geglu(h: torch.Tensor, out: torch.Tensor, F: int) -> torch.Tensor:
    """out[r, c] = h[r, c] * gelu(h[r, F + c]) for c < F, zeros in the padding columns (text2semantic.py:154-157)."""
    _chk_f32(h, out)
    assert h.is_contiguous() and out.is_contiguous() and h.shape[-1] == 2 * F and out.shape[-1] >= F and h.shape[0] == out.shape[0]
    _lib.check(_lib.load().cvx_geglu_f32(h.data_ptr(), out.data_ptr(), h.shape[0], F, out.shape[-1], _stream()), "cvx_geglu_f32")
    return out


def dwconv31_gelu_res(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor,
                      Bt: int, T: int, ragged: Optional[Ragged] = None) -> torch.Tensor:
    _chk_f32(x, w, bias, out)
    C_ = x.shape[-1]
    assert x.is_contiguous() and out.is_contiguous() and w.is_contiguous() and w.numel() == C_ * 31
    if ragged is not None:
        assert x.numel() == ragged.M * C_
        _lib.check(_lib.load().cvx_dwconv31_gelu_res_varlen_f32(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                                ragged.cu.data_ptr(), ragged.n, ragged.max_T, C_, _stream()),
                   "cvx_dwconv31_gelu_res_varlen_f32")
        return out
    _lib.check(_lib.load().cvx_dwconv31_gelu_res_f32(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                     Bt, T, C_, _stream()), "cvx_dwconv31_gelu_res_f32")
    return out


def gemm_skinny(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias=None, act: int = ACT_NONE) -> torch.Tensor:
    """out [M, N] = act(a [M <= 32, K] @ w[N, K].T + bias): the weight-streaming kernel (cvx_gemm_skinny_f32)."""
    _chk_f32(a, w, out, bias)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and tuple(out.shape) == (M, N) and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    _lib.check(_lib.load().cvx_gemm_skinny_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(bias), out.data_ptr(), out.stride(0),
                                               M, N, K, act, _stream()), "cvx_gemm_skinny_f32")
    return out


def cfg_combine_axpy(f_c, f_n, y, cond_scale: float, coef: float, out, out2=None, out3=None) -> None:
    _chk_f32(f_c, f_n, y, out, out2, out3)
    n = y.numel()
    assert f_c.numel() == n and out.numel() == n and (f_n is None or f_n.numel() == n)
    for t in (f_c, f_n, y, out, out2, out3):
        assert t is None or t.is_contiguous()
    _lib.check(_lib.load().cvx_cfg_combine_axpy_f32(f_c.data_ptr(), _p(f_n), y.data_ptr(), cond_scale, coef,
                                                    out.data_ptr(), _p(out2), _p(out3), n, _stream()),
               "cvx_cfg_combine_axpy_f32")


def embed_gather(ids: Optional[torch.Tensor], streams: int, table: torch.Tensor, cond: Optional[torch.Tensor],
                 cond_row: Optional[torch.Tensor], cond_dim: int, null_id: int, out: torch.Tensor, M: int) -> torch.Tensor:
    _chk_f32(table, cond, cond_row, out)
    if ids is not None:
        assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == M * streams
    E = table.shape[1]
    assert out.is_contiguous() and out.numel() == M * (streams * E + cond_dim)
    _lib.check(_lib.load().cvx_embed_gather_f32(_p(ids), streams, table.data_ptr(), E, table.shape[0], _p(cond),
                                                _p(cond_row), cond_dim, null_id, out.data_ptr(), M, _stream()),
               "cvx_embed_gather_f32")
    return out


def time_fourier(times: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _chk_f32(times, w, out)
    n, half = times.numel(), w.numel()
    assert out.is_contiguous() and out.numel() == n * 2 * half
    _lib.check(_lib.load().cvx_time_fourier_f32(times.data_ptr(), w.data_ptr(), out.data_ptr(), n, half, _stream()),
               "cvx_time_fourier_f32")
    return out


def _items(dst, items) -> None:
    """Fill a cvx_item_lengths field: items = (int32 CUDA tensor of per-item mel frames, mul, add) or None."""
    if items is None:
        dst.item_len_dev, dst.mul, dst.add = None, 0, 0
        return
    t, mul, add = items
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()
    dst.item_len_dev, dst.mul, dst.add = t.data_ptr(), int(mul), int(add)


def hifigan_conv1d(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *,
                   cout: int, ksize: int, dil: int = 1, pad: int = 0, up: int = 1, in_slope: float = 1.0,
                   res=None, accum=None, out_scale: float = 1.0, items=None) -> torch.Tensor:
    _chk_f32(x, wp, bias, out, res, accum)
    B, Cin, Lin = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape[0] == B and out.shape[1] == cout
    a = ConvArgs()
    a.x, a.B, a.Cin, a.Lin = x.data_ptr(), B, Cin, Lin
    a.Wp, a.bias = wp.data_ptr(), _p(bias)
    a.out, a.Cout, a.Lout = out.data_ptr(), cout, out.shape[2]
    a.ksize, a.dil, a.pad, a.up = ksize, dil, pad, up
    a.in_slope = in_slope
    a.res, a.accum, a.out_scale = _p(res), _p(accum), out_scale
    _items(a.items, items)
    _lib.check(_lib.load().cvx_hifigan_conv1d_f32(C.byref(a), _stream()), "cvx_hifigan_conv1d_f32")
    return out


def hifigan_pack_weight(w: torch.Tensor, transposed: bool) -> torch.Tensor:
    """Host-side packing of a folded conv weight into the kernel's [co_blk][chunk][k][co][16] layout."""
    w = w.detach().to("cpu", torch.float32).contiguous()
    if transposed:
        cin, cout, k = w.shape
    else:
        cout, cin, k = w.shape
    lib = _lib.load()
    n = lib.cvx_hifigan_packed_weight_floats(cout, cin, k)
    wp = torch.empty(n, dtype=torch.float32)
    _lib.check(lib.cvx_hifigan_pack_weight_f32(w.data_ptr(), cout, cin, k, int(transposed), wp.data_ptr()),
               "cvx_hifigan_pack_weight_f32")
    return wp


def hifigan_pack_conv_transpose1d(w: torch.Tensor, stride: int, padding: int) -> torch.Tensor:
    """ConvTranspose1d weight [Cin, Cout, k] -> the phase-major packed weight of cvx_hifigan_conv_transpose1d_f32 (host side)."""
    w = w.detach().to("cpu", torch.float32).contiguous()
    cin, cout, k = w.shape
    lib = _lib.load()
    wp = torch.empty(lib.cvx_hifigan_conv_transpose1d_packed_floats(cout, cin, k, stride, padding), dtype=torch.float32)
    _lib.check(lib.cvx_hifigan_pack_conv_transpose1d_f32(w.data_ptr(), cin, cout, k, stride, padding, wp.data_ptr()),
               "cvx_hifigan_pack_conv_transpose1d_f32")
    return wp


def hifigan_conv_transpose1d(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, cout: int, ksize: int,
                             stride: int, padding: int, in_slope: float = 1.0, amax_bits: Optional[torch.Tensor] = None,
                             items=None) -> torch.Tensor:
    """out = ConvTranspose1d(leaky_relu(x, in_slope)) in polyphase form; amax_bits (int32[1] on the device, zero): receives the
    bit pattern of max|out| (pow2_scale_from_amax turns it into an activation pre-scale and zeroes it)."""
    _chk_f32(x, wp, bias, out)
    B, Cin, Lin = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape[0] == B and out.shape[1] == cout
    a = ConvArgs()
    a.x, a.B, a.Cin, a.Lin = x.data_ptr(), B, Cin, Lin
    a.Wp, a.bias = wp.data_ptr(), _p(bias)
    a.out, a.Cout, a.Lout = out.data_ptr(), cout, out.shape[2]
    a.ksize, a.dil, a.pad, a.up = ksize, 1, ksize - 1 - padding, stride
    a.in_slope, a.out_scale = in_slope, 1.0
    _items(a.items, items)
    _lib.check(_lib.load().cvx_hifigan_conv_transpose1d_f32(C.byref(a), _p(amax_bits), _stream()), "cvx_hifigan_conv_transpose1d_f32")
    return out


def pow2_scale_from_amax(amax_bits: torch.Tensor, target: float, scale: torch.Tensor) -> torch.Tensor:
    """scale[0] = 2^round(log2(target / amax)) from the bit pattern a producer kernel left in amax_bits; amax_bits is zeroed."""
    assert amax_bits.is_cuda and amax_bits.dtype == torch.int32 and scale.is_cuda and scale.dtype == torch.float32
    _lib.check(_lib.load().cvx_pow2_scale_from_amax_f32(amax_bits.data_ptr(), float(target), scale.data_ptr(), _stream()),
               "cvx_pow2_scale_from_amax_f32")
    return scale


HIFI_HALO_L = 32        # zero rows in front of position 0 of every channels-last vocoder buffer (>= largest pad, 25)


def hifigan_cl_rows(L: int) -> int:
    """Rows per batch item of a channels-last vocoder buffer for L valid positions."""
    return HIFI_HALO_L + (L + 255) // 256 * 256 + 64


def hifigan_pack_weight_f16x3(w: torch.Tensor):
    """Conv1d weight [Cout, Cin, k] (fp32, on the GPU) -> (w_hi, w_lo, 1/scale, Np, Cp_in): pre-scaled split pair
    packed [Cp_in/32][k][Np][32] for cvx_hifigan_conv1d_f16x3.  Load-time plumbing."""
    cout, cin, k = w.shape
    tile = lambda c: 32 if c <= 32 else 64 if c <= 64 else 128 if c <= 128 else 256
    assert cout <= 256 and cin <= 256, "conv1d_f16x3 supports up to 256 channels"
    np_, cp = tile(cout), tile(cin)          # a stage's buffers are as wide as its output tile, for inputs too
    full = torch.zeros(np_, cp, k, dtype=torch.float32, device=w.device)
    full[:cout, :cin] = w
    packed = full.reshape(np_, cp // 32, 32, k).permute(1, 3, 0, 2).contiguous()      # [chunk][tap][co][ci]
    hi, lo, inv = split_f16(packed)
    return hi, lo, inv, np_, cp


def hifigan_pack_conv_transpose1d_f16x3(w: torch.Tensor, bias: torch.Tensor, stride: int, padding: int, cp_in: Optional[int] = None):
    """ConvTranspose1d weight [Cin, Cout, K] (fp32, on the GPU) -> the stride-1 form of cvx_hifigan_conv_transpose1d_f16x3
    (include/covomix_hip.h): dict(w_hi, w_lo, acc_scale, bias [stride * Np_out], np_out, cp_in, stride, tile_np, taps, pad,
    w_off).  Output position stride*m + r sees kernel taps c + stride*j (c = (r + padding) % stride) at inputs m + a - j
    (a = (r + padding) // stride).  Phases are grouped into column tiles (<= 256 columns, a whole number of phases) so that
    the taps a tile runs but a phase does not see - zero weights, wasted MFMAs - are fewest.  Load-time plumbing."""
    cin, cout, K = w.shape
    s = int(stride)
    tile = lambda c: 32 if c <= 32 else 64 if c <= 64 else 128 if c <= 128 else 256
    assert cout <= 256 and 1 <= s <= 8
    np_out, cp = tile(cout), cp_in or (cin + 31) // 32 * 32          # cp_in: channel count of the input buffers (>= cin, x32)
    assert cp >= cin and cp % 32 == 0
    ph = []                                               # per phase: (c, a, J) - J taps, input offsets a - J + 1 .. a
    for r in range(s):
        q = r + padding
        c, a = q % s, q // s
        ph.append((c, a, max(0, -(-(K - c) // s))))
    def plan(ppt):                                        # tiles of ppt phases: (first phase, lowest offset, taps)
        tiles = []
        for r0 in range(0, s, ppt):
            live = [(a - J + 1, a) for (c, a, J) in ph[r0:r0 + ppt] if J > 0] or [(0, 0)]
            lo, hi = min(l for l, _ in live), max(h for _, h in live)
            tiles.append((r0, lo, hi - lo + 1))
        return tiles
    cands = [ppt for ppt in range(1, s + 1) if s % ppt == 0 and ppt * np_out in (32, 64, 128, 256) and s // ppt <= 8]
    # matrix work of a plan (taps x columns, summed over its tiles) plus a per-tile term for what every extra tile costs
    # whatever its width (its own activation tiles, blocks and epilogues): the narrow layers prefer ONE tile with a zero tap
    ppt = min(cands, key=lambda n: (sum(t[2] * n * np_out + 256 for t in plan(n)), n))
    tiles, tile_np = plan(ppt), ppt * np_out
    blocks, w_off, taps, pads, off = [], [], [], [], 0
    for r0, lo, nt in tiles:
        full = torch.zeros(tile_np, cp, nt, dtype=torch.float32, device=w.device)      # conv1d weight [co, ci, tap]
        for rl in range(ppt):
            c, a, J = ph[r0 + rl]
            for t in range(nt):
                j = a - (lo + t)
                if 0 <= j < J:
                    full[rl * np_out: rl * np_out + cout, :cin, t] = w[:, :, s * j + c].t()
        blocks.append(full.reshape(tile_np, cp // 32, 32, nt).permute(1, 3, 0, 2).contiguous().reshape(-1))   # [chunk][tap][co][ci]
        w_off.append(off); taps.append(nt); pads.append(-lo)
        off += blocks[-1].numel()
    hi, lo_, inv = split_f16(torch.cat(blocks))
    b = torch.zeros(s, np_out, dtype=torch.float32, device=w.device)
    b[:, :cout] = bias.to(w.device).float()
    return dict(w_hi=hi, w_lo=lo_, acc_scale=inv, bias=b.reshape(-1).contiguous(), np_out=np_out, cp_in=cp, stride=s, tile_np=tile_np,
                taps=taps, pad=pads, w_off=w_off, ksize=K, padding=padding)


def hifigan_conv_transpose1d_f16x3(z, pk: dict, B: int, L_in: int, out: torch.Tensor, L_out: int, *, z_scale=None, amax_bits=None,
                                   items=None) -> torch.Tensor:
    """out (fp32 channels-last [B, Lp_out, Np_out]) = ConvTranspose1d of the activation whose split(leaky_relu(.) * z_scale) pair is
    z = (hi, lo) [B, Lp_in, Cp_in]; pk from hifigan_pack_conv_transpose1d_f16x3.  items: valid OUTPUT positions per item."""
    ensure_saturation_bound()
    zh, zl = z
    assert zh.dtype == torch.float16 and zh.is_contiguous() and zl.is_contiguous() and zh.shape == zl.shape
    assert zh.shape[0] == B and zh.shape[2] == pk["cp_in"], (tuple(zh.shape), pk["cp_in"])
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == B and out.shape[2] == pk["np_out"]
    a = _lib.ConvT16Args()
    a.z_hi, a.z_lo = zh.data_ptr(), zl.data_ptr()
    a.B, a.L_in, a.Lp_in, a.Cp_in, a.halo_in = B, L_in, zh.shape[1], pk["cp_in"], HIFI_HALO_L
    a.w_hi, a.w_lo, a.acc_scale, a.bias = pk["w_hi"].data_ptr(), pk["w_lo"].data_ptr(), pk["acc_scale"], pk["bias"].data_ptr()
    a.Np_out, a.stride, a.n_tiles, a.tile_np = pk["np_out"], pk["stride"], len(pk["taps"]), pk["tile_np"]
    for t in range(len(pk["taps"])):
        a.tile_taps[t], a.tile_pad[t], a.tile_w_off[t] = pk["taps"][t], pk["pad"][t], pk["w_off"][t]
    a.out, a.L_out, a.Lp_out, a.halo_out = out.data_ptr(), L_out, out.shape[1], HIFI_HALO_L
    a.z_scale_dev, a.amax_bits_dev = _sp(z_scale), _p(amax_bits)
    _items(a.items, items)
    _lib.check(_lib.load().cvx_hifigan_conv_transpose1d_f16x3(C.byref(a), _stream()), "cvx_hifigan_conv_transpose1d_f16x3")
    return out


def hifigan_split_channels_last(x_cl: torch.Tensor, z, slope: float, z_scale=None) -> None:
    """z = split(leaky_relu(x_cl, slope) * z_scale) over a whole fp32 channels-last buffer."""
    ensure_saturation_bound()
    zh, zl = z
    assert x_cl.dtype == torch.float32 and x_cl.is_contiguous() and zh.dtype == torch.float16 and zh.is_contiguous() and zl.is_contiguous()
    assert zh.shape == x_cl.shape and zl.shape == x_cl.shape and x_cl.numel() % 4 == 0
    _lib.check(_lib.load().cvx_hifigan_split_channels_last(x_cl.data_ptr(), zh.data_ptr(), zl.data_ptr(), x_cl.numel(), slope,
                                                           _sp(z_scale), _stream()), "cvx_hifigan_split_channels_last")


def hifigan_post_channels_last(x_cl: torch.Tensor, C_: int, L: int, w: torch.Tensor, bias: float, out: torch.Tensor, slope: float = 0.01) -> torch.Tensor:
    """hifigan_post reading the channels-last stage output [B, Lp, Np] (C_ valid channels, L valid positions); out [B, 1, L]."""
    _chk_f32(x_cl, w, out)
    B, Lp, np_ = x_cl.shape
    assert x_cl.is_contiguous() and out.is_contiguous() and out.numel() == B * L and w.numel() == C_ * 7
    _lib.check(_lib.load().cvx_hifigan_post_channels_last_f32(x_cl.data_ptr(), w.data_ptr(), float(bias), out.data_ptr(), B, C_, np_, L, Lp,
                                                              HIFI_HALO_L, slope, _stream()), "cvx_hifigan_post_channels_last_f32")
    return out


def amax_pow2_scale(x: torch.Tensor, target: float, scale: torch.Tensor, scratch: torch.Tensor) -> torch.Tensor:
    """scale[0] = 2^round(log2(target / max|x|)) computed on the device (scratch: one int32 element)."""
    _chk_f32(x, scale)
    assert x.is_contiguous() and scale.numel() == 1 and scratch.is_cuda and scratch.dtype == torch.int32 and scratch.numel() >= 1
    _lib.check(_lib.load().cvx_amax_pow2_scale_f32(x.data_ptr(), x.numel(), float(target), scale.data_ptr(), scratch.data_ptr(), _stream()),
               "cvx_amax_pow2_scale_f32")
    return scale


def hifigan_conv1d_f16x3(z, wpk, bias, B: int, L: int, *, ksize: int, dil: int, res=None, accum=None, out_x=None,
                         out_scale: float = 1.0, out_z=None, z_slope: float = 0.1, z_scale=None, items=None) -> None:
    """z = (hi, lo) channels-last [B, Lp, Cp_in]; wpk from hifigan_pack_weight_f16x3; bias [Np] (zero padded)."""
    ensure_saturation_bound()
    zh, zl = z
    w_hi, w_lo, inv, np_, cp = wpk
    assert zh.dtype == torch.float16 and zh.is_contiguous() and zl.is_contiguous() and zh.shape == zl.shape
    assert zh.shape[0] == B and zh.shape[2] == cp and bias.numel() == np_
    a = _lib.Conv16Args()
    a.z_hi, a.z_lo = zh.data_ptr(), zl.data_ptr()
    a.B, a.L, a.Lp, a.Cp_in, a.halo_l = B, L, zh.shape[1], cp, HIFI_HALO_L
    a.w_hi, a.w_lo, a.acc_scale, a.bias = w_hi.data_ptr(), w_lo.data_ptr(), inv, bias.data_ptr()
    a.Np, a.ksize, a.dil = np_, ksize, dil
    for t in (res, accum, out_x):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (B, zh.shape[1], np_))
    a.res, a.accum, a.out_x, a.out_scale = _p(res), _p(accum), _p(out_x), out_scale
    if out_z is not None:
        oh, ol = out_z
        assert oh.dtype == torch.float16 and oh.is_contiguous() and ol.is_contiguous() and tuple(oh.shape) == (B, zh.shape[1], np_)
        a.out_zhi, a.out_zlo = oh.data_ptr(), ol.data_ptr()
    else:
        a.out_zhi, a.out_zlo = None, None
    a.z_slope = z_slope
    a.z_scale_dev = _sp(z_scale)
    _items(a.items, items)
    _lib.check(_lib.load().cvx_hifigan_conv1d_f16x3(C.byref(a), _stream()), "cvx_hifigan_conv1d_f16x3")


def _fill_resblock16(a, x_cl, z, block, B: int, L: int, scratch: dict, accum, out, out_scale: float, z_scale, items) -> None:
    narrow = x_cl.shape[2] <= 64              # fused pair kernel: no split pairs in HBM (z / t / rz0 / rz1 unused)
    zh, zl = z if z is not None else (None, None)
    assert narrow or z is not None
    a.x, a.z_hi, a.z_lo = x_cl.data_ptr(), _p(zh), _p(zl)
    a.B, a.L, a.Lp, a.Np, a.halo_l = B, L, x_cl.shape[1], x_cl.shape[2], HIFI_HALO_L
    for m, (c1, c2) in enumerate(block):
        for dst, c in ((a.c1[m], c1), (a.c2[m], c2)):
            w_hi, w_lo, inv, np_, cp = c.w16
            assert np_ == cp == x_cl.shape[2] and c.bias16.numel() == np_
            dst.w_hi, dst.w_lo, dst.acc_scale, dst.bias = w_hi.data_ptr(), w_lo.data_ptr(), inv, c.bias16.data_ptr()
        a.dil[m] = c1.dil
        assert c2.dil == 1 and c1.k == c2.k
    a.ksize = block[0][0].k
    a.xa, a.xb = scratch["r0"].data_ptr(), scratch["r1"].data_ptr()
    if not narrow:
        a.t_hi, a.t_lo = scratch["t"][0].data_ptr(), scratch["t"][1].data_ptr()
        a.za_hi, a.za_lo = scratch["rz0"][0].data_ptr(), scratch["rz0"][1].data_ptr()
        a.zb_hi, a.zb_lo = scratch["rz1"][0].data_ptr(), scratch["rz1"][1].data_ptr()
    a.accum, a.out, a.out_scale = _p(accum), out.data_ptr(), out_scale
    a.z_scale_dev = _sp(z_scale)
    _items(a.items, items)


def hifigan_resblock_f16x3(x_cl, z, block, B: int, L: int, scratch: dict, *, accum=None, out=None, out_scale: float = 1.0, z_scale=None,
                           items=None) -> None:
    """One ResBlock1 (three conv pairs) through the operator-level C entry point cvx_hifigan_resblock_f16x3.
    block: list of 3 (c1, c2) pairs of objects with .w16 = hifigan_pack_weight_f16x3(...), .bias16, .k, .dil;
    scratch: dict with t, rz0, rz1 (split pairs) and r0, r1 (fp32), all [B, Lp, Np] like x_cl / z."""
    ensure_saturation_bound()
    a = _lib.Resblock16Args()
    _fill_resblock16(a, x_cl, z, block, B, L, scratch, accum, out, out_scale, z_scale, items)
    _lib.check(_lib.load().cvx_hifigan_resblock_f16x3(C.byref(a), _stream()), "cvx_hifigan_resblock_f16x3")


def hifigan_resblock_stage_f16x3(x_cl, z, blocks, B: int, L: int, scratches, out, *, out_scale: float = 1.0, z_scale=None, items=None) -> None:
    """out = out_scale * sum_j ResBlock1_j(x) for the (up to 3) ResBlocks of one generator stage (models.py:104-110) through
    cvx_hifigan_resblock_stage_f16x3: the convolutions of different blocks that do not depend on each other share launches; bit-identical
    to calling hifigan_resblock_f16x3 block after block (accum = out from the second block on, out_scale on the last).
    scratches: one scratch dict per block (t, rz0, rz1, r0, r1 - their own buffers)."""
    ensure_saturation_bound()
    n = len(blocks)
    arr = (_lib.Resblock16Args * n)()
    for j, block in enumerate(blocks):
        _fill_resblock16(arr[j], x_cl, z, block, B, L, scratches[j], out if j > 0 else None, out, out_scale if j == n - 1 else 1.0, z_scale, items)
    _lib.check(_lib.load().cvx_hifigan_resblock_stage_f16x3(arr, n, _stream()), "cvx_hifigan_resblock_stage_f16x3")


def hifigan_resblock_pair_f16x3(x_cl, c1, c2, B: int, L: int, out, *, accum=None, out_scale: float = 1.0, z_scale=None, flags: int = 0,
                                items=None) -> None:
    """out = (c2(lrelu(c1(lrelu(x)))) + x (+ accum)) * out_scale as one kernel (cvx_hifigan_resblock_pair_f16x3; Np = 32 / 64).
    x_cl / out / accum: fp32 channels-last [B, Lp, Np]; c1, c2: objects with .w16, .bias16, .k, .dil (c2.dil == 1)."""
    ensure_saturation_bound()
    a = _lib.Respair16Args()
    a.x = x_cl.data_ptr()
    a.B, a.L, a.Lp, a.Np, a.halo_l = B, L, x_cl.shape[1], x_cl.shape[2], HIFI_HALO_L
    for dst, c in ((a.c1, c1), (a.c2, c2)):
        w_hi, w_lo, inv, np_, cp = c.w16
        assert np_ == cp == x_cl.shape[2] and c.bias16.numel() == np_
        dst.w_hi, dst.w_lo, dst.acc_scale, dst.bias = w_hi.data_ptr(), w_lo.data_ptr(), inv, c.bias16.data_ptr()
    assert c2.dil == 1 and c1.k == c2.k
    a.ksize, a.dil = c1.k, c1.dil
    a.accum, a.out, a.out_scale = _p(accum), out.data_ptr(), out_scale
    a.z_scale_dev = _sp(z_scale)
    a.flags = flags
    _items(a.items, items)
    _lib.check(_lib.load().cvx_hifigan_resblock_pair_f16x3(C.byref(a), _stream()), "cvx_hifigan_resblock_pair_f16x3")


def hifigan_to_channels_last(x: torch.Tensor, x_cl: Optional[torch.Tensor], z, slope: float, z_scale=None) -> None:
    """x [B, C, L] fp32 channel-major -> x_cl [B, Lp, Cp] fp32 and / or z = split(leaky_relu(x)) [B, Lp, Cp]."""
    ensure_saturation_bound()
    _chk_f32(x, x_cl)
    B, Cc, L = x.shape
    ref = x_cl if x_cl is not None else z[0]
    Lp, Cp = ref.shape[1], ref.shape[2]
    assert x.is_contiguous() and ref.is_contiguous() and ref.shape[0] == B
    zh, zl = z if z is not None else (None, None)
    _lib.check(_lib.load().cvx_hifigan_to_channels_last_scaled(x.data_ptr(), _p(x_cl), _p(zh), _p(zl), B, Cc, L, Lp, Cp, HIFI_HALO_L,
                                                               slope, _sp(z_scale), _stream()), "cvx_hifigan_to_channels_last")


def hifigan_from_channels_last(x_cl: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _chk_f32(x_cl, x)
    B, Cc, L = x.shape
    assert x.is_contiguous() and x_cl.is_contiguous() and x_cl.shape[0] == B
    _lib.check(_lib.load().cvx_hifigan_from_channels_last(x_cl.data_ptr(), x.data_ptr(), B, Cc, L, x_cl.shape[1], x_cl.shape[2],
                                                          HIFI_HALO_L, _stream()), "cvx_hifigan_from_channels_last")
    return x


def hifigan_post(x: torch.Tensor, w: torch.Tensor, bias: float, out: torch.Tensor, slope: float = 0.01) -> torch.Tensor:
    _chk_f32(x, w, out)
    B, Cin, L = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.numel() == B * L and w.numel() == Cin * 7
    _lib.check(_lib.load().cvx_hifigan_post_f32(x.data_ptr(), w.data_ptr(), bias, out.data_ptr(), B, Cin, L, slope,
                                                _stream()), "cvx_hifigan_post_f32")
    return out


def wav_to_int16(wav: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk_f32(wav)
    assert wav.is_contiguous()
    if out is None:
        out = torch.empty(wav.shape, dtype=torch.int16, device=wav.device)
    _lib.check(_lib.load().cvx_wav_to_int16(wav.data_ptr(), out.data_ptr(), wav.numel(), _stream()), "cvx_wav_to_int16")
    return out


# ---------------------------------------------------------------- HuBERT tokeniser (N4)
def hubert_conv0_gn_gelu(wav: torch.Tensor, w: torch.Tensor, gn_gamma: torch.Tensor, gn_beta: torch.Tensor,
                         stride: int, eps: float = 1e-5) -> torch.Tensor:
    """First conv layer + GroupNorm(C, C) + GELU of the HuBERT feature extractor: wav [n] -> [L, C] channels-last."""
    _chk_f32(wav, w, gn_gamma, gn_beta)
    assert wav.dim() == 1 and wav.is_contiguous() and w.is_contiguous() and w.dim() == 2
    C, k = w.shape
    n = wav.numel()
    L = (n - k) // stride + 1
    lib = _lib.load()
    out = torch.empty(max(L, 0), C, dtype=torch.float32, device=wav.device)
    ws = torch.empty(int(lib.cvx_hubert_conv0_workspace_floats(max(L, 1), C)), dtype=torch.float32, device=wav.device)
    _lib.check(lib.cvx_hubert_conv0_gn_gelu_f32(wav.data_ptr(), n, w.data_ptr(), C, k, stride, gn_gamma.data_ptr(),
                                                gn_beta.data_ptr(), eps, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "cvx_hubert_conv0_gn_gelu_f32")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: Optional[torch.Tensor] = None,
              eps: float = 1e-5) -> torch.Tensor:
    _chk_f32(x, gamma, beta, out)
    assert x.is_contiguous() and gamma.is_contiguous() and beta.is_contiguous()
    D = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    assert out.is_contiguous() and out.shape == x.shape
    _lib.check(_lib.load().cvx_layernorm_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                             x.numel() // D, D, eps, _stream()), "cvx_layernorm_f32")
    return out


def hubert_group_pack(x: torch.Tensor, groups: int, halo: int) -> torch.Tensor:
    """x [T, D] -> [groups, T + 2*halo, D/groups] with zero halos (operand of the grouped positional convolution)."""
    _chk_f32(x)
    assert x.dim() == 2 and x.is_contiguous()
    T, D = x.shape
    out = torch.empty(groups, T + 2 * halo, D // groups, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().cvx_hubert_group_pack_f32(x.data_ptr(), out.data_ptr(), T, D, groups, halo, _stream()),
               "cvx_hubert_group_pack_f32")
    return out


def kmeans_argmin(x: torch.Tensor, dots: torch.Tensor, cnorm: torch.Tensor, with_margin: bool = False):
    """labels[t] = argmin_j (|x_t|^2 - 2 dots[t, j]) + cnorm[j]  (int64); optionally also the runner-up margin."""
    _chk_f32(x, dots, cnorm)
    assert x.dim() == 2 and x.is_contiguous() and dots.is_contiguous() and cnorm.is_contiguous()
    T, D = x.shape
    K = cnorm.numel()
    assert dots.shape == (T, K)
    labels = torch.empty(T, dtype=torch.int64, device=x.device)
    margin = torch.empty(T, dtype=torch.float32, device=x.device) if with_margin else None
    if T == 0:
        return (labels, margin) if with_margin else labels
    _lib.check(_lib.load().cvx_kmeans_argmin_f32(x.data_ptr(), dots.data_ptr(), cnorm.data_ptr(), labels.data_ptr(),
                                                 _p(margin), T, D, K, _stream()), "cvx_kmeans_argmin_f32")
    return (labels, margin) if with_margin else labels
