"""CPU oracle for the CoVoMix mel-generation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional, plain-PyTorch fp32 restatement of the
reference algorithm for the path named in BASELINE.json (VoMix/VoSingle vector
field -> CFG -> fixed-grid midpoint ODE -> HiFi-GAN generator -> int16).  It is
the *checker*: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it.  The product path (neurips2024-covomix_amd/) never does, and
fails loudly when its HIP library is missing.

Pinning status
  * vector field, CFG, HiFi-GAN, weight-norm fold, int16 cast, token assembly:
    PINNED - checked <=1e-5 rel-L2 (bit-exact for integer paths) against the
    imported reference modules in the build container by
    tests/golden/make_golden.py, which also wrote tests/golden/*.npz.
  * ODE integrator: the arithmetic lives in third-party `torchdiffeq`
    (un-pinned version, absent from /root/reference and from this image;
    call site acoustic.py:656, kwargs :586-591).  Restated here from its
    published fixed-grid algorithm; checked against an analytic ODE and by
    driving the *imported* reference vector field.  Parity at that boundary is
    "unpinned" by any reference test (the reference has none).

All `file:line` citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# shapes / hyper-parameters recovered from a state dict
# --------------------------------------------------------------------------
def acoustic_dims(sd: SD) -> dict:
    """Recover the CoVoMix hyper-parameters from parameter shapes
    (names per covomix/covomix_model/acoustic.py:326-406)."""
    dim = sd["to_embed.weight"].shape[0]
    e_in = sd["to_embed.weight"].shape[1]
    dim_cond = sd["null_cond"].shape[0]
    dim_emb = sd["to_phoneme_emb.weight"].shape[1]
    depth = 0
    while f"transformer.layers.{depth}.2.to_qkv.weight" in sd:
        depth += 1
    dim_head = 2 * sd["transformer.rotary_emb.inv_freq"].shape[0]
    heads = sd["transformer.layers.0.2.to_qkv.weight"].shape[0] // (3 * dim_head)
    dim_out = sd["to_pred.weight"].shape[0]
    streams = (e_in - dim_out - dim_cond) // dim_emb
    assert dim_out + streams * dim_emb + dim_cond == e_in, "unsupported to_embed layout"
    return dict(dim=dim, e_in=e_in, dim_cond=dim_cond, dim_emb=dim_emb, depth=depth,
                dim_head=dim_head, heads=heads, dim_out=dim_out, streams=streams,
                null_id=sd["to_phoneme_emb.weight"].shape[0] - 1,
                conv_k=sd["conv_embed.dw_conv1d.0.weight"].shape[-1])


# --------------------------------------------------------------------------
# vector field  (acoustic.py:430-521, inference subset)
# --------------------------------------------------------------------------
def _l2_unit(x: Tensor) -> Tensor:
    # F.normalize(x, dim=-1): x / max(||x||_2, 1e-12)   (acoustic.py:175,199)
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def time_embedding(sd: SD, times: Tensor) -> Tensor:
    """LearnedSinusoidalPosEmb -> Linear -> SiLU  (acoustic.py:98-111, :361-365)."""
    w = sd["sinu_pos_emb.0.weights"]
    ang = times[:, None] * w[None, :] * 2 * math.pi
    four = torch.cat((ang.sin(), ang.cos()), dim=-1)
    return F.silu(F.linear(four, sd["sinu_pos_emb.1.weight"], sd["sinu_pos_emb.1.bias"]))


def _ada_norm(sd: SD, prefix: str, x: Tensor, temb: Tensor) -> Tensor:
    # AdaptiveRMSNorm.forward (acoustic.py:198-204)
    scale = x.shape[-1] ** 0.5
    g = F.linear(temb, sd[prefix + ".to_gamma.weight"], sd[prefix + ".to_gamma.bias"])
    b = F.linear(temb, sd[prefix + ".to_beta.weight"], sd[prefix + ".to_beta.bias"])
    return _l2_unit(x) * scale * g[:, None, :] + b[:, None, :]


def rope_angles(sd: SD, n: int) -> Tensor:
    # RotaryEmbedding.forward (acoustic.py:126-130): freqs = cat(pos x inv_freq, same)
    pos = torch.arange(n, dtype=sd["transformer.rotary_emb.inv_freq"].dtype)
    f = pos[:, None] * sd["transformer.rotary_emb.inv_freq"][None, :]
    return torch.cat((f, f), dim=-1)


def _rope(ang: Tensor, t: Tensor) -> Tensor:
    # half-split rotation (acoustic.py:132-137)
    h = t.shape[-1] // 2
    rot = torch.cat((-t[..., h:], t[..., :h]), dim=-1)
    return t * ang.cos() + rot * ang.sin()


def _attention(sd: SD, prefix: str, x: Tensor, ang: Tensor, heads: int) -> Tensor:
    # Attention.forward (acoustic.py:225-237) + Attend.forward non-flash (attend.py:108-126)
    b, n, _ = x.shape
    qkv = F.linear(x, sd[prefix + ".to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    q, k, v = (t.reshape(b, n, heads, -1).transpose(1, 2) for t in (q, k, v))
    q, k = _rope(ang, q), _rope(ang, k)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    out = torch.matmul(sim.softmax(dim=-1), v)
    out = out.transpose(1, 2).reshape(b, n, -1)
    return F.linear(out, sd[prefix + ".to_out.weight"])


def acoustic_forward(sd: SD, x: Tensor, times: Tensor, phoneme_ids: Tensor, cond: Tensor,
                     drop_cond: bool) -> Tensor:
    """One evaluation of CoVoMix.forward with target=None (acoustic.py:430-521).

    drop_cond=False <=> cond_drop_prob=0., True <=> cond_drop_prob=1. (the only two
    values forward_with_cond_scale ever passes, :421,:426).  The RNG draws the
    reference burns for its unused training mask (:460-466) are not reproduced.
    """
    d = acoustic_dims(sd)
    bsz, n, _ = cond.shape
    if times.ndim == 0 or times.numel() == 1:
        times = times.reshape(1).expand(bsz)                       # :452-456
    if drop_cond:                                                   # :473-494
        cond = sd["null_cond"].expand_as(cond)
        phoneme_ids = torch.full_like(phoneme_ids, d["null_id"])
    emb = F.embedding(phoneme_ids, sd["to_phoneme_emb.weight"])     # :496
    if emb.ndim == 4:                                               # :499-500
        emb = emb.reshape(bsz, n, -1)
    h = F.linear(torch.cat((x, emb, cond), dim=-1), sd["to_embed.weight"], sd["to_embed.bias"])
    # ConvPositionEmbed + residual (:141-161, :508)
    k = d["conv_k"]
    c = F.conv1d(h.transpose(1, 2), sd["conv_embed.dw_conv1d.0.weight"],
                 sd["conv_embed.dw_conv1d.0.bias"], padding=k // 2, groups=h.shape[-1])
    h = F.gelu(c).transpose(1, 2) + h
    temb = time_embedding(sd, times.to(h.dtype))                    # :510
    # Transformer.forward (:288-318)
    ang = rope_angles(sd, n)
    skips = []
    for i in range(d["depth"]):
        p = f"transformer.layers.{i}"
        if (p + ".0.weight") in sd:                                 # has_skip, :306-310
            h = F.linear(torch.cat((h, skips.pop()), dim=-1), sd[p + ".0.weight"], sd[p + ".0.bias"])
        else:
            skips.append(h)
        h = _attention(sd, p + ".2", _ada_norm(sd, p + ".1", h, temb), ang, d["heads"]) + h
        f = _ada_norm(sd, p + ".3", h, temb)
        f = F.linear(F.gelu(F.linear(f, sd[p + ".4.0.weight"], sd[p + ".4.0.bias"])),
                     sd[p + ".4.2.weight"], sd[p + ".4.2.bias"])
        h = f + h
    h = _l2_unit(h) * (h.shape[-1] ** 0.5) * sd["transformer.final_norm.gamma"]   # :175,:318
    return F.linear(h, sd["to_pred.weight"])                        # :516


def forward_with_cond_scale(sd: SD, x: Tensor, times: Tensor, phoneme_ids: Tensor, cond: Tensor,
                            cond_scale: float) -> Tensor:
    """CFG combine exactly as acoustic.py:414-428: f_c*(1+s) - s*f_null, and the
    null branch is skipped only when s == 1.0."""
    f_c = acoustic_forward(sd, x, times, phoneme_ids, cond, drop_cond=False)
    if cond_scale == 1.0:
        return f_c
    f_n = acoustic_forward(sd, x, times, phoneme_ids, cond, drop_cond=True)
    return f_c * (1 + cond_scale) - cond_scale * f_n


# --------------------------------------------------------------------------
# fixed-grid ODE integration (third-party torchdiffeq semantics; see header)
# --------------------------------------------------------------------------
def fixed_grid(step_size: float = 0.0625, t0: float = 0.0, t1: float = 1.0) -> Tensor:
    """torchdiffeq fixed-grid constructor for options={'step_size': h}:
    niters = ceil((t1-t0)/h + 1); grid = arange(niters)*h + t0; grid[-1] = t1."""
    n = int(math.ceil((t1 - t0) / step_size + 1))
    g = torch.arange(0, n, dtype=torch.float32) * step_size + t0
    g[-1] = t1
    return g


def odeint_fixed(fn: Callable[[Tensor, Tensor], Tensor], y0: Tensor, grid: Tensor,
                 method: str = "midpoint") -> Tensor:
    """Integrate dy/dt = fn(t, y) over `grid`, return y(grid[-1]).

    midpoint (what acoustic.py:572,656 selects): per step
        f0 = fn(t0, y); y_mid = y + f0*(dt/2); y <- y + dt*fn(t0+dt/2, y_mid)
    euler (offered by the build, not present in the reference): y <- y + dt*fn(t0, y)
    The reference asks for outputs at t in {0,.5,1} (acoustic.py:651); with
    step 1/16 they fall on grid points so no interpolation occurs and the last
    state is returned (:657).
    """
    y = y0
    for a, b in zip(grid[:-1], grid[1:]):
        dt = b - a
        if method == "midpoint":
            half = 0.5 * dt
            f0 = fn(a, y)
            y_mid = y + f0 * half
            y = y + dt * fn(a + half, y_mid)
        elif method == "euler":
            y = y + dt * fn(a, y)
        else:
            raise ValueError(method)
    return y


def sample(sd: SD, phoneme_ids: Tensor, cond: Tensor, y0: Tensor, cond_scale: float,
           nfe: int = 32, method: str = "midpoint") -> Tensor:
    """ConditionalFlowMatcherWrapper.sample (acoustic.py:597-688) given the noise y0.
    `nfe` = number of CFG-combined vector-field evaluations ("32-step" = 16 midpoint steps)."""
    steps = nfe // 2 if method == "midpoint" else nfe
    grid = fixed_grid(1.0 / steps)
    fn = lambda t, x: forward_with_cond_scale(sd, x, t, phoneme_ids, cond, cond_scale)
    return odeint_fixed(fn, y0, grid, method)


# --------------------------------------------------------------------------
# HiFi-GAN generator  (covomix/vocoder/models.py:75-125, config hifi-gan/config_covomix.json)
# --------------------------------------------------------------------------
def fold_weight_norm(sd: SD) -> SD:
    """remove_weight_norm: w = v * g / ||v||, norm over every dim but 0
    (for ConvTranspose1d that is per INPUT channel).  models.py:118-125."""
    out: SD = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            wv = sd[base + ".weight_v"]
            nrm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (wv.ndim - 1)))
            out[base + ".weight"] = wv * (v / nrm)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def hifigan_forward(sd: SD, h: dict, mel: Tensor) -> Tensor:
    """Generator.forward on folded weights.  mel [B,80,T] or [80,T] -> [B,1,L] or [1,L]."""
    lrelu = 0.1                                                    # models.py:8
    nk = len(h["resblock_kernel_sizes"])
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, lrelu)
        # PyTorch 2.10 CPU / oneDNN computes THIS op wrong for some shapes when more than one thread runs it (ups.0 of
        # config_covomix, Cin 500 -> Cout 250, k 8, stride 5, 88 or 120 input frames: max error 0.9 of max |y| 3.9 against the
        # fp64 evaluation; 1 thread or the native kernel: 1e-6).  The oracle is the yardstick, so the transposed convolutions run
        # on the native kernel (tools/voc_flaky_probe.py found it: the HIP path agreed with fp64, the fp32 oracle did not).
        prev = torch._C._get_mkldnn_enabled()                          # (torch.backends.mkldnn.flags warns about Intel GPUs on every entry)
        torch._C._set_mkldnn_enabled(False)
        try:
            x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u,
                                   padding=(k - u) // 2)
        finally:
            torch._C._set_mkldnn_enabled(prev)
        xs = None
        for j, (rk, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            r = x
            p = f"resblocks.{i * nk + j}"
            for m, dd in enumerate(dil):                           # ResBlock1.forward :35-42
                t = F.leaky_relu(r, lrelu)
                t = F.conv1d(t, sd[f"{p}.convs1.{m}.weight"], sd[f"{p}.convs1.{m}.bias"],
                             dilation=dd, padding=(rk * dd - dd) // 2)
                t = F.leaky_relu(t, lrelu)
                t = F.conv1d(t, sd[f"{p}.convs2.{m}.weight"], sd[f"{p}.convs2.{m}.bias"],
                             padding=(rk - 1) // 2)
                r = t + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)                                            # default slope 0.01 (:112)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def wav_to_int16(y: Tensor):
    """mel_decode_to_wav tail (monologue_generation.py:52-59): squeeze, *32768, astype(int16)."""
    return (y.squeeze() * 32768.0).cpu().numpy().astype("int16")


# --------------------------------------------------------------------------
# integer token / prompt assembly (bit-exact path)
# --------------------------------------------------------------------------
def assemble_dialogue(sem_a: Tensor, sem_b: Tensor, pred_a: Tensor, pred_b: Tensor,
                      mel_a: Tensor, mel_b: Tensor):
    """covomix() input assembly, monologue_generation.py:263-295 / dialogue_generation.py:287-320.
    sem_*: prompt tokens i64[Tp*]; pred_*: predicted tokens; mel_*: prompt mel f32[Tp*,80]."""
    tp = min(mel_a.shape[0], mel_b.shape[0])
    a = torch.cat((sem_a[:tp], pred_a))
    b = torch.cat((sem_b[:tp], pred_b))
    n = max(a.shape[0], b.shape[0])
    a = F.pad(a, (0, n - a.shape[0]), value=157)
    b = F.pad(b, (0, n - b.shape[0]), value=157)
    ids = torch.stack((a, b), dim=-1).clamp(max=501)
    mask = torch.zeros(n, dtype=torch.bool)
    mask[tp:] = True
    mel = torch.zeros(n, 160)
    mel[:tp] = torch.cat((mel_a[:tp], mel_b[:tp]), dim=-1)
    return ids, mel, mask


def assemble_monologue(sem: Tensor, pred: Tensor, mel_prompt: Tensor):
    """covosingle() input assembly, monologue_generation.py:161-166."""
    ids = torch.cat((sem, pred)).clamp(max=501)
    mel = torch.zeros(ids.shape[0], 80)
    mel[: mel_prompt.shape[0]] = mel_prompt
    mask = torch.zeros(ids.shape[0], dtype=torch.bool)
    mask[mel_prompt.shape[0]:] = True
    return ids, mel, mask


def select_generated(sampled: Tensor, mask: Tensor) -> Tensor:
    """monologue_generation.py:299-300: sampled[:, mask, :] -> [80, Tgen] for the vocoder."""
    return sampled[:, mask, :].permute(0, 2, 1).squeeze(0)
