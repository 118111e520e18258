"""Platform-stable synthetic checkpoints for parity tests and the benchmark.

No pretrained CoVoMix weights exist (reference README.md:30), and full-width
weights (VoMix 1.0 GB, HiFi-GAN 48 MB) cannot be committed, so every test and
bench run regenerates *identical* weights from this recipe:

    value(name) = RandomState(crc32(name) ^ seed).standard_normal(shape) * scale(name) + shift(name)

`numpy.random.RandomState` streams are frozen by NumPy, so the container that
produced tests/golden/*.npz (by loading these tensors into the *reference*
modules) and the GPU box see bit-identical parameters.

The recipe de-trivialises the identity-at-init parameters of the reference
(AdaptiveRMSNorm to_gamma/to_beta acoustic.py:192-196, null_cond :382, HiFi-GAN
N(0,0.01) conv init vocoder/utils.py:22-25) so that time conditioning, the CFG
null branch and the vocoder's dynamic range are all exercised.

Parameter names/shapes/order follow the reference modules:
  acoustic: covomix/covomix_model/acoustic.py:326-406 (CoVoMix.__init__), :250-286 (Transformer)
  vocoder : covomix/vocoder/models.py:75-98 (Generator.__init__), :11-33 (ResBlock1)
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Optional, Dict, Tuple

import numpy as np

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def acoustic_param_shapes(dim=1024, dim_cond=160, dim_emb=1024, depth=8, heads=16, dim_head=64,
                          streams=2, dim_out=80, num_phoneme_tokens=502, conv_k=31,
                          time_hidden=None) -> Shapes:
    """Ordered name->shape map in nn.Module.parameters() order of the reference CoVoMix.

    VoMix  (twocondition_oneoutput): dim_cond=160, streams=2, dim_out=80 -> E_in = 2288
    VoSingle                        : dim_cond=80,  streams=1, dim_out=80 -> E_in = 1184
    """
    th = time_hidden or dim * 4
    e_in = dim_out + streams * dim_emb + dim_cond
    s = OrderedDict()
    s["null_cond"] = (dim_cond,)
    s["sinu_pos_emb.0.weights"] = (dim // 2,)
    s["sinu_pos_emb.1.weight"] = (th, dim)
    s["sinu_pos_emb.1.bias"] = (th,)
    s["to_phoneme_emb.weight"] = (num_phoneme_tokens + 1, dim_emb)
    s["to_embed.weight"] = (dim, e_in)
    s["to_embed.bias"] = (dim,)
    s["conv_embed.dw_conv1d.0.weight"] = (dim, 1, conv_k)
    s["conv_embed.dw_conv1d.0.bias"] = (dim,)
    inner = heads * dim_head
    for i in range(depth):
        p = f"transformer.layers.{i}"
        if i + 1 > depth // 2:
            s[p + ".0.weight"] = (dim, 2 * dim)
            s[p + ".0.bias"] = (dim,)
        for n in (1, 3):
            s[f"{p}.{n}.to_gamma.weight"] = (dim, th)
            s[f"{p}.{n}.to_gamma.bias"] = (dim,)
            s[f"{p}.{n}.to_beta.weight"] = (dim, th)
            s[f"{p}.{n}.to_beta.bias"] = (dim,)
            if n == 1:
                s[p + ".2.to_qkv.weight"] = (3 * inner, dim)
                s[p + ".2.to_out.weight"] = (dim, inner)
        s[p + ".4.0.weight"] = (4 * dim, dim)
        s[p + ".4.0.bias"] = (4 * dim,)
        s[p + ".4.2.weight"] = (dim, 4 * dim)
        s[p + ".4.2.bias"] = (dim,)
    s["transformer.final_norm.gamma"] = (dim,)
    s["to_pred.weight"] = (dim_out, dim)
    return s


def hifigan_param_shapes(h: dict) -> Shapes:
    """Ordered name->shape map of Generator.state_dict() *before* remove_weight_norm
    (bias, weight_g, weight_v per conv - the layout a `g_xxxxxxxx` checkpoint has)."""
    s = OrderedDict()
    c0 = h["upsample_initial_channel"]

    def conv(name, cout, cin, k, transposed=False):
        s[name + ".bias"] = (cout,)
        d0 = cin if transposed else cout
        s[name + ".weight_g"] = (d0, 1, 1)
        s[name + ".weight_v"] = (cin, cout, k) if transposed else (cout, cin, k)

    conv("conv_pre", c0, h.get("num_mels", 80), 7)
    for i, k in enumerate(h["upsample_kernel_sizes"]):
        conv(f"ups.{i}", c0 // 2 ** (i + 1), c0 // 2 ** i, k, transposed=True)
    nk = len(h["resblock_kernel_sizes"])
    ch = c0
    for i in range(len(h["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            for grp in ("convs1", "convs2"):
                for m in range(len(h["resblock_dilation_sizes"][j])):
                    conv(f"resblocks.{i * nk + j}.{grp}.{m}", ch, ch, k)
    conv("conv_post", 1, ch, 7)
    return s


def _rule(name: str, shape) -> Tuple[float, float]:
    """(scale, shift) of the recipe for parameter `name`."""
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    # ---- acoustic model
    if name == "null_cond":
        return 1.0, 0.0
    if name.endswith("to_gamma.weight"):
        return 0.015, 0.0
    if name.endswith("to_gamma.bias"):
        return 0.1, 1.0
    if name.endswith("to_beta.weight"):
        return 0.01, 0.0
    if name.endswith("to_beta.bias"):
        return 0.1, 0.0
    if name.endswith("final_norm.gamma"):
        return 0.1, 1.0
    if name == "sinu_pos_emb.0.weights" or name == "to_phoneme_emb.weight":
        return 1.0, 0.0
    # ---- text2semantic (N1)
    if name in ("semantic_token_emb.weight", "token_emb.text.weight"):
        return 0.1, 0.0                      # tied logits: std(logit) ~ 0.1 * sqrt(dim) ~ 2.3
    if name.startswith("start_token.") or name.endswith(".null_kv"):
        return 1.0, 0.0
    if name.endswith(".gamma"):
        return 0.1, 1.0
    # ---- HuBERT tokeniser (N4)
    if "layer_norm." in name or name.startswith("feature_extractor.conv_layers.0.2."):
        return (0.1, 1.0) if name.endswith(".weight") else (0.05, 0.0)
    if name.startswith("feature_extractor.conv_layers."):
        return float(np.sqrt(2.0 / max(fan_in, 1))), 0.0
    if ".self_attn.q_proj.weight" in name or ".self_attn.k_proj.weight" in name:
        return 3.0 / np.sqrt(max(fan_in, 1)), 0.0          # peaky attention: a random post-LN stack otherwise collapses over frames
    if ".self_attn.v_proj.weight" in name or ".self_attn.out_proj.weight" in name:
        return 0.6 / np.sqrt(max(fan_in, 1)), 0.0
    # ---- vocoder (weight-norm parametrisation)
    if name.endswith(".weight_g"):
        g = 0.3 if name.startswith("conv_post") else 1.2
        return 0.2 * g, g
    if name.endswith(".weight_v"):
        return 1.0, 0.0
    # ---- generic
    if name.endswith(".bias"):
        return 0.02, 0.0
    return 1.0 / np.sqrt(max(fan_in, 1)), 0.0


_ARRAY_CACHE: "OrderedDict" = OrderedDict()      # (name, shape, seed) -> array: the recipe is deterministic and RandomState is slow
_ARRAY_CACHE_BYTES = [0]                          # (the 250 M parameters of VoMix take 3-10 s; tests and bench.py ask for them again and again)
_ARRAY_CACHE_LIMIT = 5 << 30


def synth_array(name: str, shape, seed: int = 0) -> np.ndarray:
    """One recipe tensor.  Results are kept in a process-wide cache of at most 5 GB (least recently used first out) and the SAME
    array is handed out again: treat it as read-only (copy before writing into it - every caller in this repo does)."""
    key = (name, tuple(shape), int(seed))
    hit = _ARRAY_CACHE.get(key)
    if hit is not None:
        _ARRAY_CACHE.move_to_end(key)
        return hit
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0xFFFFFFFF)
    scale, shift = _rule(name, tuple(shape))
    a = (rs.standard_normal(tuple(shape)) * scale + shift).astype(np.float32)
    _ARRAY_CACHE[key] = a
    _ARRAY_CACHE_BYTES[0] += a.nbytes
    while _ARRAY_CACHE_BYTES[0] > _ARRAY_CACHE_LIMIT and len(_ARRAY_CACHE) > 1:
        _, old = _ARRAY_CACHE.popitem(last=False)
        _ARRAY_CACHE_BYTES[0] -= old.nbytes
    return a


def synth_state_dict(shapes: Shapes, seed: int = 0) -> "Dict[str, np.ndarray]":
    return OrderedDict((k, synth_array(k, v, seed)) for k, v in shapes.items())


def t2s_param_shapes(two_output: bool = False, dim: int = 512, dim_target: Optional[int] = None, source_depth: int = 4,
                     target_depth: int = 4, heads: int = 8, num_text: int = 30530, num_semantic: int = 501) -> Shapes:
    """Parameters of the reference TextToSemantic in nn.Module.parameters() order (text2semantic.py:405-600;
    running_command/T2S_CoSingle.sh: dim 512; T2S_CoMix.sh: two_output, target dim 1024).  The rotary `freqs`
    parameter is shared by the layers of one transformer and appears once (under layers.0.0)."""
    dim_t = dim_target or dim
    inner = heads * 64
    emb = dim_t // 2 if two_output else dim_t
    sh: Shapes = OrderedDict()
    sh["semantic_token_emb.weight"] = (num_semantic + 1, emb)
    sh["token_emb.text.weight"] = (num_text + 1, dim)
    sh["start_token.speech"] = (dim_t,)
    sh["start_token.text"] = (dim,)

    def attn(p, d, d_ctx, first, null):
        if first:
            sh[p + ".rotary_emb.freqs"] = (32,)
        if null:
            sh[p + ".null_kv"] = (2, heads, 1, 64)
        sh[p + ".norm.gamma"] = (d,)
        sh[p + ".to_q.0.weight"] = (inner, d)
        sh[p + ".to_kv.0.weight"] = (2 * inner, d_ctx)
        sh[p + ".to_out.weight"] = (d, inner)

    def ff(p, d):
        di = int(d * 4 * 2 / 3)
        sh[p + ".0.gamma"] = (d,)
        sh[p + ".1.weight"] = (2 * di, d)
        sh[p + ".1.bias"] = (2 * di,)
        sh[p + ".4.weight"] = (d, di)
        sh[p + ".4.bias"] = (d,)

    for i in range(source_depth):
        attn(f"source_transformer.layers.{i}.0", dim, dim, i == 0, False)
        ff(f"source_transformer.layers.{i}.2", dim)
    sh["source_transformer.final_norm.gamma"] = (dim,)
    for i in range(target_depth):
        attn(f"target_transformer.layers.{i}.0", dim_t, dim_t, i == 0, False)
        attn(f"target_transformer.layers.{i}.1", dim_t, dim, False, True)
        ff(f"target_transformer.layers.{i}.2", dim_t)
    sh["target_transformer.final_norm.gamma"] = (dim_t,)
    return sh


def t2s_state_dict(shapes: Shapes, seed: int = 0) -> "Dict[str, np.ndarray]":
    """Recipe weights with the two rotary `freqs` parameters set to their real (non-learned) values."""
    sd = synth_state_dict(shapes, seed)
    for k in sd:
        if k.endswith("rotary_emb.freqs"):
            sd[k] = np.asarray(rotary_inv_freq(64), dtype=np.float32)
    return sd


def hubert_param_shapes(conv_layers=((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2, dim: int = 768, ffn: int = 3072,
                        depth: int = 12, conv_pos: int = 128, conv_pos_groups: int = 16) -> Shapes:
    """state_dict of the reference HubertModel at HuBERT-Base geometry, in its own order
    (fairseq-hubert/fairseq/models/hubert/hubert.py:249-329, models/wav2vec/wav2vec2.py:844-947,1012-1063,1261-1306;
    extractor_mode "default": GroupNorm after the first conv only, no conv bias; post-LN encoder layers)."""
    sh: Shapes = OrderedDict()
    sh["mask_emb"] = (dim,)
    cin = 1
    for i, (c, k, _s) in enumerate(conv_layers):
        sh[f"feature_extractor.conv_layers.{i}.0.weight"] = (c, cin, k)
        if i == 0:
            sh["feature_extractor.conv_layers.0.2.weight"] = (c,)
            sh["feature_extractor.conv_layers.0.2.bias"] = (c,)
        cin = c
    sh["post_extract_proj.weight"] = (dim, cin)
    sh["post_extract_proj.bias"] = (dim,)
    sh["encoder.pos_conv.0.bias"] = (dim,)
    sh["encoder.pos_conv.0.weight_g"] = (1, 1, conv_pos)
    sh["encoder.pos_conv.0.weight_v"] = (dim, dim // conv_pos_groups, conv_pos)
    for i in range(depth):
        p = f"encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            sh[p + f"self_attn.{nm}.weight"] = (dim, dim)
            sh[p + f"self_attn.{nm}.bias"] = (dim,)
        sh[p + "self_attn_layer_norm.weight"] = (dim,)
        sh[p + "self_attn_layer_norm.bias"] = (dim,)
        sh[p + "fc1.weight"] = (ffn, dim)
        sh[p + "fc1.bias"] = (ffn,)
        sh[p + "fc2.weight"] = (dim, ffn)
        sh[p + "fc2.bias"] = (dim,)
        sh[p + "final_layer_norm.weight"] = (dim,)
        sh[p + "final_layer_norm.bias"] = (dim,)
    sh["encoder.layer_norm.weight"] = (dim,)
    sh["encoder.layer_norm.bias"] = (dim,)
    sh["layer_norm.weight"] = (cin,)
    sh["layer_norm.bias"] = (cin,)
    sh["final_proj.weight"] = (dim, dim)
    sh["final_proj.bias"] = (dim,)
    return sh


def hubert_state_dict(seed: int = 0, **geometry) -> "Dict[str, np.ndarray]":
    return synth_state_dict(hubert_param_shapes(**geometry), seed)


def hubert_kmeans_centers(seed: int = 0, n_clusters: int = 500, dim: int = 768) -> np.ndarray:
    """Stand-in for `km_model.cluster_centers_` (dump_km_label.py:27-28): float32 [n_clusters, dim]."""
    rs = np.random.RandomState((zlib.crc32(b"hubert.kmeans.cluster_centers_") ^ seed) & 0xFFFFFFFF)
    return rs.standard_normal((n_clusters, dim)).astype(np.float32)


def hubert_kmeans_centers_near(features: np.ndarray, seed: int = 0, n_clusters: int = 500, jitter: float = 0.35) -> np.ndarray:
    """Cluster centres that LIVE WHERE THE FEATURES DO: centre j = frame (j mod T) of `features` [T, dim] plus Gaussian jitter of
    `jitter` x the features' RMS.  With N(0, 1) centres (hubert_kmeans_centers) a random-weight network's layer-12 frames all
    fall into one or two cells - the labels hardly test the argmin (round-3 review); with these every frame has its own handful
    of nearby centres (about n_clusters / T jittered copies of itself and of its neighbours), so a fixture sees dozens of distinct
    labels with margins from ~0 upwards."""
    feats = np.asarray(features, dtype=np.float32)
    rs = np.random.RandomState((zlib.crc32(b"hubert.kmeans.centers_near") ^ seed) & 0xFFFFFFFF)
    rms = float(np.sqrt((feats.astype(np.float64) ** 2).mean()))
    idx = np.arange(n_clusters) % feats.shape[0]
    return (feats[idx] + jitter * rms * rs.standard_normal((n_clusters, feats.shape[1]))).astype(np.float32)


def rotary_inv_freq(dim_head: int = 64, theta: float = 10000.0) -> np.ndarray:
    """The `transformer.rotary_emb.inv_freq` buffer (acoustic.py:117-120), computed the
    way torch does it in fp32: 1 / theta ** (arange(0, d, 2) / d)."""
    import torch
    return (1.0 / (theta ** (torch.arange(0, dim_head, 2).float() / dim_head))).numpy()


HIFIGAN_COVOMIX_CONFIG = {
    # hifi-gan/config_covomix.json:1-37 (the fields Generator reads)
    "resblock": "1",
    "upsample_rates": [5, 4, 4, 2],
    "upsample_kernel_sizes": [8, 8, 4, 4],
    "upsample_initial_channel": 500,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "num_mels": 80,
    "sampling_rate": 8000,
    "hop_size": 160,
    # the fields its command-line callers read (hifi-gan/inference.py:30-31, inference_e2e.py:84)
    "n_fft": 480,
    "win_size": 480,
    "fmin": 0,
    "fmax": 4000,
    "seed": 1234,
}


def synthetic_inputs(kind: str, batch: int, frames: int, prompt: int, seed: int = 1234):
    """SURVEY.md section 8(d) synthetic inputs (seeded torch CPU generator).

    kind = 'vomix'   : ids [B,T,2] (stream B is 157 on alternate 100-frame turns), cond [B,T,160]
    kind = 'vosingle': ids [B,T], cond [B,T,80]
    Returns dict(phoneme_ids, cond, mask, y0).  mel ~ N(-6, 2^2) clipped to [-11.52, 2].
    """
    import torch
    g = torch.Generator().manual_seed(seed)
    two = kind == "vomix"
    cdim = 160 if two else 80
    ids = torch.randint(0, 501, (batch, frames, 2) if two else (batch, frames), generator=g)
    if two:
        turn = (torch.arange(frames) // 100) % 2
        ids[:, turn == 0, 1] = 157
        ids[:, turn == 1, 0] = 157
    mel = (torch.randn(batch, frames, cdim, generator=g) * 2.0 - 6.0).clamp(-11.52, 2.0)
    cond = torch.zeros(batch, frames, cdim)
    cond[:, :prompt] = mel[:, :prompt]
    mask = torch.zeros(batch, frames, dtype=torch.bool)
    mask[:, prompt:] = True
    y0 = torch.randn(batch, frames, 80, generator=g)
    return dict(phoneme_ids=ids, cond=cond, mask=mask, y0=y0)
