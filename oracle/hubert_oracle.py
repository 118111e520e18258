"""CPU oracle for SURVEY.md section 8(f) row N4: HuBERT layer-L features + k-means labels (the prompt tokeniser).

TEST INFRASTRUCTURE ONLY.  Imported by tests/ (and nothing else); the product path (covomix_amd.hubert) never
touches it.  Parity pinned: tests/test_hubert_oracle.py checks this restatement against tests/golden/hubert_base.npz,
which tests/golden/make_golden_hubert.py produced by running the reference's own HubertModel / ApplyKmeans classes.

Restates, in plain torch CPU ops (fp32 by default, fp64 on request):
  * HubertFeatureReader.get_feats            fairseq-hubert/examples/textless_nlp/gslm/speech2unit/pretrained/hubert_feature_reader.py:58-78
      optional F.layer_norm over the whole waveform (:66-67), chunks of max_chunk samples (:70-77)
  * HubertModel.forward(features_only, mask=False, output_layer=L)   fairseq-hubert/fairseq/models/hubert/hubert.py:433-480
      conv feature extractor -> transpose -> LayerNorm(512) -> post_extract_proj -> encoder(layer = L - 1)
  * ConvFeatureExtractionModel ("default" mode)   fairseq/models/wav2vec/wav2vec2.py:844-923
      7 x Conv1d (no bias) + GELU, GroupNorm(512, 512) after the first conv only
  * TransformerEncoder.extract_features        wav2vec2.py:1078-1163
      x + GELU(SamePad(weight-normed grouped Conv1d k=128 g=16))  (:925-946), LayerNorm, 12 post-LN layers
      (padding to a multiple of 2 frames adds one masked key and is dropped again: no effect on the T real frames)
  * TransformerSentenceEncoderLayer (layer_norm_first=False)   wav2vec2.py:1343-1370
      x = LN(x + out_proj(softmax((q_proj x) * d^-0.5 . k_proj x) v_proj x));  x = LN(x + fc2(gelu(fc1 x)))
  * ApplyKmeans.__call__                         fairseq-hubert/examples/hubert/simple_kmeans/dump_km_label.py:25-43
      argmin_j ( |x|^2 - 2 x.C_j + |C_j|^2 )
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

CONV_LAYERS = ((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2      # hubert.py:136-137 default


def _t(sd, name, dtype):
    return torch.as_tensor(np.asarray(sd[name])).to(dtype)


def conv_features(sd: Dict[str, np.ndarray], wav: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """wav [B, n] -> [B, T, 512]  (wav2vec2.py:914-923 + the transpose of hubert.py:451)."""
    x = wav.to(dtype).unsqueeze(1)
    for i, (_c, _k, stride) in enumerate(CONV_LAYERS):
        x = F.conv1d(x, _t(sd, f"feature_extractor.conv_layers.{i}.0.weight", dtype), stride=stride)
        if i == 0:                                                              # Fp32GroupNorm(dim, dim): one group per channel
            x = F.group_norm(x, x.shape[1], _t(sd, "feature_extractor.conv_layers.0.2.weight", dtype),
                             _t(sd, "feature_extractor.conv_layers.0.2.bias", dtype), 1e-5)
        x = F.gelu(x)
    return x.transpose(1, 2)


def pos_conv_weight(sd, dtype=torch.float32) -> torch.Tensor:
    """weight_norm(dim=2): w = v * g / ||v|| with the norm over dims (0, 1) for every kernel position (wav2vec2.py:939)."""
    v = _t(sd, "encoder.pos_conv.0.weight_v", torch.float64)
    g = _t(sd, "encoder.pos_conv.0.weight_g", torch.float64)
    return (v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())).to(dtype)


def encoder(sd, x: torch.Tensor, n_layers: int, heads: int = 12, groups: int = 16, dtype=torch.float32) -> torch.Tensor:
    """x [B, T, D] -> output of encoder layer n_layers (1-based), wav2vec2.py:1078-1163."""
    B, T, D = x.shape
    w = pos_conv_weight(sd, dtype)
    k = w.shape[-1]
    pc = F.conv1d(x.transpose(1, 2), w, _t(sd, "encoder.pos_conv.0.bias", dtype), padding=k // 2, groups=groups)
    if k % 2 == 0:
        pc = pc[:, :, :-1]                                                      # SamePad
    x = x + F.gelu(pc).transpose(1, 2)
    x = F.layer_norm(x, (D,), _t(sd, "encoder.layer_norm.weight", dtype), _t(sd, "encoder.layer_norm.bias", dtype), 1e-5)
    dh = D // heads
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        lin = lambda t, nm: F.linear(t, _t(sd, p + nm + ".weight", dtype), _t(sd, p + nm + ".bias", dtype))
        q = (lin(x, "self_attn.q_proj") * dh ** -0.5).view(B, T, heads, dh).transpose(1, 2)
        kk = lin(x, "self_attn.k_proj").view(B, T, heads, dh).transpose(1, 2)
        v = lin(x, "self_attn.v_proj").view(B, T, heads, dh).transpose(1, 2)
        a = torch.softmax(q @ kk.transpose(-1, -2), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, D)
        x = x + lin(a, "self_attn.out_proj")
        x = F.layer_norm(x, (D,), _t(sd, p + "self_attn_layer_norm.weight", dtype), _t(sd, p + "self_attn_layer_norm.bias", dtype), 1e-5)
        h = F.gelu(lin(x, "fc1"))
        x = x + lin(h, "fc2")
        x = F.layer_norm(x, (D,), _t(sd, p + "final_layer_norm.weight", dtype), _t(sd, p + "final_layer_norm.bias", dtype), 1e-5)
    return x


def extract_features(sd, wav: torch.Tensor, output_layer: int = 12, dtype=torch.float32) -> torch.Tensor:
    """HubertModel.extract_features(source, mask=False, output_layer) -> [B, T, 768]  (hubert.py:433-480, 533-549)."""
    f = conv_features(sd, wav, dtype)
    f = F.layer_norm(f, (f.shape[-1],), _t(sd, "layer_norm.weight", dtype), _t(sd, "layer_norm.bias", dtype), 1e-5)
    x = F.linear(f, _t(sd, "post_extract_proj.weight", dtype), _t(sd, "post_extract_proj.bias", dtype))
    return encoder(sd, x, output_layer, dtype=dtype)


def get_feats(sd, wav: np.ndarray, layer: int = 12, normalize: bool = False, max_chunk: int = 1600000,
              dtype=torch.float32) -> torch.Tensor:
    """hubert_feature_reader.py:58-78 on an already loaded mono waveform -> [T, 768]."""
    x = torch.from_numpy(np.asarray(wav)).float()
    if normalize:
        x = F.layer_norm(x, x.shape)
    x = x.view(1, -1)
    out = [extract_features(sd, x[:, s: s + max_chunk], layer, dtype) for s in range(0, x.shape[1], max_chunk)]
    return torch.cat(out, 1).squeeze(0)


def apply_kmeans(centers: np.ndarray, feats: torch.Tensor) -> np.ndarray:
    """dump_km_label.py:25-43 with C = cluster_centers_.T."""
    C = torch.from_numpy(np.ascontiguousarray(np.asarray(centers).T)).to(feats.dtype)
    cnorm = (C ** 2).sum(0, keepdim=True)
    dist = feats.pow(2).sum(1, keepdim=True) - 2 * torch.matmul(feats, C) + cnorm
    return dist.argmin(dim=1).numpy()


def frames_for(n_samples: int) -> int:
    """Number of 20-ms frames the conv stack yields for n_samples (wav2vec2.py:579-594)."""
    n = n_samples
    for _c, k, s in CONV_LAYERS:
        n = (n - k) // s + 1
    return max(n, 0)


def sinc_resample(wav: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """torchaudio.functional.resample (sinc_interp_hann defaults; what torchaudio.transforms.Resample applies in
    hubert_feature_reader.py:38-41), restated from the published algorithm.  PARITY UNPINNED: torchaudio is a
    third-party package that is in neither this image nor the reference tree; tests check filter properties and
    agreement with scipy's polyphase resampler instead."""
    import math
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if orig == new:
        return np.asarray(wav, dtype=np.float32)
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base / orig)
    kernels = kernels.to(torch.float32)
    x = torch.from_numpy(np.asarray(wav, dtype=np.float32)).view(1, -1)
    n = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernels, stride=orig).transpose(1, 2).reshape(1, -1)
    return y[0, : math.ceil(new * n / orig)].numpy()
