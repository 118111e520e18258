"""HuBERT layer-L features + k-means labels on MI355X: the prompt tokeniser.

Host-side mirror of the reference's classes, same names / arguments / return types:
  HubertTokenizer      fairseq-hubert/examples/textless_nlp/dgslm/dgslm_utils.py:19-44   (wav2code, wav2codes)
  HubertFeatureReader  fairseq-hubert/examples/textless_nlp/gslm/speech2unit/pretrained/hubert_feature_reader.py:15-78
  ApplyKmeans          fairseq-hubert/examples/hubert/simple_kmeans/dump_km_label.py:25-50
as driven by fairseq-hubert/get_fisher_semantic_tokens.py:30-40 (`encoder.wav2code(file, 1)` -> "<name>.hubert_code.npy").

The network (fairseq HubertModel.extract_features, hubert.py:433-480,533-549; wav2vec2.py:844-946,1078-1163,1343-1370)
runs entirely through the C ABI of libcovomix_hip.so, activations channels-last [frames, channels] fp32:
  conv layer 0 + GroupNorm + GELU      cvx_hubert_conv0_gn_gelu_f32
  conv layers 1-6 (+GELU)              cvx_gemm_bias_act_f32 over OVERLAPPING rows (lda = stride*C, K = k*C)
  LayerNorm(512), post_extract_proj    cvx_layernorm_f32, GEMM
  positional conv (k=128, 16 groups)   cvx_hubert_group_pack_f32 + one GEMM per group (bias + GELU + residual epilogue)
  12 post-LN layers                    GEMM (q|k|v fused), cvx_attention_f32, GEMM(+residual), LayerNorm, GEMM(+GELU),
                                       GEMM(+residual), LayerNorm
  k-means                              GEMM against cluster_centers_, cvx_kmeans_argmin_f32
torch is used for device memory and one-time weight re-layouts only.  There is no CPU path: `use_cuda=False` raises.
"""
from __future__ import annotations

import ast
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import ops

DEFAULT_CONV_LAYERS = "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"       # hubert.py:136-137


def parse_conv_layers(spec: str) -> List[Tuple[int, int, int]]:
    """The reference eval()s cfg.conv_feature_layers (hubert.py:258); here only list / tuple / int / + / * are accepted."""
    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, int):
            return n.value
        if isinstance(n, (ast.List, ast.Tuple)):
            v = [ev(e) for e in n.elts]
            return v if isinstance(n, ast.List) else tuple(v)
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Add):
            return ev(n.left) + ev(n.right)
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Mult):
            return ev(n.left) * ev(n.right)
        raise ValueError(f"conv_feature_layers: unsupported expression {ast.dump(n)}")
    layers = ev(ast.parse(spec.strip(), mode="eval"))
    out = [tuple(int(v) for v in cl) for cl in layers]
    assert all(len(cl) == 3 for cl in out), "invalid conv definition: " + str(out)       # wav2vec2.py:896
    return out


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Filter bank of torchaudio.functional.resample (resampling_method "sinc_interp_hann", the defaults of
    torchaudio.transforms.Resample that hubert_feature_reader.py:39 constructs) -> (kernels [new, 2*width + orig] fp32,
    width, orig, new) with the rates divided by their gcd.  torchaudio is third-party and in neither this image nor the
    reference tree: restated from its published algorithm, PARITY UNPINNED."""
    import math
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    return (k * window * (base / orig)).astype(np.float32), width, orig, new


def resample(wav: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """1-D fp32 waveform on the GPU -> resampled waveform, length ceil(new * n / orig)."""
    if int(orig_freq) == int(new_freq):
        return wav
    from . import _lib
    kern, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    wav = wav.reshape(-1).float().contiguous()
    n = wav.numel()
    n_out = -(-new * n // orig)
    out = torch.empty(n_out, dtype=torch.float32, device=wav.device)
    k = torch.from_numpy(kern).to(wav.device)
    _lib.check(_lib.load().cvx_resample_fir_f32(wav.data_ptr(), n, k.data_ptr(), new, orig, width, out.data_ptr(), n_out,
                                                ops._stream()), "cvx_resample_fir_f32")
    return out


class HubertEncoder:
    """Device-resident HuBERT weights in the layouts the kernels read + the forward pass."""

    def __init__(self, state_dict: Dict[str, Union[np.ndarray, torch.Tensor]], model_cfg: Optional[dict] = None,
                 device: Union[str, torch.device] = "cuda", precision: str = "f16x3", stepped: bool = False):
        """stepped=True drives every kernel from Python (tests, tools); the default enqueues the whole network from one
        C call (cvx_hubert_extract_features), which is what keeps the GPU busy.  precision: "f16x3" (default) runs every GEMM with more than 64 rows on the split-precision MFMA kernel
        (fp32 operands as fp16 hi/lo pairs, three products, fp32 accumulate - 16 k per accumulate step, measured closer
        to an fp64 evaluation than the fp32 MFMA kernel, which adds 2 k per step); "fp32" uses cvx_gemm_bias_act_f32 only."""
        assert precision in ("f16x3", "fp32")
        self.precision = precision
        self.stepped = bool(stepped) or precision == "fp32"      # the C entry point is the f16x3 path
        self._cmodel = None
        self._ws: Optional[torch.Tensor] = None
        cfg = dict(model_cfg or {})
        if cfg.get("layer_norm_first", False) or cfg.get("extractor_mode", "default") != "default":
            raise NotImplementedError("only HuBERT-Base style checkpoints (extractor_mode=default, post-LN) are supported")
        if cfg.get("conv_bias", False) or int(cfg.get("pos_conv_depth", 1)) != 1:
            raise NotImplementedError("conv_bias / pos_conv_depth > 1 checkpoints are not supported")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("covomix_amd.hubert has no CPU path (device must be a GPU)")
        f = lambda n: torch.as_tensor(np.asarray(state_dict[n]) if not isinstance(state_dict[n], torch.Tensor) else state_dict[n]) \
            .to(device=self.device, dtype=torch.float32).contiguous()
        self.conv_layers = parse_conv_layers(cfg.get("conv_feature_layers", DEFAULT_CONV_LAYERS))
        self.heads = int(cfg.get("encoder_attention_heads", 12))
        self.groups = int(cfg.get("conv_pos_groups", 16))
        # --- conv feature extractor (wav2vec2.py:844-923)
        w0 = f("feature_extractor.conv_layers.0.0.weight")
        assert w0.shape == (self.conv_layers[0][0], 1, self.conv_layers[0][1])
        self.w0 = w0.reshape(w0.shape[0], w0.shape[2]).contiguous()
        self.gn_w, self.gn_b = f("feature_extractor.conv_layers.0.2.weight"), f("feature_extractor.conv_layers.0.2.bias")
        self.conv_w = []
        for i, (c, k, _s) in enumerate(self.conv_layers[1:], start=1):
            w = f(f"feature_extractor.conv_layers.{i}.0.weight")                          # [C_out, C_in, k]
            assert w.shape[0] == c and w.shape[2] == k
            self.conv_w.append(w.permute(0, 2, 1).reshape(c, k * w.shape[1]).contiguous())   # row = (tap, channel): one run of the input
        self.ln_w, self.ln_b = f("layer_norm.weight"), f("layer_norm.bias")
        self.proj_w, self.proj_b = f("post_extract_proj.weight"), f("post_extract_proj.bias")
        self.dim = self.proj_w.shape[0]
        assert self.dim % (64 * self.heads) == 0 and self.dim // self.heads == 64, "attention kernel: head dim must be 64"
        # --- positional convolution: fold weight_norm(dim=2) (wav2vec2.py:939), split by group, tap-major rows
        v, g = f("encoder.pos_conv.0.weight_v"), f("encoder.pos_conv.0.weight_g")
        w = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        self.pos_k = w.shape[2]
        cg = self.dim // self.groups
        assert w.shape == (self.dim, cg, self.pos_k)
        self.pos_w = [w[gi * cg:(gi + 1) * cg].permute(0, 2, 1).reshape(cg, self.pos_k * cg).contiguous() for gi in range(self.groups)]
        self.pos_b = f("encoder.pos_conv.0.bias")
        self.enc_ln = (f("encoder.layer_norm.weight"), f("encoder.layer_norm.bias"))
        # --- transformer layers (wav2vec2.py:1261-1370)
        self.layers = []
        i = 0
        while f"encoder.layers.{i}.fc1.weight" in state_dict:
            p = f"encoder.layers.{i}."
            self.layers.append(dict(
                wqkv=torch.cat([f(p + f"self_attn.{n}_proj.weight") for n in "qkv"], 0).contiguous(),
                bqkv=torch.cat([f(p + f"self_attn.{n}_proj.bias") for n in "qkv"], 0).contiguous(),
                wo=f(p + "self_attn.out_proj.weight"), bo=f(p + "self_attn.out_proj.bias"),
                ln1=(f(p + "self_attn_layer_norm.weight"), f(p + "self_attn_layer_norm.bias")),
                w1=f(p + "fc1.weight"), b1=f(p + "fc1.bias"), w2=f(p + "fc2.weight"), b2=f(p + "fc2.bias"),
                ln2=(f(p + "final_layer_norm.weight"), f(p + "final_layer_norm.bias"))))
            i += 1

        self._split: Dict[int, tuple] = {}

    def _gemm(self, a, w, out, a16=None, **kw):
        """One nn.Linear / conv-as-GEMM.  f16x3 mode: split weights are made once per weight tensor (load-time packing)
        and the A operand is handed over as a pre-split (fp16 hi, fp16 lo) pair - `a16`, or split here when the producer
        only wrote fp32 - so that every tile arrives by LDS-DMA and small problems may split K (see the header)."""
        if self.precision == "f16x3" and w.shape[1] % 32 == 0:
            ws = self._split.get(id(w))
            if ws is None:
                ws = self._split[id(w)] = ops.split_f16(w)
            if a16 is None:
                a16 = ops.split_act_f16(a)
            return ops.gemm(a, w, out, w_split=ws, a_split=a16, **kw)
        kw.pop("out_split", None)
        kw.pop("write_f32", None)
        return ops.gemm(a, w, out, **kw)

    @property
    def _f16x3(self) -> bool:
        return self.precision == "f16x3"

    @staticmethod
    def _windows(pair, rows: int, width: int, stride: int):
        """Overlapping-row views (rows x width, row stride `stride`) of both halves of a split pair."""
        return tuple(t.as_strided((rows, width), (stride, 1), t.storage_offset()) for t in pair)

    # ------------------------------------------------------------------ pieces
    def n_frames(self, n_samples: int) -> int:
        n = n_samples
        for _c, k, s in self.conv_layers:
            n = (n - k) // s + 1 if n >= k else 0
        return max(n, 0)

    def conv_features(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [n] fp32 on the device -> [T, 512] (ConvFeatureExtractionModel.forward + transpose)."""
        x = ops.hubert_conv0_gn_gelu(wav, self.w0, self.gn_w, self.gn_b, self.conv_layers[0][2])
        x16 = ops.split_act_f16(x) if self._f16x3 else None
        last = len(self.conv_w) - 1
        for i, ((c, k, s), w) in enumerate(zip(self.conv_layers[1:], self.conv_w)):
            L, cin = x.shape
            Lout = (L - k) // s + 1
            a = x.as_strided((Lout, k * cin), (s * cin, 1))           # im2col row = k consecutive channels-last frames
            y = torch.empty(Lout, c, dtype=torch.float32, device=x.device)
            if self._f16x3 and k * cin % 32 == 0:
                # the GELU epilogue writes the split pair the next layer reads; fp32 only after the last layer
                y16 = None if i == last else (torch.empty(Lout, c, dtype=torch.float16, device=x.device),
                                              torch.empty(Lout, c, dtype=torch.float16, device=x.device))
                self._gemm(a, w, y, a16=self._windows(x16, Lout, k * cin, s * cin), act=ops.ACT_GELU,
                           out_split=y16, write_f32=(i == last))
                x16 = y16
            else:
                self._gemm(a, w, y, act=ops.ACT_GELU)
            x = y
        return x

    def encode(self, feats: torch.Tensor, output_layer: Optional[int] = None) -> torch.Tensor:
        """[T, 512] conv features -> output of transformer layer `output_layer` (1-based; None = last), [T, D]."""
        T = feats.shape[0]
        D, G, k = self.dim, self.groups, self.pos_k
        cg = D // G
        dev = feats.device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        new16 = lambda *shape: (torch.empty(*shape, dtype=torch.float16, device=dev), torch.empty(*shape, dtype=torch.float16, device=dev))
        h = self._gemm(ops.layernorm(feats, self.ln_w, self.ln_b), self.proj_w, new(T, D), bias=self.proj_b)
        packed = ops.hubert_group_pack(h, G, k // 2)                  # [G, T + k, cg], zero halos
        packed16 = ops.split_act_f16(packed.view(-1, cg)) if self._f16x3 else None
        x = new(T, D)
        for g in range(G):
            a = packed[g].as_strided((T, k * cg), (cg, 1))
            a16 = None
            if packed16 is not None:
                a16 = tuple(t[g * (T + k):(g + 1) * (T + k)].as_strided((T, k * cg), (cg, 1), t.storage_offset() + g * (T + k) * cg)
                            for t in packed16)
            self._gemm(a, self.pos_w[g], x[:, g * cg:(g + 1) * cg], a16=a16, bias=self.pos_b[g * cg:(g + 1) * cg], act=ops.ACT_GELU,
                       residual=h[:, g * cg:(g + 1) * cg])             # x = h + gelu(conv(h) + b)
        x = ops.layernorm(x, *self.enc_ln)
        n_layers = len(self.layers) if output_layer is None else int(output_layer)
        assert 0 <= n_layers <= len(self.layers), f"output_layer {output_layer} out of range"
        qkv, att, y = new(T, 3 * D), new(T, D), new(T, D)
        F = self.layers[0]["w1"].shape[0] if self.layers else 0
        ff = new(T, F)
        att16, ff16 = (new16(T, D), new16(T, F)) if self._f16x3 else (None, None)
        for lyr in self.layers[:n_layers]:
            self._gemm(x, lyr["wqkv"], qkv, bias=lyr["bqkv"])
            if att16 is not None:                                      # the attention writes the split pair out_proj reads
                ops.attention(qkv, None, 1, T, self.heads, 64 ** -0.5, out_split=att16)
            else:
                ops.attention(qkv, att, 1, T, self.heads, 64 ** -0.5)
            self._gemm(att, lyr["wo"], y, a16=att16, bias=lyr["bo"], residual=x)
            x = ops.layernorm(y, *lyr["ln1"], out=x)
            if ff16 is not None:
                self._gemm(x, lyr["w1"], ff, bias=lyr["b1"], act=ops.ACT_GELU, out_split=ff16, write_f32=False)
            else:
                self._gemm(x, lyr["w1"], ff, bias=lyr["b1"], act=ops.ACT_GELU)
            self._gemm(ff, lyr["w2"], y, a16=ff16, bias=lyr["b2"], residual=x)
            x = ops.layernorm(y, *lyr["ln2"], out=x)
        return x

    @ops.gated
    def extract_features(self, source: torch.Tensor, output_layer: Optional[int] = None) -> torch.Tensor:
        """HubertModel.extract_features(source [1, n] or [n], mask=False, output_layer) -> [T, D]."""
        wav = source.reshape(-1).to(device=self.device, dtype=torch.float32).contiguous()
        T = self.n_frames(wav.numel())
        if T == 0:
            return torch.empty(0, self.dim, dtype=torch.float32, device=self.device)
        if self.stepped:
            return self.encode(self.conv_features(wav), output_layer)
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        m = self._c_model()
        n_layers = len(self.layers) if output_layer is None else int(output_layer)
        assert 0 <= n_layers <= len(self.layers), f"output_layer {output_layer} out of range"
        need = int(lib.cvx_hubert_workspace_bytes(C.byref(m), wav.numel()))
        if need < 0:
            _lib.check(-22, "cvx_hubert_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None                                        # release before growing
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(T, self.dim, dtype=torch.float32, device=self.device)
        _lib.check(lib.cvx_hubert_extract_features(C.byref(m), wav.data_ptr(), wav.numel(), n_layers, out.data_ptr(),
                                                   self._ws.data_ptr(), self._ws.numel(), ops._stream()),
                   "cvx_hubert_extract_features")
        return out

    def _c_model(self):
        """cvx_hubert_model over this object's device tensors (built once; the tensors stay referenced by self)."""
        if self._cmodel is not None:
            return self._cmodel
        from . import _lib
        keep = self._keep = []

        def lin(w, bias=None):
            ws = self._split.get(id(w))
            if ws is None:
                ws = self._split[id(w)] = ops.split_f16(w)
            hi, lo, inv = ws
            keep.extend((w, hi, lo, bias))
            L = _lib.Linear()
            L.w, L.w_hi, L.w_lo, L.inv_scale = w.data_ptr(), hi.data_ptr(), lo.data_ptr(), inv
            L.bias = bias.data_ptr() if bias is not None else None
            L.N, L.K = w.shape
            return L
        m = _lib.HubertModel()
        m.n_conv = len(self.conv_layers)
        assert m.n_conv <= 8
        for i, (c, k, s) in enumerate(self.conv_layers):
            m.conv_c[i], m.conv_k[i], m.conv_stride[i] = c, k, s
        m.conv0_w, m.gn_g, m.gn_b = self.w0.data_ptr(), self.gn_w.data_ptr(), self.gn_b.data_ptr()
        for i, w in enumerate(self.conv_w, start=1):
            m.conv[i] = lin(w)
        m.ln_g, m.ln_b = self.ln_w.data_ptr(), self.ln_b.data_ptr()
        m.proj = lin(self.proj_w, self.proj_b)
        m.dim, m.heads, m.pos_k, m.pos_groups = self.dim, self.heads, self.pos_k, self.groups
        cg = self.dim // self.groups
        self._pos_bias = [self.pos_b[g * cg:(g + 1) * cg].contiguous() for g in range(self.groups)]
        self._c_pos = (_lib.Linear * self.groups)(*[lin(self.pos_w[g], self._pos_bias[g]) for g in range(self.groups)])
        m.pos = self._c_pos
        m.enc_ln_g, m.enc_ln_b = self.enc_ln[0].data_ptr(), self.enc_ln[1].data_ptr()
        cl = []
        for lyr in self.layers:
            L = _lib.HubertLayer()
            L.qkv, L.out = lin(lyr["wqkv"], lyr["bqkv"]), lin(lyr["wo"], lyr["bo"])
            L.fc1, L.fc2 = lin(lyr["w1"], lyr["b1"]), lin(lyr["w2"], lyr["b2"])
            L.ln1_g, L.ln1_b = lyr["ln1"][0].data_ptr(), lyr["ln1"][1].data_ptr()
            L.ln2_g, L.ln2_b = lyr["ln2"][0].data_ptr(), lyr["ln2"][1].data_ptr()
            cl.append(L)
        self._c_layers = (_lib.HubertLayer * max(len(cl), 1))(*cl)
        m.n_layers, m.layers = len(cl), self._c_layers
        self._cmodel = m
        return m


class HubertFeatureReader:
    """Wrapper class to run inference on a HuBERT checkpoint (hubert_feature_reader.py:15-78)."""

    def __init__(self, checkpoint_path, layer, max_chunk=1600000, use_cuda=True):
        if not use_cuda:
            raise RuntimeError("covomix_amd.hubert has no CPU path (use_cuda must be True)")
        # fairseq checkpoints: {"cfg": plain dict (trainer.py:400-412), "model": state_dict, ...}
        state = torch.load(checkpoint_path, map_location="cpu", weights_only=False) if isinstance(checkpoint_path, str) else checkpoint_path
        cfg = state.get("cfg") or {}
        self.task_cfg = dict(cfg.get("task") or {})
        self.sample_rate = int(self.task_cfg.get("sample_rate", 16000))
        self.normalize = bool(self.task_cfg.get("normalize", False))
        self.model = HubertEncoder(state["model"], dict(cfg.get("model") or {}), device="cuda")
        self.layer = layer
        self.max_chunk = max_chunk
        self.use_cuda = use_cuda

    def read_audio(self, path, ref_len=None, channel_id=None):
        """-> float32 mono waveform at the checkpoint's sample rate (torchaudio.load semantics: int16 / 32768)."""
        from .audio import read_wav
        sr, wav = read_wav(path)                   # float by the STORED sample type (int16 / 2^15, int32 / 2^31, uint8, float)
        if wav.ndim == 2:                          # [n, channels]; the reference asserts mono after squeeze(0) (:51-52)
            if wav.shape[1] == 1:
                wav = wav[:, 0]
            elif channel_id is not None:
                wav = wav[:, int(channel_id) - 1]
        assert wav.ndim == 1, wav.ndim
        if sr != self.sample_rate:                 # resample if needed (:38-41), on the GPU
            wav = resample(torch.from_numpy(np.ascontiguousarray(wav)).cuda(), sr, self.sample_rate).cpu().numpy()
        if ref_len is not None and abs(ref_len - len(wav)) > 160:
            print(f"ref {ref_len} != read {len(wav)} ({path})")
        return np.ascontiguousarray(wav)

    @ops.gated
    def get_feats(self, file_path, ref_len=None, channel_id=None):
        """-> [T, D] fp32 features of layer `self.layer` on the GPU.  `file_path` may also be a waveform array."""
        x = self.read_audio(file_path, ref_len, channel_id) if isinstance(file_path, str) else np.asarray(file_path, dtype=np.float32)
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(x)).float().cuda()
            if self.normalize:                                     # F.layer_norm(x, x.shape) over the whole waveform (:66-67)
                x = _waveform_layer_norm(x)
            x = x.view(1, -1)
            feat = []
            for start in range(0, x.size(1), self.max_chunk):
                feat.append(self.model.extract_features(x[:, start: start + self.max_chunk], output_layer=self.layer))
        return torch.cat(feat, 0)


def _waveform_layer_norm(x: torch.Tensor) -> torch.Tensor:
    """F.layer_norm over the whole 1-D waveform (no affine, eps 1e-5).  A length-n reduction of a vector that is read
    once per utterance: torch reductions (device plumbing), not a kernel of this library."""
    m = x.mean()
    v = (x - m).pow(2).mean()
    return (x - m) / torch.sqrt(v + 1e-5)


class ApplyKmeans(object):
    """dump_km_label.py:25-50: nearest cluster centre under the squared Euclidean distance."""

    def __init__(self, km_path):
        if isinstance(km_path, str):
            import joblib
            self.km_model = joblib.load(km_path)
            centers = self.km_model.cluster_centers_
        else:
            centers = km_path                                       # an array of centres [n_clusters, D]
        self.C_np = np.asarray(centers).transpose()
        self.Cnorm_np = (self.C_np ** 2).sum(0, keepdims=True)
        self.centers = torch.from_numpy(np.ascontiguousarray(np.asarray(centers), dtype=np.float32)).cuda()   # [K, D] = W of x.C
        self.Cnorm = torch.from_numpy(np.ascontiguousarray(self.Cnorm_np.reshape(-1), dtype=np.float32)).cuda()

    @ops.gated
    def labels(self, x: torch.Tensor, with_margin: bool = False):
        x = x.to(device=self.centers.device, dtype=torch.float32).contiguous()
        dots = torch.empty(x.shape[0], self.centers.shape[0], dtype=torch.float32, device=x.device)
        if x.shape[0]:
            ops.gemm(x, self.centers, dots)
        return ops.kmeans_argmin(x, dots, self.Cnorm, with_margin=with_margin)

    @ops.gated
    def __call__(self, x):
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return self.labels(x).cpu().numpy()


class HubertTokenizer:
    """dgslm_utils.py:19-44."""

    def __init__(self, hubert_path, hubert_layer, km_path, use_cuda=True):
        self.feature_extractor = HubertFeatureReader(hubert_path, hubert_layer, use_cuda=use_cuda)
        self.quantizer = ApplyKmeans(km_path)

    def wav2code(self, path, channel_id=1):
        feat = self.feature_extractor.get_feats(path, channel_id=channel_id)
        code = self.quantizer(feat)
        return ' '.join(map(str, code))

    def wav2codes(self, path):
        return [self.wav2code(path, channel_id=1), self.wav2code(path, channel_id=2)]


def tokenize_directory(process_dir: str, target_dir: str, hubert_path: str, km_path: str, hubert_layer: int = 12) -> int:
    """fairseq-hubert/get_fisher_semantic_tokens.py:30-40: every *.wav -> <name>.hubert_code.npy (array of str codes)."""
    import glob
    import os
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:                                           # independent files: one GPU per rank, no collective
        torch.cuda.set_device(local)
    encoder = HubertTokenizer(hubert_path=hubert_path, hubert_layer=hubert_layer, km_path=km_path)
    files = sorted(glob.glob(os.path.join(process_dir, "*.wav")))[rank::world]
    os.makedirs(target_dir, exist_ok=True)
    for process_file in files:
        codes = encoder.wav2code(process_file, 1).split(" ")
        file_name = process_file.split("/")[-1].split(".")[0]
        np.save(os.path.join(target_dir, file_name + '.hubert_code'), np.array(codes))
    return len(files)
