#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8(f) row N4: the HuBERT layer-12 + k-means(500) prompt tokeniser.

Runs ONLY in the build container: it imports the reference's own HubertModel
(/root/reference/fairseq-hubert/fairseq/models/hubert/hubert.py + models/wav2vec/wav2vec2.py + the fairseq.modules
they use) and ApplyKmeans (examples/hubert/simple_kmeans/dump_km_label.py).  The `fairseq` package __init__ files pull
in hydra / omegaconf / the whole toolkit (absent here), so the package objects are replaced by empty stand-ins whose
__path__ points at the real directories: every model / module source file that matters is the reference's, only
registration / dataclass / distributed plumbing is stubbed.

Weights: full HuBERT-Base geometry (the dGSLM `hubert_fisher.pt` model: 7 conv layers, 12 x 768 post-LN transformer),
values from the build-owned recipe covomix_amd.synthetic.hubert_state_dict (regenerated identically on the GPU box);
only inputs + outputs are committed.

usage: python tests/golden/make_golden_hubert.py
"""
from __future__ import annotations

import dataclasses
import importlib
import os
import sys
import types
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/fairseq-hubert"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def _pkg(name: str, real_dir: Optional[str]):
    m = types.ModuleType(name)
    m.__path__ = [real_dir] if real_dir else []
    sys.modules[name] = m
    if "." in name:
        parent, _, leaf = name.rpartition(".")
        setattr(sys.modules[parent], leaf, m)
    return m


def install_fairseq_stand_ins():
    fs = os.path.join(REF, "fairseq")
    _pkg("fairseq", fs)
    # ---- fairseq.utils: only what the model files touch at import / construction / inference time
    u = _pkg("fairseq.utils", None)
    import torch.nn.functional as F

    def gelu(x):                       # fairseq/modules/gelu.py:gelu
        return F.gelu(x.float()).type_as(x)

    u.get_activation_fn = lambda name: {"gelu": gelu, "relu": F.relu}[name]
    u.get_available_activation_fns = lambda: ["relu", "gelu", "gelu_fast", "gelu_accurate", "tanh", "linear"]
    u.buffered_arange = lambda n: torch.arange(n)
    u.is_xla_tensor = lambda t: False

    def index_put(t, mask, v):
        t[mask] = v
        return t

    u.index_put = index_put
    u.eval_str_dict = lambda x, type=dict: x              # only called with xformers_att_config=None
    u.softmax = lambda x, dim, onnx_trace=False: F.softmax(x, dim=dim, dtype=torch.float32)
    # ---- plumbing packages
    d = _pkg("fairseq.data", None)
    du = _pkg("fairseq.data.data_utils", None)
    du.compute_mask_indices = None
    dd = _pkg("fairseq.data.dictionary", None)
    dd.Dictionary = type("Dictionary", (), {})
    dc = _pkg("fairseq.dataclass", None)

    @dataclasses.dataclass
    class FairseqDataclass:
        _name: Optional[str] = None

    dc.FairseqDataclass = FairseqDataclass
    dc.ChoiceEnum = lambda choices: str
    dist = _pkg("fairseq.distributed", None)
    dist.fsdp_wrap = lambda m, **kw: m
    fsdp = _pkg("fairseq.distributed.fully_sharded_data_parallel", None)
    fsdp.FullyShardedDataParallel = type("FullyShardedDataParallel", (nn.Module,), {})
    om = types.ModuleType("omegaconf")
    om.II = lambda s: None
    sys.modules["omegaconf"] = om
    tasks = _pkg("fairseq.tasks", None)
    hp = _pkg("fairseq.tasks.hubert_pretraining", None)
    hp.HubertPretrainingConfig = type("HubertPretrainingConfig", (), {})
    hp.HubertPretrainingTask = type("HubertPretrainingTask", (), {})
    # ---- fairseq.models: registration plumbing stubbed, model files real
    models = _pkg("fairseq.models", os.path.join(fs, "models"))
    models.BaseFairseqModel = type("BaseFairseqModel", (nn.Module,), {})
    models.register_model = lambda name, dataclass=None: (lambda cls: cls)
    inc = _pkg("fairseq.models.fairseq_incremental_decoder", None)

    class FairseqIncrementalDecoder(nn.Module):          # fairseq/models/fairseq_incremental_decoder.py: base class only
        def __init__(self, dictionary=None):
            super().__init__()

        def init_incremental_state(self):                 # added by @with_incremental_state upstream; unused (no decoding)
            pass

    inc.FairseqIncrementalDecoder = FairseqIncrementalDecoder
    _pkg("fairseq.models.wav2vec", os.path.join(fs, "models", "wav2vec"))
    _pkg("fairseq.models.hubert", os.path.join(fs, "models", "hubert"))
    # ---- fairseq.modules: the real source files, imported one by one (the package __init__ imports the world)
    mods = _pkg("fairseq.modules", os.path.join(fs, "modules"))
    for sub, names in (("layer_norm", ("LayerNorm", "Fp32LayerNorm")), ("fp32_group_norm", ("Fp32GroupNorm",)),
                       ("same_pad", ("SamePad",)), ("transpose_last", ("TransposeLast",)),
                       ("grad_multiply", ("GradMultiply",)), ("fairseq_dropout", ("FairseqDropout",)),
                       ("quant_noise", ("quant_noise",)), ("multihead_attention", ("MultiheadAttention",))):
        m = importlib.import_module(f"fairseq.modules.{sub}")
        for n in names:
            setattr(mods, n, getattr(m, n))
    mods.GumbelVectorQuantizer = type("GumbelVectorQuantizer", (nn.Module,), {})
    mods.RelPositionalEncoding = type("RelPositionalEncoding", (nn.Module,), {})
    ca = _pkg("fairseq.modules.checkpoint_activations", None)
    ca.checkpoint_wrapper = lambda m, **kw: m
    cl = _pkg("fairseq.modules.conformer_layer", None)
    cl.ConformerWav2Vec2EncoderLayer = type("ConformerWav2Vec2EncoderLayer", (nn.Module,), {})
    tse = _pkg("fairseq.modules.transformer_sentence_encoder", None)
    tse.init_bert_params = lambda module: None            # every weight is overwritten by the recipe below


def build_reference_model():
    install_fairseq_stand_ins()
    hub = importlib.import_module("fairseq.models.hubert.hubert")
    cfg = hub.HubertConfig()
    cfg.label_rate = 50.0
    cfg.dropout = cfg.attention_dropout = 0.0
    cfg.required_seq_len_multiple = getattr(cfg, "required_seq_len_multiple", 2)
    task_cfg = types.SimpleNamespace(sample_rate=16000, normalize=False)
    model = hub.HubertModel(cfg, task_cfg, [None]).eval()
    return model, cfg, task_cfg


def reference_apply_kmeans(centers: np.ndarray):
    """The reference's ApplyKmeans (dump_km_label.py:25-50) on a joblib-dumped object with `cluster_centers_`."""
    import joblib
    import tempfile
    sys.path.insert(0, REF)
    src = open(os.path.join(REF, "examples/hubert/simple_kmeans/dump_km_label.py")).read()
    ns: dict = {}
    head = src.split("def get_feat_iterator")[0]
    exec(compile(head, "dump_km_label.py", "exec"), ns)
    km = types.SimpleNamespace(cluster_centers_=centers)
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        joblib.dump(km, f.name)
        return ns["ApplyKmeans"](f.name)


def main():
    sys.path.insert(0, REPO)
    import importlib.util
    spec = importlib.util.spec_from_file_location("cvx_synthetic", os.path.join(REPO, "neurips2024-covomix_amd", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)

    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, cfg, task_cfg = build_reference_model()
    names = [n for n, _ in model.state_dict().items()]
    sd = syn.hubert_state_dict(seed=0)
    missing = [n for n in names if n not in sd]
    assert not missing, missing
    model.load_state_dict({n: torch.from_numpy(sd[n]) for n in names}, strict=True)
    centers = syn.hubert_kmeans_centers(seed=0)
    km = reference_apply_kmeans(centers)

    out = {}
    rng = np.random.RandomState(4242)
    for tag, n in (("a", 8000), ("b", 5215), ("c", 16400)):
        # speech-like: a few decaying sinusoids + noise + a DC offset, at int16-normalised scale
        t = np.arange(n) / 16000.0
        wav = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in
                  zip(rng.uniform(0.02, 0.2, 5), rng.uniform(80, 3800, 5), rng.uniform(0, 6.28, 5)))
        wav = (wav * np.exp(-1.5 * t) + 0.02 * rng.standard_normal(n) + 0.01).astype(np.float32)
        x = torch.from_numpy(wav).view(1, -1)
        with torch.no_grad():
            conv = model.forward_features(x)                                   # [1, 512, T]
            feats = {}
            for layer in (1, 6, 12):
                f, _ = model.extract_features(source=x, padding_mask=None, mask=False, output_layer=layer)
                feats[layer] = f.squeeze(0).numpy()
            codes = km(torch.from_numpy(feats[12]))
            f12 = torch.from_numpy(feats[12])
            dist = f12.pow(2).sum(1, keepdim=True) - 2 * torch.matmul(f12, km.C) + km.Cnorm
            top2 = torch.topk(dist, 2, dim=1, largest=False).values
        out[f"{tag}_wav"] = wav
        out[f"{tag}_conv"] = conv.squeeze(0).transpose(0, 1).contiguous().numpy()   # [T, 512]
        for layer in (1, 6, 12):
            out[f"{tag}_feat{layer}"] = feats[layer]
        out[f"{tag}_codes"] = np.asarray(codes, dtype=np.int64)
        out[f"{tag}_margin"] = (top2[:, 1] - top2[:, 0]).numpy()
        print(tag, n, "frames", feats[12].shape, "codes", codes[:12], "min margin", float(out[f"{tag}_margin"].min()),
              "feat12 rms", float(np.sqrt((feats[12] ** 2).mean())), "conv rms", float(np.sqrt((out[f'{tag}_conv'] ** 2).mean())))
    # second k-means fixture (round-3 review: with N(0, 1) centres nearly every frame gets the same label): centres drawn from
    # the fixtures' OWN layer-12 frames (all three waveforms pooled, jittered: synthetic.hubert_kmeans_centers_near), labelled by
    # the reference's ApplyKmeans - dozens of distinct labels, none dominant
    pool = np.concatenate([out[f"{t_}_feat12"] for t_ in ("a", "b", "c")], 0)
    km2 = reference_apply_kmeans(syn.hubert_kmeans_centers_near(pool, seed=0))
    for tag in ("a", "b", "c"):
        f12 = torch.from_numpy(out[f"{tag}_feat12"])
        with torch.no_grad():
            codes2 = km2(f12)
            dist = f12.pow(2).sum(1, keepdim=True) - 2 * torch.matmul(f12, km2.C) + km2.Cnorm
            top2 = torch.topk(dist, 2, dim=1, largest=False).values
        out[f"{tag}_codes_near"] = np.asarray(codes2, dtype=np.int64)
        out[f"{tag}_margin_near"] = (top2[:, 1] - top2[:, 0]).numpy()
    allc = np.concatenate([out[f"{t_}_codes_near"] for t_ in ("a", "b", "c")])
    cnt = np.bincount(allc, minlength=500)
    print("near-centre labels:", len(np.unique(allc)), "distinct of", allc.size, "frames; most frequent label holds",
          f"{cnt.max() / allc.size:.1%}; min margin", float(min(out[f"{t_}_margin_near"].min() for t_ in ("a", "b", "c"))))
    assert len(np.unique(allc)) >= 50 and cnt.max() <= 0.2 * allc.size
    # normalize=True path of get_feats (hubert_feature_reader.py:66-67): F.layer_norm over the whole waveform
    x = torch.from_numpy(out["a_wav"])
    with torch.no_grad():
        xn = torch.nn.functional.layer_norm(x, x.shape).view(1, -1)
        f, _ = model.extract_features(source=xn, padding_mask=None, mask=False, output_layer=12)
    out["a_feat12_normalized"] = f.squeeze(0).numpy()
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "hubert_base.npz"), **out)
    print("wrote", os.path.join(HERE, "hubert_base.npz"), os.path.getsize(os.path.join(HERE, "hubert_base.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
