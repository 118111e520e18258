"""`CoVoMixModel` facade - the Python-level drop-in boundary.

Mirrors the inference surface of the reference LightningModule
(covomix/conditional_model.py:38-321) without Lightning / torch_ema / diffusers:

    model = CoVoMixModel.load_from_checkpoint(ckpt, base_dir='', batch_size=16, num_workers=0)
    model.eval()                  # swaps in the EMA shadow weights (conditional_model.py:203-217)
    model = model.to(device)
    mel = model.synthesis_sample(phoneme_ids=..., cond=..., mask=..., cond_scale=0.7)   # :295-302
    tok = t2s_model.synthesis_sample_text2semantic(grapheme_token_ids)                  # :313-321 (text2semantic ckpt)

Checkpoint layout accepted (what Lightning's ModelCheckpoint + on_save_checkpoint write, :150,:200-201):
    ckpt['state_dict']        keys 'cfm_wrapper.CoVoMix.<param>' (acoustic) or 'cfm_wrapper.model.<param>' (text2semantic:
                              TextToSemanticWrapper.model, text2semantic.py:1205-1213; tied / shared tensors appear
                              under several names - token_emb.speech.*, to_logits.*, per-layer rotary_emb.freqs)
    ckpt['hyper_parameters']  dict (may reference classes such as covomix.data_module.SpecsDataModule
                              that do not exist here -> unpickled as inert stubs)
    ckpt['ema']               torch_ema state: {'decay','num_updates','shadow_params': [...], 'collected_params'}
                              with shadow_params in nn.Module.parameters() order
If 'ema' is missing the plain state_dict is used, with a warning (:192-198).
"""
from __future__ import annotations

import os
import pickle
import types
import warnings
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import ops
from .acoustic import FlowMatchingSampler, VectorField

_PREFIX = "cfm_wrapper.CoVoMix."
_PREFIX_T2S = "cfm_wrapper.model."


class _StubUnpickler(pickle.Unpickler):
    """Unpickler that turns references to classes/functions of modules that are not installed here
    (pytorch_lightning callbacks, covomix.data_module.SpecsDataModule, ...) into inert placeholders."""

    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return type(name, (), {"__module__": module, "__init__": lambda self, *a, **k: None,
                                   "__setstate__": lambda self, s: None})


_stub_pickle = types.SimpleNamespace(Unpickler=_StubUnpickler, load=lambda f, **kw: _StubUnpickler(f, **kw).load(),
                                     __name__="covomix_amd_stub_pickle")


def _torch_load(path: str):
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_stub_pickle)


def parameter_order(sd_keys) -> list:
    """Names in nn.Module.parameters() order for the reference CoVoMix (buffers such as
    rotary_emb.inv_freq are not parameters)."""
    return [k for k in sd_keys if not k.endswith("rotary_emb.inv_freq")]


def t2s_parameter_order(sd_keys) -> list:
    """nn.Module.parameters() order of the reference TextToSemantic: state_dict order minus the aliases of tied /
    shared parameters (to_logits.* and token_emb.speech.* are the embedding tables, text2semantic.py:524-552; one
    RotaryEmbedding per transformer is shared by its layers, :291-299)."""
    return [k for k in sd_keys if not (k.startswith("token_emb.speech.") or k.startswith("to_logits.")
                                       or (k.endswith("rotary_emb.freqs") and ".layers.0.0." not in k))]


class CoVoMixModel:
    def __init__(self, state_dict: Dict[str, torch.Tensor], hparams: Optional[dict] = None,
                 ema_shadow: Optional[list] = None, nfe: int = 32, ode_method: str = "midpoint",
                 precision: Optional[str] = None):
        """state_dict: un-prefixed CoVoMix parameter names (acoustic.py:326-406)."""
        self.hparams = dict(hparams or {})
        self.is_text2semantic = "token_emb.text.weight" in state_dict
        if bool(self.hparams.get("text2semantic", self.is_text2semantic)) != self.is_text2semantic:
            raise ValueError("hyper_parameters['text2semantic'] disagrees with the parameter names of the state_dict")
        self._raw = OrderedDict((k, v.detach().cpu()) for k, v in state_dict.items())
        self._ema = None
        if ema_shadow is not None:
            names = (t2s_parameter_order if self.is_text2semantic else parameter_order)(self._raw.keys())
            if len(names) != len(ema_shadow):
                raise ValueError(f"EMA has {len(ema_shadow)} shadow params, model has {len(names)} parameters")
            self._ema = OrderedDict(self._raw)
            for n, t in zip(names, ema_shadow):
                if tuple(t.shape) != tuple(self._raw[n].shape):
                    raise ValueError(f"EMA shadow param shape mismatch at {n}")
                self._ema[n] = t.detach().cpu()
        self._error_loading_ema = ema_shadow is None
        self._use_ema = False
        self.device = torch.device("cpu")
        self.nfe, self.ode_method = nfe, ode_method
        self.precision = precision or os.environ.get("CVX_PRECISION", "f16x3")
        self._field: Optional[VectorField] = None
        self._field_fp32: Optional[VectorField] = None      # built only if a split-precision call ever saturates
        self._t2s = None

    # ---- construction ---------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, base_dir='', batch_size=16, num_workers=0, **kwargs):
        assert os.path.isfile(checkpoint_path), checkpoint_path      # mirrors monologue_generation.py:46
        ckpt = _torch_load(checkpoint_path)
        sd = OrderedDict((k[len(_PREFIX):], v) for k, v in ckpt["state_dict"].items() if k.startswith(_PREFIX))
        if not sd:                                                   # text2semantic checkpoint (CoSingle / CoMix)
            sd = OrderedDict((k[len(_PREFIX_T2S):], v) for k, v in ckpt["state_dict"].items() if k.startswith(_PREFIX_T2S))
        if not sd:
            raise KeyError(f"no '{_PREFIX}*' or '{_PREFIX_T2S}*' entries in checkpoint state_dict")
        ema = ckpt.get("ema")
        shadow = None
        if ema is not None:
            shadow = ema["shadow_params"]
        else:
            warnings.warn("EMA state_dict not found in checkpoint!")
        hp = ckpt.get("hyper_parameters", {})
        hp = {k: v for k, v in dict(hp).items() if isinstance(v, (int, float, str, bool, type(None)))}
        return cls(sd, hparams=hp, ema_shadow=shadow, **kwargs)

    @classmethod
    def from_state_dict(cls, state_dict, **kwargs):
        return cls(state_dict, **kwargs)

    # ---- nn.Module-like surface -------------------------------------------------------------
    def train(self, mode: bool = True, no_ema: bool = False):
        use = (not mode) and (not no_ema) and (not self._error_loading_ema)
        if use != self._use_ema:
            self._use_ema = use
            self._field = None
            self._field_fp32 = None
            self._t2s = None
        return self

    def eval(self, no_ema: bool = False):
        return self.train(False, no_ema=no_ema)

    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self.device = device
            self._field = None
            self._field_fp32 = None
            self._t2s = None
        return self

    def active_state_dict(self) -> Dict[str, torch.Tensor]:
        return self._ema if (self._use_ema and self._ema is not None) else self._raw

    def _get_field(self) -> VectorField:
        if self.is_text2semantic:
            raise TypeError("this checkpoint is a text2semantic model: use synthesis_sample_text2semantic")
        if self._field is None:
            if self.device.type != "cuda":
                from ._lib import CovomixHipError
                raise CovomixHipError("CoVoMixModel must be on a GPU (`.to('cuda')`): covomix_amd has no CPU path")
            self._field = VectorField(self.active_state_dict(), self.device, precision=self.precision)
        return self._field

    # ---- sampling -----------------------------------------------------------------------------
    @ops.gated
    @torch.no_grad()
    def synthesis_sample(self, phoneme_ids, cond, mask, cond_scale, y0=None):
        """reference conditional_model.py:295-302 -> ConditionalFlowMatcherWrapper.sample.
        `mask` is accepted and unused, exactly as in the reference (acoustic.py:597-688).
        `y0` (optional) fixes the initial noise; parity is defined given y0.
        Extension: LISTS of per-utterance tensors (phoneme_ids[i] [T_i(, streams)], cond[i] [T_i, C], y0[i] [T_i, dim_out],
        lengths may differ) run as one packed ragged batch and return a list; each result equals that utterance's B = 1
        call up to fp32 summation order (the reference loops over utterances, monologue_generation.py:259-304)."""
        from . import ops
        field = self._get_field()
        ragged = isinstance(cond, (list, tuple))     # extension: utterances of different length in one packed launch

        def run(f):
            sampler = FlowMatchingSampler(f, nfe=self.nfe, method=self.ode_method)
            if ragged:
                return sampler.sample_ragged(phoneme_ids=list(phoneme_ids), cond=list(cond), cond_scale=cond_scale,
                                             y0=None if y0 is None else list(y0))
            return sampler.sample(phoneme_ids=phoneme_ids, cond=cond, mask=mask, cond_scale=cond_scale, y0=y0)
        # Split-precision activations live in a window of 2^12 around the RMS the gain model predicts (acoustic.py,
        # _activation_scales).  A checkpoint with an outlier row / channel can leave it: the kernels then clamp and raise the
        # device's sticky saturation flag (include/covomix_hip.h).  One flag read per call (the only host synchronisation of
        # the solve); a flagged call is re-run on the exact-fp32 kernels (or raises: CVX_ON_SATURATION=raise) - never
        # returned as is.
        checked = field.precision != "fp32" and ops.saturation_checked() and (not ragged or len(cond) > 0)
        if ragged and y0 is None and checked:       # a re-run must see the same noise
            y0 = [torch.randn(c.shape[0], field.d["dim_out"], device=self.device) for c in cond]
        elif not ragged and y0 is None and checked:
            y0 = torch.randn(cond.shape[0], cond.shape[1], field.d["dim_out"], device=self.device)
        if checked:
            ops.saturation_reset()
        out = run(field)
        if checked and ops.saturation_query():
            out = run(self._saturated(field.precision))
        if ragged:
            return [o.to(c.device) if c.device != o.device else o for o, c in zip(out, cond)]
        return out.to(cond.device) if cond.device != out.device else out

    def _saturated(self, precision: str) -> VectorField:
        """Called when a split-precision solve raised the saturation flag: warn (or raise) and hand back the exact-fp32
        field (built on first use: one more device copy of the weights)."""
        msg = (f"covomix_amd: split-precision ('{precision}') activations left the fp16 window (saturating store): this "
               "checkpoint has outlier weights the gain model does not predict")
        if os.environ.get("CVX_ON_SATURATION", "fp32") == "raise":
            from ._lib import CovomixHipError
            raise CovomixHipError(msg + " (CVX_ON_SATURATION=raise)")
        warnings.warn(msg + "; re-running this call on the exact-fp32 kernels (construct the model with precision='fp32' to avoid the double work)")
        if self._field_fp32 is None:
            self._field_fp32 = VectorField(self.active_state_dict(), self.device, precision="fp32")
        return self._field_fp32

    def _get_t2s(self):
        """The device-resident text2semantic decoder (built on first use)."""
        if not self.is_text2semantic:
            raise TypeError("this checkpoint is an acoustic model: use synthesis_sample")
        if self._t2s is None:
            if self.device.type != "cuda":
                from ._lib import CovomixHipError
                raise CovomixHipError("CoVoMixModel must be on a GPU (`.to('cuda')`): covomix_amd has no CPU path")
            from .t2s import TextToSemanticDecoder
            self._t2s = TextToSemanticDecoder(self.active_state_dict(), self.device)
        return self._t2s

    @ops.gated
    @torch.no_grad()
    def synthesis_sample_text2semantic(self, grapheme_token_ids, temprature=1.0, cond_scale=1.0, beam_search_decode=False,
                                       prompt_mel=None, uniforms=None, generator=None, max_length=None, slots=64):
        """reference conditional_model.py:313-321 -> TextToSemanticWrapper.sample (text2semantic.py:1237-1251): the
        sampled semantic tokens as one flat int64 tensor (two-output models: stream 1 then stream 2) on the input's
        device.  (`temprature` is the reference's spelling.)  `uniforms` / `generator` (optional) fix the U(0,1) draws
        behind the Gumbel noise; parity is defined given them.  A LIST of id tensors (any number) is decoded through `slots`
        continuously refilled decode slots (t2s.generate_many) and returns a list (the reference decodes utterances one by
        one; the tokens are the same)."""
        if not self.is_text2semantic:
            raise TypeError("this checkpoint is an acoustic model: use synthesis_sample")
        assert cond_scale >= 1., "cond_scale >= 1 (text2semantic.py:683)"
        # text2semantic.py:684: guidance needs a model trained with condition dropping - the checkpoint's `cond_drop_prob`
        # hyper-parameter (conditional_model.py:52, :78, :114); checkpoints without the key were built with the default 0
        assert not (cond_scale > 1 and float(self.hparams.get("cond_drop_prob", 0.0)) == 0.0), \
            ("you need to train with conditional drop probability greater than 0 to use classifier free guidance at inference "
             "(text2semantic.py:684): this checkpoint's hyper_parameters['cond_drop_prob'] is 0 or absent")
        if beam_search_decode:
            raise NotImplementedError("beam search decoding is not built (the generation scripts sample)")
        self._get_t2s()
        ids = grapheme_token_ids
        if isinstance(ids, (list, tuple)):          # extension: several utterances decoded together (bit-identical tokens)
            ids = list(ids)
            if float(cond_scale) > 1.0:             # guidance: slot pairs in lock step, up to 32 utterances per pass
                res = []
                for w in range(0, len(ids), 32):
                    res += self._t2s.generate_batch(ids[w:w + 32], None if uniforms is None else list(uniforms[w:w + 32]), max_length,
                                                    float(temprature), generator, cond_scale=float(cond_scale))
            else:                                   # any number of utterances through 64 continuously refilled decode slots
                res = self._t2s.generate_many(ids, uniforms, max_length, float(temprature), generator, slots=slots)
            return [r[0].to(i.device) for r, i in zip(res, ids)]
        out = self._t2s.generate(ids, uniforms=uniforms, max_length=max_length, temperature=float(temprature), generator=generator,
                                 cond_scale=float(cond_scale))
        return out.to(ids.device) if ids.device != out.device else out
