"""HiFi-GAN generator on the MI355X kernels - drop-in for the reference `Generator`.

Mirrors covomix/vocoder/models.py:75-125 (Generator), :11-48 (ResBlock1) and
covomix/vocoder/env.py:5-8 (AttrDict) of the reference:

    h = AttrDict(json.load(open('vocoder_config.json')))
    generator = Generator(h).to(device)
    generator.load_state_dict(torch.load(ckpt)['generator'])      # weight_g / weight_v / bias
    generator.eval(); generator.remove_weight_norm()
    wav = generator(mel)            # mel [B,80,T] -> [B,1,160T+32];  [80,T] -> [1,160T+32]

Only resblock == '1' (what config_covomix.json selects) is implemented.  Each conv launch
fuses the preceding leaky_relu, bias, the ResBlock residual add and the running
`xs += resblock(x)` / `xs / num_kernels` of Generator.forward (models.py:104-110).

precision (constructor argument or env CVX_VOCODER_PRECISION):
  'f16x3' (default)  the 72 ResBlock convolutions (97 % of the FLOPs) run on the fp16 matrix pipe with split-precision
                     operands (cvx_hifigan_conv1d_f16x3, channels-last activations with zero halos); conv_pre, the four
                     ConvTranspose1d upsamplers and conv_post stay on the fp32 kernels, with a layout converter on
                     either side of every ResBlock stage;
  'fp32'             everything on v_mfma_f32_32x32x2_f32 (cvx_hifigan_conv1d_f32).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import ops

LRELU_SLOPE = 0.1   # models.py:8


class AttrDict(dict):
    """dict with attribute access (reference env.py:5-8)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


GROUP_STAGE = True      # wide stages: the ResBlocks of a stage share launches (tests flip it to compare against block after block)


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    return int((kernel_size * dilation - dilation) / 2)     # vocoder/utils.py:34-35


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """weight = weight_v * weight_g / ||weight_v|| with the norm over every dim but 0
    (torch._weight_norm, dim=0 - per INPUT channel for ConvTranspose1d).  Load-time plumbing."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            wv = sd[base + ".weight_v"].float()
            nrm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (wv.ndim - 1)))
            out[base + ".weight"] = wv * (v.float() / nrm)
        elif not k.endswith(".weight_v"):
            out[k] = v
    return out


class _Conv:
    __slots__ = ("wp", "bias", "cout", "cin", "k", "dil", "pad", "up", "lout_fn", "w16", "bias16", "wp_poly", "padding", "w16t")


class Generator:
    def __init__(self, h, precision: Optional[str] = None):
        if str(h["resblock"]) != "1":
            raise NotImplementedError("only ResBlock1 (resblock == '1') is supported, as in config_covomix.json")
        self.precision = precision or os.environ.get("CVX_VOCODER_PRECISION", "f16x3")
        if self.precision not in ("f16x3", "fp32"):
            raise ValueError(f"vocoder precision must be 'f16x3' or 'fp32', got {self.precision!r}")
        self._cl: Dict[tuple, dict] = {}
        self.act_scales = True               # measured per-stage power-of-two pre-scales of the split pairs
        # channels-last pipeline (round 3): the upsamplers on the split pipe too, no layout converters between the stages
        self.cl_pipeline = True
        self.h = h
        self.num_kernels = len(h["resblock_kernel_sizes"])
        self.num_upsamples = len(h["upsample_rates"])
        self.device = torch.device("cpu")
        self._sd: Optional[Dict[str, torch.Tensor]] = None
        self._has_weight_norm = False
        self._packed = None
        self._fp32_twin: Optional["Generator"] = None       # built only if a split-precision call ever saturates

    # ---- nn.Module-like surface used by the reference scripts -------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        self._sd = {k: v.detach().to("cpu") for k, v in sd.items()}
        self._has_weight_norm = any(k.endswith(".weight_g") for k in self._sd)
        self._packed = None
        self._fp32_twin = None
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self.device = device
            self._packed = None
            self._fp32_twin = None
        return self

    def remove_weight_norm(self):
        print('Removing weight norm...')
        if self._has_weight_norm:
            self._sd = fold_weight_norm(self._sd)
            self._has_weight_norm = False
            self._packed = None
            self._fp32_twin = None
        return self

    # ---- weight packing -----------------------------------------------------------------
    def _conv(self, name: str, dil: int = 1, pad: int = 0, transposed: bool = False, up: int = 1) -> _Conv:
        sd = self._sd
        w = sd[name + ".weight"].float()
        c = _Conv()
        if transposed:
            c.cin, c.cout, c.k = w.shape
            c.pad = c.k - 1 - pad            # stride-1 conv over the zero-stuffed input, flipped kernel
        else:
            c.cout, c.cin, c.k = w.shape
            c.pad = pad
        c.dil, c.up = dil, up
        c.wp = ops.hifigan_pack_weight(w, transposed).to(self.device)
        c.wp_poly, c.padding = None, pad
        if transposed and up > 1:                 # polyphase form: 1/stride of the zero-stuffed form's matrix work
            c.wp_poly = ops.hifigan_pack_conv_transpose1d(w, up, pad).to(self.device)
        c.bias = sd[name + ".bias"].float().to(self.device).contiguous()
        c.w16 = c.bias16 = c.w16t = None
        if self.precision == "f16x3" and transposed and 1 <= c.k - 2 * pad <= 2 * up and c.cout <= 256 and up <= 8:
            cp_in = (32 if c.cin <= 32 else 64 if c.cin <= 64 else 128 if c.cin <= 128 else 256) if c.cin <= 256 else (c.cin + 31) // 32 * 32
            c.w16t = ops.hifigan_pack_conv_transpose1d_f16x3(w.to(self.device), c.bias, up, pad, cp_in)   # (input = a stage's Np-wide buffers)
        if (self.precision == "f16x3" and not transposed and up == 1 and name.startswith("resblocks.") and c.cout <= 256
                and (c.k - 1) * dil <= 50 and (c.k - 1) * dil % 2 == 0):
            c.w16 = ops.hifigan_pack_weight_f16x3(w.to(self.device))
            c.bias16 = torch.zeros(c.w16[3], dtype=torch.float32, device=self.device)
            c.bias16[: c.cout] = c.bias
        return c

    def pack(self) -> "Generator":
        """Fold weight norm and lay the weights out for the kernels NOW (otherwise the first forward does it - with host-to-device
        copies that wait behind whatever the stream is still running, e.g. the solve whose mel this call is about to vocode)."""
        if self._packed is None:
            self._pack()
        return self

    def _pack(self):
        if self._sd is None:
            raise RuntimeError("Generator: load_state_dict() has not been called")
        if self._has_weight_norm:
            # the reference can run with weight norm still attached (same function); fold on the fly
            self._sd = fold_weight_norm(self._sd)
            self._has_weight_norm = False
        if self.device.type != "cuda":
            raise ops._lib.CovomixHipError("Generator must be moved to a GPU (`.to('cuda')`): covomix_amd has no CPU path")
        h = self.h
        pk = dict(pre=self._conv("conv_pre", pad=3), ups=[], res=[])
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            pk["ups"].append(self._conv(f"ups.{i}", pad=(k - u) // 2, transposed=True, up=u))
            blocks = []
            for j, (rk, dils) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
                p = f"resblocks.{i * self.num_kernels + j}"
                blocks.append([(self._conv(f"{p}.convs1.{m}", dil=d, pad=get_padding(rk, d)),
                                self._conv(f"{p}.convs2.{m}", dil=1, pad=get_padding(rk, 1)))
                               for m, d in enumerate(dils)])
            pk["res"].append(blocks)
        pk["post_w"] = self._sd["conv_post.weight"].float().reshape(-1, 7).contiguous().to(self.device)
        pk["post_b"] = float(self._sd["conv_post.bias"].float().reshape(-1)[0])
        self._packed = pk

    # ---- forward ------------------------------------------------------------------------
    def _run(self, c: _Conv, x: torch.Tensor, out: Optional[torch.Tensor] = None, *, in_slope=1.0, res=None,
             accum=None, out_scale=1.0, items=None) -> torch.Tensor:
        B, _, lin = x.shape
        lout = (lin - 1) * c.up + 1 + 2 * c.pad - (c.k - 1) * c.dil
        if out is None:
            out = torch.empty(B, c.cout, lout, dtype=torch.float32, device=x.device)
        return ops.hifigan_conv1d(x, c.wp, c.bias, out, cout=c.cout, ksize=c.k, dil=c.dil, pad=c.pad, up=c.up,
                                  in_slope=in_slope, res=res, accum=accum, out_scale=out_scale, items=items)

    @staticmethod
    def _affine(c: _Conv, mul: int, add: int) -> tuple:
        """Output length of convolution c as a function of the item's mel frames T, given its input length mul * T + add."""
        # lout = (lin - 1) * up + 1 + 2 * pad - (k - 1) * dil
        return mul * c.up, (add - 1) * c.up + 1 + 2 * c.pad - (c.k - 1) * c.dil

    def output_length(self, frames: int) -> int:
        """Samples the generator returns for a mel of `frames` frames."""
        if self._packed is None:
            self._pack()
        mul, add = 1, 0
        for c in [self._packed["pre"]] + self._packed["ups"]:
            mul, add = self._affine(c, mul, add)
        return mul * frames + add

    @ops.gated
    @torch.no_grad()
    def __call__(self, mel: torch.Tensor, lengths=None) -> torch.Tensor:
        """mel [B, 80, T] -> [B, 1, L]  ([80, T] -> [1, L]), models.py:100-116.
        lengths (extension, ragged batch): item b only has lengths[b] <= T valid frames (the rest of its mel must be zero
        padding); its first output_length(lengths[b]) samples are then exactly what a B = 1 call on its own frames returns
        (every kernel writes zeros behind a shorter item's end - the zero padding the B = 1 run sees there), the samples
        behind them are unspecified.  `ragged()` wraps the padding and slicing.
        The split-precision stages keep their activations within 2^6 of the measured max|x| of the stage input; when a
        ResBlock intermediate outgrows that, the kernels clamp and raise the device's sticky saturation flag
        (include/covomix_hip.h).  One flag read per call; a flagged call is re-run on the all-fp32 kernels (or raises:
        CVX_ON_SATURATION=raise) - never returned as is."""
        checked = self.precision != "fp32" and ops.saturation_checked() and self.device.type == "cuda"
        if not checked:
            return self._forward(mel, lengths)
        with torch.cuda.device(self.device):
            ops.saturation_reset()
            y = self._forward(mel, lengths)
            if not ops.saturation_query():
                return y
        msg = ("covomix_amd: the split-precision vocoder stages saturated (an intermediate left the fp16 window around the "
               "measured stage input)")
        if os.environ.get("CVX_ON_SATURATION", "fp32") == "raise":
            raise ops._lib.CovomixHipError(msg + " (CVX_ON_SATURATION=raise)")
        import warnings
        warnings.warn(msg + "; re-running this call on the all-fp32 kernels")
        if self._fp32_twin is None:
            twin = Generator(self.h, precision="fp32").to(self.device)
            twin._sd, twin._has_weight_norm = self._sd, self._has_weight_norm
            self._fp32_twin = twin
        return self._fp32_twin._forward(mel, lengths)

    @torch.no_grad()
    def ragged(self, mels) -> list:
        """mels: list of [80, T_b] tensors of different length -> list of [1, L_b] waveforms from ONE batched call (each
        equal to generator(mels[b]) up to fp32 summation order)."""
        if not mels:
            return []
        T = [int(m.shape[-1]) for m in mels]
        pad = torch.zeros(len(mels), mels[0].shape[0], max(T), dtype=torch.float32, device=self.device)
        for b, m in enumerate(mels):
            pad[b, :, : T[b]] = m
        y = self(pad, lengths=T) if len(set(T)) > 1 else self(pad)
        return [y[b, :, : self.output_length(T[b])] for b in range(len(mels))]

    def _forward(self, mel: torch.Tensor, lengths=None) -> torch.Tensor:
        if self._packed is None:
            self._pack()
        pk = self._packed
        unbatched = mel.ndim == 2
        x = mel.to(device=self.device, dtype=torch.float32)
        if unbatched:
            x = x.unsqueeze(0)
        x = x.contiguous()
        lens, mul, add = None, 1, 0               # ragged batch: item b is valid on mul * lens[b] + add positions of x
        if lengths is not None:
            if len(lengths) != x.shape[0] or min(lengths) < 1 or max(lengths) > x.shape[2]:
                raise ValueError(f"lengths must hold one frame count in [1, {x.shape[2]}] per batch item")
            lens = ops.h2d(torch.tensor([int(t) for t in lengths], dtype=torch.int32), self.device)
        it = lambda: None if lens is None else (lens, mul, add)
        mul, add = self._affine(pk["pre"], mul, add)
        x = self._run(pk["pre"], x, items=it())
        if (self.cl_pipeline and self.precision == "f16x3" and all(u.w16t is not None for u in pk["ups"]) and pk["ups"][-1].cout <= 64
                and all(c.w16 is not None and len(block) == 3 for blocks in pk["res"] for block in blocks for pair in block for c in pair)):
            y = self._forward_channels_last(pk, x, lens, mul, add)
            return y.squeeze(0) if unbatched else y
        for i in range(self.num_upsamples):
            up = pk["ups"][i]
            mul, add = self._affine(up, mul, add)
            split = all(c.w16 is not None for block in pk["res"][i] for pair in block for c in pair)
            if up.wp_poly is not None:                                      # leaky_relu + ConvTranspose1d (polyphase)
                B, _, lin = x.shape
                lout = (lin - 1) * up.up + 1 + 2 * up.pad - (up.k - 1)
                buf = self._cl_buffers(B, up.cout, lout) if split and self.act_scales else None
                y = torch.empty(B, up.cout, lout, dtype=torch.float32, device=x.device)
                x = ops.hifigan_conv_transpose1d(x, up.wp_poly, up.bias, y, cout=up.cout, ksize=up.k, stride=up.up, padding=up.padding,
                                                 in_slope=LRELU_SLOPE, amax_bits=buf["zs_scratch"] if buf is not None else None, items=it())
                if split:
                    x = self._resblocks_f16x3(pk["res"][i], x, measured=buf is not None, items=it())
                    continue
            else:
                x = self._run(up, x, in_slope=LRELU_SLOPE, items=it())
                if split:
                    x = self._resblocks_f16x3(pk["res"][i], x, items=it())
                    continue
            t = torch.empty_like(x)
            r = torch.empty_like(x)
            xs = torch.empty_like(x)
            nblk = len(pk["res"][i])
            for j, block in enumerate(pk["res"][i]):
                cur = x
                for m, (c1, c2) in enumerate(block):
                    self._run(c1, cur, t, in_slope=LRELU_SLOPE, items=it())
                    if m + 1 < len(block):
                        self._run(c2, t, r, in_slope=LRELU_SLOPE, res=cur, items=it())
                        cur = r
                    else:                                                    # last pair: fold into xs
                        self._run(c2, t, xs, in_slope=LRELU_SLOPE, res=cur, accum=xs if j > 0 else None,
                                  out_scale=(1.0 / self.num_kernels) if j == nblk - 1 else 1.0, items=it())
            x = xs
        B, _, L = x.shape
        y = torch.empty(B, 1, L, dtype=torch.float32, device=x.device)
        ops.hifigan_post(x, pk["post_w"], pk["post_b"], y, slope=0.01)       # F.leaky_relu default slope (:112)
        return y.squeeze(0) if unbatched else y

    forward = __call__

    # ---- ResBlock stage on the split-precision kernel ---------------------------------------
    def _cl_buffers(self, B: int, C: int, L: int) -> dict:
        """Channels-last buffers of one stage, zero-initialised ONCE (the kernels only ever write valid rows, and
        write zeros into the padded channels), cached per shape."""
        key = (B, C, L)
        buf = self._cl.get(key)
        if buf is None:
            Lp, Cp = ops.hifigan_cl_rows(L), (C + 31) // 32 * 32
            np_ = 32 if C <= 32 else 64 if C <= 64 else 128 if C <= 128 else 256
            assert np_ == Cp or Cp < np_          # Cp (input padding, x32) <= Np (output tile); use Np for both
            Cp = np_
            f32 = lambda: torch.zeros(B, Lp, Cp, dtype=torch.float32, device=self.device)
            f16 = lambda: (torch.zeros(B, Lp, Cp, dtype=torch.float16, device=self.device),
                           torch.zeros(B, Lp, Cp, dtype=torch.float16, device=self.device))
            buf = dict(x0=f32(), r0=f32(), r1=f32(), xs=f32())
            buf["zs"] = torch.ones(1, dtype=torch.float32, device=self.device)
            buf["zs_scratch"] = torch.zeros(1, dtype=torch.int32, device=self.device)
            if Cp > 64:                           # (the narrow stages run one kernel per conv pair: no split pairs in HBM)
                buf.update(z0=f16(), t=f16(), rz0=f16(), rz1=f16())
                # the ResBlocks of a wide stage share launches (cvx_hifigan_resblock_stage_f16x3): every block needs its OWN scratch
                buf["blk"] = [dict(t=buf["t"], rz0=buf["rz0"], rz1=buf["rz1"], r0=buf["r0"], r1=buf["r1"])] + \
                             [dict(t=f16(), rz0=f16(), rz1=f16(), r0=f32(), r1=f32()) for _ in range(self.num_kernels - 1)]
            if len(self._cl) >= 12:
                self._cl.clear()
            self._cl[key] = buf
        return buf

    def _resblocks_f16x3(self, blocks, x: torch.Tensor, measured: bool = False, items=None) -> torch.Tensor:
        """xs = sum_j ResBlock1_j(x) / num_kernels  (models.py:104-110, :35-42) for one upsampling stage."""
        B, C, L = x.shape
        buf = self._cl_buffers(B, C, L)
        # activation pre-scale of this stage's split pairs: the power of two that brings max|x| of the stage input to 2^10
        # (measured on the device; the intermediates of a ResBlock stay within a few binades of its input)
        # (measured: the ConvTranspose1d that produced x left max|x| in zs_scratch)
        if not self.act_scales:
            zs = None
        elif measured:
            zs = ops.pow2_scale_from_amax(buf["zs_scratch"], 1024.0, buf["zs"])
        else:
            zs = ops.amax_pow2_scale(x, 1024.0, buf["zs"], buf["zs_scratch"])
        ops.hifigan_to_channels_last(x, buf["x0"], buf.get("z0"), LRELU_SLOPE, z_scale=zs)
        self._resblocks_cl(blocks, buf, B, L, zs, items)
        out = torch.empty_like(x)
        return ops.hifigan_from_channels_last(buf["xs"], out)

    def _resblocks_cl(self, blocks, buf: dict, B: int, L: int, zs, items=None) -> None:
        """buf["xs"] = sum_j ResBlock1_j(x) / num_kernels from buf["x0"] (fp32) and, on the wide stages, buf["z0"] =
        split(leaky_relu(x0) * zs): all channels-last."""
        narrow = "z0" not in buf
        nblk = len(blocks)
        if not narrow and 1 < nblk <= 3 and nblk <= len(buf.get("blk", ())) and all(len(b) == 3 for b in blocks) and GROUP_STAGE:
            # wide stage: the blocks' independent convolutions share launches (18 -> 8 per stage), same bits as block after block
            ops.hifigan_resblock_stage_f16x3(buf["x0"], buf["z0"], blocks, B, L, buf["blk"][:nblk], buf["xs"],
                                             out_scale=1.0 / self.num_kernels, z_scale=zs, items=items)
            return
        for j, block in enumerate(blocks):
            if len(block) == 3:                   # one C call per ResBlock: cvx_hifigan_resblock_f16x3 (6 launches, or 3 fused pairs)
                ops.hifigan_resblock_f16x3(buf["x0"], buf.get("z0"), block, B, L, buf, accum=buf["xs"] if j > 0 else None, out=buf["xs"],
                                           out_scale=(1.0 / self.num_kernels) if j == nblk - 1 else 1.0, z_scale=zs, items=items)
                continue
            if narrow:                            # other dilation counts: pair by pair
                cur = buf["x0"]
                for m, (c1, c2) in enumerate(block):
                    if m + 1 < len(block):
                        nxt = buf["r0"] if m % 2 == 0 else buf["r1"]
                        ops.hifigan_resblock_pair_f16x3(cur, c1, c2, B, L, nxt, z_scale=zs, items=items)
                        cur = nxt
                    else:
                        ops.hifigan_resblock_pair_f16x3(cur, c1, c2, B, L, buf["xs"], accum=buf["xs"] if j > 0 else None,
                                                        out_scale=(1.0 / self.num_kernels) if j == nblk - 1 else 1.0, z_scale=zs, items=items)
                continue
            cur_x, cur_z = buf["x0"], buf["z0"]   # other dilation counts: convolution by convolution
            for m, (c1, c2) in enumerate(block):
                ops.hifigan_conv1d_f16x3(cur_z, c1.w16, c1.bias16, B, L, ksize=c1.k, dil=c1.dil, out_z=buf["t"], z_slope=LRELU_SLOPE, z_scale=zs, items=items)
                if m + 1 < len(block):
                    ox, oz = (buf["r0"], buf["rz0"]) if m % 2 == 0 else (buf["r1"], buf["rz1"])
                    ops.hifigan_conv1d_f16x3(buf["t"], c2.w16, c2.bias16, B, L, ksize=c2.k, dil=c2.dil, res=cur_x, out_x=ox,
                                             out_z=oz, z_slope=LRELU_SLOPE, z_scale=zs, items=items)
                    cur_x, cur_z = ox, oz
                else:
                    ops.hifigan_conv1d_f16x3(buf["t"], c2.w16, c2.bias16, B, L, ksize=c2.k, dil=c2.dil, res=cur_x,
                                             accum=buf["xs"] if j > 0 else None, out_x=buf["xs"],
                                             out_scale=(1.0 / self.num_kernels) if j == nblk - 1 else 1.0, z_scale=zs, items=items)

    def _forward_channels_last(self, pk, x: torch.Tensor, lens, mul: int, add: int) -> torch.Tensor:
        """Everything behind conv_pre on channels-last buffers: per stage
            ConvTranspose1d on the split pipe (stride-1 form; leaves max|out| on the device) -> fp32 x0
            -> [wide stages] z0 = split(leaky_relu(x0) * zs)  -> the ResBlocks -> xs
            -> max|xs| -> the next upsampler's input pair split(leaky_relu(xs) * zs'),
        then conv_post + tanh straight from the last xs.  No channel-major tensor exists between conv_pre and the waveform
        (round 2 converted twice per stage and ran the upsamplers on the fp32 pipe: 1.9 of 11.4 ms)."""
        B, C, L = x.shape
        it = lambda: None if lens is None else (lens, mul, add)
        up0 = pk["ups"][0].w16t
        key = ("in", B, up0["cp_in"], L)
        zin = self._cl.get(key)
        if zin is None:
            shape = (B, ops.hifigan_cl_rows(L), up0["cp_in"])
            zin = dict(z=(torch.zeros(shape, dtype=torch.float16, device=self.device), torch.zeros(shape, dtype=torch.float16, device=self.device)),
                       zs=torch.ones(1, dtype=torch.float32, device=self.device), scr=torch.zeros(1, dtype=torch.int32, device=self.device))
            self._cl[key] = zin
        cur_zs = ops.amax_pow2_scale(x, 1024.0, zin["zs"], zin["scr"]) if self.act_scales else None
        ops.hifigan_to_channels_last(x, None, zin["z"], LRELU_SLOPE, z_scale=cur_zs)
        cur_z, lin = zin["z"], L
        for i in range(self.num_upsamples):
            up = pk["ups"][i]
            mul, add = self._affine(up, mul, add)
            lout = (lin - 1) * up.up + up.k - 2 * up.padding
            buf = self._cl_buffers(B, up.cout, lout)
            ops.hifigan_conv_transpose1d_f16x3(cur_z, up.w16t, B, lin, buf["x0"], lout, z_scale=cur_zs,
                                               amax_bits=buf["zs_scratch"] if self.act_scales else None, items=it())
            zs = ops.pow2_scale_from_amax(buf["zs_scratch"], 1024.0, buf["zs"]) if self.act_scales else None
            if "z0" in buf:
                ops.hifigan_split_channels_last(buf["x0"], buf["z0"], LRELU_SLOPE, z_scale=zs)
            self._resblocks_cl(pk["res"][i], buf, B, lout, zs, it())
            if i + 1 < self.num_upsamples:
                if "zn" not in buf:                   # the next upsampler's input pair (the wide stages lend their t pair)
                    buf["zn"] = buf["t"] if "t" in buf else (torch.zeros_like(buf["xs"], dtype=torch.float16), torch.zeros_like(buf["xs"], dtype=torch.float16))
                    buf["zs_out"] = torch.ones(1, dtype=torch.float32, device=self.device)
                cur_zs = ops.amax_pow2_scale(buf["xs"], 1024.0, buf["zs_out"], buf["zs_scratch"]) if self.act_scales else None
                ops.hifigan_split_channels_last(buf["xs"], buf["zn"], LRELU_SLOPE, z_scale=cur_zs)
                cur_z, lin = buf["zn"], lout
        y = torch.empty(B, 1, lout, dtype=torch.float32, device=x.device)
        return ops.hifigan_post_channels_last(buf["xs"], pk["ups"][-1].cout, lout, pk["post_w"], pk["post_b"], y, slope=0.01)


def mel_decode_to_wav(generator: Generator, mel: torch.Tensor):
    """reference monologue_generation.py:52-59: generator(mel) -> squeeze -> *32768 -> int16 numpy."""
    y = generator(mel)
    pcm = ops.wav_to_int16(y.squeeze().contiguous())
    return pcm.cpu().numpy()
