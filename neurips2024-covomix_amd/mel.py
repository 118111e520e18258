"""Prompt mel extraction on the GPU - SURVEY.md section 8f row N3.

Host mirror of `extract_mel` / `mel_spectrogram` (monologue_generation.py:62-74, data_preparation/generate_mel.py:49-72)
with the constants the generation scripts set (monologue_generation.py:349-357): 8 kHz, n_fft = win = 480, hop 160,
80 mel bins, fmin 0, fmax 4000.

    mel = extract_mel(wav)          # wav float32 [n] in [-1, 1] (or a path to an 8 kHz wav file) -> [80, n // 160]

The STFT is not an FFT here: with a 480-sample window and 241 bins it is a [T, 480] x [480, 482] matrix product, i.e.
one call of the fp32 MFMA GEMM on the reflect-padded signal viewed as overlapping rows (row stride = hop), with the hann
window folded into the cos | sin basis; magnitude, the [80, 241] mel projection (second GEMM) and log(clamp) follow.

The mel basis is `librosa.filters.mel` in the reference (third-party, not available to this build): `slaney_mel_basis`
restates its published algorithm (Slaney scale, slaney normalisation); parity of this row is therefore pinned against
torch.stft + an independent implementation of that filter bank, not against the reference's own output.
"""
from __future__ import annotations

import math
from typing import Dict, Union

import numpy as np
import torch

from . import _lib, ops

SR, N_FFT, HOP, WIN, N_MELS, FMIN, FMAX = 8000, 480, 160, 480, 80, 0.0, 4000.0
_NB = N_FFT // 2 + 1                      # 241 frequency bins
_NBP = (_NB + 3) // 4 * 4                  # K of the mel GEMM padded to a multiple of 4


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr: int = SR, n_fft: int = N_FFT, n_mels: int = N_MELS, fmin: float = FMIN, fmax: float = FMAX) -> np.ndarray:
    """[n_mels, n_fft // 2 + 1] float32: librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults
    (Slaney mel scale, triangular filters on the FFT bin frequencies, norm='slaney')."""
    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        w[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    return (w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]).astype(np.float32)


_CONST: Dict[torch.device, tuple] = {}


def _constants(device: torch.device):
    c = _CONST.get(device)
    if c is None:
        n = torch.arange(N_FFT, dtype=torch.float64)
        k = torch.arange(_NB, dtype=torch.float64)
        ang = (2.0 * math.pi / N_FFT) * k[:, None] * n[None, :]
        win = torch.hann_window(WIN, dtype=torch.float64)
        dft = torch.cat((torch.cos(ang) * win, torch.sin(ang) * win), dim=0).float().contiguous().to(device)      # [482, 480]
        basis = torch.zeros(N_MELS, _NBP, dtype=torch.float32)
        basis[:, :_NB] = torch.from_numpy(slaney_mel_basis())
        c = _CONST[device] = (dft, basis.to(device))
    return c


@torch.no_grad()
def mel_spectrogram(y: torch.Tensor) -> torch.Tensor:
    """y [B, n] or [n] float32 in [-1, 1] on the GPU -> [B, 80, T] (or [80, T]) log-mel, T = n // 160."""
    if y.device.type != "cuda":
        raise _lib.CovomixHipError("mel_spectrogram needs the waveform on a GPU: covomix_amd has no CPU path")
    single = y.ndim == 1
    y = (y[None] if single else y).to(torch.float32)
    dft, basis = _constants(y.device)
    pad = (N_FFT - HOP) // 2
    yr = torch.nn.functional.pad(y[:, None], (pad, pad), mode="reflect")[:, 0]                    # index plumbing only
    yp = torch.zeros(yr.shape[0], (yr.shape[1] + 3) // 4 * 4, dtype=torch.float32, device=y.device)      # 16-byte aligned rows
    yp[:, : yr.shape[1]] = yr
    n_pad = yr.shape[1]
    out = []
    for b in range(yp.shape[0]):
        sig = yp[b]
        T = (n_pad - N_FFT) // HOP + 1
        frames = sig.as_strided((T, N_FFT), (HOP, 1))                  # overlapping rows: frame t = sig[160 t : 160 t + 480]
        spec = torch.empty(T, 2 * _NB, dtype=torch.float32, device=y.device)
        ops.gemm(frames, dft, spec)
        mag = torch.empty(T, _NBP, dtype=torch.float32, device=y.device)
        _lib.check(_lib.load().cvx_mel_magnitude_f32(spec.data_ptr(), mag.data_ptr(), T, _NB, _NBP,
                                                     torch.cuda.current_stream().cuda_stream), "cvx_mel_magnitude_f32")
        proj = torch.empty(T, N_MELS, dtype=torch.float32, device=y.device)
        ops.gemm(mag, basis, proj)
        mel = torch.empty(N_MELS, T, dtype=torch.float32, device=y.device)
        _lib.check(_lib.load().cvx_mel_log_transpose_f32(proj.data_ptr(), mel.data_ptr(), T, N_MELS,
                                                         torch.cuda.current_stream().cuda_stream), "cvx_mel_log_transpose_f32")
        out.append(mel)
    return out[0] if single else torch.stack(out)


def extract_mel(x: Union[str, np.ndarray, torch.Tensor], device: Union[str, torch.device] = "cuda", channel_idx=None) -> torch.Tensor:
    """monologue_generation.py:62-74: wav -> clip to [-1, 1] -> mel_spectrogram -> [80, T] on the CPU.  A path must be an
    8 kHz wav file (the reference resamples with librosa, which this build does not ship)."""
    if isinstance(x, str):
        from scipy.io.wavfile import read
        sr, data = read(x)
        if sr != SR:
            raise ValueError(f"{x}: {sr} Hz - prompts must be {SR} Hz (resample first; the reference uses librosa.load(sr=8000))")
        if data.ndim == 2:
            data = data[:, channel_idx] if channel_idx is not None else data.mean(axis=1)
        wav = data.astype(np.float32) / 32768.0 if data.dtype == np.int16 else data.astype(np.float32)
    else:
        wav = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        wav = wav.astype(np.float32)
    wav = torch.from_numpy(np.clip(wav, -1, 1)).to(device)
    return mel_spectrogram(wav).cpu()
