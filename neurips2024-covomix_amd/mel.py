"""Prompt mel extraction on the GPU - SURVEY.md section 8f row N3.

Host mirror of `extract_mel` / `mel_spectrogram` (monologue_generation.py:62-74, data_preparation/generate_mel.py:49-72 =
hifi-gan/meldataset.py:49-72), parameterised like the reference function by (sr, n_fft, hop, win, n_mels, fmin, fmax); the
defaults are the constants the generation scripts set (monologue_generation.py:349-357): 8 kHz, n_fft = win = 480,
hop 160, 80 mel bins, fmin 0, fmax 4000.

    mel = extract_mel(wav)          # wav float32 [n] in [-1, 1], or a wav file of any rate / sample type -> [80, n // 160]

The STFT is not an FFT here: with an n_fft-sample window and n_fft/2+1 bins it is a [T, n_fft] x [n_fft, 2*bins] matrix
product, i.e. one call of the fp32 MFMA GEMM on the reflect-padded signal viewed as overlapping rows (row stride = hop),
with the hann window folded into the cos | sin basis; magnitude, the mel projection (second GEMM) and log(clamp) follow.

The mel basis is `librosa.filters.mel` in the reference (third-party, not in this image): `slaney_mel_basis` restates its
published algorithm (Slaney scale, slaney normalisation).  PINNED: tests/golden/mel_ref_16k.npz holds two wav -> mel pairs
written by the reference's own function (16 kHz / 1024 / 256 / fmax 8000); this module reproduces them to <= 5e-5 on the
log-mel (tests/test_mel_gpu.py), the CPU oracle to 1e-6.  Files at another rate than `sr` are resampled on the GPU
(cvx_resample_fir_f32, windowed sinc; the reference calls librosa.load(sr=8000) - third-party, unpinned).
"""
from __future__ import annotations

import math
from typing import Dict, Union

import numpy as np
import torch

from . import _lib, ops

SR, N_FFT, HOP, WIN, N_MELS, FMIN, FMAX = 8000, 480, 160, 480, 80, 0.0, 4000.0

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


_CONST: Dict[tuple, tuple] = {}


def _constants(device: torch.device, sr: int, n_fft: int, win: int, n_mels: int, fmin: float, fmax: float):
    key = (device, sr, n_fft, win, n_mels, float(fmin), float(fmax))
    c = _CONST.get(key)
    if c is None:
        if win != n_fft:
            raise NotImplementedError("mel_spectrogram: win_size must equal n_fft (every caller of the reference does)")
        nb = n_fft // 2 + 1
        nbp = (nb + 3) // 4 * 4                                        # K of the mel GEMM padded to a multiple of 4
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(nb, dtype=torch.float64)
        ang = (2.0 * math.pi / n_fft) * k[:, None] * n[None, :]
        w = torch.hann_window(win, dtype=torch.float64)
        dft = torch.cat((torch.cos(ang) * w, torch.sin(ang) * w), dim=0).float().contiguous().to(device)      # [2*nb, n_fft]
        basis = torch.zeros(n_mels, nbp, dtype=torch.float32)
        basis[:, :nb] = torch.from_numpy(slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax))
        if len(_CONST) >= 8:
            _CONST.clear()
        c = _CONST[key] = (dft, basis.to(device), nb, nbp)
    return c


@torch.no_grad()
def mel_spectrogram(y: torch.Tensor, n_fft: int = N_FFT, num_mels: int = N_MELS, sampling_rate: int = SR, hop_size: int = HOP,
                    win_size: int = WIN, fmin: float = FMIN, fmax: float = FMAX, center: bool = False) -> torch.Tensor:
    """y [B, n] or [n] float32 in [-1, 1] on the GPU -> [B, num_mels, T] (or [num_mels, T]) log-mel, T = n // hop_size.
    Argument names and order follow the reference function (generate_mel.py:49)."""
    if y.device.type != "cuda":
        raise _lib.CovomixHipError("mel_spectrogram needs the waveform on a GPU: covomix_amd has no CPU path")
    if center:
        raise NotImplementedError("center=True is not used by the reference's callers")
    if n_fft % 4 or (n_fft - hop_size) % 2 or hop_size % 4:
        raise ValueError("mel_spectrogram: n_fft and hop_size must be multiples of 4 (16-byte aligned overlapping rows)")
    single = y.ndim == 1
    y = (y[None] if single else y).to(torch.float32)
    dft, basis, nb, nbp = _constants(y.device, int(sampling_rate), int(n_fft), int(win_size), int(num_mels), fmin, fmax)
    pad = int((n_fft - hop_size) / 2)
    yr = torch.nn.functional.pad(y[:, None], (pad, pad), mode="reflect")[:, 0]                    # index plumbing only
    yp = torch.zeros(yr.shape[0], (yr.shape[1] + 3) // 4 * 4, dtype=torch.float32, device=y.device)      # 16-byte aligned rows
    yp[:, : yr.shape[1]] = yr
    n_pad = yr.shape[1]
    out = []
    st = ops._stream()
    for b in range(yp.shape[0]):
        sig = yp[b]
        T = (n_pad - n_fft) // hop_size + 1
        frames = sig.as_strided((T, n_fft), (hop_size, 1))             # overlapping rows: frame t = sig[hop t : hop t + n_fft]
        spec = torch.empty(T, 2 * nb, dtype=torch.float32, device=y.device)
        ops.gemm(frames, dft, spec)
        mag = torch.empty(T, nbp, dtype=torch.float32, device=y.device)
        _lib.check(_lib.load().cvx_mel_magnitude_f32(spec.data_ptr(), mag.data_ptr(), T, nb, nbp, st), "cvx_mel_magnitude_f32")
        proj = torch.empty(T, num_mels, dtype=torch.float32, device=y.device)
        ops.gemm(mag, basis, proj)
        mel = torch.empty(num_mels, T, dtype=torch.float32, device=y.device)
        _lib.check(_lib.load().cvx_mel_log_transpose_f32(proj.data_ptr(), mel.data_ptr(), T, num_mels, st), "cvx_mel_log_transpose_f32")
        out.append(mel)
    return out[0] if single else torch.stack(out)


def extract_mel(x: Union[str, np.ndarray, torch.Tensor], device: Union[str, torch.device] = "cuda", channel_idx=None,
                sr: int = SR, **mel_kw) -> torch.Tensor:
    """monologue_generation.py:62-74: wav -> (mono, `sr` Hz) -> clip to [-1, 1] -> mel_spectrogram -> [80, T] on the CPU.
    A path may be a wav file of any sample rate and PCM type: samples are converted to float by their stored type, the
    channel is selected (channel_idx) or the channels averaged (librosa.load's mono=True), and other rates are resampled
    to `sr` on the GPU with the windowed-sinc FIR the HuBERT reader uses."""
    if isinstance(x, str):
        from .audio import read_wav
        file_sr, data = read_wav(x)
        if data.ndim == 2:
            data = data[:, channel_idx] if channel_idx is not None else data.mean(axis=1, dtype=np.float32)
        wav = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(device)
        if file_sr != sr:
            from .hubert import resample
            wav = resample(wav, file_sr, sr)
    else:
        wav = x.detach() if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x, dtype=np.float32))
        wav = wav.to(device=device, dtype=torch.float32)
    return mel_spectrogram(wav.clamp(-1.0, 1.0), sampling_rate=sr, **mel_kw).cpu()
