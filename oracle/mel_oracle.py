"""CPU restatement of the reference's prompt mel extraction - SURVEY.md section 8f row N3.

TEST INFRASTRUCTURE ONLY (nothing under neurips2024-covomix_amd/ imports this file).

Follows data_preparation/generate_mel.py:49-72 (`mel_spectrogram`) as called by `extract_mel`
(monologue_generation.py:62-74) with the constants of monologue_generation.py:349-357
(8 kHz, n_fft = win = 480, hop 160, 80 mels, fmin 0, fmax 4000):
    reflect-pad (n_fft - hop)/2 = 160 samples on both sides -> torch.stft(center=False, hann window) ->
    sqrt(re^2 + im^2 + 1e-9) -> mel basis [80, 241] @ magnitude -> log(clamp(., 1e-5)).

PARITY UNPINNED against the reference: the mel basis is `librosa.filters.mel` (third-party, absent from this image
and from /root/reference) and generate_mel.py cannot be imported (librosa, torchaudio, wespeakerruntime, soundfile).
`slaney_mel_basis` restates librosa's published algorithm (Slaney mel scale: linear below 1 kHz, log above with
step log(6.4)/27; triangular filters on the FFT bin frequencies; slaney area normalisation 2 / (f[i+2] - f[i])) and is
cross-checked in tests/test_mel_oracle.py against an independent implementation of the same algorithm that IS installed
here (transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")).  The STFT is torch's own.
"""
import math

import numpy as np
import torch

SR, N_FFT, HOP, WIN, N_MELS, FMIN, FMAX = 8000, 480, 160, 480, 80, 0.0, 4000.0


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX) -> np.ndarray:
    """[n_mels, n_fft//2 + 1] float32, librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) defaults (htk=False, norm='slaney')."""
    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def mel_spectrogram(y: torch.Tensor, basis: torch.Tensor = None) -> torch.Tensor:
    """y [B, n] float32 in [-1, 1] -> [B, 80, T] log-mel, T = n // 160 (generate_mel.py:49-72)."""
    if basis is None:
        basis = torch.from_numpy(slaney_mel_basis())
    pad = (N_FFT - HOP) // 2
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, N_FFT, hop_length=HOP, win_length=WIN, window=torch.hann_window(WIN), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    spec = torch.matmul(basis, spec)
    return torch.log(torch.clamp(spec, min=1e-5))
