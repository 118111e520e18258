"""CPU restatement of the reference's prompt mel extraction - SURVEY.md section 8f row N3.

TEST INFRASTRUCTURE ONLY (nothing under neurips2024-covomix_amd/ imports this file).

Follows data_preparation/generate_mel.py:49-72 (`mel_spectrogram`) as called by `extract_mel`
(monologue_generation.py:62-74) with the constants of monologue_generation.py:349-357
(8 kHz, n_fft = win = 480, hop 160, 80 mels, fmin 0, fmax 4000):
    reflect-pad (n_fft - hop)/2 = 160 samples on both sides -> torch.stft(center=False, hann window) ->
    sqrt(re^2 + im^2 + 1e-9) -> mel basis [80, 241] @ magnitude -> log(clamp(., 1e-5)).

PINNED against outputs of the reference itself: hifi-gan/hifigan_test/input_wav/*.wav -> input_mel/*.npy were written by
the reference's own `mel_spectrogram` (hifi-gan/meldataset.py:49-72 - the same function as
data_preparation/generate_mel.py:49-72, with librosa's filter bank) at 16 kHz, n_fft = win = 1024, hop 256, fmax 8000;
tests/golden/make_golden_mel.py packs the two pairs into tests/golden/mel_ref_16k.npz and tests/test_mel_oracle.py holds
this restatement to them (measured: max |log-mel difference| 9.5e-7).  The mel basis is `librosa.filters.mel`
(third-party, absent from this image): `slaney_mel_basis` restates its published algorithm (Slaney mel scale: linear
below 1 kHz, log above with step log(6.4)/27; triangular filters on the FFT bin frequencies; slaney area normalisation
2 / (f[i+2] - f[i])), additionally cross-checked against the independent implementation in
transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney").  The STFT is torch's own.  Every function
takes the analysis parameters (sr, n_fft, hop, win, n_mels, fmin, fmax); the defaults are the generation scripts' 8 kHz set.
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


def mel_spectrogram(y: torch.Tensor, basis: torch.Tensor = None, sr=SR, n_fft=N_FFT, hop=HOP, win=WIN, n_mels=N_MELS,
                    fmin=FMIN, fmax=FMAX) -> torch.Tensor:
    """y [B, n] float32 in [-1, 1] -> [B, n_mels, T] log-mel, T = n // hop (generate_mel.py:49-72 = hifi-gan/meldataset.py:49-72)."""
    if basis is None:
        basis = torch.from_numpy(slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax))
    pad = int((n_fft - hop) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    spec = torch.matmul(basis, spec)
    return torch.log(torch.clamp(spec, min=1e-5))
