"""CPU: the prompt-mel oracle (oracle/mel_oracle.py) against (i) the reference-held
fixtures tests/golden/mel_ref_16k.npz (outputs of the reference's own mel_spectrogram, hifi-gan/hifigan_test), (ii) the
independent Slaney filter bank in `transformers`, (iii) a direct DFT."""
import os

import numpy as np
import pytest
import torch

import mel_oracle as mo


def test_slaney_basis_matches_independent_implementation():
    tf = pytest.importorskip("transformers.audio_utils")
    ref = tf.mel_filter_bank(num_frequency_bins=241, num_mel_filters=80, min_frequency=0.0, max_frequency=4000.0,
                             sampling_rate=8000, norm="slaney", mel_scale="slaney").T
    mine = mo.slaney_mel_basis()
    assert mine.shape == (80, 241) and mine.dtype == np.float32
    assert np.abs(mine - ref).max() < 1e-7 * max(1.0, np.abs(ref).max()) + 1e-8
    # known properties of the librosa filter bank: non-negative, every filter has support, slaney area normalisation
    assert (mine >= 0).all() and (mine.sum(axis=1) > 0).all()


def test_mel_spectrogram_shape_and_direct_dft():
    g = torch.Generator().manual_seed(0)
    y = (torch.rand(2, 16000, generator=g) * 2 - 1) * 0.5
    mel = mo.mel_spectrogram(y)
    assert mel.shape == (2, 80, 100) and torch.isfinite(mel).all()
    # frame 3 by hand: reflect padding, hann window, 480-point DFT in float64
    pad = torch.nn.functional.pad(y[:1, None].double(), (160, 160), mode="reflect")[0, 0]
    fr = pad[3 * 160: 3 * 160 + 480] * torch.hann_window(480, dtype=torch.float64)
    k = torch.arange(241, dtype=torch.float64)[:, None] * torch.arange(480, dtype=torch.float64)[None, :] * (2 * np.pi / 480)
    mag = torch.sqrt((torch.cos(k) @ fr) ** 2 + (torch.sin(k) @ fr) ** 2 + 1e-9)
    want = torch.log(torch.clamp(torch.from_numpy(mo.slaney_mel_basis()).double() @ mag, min=1e-5))
    assert float((mel[0, :, 3].double() - want).abs().max()) < 1e-4


def test_oracle_reproduces_reference_mel_fixtures():
    """The reference's own wav -> log-mel pairs (hifi-gan/meldataset.py:49-72 with librosa's filter bank; 16 kHz, n_fft =
    win = 1024, hop 256, fmax 8000): the restatement must reproduce them (measured 9.5e-7 max abs on the log-mel)."""
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "mel_ref_16k.npz"))
    kw = dict(sr=int(g["sr"]), n_fft=int(g["n_fft"]), hop=int(g["hop"]), win=int(g["win"]), n_mels=int(g["n_mels"]),
              fmin=float(g["fmin"]), fmax=float(g["fmax"]))
    for i in range(2):
        wav = torch.from_numpy(g[f"wav{i}"].astype(np.float32) / 32768.0)[None]
        mel = mo.mel_spectrogram(wav, **kw)[0]
        ref = torch.from_numpy(g[f"mel{i}"])
        assert mel.shape == ref.shape
        err = float((mel - ref).abs().max())
        print("mel fixture", i, tuple(ref.shape), "max abs err", err)
        assert err < 5e-6
