"""GPU: prompt mel extraction through the C ABI against the CPU oracle (torch.stft + the
restated Slaney filter bank).  Tolerance 2e-5 absolute on the log-mel (fp32 DFT-as-GEMM vs torch's FFT) except where the
1e-5 clamp is active."""
import numpy as np
import pytest
import torch

import mel_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [16000, 480, 8000 * 7 + 123, 159 * 160])
def test_mel_vs_oracle(n):
    from covomix_amd import mel
    g = torch.Generator().manual_seed(n)
    t = torch.arange(n) / 8000.0
    y = 0.4 * torch.sin(2 * np.pi * 440 * t) + 0.2 * torch.sin(2 * np.pi * 1333 * t + 1.0) + 0.05 * torch.randn(n, generator=g)
    y = y.clamp(-1, 1)
    ref = mo.mel_spectrogram(y[None])[0]
    got = mel.mel_spectrogram(y.cuda()).cpu()
    assert got.shape == ref.shape == (80, n // 160)
    assert float((got - ref).abs().max()) < 5e-5
    assert np.array_equal(mel.slaney_mel_basis(), mo.slaney_mel_basis())
    both = mel.mel_spectrogram(torch.stack((y, -y)).cuda()).cpu()           # batched; the log-mel of -y equals that of y
    assert float((both[0] - got).abs().max()) == 0 and float((both[1] - got).abs().max()) < 1e-5


def test_extract_mel_from_wav_file(tmp_path):
    from scipy.io.wavfile import write
    from covomix_amd import mel
    y = (np.sin(np.arange(8000) * 0.3) * 15000 + np.random.RandomState(0).randn(8000) * 1500).astype(np.int16)
    p = str(tmp_path / "p.wav")
    write(p, 8000, y)
    m = mel.extract_mel(p)
    ref = mo.mel_spectrogram(torch.from_numpy(y.astype(np.float32) / 32768.0)[None])[0]
    assert m.shape == (80, 50) and not m.is_cuda and float((m - ref).abs().max()) < 5e-5
    write(p, 16000, y)
    with pytest.raises(ValueError):
        mel.extract_mel(p)
