"""GPU: prompt mel extraction through the C ABI against the reference-held fixtures
(tests/golden/mel_ref_16k.npz: outputs of the reference's own mel_spectrogram) and the CPU oracle (torch.stft + the
restated Slaney filter bank).  Tolerance 5e-5 absolute on the log-mel (fp32 DFT-as-GEMM vs torch's FFT)."""
import os

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
    # a 16 kHz file is resampled to 8 kHz on the GPU (the reference: librosa.load(sr=8000)); a 2:1 windowed-sinc decimation
    # of a band-limited signal must give the mel of the same signal sampled at 8 kHz
    t16 = np.arange(32000) / 16000.0
    sig = lambda t: 0.4 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1333 * t + 1.0)
    write(p, 16000, (sig(t16) * 32767).astype(np.int16))
    m16 = mel.extract_mel(p)
    ref8 = mo.mel_spectrogram(torch.from_numpy(sig(np.arange(16000) / 8000.0).astype(np.float32))[None])[0]
    assert m16.shape == (80, 100)
    assert float((m16[:, 2:-2] - ref8[:, 2:-2]).abs().mean()) < 0.05
    # stereo int16 without a channel index: the channel MEAN, scaled by the stored sample type (was clipped garbage)
    st = np.stack((y, (y // 2).astype(np.int16)), axis=1)
    write(p, 8000, st)
    ms = mel.extract_mel(p)
    refs = mo.mel_spectrogram(torch.from_numpy((st.astype(np.float32) / 32768.0).mean(axis=1))[None])[0]
    assert float((ms - refs).abs().max()) < 5e-5
    m1 = mel.extract_mel(p, channel_idx=1)
    ref1 = mo.mel_spectrogram(torch.from_numpy(st[:, 1].astype(np.float32) / 32768.0)[None])[0]
    assert float((m1 - ref1).abs().max()) < 5e-5
    # int32 and uint8 PCM
    write(p, 8000, (y.astype(np.int32) << 16))
    assert float((mel.extract_mel(p) - ref).abs().max()) < 5e-5
    write(p, 8000, ((y.astype(np.int32) >> 8) + 128).astype(np.uint8))
    assert mel.extract_mel(p).shape == (80, 50)


def test_mel_vs_reference_fixtures():
    """Row N3 pin: the reference's own wav -> log-mel pairs (16 kHz, n_fft = win = 1024, hop 256, fmax 8000)."""
    from conftest import GOLDEN
    from covomix_amd import mel
    g = np.load(os.path.join(GOLDEN, "mel_ref_16k.npz"))
    for i in range(2):
        wav = torch.from_numpy(g[f"wav{i}"].astype(np.float32) / 32768.0).cuda()
        got = mel.mel_spectrogram(wav, int(g["n_fft"]), int(g["n_mels"]), int(g["sr"]), int(g["hop"]), int(g["win"]),
                                  float(g["fmin"]), float(g["fmax"])).cpu()
        ref = torch.from_numpy(g[f"mel{i}"])
        err = float((got - ref).abs().max())
        print("mel fixture", i, "max abs err (HIP path)", err)
        assert got.shape == ref.shape and err < 5e-5
