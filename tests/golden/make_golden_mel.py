#!/usr/bin/env python3
"""Pack the reference-held prompt-mel fixtures into tests/golden/mel_ref_16k.npz (row N3 pin).

/root/reference/hifi-gan/hifigan_test/input_wav/<name>.wav and input_mel/<name>.npy are an input / output pair of the
reference's own `mel_spectrogram` (hifi-gan/meldataset.py:49-72; identical to data_preparation/generate_mel.py:49-72)
at the hifi-gan test settings: 16 kHz, n_fft = win = 1024, hop 256, 80 mels, fmin 0, fmax 8000, wav / 32768.
Only DATA is copied (int16 samples and float32 log-mels).  Run in the build container:  python tests/golden/make_golden_mel.py"""
import os

import numpy as np
from scipy.io.wavfile import read

REF = "/root/reference/hifi-gan/hifigan_test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mel_ref_16k.npz")
NAMES = ["908-31957-0024_5142-36586-0004", "908-31957-0024_5683-32865-0017"]

out = dict(sr=np.int64(16000), n_fft=np.int64(1024), hop=np.int64(256), win=np.int64(1024), n_mels=np.int64(80),
           fmin=np.float64(0.0), fmax=np.float64(8000.0))
for i, n in enumerate(NAMES):
    sr, wav = read(os.path.join(REF, "input_wav", n + ".wav"))
    mel = np.load(os.path.join(REF, "input_mel", n + ".npy"))
    assert sr == 16000 and wav.dtype == np.int16 and mel.shape == (80, wav.shape[0] // 256) and mel.dtype == np.float32
    out[f"wav{i}"], out[f"mel{i}"] = wav, mel
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
