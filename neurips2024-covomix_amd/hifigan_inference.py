"""MI355X drop-ins for the reference's HiFi-GAN command-line callers of Generator (the vocoder side of the hot path):

  inference_e2e   hifi-gan/inference_e2e.py:36-92   every <name>.npy mel [80, T] in --input_mels_dir -> <name>_generated_e2e.wav
  inference       hifi-gan/inference.py:46-80        every *.wav in --input_wavs_dir -> log-mel (meldataset.mel_spectrogram with the
                                                     config's n_fft / hop / win / fmin / fmax) -> Generator -> <name>_generated.wav
                                                     (copy synthesis; the reference's PESQ / STOI scoring needs third-party
                                                     packages and is not part of the path)
Same flags; the config is `config.json` beside --checkpoint_file (inference_e2e.py:78).  Everything numeric runs through the
C ABI (Generator, cvx_wav_to_int16, the prompt-mel kernels); there is no CPU path.
"""
from __future__ import annotations

import argparse
import glob
import json
import os

import numpy as np
import torch

from . import mel as melmod
from .vocoder import AttrDict, Generator, mel_decode_to_wav


def _load(checkpoint_file: str):
    config_file = os.path.join(os.path.split(checkpoint_file)[0], "config.json")
    with open(config_file) as f:
        h = AttrDict(json.loads(f.read()))
    if not torch.cuda.is_available():
        from ._lib import CovomixHipError
        raise CovomixHipError("HiFi-GAN inference needs an MI355X: covomix_amd has no CPU path")
    torch.manual_seed(h.seed)
    torch.cuda.manual_seed(h.seed)
    device = torch.device("cuda")
    generator = Generator(h).to(device)
    assert os.path.isfile(checkpoint_file)
    print("Loading '{}'".format(checkpoint_file))
    state = torch.load(checkpoint_file, map_location="cpu", weights_only=False)
    generator.load_state_dict(state["generator"])
    generator.eval()
    generator.remove_weight_norm()
    return h, generator, device


def inference_e2e(argv=None) -> int:
    p = argparse.ArgumentParser()
    p.add_argument("--input_mels_dir", default="test_mel_files")
    p.add_argument("--output_dir", default="generated_files_from_mel")
    p.add_argument("--checkpoint_file", required=True)
    a = p.parse_args(argv)
    from scipy.io.wavfile import write
    h, generator, device = _load(a.checkpoint_file)
    os.makedirs(a.output_dir, exist_ok=True)
    n = 0
    with torch.no_grad():
        for filname in sorted(os.listdir(a.input_mels_dir)):
            if not filname.endswith(".npy"):
                continue
            x = torch.from_numpy(np.load(os.path.join(a.input_mels_dir, filname)).astype(np.float32)).to(device).unsqueeze(0)
            audio = mel_decode_to_wav(generator, x)
            output_file = os.path.join(a.output_dir, os.path.splitext(filname)[0] + "_generated_e2e.wav")
            write(output_file, h.sampling_rate, audio)
            print(output_file)
            n += 1
    return n


def inference(argv=None) -> int:
    p = argparse.ArgumentParser()
    p.add_argument("--input_wavs_dir", default="test_files")
    p.add_argument("--output_dir", default="generated_files")
    p.add_argument("--checkpoint_file", required=True)
    a = p.parse_args(argv)
    from scipy.io.wavfile import read, write
    h, generator, device = _load(a.checkpoint_file)
    want = (melmod.SR, melmod.N_FFT, melmod.HOP, melmod.WIN, melmod.N_MELS, melmod.FMIN, melmod.FMAX)
    have = (h.sampling_rate, h.n_fft, h.hop_size, h.win_size, h.num_mels, float(h.fmin), float(h.fmax))
    if have != want:
        raise ValueError(f"mel parameters {have} differ from the ones this build extracts {want} (hifi-gan/config_covomix.json)")
    os.makedirs(a.output_dir, exist_ok=True)
    n = 0
    with torch.no_grad():
        for wavfile in sorted(glob.glob(os.path.join(a.input_wavs_dir, "*.wav"))):
            sr, data = read(wavfile)
            if sr != h.sampling_rate:
                raise ValueError(f"{wavfile}: {sr} Hz, the vocoder config says {h.sampling_rate}")
            wav = data.astype(np.float32) / 32768.0 if data.dtype == np.int16 else data.astype(np.float32)     # wav / MAX_WAV_VALUE
            x = melmod.mel_spectrogram(torch.from_numpy(wav).to(device).unsqueeze(0))                           # get_mel, inference.py:30-31
            audio = mel_decode_to_wav(generator, x)
            output_file = os.path.join(a.output_dir, os.path.splitext(os.path.basename(wavfile))[0] + "_generated.wav")
            write(output_file, h.sampling_rate, audio)
            n += 1
    return n
