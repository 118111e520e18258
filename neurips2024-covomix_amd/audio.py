"""Wav-file plumbing shared by the prompt-mel extractor (mel.py) and the HuBERT feature reader (hubert.py)."""
from __future__ import annotations

import numpy as np


def pcm_to_float(data: np.ndarray) -> np.ndarray:
    """scipy.io.wavfile sample arrays -> float32 in [-1, 1): int16 / 2^15, int32 / 2^31, uint8 (x - 128) / 128, float as is.
    Converted by the ORIGINAL dtype before any channel arithmetic (a channel mean of int16 data is float64 and must still be
    divided by 32768)."""
    if data.dtype == np.int16:
        return data.astype(np.float32) / 32768.0
    if data.dtype == np.int32:
        return (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    if data.dtype == np.uint8:
        return (data.astype(np.float32) - 128.0) / 128.0
    if np.issubdtype(data.dtype, np.floating):
        return data.astype(np.float32)
    raise ValueError(f"unsupported wav sample type {data.dtype}")


def read_wav(path: str):
    """(sample_rate, float32 array [n] or [n, channels])"""
    from scipy.io.wavfile import read
    sr, data = read(path)
    return int(sr), pcm_to_float(np.asarray(data))
