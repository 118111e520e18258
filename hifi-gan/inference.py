#!/usr/bin/env python3
"""MI355X drop-in for the reference's hifi-gan/inference.py (same flags); see neurips2024-covomix_amd/hifigan_inference.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd  # noqa: E402,F401
from covomix_amd.hifigan_inference import inference  # noqa: E402

if __name__ == "__main__":
    inference()
