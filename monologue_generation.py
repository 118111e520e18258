#!/usr/bin/env python3
"""MI355X drop-in for the reference's monologue_generation.py (same flags, monologue_generation.py:324-333).
See neurips2024-covomix_amd/generation.py for what is computed here and what is read from files."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import covomix_amd  # noqa: E402,F401
from covomix_amd.generation import run  # noqa: E402

if __name__ == "__main__":
    run(dialogue=False)
