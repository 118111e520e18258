#!/usr/bin/env python3
"""MI355X drop-in for the reference's dialogue_generation.py (same flags; two prompts `<name>_1` / `<name>_2`,
dialogue_generation.py:285-286).  See neurips2024-covomix_amd/generation.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import covomix_amd  # noqa: E402,F401
from covomix_amd.generation import run  # noqa: E402

if __name__ == "__main__":
    run(dialogue=True)
