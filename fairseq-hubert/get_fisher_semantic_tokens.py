#!/usr/bin/env python3
"""MI355X drop-in for the reference's fairseq-hubert/get_fisher_semantic_tokens.py (same flags, :20-27): every *.wav in
--process_dir -> <name>.hubert_code.npy in --target_dir (HuBERT layer 12 + k-means codes of channel 1).
See neurips2024-covomix_amd/hubert.py.  Under torchrun the files are dealt round-robin to the ranks (one GPU each)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd  # noqa: E402,F401
from covomix_amd.hubert import tokenize_directory  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--process_dir", type=str, required=True, help="Directory containing the wav data")
    parser.add_argument("--target_dir", type=str, required=True, help="Directory containing the generated semantic tokens")
    parser.add_argument("--hubert_path", type=str, required=True, help="HuBERT checkpoint (fairseq layout)")
    parser.add_argument("--km_path", type=str, required=True, help="k-means model (joblib)")
    args = parser.parse_args()
    n = tokenize_directory(args.process_dir, args.target_dir, args.hubert_path, args.km_path)
    print(f"tokenised {n} files")
