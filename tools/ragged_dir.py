#!/usr/bin/env python3
"""Dev: the ragged test directory of tests/test_generation_gpu.py::test_cli_rate_on_ragged_directory (16 utterances of distinct T in
[400, 1200], 40 % prompt, full-width VoMix + config_covomix HiFi-GAN, recipe weights) through generation.run, PASSES times, with the
per-batch wall times of run.last_stats - to be run under rocprofv3 --kernel-trace --stats for the per-kernel picture.
Env: PASSES=3, LENGTHS="400,1200,..." (frames per utterance), MAX_FRAMES (default: the CLI's own choice), OUT=/tmp/ragged_dir."""
import json, os, sys, warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covomix_amd.synthetic as syn
from covomix_amd import generation
tmp = os.environ.get("OUT", "/tmp/ragged_dir")
os.makedirs(tmp, exist_ok=True)
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.acoustic_param_shapes(), seed=0).items()}
torch.save({"state_dict": {"cfm_wrapper.CoVoMix." + k: v for k, v in sd.items()}, "hyper_parameters": {"twocondition_oneoutput": True}},
           os.path.join(tmp, "acous.ckpt"))
h = dict(syn.HIFIGAN_COVOMIX_CONFIG)
vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
os.makedirs(os.path.join(tmp, "voc"), exist_ok=True)
torch.save({"generator": vsd}, os.path.join(tmp, "voc", "g_00000001"))
json.dump(h, open(os.path.join(tmp, "voc", "vocoder_config.json"), "w"))
tdir, pdir, sdir = (os.path.join(tmp, d) for d in ("text", "prompt", "out"))
os.makedirs(tdir, exist_ok=True); os.makedirs(pdir, exist_ok=True)
g = np.random.RandomState(7)
lengths = [int(v) for v in os.environ.get("LENGTHS", "400,1200,451,1149,503,1097,555,1044,607,993,659,941,711,889,763,837").split(",")]
for i, T in enumerate(lengths):
    P = int(0.4 * T)
    for suf in ("_1", "_2"):
        np.save(os.path.join(pdir, f"u{i:02d}{suf}.hubert_code.npy"), g.randint(0, 500, size=P))
        np.save(os.path.join(pdir, f"u{i:02d}{suf}.mel.npy"), (g.randn(80, P) * 2 - 6).astype(np.float32))
    np.save(os.path.join(tdir, f"u{i:02d}.semantic.npy"), g.randint(0, 500, size=(2, T - P)))
argv = ["--acous_ckpt", os.path.join(tmp, "acous.ckpt"), "--hifigan_ckpt", os.path.join(tmp, "voc", "g_00000001"),
        "--text_dir", tdir, "--prompt_dir", pdir, "--saved_dir", sdir, "--mode", "covomix", "--seed", "1"]
if os.environ.get("MAX_FRAMES"):
    argv += ["--max_frames", os.environ["MAX_FRAMES"]]
import contextlib, io
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for k in range(int(os.environ.get("PASSES", "3")) + 1):
        with contextlib.redirect_stdout(io.StringIO()):
            generation.run(True, argv)
        st = generation.run.last_stats
        if k:           # (the first pass pays first-launch costs)
            print(f"pass {k}: {st['frames'] / st['seconds']:.0f} generated frames/s, {st['seconds'] * 1e3:.1f} ms = files {st['load_seconds'] * 1e3:.1f} ms + "
                  + ", ".join(f"({n} utt, {fr} fr, {sec * 1e3:.1f} ms)" for n, fr, _, sec in st["batches"]) + f"; max_frames {st['max_frames']}", flush=True)
