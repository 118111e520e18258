"""N > 1 through the GPU path on a box with ONE GPU: two ranks, both on cuda:0, process group on gloo
(CVX_DP_SINGLE_DEVICE=1, a test-only switch of covomix_amd.dp; RCCL cannot put two ranks on one device).
Covers what the driver's 8-GPU run exercises except the RCCL transport itself: `bench.py --gpus 2` started from a plain
shell launches its own ranks and prints one JSON line with n_gpus = 2, and the generation CLI under two ranks writes every
utterance exactly once with the audio the one-rank run writes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_generation_gpu import _write_fixture

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, CVX_DP_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def test_bench_self_launches_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 only, ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["ranks"]["elapsed_s"]) == 2 and out["config"]["parallelism"].startswith("dp2")
    assert "cpu_baseline" not in out and "fp32_exact" not in out        # N = 1 only
    # whole-job rate = frames of BOTH ranks / max elapsed
    assert abs(out["value"] - 2 * 8 * 1000 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-3
    # start-up diagnostics of the ranks block: transports and what RCCL depends on (here: the single-device test mode, so the
    # weight broadcast went through host memory on gloo and says so)
    rk = out["ranks"]
    assert rk["default_backend"] == "gloo" and rk["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and rk["host_threads_per_rank"] >= 1
    assert rk["broadcast_backend"].startswith("gloo") and "rccl_error" not in rk


def test_rccl_failure_falls_back_to_gloo_and_says_so():
    """Two ranks on ONE device with RCCL requested (CVX_DP_RCCL_ON_ONE_DEVICE=1, test only): RCCL refuses two ranks on a device -
    the kind of start-up failure a node can produce (e.g. hipIpcGetMemHandle under the legacy IPC mode).  The run must still
    complete: the proof all-reduce fails on the ranks, they agree over gloo, the weights go through host memory, and the bench
    line carries the reason."""
    env = _env()
    env.pop("CVX_DP_SINGLE_DEVICE")
    env["CVX_DP_RCCL_ON_ONE_DEVICE"] = "1"
    env["CVX_DP_RCCL_TIMEOUT_S"] = "40"             # (whether RCCL reports the duplicate device or stalls until the timeout: both must fall back)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rk = out["ranks"]
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert rk["requested_backend"] == "nccl" and rk["broadcast_backend"].startswith("gloo") and rk.get("rccl_error"), rk


def test_bench_strong_scaling_deals_64_utterances():
    """--scaling strong = BASELINE config 4 literally: 64 utterances per step dealt over the ranks (32 per rank here, four
    batches of 8)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--scaling", "strong"],
                       env=_env(), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["utterances_per_step"] == 64
    assert abs(out["value"] - 64 * 1000 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-3


def test_cli_two_ranks_write_every_utterance_once_like_one_rank(tmp_path):
    from scipy.io.wavfile import read
    tmp = str(tmp_path)
    _write_fixture(tmp, "vosingle")
    tdir, pdir = os.path.join(tmp, "text"), os.path.join(tmp, "prompt")
    os.makedirs(tdir); os.makedirs(pdir)
    g = np.random.RandomState(5)
    names = [f"utt{i}" for i in range(7)]
    for i, n in enumerate(names):                                       # all lengths distinct
        np.save(os.path.join(pdir, f"{n}.hubert_code.npy"), g.randint(0, 510, size=30))
        np.save(os.path.join(pdir, f"{n}.mel.npy"), (g.randn(80, 30) * 2 - 6).astype(np.float32))
        np.save(os.path.join(tdir, f"{n}.semantic.npy"), g.randint(0, 510, size=40 + 9 * i))
    base = [sys.executable, os.path.join(ROOT, "monologue_generation.py"), "--acous_ckpt", os.path.join(tmp, "acous.ckpt"),
            "--hifigan_ckpt", os.path.join(tmp, "voc", "g_00000001"), "--text_dir", tdir, "--prompt_dir", pdir,
            "--mode", "covosingle", "--seed", "30"]
    outs = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2"])):
        sdir = os.path.join(tmp, "out_" + tag)
        r = subprocess.run(base + ["--saved_dir", sdir] + extra, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        saved = [ln.split()[-1] for ln in r.stdout.splitlines() if ln.startswith("Saved wavfile")]
        assert sorted(os.path.basename(p) for p in saved) == sorted(n + ".wav" for n in names), (tag, saved)   # each exactly once
        if tag == "two":                                                # both ranks did some of the work
            per_rank = [ln for ln in r.stdout.splitlines() if ln.startswith("rank ")]
            assert len(per_rank) == 2 and all(" 0 utterances" not in ln for ln in per_rank), per_rank
        outs[tag] = {n: read(os.path.join(sdir, n + ".wav"))[1] for n in names}
    for n in names:
        a, b = outs["one"][n].astype(np.int32), outs["two"][n].astype(np.int32)
        assert a.shape == b.shape and a.shape[0] > 0
        # the same utterance with the same per-utterance noise; only the packing with other utterances may differ (fp32 rounding)
        print(n, "max |diff|", np.abs(a - b).max(), "fraction differing", (a != b).mean())
        assert np.abs(a - b).max() <= 2 and (a != b).mean() < 0.01, (n, np.abs(a - b).max(), (a != b).mean())
