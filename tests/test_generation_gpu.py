"""End-to-end CLI drop-in on the GPU: synthetic checkpoints in the reference's on-disk layouts
(Lightning .ckpt with EMA; HiFi-GAN g_xxxx + vocoder_config.json), prompts/tokens as files,
monologue (covosingle) and dialogue (covomix) modes; outputs checked against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_fixture(tmp, kind):
    import covomix_amd.synthetic as syn
    two = kind == "vomix"
    shapes = syn.acoustic_param_shapes(dim=128, dim_cond=160 if two else 80, dim_emb=64, depth=4, heads=2,
                                       streams=2 if two else 1)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=0).items()}
    ema = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, seed=3).items()}
    full = {"cfm_wrapper.CoVoMix." + k: v for k, v in sd.items()}
    full["cfm_wrapper.CoVoMix.transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    torch.save({"state_dict": full, "hyper_parameters": {"twocondition_oneoutput": two},
                "ema": {"decay": 0.999, "num_updates": 1, "shadow_params": list(ema.values()), "collected_params": None}},
               os.path.join(tmp, "acous.ckpt"))
    h = dict(syn.HIFIGAN_COVOMIX_CONFIG); h["upsample_initial_channel"] = 32
    vsd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.hifigan_param_shapes(h), seed=0).items()}
    os.makedirs(os.path.join(tmp, "voc"), exist_ok=True)
    torch.save({"generator": vsd}, os.path.join(tmp, "voc", "g_00000001"))
    json.dump(h, open(os.path.join(tmp, "voc", "vocoder_config.json"), "w"))
    ema["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
    return ema, vsd, h


@pytest.mark.parametrize("mode,dialogue", [("covosingle", False), ("covomix", True), ("covosinx", False)])
def test_cli_end_to_end(tmp_path, mode, dialogue, monkeypatch):
    import covomix_oracle as orc
    from covomix_amd import assembly, generation
    tmp = str(tmp_path)
    kind = "vosingle" if mode == "covosingle" else "vomix"
    ema, vsd, h = _write_fixture(tmp, kind)
    tdir, pdir, sdir = (os.path.join(tmp, d) for d in ("text", "prompt", "out"))
    os.makedirs(tdir); os.makedirs(pdir)
    g = np.random.RandomState(0)
    names = ["utt_a", "utt_b", "utt_c"]
    npred = {"utt_a": 60, "utt_b": 60, "utt_c": 37}
    for n in names:
        for suf in (["_1", "_2"] if dialogue else [""]):
            np.save(os.path.join(pdir, f"{n}{suf}.hubert_code.npy"), g.randint(0, 510, size=30))
            np.save(os.path.join(pdir, f"{n}{suf}.mel.npy"), (g.randn(80, 30) * 2 - 6).astype(np.float32))
        k = npred[n]
        np.save(os.path.join(tdir, f"{n}.semantic.npy"),
                g.randint(0, 510, size=(2, k)) if mode == "covomix" else g.randint(0, 510, size=k))
    captured = {}
    real = generation.CoVoMixModel.synthesis_sample

    def spy(self, phoneme_ids, cond, mask, cond_scale, y0=None):
        y0 = torch.randn(cond.shape[0], cond.shape[1], 80, generator=torch.Generator().manual_seed(int(cond.shape[1])))
        captured[int(cond.shape[1])] = (phoneme_ids.cpu(), cond.cpu(), mask.cpu(), y0)
        return real(self, phoneme_ids, cond, mask, cond_scale, y0=y0)
    monkeypatch.setattr(generation.CoVoMixModel, "synthesis_sample", spy)
    n = generation.run(dialogue, ["--acous_ckpt", os.path.join(tmp, "acous.ckpt"), "--hifigan_ckpt",
                                  os.path.join(tmp, "voc", "g_00000001"), "--text_dir", tdir, "--prompt_dir", pdir,
                                  "--saved_dir", sdir, "--mode", mode, "--seed", "30"])
    assert n == 3 and os.path.isfile(os.path.join(sdir, "config.txt"))
    from scipy.io.wavfile import read
    folded = orc.fold_weight_norm(vsd)
    model_nfe = 32
    for T, (ids, cond, mask, y0) in captured.items():
        assert ids.dtype == torch.int64 and int(ids.max()) <= 501
        ref_mel = orc.sample(ema, ids, cond, y0, 0.7, nfe=model_nfe)             # EMA weights are what run
        for j in range(ids.shape[0]):
            valid = assembly.select_generated_frames(ref_mel[j:j + 1], mask[j])
            ref_pcm = orc.wav_to_int16(orc.hifigan_forward(folded, h, valid))
            name = [nm for nm in names if 30 + npred[nm] == T][j]
            sr, pcm = read(os.path.join(sdir, name + ".wav"))
            assert sr == 8000 and pcm.dtype == np.int16 and pcm.shape == ref_pcm.shape == (160 * (T - 30) + 32,)
            err = np.abs(pcm.astype(np.int32) - ref_pcm.astype(np.int32))
            assert err.max() <= 64 and (err > 2).mean() < 0.01, (err.max(), (err > 2).mean())
