import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import covomix_oracle as orc, covomix_amd.synthetic as syn
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' | head -5")
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.acoustic_param_shapes(), seed=0).items()}
sd["transformer.rotary_emb.inv_freq"] = torch.from_numpy(syn.rotary_inv_freq(64))
inp = syn.synthetic_inputs("vomix", 1, 1000, 400)
tm = torch.tensor(0.25)
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    with torch.inference_mode():
        orc.forward_with_cond_scale(sd, inp["y0"], tm, inp["phoneme_ids"], inp["cond"], 0.7)
        t0 = time.perf_counter()
        orc.forward_with_cond_scale(sd, inp["y0"], tm, inp["phoneme_ids"], inp["cond"], 0.7)
        print("threads", n, "cfg eval s", round(time.perf_counter() - t0, 3), flush=True)
