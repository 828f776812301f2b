"""Prototype: N batch-32 decode/prefill chains in flight at once on separate streams (one engine each)."""
import sys, time, threading, torch
sys.path.insert(0, '.')
from dualhyp_amd import GPT, Config, GER_LORA, generate_batch
from dualhyp_amd.synth import synth_state_dict, synth_prompts
D = "cuda:0"
cfg = Config.from_name("tiny-llama-1.1b-chat", **{**GER_LORA, "dropout": 0.0})
sd = synth_state_dict(cfg, seed=1337, device=D)
def make():
    m = GPT(cfg).to(device=D, dtype=torch.bfloat16)
    m.load_state_dict(sd, strict=True, assign=True)      # share the weight storage
    m.eval(); m.set_capacity(32, 576, 32 * 512)
    return m
B, steps = 32, 6
corpus = [p.to(D) for p in synth_prompts(B * (steps + 2), 512, cfg.padded_vocab_size, seed=1337)]
for E in (1, 2, 3):
    models = [make() for _ in range(E)]
    streams = [torch.cuda.Stream() for _ in range(E)]
    def work(e, idxs, out):
        with torch.cuda.stream(streams[e]):
            for i in idxs:
                out[i] = generate_batch(models[e], corpus[i * B:(i + 1) * B], 64, temperature=0.2, top_k=1)
    out = {}
    work(0, [steps], out)
    for e in range(1, E): work(e, [steps + 1], out)      # warm every engine
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(e, list(range(e, steps, E)), out)) for e in range(E)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"engines {E}: {steps} steps in {dt*1e3:.0f} ms -> {dt/steps*1e3:.1f} ms/step, {B*steps/dt:.0f} utt/s")
    del models
