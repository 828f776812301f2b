#!/usr/bin/env python3
"""Llama-3-8B shape (BASELINE configs[4]: hs 128, 8 kv groups, d 4096, I 14336, V 128256, 32 layers),
hash weights: ragged batch generate, joint decode == alone, timing of one 10+10-hyp sized batch."""
import sys, time, torch
sys.path.insert(0, '.')
from dualhyp_amd import GPT, Config, GER_LORA, generate, generate_batch
from dualhyp_amd.synth import synth_state_dict, synth_prompts
dev = "cuda:0"
cfg = Config.from_name("Llama-3-8B-Instruct", **{**GER_LORA, "dropout": 0.0})
t0 = time.time()
sd = synth_state_dict(cfg, seed=1337, device=dev)
m = GPT(cfg).to(device=dev, dtype=torch.bfloat16)
m.load_state_dict(sd, strict=True); del sd
m.eval()
print(f"model built in {time.time()-t0:.1f}s, {sum(p.numel() for p in m.parameters())/1e9:.2f} B params", flush=True)
g = torch.Generator().manual_seed(5)
V = cfg.padded_vocab_size
prompts = [torch.randint(3, V, (int(n),), generator=g).to(dev) for n in torch.randint(20, 90, (40,), generator=g)]
out = generate_batch(m, prompts, 8, temperature=0.2, top_k=1, prefill_batch=16)
alone = generate(m, prompts[21], prompts[21].numel() + 8, temperature=0.2, top_k=1)
assert torch.equal(alone, out[21]), "joint decode row differs from the prompt alone"
small = generate_batch(m, prompts[16:32], 8, temperature=0.2, top_k=1)
assert all(torch.equal(a, b) for a, b in zip(small, out[16:32]))
print("ragged joint decode == alone: ok; ids", out[21][-8:].tolist(), flush=True)
B, T, G = 32, 1536, 64
corpus = [p.to(dev) for p in synth_prompts(B * 2, T, V, seed=1)]
generate_batch(m, corpus[:B], G, temperature=0.2, top_k=1)
torch.cuda.synchronize(); t0 = time.time()
o = generate_batch(m, corpus[B:], G, temperature=0.2, top_k=1)
torch.cuda.synchronize(); dt = time.time() - t0
assert all(x.numel() == T + G for x in o)
print(f"Llama-3-8B bf16, batch {B}, {T}-token prompts -> {G} tokens: {dt*1e3:.0f} ms = {B/dt:.1f} utt/s (1 GPU, one batch in flight)")

# several batches decoded jointly (chunked prefill of 32, one decode loop)
for Gn in (4, 8):
    corpus = [p.to(dev) for p in synth_prompts(B * Gn, T, V, seed=2)]
    if Gn == 4:
        generate_batch(m, corpus, G, temperature=0.2, top_k=1, prefill_batch=B)      # allocation + graph capture
    torch.cuda.synchronize(); t0 = time.time()
    o = generate_batch(m, corpus, G, temperature=0.2, top_k=1, prefill_batch=B)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"Llama-3-8B bf16, {Gn} batches of {B} decoded jointly: {dt*1e3:.0f} ms = {B*Gn/dt:.1f} utt/s (1 GPU)", flush=True)
