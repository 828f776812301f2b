#!/usr/bin/env python3
"""One batch of 32 in flight (BASELINE config 2 read literally): wall time per batch against the GPU-event times of its prefill and
decode phases — what the host adds.  GPU box."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import GPT, Config, GER_LORA
from dualhyp_amd.generate import generate_batch
from dualhyp_amd.synth import synth_state_dict, synth_prompts
dev = torch.device("cuda", 0)
cfg = Config.from_name("tiny-llama-1.1b-chat", **{**GER_LORA, "dropout": 0.0})
model = GPT(cfg).to(device=dev, dtype=torch.bfloat16)
model.load_state_dict(synth_state_dict(cfg, seed=1337, device=dev, embed_scale=50.0, head_tie=1.0), strict=True)
model.eval()
kw = dict(temperature=0.2, top_k=1, eos_id=None)
batches = [[p.to(dev) for p in synth_prompts(32, 512, cfg.padded_vocab_size, seed=s)] for s in range(8)]
model.set_capacity(32, 576, 32 * 512)
for b in batches[:2]:
    generate_batch(model, b, 64, prefill_batch=32, **kw)
torch.cuda.synchronize()
tm = {}
t0 = time.perf_counter()
marks = []
for b in batches[2:]:
    t1 = time.perf_counter()
    generate_batch(model, b, 64, prefill_batch=32, timing=tm, **kw)
    marks.append(time.perf_counter() - t1)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
n = len(batches) - 2
print(f"wall {wall / n * 1e3:.2f} ms per batch = {32 * n / wall:.1f} utt/s; GPU events: prefill {tm['prefill_ms'] / n:.2f} ms + decode {tm['decode_ms'] / n:.2f} ms "
      f"({tm['decode_ms'] / tm['decode_steps']:.3f} ms per step) = {(tm['prefill_ms'] + tm['decode_ms']) / n:.2f}; per-call wall {[round(m * 1e3, 1) for m in marks]}")
