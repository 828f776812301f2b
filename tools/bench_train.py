#!/usr/bin/env python3
"""LoRA fine-tune micro-step timing at the TinyLlama-1.1B shape (BASELINE configs[2] unit of work):
T = 560 tokens (512 masked prompt + 47 response + EOS), micro-batch 1, fwd + bwd, chunked CE."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from dualhyp_amd import GPT, Config, GER_LORA
from dualhyp_amd.synth import synth_state_dict, synth_prompts
from dualhyp_amd.train import prepare_for_training
from dualhyp_amd.finetune import micro_loss, FlatGradBucket
D = "cuda:0"
cfg = Config.from_name("tiny-llama-1.1b-chat", **GER_LORA)
m = GPT(cfg).to(device=D, dtype=torch.bfloat16)
m.load_state_dict(synth_state_dict(cfg, seed=1337, device=D))
m.train()
params = prepare_for_training(m)
bucket = FlatGradBucket(params)
T = 560
ids = synth_prompts(1, T, cfg.padded_vocab_size, seed=3)[0].view(1, -1).to(D)
labels = ids.clone(); labels[:, :512] = -1
def step():
    loss = micro_loss(m, ids, labels, 128)
    (loss / 32).backward()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n): l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = 2.43e12
print(f"micro-step (T={T}, B=1): {dt*1e3:.1f} ms  ~{flops/dt/1e12:.0f} TFLOP/s algorithmic  loss {l.item():.3f}  grad-bucket {bucket.flat.numel()} fp32")
print(f"peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")

# ---- the same micro-step captured once in a hipGraph and replayed (dualhyp_amd.train.GraphedTrainStep)
from dualhyp_amd.train import GraphedTrainStep
bucket.zero()
for _ in range(2): step()
ref = bucket.flat.clone(); ref_loss = l.item()
bucket.zero()
gs = GraphedTrainStep(m, bucket)
m.eval_dropout = None
lg = None
for _ in range(2): lg = gs(ids, labels, 1.0 / 32)
torch.cuda.synchronize()
print(f"graph vs eager: loss {lg.item():.4f} vs {ref_loss:.4f}; grad relerr {((bucket.flat - ref).norm() / ref.norm()).item():.3e} (dropout masks differ between the runs)")
t0 = time.perf_counter()
for _ in range(n): lg = gs(ids, labels, 1.0 / 32)
torch.cuda.synchronize()
dtg = (time.perf_counter() - t0) / n
print(f"graphed micro-step: {dtg*1e3:.1f} ms  ~{flops/dtg/1e12:.0f} TFLOP/s algorithmic")
