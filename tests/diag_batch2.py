import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from dualhyp_amd import GPT, Config, generate_batch, ops
from dualhyp_amd.synth import synth_state_dict
D = "cuda:0"
t, meta = load_golden("tiny_r4")
cfg = Config(**meta["config"])
sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], weight_scale=meta["weight_scale"], device=D)
m = GPT(cfg).to(device=D, dtype=torch.bfloat16); m.load_state_dict(sd); m.eval()
p0, p1 = t["idx0"][:17].to(D), t["idx1"].to(D)
m.set_capacity(4, 128, 512)
eng = m.engine()
eng.set_rsqrt_emulation(32, False)
G, hs = cfg.n_query_groups, cfg.head_size
_, la = eng.forward(p1, [24], [0], False, True)
ka = [eng.read(1, l, (4, G, 128, hs))[0].clone() for l in range(2)]
_, lb = eng.forward(torch.cat([p0, p1]), [17, 24], [0, 0], False, True)
kb = [eng.read(1, l, (4, G, 128, hs))[1].clone() for l in range(2)]
print("p1 prefill logits alone vs packed:", (la[0] != lb[1]).sum().item(), "k cache L0/L1 diffs:", [(a[:, :24] != b[:, :24]).sum().item() for a, b in zip(ka, kb)])
# step-by-step decode, alone vs packed, compare per-step tokens
def run(prompts, which):
    out, st = generate_batch(m, prompts, 6, temperature=0.2, top_k=1, return_state=True)
    return out[which].cpu()[-6:].tolist()
print("alone", run([p1], 0), "packed", run([p0, p1], 1), "packed-first", run([p1, p0], 0), "dup", run([p1, p1], 1))
