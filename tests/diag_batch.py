import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from dualhyp_amd import GPT, Config, generate, generate_batch
from dualhyp_amd.synth import synth_state_dict
D = "cuda:0"
t, meta = load_golden("tiny_r4")
cfg = Config(**meta["config"])
sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], weight_scale=meta["weight_scale"], device=D)
m = GPT(cfg).to(device=D, dtype=torch.bfloat16); m.load_state_dict(sd); m.eval()
p0, p1 = t["idx0"][:17].to(D), t["idx1"].to(D)
m.set_capacity(4, 128, 512)
a0 = generate_batch(m, [p0], 6, temperature=0.2, top_k=1)[0].cpu()
a1 = generate_batch(m, [p1], 6, temperature=0.2, top_k=1)[0].cpu()
b = [o.cpu() for o in generate_batch(m, [p0, p1], 6, temperature=0.2, top_k=1)]
c = [o.cpu() for o in generate_batch(m, [p1, p0], 6, temperature=0.2, top_k=1)]
dd = [o.cpu() for o in generate_batch(m, [p0, p0, p1, p1], 6, temperature=0.2, top_k=1)]
print("alone0", a0[-6:].tolist()); print("alone1", a1[-6:].tolist())
print("both  ", b[0][-6:].tolist(), b[1][-6:].tolist())
print("swap  ", c[1][-6:].tolist(), c[0][-6:].tolist())
print("quad  ", [x[-6:].tolist() for x in dd])
# prefill logits batch invariance
eng = m.engine()
_, l1 = eng.forward(p0, [17], [0], False, True)
_, l2 = eng.forward(torch.cat([p0, p1]), [17, 24], [0, 0], False, True)
_, l3 = eng.forward(torch.cat([p1, p0]), [24, 17], [0, 0], False, True)
print("prefill last logits p0 alone vs packed first:", (l1[0] != l2[0]).sum().item(), "vs packed second:", (l1[0] != l3[1]).sum().item())
