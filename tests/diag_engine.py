import sys, math, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from dualhyp_amd import ops, GPT, Config
from dualhyp_amd.synth import synth_state_dict
from oracle import ger_oracle as O
t, meta = load_golden("tiny_r4")
D = "cuda:0"
def cmp(a, b, what):
    a = a.float().cpu().reshape(-1); b = b.float().reshape(-1)
    rms = b.pow(2).mean().sqrt(); d = (a-b).abs()
    print(f"{what:40s} exact {(a==b).float().mean().item():7.2%}  relRMS {(d.pow(2).mean().sqrt()/rms).item():.2e}")
idx = t["idx0"]; T = idx.numel()
for nl in (1, 2):
    for lora in (True, False):
        c = dict(meta["config"]); c["n_layer"] = nl
        if not lora: c["r"] = 0
        cfg = Config(**c)
        sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], weight_scale=meta["weight_scale"])
        m = GPT(cfg).to(device=D, dtype=torch.bfloat16); m.load_state_dict({k: v.to(D) for k, v in sd.items()}); m.eval()
        om = O.OracleGPT(cfg, sd)
        with torch.no_grad():
            a = m(idx.view(1, -1).to(D))
            hid = m._engine.read(3, 0, (T, cfg.n_embd))
            # oracle residual stream before ln_f
            x = torch.nn.functional.embedding(idx.view(1, -1), sd["transformer.wte.weight"])
            cos, sin = O.build_rope_cache(cfg.block_size, cfg.rope_n_elem)
            for l in range(nl):
                x = om.block(l, x, cos[:T], sin[:T])
                if l == 0: x0 = x
            cmp(hid, x, f"n_layer={nl} lora={lora}: residual x")
            cmp(a, om(idx.view(1, -1)), f"n_layer={nl} lora={lora}: logits")
