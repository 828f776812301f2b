import sys, math, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from dualhyp_amd import ops, GPT, Config
from dualhyp_amd.synth import synth_state_dict
t, meta = load_golden("tiny_r4")
D = "cuda:0"
def cmp(a, b, what):
    a = a.float().reshape(-1); b = b.float().reshape(-1)
    print(f"{what:40s} exact {(a==b).float().mean().item():7.2%}  maxabs {(a-b).abs().max().item():.3e}")
idx = t["idx0"].to(D); T = idx.numel()
c = dict(meta["config"]); c["n_layer"] = 1; c["r"] = 0
cfg = Config(**c)
sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], weight_scale=meta["weight_scale"], device=D)
m = GPT(cfg).to(device=D, dtype=torch.bfloat16); m.load_state_dict(sd); m.eval()
H, G, hs, d = cfg.n_head, cfg.n_query_groups, cfg.head_size, cfg.n_embd
i32 = torch.int32
for vec in (32, 0):
    m.cpu_rsqrt_vec_width = vec
    with torch.no_grad():
        lg = m(idx.view(1, -1))
    eng = m._engine
    x_e = eng.read(3, 0, (T, d))
    kc_e = eng.read(1, 0, (1, G, 128, hs)); vt_e = eng.read(2, 0, (1, G, hs, 128))
    tail = torch.ones(T, dtype=torch.uint8, device=D) if vec else None
    p = "transformer.h.0."
    x = ops.embed(idx, sd["transformer.wte.weight"])
    n1 = ops.rmsnorm(x, sd[p+"norm_1.weight"], cfg.norm_eps, row_tail=tail)
    qkv = ops.linear(n1, sd[p+"attn.attn.linear.weight"])
    cos, sin = m.rope_cache
    kc = torch.zeros((1, G, 128, hs), dtype=torch.bfloat16, device=D); vt = torch.zeros((1, G, hs, 128), dtype=torch.bfloat16, device=D)
    q = ops.qkv_rope_cache(qkv, cos, sin, torch.zeros(T, dtype=i32, device=D), torch.arange(T, dtype=i32, device=D), kc, vt, H, G)
    cmp(kc_e, kc, f"vec={vec} k cache engine vs ops")
    cmp(vt_e, vt, f"vec={vec} vT cache engine vs ops")
    y = ops.attn_prefill(q, kc, vt, torch.zeros(1, dtype=i32, device=D), torch.zeros(1, dtype=i32, device=D), torch.tensor([T], dtype=i32, device=D), torch.zeros(1, dtype=i32, device=D), T)
    x1 = ops.linear(y, sd[p+"attn.proj.linear.weight"], resid=x)
    n2 = ops.rmsnorm(x1, sd[p+"norm_2.weight"], cfg.norm_eps, row_tail=tail)
    act = ops.linear(n2, sd[p+"mlp.fc_1.linear.weight"], epilogue=ops.EPI_SWIGLU, w2=sd[p+"mlp.fc_2.linear.weight"])
    x2 = ops.linear(act, sd[p+"mlp.proj.linear.weight"], resid=x1)
    cmp(x_e, x2, f"vec={vec} residual engine vs ops chain")
    xf = ops.rmsnorm(x2, sd["transformer.ln_f.weight"], cfg.norm_eps, row_tail=tail)
    l2 = ops.linear(xf, sd["lm_head.linear.weight"], epilogue=ops.EPI_ADAPTER, scale=sd["lm_head.adapter_scale"], bias=sd["lm_head.adapter_bias"])
    cmp(lg, l2, f"vec={vec} logits engine vs ops chain")
