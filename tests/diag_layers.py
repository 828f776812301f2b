"""Layer-by-layer HIP (ops) vs oracle on the GPU box: isolates the op that deviates."""
import sys, math, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from dualhyp_amd import ops, GPT, Config
from dualhyp_amd.synth import synth_state_dict
from oracle import ger_oracle as O
import torch.nn.functional as F
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_r4"
t, meta = load_golden(name)
cfg = Config(**meta["config"])
sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], weight_scale=meta["weight_scale"])
D = "cuda:0"
g = {k: v.to(D) for k, v in sd.items()}
idx = t["idx0"]; T = idx.numel()
def cmp(a, b, what):
    a = a.float().cpu().reshape(-1); b = b.float().reshape(-1)
    rms = b.pow(2).mean().sqrt()
    d = (a-b).abs()
    print(f"{what:28s} exact {(a==b).float().mean().item():7.2%}  relRMS {(d.pow(2).mean().sqrt()/rms).item():.2e}  max|d|/rms {(d.max()/rms).item():.2e}")
cos, sin = O.build_rope_cache(cfg.block_size, cfg.rope_n_elem)
H, G, hs, d = cfg.n_head, cfg.n_query_groups, cfg.head_size, cfg.n_embd
qpk = H // G; kvw = d // qpk; s = cfg.alpha / cfg.r
tail = (torch.arange(T) >= T // 32 * 32).to(torch.uint8).to(D)
x_ref = F.embedding(idx.view(1, -1), sd["transformer.wte.weight"])     # (1,T,d)
x_hip = ops.embed(idx.to(D), g["transformer.wte.weight"])
cmp(x_hip, x_ref, "embed")
from dualhyp_amd.gpt import _pad_rank
for l in range(cfg.n_layer):
    p = f"transformer.h.{l}."
    for mode in ("teacher", ):
        xin = x_ref[0].to(D)
        n1 = ops.rmsnorm(xin, g[p+"norm_1.weight"], cfg.norm_eps, row_tail=tail)
        n1_ref = O.rmsnorm(x_ref, sd[p+"norm_1.weight"], cfg.norm_eps)
        cmp(n1, n1_ref, f"L{l} norm_1")
        A, B = sd[p+"attn.attn.lora_A"], sd[p+"attn.attn.lora_B"]
        A48 = torch.zeros(48, d, dtype=torch.bfloat16)
        for seg in range(3): A48[16*seg:16*seg+cfg.r] = A[seg*cfg.r:(seg+1)*cfg.r]
        B16 = _pad_rank(B, cfg.r, 1)
        n1r = n1_ref[0].to(D)
        xa = ops.linear(n1r, A48.to(D))
        qkv = ops.linear(n1r, g[p+"attn.attn.linear.weight"], epilogue=ops.EPI_LORA, xa=xa, lora_b=B16.to(D), lora_scale=s, splits=(d, d+kvw))
        qkv_ref = O.lora_qkv_linear(n1_ref, sd[p+"attn.attn.linear.weight"], A, B, s, (d, kvw, kvw))
        cmp(qkv, qkv_ref, f"L{l} qkv+lora")
        # attention
        kc = torch.zeros((1, G, 128, hs), dtype=torch.bfloat16, device=D); vt = torch.zeros((1, G, hs, 128), dtype=torch.bfloat16, device=D)
        i32 = torch.int32
        q = ops.qkv_rope_cache(qkv_ref[0].to(D), cos.to(D), sin.to(D), torch.zeros(T, dtype=i32, device=D), torch.arange(T, dtype=i32, device=D), kc, vt, H, G)
        y = ops.attn_prefill(q, kc, vt, torch.zeros(1, dtype=i32, device=D), torch.zeros(1, dtype=i32, device=D), torch.tensor([T], dtype=i32, device=D), torch.zeros(1, dtype=i32, device=D), T)
        v5 = qkv_ref.view(1, T, G, qpk+2, hs).permute(0, 2, 3, 1, 4)
        qq, kk, vv = v5.split((qpk, 1, 1), dim=2)
        kk = kk.expand(1, G, qpk, T, hs); vv = vv.expand(1, G, qpk, T, hs)
        qq = qq.reshape(1, -1, T, hs); kk = kk.reshape(1, -1, T, hs); vv = vv.reshape(1, -1, T, hs)
        qq = O.apply_rope(qq, cos[:T], sin[:T]); kk = O.apply_rope(kk, cos[:T], sin[:T])
        y_ref = F.scaled_dot_product_attention(qq, kk, vv, is_causal=True, scale=1/math.sqrt(hs)).transpose(1, 2).reshape(1, T, d)
        y_f32 = F.scaled_dot_product_attention(qq.float(), kk.float(), vv.float(), is_causal=True, scale=1/math.sqrt(hs)).transpose(1, 2).reshape(1, T, d)
        cmp(y, y_ref, f"L{l} attn (vs cpu sdpa)")
        cmp(y, y_f32, f"L{l} attn hip vs fp32")
        cmp(y_ref.to(D), y_f32, f"L{l} attn cpu-bf16 vs fp32")
        # masked-cache variant the reference uses in the cache path
        Ap, Bp = sd[p+"attn.proj.lora_A"], sd[p+"attn.proj.lora_B"]
        yr = y_ref[0].to(D)
        xa = ops.linear(yr, _pad_rank(Ap, cfg.r, 0).to(D))
        x1 = ops.linear(yr, g[p+"attn.proj.linear.weight"], epilogue=ops.EPI_LORA, xa=xa, lora_b=_pad_rank(Bp, cfg.r, 1).to(D), lora_scale=s, resid=xin)
        x1_ref = x_ref + O.lora_linear(y_ref, sd[p+"attn.proj.linear.weight"], Ap, Bp, s)
        cmp(x1, x1_ref, f"L{l} proj+lora+resid")
        n2 = ops.rmsnorm(x1_ref[0].to(D), g[p+"norm_2.weight"], cfg.norm_eps, row_tail=tail)
        n2_ref = O.rmsnorm(x1_ref, sd[p+"norm_2.weight"], cfg.norm_eps)
        cmp(n2, n2_ref, f"L{l} norm_2")
        act = ops.linear(n2_ref[0].to(D), g[p+"mlp.fc_1.linear.weight"], epilogue=ops.EPI_SWIGLU, w2=g[p+"mlp.fc_2.linear.weight"])
        act_ref = F.silu(F.linear(n2_ref, sd[p+"mlp.fc_1.linear.weight"])) * F.linear(n2_ref, sd[p+"mlp.fc_2.linear.weight"])
        cmp(act, act_ref, f"L{l} swiglu")
        x2 = ops.linear(act_ref[0].to(D), g[p+"mlp.proj.linear.weight"], resid=x1_ref[0].to(D))
        x2_ref = x1_ref + F.linear(act_ref, sd[p+"mlp.proj.linear.weight"])
        cmp(x2, x2_ref, f"L{l} mlp proj+resid")
        x_ref = x2_ref
xf = ops.rmsnorm(x_ref[0].to(D), g["transformer.ln_f.weight"], cfg.norm_eps, row_tail=tail)
xf_ref = O.rmsnorm(x_ref, sd["transformer.ln_f.weight"], cfg.norm_eps)
cmp(xf, xf_ref, "ln_f")
lg = ops.linear(xf_ref[0].to(D), g["lm_head.linear.weight"], epilogue=ops.EPI_ADAPTER, scale=g["lm_head.adapter_scale"], bias=g["lm_head.adapter_bias"])
lg_ref = sd["lm_head.adapter_scale"] * (F.linear(xf_ref, sd["lm_head.linear.weight"]) + sd["lm_head.adapter_bias"])
cmp(lg, lg_ref, "lm_head")
# whole model through the engine vs oracle
m = GPT(cfg).to(device=D, dtype=torch.bfloat16); m.load_state_dict({k: v for k, v in g.items()}); m.eval()
om = O.OracleGPT(cfg, sd)
with torch.no_grad():
    a = m(idx.view(1, -1).to(D)); b = om(idx.view(1, -1))
    cmp(a, b, "engine logits no-cache B=1")
    a = m(idx.view(1, -1).to(D), torch.arange(T, device=D)); b2 = om(idx.view(1, -1), torch.arange(T))
    cmp(a, b2, "engine logits cache B=1")
    cmp(b.to(D), b2, "oracle nocache vs cache")
    hid = m._engine.read(3, 0, (T, d))
    cmp(hid, x_ref, "engine final residual x vs teacher-forced oracle x")
