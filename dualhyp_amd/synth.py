"""Deterministic synthetic weights and prompts (no checkpoints or datasets exist offline).

Values come from a counter-based splitmix64 hash written with torch int64 ops only, so the
same (name, shape, seed) yields bit-identical bf16 tensors on the CPU of this container, on
the CPU of the GPU box (oracle side) and on the MI355X itself (product side).  Distributions
follow SURVEY.md §8(d): base weights std 0.02, lora_A kaiming-uniform(a=sqrt(5))
(reference ger/lora.py:149), lora_B non-zero so the LoRA branch is live, RMSNorm weight 1.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List

import torch

_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    """Python int -> two's-complement int64 value."""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


_GOLD = _s64(0x9E3779B97F4A7C15)
_MIX1 = _s64(0xBF58476D1CE4E5B9)
_MIX2 = _s64(0x94D049BB133111EB)


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    """Logical shift right on int64 (torch's >> is arithmetic)."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def hash_u24(n: int, stream: int, device="cpu", offset: int = 0) -> torch.Tensor:
    """n 24-bit integers (as int64) from splitmix64(counter) keyed by `stream`."""
    key = _s64((stream * 0xD1342543DE82EF95 + 0x2545F4914F6CDD1D) & _M64)
    z = torch.arange(offset, offset + n, dtype=torch.int64, device=device)
    z = (z + 1) * _GOLD + key
    z = (z ^ _lsr(z, 30)) * _MIX1
    z = (z ^ _lsr(z, 27)) * _MIX2
    z = z ^ _lsr(z, 31)
    return _lsr(z, 40)


def uniform(shape, bound: float, stream: int, device="cpu", dtype=torch.bfloat16,
            chunk: int = 1 << 24) -> torch.Tensor:
    """U(-bound, bound) in fp32 from the hash, then ONE cast to `dtype`."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=dtype, device=device)
    for off in range(0, n, chunk):
        m = min(chunk, n - off)
        u = hash_u24(m, stream, device, off).to(torch.float32)  # exact: < 2**24
        u = u * (2.0 / (1 << 24)) - 1.0                          # exact in fp32
        out[off:off + m] = (u * bound).to(dtype)
    return out.view(*shape)


def stream_id(seed: int, name: str) -> int:
    return (seed << 32) ^ zlib.crc32(name.encode())


def param_shapes(cfg) -> Dict[str, tuple]:
    """State-dict keys/shapes of ger.lora.GPT (probed key set)."""
    d, hs = cfg.n_embd, cfg.head_size
    qkv = (cfg.n_head + 2 * cfg.n_query_groups) * hs
    kv = d // (cfg.n_head // cfg.n_query_groups)
    V, I = cfg.padded_vocab_size, cfg.intermediate_size
    shapes: Dict[str, tuple] = {
        "lm_head.linear.weight": (V, d),
        "lm_head.adapter_bias": (V,),
        "lm_head.adapter_scale": (V,),
        "transformer.wte.weight": (V, d),
        "transformer.ln_f.weight": (d,),
    }
    en = [cfg.to_query, cfg.to_key, cfg.to_value]
    for l in range(cfg.n_layer):
        p = f"transformer.h.{l}."
        shapes[p + "norm_1.weight"] = (d,)
        shapes[p + "norm_2.weight"] = (d,)
        shapes[p + "attn.attn.linear.weight"] = (qkv, d)
        shapes[p + "attn.proj.linear.weight"] = (d, d)
        if cfg.r > 0 and l >= cfg.lora_start_layer:
            if any(en):
                shapes[p + "attn.attn.lora_A"] = (cfg.r * sum(en), d)
                shapes[p + "attn.attn.lora_B"] = (d * en[0] + kv * en[1] + kv * en[2], cfg.r)
            if cfg.to_projection:
                shapes[p + "attn.proj.lora_A"] = (cfg.r, d)
                shapes[p + "attn.proj.lora_B"] = (d, cfg.r)
        for nm, (o, i) in (("fc_1", (I, d)), ("fc_2", (I, d)), ("proj", (d, I))):
            shapes[p + f"mlp.{nm}.linear.weight"] = (o, i)
            if cfg.r > 0 and cfg.to_mlp and l >= cfg.lora_start_layer:
                shapes[p + f"mlp.{nm}.lora_A"] = (cfg.r, i)
                shapes[p + f"mlp.{nm}.lora_B"] = (o, cfg.r)
    return shapes


TIE_MUL, TIE_ADD = 7919, 12345     # successor permutation of the tied head: sigma(v) = (TIE_MUL * v + TIE_ADD) mod V


def tie_successor(token: int, vocab: int) -> int:
    """The token a `head_tie` model emits after `token` (the v with sigma(v) == token)."""
    return ((token - TIE_ADD) * pow(TIE_MUL, -1, vocab)) % vocab


def synth_state_dict(cfg, seed: int = 1337, device="cpu", dtype=torch.bfloat16,
                     norm_jitter: float = 0.0, weight_scale: float = 1.0, head_peak: float = 0.0,
                     embed_scale: float = 1.0, head_tie: float = 0.0) -> Dict[str, torch.Tensor]:
    """Full state dict with the reference's key names.  `norm_jitter` > 0 perturbs the RMSNorm
    weights away from 1 (used by parity fixtures so the weight multiply is exercised);
    `weight_scale` widens the base/lora_B weights (tiny shapes need it for non-trivial logits);
    `head_peak` > 0 makes `lm_head.adapter_scale` (ger/lora.py:67-71) heavy-tailed, exp(head_peak * E)
    with E ~ Exp(1): i.i.d. Gaussian logits over 32000 tokens put the runner-up within 4 bf16 ulps of
    the arg-max on a quarter of the positions, which makes greedy-id parity untestable; a trained
    head is peaked, and this reproduces that with the reference's own parameter (0.5: ~2.5 % by the ulp
    measure — but a top logit is then scale x (a SMALL dot product), whose bf16 noise is scale x the noise of
    every dot product, so those margins are only ~1 sigma of what separates two bf16 implementations).
    `embed_scale` / `head_tie` give margins that ARE robust: the token embedding is scaled (50: it stays visible in
    the residual stream after 22 random layers) and lm_head[v] += head_tie * wte_unscaled[sigma(v)], so the logit of
    the successor sigma^-1(last token) stands ~10 logit-std above the rest (>= 20 sigma of bf16 noise, >= 38 ulps on
    every step at TinyLlama size); every other logit stays a context-dependent random projection."""
    sd: Dict[str, torch.Tensor] = {}
    a = 0.02 * math.sqrt(3.0) * weight_scale
    for name, shape in param_shapes(cfg).items():
        st = stream_id(seed, name)
        if name == "transformer.wte.weight" and embed_scale != 1.0:
            t = uniform(shape, a * embed_scale, st, device, dtype)
        elif name == "lm_head.linear.weight" and head_tie != 0.0:
            V = shape[0]
            assert math.gcd(TIE_MUL, V) == 1, "the successor map must be a permutation of the vocabulary"
            sigma = (TIE_MUL * torch.arange(V, dtype=torch.int64, device=device) + TIE_ADD) % V
            raw = uniform(shape, a, stream_id(seed, "transformer.wte.weight"), device, torch.float32)   # embedding before embed_scale
            t = uniform(shape, a, st, device, torch.float32)
            for r0 in range(0, V, 8192):                  # fp32 elementwise only: bit-identical on CPU and GPU
                t[r0:r0 + 8192] += head_tie * raw[sigma[r0:r0 + 8192]]
            t = t.to(dtype)
            del raw
        elif name.endswith("adapter_bias"):
            t = torch.zeros(shape, dtype=dtype, device=device)
        elif name.endswith("adapter_scale"):
            t = torch.ones(shape, dtype=dtype, device=device)
            if head_peak > 0:
                u = uniform(shape, 1.0, st, device, torch.float32).abs()       # U[0, 1)
                # -log(1-u) through a 4096-entry table indexed by the hash bits: no libm call whose
                # last bit could differ between the CPU and the GPU
                q = torch.clamp((u * 4096.0).to(torch.int64), 0, 4095)
                table = torch.tensor([math.exp(head_peak * -math.log(1.0 - (i + 0.5) / 4096.0)) for i in range(4096)],
                                     dtype=torch.float32).to(dtype).to(device)
                t = table[q]
        elif "norm" in name or "ln_f" in name:
            t = torch.ones(shape, dtype=dtype, device=device)
            if norm_jitter > 0:
                t = (1.0 + uniform(shape, norm_jitter, st, device, torch.float32)).to(dtype)
        elif name.endswith("lora_A"):
            t = uniform(shape, 1.0 / math.sqrt(shape[1]), st, device, dtype)
        else:
            t = uniform(shape, a, st, device, dtype)
        sd[name] = t
    return sd


def synth_prompts(n: int, length: int, vocab: int, seed: int = 1337, ragged: bool = False,
                  lo: int = 0, hi: int = 0) -> List[torch.Tensor]:
    """Token-id level prompts: BOS(1) + ids uniform in [3, vocab).  With
    `ragged`, lengths are spread over [lo, hi]."""
    out = []
    for i in range(n):
        T = length
        if ragged:
            T = lo + int(hash_u24(1, stream_id(seed, f"len{i}"))[0]) % (hi - lo + 1)
        body = hash_u24(T - 1, stream_id(seed, f"prompt{i}")) % (vocab - 3) + 3
        out.append(torch.cat([torch.ones(1, dtype=torch.int64), body]))
    return out
