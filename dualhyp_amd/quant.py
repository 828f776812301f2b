"""fp8 (OCP e4m3fn) serving path — BASELINE config 5 / SURVEY.md §8f-4.

`quantize_model_fp8(model)` folds every LoRA update into its base weight exactly as the reference's
`merge_lora_weights` does (ger/lora.py:707-711 -> `:152-157`, `:349-365`) and then replaces each dense weight (the
seven matrices of every block and lm_head) by e4m3 rows with one fp32 scale per output channel:

    amax = max|W[n, :]| (>= 1e-12);  Wq[n, :] = fp8_rne(W[n, :] * inv), inv = 448 / amax (one IEEE division);  scale[n] = amax * fp32(1/448)

Activations are quantised the same way per token at run time by the kernels (csrc/fp8.hip), and every product runs
on the block-scaled fp8 MFMA.  The embedding table, the norm weights and the lm_head adapter vectors stay bf16.
The model is inference-only afterwards; the bf16 weights are released.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

FP8_MAX = 448.0


def quantize_rows_fp8(w: torch.Tensor, chunk_rows: int = 8192) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (uint8 [N, K] holding e4m3fn bit patterns, fp32 [N] scales); fp32 arithmetic, round to nearest even."""
    N = w.size(0)
    q = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
    scale = torch.empty(N, dtype=torch.float32, device=w.device)
    for r0 in range(0, N, chunk_rows):
        x = w[r0:r0 + chunk_rows].float()
        amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12)
        inv = torch.full_like(amax, FP8_MAX) / amax        # tensor / tensor: one correctly rounded division
        q[r0:r0 + chunk_rows] = (x * inv).to(torch.float8_e4m3fn).view(torch.uint8)
        scale[r0:r0 + chunk_rows] = (amax * torch.tensor(1.0 / FP8_MAX, dtype=torch.float32, device=w.device)).view(-1)
    return q, scale


def dequantize_rows_fp8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float() * scale.view(-1, 1)


def quantize_model_fp8(model) -> None:
    """Serving-only, one way: merge LoRA, replace every dense weight by e4m3 rows + channel scales (non-persistent
    buffers).  The bf16 weights are gone afterwards, so `GPT.load_state_dict` and `save_checkpoint` refuse a quantised
    model (quantise a freshly loaded model instead)."""
    from .gpt import GPT, _FrozenLinear, merge_lora_weights
    assert isinstance(model, GPT)
    if getattr(model, "fp8", False):
        return
    merge_lora_weights(model)
    for m in model.modules():
        if isinstance(m, _FrozenLinear):
            q, s = quantize_rows_fp8(m.weight.data)
            m.register_buffer("weight_fp8", q, persistent=False)
            m.register_buffer("weight_scale", s, persistent=False)
            m.weight = nn.Parameter(torch.empty(0, dtype=m.weight.dtype, device=m.weight.device), requires_grad=False)
    for p in model.parameters():
        p.requires_grad_(False)
    model.fp8 = True
    model.refresh_engine()
