"""Loss / precision helpers with the reference's names (ger/utils.py)."""
from __future__ import annotations

from typing import List, Optional, Union

import torch
import torch.nn.functional as F

from .config import find_multiple  # noqa: F401  (re-export, ger/utils.py:29-33)


def num_parameters(module: torch.nn.Module, requires_grad: Optional[bool] = None) -> int:
    return sum(p.numel() for p in module.parameters() if requires_grad is None or p.requires_grad == requires_grad)


class _RowCrossEntropy(torch.autograd.Function):
    """Per-row CE (ignore_index -1 -> 0) on the GPU: dualhyp_amd/csrc/loss.hip, fp32 log-softmax of the
    bf16 (or fp32) logits as F.cross_entropy computes it under the reference's autocast."""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        from . import ops
        loss, lse = ops.cross_entropy_fwd(logits, targets)
        ctx.save_for_backward(logits, targets, lse)
        return loss

    @staticmethod
    def backward(ctx, grad_loss: torch.Tensor):
        from . import ops
        logits, targets, lse = ctx.saved_tensors
        return ops.cross_entropy_bwd(logits, targets, lse, grad_loss.contiguous().float()), None


def _row_losses(lg: torch.Tensor, tg: torch.Tensor) -> torch.Tensor:
    """fp32 loss per row, 0 where the target is -1."""
    if lg.is_cuda:
        return _RowCrossEntropy.apply(lg, tg)
    return F.cross_entropy(lg, tg, ignore_index=-1, reduction="none")   # host tensors: test / tooling path


def chunked_cross_entropy(logits: Union[torch.Tensor, List[torch.Tensor]], targets: torch.Tensor,
                          chunk_size: int = 128) -> torch.Tensor:
    """ger/utils.py:424-463.  ignore_index = -1.  Chunked variants average the per-position
    losses over ALL positions (ignored ones count as 0, quirk Q5); chunk_size == 0 averages over
    the valid positions only.  Chunking itself only bounded the reference's peak memory: the
    per-row losses are the same numbers whichever way the rows are grouped."""
    if isinstance(logits, list):
        logits = torch.cat(logits, dim=1)
    lg, tg = logits.reshape(-1, logits.size(-1)), targets.reshape(-1)
    if chunk_size == 0 and not lg.is_cuda:
        return F.cross_entropy(lg, tg, ignore_index=-1)
    per = _row_losses(lg, tg)
    if chunk_size == 0:
        out = per.sum() / (tg != -1).sum()      # 0/0 = nan when every position is ignored, as torch
    else:
        out = per.mean()
    return out.to(logits.dtype) if not lg.is_cuda else out


def get_default_supported_precision(training: bool) -> str:
    """ger/utils.py:475-489: bf16-mixed for training, bf16-true for inference (MI355X has bf16)."""
    return "bf16-mixed" if training else "bf16-true"
