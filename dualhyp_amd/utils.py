"""Loss / precision helpers with the reference's names (ger/utils.py)."""
from __future__ import annotations

from typing import List, Optional, Union

import torch
import torch.nn.functional as F

from .config import find_multiple  # noqa: F401  (re-export, ger/utils.py:29-33)


def num_parameters(module: torch.nn.Module, requires_grad: Optional[bool] = None) -> int:
    return sum(p.numel() for p in module.parameters() if requires_grad is None or p.requires_grad == requires_grad)


def chunked_cross_entropy(logits: Union[torch.Tensor, List[torch.Tensor]], targets: torch.Tensor,
                          chunk_size: int = 128) -> torch.Tensor:
    """ger/utils.py:424-463.  ignore_index = -1.  Chunked variants average the per-position
    losses over ALL positions (ignored ones count as 0, quirk Q5); chunk_size == 0 averages over
    the valid positions only."""
    if isinstance(logits, list):
        if chunk_size == 0:
            lg = torch.cat(logits, dim=1)
            return F.cross_entropy(lg.reshape(-1, lg.size(-1)), targets.reshape(-1), ignore_index=-1)
        width = logits[0].size(1)
        per = [F.cross_entropy(lc.reshape(-1, lc.size(-1)), tc.reshape(-1), ignore_index=-1, reduction="none")
               for lc, tc in zip(logits, targets.split(width, dim=1))]
        return torch.cat(per).mean()
    lg, tg = logits.reshape(-1, logits.size(-1)), targets.reshape(-1)
    if chunk_size == 0:
        return F.cross_entropy(lg, tg, ignore_index=-1)
    per = [F.cross_entropy(lc, tc, ignore_index=-1, reduction="none")
           for lc, tc in zip(lg.split(chunk_size), tg.split(chunk_size))]
    return torch.cat(per).mean()


def get_default_supported_precision(training: bool) -> str:
    """ger/utils.py:475-489: bf16-mixed for training, bf16-true for inference (MI355X has bf16)."""
    return "bf16-mixed" if training else "bf16-true"
