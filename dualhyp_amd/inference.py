"""Sharded GER / DualHyp inference + WER — the MI355X counterpart of inference/ger.py:30-124.

The reference decodes the test set one utterance at a time on every rank (no sharding, quirk Q4).
Here each rank takes a strided shard of the utterances, decodes them in batches through
`generate_batch` (one packed prefill + hipGraph decode per batch) and the ranks all-reduce four
integers (edit errors, reference words, exact matches, count) for the corpus WER; predictions are
gathered to rank 0.  No collective runs inside the decode loop."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from .wer import post_normalize, wer_counts


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Strided shard: rank r takes r, r+world, ... (balanced to within one utterance)."""
    return list(range(rank, n, world))


def extract_answer(decoded_full: str, decoded_prompt: str) -> str:
    """inference/ger.py:84-86: strip the prompt text, keep the first line."""
    return decoded_full[len(decoded_prompt):].split("\n")[0].strip()


def _reduce_counts(c: Dict[str, int], device) -> Dict[str, int]:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return c
    keys = sorted(c)
    t = torch.tensor([c[k] for k in keys], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return dict(zip(keys, t.tolist()))


def _gather(obj: Any) -> List[Any]:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out: List[Any] = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def run_inference(generate_fn: Callable[[List[torch.Tensor]], List[torch.Tensor]], examples: Sequence[Dict[str, Any]],
                  decode: Callable[[torch.Tensor], str], *, batch_size: int = 32, rank: int = 0, world: int = 1,
                  device="cpu") -> Dict[str, Any]:
    """examples[i] needs 'input_ids_no_response' (1-D ids) and 'ground_truth'.  `generate_fn` maps a
    list of prompts to a list of prompt+continuation id tensors (dualhyp_amd.generate_batch bound to a
    model; a stub in the CPU tests).  Returns corpus metrics on every rank and predictions on rank 0."""
    mine = shard_indices(len(examples), rank, world)
    preds: Dict[int, Dict[str, str]] = {}
    for b in range(0, len(mine), batch_size):
        idxs = mine[b:b + batch_size]
        prompts = [examples[i]["input_ids_no_response"] for i in idxs]
        outs = generate_fn(prompts)
        for i, p, o in zip(idxs, prompts, outs):
            preds[i] = {"inference": extract_answer(decode(o), decode(p)),
                        "ground_truth": examples[i]["ground_truth"].strip()}
    order = sorted(preds)
    pr = [preds[i]["inference"] for i in order]
    gt = [preds[i]["ground_truth"] for i in order]
    raw = _reduce_counts(wer_counts(pr, gt), device)
    post = _reduce_counts(wer_counts([post_normalize(p) for p in pr], [post_normalize(g) for g in gt]), device)
    gathered = _gather(preds)
    merged: Dict[int, Dict[str, str]] = {}
    for g in gathered:
        merged.update(g)
    n = max(raw["n"], 1)
    return {"WER": raw["errors"] / max(raw["ref_words"], 1), "gtms": raw["exact"] / n,
            "post_ST_wer": post["errors"] / max(post["ref_words"], 1), "post_gtms": post["exact"] / n,
            "n": raw["n"], "predictions": [merged[i] for i in sorted(merged)] if rank == 0 else None}
