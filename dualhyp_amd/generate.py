"""The decode loop of the reference (generate/base.py:19-82), batched and kept on the device.

`generate()` has the reference's signature and semantics for one prompt (EOS excluded from the
result, quirk Q7; `top_k=1` is a deterministic lowest-index arg-max instead of a sampled tie
break, quirk Q6).  `generate_batch()` runs many ragged prompts at once — equal to running each
alone — with one packed prefill and one hipGraph launch per generated token; the host reads the
device state back once at the end instead of once per token (generate/base.py:79).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops
from .gpt import GPT


EOS_CHECK_EVERY = 16     # decode steps between two "has every sequence finished" read-backs (generate_batch with an eos_id)


@torch.inference_mode()
def generate_batch(model: GPT, prompts: Sequence[torch.Tensor], max_new_tokens: int, *, temperature: float = 1.0,
                   top_k: Optional[int] = None, eos_id: Optional[int] = None, seed: int = 1337,
                   return_state: bool = False, prefill_batch: int = 32, timing: Optional[dict] = None):
    """prompts: 1-D int64 tensors (any lengths).  Returns a list of 1-D tensors prompt+generated,
    cut before the EOS token when one was produced.

    More than `prefill_batch` prompts are prefilled `prefill_batch` at a time (each prefill is
    MFMA-bound and fills the chip on its own) into consecutive KV-cache slots of one engine and then
    decoded TOGETHER: a decode step streams the weights once for all rows (up to 256 rows take the
    streaming kernels), which is where the time of a small-batch decode goes.  A sequence's tokens do
    not depend on how many others ride along (every kernel's per-row summation order is fixed by the
    phase, not by the packing); only the multinomial draw is keyed by the row index in the call."""
    B = len(prompts)
    assert B > 0 and max_new_tokens > 0
    lens = [int(p.numel()) for p in prompts]
    T_max = max(lens)
    need_pos = T_max + max_new_tokens - 1
    if model.max_seq_length < need_pos:
        raise NotImplementedError(f"max_seq_length {model.max_seq_length} needs to be >= {need_pos}")
    dev = model.transformer.wte.weight.device
    chunks = [(c, min(c + prefill_batch, B)) for c in range(0, B, prefill_batch)]
    eng = model.engine(B, need_pos, max(sum(lens[a:b]) for a, b in chunks), exact=B > prefill_batch)
    tok_ld = T_max + max_new_tokens
    if min(lens) == T_max:             # equal lengths: one copy
        tokens = torch.nn.functional.pad(torch.stack([p.to(dev) for p in prompts]), (0, tok_ld - T_max))
    else:                              # ragged: right-pad with 0 (pad_sequence), then out to the buffer width
        tokens = torch.nn.utils.rnn.pad_sequence([p.to(dev) for p in prompts], batch_first=True)
        tokens = torch.nn.functional.pad(tokens, (0, tok_ld - tokens.size(1)))
    tokens = tokens.contiguous()
    length = torch.tensor(lens, dtype=torch.int32, device=dev)
    done = torch.zeros(B, dtype=torch.int32, device=dev)
    eng.set_rsqrt_emulation(model.cpu_rsqrt_vec_width, whole_call=False)   # B independent batch-1 runs
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timing is not None else None
    if ev:
        ev[0].record()
    last = torch.empty((B, eng.vocab), dtype=torch.bfloat16, device=dev)
    for a, b in chunks:
        packed = torch.cat([p.to(dev).reshape(-1) for p in prompts[a:b]])
        _, last[a:b] = eng.forward(packed, lens[a:b], [0] * (b - a), want_all=False, want_last=True, slot_base=a)
    ops.sample(last, tokens, length, done, temperature=temperature, top_k=top_k, eos_id=eos_id, seed=seed, step=0)
    if ev:
        ev[1].record()
    steps_run = 0
    if max_new_tokens > 1:
        n = max_new_tokens - 1
        if eos_id is None:
            eng.decode(tokens, length, done, n, temperature, top_k, eos_id, seed, first_step=0)
            steps_run = n
        else:
            # with an EOS the loop is issued EOS_CHECK_EVERY steps at a time and ends once every sequence has finished
            # (generate/base.py:79-80 returns at the EOS; the harness asks for up to 150 tokens, inference/ger.py:71, and a
            # correction is usually 20-40): one 4-byte read-back per chunk instead of up to 5x the steps
            while steps_run < n:
                c = min(EOS_CHECK_EVERY, n - steps_run)
                eng.decode(tokens, length, done, c, temperature, top_k, eos_id, seed, first_step=steps_run)
                steps_run += c
                if steps_run < n and bool((done != 0).all()):
                    break
    if ev:
        ev[2].record()
    model._cache_len = []  # slots now hold these sequences; a later cached forward must start at 0
    length_h = length.tolist()          # the one host read-back
    done_h = done.tolist()
    if ev:   # the read-back above has synchronised the stream
        timing["prefill_ms"] = timing.get("prefill_ms", 0.0) + ev[0].elapsed_time(ev[1])
        timing["decode_ms"] = timing.get("decode_ms", 0.0) + ev[1].elapsed_time(ev[2])
        timing["decode_steps"] = timing.get("decode_steps", 0) + steps_run
        timing["decode_row_steps"] = timing.get("decode_row_steps", 0) + B * steps_run
    out: List[torch.Tensor] = []
    for i in range(B):
        n = min(length_h[i], lens[i] + max_new_tokens)
        if done_h[i] == 1:
            n -= 1                      # generate/base.py:80 returns idx[:input_pos]: EOS excluded
        out.append(tokens[i, :n])       # a view of this call's own buffer (640 clone launches per 20-batch group otherwise)
    if return_state:
        return out, dict(tokens=tokens, length=length, done=done)
    return out


@torch.inference_mode()
def generate(model: GPT, idx: torch.Tensor, max_returned_tokens: int, *, temperature: float = 1.0,
             top_k: Optional[int] = None, eos_id: Optional[int] = None) -> torch.Tensor:
    """Drop-in for generate/base.py:generate (one prompt of shape (T,))."""
    T = idx.size(0)
    assert max_returned_tokens > T
    if model.max_seq_length < max_returned_tokens - 1:
        raise NotImplementedError(f"max_seq_length {model.max_seq_length} needs to be >= {max_returned_tokens - 1}")
    return generate_batch(model, [idx], max_returned_tokens - T, temperature=temperature, top_k=top_k, eos_id=eos_id)[0]
