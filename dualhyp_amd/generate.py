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


@torch.inference_mode()
def generate_batch(model: GPT, prompts: Sequence[torch.Tensor], max_new_tokens: int, *, temperature: float = 1.0,
                   top_k: Optional[int] = None, eos_id: Optional[int] = None, seed: int = 1337,
                   return_state: bool = False):
    """prompts: 1-D int64 tensors (any lengths).  Returns a list of 1-D tensors prompt+generated,
    cut before the EOS token when one was produced."""
    B = len(prompts)
    assert B > 0 and max_new_tokens > 0
    lens = [int(p.numel()) for p in prompts]
    T_max = max(lens)
    need_pos = T_max + max_new_tokens - 1
    if model.max_seq_length < need_pos:
        raise NotImplementedError(f"max_seq_length {model.max_seq_length} needs to be >= {need_pos}")
    dev = model.transformer.wte.weight.device
    eng = model.engine(B, need_pos, sum(lens))
    tok_ld = T_max + max_new_tokens
    tokens = torch.zeros((B, tok_ld), dtype=torch.int64, device=dev)
    for i, p in enumerate(prompts):
        tokens[i, : lens[i]] = p.to(dev)
    length = torch.tensor(lens, dtype=torch.int32, device=dev)
    done = torch.zeros(B, dtype=torch.int32, device=dev)
    packed = torch.cat([p.to(dev).reshape(-1) for p in prompts])
    eng.set_rsqrt_emulation(model.cpu_rsqrt_vec_width, whole_call=False)   # B independent batch-1 runs
    _, last = eng.forward(packed, lens, [0] * B, want_all=False, want_last=True)
    ops.sample(last, tokens, length, done, temperature=temperature, top_k=top_k, eos_id=eos_id, seed=seed, step=0)
    if max_new_tokens > 1:
        eng.decode(tokens, length, done, max_new_tokens - 1, temperature, top_k, eos_id, seed, first_step=0)
    model._cache_len = []  # slots now hold these sequences; a later cached forward must start at 0
    length_h = length.tolist()          # the one host read-back
    done_h = done.tolist()
    out: List[torch.Tensor] = []
    for i in range(B):
        n = min(length_h[i], lens[i] + max_new_tokens)
        if done_h[i] == 1:
            n -= 1                      # generate/base.py:80 returns idx[:input_pos]: EOS excluded
        out.append(tokens[i, :n].clone())
    if return_state:
        return out, dict(tokens=tokens, length=length, done=done)
    return out


@torch.inference_mode()
def generate_gang(models: Sequence[GPT], batches: Sequence[Sequence[torch.Tensor]], max_new_tokens: int, *,
                  temperature: float = 1.0, top_k: Optional[int] = None, eos_id: Optional[int] = None,
                  seed: int = 1337, streams: Optional[Sequence["torch.cuda.Stream"]] = None) -> List[List[torch.Tensor]]:
    """len(batches) <= len(models) batches in flight on one GPU: the prefills run one after the other on
    the current stream (each is MFMA-bound and wants the whole chip), then every batch's decode loop runs
    concurrently on its own engine / HIP stream (each is a latency-bound chain of small launches; several
    of them interleave on the hardware).  `models` share one copy of the weights
    (dualhyp_amd.pipeline.BatchPipeline builds them).  Same results as generate_batch per batch."""
    assert 0 < len(batches) <= len(models)
    cur = torch.cuda.current_stream()
    if streams is None:
        streams = [torch.cuda.Stream() for _ in batches]
    states = []
    for m, prompts in zip(models, batches):                       # phase 1: exclusive prefills
        B = len(prompts)
        lens = [int(p.numel()) for p in prompts]
        need_pos = max(lens) + max_new_tokens - 1
        if m.max_seq_length < need_pos:
            raise NotImplementedError(f"max_seq_length {m.max_seq_length} needs to be >= {need_pos}")
        dev = m.transformer.wte.weight.device
        eng = m.engine(B, need_pos, sum(lens))
        tok_ld = max(lens) + max_new_tokens
        tokens = torch.nn.utils.rnn.pad_sequence([p.to(dev) for p in prompts], batch_first=True)
        tokens = torch.nn.functional.pad(tokens, (0, tok_ld - tokens.size(1))).contiguous()
        length = torch.tensor(lens, dtype=torch.int32, device=dev)
        done = torch.zeros(B, dtype=torch.int32, device=dev)
        packed = torch.cat([p.to(dev).reshape(-1) for p in prompts])
        eng.set_rsqrt_emulation(m.cpu_rsqrt_vec_width, whole_call=False)
        _, last = eng.forward(packed, lens, [0] * B, want_all=False, want_last=True)
        ops.sample(last, tokens, length, done, temperature=temperature, top_k=top_k, eos_id=eos_id, seed=seed, step=0)
        m._cache_len = []
        states.append((eng, tokens, length, done, lens))
    if max_new_tokens > 1:                                          # phase 2: concurrent decode loops
        for st, (eng, tokens, length, done, lens) in zip(streams, states):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                eng.decode(tokens, length, done, max_new_tokens - 1, temperature, top_k, eos_id, seed, first_step=0)
    outs: List[List[torch.Tensor]] = []
    for st, (eng, tokens, length, done, lens) in zip(streams, states):
        with torch.cuda.stream(st):
            length_h, done_h = length.tolist(), done.tolist()
        cur.wait_stream(st)
        o = []
        for i in range(len(lens)):
            n = min(length_h[i], lens[i] + max_new_tokens)
            if done_h[i] == 1:
                n -= 1
            o.append(tokens[i, :n].clone())
        outs.append(o)
    return outs


@torch.inference_mode()
def generate(model: GPT, idx: torch.Tensor, max_returned_tokens: int, *, temperature: float = 1.0,
             top_k: Optional[int] = None, eos_id: Optional[int] = None) -> torch.Tensor:
    """Drop-in for generate/base.py:generate (one prompt of shape (T,))."""
    T = idx.size(0)
    assert max_returned_tokens > T
    if model.max_seq_length < max_returned_tokens - 1:
        raise NotImplementedError(f"max_seq_length {model.max_seq_length} needs to be >= {max_returned_tokens - 1}")
    return generate_batch(model, [idx], max_returned_tokens - T, temperature=temperature, top_k=top_k, eos_id=eos_id)[0]
