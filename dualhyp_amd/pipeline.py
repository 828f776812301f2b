"""Several batches in flight on one GPU: E engines (KV cache + workspace + decode graph each) that share
ONE copy of the weights, each driven from its own host thread on its own HIP stream.

A single batch-32 decode step is a chain of ~160 dependent, latency-bound launches that leaves most of
the chip idle, and a prefill is MFMA-bound; with two independent batches in flight the hardware
interleaves the two chains (measured: 149 -> 112 ms per batch of 32 at the bench workload).  Results
do not depend on which engine ran a batch: same kernels, same reduction orders.
"""
from __future__ import annotations

import threading
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Any, List, Optional, Sequence

import torch

from .generate import generate_batch
from .gpt import GPT


class BatchPipeline:
    def __init__(self, model: GPT, n_engines: int = 2, max_batch: int = 32, s_max: Optional[int] = None,
                 max_tokens: Optional[int] = None) -> None:
        assert n_engines >= 1
        self.models: List[GPT] = [model]
        for _ in range(n_engines - 1):
            clone = type(model)(model.config)
            clone = clone.to(device=model.transformer.wte.weight.device, dtype=model.transformer.wte.weight.dtype)
            clone.load_state_dict(model.state_dict(), strict=True, assign=True)   # share storage, no copy
            clone.cpu_rsqrt_vec_width = model.cpu_rsqrt_vec_width
            clone.eval()
            self.models.append(clone)
        dev = model.transformer.wte.weight.device
        for m in self.models:
            m.set_capacity(max_batch, s_max, max_tokens)
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.models]
        self._free = list(range(n_engines))
        self._cv = threading.Condition()
        self._pool = ThreadPoolExecutor(max_workers=n_engines, thread_name_prefix="dualhyp-engine")
        self.device = dev

    def _run(self, prompts: Sequence[torch.Tensor], max_new_tokens: int, kw: dict) -> List[torch.Tensor]:
        with self._cv:
            while not self._free:
                self._cv.wait()
            e = self._free.pop()
        try:
            with torch.cuda.device(self.device), torch.cuda.stream(self.streams[e]):
                out = generate_batch(self.models[e], prompts, max_new_tokens, **kw)
            return out
        finally:
            with self._cv:
                self._free.append(e)
                self._cv.notify()

    def warm(self, prompts: Sequence[torch.Tensor], max_new_tokens: int, **kw: Any) -> None:
        """Run one batch on EVERY engine (allocates its cache/workspace, captures its decode graph)."""
        for e, m in enumerate(self.models):
            with torch.cuda.device(self.device), torch.cuda.stream(self.streams[e]):
                generate_batch(m, prompts, max_new_tokens, **kw)
        torch.cuda.synchronize(self.device)

    def submit(self, prompts: Sequence[torch.Tensor], max_new_tokens: int, **kw: Any) -> "Future[List[torch.Tensor]]":
        """Queue one batch; the future resolves to generate_batch's result."""
        return self._pool.submit(self._run, prompts, max_new_tokens, kw)

    def close(self) -> None:
        self._pool.shutdown(wait=True)
        for m in self.models[1:]:
            m._drop_engine()
