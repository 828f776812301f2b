"""RelPrompt variant of the decoder (ger/relprompt.py:182-294): the same LoRA decoder whose token
embedding grows by the three reliability tokens <<C>>/<<M>>/<<N>> (ids vocab..vocab+2,
inference/relprompt.py:341-342) while lm_head keeps the original vocabulary.  The audio/video
encoders and the NoiseMaskClassifier that PREDICT those tokens are upstream of the LLM path
(SURVEY.md §8f "next"); prompts carry the tokens already (dualhyp_amd.data.relprompt_prompt)."""
from __future__ import annotations

from typing import List, Optional, Union

import torch
import torch.nn as nn

from .gpt import GPT as _GPT


class NoiseMaskClassifier(nn.Module):
    """ger/relprompt.py:126-147 — per-chunk reliability logits (clean / mixed / noisy) from encoder features:
    Conv1d(k3) -> ReLU -> Dropout -> Conv1d(k3) -> ReLU -> AvgPool1d(pool, ceil_mode) -> Linear(hidden, 3).
    Same parameter names as the reference (`conv1`, `conv2`, `classifier`), so its checkpoints load.  Inference
    runs on the HIP path: im2col (dh_im2col3_bf16) + MFMA GEMM with the bias folded into the accumulation
    (dh_linear_bf16) twice, then the fused ReLU/pool/linear head (dh_pool_head_bf16)."""

    def __init__(self, input_dim: int, hidden_dim: int = 256, dropout: float = 0.1, pool_size: int = 10) -> None:
        super().__init__()
        self.pool_size = pool_size
        self.conv1 = nn.Conv1d(input_dim, hidden_dim, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(hidden_dim, hidden_dim, kernel_size=3, padding=1)
        self.classifier = nn.Linear(hidden_dim, 3)
        self.dropout = nn.Dropout(dropout)
        self._packed = None

    @staticmethod
    def _pack(conv: nn.Conv1d):
        """[H, ld]: columns dk*C + ci = weight[:, ci, dk], column 3C = bias, zero padded to a multiple of 64."""
        H, Cc, _ = conv.weight.shape
        ld = (3 * Cc + 1 + 63) // 64 * 64
        w = torch.zeros((H, ld), dtype=torch.bfloat16, device=conv.weight.device)
        w[:, : 3 * Cc] = conv.weight.detach().permute(0, 2, 1).reshape(H, 3 * Cc).to(torch.bfloat16)
        w[:, 3 * Cc] = conv.bias.detach().to(torch.bfloat16)
        return w, ld

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from . import ops
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("NoiseMaskClassifier: the HIP path is inference-only (call .eval())")
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or self._packed[0] != sig:
            self._packed = (sig, self._pack(self.conv1), self._pack(self.conv2))
        (w1, ld1), (w2, ld2) = self._packed[1], self._packed[2]
        B, T, _ = x.shape
        x = x.to(torch.bfloat16).contiguous()
        h1 = ops.linear(ops.im2col3(x, ld1), w1).view(B, T, -1)                 # conv1 (+bias), pre-activation
        h2 = ops.linear(ops.im2col3(h1, ld2, relu=True), w2).view(B, T, -1)     # ReLU -> conv2 (+bias)
        return ops.pool_head(h2, self.classifier.weight.detach().to(torch.bfloat16).contiguous(),
                             self.classifier.bias.detach().to(torch.bfloat16).contiguous(), self.pool_size)


class GPT(_GPT):
    def __init__(self, config) -> None:
        super().__init__(config)
        # ger/relprompt.py:212-213: audio features at 50 fps (pool 2 * pool_size), visual at 25 fps
        self.audio_noise_classifier = NoiseMaskClassifier(config.whisper_dim, pool_size=2 * config.pool_size)
        self.visual_noise_classifier = NoiseMaskClassifier(config.raven_dim, pool_size=config.pool_size)

    def resize_token_embeddings(self, new_vocab_size: int) -> None:
        """ADD `new_vocab_size` rows to wte (sic: ger/relprompt.py:215-230 grows by, not to)."""
        old = self.transformer.wte
        old_num, dim = old.weight.shape
        if new_vocab_size == old_num:
            return
        new = nn.Embedding(old_num + new_vocab_size, dim, device=old.weight.device, dtype=old.weight.dtype)
        new.weight.data[:old_num].copy_(old.weight.data)
        nn.init.normal_(new.weight.data[old_num:], mean=0.0, std=old.weight.data.float().std().item())
        self.transformer.wte = new
        self._drop_engine()

    def forward(self, idx: torch.Tensor, audio_query: Optional[torch.Tensor] = None,
                lip_query: Optional[torch.Tensor] = None, max_seq_length: Optional[int] = None,
                input_pos: Optional[torch.Tensor] = None,
                lm_head_chunk_size: int = 0) -> Union[torch.Tensor, List[torch.Tensor]]:
        # audio_query / lip_query are accepted and unused, as in the reference (its av_prompt_length
        # is always 0: ger/relprompt.py:262-268), so positions are not shifted
        return super().forward(idx, input_pos=input_pos, lm_head_chunk_size=lm_head_chunk_size)
