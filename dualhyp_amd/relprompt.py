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


class GPT(_GPT):
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
