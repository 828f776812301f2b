"""RelPrompt variant of the decoder (ger/relprompt.py:182-294): the same LoRA decoder whose token
embedding grows by the three reliability tokens <<C>>/<<M>>/<<N>> (ids vocab..vocab+2,
inference/relprompt.py:341-342) while lm_head keeps the original vocabulary.  The audio/video
encoders that feed the NoiseMaskClassifier are upstream of the LLM path (out of scope); the classifiers
themselves run and TRAIN here on encoder features: forward and backward on the HIP kernels
(csrc/classifier.hip + the MFMA GEMM), the mask cross entropy weighted 0.02 and the second learning-rate group
of finetune/relprompt.py:175-195,356-403 live in `mask_loss` below and in dualhyp_amd.finetune.fit."""
from __future__ import annotations

from typing import List, Optional, Union

import torch
import torch.nn as nn

from .gpt import GPT as _GPT


def _pack_conv(weight: torch.Tensor, bias: torch.Tensor):
    """[H, ld] bf16: columns dk*C + ci = weight[:, ci, dk], column 3C = bias, zero padded to a multiple of 64."""
    H, Cc, _ = weight.shape
    ld = (3 * Cc + 1 + 63) // 64 * 64
    w = torch.zeros((H, ld), dtype=torch.bfloat16, device=weight.device)
    w[:, : 3 * Cc] = weight.detach().permute(0, 2, 1).reshape(H, 3 * Cc).to(torch.bfloat16)
    w[:, 3 * Cc] = bias.detach().to(torch.bfloat16)
    return w, ld


class _ClassifierFn(torch.autograd.Function):
    """NoiseMaskClassifier forward + backward on the HIP kernels.  Activations and activation gradients are bf16 (the
    rounding points of the reference's bf16 forward: conv output, pooled mean, logits), parameter gradients are
    accumulated in fp32 over the tokens (dh_tn_accum_f32) — the same split as the LoRA fine-tune path (train.py)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, wc, bc, pool: int, p_drop: float, training: bool):
        from . import ops
        B, T, C = x.shape
        H = w1.size(0)
        (W1, ld1), (W2, ld2) = _pack_conv(w1, b1), _pack_conv(w2, b2)
        wcb = wc.detach().to(torch.bfloat16).contiguous()
        x = x.detach().to(torch.bfloat16).contiguous()
        col1 = ops.im2col3(x, ld1)
        h1 = ops.linear(col1, W1).view(B, T, H)                       # conv1 (+bias), pre-activation
        mask = None
        if training and p_drop > 0.0:                                 # ger/relprompt.py:140: dropout between the two convs
            mask = (torch.rand(h1.shape, device=h1.device) >= p_drop).to(torch.bfloat16) * (1.0 / (1.0 - p_drop))
            col2 = ops.im2col3(torch.relu(h1) * mask, ld2)
        else:
            col2 = ops.im2col3(h1, ld2, relu=True)
        h2 = ops.linear(col2, W2).view(B, T, H)                       # conv2 (+bias), pre-activation
        logits = ops.pool_head(h2, wcb, bc.detach().to(torch.bfloat16).contiguous(), pool)
        ctx.save_for_backward(col1, h1, col2, h2, W2, wcb)
        ctx.mask, ctx.pool, ctx.dims = mask, pool, (B, T, C, H)
        ctx.dtypes = (w1.dtype, b1.dtype, w2.dtype, b2.dtype, wc.dtype, bc.dtype)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        from . import ops
        col1, h1, col2, h2, W2, wcb = ctx.saved_tensors
        B, T, C, H = ctx.dims
        dev = h2.device
        dl = dlogits.float().contiguous()                                        # [B, P, 3]
        dh2, pooled = ops.pool_head_bwd(h2, wcb, dl, ctx.pool)
        # Linear(H, 3): gWc[j, c] = sum_bp dl[bp, j] pooled[bp, c]; the 3 rows ride a 16-row bf16 operand
        dl16 = torch.zeros((dl.numel() // 3, 16), dtype=torch.bfloat16, device=dev)
        dl16[:, :3] = dl.view(-1, 3)
        gwc = torch.empty((16, H), dtype=torch.float32, device=dev)
        ops.tn_accum(dl16, pooled, gwc, accumulate=False)
        gbc = dl.view(-1, 3).sum(0)
        # conv2 as a GEMM over col2: dW2p = dh2^T col2 (the ones column gives the bias gradient), dcol2 = dh2 W2p
        n = B * T
        gW2 = torch.empty((H, col2.size(1)), dtype=torch.float32, device=dev)
        ops.tn_accum(dh2.view(n, H), col2, gW2, accumulate=False)
        dcol2 = ops.linear(dh2.view(n, H), W2.t().contiguous())                 # [n, ld2]
        dh1 = ops.col2im3(dcol2, B, T, H, pre=h1, mask=ctx.mask)                 # through dropout and ReLU
        gW1 = torch.empty((H, col1.size(1)), dtype=torch.float32, device=dev)
        ops.tn_accum(dh1.view(n, H), col1, gW1, accumulate=False)

        def unpack(g, cin):
            return g[:, : 3 * cin].reshape(H, 3, cin).permute(0, 2, 1).contiguous(), g[:, 3 * cin].contiguous()
        gw1, gb1 = unpack(gW1, C)
        gw2, gb2 = unpack(gW2, H)
        grads = [gw1, gb1, gw2, gb2, gwc[:3].contiguous(), gbc]
        grads = [g.to(dt) for g, dt in zip(grads, ctx.dtypes)]
        return (None, *grads, None, None, None)


def mask_loss(audio_logits: torch.Tensor, visual_logits: torch.Tensor, audio_targets: torch.Tensor,
              visual_targets: torch.Tensor) -> torch.Tensor:
    """finetune/relprompt.py:361-387: cross entropy of the per-chunk reliability logits [B, P, 3] against the
    class indices (<<C>> 0, <<M>> 1, <<N>> 2) [B, P'], the longer of prediction / target trimmed to the shorter,
    audio + visual.  The caller weights it by `mask_loss_weight` (0.02) and adds it to the LM loss (`:400-403`)."""
    import torch.nn.functional as F

    def one(lg, tg):
        n = min(lg.size(1), tg.size(1))
        return F.cross_entropy(lg[:, :n].float().reshape(-1, 3), tg[:, :n].reshape(-1))
    return one(audio_logits, audio_targets) + one(visual_logits, visual_targets)


def labels_to_indices(labels_list, device, prefix: str = "") -> torch.Tensor:
    """finetune/relprompt.py:73-79: '<<C>>' -> 0, '<<M>>' -> 1, anything else -> 2."""
    rows = [[0 if l == f"<<{prefix}C>>" else (1 if l == f"<<{prefix}M>>" else 2) for l in labels] for labels in labels_list]
    return torch.tensor(rows, device=device)


@torch.no_grad()
def predicted_mask_prompt(model: nn.Module, prompt_with_placeholders: str, audio_enc_features: torch.Tensor,
                          visual_enc_features: torch.Tensor, mask_tokens=("<<C>>", "<<M>>", "<<N>>")):
    """inference/relprompt.py:113-153 for one utterance: per-chunk reliability classes from the two classifiers (arg-max of
    their logits) become the mask tokens of the prompt.  -> (prompt, audio class indices, visual class indices)."""
    dev = next(model.parameters()).device
    a = model.audio_noise_classifier(audio_enc_features.to(dev).unsqueeze(0)).float().argmax(-1)[0].cpu()
    v = model.visual_noise_classifier(visual_enc_features.to(dev).unsqueeze(0)).float().argmax(-1)[0].cpu()
    prompt = (prompt_with_placeholders.replace("<<<ASR_MASKS>>>", "".join(mask_tokens[i] for i in a.tolist()))
              .replace("<<<VSR_MASKS>>>", "".join(mask_tokens[i] for i in v.tolist())))
    return prompt, a, v


def mark_only_lora_as_trainable(model: nn.Module, bias: str = "none") -> None:
    """ger/relprompt.py:79-119: as ger/lora.py's, but the noise classifiers stay trainable."""
    from .gpt import mark_only_lora_as_trainable as base
    base(model, bias)
    for n, p in model.named_parameters():
        if "noise_classifier" in n:
            p.requires_grad = True


def classifier_parameters(model: nn.Module) -> List[torch.nn.Parameter]:
    return [p for n, p in model.named_parameters() if "noise_classifier" in n]


def prepare_classifiers_for_training(model: nn.Module) -> List[torch.nn.Parameter]:
    """fp32 masters for the classifier parameters (what bf16-mixed keeps in fp32 and updates), requires_grad on."""
    ps = classifier_parameters(model)
    for p in ps:
        p.data = p.data.float()
        p.requires_grad_(True)
    return ps


class NoiseMaskClassifier(nn.Module):
    """ger/relprompt.py:126-147 — per-chunk reliability logits (clean / mixed / noisy) from encoder features:
    Conv1d(k3) -> ReLU -> Dropout -> Conv1d(k3) -> ReLU -> AvgPool1d(pool, ceil_mode) -> Linear(hidden, 3).
    Same parameter names as the reference (`conv1`, `conv2`, `classifier`), so its checkpoints load.  Runs on the
    HIP path: im2col (dh_im2col3_bf16) + MFMA GEMM with the bias folded into the accumulation (dh_linear_bf16) twice,
    then the fused ReLU/pool/linear head (dh_pool_head_bf16); with gradients enabled the same kernels run inside
    `_ClassifierFn`, whose backward (dh_pool_head_bwd_bf16, dh_col2im3_bf16, dh_tn_accum_f32, dh_linear_bf16) returns the
    gradients of all six parameters."""

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
        return _pack_conv(conv.weight, conv.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from . import ops
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # fine-tune (finetune/relprompt.py:356-357): differentiable, dropout active in train mode
            return _ClassifierFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                       self.classifier.weight, self.classifier.bias, self.pool_size, float(self.dropout.p),
                                       self.training)
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("NoiseMaskClassifier: a train-mode forward without gradients would apply dropout; call .eval()")
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
