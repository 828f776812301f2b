"""LoRA fine-tune path of the decoder (finetune/ger.py:212-353) on the HIP kernels.

`forward_train` is what `GPT.forward` dispatches to when gradients are enabled: the no-cache forward
of ger/lora.py:538-549 built from the same kernels as inference (tiled MFMA GEMMs with the fused
LoRA epilogue, RMSNorm, rope, flash attention) while keeping the activations the backward needs;
its autograd node runs the hand-written backward kernels (csrc/train_kernels.hip,
csrc/attention_bwd.hip) plus the dX GEMMs on transposed copies of the frozen weights, and returns
gradients for the LoRA parameters only — the base model is frozen
(`mark_only_lora_as_trainable`, ger/lora.py:405-439), so no dW of a dense layer is ever formed.

Numerics: activations and activation gradients are bf16 (the kernels' rounding points are those of
the reference's bf16 forward); LoRA gradients are accumulated in fp32.  The reference's fine-tune runs
under bf16-mixed autocast (fp32 residual stream); DESIGN.md lists that as a documented difference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import os

import torch

from . import ops
from .gpt import GPT, LoRALinear, LoRAQKVLinear, _pad_rank

BF = torch.bfloat16


class _Frozen:
    """Transposed copies of the frozen weights for the dX GEMMs (y = x W^T  =>  dx = dy W, computed as
    dy · (W^T)^T by the same K-contiguous NT kernels).  Built once; 1x the model's size in HBM."""

    def __init__(self, model: GPT) -> None:
        self.sig = tuple(p.data_ptr() for n, p in model.named_parameters() if "lora_" not in n)
        self.lm_T = model.lm_head.linear.weight.data.t().contiguous()                      # [d, V]
        self.scale_is_one = bool((model.lm_head.adapter_scale.data == 1).all())
        self.layers = []
        for blk in model.transformer.h:
            w1, w2 = blk.mlp.fc_1.linear.weight.data, blk.mlp.fc_2.linear.weight.data
            self.layers.append(dict(
                qkv_T=blk.attn.attn.linear.weight.data.t().contiguous(),                    # [d, qkv]
                proj_T=blk.attn.proj.linear.weight.data.t().contiguous(),                   # [d, d]
                w12_T=torch.cat([w1, w2], dim=0).t().contiguous(),                          # [d, 2I]
                mlp_T=blk.mlp.proj.linear.weight.data.t().contiguous(),                     # [I, d]
            ))


def _frozen(model: GPT) -> _Frozen:
    fz = getattr(model, "_train_frozen", None)
    sig = tuple(p.data_ptr() for n, p in model.named_parameters() if "lora_" not in n)
    if fz is None or fz.sig != sig:
        fz = _Frozen(model)
        model._train_frozen = fz
    return fz


def _lora_views(mod, qkv: bool):
    """bf16, rank-padded operands of one LoRA module for the forward and backward kernels, rebuilt only when
    the fp32 masters changed (an optimizer step), not at every micro-step: (A, B, A^T, B^T[, block B^T]).
    Under hipGraph capture nothing is cached — the casts must be part of the captured step."""
    key = (mod.lora_A.data_ptr(), mod.lora_A._version, mod.lora_B.data_ptr(), mod.lora_B._version)
    capturing = torch.cuda.is_current_stream_capturing()
    hit = getattr(mod, "_train_views", None)
    if hit is not None and hit[0] == key and not capturing:
        return hit[1]
    A, B = mod.padded_lora()
    At, Bt = A.t().contiguous(), B.t().contiguous()
    views = [A, B, At, Bt]
    if qkv:
        qd = B.size(0)
        s0, s1 = mod.splits
        bounds = (0, s0, s1, qd)
        Bblk = torch.zeros((64, qd), dtype=BF, device=B.device)      # block-structured B^T, rank slots 48..63 zero
        for seg in range(3):
            Bblk[16 * seg:16 * seg + 16, bounds[seg]:bounds[seg + 1]] = B[bounds[seg]:bounds[seg + 1]].t()
        views += [Bblk, _pad64(At)]
    else:
        views += [_pad64(At)]
    if not capturing:
        mod._train_views = (key, views)
    return views


def _all_lora_views(model: GPT):
    """The operands `_lora_views` builds for ONE module, for every layer at once: the fp32 masters of all layers are stacked and
    cast / transposed / padded as [n_layer, ...] tensors (about 25 small kernels per micro-step instead of about 15 per layer — at 22
    layers 2 ms of launches).  Returns [(qkv views, proj views)] per layer — slices of the stacked tensors, the same values as
    `_lora_views` — or None when the layers are not uniform rank-16 all-enabled adapters (then the caller builds them per layer)."""
    if os.environ.get("DUALHYP_NO_BATCHED_VIEWS"):      # same-box A/B
        return None
    blocks = list(model.transformer.h)
    qs, ps = [b.attn.attn for b in blocks], [b.attn.proj for b in blocks]
    def uniform(ms):
        return (all(getattr(m, "lora_active", False) and getattr(m, "r", 0) == 16 for m in ms)
                and len({(tuple(m.lora_A.shape), tuple(m.lora_B.shape)) for m in ms}) == 1)
    if not blocks or not uniform(qs) or not uniform(ps) or not all(all(m.enable_lora) for m in qs):
        return None
    if len({tuple(m.splits) for m in qs}) != 1:
        return None

    def base(ms):
        A = torch.stack([m.lora_A.data for m in ms]).to(BF)              # [L, r*, in]
        B = torch.stack([m.lora_B.data for m in ms]).to(BF)              # [L, out, 16]
        At, Bt = A.transpose(1, 2).contiguous(), B.transpose(1, 2).contiguous()
        At64 = At.new_zeros((*At.shape[:-1], 64))
        At64[..., :At.size(-1)] = At
        return A, B, At, Bt, At64
    qA, qB, qAt, qBt, qAt64 = base(qs)
    s0, s1 = qs[0].splits
    qd = qB.size(1)
    bounds = (0, s0, s1, qd)
    Bblk = qB.new_zeros((len(blocks), 64, qd))                           # block-structured B^T, rank slots 48..63 zero
    for seg in range(3):
        Bblk[:, 16 * seg:16 * seg + 16, bounds[seg]:bounds[seg + 1]] = qBt[:, :, bounds[seg]:bounds[seg + 1]]
    pA, pB, pAt, pBt, pAt64 = base(ps)
    return [([qA[i], qB[i], qAt[i], qBt[i], Bblk[i], qAt64[i]], [pA[i], pB[i], pAt[i], pBt[i], pAt64[i]]) for i in range(len(blocks))]


def lora_parameters(model: GPT) -> List[torch.nn.Parameter]:
    """The trainable tensors in the fixed order the autograd node uses."""
    out = []
    for blk in model.transformer.h:
        for m in (blk.attn.attn, blk.attn.proj):
            if getattr(m, "r", 0) > 0 and hasattr(m, "lora_A"):
                out += [m.lora_A, m.lora_B]
    return out


def prepare_for_training(model: GPT) -> List[torch.nn.Parameter]:
    """bf16 frozen base + fp32 LoRA masters (what bf16-mixed keeps in fp32 and actually updates)."""
    from .gpt import mark_only_lora_as_trainable
    mark_only_lora_as_trainable(model)
    ps = lora_parameters(model)
    for p in ps:
        p.data = p.data.float()
        p.requires_grad_(True)
    model._drop_engine()
    return ps


class _Layer:
    __slots__ = ("x", "n1", "n1d", "xa", "q", "k", "v", "y", "lse", "xa2", "yd", "x1", "n2", "g", "u", "act", "mask1", "mask2", "vq", "vp")


def _drop(x: torch.Tensor, p: float, training: bool, model=None, call_id: int = 0):
    """LoRA-branch dropout (ger/lora.py:96,165,391): mask scaled by 1/(1-p), result rounded to bf16.  The mask is drawn and
    applied by ONE HIP kernel (ops.dropout, Philox keyed by the model's dropout seed and the call site, counted by a device-side
    micro-step counter that `_DecoderFn.forward` bumps once per forward — so a captured hipGraph replays with fresh masks);
    round 3 drew it with torch.rand and two ATen multiplies."""
    if not training or p <= 0.0:
        return x, None
    seed, step = _dropout_state(model, x.device)
    return ops.dropout(x, p, seed, call_id, step)


def _dropout_state(model, dev):
    """(seed, device step counter) of a model's LoRA dropout: seeded from torch's generator at first use (torch.manual_seed makes
    a run reproducible), the counter an int64 tensor on the device."""
    st = getattr(model, "_dropout_rng", None)
    if st is None or st[1].device != torch.device(dev):
        st = (int(torch.initial_seed()) & (2 ** 63 - 1), torch.zeros(1, dtype=torch.int64, device=dev))
        model._dropout_rng = st
    return st


def _masked(lo: torch.Tensor, mask: Optional[torch.Tensor], s: float) -> torch.Tensor:
    """bf16(lo * mask * s) as torch computes it left to right; a multiplication by exactly 1.0 changes no bit and is skipped
    (alpha == r in both reference harnesses: one elementwise pass over [tokens, d] less per adapter and layer)."""
    if mask is not None:
        lo = lo * mask
    if s != 1.0:
        lo = lo * s
    return lo.contiguous()


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: GPT, idx: torch.Tensor, *lora_ps: torch.Tensor) -> torch.Tensor:
        cfg = model.config
        dev = idx.device
        B, T = idx.shape
        n_tok = B * T
        d, H, G, hs = cfg.n_embd, cfg.n_head, cfg.n_query_groups, cfg.head_size
        i32 = torch.int32
        if model.rope_cache is None or model.rope_cache[0].device != dev:
            model.rope_cache = model.build_rope_cache(idx)
        cos, sin = model.rope_cache
        tail = getattr(model, "_row_tail_override", None)      # GraphedTrainStep: flags of the UNPADDED call
        V = model.cpu_rsqrt_vec_width
        if tail is None and V:
            tail = (torch.arange(n_tok, device=dev) >= n_tok // V * V).to(torch.uint8)
        tok_slot = torch.arange(B, dtype=i32, device=dev).repeat_interleave(T)
        tok_pos = torch.arange(T, dtype=i32, device=dev).repeat(B)
        seq_slot = torch.arange(B, dtype=i32, device=dev)
        q_start = seq_slot * T
        q_len = torch.full((B,), T, dtype=i32, device=dev)
        zeros = torch.zeros(B, dtype=i32, device=dev)
        s_max = -(-T // 64) * 64
        kc = torch.zeros((B, G, s_max, hs), dtype=BF, device=dev)       # scratch, reused by every layer
        vt = torch.zeros((B, G, hs, s_max), dtype=BF, device=dev)
        p_drop, training = cfg.dropout, model.training
        saved: List[_Layer] = []
        if training and p_drop > 0.0:
            _dropout_state(model, dev)[1].add_(1)      # one micro-step: a torch op, so a hipGraph capture replays the bump too
        x = ops.embed(idx.reshape(-1), model.transformer.wte.weight.data)
        all_views = _all_lora_views(model)          # every layer's LoRA operands from stacked masters (None: per layer below)
        for blk in model.transformer.h:
            L = _Layer()
            qkv_m, proj_m = blk.attn.attn, blk.attn.proj
            L.x = x
            L.n1 = ops.rmsnorm(x, blk.norm_1.weight.data, cfg.norm_eps, row_tail=tail)
            if qkv_m.lora_active:
                L.vq = all_views[len(saved)][0] if all_views is not None else _lora_views(qkv_m, True)      # kept for the backward of the same micro-step (same weights)
                A48, B16 = L.vq[:2]
                L.n1d, L.mask1 = _drop(L.n1, p_drop, training, model, 2 * len(saved))
                L.xa = ops.linear(L.n1d, A48)
                qkv = ops.linear(L.n1, qkv_m.linear.weight.data, epilogue=ops.EPI_LORA, xa=L.xa, lora_b=B16,
                                 lora_scale=qkv_m.scaling, splits=qkv_m.splits)
            else:
                L.n1d = L.mask1 = L.xa = None
                qkv = ops.linear(L.n1, qkv_m.linear.weight.data)
            L.k = torch.empty((n_tok, G, hs), dtype=BF, device=dev)
            L.v = torch.empty_like(L.k)
            L.q = ops.qkv_rope_cache(qkv, cos, sin, tok_slot, tok_pos, kc, vt, H, G, k_out=L.k, v_out=L.v)
            L.lse = torch.empty((n_tok, H), dtype=torch.float32, device=dev)
            L.y = ops.attn_prefill(L.q, kc, vt, seq_slot, q_start, q_len, zeros, T, lse=L.lse)
            if proj_m.lora_active:
                L.vp = all_views[len(saved)][1] if all_views is not None else _lora_views(proj_m, False)
                Ap, Bp = L.vp[:2]
                L.yd, L.mask2 = _drop(L.y, p_drop, training, model, 2 * len(saved) + 1)
                L.xa2 = ops.linear(L.yd, Ap)
                L.x1 = ops.linear(L.y, proj_m.linear.weight.data, epilogue=ops.EPI_LORA, xa=L.xa2, lora_b=Bp,
                                  lora_scale=proj_m.scaling, resid=x)
            else:
                L.yd = L.mask2 = L.xa2 = None
                L.x1 = ops.linear(L.y, proj_m.linear.weight.data, resid=x)
            L.n2 = ops.rmsnorm(L.x1, blk.norm_2.weight.data, cfg.norm_eps, row_tail=tail)
            L.act, L.g, L.u = ops.linear_swiglu_train(L.n2, blk.mlp.fc_1.linear.weight.data, blk.mlp.fc_2.linear.weight.data)
            x = ops.linear(L.act, blk.mlp.proj.linear.weight.data, resid=L.x1)
            saved.append(L)
        xf = ops.rmsnorm(x, model.transformer.ln_f.weight.data, cfg.norm_eps, row_tail=tail)
        ctx.model, ctx.saved, ctx.x_last = model, saved, x
        ctx.meta = (B, T, cos, sin, tok_pos, q_start, q_len)
        ctx.param_dtypes = [p.dtype for p in lora_ps]
        if getattr(ctx, "hidden_only", False):      # GraphedTrainStep: lm_head + loss on the rows that carry a target only
            return xf
        logits = ops.linear(xf, model.lm_head.linear.weight.data, epilogue=ops.EPI_ADAPTER,
                            scale=model.lm_head.adapter_scale.data, bias=model.lm_head.adapter_bias.data)
        return logits.view(B, T, -1)

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        model: GPT = ctx.model
        cfg = model.config
        B, T, cos, sin, tok_pos, q_start, q_len = ctx.meta
        d, H, G, hs, I = cfg.n_embd, cfg.n_head, cfg.n_query_groups, cfg.head_size, cfg.intermediate_size
        fz = _frozen(model)
        dev = dlogits.device
        if getattr(ctx, "hidden_only", False):
            dxf = dlogits                                                  # already d loss / d ln_f(x), [B*T, d]
        else:
            dz = dlogits.reshape(B * T, -1).to(BF)
            scale_vec = model.lm_head.adapter_scale.data
            if not fz.scale_is_one:                                        # (frozen: checked once, no host sync here)
                dz = dz * scale_vec                                        # d(scale*(z+bias))/dz
            dz = dz.contiguous()
            dxf = ops.linear(dz, fz.lm_T)
        dx = ops.rmsnorm_bwd(dxf, ctx.x_last, model.transformer.ln_f.weight.data, cfg.norm_eps)
        grads: List[Optional[torch.Tensor]] = []
        per_layer: List[List[Optional[torch.Tensor]]] = []
        bwd_plan = ops.attn_bwd_plan(q_start, q_len, B * T, [T] * B)      # one set of index tensors for all layers
        go = getattr(ctx, "grad_out", None)          # GraphedTrainStep: {id(parameter): fp32 accumulator of its shape}
        for li in range(cfg.n_layer - 1, -1, -1):
            blk, L, W = model.transformer.h[li], ctx.saved[li], fz.layers[li]
            qkv_m, proj_m = blk.attn.attn, blk.attn.proj
            # ---- MLP: x2 = x1 + proj(silu(fc_1 n2) * fc_2 n2)
            dact = ops.linear(dx, W["mlp_T"])
            dgu = ops.swiglu_bwd(dact, L.g, L.u)
            dn2 = ops.linear(dgu, W["w12_T"])
            dx1 = ops.rmsnorm_bwd(dn2, L.x1, blk.norm_2.weight.data, cfg.norm_eps, dres=dx)
            # ---- attention output projection (+LoRA): x1 = x + y Wp^T + s (drop(y) Ap^T) Bp^T
            gA2 = gB2 = None
            if proj_m.lora_active:
                s = proj_m.scaling
                Ap, Bp, ApT, BpT, ApT64 = L.vp                          # [16,d], [d,16] and transposes (built by the forward)
                t = ops.linear(dx1, BpT)                                # dx1 · Bp           [n,16]
                if L.mask2 is None:
                    dy = ops.linear(dx1, W["proj_T"], epilogue=ops.EPI_LORA, xa=t, lora_b=ApT, lora_scale=s)
                else:
                    lo = ops.linear_mul(_pad64(t), ApT64, L.mask2)          # (t Ap) * mask, rounded as the two steps
                    dy = ops.linear(dx1, W["proj_T"], resid=_masked(lo, None, s))
                direct2 = go is not None and proj_m.r == 16 and go[id(proj_m.lora_A)].dtype == torch.float32
                if direct2:                                                      # straight into the caller's accumulators
                    gA2, gB2 = go[id(proj_m.lora_A)], go[id(proj_m.lora_B)]
                else:
                    gB2 = torch.empty((d, 16), dtype=torch.float32, device=dev)      # written whole (accumulate=False)
                    gA2 = torch.empty((16, d), dtype=torch.float32, device=dev)
                ops.tn_accum(dx1, L.xa2, gB2, scale=s, accumulate=direct2)
                ops.tn_accum(t, L.yd, gA2, scale=s, accumulate=direct2)
            else:
                dy = ops.linear(dx1, W["proj_T"])
            # ---- attention + rope
            dq, dk, dv = ops.attn_bwd(L.q, L.k, L.v, L.y, dy, L.lse, q_start, q_len, T, lens=[T] * B, plan=bwd_plan)
            dqkv = ops.qkv_rope_bwd(dq, dk, dv, cos, sin, tok_pos)
            # ---- fused qkv projection (+LoRA, contiguous [Q|K|V] delta placement, quirk Q2)
            gA1 = gB1 = None
            if qkv_m.lora_active:
                s = qkv_m.scaling
                A48, B16, _, _, Bblk, A48T64 = L.vq                      # [48,d], [qkv,16], block B^T [64,qkv], A^T [d,64]
                qd = B16.size(0)
                s0, s1 = qkv_m.splits
                bounds = (0, s0, s1, qd)
                t3 = ops.linear(dqkv, Bblk)                              # [n,64], cols 16seg.. = dqkv[:,seg] · B_seg
                lo = ops.linear(t3, A48T64) if L.mask1 is None else ops.linear_mul(t3, A48T64, L.mask1)      # [n,d] = (t3 · A48) * mask
                dn1 = ops.linear(dqkv, W["qkv_T"], resid=_masked(lo, None, s))
                direct1 = (go is not None and qkv_m.r == 16 and all(qkv_m.enable_lora) and go[id(qkv_m.lora_A)].dtype == torch.float32)
                if direct1:
                    gA1, gB1 = go[id(qkv_m.lora_A)], go[id(qkv_m.lora_B)]
                else:
                    gB1 = torch.empty((qd, 16), dtype=torch.float32, device=dev)    # every segment written whole below
                    gA1 = torch.empty((48, d), dtype=torch.float32, device=dev)
                if s0 % 128 == 0 and s1 % 128 == 0 and dqkv.size(0) >= 64:
                    ops.tn_accum(dqkv, L.xa, gB1, scale=s, accumulate=direct1, splits=(s0, s1))      # the three segments in one launch
                else:
                    for seg in range(3):
                        ops.tn_accum(dqkv[:, bounds[seg]:bounds[seg + 1]], L.xa[:, 16 * seg:16 * seg + 16],
                                     gB1[bounds[seg]:bounds[seg + 1]], scale=s, accumulate=direct1)
                ops.tn_accum(t3[:, :48], L.n1d, gA1, scale=s, accumulate=direct1)
            else:
                dn1 = ops.linear(dqkv, W["qkv_T"])
            dx = ops.rmsnorm_bwd(dn1, L.x, blk.norm_1.weight.data, cfg.norm_eps, dres=dx1)
            # ---- map the rank-padded gradients back onto the parameters' shapes
            lg: List[Optional[torch.Tensor]] = []
            if qkv_m.lora_active and direct1:
                lg += [None, None]                                           # already in the caller's accumulators
            elif qkv_m.lora_active:
                r, en = qkv_m.r, qkv_m.enable_lora
                lg.append(torch.cat([gA1[16 * seg:16 * seg + r] for seg in range(3) if en[seg]], dim=0))
                s0, s1 = qkv_m.splits
                bounds = (0, s0, s1, gB1.size(0))
                lg.append(torch.cat([gB1[bounds[seg]:bounds[seg + 1], :r] for seg in range(3) if en[seg]], dim=0))
            if proj_m.lora_active and direct2:
                lg += [None, None]
            elif proj_m.lora_active:
                lg += [gA2[:proj_m.r], gB2[:, :proj_m.r]]
            per_layer.append(lg)
            ctx.saved[li] = None                                         # free this layer's activations
        for lg in reversed(per_layer):
            grads += lg
        grads = [g.to(dt) if g is not None else None for g, dt in zip(grads, ctx.param_dtypes)]
        return (None, None, *grads)


def _pad64(t: torch.Tensor) -> torch.Tensor:
    """Zero-pad the contraction (last) dimension to a multiple of 64 (the GEMM kernels' K granule)."""
    k = t.size(-1)
    if k % 64 == 0:
        return t.contiguous()
    out = t.new_zeros((*t.shape[:-1], -(-k // 64) * 64))
    out[..., :k] = t
    return out


class _Ctx:
    """Stands in for autograd's ctx when the node is driven by hand (GraphedTrainStep)."""


class GraphedTrainStep:
    """LoRA micro-steps (micro-batch 1, as finetune/ger.py runs them) — forward, the reference's chunked cross entropy
    (mean over ALL T-1 positions, Q5), backward, accumulation into the flat gradient bucket — captured ONCE per
    (sequences, padded length) as a hipGraph and replayed.  The eager autograd path issues ~1500 launches from Python
    per micro-step and is host-bound (48 ms at T = 560); the replay is one launch.

    PACKING: a call may carry P sequences = P micro-batches of the same accumulation window.  They run as ONE packed
    forward / backward (var-len causal attention keeps the sequences apart, every other kernel is row-wise), so the
    GEMMs see M = P x T rows and take the 256-tile kernels instead of walking K with 6-27 blocks on 256 CUs
    (M = 560: 0.04 of the MFMA peak).  `micro_batch_size 1` semantics are kept: each sequence's loss is its own mean
    over its own T_i - 1 positions (Q5), each is scaled by the same `loss_scale` (1 / accumulation), and — every op
    after the loss being linear in the per-row gradient — what is added to the bucket is the SUM of the P sequential
    micro-steps' gradients, up to the fp32 summation order of the token contraction.

    Sequences are right-padded to a common multiple of `pad_to` with ignored positions: causal attention and the
    row-wise kernels make real positions independent of the padding, ignored rows get zero gradient."""

    def __init__(self, model: GPT, bucket, pad_to: int = 64, max_graphs: int = 8) -> None:
        self.model, self.bucket, self.pad_to = model, bucket, pad_to
        self.params = lora_parameters(model)
        n = len(self.params)
        assert [id(p) for p in self.params] == [id(p) for p in bucket.params[:n]], "bucket must start with lora_parameters(model) in order"
        # Graph cache.  A captured micro-step holds every saved activation of its P x T_pad rows (~1.3 MB per row
        # on TinyLlama: ~6 GB for 8 x 576); ragged data produce tens of (T_pad, n_rows) keys.  So: (1) ALL graphs capture into ONE
        # memory pool — they never replay concurrently and nothing allocated inside `_body` outlives it, so the pool is as large as
        # the largest graph, not their sum; (2) the cache is an LRU of at most `max_graphs` entries (an evicted key is captured
        # again when it comes back: ~1 s); (3) the head's row count is rounded to 256 so fewer keys exist.
        import collections
        self._graphs = collections.OrderedDict()
        self.max_graphs = max(1, int(max_graphs))
        self._pool = None
        self.captures = 0

    def _body(self, st) -> None:
        model, V = self.model, self.model.config.padded_vocab_size
        ctx = _Ctx()
        ctx.hidden_only = True
        model._row_tail_override = st["tail"] if model.cpu_rsqrt_vec_width else None
        try:
            xf = _DecoderFn.forward(ctx, model, st["ids"], *self.params)        # ln_f(x) of every row, [P * T_pad, d]
        finally:
            model._row_tail_override = None
        # lm_head, cross entropy and their backward on the rows that carry a target (`rows`: those first, in order, then ignored
        # rows up to a static count).  A row without a target has loss 0 and a zero logit gradient (ignore_index = -1,
        # ger/utils.py:424-463), so the scattered results are bit for bit what the all-rows pass produced — the fine-tune data
        # label only the response (~50 of ~560 positions), and the V = 32000 head is the widest product of the step.
        rows = st["rows"]
        fz = _frozen(model)
        lg = ops.linear(xf.index_select(0, rows), model.lm_head.linear.weight.data, epilogue=ops.EPI_ADAPTER,
                        scale=model.lm_head.adapter_scale.data, bias=model.lm_head.adapter_bias.data)
        tg_c = st["targets"].index_select(0, rows)
        per_c, lse = ops.cross_entropy_fwd(lg, tg_c)
        per = torch.zeros_like(st["grow"]).index_copy_(0, rows, per_c.to(torch.float32))
        st["loss"].copy_(per.view(st["ids"].size(0), -1).sum(1) * st["inv_count"])
        dz = ops.cross_entropy_bwd(lg, tg_c, lse, st["grow"].index_select(0, rows)).to(BF)
        if not fz.scale_is_one:
            dz = dz * model.lm_head.adapter_scale.data
        dxf = torch.zeros_like(xf).index_copy_(0, rows, ops.linear(dz.contiguous(), fz.lm_T))
        # the backward accumulates rank-16 all-enabled adapters' gradients straight into the bucket's views (the token contraction's
        # last step is `out += scale * sum` either way: the same bits as computing them apart and adding), and returns None for those
        off, views = 0, []
        for p in self.params:
            views.append(self.bucket.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        ctx.grad_out = None if os.environ.get("DUALHYP_NO_DIRECT_GRADS") else {id(p): v for p, v in zip(self.params, views)}      # (the variable: same-box A/B)
        grads = _DecoderFn.backward(ctx, dxf)[2:]
        for v, g in zip(views, grads):
            if g is not None:
                v.add_(g.reshape(v.shape).to(torch.float32))

    def _state(self, P: int, T_pad: int, dev, n_rows: int):
        n = P * T_pad
        st = dict(rows=torch.arange(n_rows, dtype=torch.int64, device=dev),
                  ids=torch.zeros((P, T_pad), dtype=torch.int64, device=dev),
                  targets=torch.full((n,), -1, dtype=torch.int64, device=dev),
                  grow=torch.zeros(n, dtype=torch.float32, device=dev),
                  tail=torch.zeros(n, dtype=torch.uint8, device=dev),
                  inv_count=torch.zeros(P, dtype=torch.float32, device=dev),
                  loss=torch.zeros(P, dtype=torch.float32, device=dev))
        return st

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, labels: torch.Tensor, loss_scale: float = 1.0,
                 lengths: Optional[Sequence[int]] = None, n_targets: Optional[int] = None) -> torch.Tensor:
        """input_ids / labels [P, T] (right-padded: ids 0, labels -1), `lengths` the true lengths (default T for all).
        -> the P micro-steps' losses (device tensor [P], finetune/ger.py:278-281); d(loss_scale * sum of them) is ADDED to
        the bucket."""
        assert input_ids.dim() == 2 and labels.shape == input_ids.shape
        P, T = input_ids.shape
        lengths = [T] * P if lengths is None else [int(n) for n in lengths]
        assert len(lengths) == P and all(2 <= n <= T for n in lengths)
        T_pad = -(-max(lengths) // self.pad_to) * self.pad_to
        dev = input_ids.device
        Vw = self.model.cpu_rsqrt_vec_width
        # rows that carry a target (labels[:, 1:] != -1 below the sequence's end): the caller's count when it knows it (no host
        # sync), else counted here; rounded up to whole 128-row tiles — the head's GEMM stays in the tiled class whatever the count
        W = min(T, T_pad)
        if n_targets is None:
            lens_h = torch.tensor(lengths, device=labels.device).view(-1, 1)
            colh = torch.arange(1, W, device=labels.device).view(1, -1)
            n_targets = int(((labels[:, 1:W] != -1) & (colh < lens_h)).sum())
        n_rows = min(P * T_pad, max(128, -(-int(n_targets) // 256) * 256))
        gkey = (P, T_pad, Vw, n_rows)        # the rsqrt-emulation switch is resolved while capturing: part of the key
        ent = self._graphs.get(gkey)
        if ent is None:
            while len(self._graphs) >= self.max_graphs:          # least recently used first; its blocks go back to the shared pool
                _, old = self._graphs.popitem(last=False)
                old[1] = None
                old[0].clear()
            st = self._state(P, T_pad, dev, n_rows)
            ent = self._graphs[gkey] = [st, None]
        else:
            self._graphs.move_to_end(gkey)
        st = ent[0]
        st["ids"].zero_()
        st["ids"][:, :W].copy_(input_ids[:, :W])
        tg = st["targets"].view(P, T_pad)
        tg.fill_(-1)
        tg[:, : W - 1].copy_(labels[:, 1:W])
        # per sequence: positions >= T_i - 1 carry no target, the loss is the mean over its T_i - 1 positions (Q5)
        lens_d = torch.tensor(lengths, device=dev)
        col = torch.arange(T_pad, device=dev).view(1, -1)
        tg.masked_fill_(col >= (lens_d.view(-1, 1) - 1), -1)
        st["ids"].masked_fill_(col >= lens_d.view(-1, 1), 0)
        # target rows first (stable: in order), ignored rows behind them; the first n_rows of that order go through the head
        st["rows"].copy_(torch.sort((st["targets"] < 0).to(torch.int8), stable=True).indices[:n_rows])
        inv = 1.0 / (lens_d - 1).clamp_min(1).to(torch.float32)
        st["inv_count"].copy_(inv)
        st["grow"].view(P, T_pad).copy_((inv * loss_scale).view(-1, 1).expand(P, T_pad))
        if Vw:
            # Q11: a micro-batch-1 call of the reference has T_i rows; its rows past the last whole vector take the scalar loop
            st["tail"].view(P, T_pad).copy_(((col >= (lens_d // Vw * Vw).view(-1, 1)) & (col < lens_d.view(-1, 1))).to(torch.uint8))
        if ent[1] is None:
            # A caller-supplied n_targets is trusted on replays (counting on the device would be a host sync per step); it is
            # CHECKED once per graph key, here, where a capture costs a second anyway: an undercount would silently drop target
            # rows from the loss and the gradient.
            have = int((st["targets"] >= 0).sum())
            if have > n_rows:
                del self._graphs[gkey]
                raise ValueError(f"GraphedTrainStep: n_targets={n_targets} but the labels carry {have} targets (> {n_rows} head rows)")
            # warm-up outside capture (allocations, lazy initialisation), with its gradient contribution undone
            keep = self.bucket.flat.clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._body(st)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.bucket.flat.copy_(keep)
            g = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self._pool):
                self._body(st)
            self.captures += 1
            self.bucket.flat.copy_(keep)          # capture does not execute, but keep the invariant explicit
            ent[1] = g
        ent[1].replay()
        return st["loss"].clone()


def forward_train(model: GPT, idx: torch.Tensor, lm_head_chunk_size: int = 0) -> Union[torch.Tensor, List[torch.Tensor]]:
    """Differentiable no-cache forward (ger/lora.py:538-549); gradients flow to the LoRA parameters."""
    ps = lora_parameters(model)
    assert ps, "forward_train: the model has no active LoRA parameters to train"
    logits = _DecoderFn.apply(model, idx, *ps)
    if lm_head_chunk_size > 0:
        return list(logits.split(lm_head_chunk_size, dim=1))
    return logits
