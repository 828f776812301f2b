"""Tensor-level wrappers over the C ABI (include/dualhyp_hip.h).

Every function takes CUDA (ROCm) bf16/int tensors, launches on torch's current stream and
returns torch tensors.  They are thin: shape checks + pointer passing.  No fallbacks — a CPU
tensor or a missing library raises.
"""
from __future__ import annotations

import os
from typing import Sequence, Optional, Tuple

import torch

from . import _lib
from ._lib import check

EPI_PLAIN, EPI_LORA, EPI_SWIGLU, EPI_ADAPTER = 0, 1, 2, 3


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype=torch.bfloat16, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.DualHypHipError(f"{name} must live on the GPU: the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


class _Keep(list):
    """Pointer of a validated (contiguous, right dtype, on the GPU) view of `t`, with the view kept alive until the
    wrapper returns: `.contiguous()` may allocate a temporary, and a temporary freed before the launch is enqueued
    could be handed to the next argument by the caching allocator (two kernel arguments would then alias)."""

    def __call__(self, t: Optional[torch.Tensor], dtype=torch.bfloat16, name: str = "tensor"):
        if t is None:
            return None
        t = _dev(t, dtype, name)
        self.append(t)
        return t.data_ptr()


def embed(ids: torch.Tensor, wte: torch.Tensor) -> torch.Tensor:
    ids = _dev(ids.reshape(-1), torch.int64, "ids")
    wte = _dev(wte, name="wte")
    out = torch.empty((ids.numel(), wte.size(1)), dtype=torch.bfloat16, device=wte.device)
    check(_lib.load().dh_embed_bf16(_p(ids), _p(wte), _p(out), ids.numel(), wte.size(1), wte.size(0), _stream()))
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, resid: Optional[torch.Tensor] = None,
            return_sum: bool = False, row_tail: Optional[torch.Tensor] = None):
    x = _dev(x, name="x")
    w = _dev(w, name="weight")
    d = x.size(-1)
    rows = x.numel() // d
    out = torch.empty_like(x)
    s = torch.empty_like(x) if (resid is not None and return_sum) else None
    if resid is not None:
        resid = _dev(resid, name="resid")
    if row_tail is not None:
        row_tail = _dev(row_tail.reshape(-1), torch.uint8, "row_tail")
        assert row_tail.numel() == rows
    check(_lib.load().dh_rmsnorm_bf16(_p(x), _p(resid), _p(w), _p(out), _p(s), rows, d, float(eps), _p(row_tail),
                                      _stream()))
    return (out, s) if return_sum else out


def linear(x: torch.Tensor, w: torch.Tensor, *, epilogue: int = EPI_PLAIN, w2: Optional[torch.Tensor] = None,
           xa: Optional[torch.Tensor] = None, lora_b: Optional[torch.Tensor] = None, lora_scale: float = 1.0,
           splits: Optional[Tuple[int, int]] = None, scale: Optional[torch.Tensor] = None,
           bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epilogue(x @ w.T); see dh_linear_bf16 in include/dualhyp_hip.h."""
    x = _dev(x, name="x")
    w = _dev(w, name="w")
    K = x.size(-1)
    M = x.numel() // K
    N = w.size(0)
    assert w.size(1) == K, f"weight is {tuple(w.shape)}, input feature size {K}"
    y = out if out is not None else torch.empty((*x.shape[:-1], N), dtype=torch.bfloat16, device=x.device)
    s0, s1 = splits if splits is not None else (N, N)
    xa_ld = 0
    if xa is not None:
        xa = _dev(xa, name="xa")
        xa_ld = xa.size(-1)
    for t, nm in ((w2, "w2"), (lora_b, "lora_b"), (scale, "scale"), (bias, "bias"), (resid, "resid")):
        if t is not None:
            _dev(t, name=nm)
            assert t.is_contiguous(), f"{nm} must be contiguous"
    check(_lib.load().dh_linear_bf16(_p(x), _p(w), _p(y), M, N, K, epilogue, _p(w2), _p(xa), xa_ld, _p(lora_b),
                                     float(lora_scale), s0, s1, _p(scale), _p(bias), _p(resid), _stream()))
    return y


def qkv_rope_cache(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, tok_slot: torch.Tensor,
                   tok_pos: torch.Tensor, k_cache: torch.Tensor, vT_cache: torch.Tensor, n_head: int,
                   n_groups: int, k_out: Optional[torch.Tensor] = None, v_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> rotated q [n_tok, n_head, hs]; appends k / v^T into the caches in place (and, for training,
    plain copies into k_out / v_out [n_tok, n_groups, hs])."""
    k = _Keep()
    qkv = _dev(qkv, name="qkv")
    hs = k_cache.size(-1)
    s_max = k_cache.size(-2)
    n_tok = qkv.numel() // ((n_head + 2 * n_groups) * hs)
    q = torch.empty((n_tok, n_head, hs), dtype=torch.bfloat16, device=qkv.device)
    check(_lib.load().dh_qkv_rope_cache_bf16(_p(qkv), k(cos), k(sin), k(tok_slot, torch.int32),
                                             k(tok_pos, torch.int32), _p(q), _p(k_cache), _p(vT_cache), _p(k_out),
                                             _p(v_out), n_tok, n_head, n_groups, hs, s_max, _stream()))
    return q


def linear_qkv_rope_cache(x: torch.Tensor, w: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, tok_slot: torch.Tensor,
                          tok_pos: torch.Tensor, k_cache: torch.Tensor, vT_cache: torch.Tensor, n_head: int, n_groups: int,
                          xa: Optional[torch.Tensor] = None, lora_b: Optional[torch.Tensor] = None,
                          lora_scale: float = 1.0) -> torch.Tensor:
    """QKV projection (+LoRA) with rope and KV-cache append in the GEMM epilogue -> rotated q [M, n_head, hs];
    dh_linear_qkv_rope_cache_bf16 (large M only)."""
    k = _Keep()
    x = _dev(x, name="x")
    K = x.size(-1)
    M = x.numel() // K
    hs, s_max = k_cache.size(-1), k_cache.size(-2)
    q = torch.empty((M, n_head, hs), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().dh_linear_qkv_rope_cache_bf16(_p(x), k(w, name="w"), M, K, k(xa, name="xa"), 0 if xa is None else xa.size(-1),
                                                    k(lora_b, name="lora_b"), float(lora_scale), k(cos), k(sin),
                                                    k(tok_slot, torch.int32), k(tok_pos, torch.int32), _p(q), _p(k_cache),
                                                    _p(vT_cache), n_head, n_groups, hs, s_max, _stream()))
    return q


def linear_lora(x: torch.Tensor, w: torch.Tensor, lora_a: torch.Tensor, lora_b: torch.Tensor, *, lora_scale: float = 1.0,
                splits: Optional[Tuple[int, int]] = None, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x W^T + scale * bf16(x A^T) B^T (+ resid) with the down-projection computed by the library (inside the GEMM's K loop on the
    4-wave 256-tile kernel, else one more launch); lora_a [16 * nseg, K], lora_b [N, 16]; dh_linear_lora_bf16."""
    x, w, lora_a, lora_b = _dev(x, name="x"), _dev(w, name="w"), _dev(lora_a, name="lora_a"), _dev(lora_b, name="lora_b")
    K = x.size(-1)
    M, N = x.numel() // K, w.size(0)
    s0, s1 = splits if splits is not None else (N, N)
    nseg = 1 + (s0 < N) + (s1 < N)
    assert lora_a.shape == (16 * nseg, K) and lora_b.shape == (N, 16), (lora_a.shape, lora_b.shape)
    if resid is not None:
        _dev(resid, name="resid")
    y = torch.empty((*x.shape[:-1], N), dtype=torch.bfloat16, device=x.device)
    work = torch.empty((M, 16 * nseg), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().dh_linear_lora_bf16(_p(x), _p(w), _p(y), M, N, K, _p(lora_a), _p(lora_b), float(lora_scale), s0, s1, _p(resid),
                                          _p(work), _stream()))
    return y


def linear_qkv_lora_rope_cache(x: torch.Tensor, w: torch.Tensor, lora_a: torch.Tensor, lora_b: torch.Tensor, cos: torch.Tensor,
                               sin: torch.Tensor, tok_slot: torch.Tensor, tok_pos: torch.Tensor, k_cache: torch.Tensor,
                               vT_cache: torch.Tensor, n_head: int, n_groups: int, *, lora_scale: float = 1.0) -> torch.Tensor:
    """linear_qkv_rope_cache with the LoRA down-projection computed by the library (dh_linear_qkv_lora_rope_cache_bf16)."""
    k = _Keep()
    M, K = x.shape
    hs = k_cache.size(-1)
    q = torch.empty((M, n_head, hs), dtype=torch.bfloat16, device=x.device)
    work = torch.empty((M, 48), dtype=torch.bfloat16, device=x.device)
    i32 = torch.int32
    check(_lib.load().dh_linear_qkv_lora_rope_cache_bf16(k(x), k(w), M, K, k(lora_a), k(lora_b), float(lora_scale), k(cos), k(sin),
                                                         k(tok_slot, i32), k(tok_pos, i32), _p(q), _p(k_cache), _p(vT_cache), n_head,
                                                         n_groups, hs, k_cache.size(2), _p(work), _stream()))
    return q


def attn_prefill(q: torch.Tensor, k_cache: torch.Tensor, vT_cache: torch.Tensor, seq_slot: torch.Tensor,
                 q_start: torch.Tensor, q_len: torch.Tensor, kv_pos0: torch.Tensor, max_q_len: int,
                 lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    k = _Keep()
    n_tok, n_head, hs = q.shape
    n_groups, s_max = k_cache.size(1), k_cache.size(2)
    y = torch.empty((n_tok, n_head * hs), dtype=torch.bfloat16, device=q.device)
    i32 = torch.int32
    check(_lib.load().dh_attn_prefill_bf16(k(q), _p(k_cache), _p(vT_cache), k(seq_slot, i32),
                                           k(q_start, i32), k(q_len, i32), k(kv_pos0, i32), _p(y), _p(lse),
                                           seq_slot.numel(), int(max_q_len), n_head, n_groups, hs, s_max, _stream()))
    return y


def attn_decode(q: torch.Tensor, k_cache: torch.Tensor, vT_cache: torch.Tensor, seq_slot: torch.Tensor,
                kv_len: torch.Tensor) -> torch.Tensor:
    k = _Keep()
    n_seq, n_head, hs = q.shape
    n_groups, s_max = k_cache.size(1), k_cache.size(2)
    lib = _lib.load()
    work = torch.empty(lib.dh_attn_decode_work_bytes(n_seq, n_head, hs, s_max), dtype=torch.uint8, device=q.device)
    y = torch.empty((n_seq, n_head * hs), dtype=torch.bfloat16, device=q.device)
    i32 = torch.int32
    check(lib.dh_attn_decode_bf16(k(q), _p(k_cache), _p(vT_cache), k(seq_slot, i32),
                                  k(kv_len, i32), _p(y), _p(work), n_seq, n_head, n_groups, hs, s_max, _stream()))
    return y


def sample(logits: torch.Tensor, tokens: torch.Tensor, length: torch.Tensor, done: torch.Tensor, *,
           temperature: float = 1.0, top_k: Optional[int] = None, eos_id: Optional[int] = None, seed: int = 0,
           step: int = 0) -> None:
    """Append one token per sequence in place (tokens/length/done); see dh_sample_bf16."""
    k = _Keep()
    logits = _dev(logits, name="logits")
    n_seq, vocab = logits.shape
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and tokens.size(0) == n_seq
    check(_lib.load().dh_sample_bf16(_p(logits), vocab, _p(tokens), tokens.size(1), k(length, torch.int32),
                                     k(done, torch.int32), n_seq, float(temperature),
                                     0 if top_k is None else int(top_k), -1 if eos_id is None else int(eos_id),
                                     int(seed) & ((1 << 64) - 1), int(step), _stream()))


def quant_rows_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(uint8 e4m3 [rows, K], fp32 scale [rows]) of bf16 rows; dh_quant_rows_fp8."""
    x = _dev(x, name="x")
    K = x.size(-1)
    rows = x.numel() // K
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    s = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.load().dh_quant_rows_fp8(_p(x), _p(q), _p(s), rows, K, _stream()))
    return q, s


def rmsnorm_quant_fp8(x: torch.Tensor, w: torch.Tensor, eps: float, row_tail: Optional[torch.Tensor] = None):
    """(bf16 RMSNorm rows, their e4m3 bytes, fp32 scales); dh_rmsnorm_quant_fp8."""
    k = _Keep()
    x = _dev(x, name="x")
    d = x.size(-1)
    rows = x.numel() // d
    xn = torch.empty_like(x)
    q = torch.empty((rows, d), dtype=torch.uint8, device=x.device)
    s = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.load().dh_rmsnorm_quant_fp8(_p(x), k(w, name="weight"), _p(xn), _p(q), _p(s), rows, d, float(eps),
                                           k(row_tail, torch.uint8, "row_tail"), _stream()))
    return xn, q, s


def linear_fp8(xq: torch.Tensor, x_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, *, epilogue: int = EPI_PLAIN,
               w2q: Optional[torch.Tensor] = None, w2_scale: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
               bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, kernel: int = 0) -> torch.Tensor:
    """bf16 [M, N] = epilogue((xq . wq^T) * x_scale[m] * w_scale[n]) on the fp8 MFMA; dh_linear_fp8_ex (kernel: 0 by row
    count, 1 tiled, 2 streaming)."""
    k = _Keep()
    M, K = xq.shape
    N = wq.size(0)
    assert wq.size(1) == K and xq.dtype == torch.uint8 and wq.dtype == torch.uint8
    y = torch.empty((M, N), dtype=torch.bfloat16, device=xq.device)
    check(_lib.load().dh_linear_fp8_ex(k(xq, torch.uint8, "xq"), k(x_scale, torch.float32, "x_scale"), k(wq, torch.uint8, "wq"),
                                       k(w_scale, torch.float32, "w_scale"), _p(y), M, N, K, epilogue, k(w2q, torch.uint8, "w2q"),
                                       k(w2_scale, torch.float32, "w2_scale"), k(scale, name="scale"), k(bias, name="bias"),
                                       k(resid, name="resid"), int(kernel), _stream()))
    return y


def im2col3(x: torch.Tensor, ld: int, relu: bool = False) -> torch.Tensor:
    """[B*T, ld] im2col matrix of a k=3, pad=1 convolution over x [B,T,C] with a ones column at 3C; dh_im2col3_bf16."""
    x = _dev(x, name="x")
    B, T, Cc = x.shape
    out = torch.empty((B * T, ld), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().dh_im2col3_bf16(_p(x), _p(out), B, T, Cc, ld, int(relu), _stream()))
    return out


def pool_head(h: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, pool: int) -> torch.Tensor:
    """ReLU -> AvgPool1d(pool, ceil) -> Linear(H, 3) on h [B,T,H]; dh_pool_head_bf16."""
    k = _Keep()
    h = _dev(h, name="h")
    B, T, H = h.shape
    out = torch.empty((B, (T + pool - 1) // pool, 3), dtype=torch.bfloat16, device=h.device)
    check(_lib.load().dh_pool_head_bf16(_p(h), k(w, name="w"), k(bias, name="bias"), _p(out), B, T, H, int(pool),
                                        _stream()))
    return out


def pool_head_bwd(h: torch.Tensor, w: torch.Tensor, dlogits: torch.Tensor, pool: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dh bf16 [B,T,H], pooled bf16 [B*P,H]) from dlogits fp32 [B,P,3]; dh_pool_head_bwd_bf16."""
    k = _Keep()
    h = _dev(h, name="h")
    B, T, H = h.shape
    P = (T + pool - 1) // pool
    assert dlogits.shape == (B, P, 3)
    dh = torch.empty_like(h)
    pooled = torch.empty((B * P, H), dtype=torch.bfloat16, device=h.device)
    check(_lib.load().dh_pool_head_bwd_bf16(_p(h), k(w, name="w"), k(dlogits, torch.float32, "dlogits"), _p(dh), _p(pooled),
                                            B, T, H, int(pool), _stream()))
    return dh, pooled


def col2im3(dcol: torch.Tensor, B: int, T: int, C: int, pre: Optional[torch.Tensor] = None,
            mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx bf16 [B,T,C]: backward of im2col3 through the ReLU of `pre` and the dropout `mask` in front of it; dh_col2im3_bf16."""
    k = _Keep()
    dcol = _dev(dcol, name="dcol")
    assert dcol.dim() == 2 and dcol.size(0) == B * T and dcol.size(1) >= 3 * C
    dx = torch.empty((B, T, C), dtype=torch.bfloat16, device=dcol.device)
    check(_lib.load().dh_col2im3_bf16(_p(dcol), k(pre, name="pre"), k(mask, name="mask"), _p(dx), B, T, C, dcol.size(1), _stream()))
    return dx


def cross_entropy_fwd(logits: torch.Tensor, targets: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(loss_per_row fp32 [rows], lse fp32 [rows]); ignored rows (target -1) give 0; dh_cross_entropy_fwd."""
    assert logits.dtype in (torch.bfloat16, torch.float32), f"logits must be bf16 or fp32, got {logits.dtype}"
    lg = _dev(logits.reshape(-1, logits.size(-1)), logits.dtype, "logits")
    tg = _dev(targets.reshape(-1), torch.int64, "targets")
    assert tg.numel() == lg.size(0), f"{tg.numel()} targets for {lg.size(0)} rows"
    loss = torch.empty(lg.size(0), dtype=torch.float32, device=lg.device)
    lse = torch.empty_like(loss)
    check(_lib.load().dh_cross_entropy_fwd(_p(lg), int(lg.dtype == torch.float32), _p(tg), _p(loss), _p(lse), lg.size(0),
                                           lg.size(1), _stream()))
    return loss, lse


def cross_entropy_bwd(logits: torch.Tensor, targets: torch.Tensor, lse: torch.Tensor, grad_row: torch.Tensor) -> torch.Tensor:
    """dlogits (same shape / dtype as logits); dh_cross_entropy_bwd."""
    k = _Keep()
    lg = _dev(logits.reshape(-1, logits.size(-1)), logits.dtype, "logits")
    tg = _dev(targets.reshape(-1), torch.int64, "targets")
    out = torch.empty_like(lg)
    check(_lib.load().dh_cross_entropy_bwd(_p(lg), int(lg.dtype == torch.float32), _p(tg), k(lse, torch.float32, "lse"),
                                           k(grad_row, torch.float32, "grad_row"), _p(out), lg.size(0), lg.size(1),
                                           _stream()))
    return out.view(logits.shape)


def linear_partial(x: torch.Tensor, w: torch.Tensor, w_ext: Optional[torch.Tensor] = None, ksplit: int = 1) -> torch.Tensor:
    """fp32 partial sums [ksplit, M, n_main+n_ext] of x @ [w; w_ext].T (decode rows, M <= 4096); dh_linear_partial_bf16."""
    x, w = _dev(x, name="x"), _dev(w, name="w")
    K = x.size(-1)
    M = x.numel() // K
    n_ext = 0 if w_ext is None else _dev(w_ext, name="w_ext").size(0)
    y = torch.empty((ksplit, M, w.size(0) + n_ext), dtype=torch.float32, device=x.device)
    check(_lib.load().dh_linear_partial_bf16(_p(x), _p(w), _p(w_ext), _p(y), M, w.size(0), n_ext, K, ksplit, _stream()))
    return y


def linear_partial_pairs(x: torch.Tensor, w: torch.Tensor, w_ext: Optional[torch.Tensor] = None, ksplit: int = 1) -> torch.Tensor:
    """fp32 [(ksplit+1)//2, M, n_main+n_ext]: sums of adjacent pairs of `linear_partial`'s slices, from the tiled split-K
    kernel; dh_linear_partial_pairs_bf16."""
    x, w = _dev(x, name="x"), _dev(w, name="w")
    K = x.size(-1)
    M = x.numel() // K
    n_ext = 0 if w_ext is None else _dev(w_ext, name="w_ext").size(0)
    y = torch.empty(((ksplit + 1) // 2, M, w.size(0) + n_ext), dtype=torch.float32, device=x.device)
    check(_lib.load().dh_linear_partial_pairs_bf16(_p(x), _p(w), _p(w_ext), _p(y), M, w.size(0), n_ext, K, ksplit, _stream()))
    return y


def combine_partials(parts: torch.Tensor, pairs: bool = True) -> torch.Tensor:
    """The decode family's K-slice combine order on the fp32 partials [n, M, N] (include/dualhyp_hip.h): adjacent slices in
    pairs, pair sums in index order (`pairs=False`: the inputs already are pair sums).  torch fp32 adds: test helper."""
    if pairs:
        n = parts.size(0)
        parts = torch.stack([parts[i] + parts[i + 1] if i + 1 < n else parts[i] for i in range(0, n, 2)])
    out = parts[0].clone()
    for p in parts[1:]:
        out = out + p
    return out


def linear_chain(x: torch.Tensor, w: torch.Tensor, w_ext: Optional[torch.Tensor] = None, ksplit: int = 1) -> torch.Tensor:
    """fp32 [M, n_main+n_ext]: the ksplit slices of `linear_partial` combined in the family's order (`combine_partials`)
    by one launch of the tiled kernel (M >= 65); dh_linear_chain_bf16."""
    x, w = _dev(x, name="x"), _dev(w, name="w")
    K = x.size(-1)
    M = x.numel() // K
    n_ext = 0 if w_ext is None else _dev(w_ext, name="w_ext").size(0)
    y = torch.empty((M, w.size(0) + n_ext), dtype=torch.float32, device=x.device)
    check(_lib.load().dh_linear_chain_bf16(_p(x), _p(w), _p(w_ext), _p(y), M, w.size(0), n_ext, K, ksplit, _stream()))
    return y


def finish_norm(h32: torch.Tensor, d: int, x_resid: torch.Tensor, w_norm: torch.Tensor, eps: float,
                lora_b: Optional[torch.Tensor] = None, lora_scale: float = 1.0,
                row_tail: Optional[torch.Tensor] = None, pairs: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(x_out, xn_out) = LoRA finish + residual + RMSNorm of fp32 partials (`pairs`: they are K-slices, to be added in
    adjacent pairs first; False: pair sums / the total); dh_finish_norm_bf16."""
    k = _Keep()
    n_part, rows, ld = h32.shape
    x_resid = _dev(x_resid, name="x_resid")
    x_out, xn_out = torch.empty_like(x_resid), torch.empty_like(x_resid)
    check(_lib.load().dh_finish_norm_bf16(_p(h32), n_part, int(pairs), rows, d, ld - d, _p(lora_b), float(lora_scale), _p(x_resid),
                                          k(w_norm), _p(x_out), _p(xn_out), float(eps), _p(row_tail), _stream()))
    return x_out, xn_out


def attn_decode_fused(qkv32: torch.Tensor, qkv_dim: int, lora_b: Optional[torch.Tensor], lora_scale: float,
                      splits: Tuple[int, int], cos: torch.Tensor, sin: torch.Tensor, seq_slot: torch.Tensor,
                      kv_len: torch.Tensor, k_cache: torch.Tensor, vT_cache: torch.Tensor, n_head: int,
                      pairs: bool = True) -> torch.Tensor:
    """Decode-step attention sub-layer from fp32 qkv partials (`pairs` as in finish_norm); dh_attn_decode_fused_bf16."""
    k = _Keep()
    n_part, n_seq, ld = qkv32.shape
    n_groups, s_max, hs = k_cache.size(1), k_cache.size(2), k_cache.size(3)
    y = torch.empty((n_seq, n_head * hs), dtype=torch.bfloat16, device=qkv32.device)
    i32 = torch.int32
    check(_lib.load().dh_attn_decode_fused_bf16(_p(qkv32), n_part, int(pairs), n_seq, qkv_dim, ld - qkv_dim, _p(lora_b),
                                                float(lora_scale), splits[0], splits[1], k(cos), k(sin),
                                                k(seq_slot, i32), k(kv_len, i32), _p(k_cache), _p(vT_cache),
                                                _p(y), n_head, n_groups, hs, s_max, _stream()))
    return y


def kcache_to_plain(kc: torch.Tensor) -> torch.Tensor:
    """K cache [slot, group, s_max, hs] as stored (MFMA-fragment order, csrc/common.h kfrag_off)
    -> plain [slot, group, key, channel].  Test/debug helper."""
    B, G, S, hs = kc.shape
    v = kc.reshape(B, G, S // 32, hs // 16, 2, 32, 8)            # tile, ks, lh, lr, e
    return v.permute(0, 1, 2, 5, 3, 4, 6).reshape(B, G, S, hs)


def vcache_to_plain(vt: torch.Tensor) -> torch.Tensor:
    """V^T cache [slot, group, hs, s_max] as stored (fragment order, vfrag_off) -> plain
    [slot, group, channel, key].  Test/debug helper."""
    B, G, hs, S = vt.shape
    v = vt.reshape(B, G, S // 32, hs // 32, 2, 2, 32, 2, 4)      # tile, dt, s2, lh, lr, jh, jl
    return v.permute(0, 1, 3, 6, 2, 4, 7, 5, 8).reshape(B, G, hs, S)


# ------------------------------------------------------------------------------------------ training backward
def dropout(x: torch.Tensor, p: float, seed: int, call_id: int, step: Optional[torch.Tensor] = None):
    """(bf16(x * m), m): LoRA-branch dropout with the mask drawn in the kernel (Philox keyed by seed / call_id, counted by the
    device counter `step` — an int64 tensor of one element — and the element index); dh_dropout_bf16."""
    x = _dev(x, name="x")
    y, m = torch.empty_like(x), torch.empty_like(x)
    check(_lib.load().dh_dropout_bf16(_p(x), _p(y), _p(m), x.numel(), float(p), int(seed) & (2 ** 64 - 1), int(call_id) & 0xffffffff,
                                      _p(step) if step is not None else None, _stream()))
    return y, m


def swiglu_fwd(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    k = _Keep()
    act = torch.empty_like(g)
    check(_lib.load().dh_swiglu_fwd_bf16(k(g), k(u), _p(act), g.numel(), _stream()))
    return act


def linear_swiglu_train(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor):
    """-> (act, g, u): the training forward of fc_1 / fc_2 (ger/model.py:313-315) in one GEMM launch that also keeps the rounded
    pre-activations for the backward; the bits of linear(x, w1), linear(x, w2), swiglu_fwd(g, u)."""
    k = _Keep()
    M, K = x.shape
    I = w1.size(0)
    act = torch.empty((M, I), dtype=torch.bfloat16, device=x.device)
    g, u = torch.empty_like(act), torch.empty_like(act)
    check(_lib.load().dh_linear_swiglu_train_bf16(k(x), k(w1), k(w2), _p(act), _p(g), _p(u), M, I, K, _stream()))
    return act, g, u


def linear_mul(x: torch.Tensor, w: torch.Tensor, mul: torch.Tensor) -> torch.Tensor:
    """bf16(bf16(x @ w.T) * mul): the multiply inside the GEMM's epilogue for large shapes (dh_linear_mul_bf16), else linear then *."""
    M, K = x.shape
    N = w.size(0)
    if (M >= 256 and N >= 256 and -(-M // 256) * -(-N // 256) >= 128 and mul.is_contiguous() and mul.shape == (M, N)
            and not os.environ.get("DUALHYP_NO_LINEAR_MUL")):        # (the variable: same-box A/B of the fused multiply)
        k = _Keep()
        y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
        check(_lib.load().dh_linear_mul_bf16(k(x), k(w), _p(y), M, N, K, k(mul), _stream()))
        return y
    return linear(x, w) * mul


def swiglu_bwd(dact: torch.Tensor, g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    k = _Keep()
    rows, I = g.shape
    out = torch.empty((rows, 2 * I), dtype=torch.bfloat16, device=g.device)
    check(_lib.load().dh_swiglu_bwd_bf16(k(dact), k(g), k(u), _p(out), rows, I, _stream()))
    return out


def rmsnorm_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, eps: float, dres: Optional[torch.Tensor] = None) -> torch.Tensor:
    k = _Keep()
    d = x.size(-1)
    dx = torch.empty_like(x)
    check(_lib.load().dh_rmsnorm_bwd_bf16(k(dy), k(x), k(w), k(dres, name="dres"), _p(dx), x.numel() // d, d,
                                          float(eps), _stream()))
    return dx


def qkv_rope_bwd(dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                 tok_pos: torch.Tensor) -> torch.Tensor:
    k = _Keep()
    n_tok, n_head, hs = dq.shape
    n_groups = dk.size(1)
    out = torch.empty((n_tok, (n_head + 2 * n_groups) * hs), dtype=torch.bfloat16, device=dq.device)
    check(_lib.load().dh_qkv_rope_bwd_bf16(k(dq), k(dk), k(dv), k(cos), k(sin),
                                           k(tok_pos, torch.int32), _p(out), n_tok, n_head, n_groups, hs, _stream()))
    return out


def tn_accum(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, scale: float = 1.0, accumulate: bool = True,
             splits: Optional[Tuple[int, int]] = None) -> None:
    """out[M,N] (+)= scale * a[T,M].T @ b[T,N]; a/b may be column slices of wider row-major matrices.
    splits = (s0, s1): out[M,16], rows m of segment seg(m) = (m >= s0) + (m >= s1) contract with b[:, 16 seg : 16 seg + 16]."""
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.dtype == torch.float32 and out.stride(1) == 1
    T, M = a.shape
    N = b.size(1) if splits is None else 16
    lib = _lib.load()
    wb = lib.dh_tn_accum_work_bytes(T, M, N)       # packed micro-batches: the token loop is split over the grid
    work = torch.empty(wb // 4, dtype=torch.float32, device=a.device) if wb else None
    if splits is not None:
        check(lib.dh_tn_accum_seg_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                                      T, M, int(splits[0]), int(splits[1]), float(scale), int(accumulate), _p(work), _stream()))
        return
    check(lib.dh_tn_accum_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                              T, M, N, float(scale), int(accumulate), _p(work), _stream()))


def attn_bwd_plan(q_start: torch.Tensor, q_len: torch.Tensor, n_tok: int, lens: Sequence[int]) -> dict:
    """Index tensors of the padded, fragment-ordered copies dh_attn_bwd_bf16 reads — the same for every layer of a micro-step, so the
    training step builds them once: pad_start[i] (sequence i starts at a multiple of 32), n_pad, and the inverse map
    pad_tok[p] = token of padded position p (-1 = padding)."""
    dev = q_len.device
    n_pad = sum(-(-n // 32) * 32 for n in lens)
    pads_dev = (q_len + 31) // 32 * 32
    pad_start = (torch.cumsum(pads_dev, 0) - pads_dev).to(torch.int32)
    tok_seq = torch.repeat_interleave(torch.arange(len(lens), dtype=torch.int64, device=dev), q_len.to(torch.int64), output_size=n_tok)
    tok = torch.arange(n_tok, dtype=torch.int64, device=dev)
    pp = pad_start.to(torch.int64)[tok_seq] + (tok - q_start.to(torch.int64)[tok_seq])
    pad_tok = torch.full((n_pad,), -1, dtype=torch.int32, device=dev)
    pad_tok[pp] = tok.to(torch.int32)
    return {"n_pad": n_pad, "pad_start": pad_start, "pad_tok": pad_tok, "n_seq": len(lens)}


def attn_bwd(q, k, v, out, dout, lse, q_start, q_len, max_q_len: int, lens: Optional[Sequence[int]] = None, plan: Optional[dict] = None):
    """-> (dq, dk, dv) of the causal GQA attention of a packed batch (sequences attend to themselves).
    `lens` = q_len on the host; when given nothing here touches the host (hipGraph capture of a training step).
    `plan` = attn_bwd_plan(...) of the same batch (built once per micro-step by the caller; built here when absent)."""
    keep = _Keep()
    n_tok, H, hs = q.shape
    G = k.size(1)
    lib = _lib.load()
    dev = q.device
    if plan is None:
        if lens is None:
            lens = q_len.tolist()
        plan = attn_bwd_plan(q_start, q_len, n_tok, lens)
    n_pad, pad_start, pad_tok = plan["n_pad"], plan["pad_start"], plan["pad_tok"]
    dout = _dev(dout.reshape(n_tok, H, hs))
    dsum = torch.empty((n_tok, H), dtype=torch.float32, device=dev)
    check(lib.dh_rowdot_f32(_p(dout), keep(out.reshape(n_tok, H, hs)), _p(dsum), n_tok * H, hs, _stream()))

    def tpad(src, heads):
        dst = torch.empty((heads, hs, n_pad), dtype=torch.bfloat16, device=dev)      # padding is written (zeros) by the kernel
        check(lib.dh_transpose_frag_bf16(keep(src), _p(dst), _p(pad_tok), heads, hs, n_pad, _stream()))
        return dst
    need = lib.dh_attn_bwd_transposes(H, G, hs, n_pad)      # hs 64: the dk/dv kernel transposes its q / dO tiles in LDS
    qT, doT = (tpad(q, H), tpad(dout, H)) if need & 1 else (None, None)
    kT = tpad(k, G)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    check(lib.dh_attn_bwd_bf16(keep(q, name="q"), keep(k, name="k"), keep(v, name="v"), _p(dout), _p(qT), _p(doT), _p(kT), _p(lse), _p(dsum), _p(q_start),
                               _p(q_len), _p(pad_start), _p(dq), _p(dk), _p(dv), plan["n_seq"], int(max_q_len), H, G, hs,
                               n_pad, _stream()))
    return dq, dk, dv
