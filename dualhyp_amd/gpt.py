"""Host-side mirror of the reference's LoRA decoder (ger/lora.py GPT + ger/model.py).

Same constructor, parameter names, state-dict keys and `forward()` signature as
`ger.lora.GPT` (ger/lora.py:475-549) so `finetune.ger` / `inference.ger`-style callers work
unchanged — but the modules only HOLD parameters.  All arithmetic runs in libdualhyp_hip.so:
`GPT.forward` hands the packed batch to the native engine (dualhyp_amd/csrc/engine.hip), which
sequences the HIP kernels for every layer.  There is no eager/PyTorch fallback: on CPU tensors
or without the built library, forward raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _lib, ops
from .config import Config

KVCache = Tuple[torch.Tensor, torch.Tensor]
RoPECache = Tuple[torch.Tensor, torch.Tensor]


def build_rope_cache(seq_len: int, n_elem: int, dtype: torch.dtype = torch.bfloat16, device="cpu",
                     base: int = 10000, condense_ratio: int = 1) -> RoPECache:
    """cos/sin tables [seq_len, n_elem] (ger/model.py:319-346).  Built once on the host with the
    same fp32 formula as the reference and rounded to bf16 whatever the model dtype (quirk Q1),
    then moved to `device`; the kernels only read them."""
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2) / n_elem))
    ang = torch.outer(torch.arange(seq_len) / condense_ratio, theta).repeat(1, 2)
    cos, sin = torch.cos(ang), torch.sin(ang)
    if dtype == torch.bfloat16:
        cos, sin = cos.bfloat16(), sin.bfloat16()
    elif dtype in (torch.float16, torch.int8):
        cos, sin = cos.half(), sin.half()
    return cos.to(device), sin.to(device)


def map_old_state_dict_weights(state_dict: Dict, mapping: Dict[str, str], prefix: str) -> Dict:
    """Rename pre-LoRA checkpoint keys (ger/utils.py:466-472)."""
    for old, new in mapping.items():
        k = prefix + old
        if k in state_dict:
            state_dict[prefix + new] = state_dict.pop(k)
    return state_dict


# ------------------------------------------------------------------------------------------ modules
class RMSNorm(nn.Module):
    """ger/rmsnorm.py: weight-only RMS norm, computed in the storage dtype by dh_rmsnorm_bf16."""

    def __init__(self, size: int, dim: int = -1, eps: float = 1e-5) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(size))
        self.eps = eps
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.rmsnorm(x, self.weight, self.eps)

    def reset_parameters(self) -> None:
        nn.init.ones_(self.weight)


class _FrozenLinear(nn.Module):
    """Parameter holder for a bias-free dense layer ('linear.weight' in the state dict)."""

    def __init__(self, in_features: int, out_features: int) -> None:
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class AdapterV2Linear(nn.Module):
    """lm_head wrapper: scale * (W x + bias) (ger/lora.py:63-75); scale=1, bias=0, frozen."""

    def __init__(self, in_features: int, out_features: int, **_: Any) -> None:
        super().__init__()
        self.linear = _FrozenLinear(in_features, out_features)
        self.adapter_bias = nn.Parameter(torch.zeros(out_features), requires_grad=False)
        self.adapter_scale = nn.Parameter(torch.ones(out_features), requires_grad=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear(x, self.linear.weight, epilogue=ops.EPI_ADAPTER, scale=self.adapter_scale,
                          bias=self.adapter_bias)

    def reset_parameters(self) -> None:
        nn.init.zeros_(self.adapter_bias)
        nn.init.ones_(self.adapter_scale)


class LoRALayer(nn.Module):
    def __init__(self, r: int, lora_alpha: int, lora_dropout: float) -> None:
        super().__init__()
        assert r >= 0
        self.r, self.lora_alpha, self.lora_dropout_p = r, lora_alpha, float(lora_dropout)
        self.merged = False


def _pad_rank(t: torch.Tensor, r: int, dim: int) -> torch.Tensor:
    """Zero-pad the rank axis to 16 (what the rank-16 MFMA epilogue consumes); exact."""
    if r == 16:
        return t
    shape = list(t.shape)
    shape[dim] = 16
    out = t.new_zeros(shape)
    out.narrow(dim, 0, r).copy_(t)
    return out


class LoRALinear(LoRALayer):
    """y = W x + (alpha/r) B A x (ger/lora.py:103-166); r == 0 is a plain linear."""

    def __init__(self, in_features: int, out_features: int, r: int = 0, lora_alpha: int = 1,
                 lora_dropout: float = 0.0, **_: Any) -> None:
        super().__init__(r, lora_alpha, lora_dropout)
        self.linear = _FrozenLinear(in_features, out_features)
        if r > 0:
            self.lora_A = nn.Parameter(torch.zeros(r, in_features))
            self.lora_B = nn.Parameter(torch.zeros(out_features, r))
            self.scaling = lora_alpha / r
            self.reset_parameters()

    def reset_parameters(self) -> None:
        if hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    @property
    def lora_active(self) -> bool:
        return self.r > 0 and not self.merged and hasattr(self, "lora_A")

    def padded_lora(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(A [16, in], B [out, 16]) contiguous, rank zero-padded."""
        bf = torch.bfloat16   # LoRA parameters may be fp32 masters while fine-tuning; kernels take bf16
        return (_pad_rank(self.lora_A.data.to(bf), self.r, 0).contiguous(),
                _pad_rank(self.lora_B.data.to(bf), self.r, 1).contiguous())

    def merge(self) -> None:
        """W += (B A) * scaling (ger/lora.py:152-157)."""
        if self.r > 0 and not self.merged:
            self.linear.weight.data += (self.lora_B.data @ self.lora_A.data) * self.scaling
            self.merged = True

    def forward(self, x: torch.Tensor, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not self.lora_active:
            return ops.linear(x, self.linear.weight, resid=resid)
        A, B = self.padded_lora()
        xa = ops.linear(x, A)
        return ops.linear(x, self.linear.weight, epilogue=ops.EPI_LORA, xa=xa, lora_b=B,
                          lora_scale=self.scaling, resid=resid)


class LoRAQKVLinear(LoRALinear):
    """Fused q/k/v projection with per-matrix LoRA (ger/lora.py:169-402).  The base weight rows
    are group-interleaved, the LoRA delta is laid out contiguously [Q|K|V] and added as is
    (quirk Q2) — dh_linear_bf16's `splits` reproduce that."""

    def __init__(self, in_features: int, out_features: int, n_head: int, n_query_groups: int, r: int = 0,
                 lora_alpha: int = 1, lora_dropout: float = 0.0,
                 enable_lora: Union[bool, Tuple[bool, bool, bool]] = False, **_: Any) -> None:
        LoRALayer.__init__(self, r, lora_alpha, lora_dropout)
        self.linear = _FrozenLinear(in_features, out_features)
        self.n_head, self.n_query_groups = n_head, n_query_groups
        if isinstance(enable_lora, bool):
            enable_lora = [enable_lora] * 3
        assert len(enable_lora) == 3
        self.enable_lora = list(enable_lora)
        self.kv_embd_size = in_features // (n_head // n_query_groups)
        if r > 0 and any(self.enable_lora):
            shapes = (in_features * self.enable_lora[0], self.kv_embd_size * self.enable_lora[1],
                      self.kv_embd_size * self.enable_lora[2])
            self.qkv_shapes = [s for s in shapes if s]
            self.lora_A = nn.Parameter(torch.zeros(r * sum(self.enable_lora), in_features))
            self.lora_B = nn.Parameter(torch.zeros(sum(self.qkv_shapes), r))
            self.scaling = lora_alpha / r
            self.reset_parameters()

    @property
    def splits(self) -> Tuple[int, int]:
        d = self.linear.in_features
        return d, d + self.kv_embd_size

    def padded_lora(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(A [48, in], B [out, 16]); disabled q/k/v segments are zero rows, which reproduces
        zero_pad (ger/lora.py:272-312) exactly."""
        bf = torch.bfloat16
        if self.r == 16 and all(self.enable_lora):
            return self.lora_A.data.to(bf).contiguous(), self.lora_B.data.to(bf).contiguous()
        d_in, r = self.linear.in_features, self.r
        A = torch.zeros(48, d_in, dtype=bf, device=self.lora_A.device)
        B = torch.zeros(self.linear.out_features, 16, dtype=bf, device=self.lora_B.device)
        seg_rows = (d_in, self.kv_embd_size, self.kv_embd_size)
        a_off = b_off = row0 = 0
        for seg, en in enumerate(self.enable_lora):
            if en:
                A[16 * seg:16 * seg + r] = self.lora_A.data[a_off:a_off + r]
                B[row0:row0 + seg_rows[seg], :r] = self.lora_B.data[b_off:b_off + seg_rows[seg]]
                a_off += r
                b_off += seg_rows[seg]
            row0 += seg_rows[seg]
        return A, B

    def delta_w(self) -> torch.Tensor:
        """Dense (out, in) LoRA update before scaling, contiguous [Q|K|V] rows (ger/lora.py:357-364)."""
        A, B = self.padded_lora()
        d_in = self.linear.in_features
        segs = (d_in, self.kv_embd_size, self.kv_embd_size)
        parts, row0 = [], 0
        for seg, n in enumerate(segs):
            parts.append(B[row0:row0 + n].float() @ A[16 * seg:16 * seg + 16].float())
            row0 += n
        return torch.cat(parts, dim=0)

    def merge(self) -> None:
        if self.r > 0 and any(self.enable_lora) and not self.merged:
            self.linear.weight.data += (self.delta_w() * self.scaling).to(self.linear.weight.dtype)
            self.merged = True

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not (self.lora_active and any(self.enable_lora)):
            return ops.linear(x, self.linear.weight)
        A, B = self.padded_lora()
        xa = ops.linear(x, A)
        return ops.linear(x, self.linear.weight, epilogue=ops.EPI_LORA, xa=xa, lora_b=B,
                          lora_scale=self.scaling, splits=self.splits)


class CausalSelfAttention(nn.Module):
    def __init__(self, config: Config) -> None:
        super().__init__()
        shape = (config.n_head + 2 * config.n_query_groups) * config.head_size
        self.attn = LoRAQKVLinear(config.n_embd, shape, n_head=config.n_head, n_query_groups=config.n_query_groups,
                                  r=config.r, lora_alpha=config.alpha, lora_dropout=config.dropout,
                                  enable_lora=(config.to_query, config.to_key, config.to_value))
        self.proj = LoRALinear(config.n_embd, config.n_embd, r=(config.r if config.to_projection else 0),
                               lora_alpha=config.alpha, lora_dropout=config.dropout)
        self.config = config

    def _load_from_state_dict(self, state_dict: Dict, prefix: str, *args: Any, **kwargs: Any) -> None:
        mapping = {"attn.weight": "attn.linear.weight", "proj.weight": "proj.linear.weight"}
        super()._load_from_state_dict(map_old_state_dict_weights(state_dict, mapping, prefix), prefix, *args, **kwargs)


class LLaMAMLP(nn.Module):
    def __init__(self, config: Config) -> None:
        super().__init__()
        r = config.r if config.to_mlp else 0
        self.fc_1 = LoRALinear(config.n_embd, config.intermediate_size, r=r, lora_alpha=config.alpha)
        self.fc_2 = LoRALinear(config.n_embd, config.intermediate_size, r=r, lora_alpha=config.alpha)
        self.proj = LoRALinear(config.intermediate_size, config.n_embd, r=r, lora_alpha=config.alpha)

    def forward(self, x: torch.Tensor, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
        """proj(silu(fc_1 x) * fc_2 x) (ger/model.py:312-316) as two launches."""
        act = ops.linear(x, self.fc_1.linear.weight, epilogue=ops.EPI_SWIGLU, w2=self.fc_2.linear.weight)
        return ops.linear(act, self.proj.linear.weight, resid=resid)

    def _load_from_state_dict(self, state_dict: Dict, prefix: str, *args: Any, **kwargs: Any) -> None:
        mapping = {f"{n}.weight": f"{n}.linear.weight" for n in ("fc_1", "fc_2", "proj")}
        super()._load_from_state_dict(map_old_state_dict_weights(state_dict, mapping, prefix), prefix, *args, **kwargs)


class Block(nn.Module):
    def __init__(self, config: Config, block_idx: int = 0) -> None:
        super().__init__()
        self.norm_1 = RMSNorm(config.n_embd, eps=config.norm_eps)
        self.attn = CausalSelfAttention(config)
        self.norm_2 = RMSNorm(config.n_embd, eps=config.norm_eps)
        self.mlp = LLaMAMLP(config)
        self.config = config


# ------------------------------------------------------------------------------------------ engine handle
class _Engine:
    """Owns a dh_engine* and the derived (rank-padded) LoRA tensors it points at."""

    def __init__(self, model: "GPT", max_batch: int, s_max: int, max_tokens: int) -> None:
        cfg = model.config
        lib = _lib.load()
        self.keep: List[torch.Tensor] = []      # tensors whose storage the engine references
        layers = (_lib.LayerWeights * cfg.n_layer)()

        def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
            if t is None:
                return None
            if not t.is_cuda or t.dtype != torch.bfloat16:
                raise _lib.DualHypHipError(
                    "GPT parameters must be bf16 tensors on the GPU for the HIP path "
                    f"(found {t.dtype} on {t.device}); there is no CPU fallback")
            t = t.contiguous()
            self.keep.append(t)
            return t.data_ptr()

        fp8 = bool(getattr(model, "fp8", False))

        def wptr(lin) -> Optional[int]:
            """bf16 weight, or the e4m3 bytes of a quantised layer (dualhyp_amd.quant)"""
            if not fp8:
                return ptr(lin.weight.data)
            t = lin.weight_fp8
            assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
            self.keep.append(t)
            return t.data_ptr()

        def sptr(lin) -> Optional[int]:
            if not fp8:
                return None
            t = lin.weight_scale
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
            self.keep.append(t)
            return t.data_ptr()

        scale = 0.0
        for l, blk in enumerate(model.transformer.h):
            w = layers[l]
            w.norm_1, w.norm_2 = ptr(blk.norm_1.weight.data), ptr(blk.norm_2.weight.data)
            qkv, proj = blk.attn.attn, blk.attn.proj
            w.attn_w, w.proj_w = wptr(qkv.linear), wptr(proj.linear)
            w.attn_ws, w.proj_ws = sptr(qkv.linear), sptr(proj.linear)
            if qkv.lora_active and any(qkv.enable_lora):
                A, B = qkv.padded_lora()
                w.attn_lora_a, w.attn_lora_b = ptr(A), ptr(B)
                scale = qkv.scaling
            if proj.lora_active:
                A, B = proj.padded_lora()
                w.proj_lora_a, w.proj_lora_b = ptr(A), ptr(B)
                scale = proj.scaling
            w.fc_1, w.fc_2 = wptr(blk.mlp.fc_1.linear), wptr(blk.mlp.fc_2.linear)
            w.mlp_proj = wptr(blk.mlp.proj.linear)
            w.fc_1_ws, w.fc_2_ws, w.mlp_proj_ws = sptr(blk.mlp.fc_1.linear), sptr(blk.mlp.fc_2.linear), sptr(blk.mlp.proj.linear)
        cos, sin = model.rope_cache
        desc = _lib.ModelDesc(
            n_layer=cfg.n_layer, n_head=cfg.n_head, n_groups=cfg.n_query_groups, head_size=cfg.head_size,
            n_embd=cfg.n_embd, intermediate=cfg.intermediate_size, vocab=model.lm_head.adapter_scale.size(0),
            block_size=cfg.block_size, norm_eps=cfg.norm_eps, lora_scale=scale,
            wte=ptr(model.transformer.wte.weight.data), wte_rows=model.transformer.wte.weight.size(0),
            ln_f=ptr(model.transformer.ln_f.weight.data), rope_cos=ptr(cos), rope_sin=ptr(sin),
            lm_head=wptr(model.lm_head.linear), adapter_scale=ptr(model.lm_head.adapter_scale.data),
            adapter_bias=ptr(model.lm_head.adapter_bias.data), h_layers=layers, lm_head_ws=sptr(model.lm_head.linear))
        self.max_batch, self.s_max, self.max_tokens = max_batch, s_max, max_tokens
        self.vocab = desc.vocab
        self.device = model.transformer.wte.weight.device
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.dh_engine_create(C.byref(desc), max_batch, s_max, max_tokens, C.byref(handle)))
        self.handle = handle
        self.lib = lib
        self.signature = model._param_signature()

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.dh_engine_destroy(self.handle)
            self.handle = None

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def forward(self, ids: torch.Tensor, seq_len: List[int], pos0: List[int], want_all: bool,
                want_last: bool, slot_base: int = 0) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        n = len(seq_len)
        n_tok = int(sum(seq_len))
        ids = ids.reshape(-1)
        assert ids.numel() == n_tok and ids.dtype == torch.int64 and ids.is_cuda
        ids = ids.contiguous()
        la = torch.empty((n_tok, self.vocab), dtype=torch.bfloat16, device=self.device) if want_all else None
        ll = torch.empty((n, self.vocab), dtype=torch.bfloat16, device=self.device) if want_last else None
        a_len = (C.c_int32 * n)(*seq_len)
        a_pos = (C.c_int32 * n)(*pos0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dh_engine_forward_at(self.handle, ids.data_ptr(), a_len, a_pos, n, int(slot_base),
                                                     ops._p(la), ops._p(ll), torch.cuda.current_stream().cuda_stream))
        return la, ll

    def decode(self, tokens: torch.Tensor, length: torch.Tensor, done: torch.Tensor, n_steps: int, temperature: float,
               top_k: Optional[int], eos_id: Optional[int], seed: int, first_step: int = 0) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dh_engine_decode(
                self.handle, tokens.data_ptr(), tokens.size(1), length.data_ptr(), done.data_ptr(), tokens.size(0),
                int(n_steps), float(temperature), 0 if top_k is None else int(top_k),
                -1 if eos_id is None else int(eos_id), int(seed) & ((1 << 64) - 1), int(first_step),
                torch.cuda.current_stream().cuda_stream))

    def set_rsqrt_emulation(self, vec_width: int, whole_call: bool) -> None:
        _lib.check(self.lib.dh_engine_set_cpu_rsqrt_emulation(self.handle, int(vec_width), int(whole_call)))

    def set_timing(self, on: bool) -> None:
        _lib.check(self.lib.dh_engine_set_timing(self.handle, int(on)))

    def get_timing(self, which: int) -> Tuple[float, int]:
        """(milliseconds, launches) of kernel class `which` since set_timing(True); see dh_engine_get_timing."""
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self.lib.dh_engine_get_timing(self.handle, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def read(self, what: int, layer: int, shape) -> torch.Tensor:
        """Copy of engine state (dh_engine_read): 0 ln_f(x), 1 K cache, 2 V^T cache, 3 residual x."""
        out = torch.empty(shape, dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dh_engine_read(self.handle, what, layer, out.data_ptr(), out.numel() * 2,
                                               torch.cuda.current_stream().cuda_stream))
        return out


# ------------------------------------------------------------------------------------------ GPT
class GPT(nn.Module):
    """Drop-in for ger.lora.GPT (ger/lora.py:475-565)."""

    def __init__(self, config: Config) -> None:
        super().__init__()
        assert config.padded_vocab_size is not None
        config.check_supported()
        self.config = config
        self.lm_head = AdapterV2Linear(config.n_embd, config.padded_vocab_size)
        self.transformer = nn.ModuleDict(dict(
            wte=nn.Embedding(config.padded_vocab_size, config.n_embd),
            h=nn.ModuleList(Block(config, i) for i in range(config.n_layer)),
            ln_f=RMSNorm(config.n_embd, eps=config.norm_eps),
        ))
        self.rope_cache: Optional[RoPECache] = None
        self.max_seq_length = config.block_size
        self.mask_cache: Optional[torch.Tensor] = None   # never built: causality is implicit in the kernels
        self.kv_caches: List[KVCache] = []                # kept for API parity; the engine owns the cache
        # 0 (the product default): RMSNorm rounds rsqrt once, as a GPU run of the reference does.  Tests that
        # compare with the reference's CPU tensors set 32 (AVX-512 hosts; 16 = AVX2): torch's CPU bf16 rsqrt
        # rounds twice in its scalar tail loop and dh_rmsnorm_bf16 can reproduce that per row.
        self.cpu_rsqrt_vec_width = 0
        self._engine: Optional[_Engine] = None
        self._capacity = dict(max_batch=1, s_max=0, max_tokens=0)
        self._cache_len: List[int] = []                   # tokens currently valid per cache slot

    # ---- construction helpers --------------------------------------------------------------
    @classmethod
    def from_name(cls, name: str, **kwargs: Any) -> "GPT":
        return cls(Config.from_name(name, **kwargs))

    def _load_from_state_dict(self, state_dict: Dict, prefix: str, *args: Any, **kwargs: Any) -> None:
        mapping = {"lm_head.weight": "lm_head.linear.weight"}
        super()._load_from_state_dict(map_old_state_dict_weights(state_dict, mapping, prefix), prefix, *args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        # accept both 'x.linear.weight' (reference key) and our holder's 'x.linear.weight' (identical)
        if getattr(self, "fp8", False):
            raise RuntimeError("load_state_dict: this model was quantised to fp8 (quantize_model_fp8): its bf16 weights no longer "
                               "exist; load the checkpoint into a fresh GPT and quantise that")
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._drop_engine()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._drop_engine()
        self.rope_cache = None
        return out

    def build_rope_cache(self, idx: torch.Tensor) -> RoPECache:
        return build_rope_cache(self.config.block_size, self.config.rope_n_elem, torch.bfloat16, idx.device)

    def reset_cache(self) -> None:
        """Forget the cached keys/values (ger/model.py:58-63).  Stale cache contents are never
        attended to (causal bound), so nothing has to be cleared."""
        self.kv_caches.clear()
        self._cache_len = []

    # ---- engine management -----------------------------------------------------------------
    def _param_signature(self) -> Tuple:
        # data_ptr + version: in-place updates of LoRA parameters (optimizer steps) invalidate the
        # engine's rank-padded copies too
        return tuple((p.data_ptr(), p.dtype, str(p.device), p._version) for p in self.parameters())

    def _drop_engine(self) -> None:
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None
        self._cache_len = []

    def refresh_engine(self) -> None:
        """Call after changing LoRA parameters in place (the engine keeps rank-padded copies)."""
        self._drop_engine()

    def set_capacity(self, max_batch: int, s_max: Optional[int] = None, max_tokens: Optional[int] = None) -> None:
        """Pre-size the KV cache / workspace: `max_batch` sequences of up to `s_max` positions
        (default: max_seq_length, as the reference allocates, ger/lora.py:542) and `max_tokens`
        packed tokens per forward.  Re-sizing drops the cache contents."""
        s_max = self.max_seq_length if s_max is None else s_max
        s_max = -(-min(s_max, self.config.block_size) // 64) * 64
        max_tokens = max_batch * min(s_max, 1024) if max_tokens is None else max_tokens
        cap = dict(max_batch=max_batch, s_max=s_max, max_tokens=max(max_tokens, max_batch))
        if cap != self._capacity:
            self._capacity = cap
            self._drop_engine()

    def engine(self, need_batch: int = 1, need_pos: int = 1, need_tokens: int = 1, exact: bool = False) -> _Engine:
        """The engine, grown (cache contents dropped) when the request does not fit.  By default the cache is
        sized like the reference's (max_seq_length positions, ger/lora.py:542: later calls at higher positions
        must not re-allocate) with room for max_batch x 1024 packed tokens; `exact=True` (generate_batch, which
        knows its whole need up front) allocates what was asked for — 2048 sequences x 640 positions is 30 GB,
        at max_seq_length it would be 94 GB plus 77 GB of workspace."""
        cap = self._capacity
        if cap["s_max"] == 0 or need_batch > cap["max_batch"] or need_pos > cap["s_max"] or need_tokens > cap["max_tokens"]:
            mb = max(cap["max_batch"], need_batch)
            if exact:   # nothing inherited from an earlier, reference-sized allocation
                self.set_capacity(need_batch, need_pos, max(need_tokens, need_batch))
            else:
                s_need = max(cap["s_max"], need_pos, self.max_seq_length)
                self.set_capacity(mb, s_need, max(cap["max_tokens"], need_tokens, mb * min(s_need, 1024)))
        if self._engine is not None and self._engine.signature != self._param_signature():
            self._drop_engine()
        if self._engine is None:
            w = self.transformer.wte.weight
            if not w.is_cuda:
                raise _lib.DualHypHipError("dualhyp_amd.GPT runs only on the GPU (model.to('cuda')); no CPU fallback")
            if self.rope_cache is None or self.rope_cache[0].device != w.device:
                self.rope_cache = self.build_rope_cache(w)
            self._engine = _Engine(self, **self._capacity)
        return self._engine

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, idx: torch.Tensor, input_pos: Optional[torch.Tensor] = None,
                lm_head_chunk_size: int = 0) -> Union[torch.Tensor, List[torch.Tensor]]:
        """ger/lora.py:504-549.  With `input_pos` (consecutive positions) the batch runs through
        the KV cache; without, every row is a fresh causal sequence."""
        B, T = idx.shape
        block_size = self.config.block_size
        use_kv_cache = input_pos is not None
        if use_kv_cache:
            assert self.max_seq_length >= T, f"Cannot forward sequence of length {T}, max seq length is only {self.max_seq_length}"
        assert self.max_seq_length <= block_size, f"Cannot attend to {self.max_seq_length}, block size is only {block_size}"
        assert block_size >= T, f"Cannot forward sequence of length {T}, block size is only {block_size}"
        if not use_kv_cache and torch.is_grad_enabled():
            # autograd path (LoRA fine-tune, finetune/ger.py:278): only the no-cache forward trains, and only
            # when there is an active LoRA parameter asking for a gradient.  A cached call (input_pos given) is
            # incremental decoding whatever the grad mode — the reference allows it outside no_grad() — and
            # runs through the engine below, which returns logits without a graph.
            from .train import forward_train, lora_parameters
            if not getattr(self, "fp8", False) and any(p.requires_grad for p in lora_parameters(self)):
                return forward_train(self, idx, lm_head_chunk_size)
        if use_kv_cache:
            pos = input_pos.tolist() if input_pos.numel() <= 2 else [int(input_pos[0]), int(input_pos[-1])]
            p0 = pos[0]
            if pos[-1] - p0 + 1 != T or input_pos.numel() != T:
                raise NotImplementedError("input_pos must be T consecutive positions")
            if p0 + T > self.max_seq_length:
                raise NotImplementedError(f"position {p0 + T - 1} exceeds max_seq_length {self.max_seq_length}")
        else:
            p0 = 0
        eng = self.engine(B, p0 + T, B * T)
        if use_kv_cache:
            if p0 > 0 and (len(self._cache_len) < B or any(c != p0 for c in self._cache_len[:B])):
                raise RuntimeError(f"KV cache holds {self._cache_len[:B]} tokens but input_pos starts at {p0}")
        eng.set_rsqrt_emulation(self.cpu_rsqrt_vec_width, whole_call=True)
        la, _ = eng.forward(idx, [T] * B, [p0] * B, want_all=True, want_last=False)
        self._cache_len = [p0 + T] * B if use_kv_cache else []
        logits = la.view(B, T, -1)
        if lm_head_chunk_size > 0:
            return list(logits.split(lm_head_chunk_size, dim=1))
        return logits


def mark_only_lora_as_trainable(model: nn.Module, bias: str = "none") -> None:
    """ger/lora.py:405-439."""
    for n, p in model.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False
    if bias == "none":
        return
    if bias == "all":
        for n, p in model.named_parameters():
            if "bias" in n:
                p.requires_grad = True
    elif bias == "lora_only":
        for m in model.modules():
            if isinstance(m, LoRALayer) and getattr(m, "bias", None) is not None:
                m.bias.requires_grad = True
    else:
        raise NotImplementedError


def lora_filter(key: str, value: Any) -> bool:
    return "lora_" in key


def merge_lora_weights(model: GPT) -> None:
    """Fold every LoRA update into its base weight (ger/lora.py:707-711)."""
    for m in model.modules():
        if isinstance(m, LoRALinear):
            m.merge()
    model.refresh_engine()
