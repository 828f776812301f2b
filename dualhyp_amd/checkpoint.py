"""Checkpoint I/O in the reference's formats.

* `lit_model.pth` — the base model the reference loads (`finetune/ger.py:122-124`, `lazy_load` +
  `load_state_dict(strict=False)`): a torch zip-pickle of a FLAT state dict with lit-gpt keys
  (`transformer.h.{l}.attn.attn.weight` ...; `GPT.load_state_dict` maps them onto the LoRA-wrapped
  `...attn.attn.linear.weight`, `ger/lora.py:561-565,619-628,693-704`).
* `best_model.pth` / `lit_model_lora_finetuned.pth` — what `fabric.save(path, {"model": model})`
  writes (`finetune/ger.py:356-358`) and `inference/ger.py:32-35` reads back as
  `checkpoint["model"]`: the WHOLE state dict (base + LoRA) under the key "model".
* Hugging Face Llama checkpoints -> lit layout (`scripts/convert_hf_checkpoint.py:117-202,313-373`):
  key renames, and q/k/v projections interleaved PER QUERY GROUP into one fused matrix
  `[q_0 .. q_{q_per_kv-1} k v]_g` — the layout `dh_qkv_rope_cache_bf16` and the LoRA-QKV zero-pad
  (quirk Q2) expect.  Pinned against the reference's own converter by
  `tests/golden/convert_hf_llama.safetensors`.

Nothing here touches the GPU: tensors are converted / memory-mapped on the host and handed to
`GPT.load_state_dict`.
"""
from __future__ import annotations

import json
import re
from pathlib import Path
from typing import Dict, Iterable, List, Mapping, Optional, Union

import torch

from .config import Config

_LAYER = re.compile(r"^model\.layers\.(\d+)\.(.+)$")

# HF name (layer prefix stripped) -> lit name; None = dropped (rotary tables are rebuilt from the config)
_HF_LAYER_MAP = {
    "input_layernorm.weight": "norm_1.weight",
    "input_layernorm.bias": "norm_1.bias",
    "self_attn.o_proj.weight": "attn.proj.weight",
    "self_attn.rotary_emb.inv_freq": None,
    "post_attention_layernorm.weight": "norm_2.weight",
    "post_attention_layernorm.bias": "norm_2.bias",
    "mlp.gate_proj.weight": "mlp.fc_1.weight",
    "mlp.up_proj.weight": "mlp.fc_2.weight",
    "mlp.down_proj.weight": "mlp.proj.weight",
}
_HF_TOP_MAP = {
    "model.embed_tokens.weight": "transformer.wte.weight",
    "model.norm.weight": "transformer.ln_f.weight",
    "model.norm.bias": "transformer.ln_f.bias",
    "lm_head.weight": "lm_head.weight",
}
_QKV = {"self_attn.q_proj.weight": 0, "self_attn.k_proj.weight": 1, "self_attn.v_proj.weight": 2}


def interleave_qkv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, config: Config) -> torch.Tensor:
    """Separate HF projections -> the fused lit matrix: for every query group its q_per_kv query
    heads, then its key head, then its value head (scripts/convert_hf_checkpoint.py:187-199)."""
    hs, G = config.head_size, config.n_query_groups
    q_per_kv = config.n_head // G
    assert q.size(0) == config.n_head * hs and k.size(0) == G * hs and v.size(0) == G * hs, \
        f"q/k/v rows {q.size(0)}/{k.size(0)}/{v.size(0)} do not match n_head={config.n_head}, groups={G}, head_size={hs}"
    qs, ks, vs = q.split(hs * q_per_kv), k.split(hs), v.split(hs)
    return torch.cat([t for group in zip(qs, ks, vs) for t in group])


def split_qkv(qkv: torch.Tensor, config: Config):
    """Inverse of `interleave_qkv` (lit -> HF export)."""
    hs, G = config.head_size, config.n_query_groups
    q_per_kv = config.n_head // G
    blocks = qkv.view(G, (q_per_kv + 2) * hs, -1)
    q = blocks[:, : q_per_kv * hs].reshape(config.n_head * hs, -1)
    k = blocks[:, q_per_kv * hs:(q_per_kv + 1) * hs].reshape(G * hs, -1)
    v = blocks[:, (q_per_kv + 1) * hs:].reshape(G * hs, -1)
    return q.contiguous(), k.contiguous(), v.contiguous()


class HFLlamaConverter:
    """Feed the shards of an HF Llama checkpoint in any order (`add`), then `finish()`.  A layer's
    q, k and v may arrive in different shards; they are fused once all three are present."""

    def __init__(self, config: Config, dtype: Optional[torch.dtype] = None) -> None:
        if config._mlp_class != "LLaMAMLP":
            raise NotImplementedError(f"HF conversion is implemented for LLaMAMLP models, not {config._mlp_class}")
        self.config, self.dtype = config, dtype
        self.state: Dict[str, torch.Tensor] = {}
        self._qkv: Dict[int, List[Optional[torch.Tensor]]] = {}

    def _cast(self, t: torch.Tensor) -> torch.Tensor:
        return t if self.dtype is None or t.dtype == self.dtype else t.to(self.dtype)

    def add(self, hf_weights: Mapping[str, torch.Tensor]) -> None:
        for name, param in hf_weights.items():
            m = _LAYER.match(name)
            if m:
                l, rest = int(m.group(1)), m.group(2)
                if rest in _QKV:
                    self._qkv.setdefault(l, [None, None, None])[_QKV[rest]] = param
                    continue
                if rest not in _HF_LAYER_MAP:
                    raise KeyError(f"unexpected HF Llama tensor {name!r}")
                to = _HF_LAYER_MAP[rest]
                if to is None:
                    continue
                self.state[f"transformer.h.{l}.{to}"] = self._cast(param)
            else:
                if name not in _HF_TOP_MAP:
                    raise KeyError(f"unexpected HF Llama tensor {name!r}")
                self.state[_HF_TOP_MAP[name]] = self._cast(param)
        for l, (q, k, v) in list(self._qkv.items()):
            if q is None or k is None or v is None:
                continue   # split across shards
            self.state[f"transformer.h.{l}.attn.attn.weight"] = interleave_qkv(self._cast(q), self._cast(k), self._cast(v), self.config)
            del self._qkv[l]

    def finish(self) -> Dict[str, torch.Tensor]:
        if self._qkv:
            raise ValueError(f"layers {sorted(self._qkv)} are missing one of q_proj / k_proj / v_proj")
        if "lm_head.weight" not in self.state:   # tied embeddings
            self.state["lm_head.weight"] = self.state["transformer.wte.weight"]
        return self.state


def convert_hf_llama(hf_weights: Union[Mapping[str, torch.Tensor], Iterable[Mapping[str, torch.Tensor]]], config: Config,
                     dtype: Optional[torch.dtype] = None) -> Dict[str, torch.Tensor]:
    conv = HFLlamaConverter(config, dtype)
    for shard in ([hf_weights] if isinstance(hf_weights, Mapping) else hf_weights):
        conv.add(shard)
    return conv.finish()


def _read_shard(path: Path) -> Dict[str, torch.Tensor]:
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path))
    return torch.load(str(path), map_location="cpu", mmap=True, weights_only=True)


def convert_hf_checkpoint(checkpoint_dir: Union[str, Path], model_name: Optional[str] = None,
                          dtype: Optional[Union[str, torch.dtype]] = None) -> Path:
    """`python scripts/convert_hf_checkpoint.py --checkpoint_dir DIR` of the reference: reads DIR's
    `*.bin` (or `*.safetensors`) shards, writes `DIR/lit_model.pth` + `DIR/lit_config.json`."""
    d = Path(checkpoint_dir)
    name = model_name or d.name
    if isinstance(dtype, str):
        dtype = getattr(torch, dtype)
    config = Config.from_name(name)
    (d / "lit_config.json").write_text(json.dumps(config.to_dict()))
    index = d / "pytorch_model.bin.index.json"
    if index.is_file():
        files = {d / f for f in json.loads(index.read_text())["weight_map"].values()}
    else:
        files = {f for f in d.glob("*.bin") if f.name != "training_args.bin"} or set(d.glob("*.safetensors"))
    if not files:
        raise ValueError(f"Expected {str(d)!r} to contain .bin or .safetensors files")
    conv = HFLlamaConverter(config, dtype)
    for f in sorted(files):
        conv.add(_read_shard(f))
    out = d / "lit_model.pth"
    torch.save(conv.finish(), str(out))
    return out


def load_checkpoint(path: Union[str, Path]) -> Dict[str, torch.Tensor]:
    """State dict of a `lit_model.pth` (flat) or of a fine-tuned `best_model.pth` (`{"model": sd}`),
    memory-mapped.  Pass the result to `GPT.load_state_dict(sd, strict=False)`."""
    p = Path(path)
    if not p.is_file():
        raise FileNotFoundError(f"Path {str(p)!r} does not exist or is not a file.")
    try:
        ck = torch.load(str(p), map_location="cpu", mmap=True, weights_only=True)
    except RuntimeError:   # legacy (non-zip) pickles cannot be mapped
        ck = torch.load(str(p), map_location="cpu", weights_only=True)
    if isinstance(ck, dict) and "model" in ck and isinstance(ck["model"], dict):
        ck = ck["model"]
    return ck


def save_checkpoint(model, path: Union[str, Path], lora_only: bool = False) -> None:
    """`fabric.save(path, {"model": model})` (finetune/ger.py:356-358): the whole state dict under
    "model", reference keys.  `lora_only=True` applies `lora_filter` (what the reference imports at
    finetune/ger.py:26 but never uses) — a 18 MB file instead of 2.2 GB for TinyLlama."""
    if getattr(model, "fp8", False):
        raise RuntimeError("save_checkpoint: the model was quantised to fp8 (quantize_model_fp8); its state dict holds empty "
                           "weights — save the bf16 model before quantising")
    sd = {k: v.detach().to("cpu") for k, v in model.state_dict().items() if not lora_only or "lora_" in k}
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save({"model": sd}, str(path))


def export_hf_llama(state_dict: Mapping[str, torch.Tensor], config: Config) -> Dict[str, torch.Tensor]:
    """lit -> HF names (merged weights; LoRA tensors are skipped — call merge_lora_weights first)."""
    inv_layer = {v: k for k, v in _HF_LAYER_MAP.items() if v is not None}
    inv_top = {v: k for k, v in _HF_TOP_MAP.items()}
    out: Dict[str, torch.Tensor] = {}
    for name, t in state_dict.items():
        if "lora_" in name or "adapter_" in name:
            continue
        name = name.replace(".linear.weight", ".weight")
        m = re.match(r"^transformer\.h\.(\d+)\.(.+)$", name)
        if m:
            l, rest = int(m.group(1)), m.group(2)
            if rest == "attn.attn.weight":
                q, k, v = split_qkv(t, config)
                for nm, tt in (("q_proj", q), ("k_proj", k), ("v_proj", v)):
                    out[f"model.layers.{l}.self_attn.{nm}.weight"] = tt
            else:
                out[f"model.layers.{l}.{inv_layer[rest]}"] = t
        else:
            out[inv_top[name]] = t
    return out


if __name__ == "__main__":   # python -m dualhyp_amd.checkpoint --checkpoint_dir checkpoints/TinyLlama/TinyLlama-1.1B-Chat-v1.0
    import argparse
    ap = argparse.ArgumentParser(description="HF Llama checkpoint -> lit_model.pth + lit_config.json")
    ap.add_argument("--checkpoint_dir", required=True)
    ap.add_argument("--model_name", default=None)
    ap.add_argument("--dtype", default=None)
    a = ap.parse_args()
    print(convert_hf_checkpoint(a.checkpoint_dir, a.model_name, a.dtype))
