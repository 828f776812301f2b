"""Model hyper-parameters for the DualHyp LLM hot path.

Mirrors the API surface of the reference's `Config` (ger/config.py:16-157) merged with the
LoRA fields of `ger.lora.Config` (ger/lora.py:446-472): same field names, same derived
values (`head_size`, `padded_vocab_size`, `n_query_groups`, `rope_n_elem`), same
`from_name(name, **kwargs)` lookup by table name or HF name.  Only the two model families
the hot path is quoted on are tabulated (TinyLlama-1.1B ger/config.py:1542-1567 and
Llama-3-8B ger/config.py:801-818) plus small shapes used by the parity tests.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, asdict
from pathlib import Path
from typing import Any, Dict, Optional, Union


def find_multiple(n: int, k: int) -> int:
    """Smallest multiple of k that is >= n (ger/utils.py:29-33)."""
    assert k > 0
    r = n % k
    return n if r == 0 else n + k - r


@dataclass
class Config:
    name: str = ""
    hf_config: dict = field(default_factory=dict)
    scale_embeddings: bool = False
    block_size: int = 4096
    vocab_size: int = 50254
    padding_multiple: int = 512
    padded_vocab_size: Optional[int] = None
    n_layer: int = 16
    n_head: int = 32
    head_size: Optional[int] = None
    n_embd: int = 4096
    rotary_percentage: float = 0.25
    parallel_residual: bool = True
    bias: bool = True
    lm_head_bias: bool = False
    n_query_groups: Optional[int] = None
    shared_attention_norm: bool = False
    _norm_class: str = "LayerNorm"
    norm_eps: float = 1e-5
    _mlp_class: str = "GptNeoxMLP"
    gelu_approximate: str = "none"
    intermediate_size: Optional[int] = None
    rope_condense_ratio: int = 1
    rope_base: int = 10000
    n_expert: int = 0
    n_expert_per_token: int = 0
    # LoRA (ger/lora.py:459-468)
    r: int = 0
    alpha: int = 1
    dropout: float = 0.0
    to_query: bool = False
    to_key: bool = False
    to_value: bool = False
    to_projection: bool = False
    to_mlp: bool = False
    to_head: bool = False
    lora_start_layer: int = 0
    # RelPrompt reliability predictors (ger/relprompt.py Config): encoder feature widths and chunk pooling
    whisper_dim: int = 1280
    raven_dim: int = 1024
    pool_size: int = 10

    def __post_init__(self) -> None:
        if not self.name:
            self.name = self.hf_config.get("name", self.name)
        if self.head_size is None:
            assert self.n_embd % self.n_head == 0
            self.head_size = self.n_embd // self.n_head
        if self.padded_vocab_size is None:
            self.padded_vocab_size = find_multiple(self.vocab_size, self.padding_multiple)
        else:
            self.vocab_size = min(self.vocab_size, self.padded_vocab_size)
        if self.n_query_groups is not None:
            assert self.n_head % self.n_query_groups == 0
        else:
            self.n_query_groups = self.n_head
        if self.intermediate_size is None:
            if self._mlp_class == "LLaMAMLP":
                raise ValueError("The config needs to set the `intermediate_size`")
            self.intermediate_size = 4 * self.n_embd
        self.rope_n_elem = int(self.rotary_percentage * self.head_size)

    # -- what the HIP path implements ------------------------------------------------------
    def check_supported(self) -> None:
        """The HIP decoder covers the Llama family the hot path is quoted on; everything
        else in the reference's table is out of scope and fails loudly."""
        problems = []
        if self._norm_class != "RMSNorm":
            problems.append("_norm_class must be RMSNorm")
        if self._mlp_class != "LLaMAMLP":
            problems.append("_mlp_class must be LLaMAMLP")
        if self.parallel_residual or self.shared_attention_norm:
            problems.append("parallel_residual/shared_attention_norm unsupported")
        if self.bias or self.lm_head_bias:
            problems.append("bias unsupported")
        if self.rotary_percentage != 1.0:
            problems.append("rotary_percentage must be 1.0")
        if self.head_size not in (64, 128):
            problems.append("head_size must be 64 or 128")
        if self.n_embd % 64 or self.intermediate_size % 64 or self.padded_vocab_size % 64:
            problems.append("n_embd/intermediate_size/padded_vocab_size must be multiples of 64")
        if self.to_mlp or self.to_head:
            problems.append("LoRA on mlp/head is not on the hot path (inference/ger.py:150-153)")
        if self.r > 16:
            problems.append("LoRA rank > 16 unsupported")
        if self.lora_start_layer != 0:
            problems.append("lora_start_layer != 0 unsupported")
        if self.scale_embeddings or self.n_expert:
            problems.append("scale_embeddings / MoE unsupported")
        if problems:
            raise NotImplementedError(f"Config {self.name!r} is outside the HIP hot path: " + "; ".join(problems))

    # -- constructors ----------------------------------------------------------------------
    @classmethod
    def from_name(cls, name: str, **kwargs: Any) -> "Config":
        if name in name_to_config:
            conf = name_to_config[name]
        else:
            try:
                conf = next(c for c in configs if name == c["hf_config"]["name"])
            except StopIteration:
                raise ValueError(f"{name!r} is not a supported config name")
        conf = dict(conf)
        if "condense_ratio" in kwargs:  # legacy spelling
            kwargs["rope_condense_ratio"] = kwargs.pop("condense_ratio")
        conf.update(kwargs)
        return cls(**conf)

    @classmethod
    def from_json(cls, path: Union[str, Path], **kwargs: Any) -> "Config":
        with open(path, encoding="utf-8") as fp:
            js = json.load(fp)
        for d in (js, kwargs):
            if "condense_ratio" in d:
                d["rope_condense_ratio"] = d.pop("condense_ratio")
        if "org" in js:
            js["hf_config"] = {"name": js["name"], "org": js.pop("org")}
        if "org" in kwargs:
            kwargs["hf_config"] = {"name": kwargs.get("name", js["name"]), "org": kwargs.pop("org")}
        js.pop("rope_n_elem", None)
        js.update(kwargs)
        return cls(**js)

    @classmethod
    def from_checkpoint(cls, path: Path, **kwargs: Any) -> "Config":
        path = Path(path)
        if (p := path / "lit_config.json").is_file():
            return cls.from_json(p, **kwargs)
        if path.name in name_to_config:
            return cls.from_name(path.name, **kwargs)
        raise FileNotFoundError(f"For {str(path)!r} neither 'lit_config.json' nor matching config exists.")

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        return d


def _llama(**kw) -> dict:
    base = dict(rotary_percentage=1.0, parallel_residual=False, bias=False,
                _norm_class="RMSNorm", _mlp_class="LLaMAMLP")
    base.update(kw)
    return base


configs = []

# TinyLlama 1.1B (ger/config.py:1542-1567): both the base and the -Chat-v1.0 names.
for _kind, _post in (("", "-intermediate-step-1431k-3T"), ("-chat", "-Chat-v1.0")):
    configs.append(_llama(
        name=f"tiny-llama-1.1b{_kind}",
        hf_config=dict(org="TinyLlama", name=f"TinyLlama-1.1B{_post}"),
        block_size=2048, vocab_size=32000, padding_multiple=64, n_layer=22, n_head=32,
        n_embd=2048, norm_eps=1e-5, intermediate_size=5632, n_query_groups=4))

# Llama 3 8B (ger/config.py:801-818); rope_base is carried but ignored, as in the reference
# (quirk Q1: ger/model.py:120-126 never forwards it).
for _kind in ("", "-Instruct"):
    configs.append(_llama(
        name=f"Llama-3-8B{_kind}",
        hf_config=dict(org="meta-llama", name=f"Meta-Llama-3-8B{_kind}"),
        block_size=8192, vocab_size=128000, padded_vocab_size=128256, n_layer=32, n_head=32,
        n_query_groups=8, n_embd=4096, intermediate_size=14336, rope_base=500000))

# Small shapes for parity tests (not in the reference's table).
configs.append(_llama(name="parity-tiny", hf_config=dict(org="dualhyp_amd", name="parity-tiny"),
                      block_size=128, vocab_size=256, padding_multiple=64, n_layer=2, n_head=4,
                      n_embd=256, intermediate_size=384, n_query_groups=2))
configs.append(_llama(name="parity-block", hf_config=dict(org="dualhyp_amd", name="parity-block"),
                      block_size=2048, vocab_size=32000, padding_multiple=64, n_layer=1, n_head=32,
                      n_embd=2048, intermediate_size=5632, n_query_groups=4))
configs.append(_llama(name="parity-hs128", hf_config=dict(org="dualhyp_amd", name="parity-hs128"),
                      block_size=256, vocab_size=512, padding_multiple=64, n_layer=2, n_head=4,
                      n_embd=512, intermediate_size=768, n_query_groups=2))

# room for a byte-tokenised DualHyp prompt (~700 tokens) + 150 generated: harness tests (tests/test_harness.py)
configs.append(_llama(name="parity-harness", hf_config=dict(org="dualhyp_amd", name="parity-harness"),
                      block_size=1024, vocab_size=300, padding_multiple=64, n_layer=2, n_head=4,
                      n_embd=256, intermediate_size=384, n_query_groups=2))

name_to_config = {c["name"]: c for c in configs}

# LoRA settings used by both reference harnesses (inference/ger.py:145-153, finetune/ger.py:386-394)
GER_LORA = dict(r=16, alpha=16, dropout=0.05, to_query=True, to_key=True, to_value=True,
                to_projection=True, to_mlp=False, to_head=False)
