"""dualhyp_amd — MI355X-native LLM hot path of DualHyp (generative error correction decoder).

Public surface mirrors the reference (ger.lora / generate.base / ger.utils):
    Config, GPT, generate, generate_batch, chunked_cross_entropy,
    mark_only_lora_as_trainable, lora_filter, merge_lora_weights
"""
from .config import Config, GER_LORA  # noqa: F401
from .gpt import GPT, mark_only_lora_as_trainable, lora_filter, merge_lora_weights, build_rope_cache  # noqa: F401
from .generate import generate, generate_batch  # noqa: F401
from .utils import chunked_cross_entropy  # noqa: F401
from .checkpoint import load_checkpoint, save_checkpoint, convert_hf_checkpoint  # noqa: F401
from .quant import quantize_model_fp8  # noqa: F401

__version__ = "0.1.0"
