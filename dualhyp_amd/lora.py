"""Import-compatible alias of the reference module name `ger.lora`."""
from .config import Config  # noqa: F401
from .gpt import (GPT, Block, CausalSelfAttention, LLaMAMLP, LoRALayer, LoRALinear, LoRAQKVLinear,  # noqa: F401
                  AdapterV2Linear, mark_only_lora_as_trainable, lora_filter, merge_lora_weights)
