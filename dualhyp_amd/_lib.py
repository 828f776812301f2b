"""ctypes binding of libdualhyp_hip.so (the C ABI in include/dualhyp_hip.h).

There is NO fallback: if the library is missing or a symbol is absent, importing an op raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_HERE = Path(__file__).resolve().parent
import os as _os
# DUALHYP_HIP_LIB: another build of the same ABI (kernel A/B comparisons inside one gpurun call)
LIB_PATH = Path(_os.environ["DUALHYP_HIP_LIB"]) if _os.environ.get("DUALHYP_HIP_LIB") else _HERE / "lib" / "libdualhyp_hip.so"

P = C.c_void_p
I = C.c_int
I64 = C.c_int64
U64 = C.c_uint64
F = C.c_float


class LayerWeights(C.Structure):
    _fields_ = [(n, P) for n in ("norm_1", "norm_2", "attn_w", "attn_lora_a", "attn_lora_b", "proj_w",
                                 "proj_lora_a", "proj_lora_b", "fc_1", "fc_2", "mlp_proj",
                                 "attn_ws", "proj_ws", "fc_1_ws", "fc_2_ws", "mlp_proj_ws")]


class ModelDesc(C.Structure):
    _fields_ = [("n_layer", C.c_int32), ("n_head", C.c_int32), ("n_groups", C.c_int32), ("head_size", C.c_int32),
                ("n_embd", C.c_int32), ("intermediate", C.c_int32), ("vocab", C.c_int32), ("block_size", C.c_int32),
                ("norm_eps", F), ("lora_scale", F), ("wte", P), ("wte_rows", C.c_int32), ("ln_f", P),
                ("rope_cos", P), ("rope_sin", P), ("lm_head", P), ("adapter_scale", P), ("adapter_bias", P),
                ("h_layers", C.POINTER(LayerWeights)), ("lm_head_ws", P)]


# name -> (restype, argtypes); must list every function declared in include/dualhyp_hip.h
SIGNATURES = {
    "dh_abi_version": (I, []),
    "dh_set_tuning": (I, [I, I]),
    "dh_last_error": (C.c_char_p, []),
    "dh_device_info": (I, [C.c_char_p, I, C.POINTER(I), C.POINTER(I64)]),
    "dh_embed_bf16": (I, [P, P, P, I, I, I, P]),
    "dh_rmsnorm_bf16": (I, [P, P, P, P, P, I, I, F, P, P]),
    "dh_qkv_rope_cache_bf16": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dh_linear_bf16": (I, [P, P, P, I, I, I, I, P, P, I, P, F, I, I, P, P, P, P]),
    "dh_linear_qkv_rope_cache_bf16": (I, [P, P, I, I, P, I, P, F, P, P, P, P, P, P, P, I, I, I, I, P]),
    "dh_linear_lora_bf16": (I, [P, P, P, I, I, I, P, P, F, I, I, P, P, P]),
    "dh_linear_qkv_lora_rope_cache_bf16": (I, [P, P, I, I, P, P, F, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    "dh_linear_partial_bf16": (I, [P, P, P, P, I, I, I, I, I, P]),
    "dh_linear_chain_bf16": (I, [P, P, P, P, I, I, I, I, I, P]),
    "dh_linear_partial_pairs_bf16": (I, [P, P, P, P, I, I, I, I, I, P]),
    "dh_finish_norm_bf16": (I, [P, I, I, I, I, I, P, F, P, P, P, P, F, P, P]),
    "dh_attn_decode_fused_bf16": (I, [P, I, I, I, I, I, P, F, I, I, P, P, P, P, P, P, P, I, I, I, I, P]),
    "dh_attn_prefill_bf16": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "dh_dropout_bf16": (I, [P, P, P, I64, F, C.c_uint64, C.c_uint32, P, P]),
    "dh_swiglu_fwd_bf16": (I, [P, P, P, I64, P]),
    "dh_linear_swiglu_train_bf16": (I, [P, P, P, P, P, P, I, I, I, P]),
    "dh_linear_mul_bf16": (I, [P, P, P, I, I, I, P, P]),
    "dh_swiglu_bwd_bf16": (I, [P, P, P, P, I, I, P]),
    "dh_rmsnorm_bwd_bf16": (I, [P, P, P, P, P, I, I, F, P]),
    "dh_qkv_rope_bwd_bf16": (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    "dh_tn_accum_f32": (I, [P, I, P, I, P, I, I, I, I, F, I, P, P]),
    "dh_tn_accum_work_bytes": (I64, [I, I, I]),
    "dh_tn_accum_seg_f32": (I, [P, I, P, I, P, I, I, I, I, I, F, I, P, P]),
    "dh_rowdot_f32": (I, [P, P, P, I64, I, P]),
    "dh_transpose_pad_bf16": (I, [P, P, P, P, P, I, I, I, I, P]),
    "dh_transpose_frag_bf16": (I, [P, P, P, I, I, I, P]),
    "dh_attn_bwd_transposes": (I, [I, I, I, I]),
    "dh_attn_bwd_bf16": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "dh_attn_decode_work_bytes": (I64, [I, I, I, I]),
    "dh_attn_decode_bf16": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dh_sample_bf16": (I, [P, I, P, I, P, P, I, F, I, I64, U64, I, P]),
    "dh_quant_rows_fp8": (I, [P, P, P, I, I, P]),
    "dh_rmsnorm_quant_fp8": (I, [P, P, P, P, P, I, I, F, P, P]),
    "dh_linear_fp8": (I, [P, P, P, P, P, I, I, I, I, P, P, P, P, P, P]),
    "dh_linear_fp8_ex": (I, [P, P, P, P, P, I, I, I, I, P, P, P, P, P, I, P]),
    "dh_linear_fp8_f32": (I, [P, P, P, P, P, I, I, I, P]),
    "dh_engine_create": (I, [C.POINTER(ModelDesc), I, I, I, C.POINTER(P)]),
    "dh_engine_destroy": (None, [P]),
    "dh_engine_device_bytes": (I64, [P]),
    "dh_im2col3_bf16": (I, [P, P, I, I, I, I, I, P]),
    "dh_pool_head_bf16": (I, [P, P, P, P, I, I, I, I, P]),
    "dh_pool_head_bwd_bf16": (I, [P, P, P, P, P, I, I, I, I, P]),
    "dh_col2im3_bf16": (I, [P, P, P, P, I, I, I, I, P]),
    "dh_cross_entropy_fwd": (I, [P, I, P, P, P, I, I, P]),
    "dh_cross_entropy_bwd": (I, [P, I, P, P, P, P, I, I, P]),
    "dh_engine_forward": (I, [P, P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), I, P, P, P]),
    "dh_engine_forward_at": (I, [P, P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), I, I, P, P, P]),
    "dh_engine_set_cpu_rsqrt_emulation": (I, [P, I, I]),
    "dh_engine_decode": (I, [P, P, I, P, P, I, I, F, I, I64, U64, I, P]),
    "dh_engine_read": (I, [P, I, I, P, I64, P]),
    "dh_engine_set_timing": (I, [P, I]),
    "dh_engine_get_timing": (I, [P, I, C.POINTER(C.c_double), C.POINTER(I64)]),
}

_lib = None


class DualHypHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the library and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise DualHypHipError(
            f"{LIB_PATH} not found: the HIP library is not built. Run `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DualHypHipError(f"libdualhyp_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.dh_abi_version() != 6:
        raise DualHypHipError("libdualhyp_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().dh_last_error().decode(errors="replace")
        raise DualHypHipError(msg or f"libdualhyp_hip call failed with code {rc}")
