"""Sharded GER / DualHyp inference + WER — the MI355X counterpart of inference/ger.py:30-124.

The reference decodes the test set one utterance at a time on every rank (no sharding, quirk Q4).
Here each rank takes a strided shard of the utterances, decodes them in batches through
`generate_batch` (one packed prefill + hipGraph decode per batch) and the ranks all-reduce four
integers (edit errors, reference words, exact matches, count) for the corpus WER; predictions are
gathered to rank 0.  No collective runs inside the decode loop."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from .wer import post_normalize, wer_counts


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Strided shard: rank r takes r, r+world, ... (balanced to within one utterance)."""
    return list(range(rank, n, world))


def extract_answer(decoded_full: str, decoded_prompt: str) -> str:
    """inference/ger.py:84-86: strip the prompt text, keep the first line."""
    return decoded_full[len(decoded_prompt):].split("\n")[0].strip()


def _reduce_counts(c: Dict[str, int], device) -> Dict[str, int]:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return c
    keys = sorted(c)
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([c[k] for k in keys], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return dict(zip(keys, t.tolist()))


def _gather(obj: Any) -> List[Any]:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out: List[Any] = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def run_inference(generate_fn: Callable[[List[torch.Tensor]], List[torch.Tensor]], examples: Sequence[Dict[str, Any]],
                  decode: Callable[[torch.Tensor], str], *, batch_size: int = 32, rank: int = 0, world: int = 1,
                  device="cpu") -> Dict[str, Any]:
    """examples[i] needs 'input_ids_no_response' (1-D ids) and 'ground_truth'.  `generate_fn` maps a
    list of prompts to a list of prompt+continuation id tensors (dualhyp_amd.generate_batch bound to a
    model; a stub in the CPU tests).  Returns corpus metrics on every rank and predictions on rank 0."""
    mine = shard_indices(len(examples), rank, world)
    preds: Dict[int, Dict[str, str]] = {}
    for b in range(0, len(mine), batch_size):
        idxs = mine[b:b + batch_size]
        prompts = [examples[i]["input_ids_no_response"] for i in idxs]
        outs = generate_fn(prompts)
        for i, p, o in zip(idxs, prompts, outs):
            preds[i] = {"inference": extract_answer(decode(o), decode(p)),
                        "ground_truth": examples[i]["ground_truth"].strip()}
    order = sorted(preds)
    pr = [preds[i]["inference"] for i in order]
    gt = [preds[i]["ground_truth"] for i in order]
    raw = _reduce_counts(wer_counts(pr, gt), device)
    post = _reduce_counts(wer_counts([post_normalize(p) for p in pr], [post_normalize(g) for g in gt]), device)
    gathered = _gather(preds)
    merged: Dict[int, Dict[str, str]] = {}
    for g in gathered:
        merged.update(g)
    n = max(raw["n"], 1)
    return {"WER": raw["errors"] / max(raw["ref_words"], 1), "gtms": raw["exact"] / n,
            "post_ST_wer": post["errors"] / max(post["ref_words"], 1), "post_gtms": post["exact"] / n,
            "n": raw["n"], "predictions": [merged[i] for i in sorted(merged)] if rank == 0 else None}


# ------------------------------------------------------------------------------------------ harness
def add_lora_arguments(parser) -> None:
    """The LoRA flags shared by both reference harnesses (inference/ger.py:145-153, finetune/ger.py:386-394).
    `type=bool` is the reference's: any non-empty string is True, so the flags are effectively constants."""
    parser.add_argument("--lora_r", type=int, default=16)
    parser.add_argument("--lora_alpha", type=int, default=16)
    parser.add_argument("--lora_dropout", type=float, default=0.05)
    parser.add_argument("--lora_query", type=bool, default=True)
    parser.add_argument("--lora_key", type=bool, default=True)
    parser.add_argument("--lora_value", type=bool, default=True)
    parser.add_argument("--lora_projection", type=bool, default=True)
    parser.add_argument("--lora_mlp", type=bool, default=False)
    parser.add_argument("--lora_head", type=bool, default=False)


def config_from_args(args):
    """Config.from_name(checkpoint_dir.name, r=..., ...) as inference/ger.py:177-190 / finetune/ger.py:101-112."""
    from pathlib import Path
    from .config import Config
    name = getattr(args, "config_name", None) or Path(args.llm_checkpoint).name
    cfg = Config.from_name(name, r=args.lora_r, alpha=args.lora_alpha, dropout=args.lora_dropout, to_query=args.lora_query,
                           to_key=args.lora_key, to_value=args.lora_value, to_projection=args.lora_projection,
                           to_mlp=args.lora_mlp, to_head=args.lora_head)
    if "llama-3" in cfg.name.lower():
        cfg.block_size = 4096                      # inference/ger.py:189-190
    return cfg


def init_distributed(n_devices: int):
    """One process per GPU (the reference's Fabric launch): -> (rank, world, device).  Under torchrun the env is read;
    `--d N` without a launcher re-runs this module under torch.distributed.run BEFORE anything touches the GPU."""
    import os
    import sys
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and n_devices > 1:
        import subprocess
        # --standalone lets the launcher pick (and hold) its own rendezvous port on 127.0.0.1: no bind/close race.  The
        # target is this run's entry point: `-m package.module` when started that way, else the script path
        # (__main__.__spec__ is None for `python dualhyp_amd/finetune.py` or when called from another entry point)
        spec = getattr(sys.modules.get("__main__"), "__spec__", None)
        target = ["-m", spec.name] if spec is not None and spec.name else [os.path.abspath(sys.argv[0])]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={n_devices}", *target, *sys.argv[1:]]
        sys.exit(subprocess.run(cmd).returncode)
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("DUALHYP_DP_REHEARSAL") == "1"    # every rank on cuda:0, gloo (one-GPU boxes, tests)
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
    return rank, world, dev


def result(adapter_path: str, model, tokenizer, args, rank: int = 0, world: int = 1) -> Dict[str, Any]:
    """inference/ger.py:30-124: load the fine-tuned state dict, decode the test JSON, WER, predictions file."""
    import json
    import os
    from pathlib import Path
    from .checkpoint import load_checkpoint
    from .data import HypothesesDataset
    from .generate import generate_batch
    if adapter_path:
        sd = load_checkpoint(adapter_path)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if rank == 0:
            print("Missing:", missing)
            print("Unexpected:", unexpected)
    fmt = args.prompts_format
    if args.dual_hypotheses and "Dual" not in fmt and fmt != "RelPrompt":
        print("Warning: dual hypotheses is enabled, but prompts format is not Dual.")
    # RelPrompt with encoder features at hand: the prompt's mask tokens are PREDICTED by the reliability classifiers
    # (inference/relprompt.py:113-153); without features the ground-truth chunk labels of the corruption records are used
    feats_dir = getattr(args, "enc_features_dir", None) if fmt == "RelPrompt" else None
    enc_features = None
    _variants: dict = {}          # Uid -> its items (filled from the dataset below; read when a feature file is looked up)
    if feats_dir:
        def enc_features(s1, s2, _d=Path(feats_dir)):
            # <dir>/<Uid>.pt = {'audio', 'visual'} when the Uid has one variant; with several, the audio features of s1's corruption
            # and the visual features of s2's come from <Uid>.<hash of that corruption record>.pt (data.feature_key)
            from .data import feature_key
            n_var = len(_variants.get(s1["Uid"], (s1,)))
            fa = torch.load(_d / f"{feature_key(s1, 'Audio_Corruption', n_var)}.pt", map_location="cpu")
            fv = fa if n_var <= 1 else torch.load(_d / f"{feature_key(s2, 'Visual_Corruption', n_var)}.pt", map_location="cpu")
            return fa["audio"].float(), fv["visual"].float()
    ds = HypothesesDataset(args.test_path, tokenizer, prompts_format=fmt if (args.dual_hypotheses or fmt == "RelPrompt") else "GER",
                           nhyps_key=args.nhyps_key, max_nhyps=args.max_nhyps, language=args.language, seed=args.seed,
                           mask_threshold=getattr(args, "mask_threshold", None), time_window=getattr(args, "time_window", 0.4),
                           enc_features=enc_features, leave_masks=bool(feats_dir))
    _variants.update(ds.uid2sample)
    examples = [ds[i] for i in range(len(ds))]
    mask_stats = None
    if feats_dir:
        from .relprompt import predicted_mask_prompt
        hit = {"audio": [0, 0], "visual": [0, 0]}
        for ex in examples:
            prompt, a, v = predicted_mask_prompt(model, ex["input_no_response"], ex["audio_enc_features"], ex["visual_enc_features"])
            ex["input_no_response"] = prompt
            ex["input_ids_no_response"] = torch.tensor(tokenizer.encode(prompt), dtype=torch.int64)    # re-encoded with the predicted masks
            for name, pred, tgt in (("audio", a, ex["audio_mask_targets"]), ("visual", v, ex["visual_mask_targets"])):
                n = min(pred.numel(), tgt.numel())                                                   # trimmed to the shorter, as the reference
                hit[name][0] += int((pred[:n] == tgt[:n]).sum())
                hit[name][1] += n
        mask_stats = {f"{k}_mask_accuracy": h[0] / max(h[1], 1) for k, h in hit.items()}
    eos = tokenizer.eos_token_id

    # --decode_batch is a throughput knob tuned on TinyLlama (2 x 22.5 KB of KV per position); the KV cache is allocated for
    # decode_batch x (longest prompt + max_new_tokens) positions, so bound it by what the device has free (Llama-3-8B: 131 KB per
    # position — 640 x 1700 positions would be 140 GB)
    if torch.cuda.is_available() and model.transformer.wte.weight.is_cuda:
        c_ = model.config
        kv_per_pos = c_.n_layer * 2 * c_.n_query_groups * c_.head_size * 2
        longest = max((int(e["input_ids_no_response"].numel()) for e in examples), default=1) + args.max_new_tokens
        free, _total = torch.cuda.mem_get_info(model.transformer.wte.weight.device)
        fit = int(0.8 * free // max(kv_per_pos * longest, 1))
        if fit < 1:
            raise RuntimeError(f"not enough device memory for one sequence of {longest} positions ({kv_per_pos * longest / 2**30:.1f} GiB of KV cache)")
        if fit < args.decode_batch:
            print(f"[dualhyp_amd] --decode_batch {args.decode_batch} -> {fit}: {kv_per_pos * longest / 2**20:.0f} MiB of KV cache per sequence, "
                  f"{free / 2**30:.0f} GiB free")
            args.decode_batch = fit

    def gen(prompts):
        dev = model.transformer.wte.weight.device
        outs = generate_batch(model, [p.to(dev) for p in prompts], args.max_new_tokens, temperature=0.2, top_k=1, eos_id=eos,
                              prefill_batch=max(1, min(args.prefill_batch, args.decode_batch)))
        return [o.cpu() for o in outs]

    out = run_inference(gen, examples, tokenizer.decode, batch_size=args.decode_batch, rank=rank, world=world,
                        device="cpu" if os.environ.get("DUALHYP_DP_REHEARSAL") == "1" or world == 1 else model.transformer.wte.weight.device)
    out["adapter_path"] = adapter_path
    if mask_stats:
        out.update(mask_stats)
        if rank == 0:
            print("reliability masks predicted by the classifiers:", mask_stats)
    if rank == 0:
        n = out["n"]
        to_json = list(out["predictions"])
        to_json.append({"wer": out["WER"], "gtms": f"{round(out['gtms'] * n)}/{n}"})
        to_json.append({"post_wer": out["post_ST_wer"], "post_gtms": out["post_gtms"]})
        os.makedirs(args.predict_dir, exist_ok=True)
        stem = Path(adapter_path).name.replace(".pth", "") if adapter_path else "random_init"
        path = Path(args.predict_dir) / f"{stem}.json"
        path.write_text(json.dumps(to_json, indent=4, ensure_ascii=False))
        out["predictions_file"] = str(path)
        print(f"\nFor {adapter_path}\nWER is {out['WER']}\nGround truth matches is {round(out['gtms'] * n)}/{n}")
        print(f"the post string normalization wer is\nWER {out['post_ST_wer']}\nResults in {path}")
    return out


def main(argv: Optional[Sequence[str]] = None) -> Dict[str, Any]:
    """`python -m dualhyp_amd.inference --test_path x.json --model_path runs/exp/best_model.pth --llm_checkpoint
    checkpoints/TinyLlama/TinyLlama-1.1B-Chat-v1.0 --dual_hypotheses --prompts_format DualHyp` — the flags of
    inference/ger.py:129-153; `--d N` shards the utterances over N GPUs (one process each)."""
    import argparse
    import random
    from pathlib import Path
    p = argparse.ArgumentParser(prog="python -m dualhyp_amd.inference")
    p.add_argument("--test_path", type=str, required=True)
    p.add_argument("--model_path", type=str, default="", help="fine-tuned checkpoint ({'model': state_dict}); empty with --random_init")
    p.add_argument("--llm_checkpoint", type=str, default="checkpoints/TinyLlama/TinyLlama-1.1B-Chat-v1.0")
    p.add_argument("--nhyps_key", type=str, default="nhyps_asr")
    p.add_argument("--dual_hypotheses", action="store_true")
    p.add_argument("--max_nhyps", type=int, default=None)
    p.add_argument("--d", type=int, default=1, help="number of GPUs")
    p.add_argument("--audio_corruption_disabled", action="store_true", help="accepted for compatibility: the LLM path never reads the media")
    p.add_argument("--visual_corruption_disabled", action="store_true", help="accepted for compatibility")
    p.add_argument("--seed", type=int, default=1337)
    p.add_argument("--prompts_format", type=str, default="GER")
    p.add_argument("--apply_chat_template", action="store_true", help="unsupported here (phi-3.5 only in the reference)")
    p.add_argument("--language", type=str, default=None)
    add_lora_arguments(p)
    # additions of this build
    p.add_argument("--tokenizer", choices=("auto", "hf", "byte"), default="auto")
    p.add_argument("--config_name", type=str, default=None, help="Config.from_name key (default: the checkpoint directory's name)")
    p.add_argument("--random_init", action="store_true", help="synthetic weights from the counter hash instead of --model_path")
    p.add_argument("--decode_batch", type=int, default=640,
                   help="utterances decoded jointly (one weight stream per step for all of them; bench.py: 260 utt/s at 32, 770 at 640); "
                        "a sequence's tokens do not depend on it")
    p.add_argument("--prefill_batch", type=int, default=64, help="utterances per packed prefill launch inside a decode batch")
    p.add_argument("--max_new_tokens", type=int, default=150, help="inference/ger.py:71")
    p.add_argument("--predict_dir", type=str, default=None)
    # RelPrompt (inference/relprompt.py): chunk geometry of the reliability masks; with --enc_features_dir (<dir>/<Uid>.pt =
    # {'audio': [T, whisper_dim], 'visual': [T, raven_dim]}: the encoders are upstream of this path) the masks are predicted
    p.add_argument("--mask_threshold", type=int, default=None)
    p.add_argument("--time_window", type=float, default=0.4)
    p.add_argument("--pool_size", type=int, default=10)
    p.add_argument("--enc_features_dir", type=str, default=None)
    args = p.parse_args(argv)
    if args.apply_chat_template:
        raise NotImplementedError("--apply_chat_template is outside the hot path")
    rank, world, dev = init_distributed(args.d)
    random.seed(args.seed)
    torch.manual_seed(args.seed)
    from .gpt import GPT
    from .relprompt import GPT as RelGPT
    from .tokenizer import load_tokenizer
    cfg = config_from_args(args)
    tokenizer = load_tokenizer(args.llm_checkpoint, args.tokenizer)
    rel = args.prompts_format == "RelPrompt"
    if rel:
        cfg.pool_size = args.pool_size
    model = (RelGPT if rel else GPT)(cfg)
    if rel:                                        # inference/relprompt.py:341-342
        tokenizer.add_reliability_tokens(cfg.padded_vocab_size)
        model.resize_token_embeddings(3)
    if args.random_init:
        from .synth import synth_state_dict
        sd = synth_state_dict(cfg, seed=args.seed, embed_scale=50.0, head_tie=1.0)
        if rel:                                    # keep the freshly drawn reliability rows
            sd["transformer.wte.weight"] = torch.cat([sd["transformer.wte.weight"], model.transformer.wte.weight.data[-3:].to(torch.bfloat16)])
        model.load_state_dict(sd, strict=not rel)
    model = model.to(device=dev, dtype=torch.bfloat16)
    model.eval()
    if args.predict_dir is None:
        args.predict_dir = str(Path(args.model_path).parent / "predictions") if args.model_path else "predictions"
    out = result(args.model_path, model, tokenizer, args, rank, world)
    if rank == 0:
        print("Model: ", args.model_path, "WER: ", out["WER"] * 100, "WER_post: ", out["post_ST_wer"] * 100, "GTM: ", out["gtms"] * 100,
              "GTM_post: ", out["post_gtms"] * 100)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
