"""LoRA fine-tune loop (finetune/ger.py:86-353) for one process per GPU.

What is kept from the reference: AdamW(lr, weight_decay=0.02) on the LoRA parameters only
(finetune/ger.py:126-133), linear warm-up to `lr` then constant or cosine (`:255-266`), loss
= chunked CE over lm-head chunks of 128 with the per-position mean of quirk Q5 (`:278-281`),
`loss / gradient_accumulation_iters` (`:285`), validation = unchunked CE on the valid positions
(`:331-353`), best-checkpoint-on-val-loss as `{"model": state_dict}` (`:312-316,356-358`).

What is deliberately different:
  * data is SHARDED: every rank walks a strided slice of one seed-1337 permutation per epoch, so
    1 GPU x 32 accumulation and 8 GPUs x 4 see the same global batch (the reference gives every rank
    the full shuffled set);
  * gradients ARE synchronised: one all-reduce (RCCL over xGMI) of a single flat fp32 bucket holding
    every LoRA gradient (4 505 600 elements for TinyLlama r=16) per optimizer step — the reference's
    `no_backward_sync` flag is true on every micro-step, so with d > 1 it never reduces;
  * the accumulation off-by-one (an optimizer step every 31 micro-batches while dividing by 32) is
    reproduced only with `reference_accumulation=True` (default False: step every `accum`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence

import torch

from .utils import chunked_cross_entropy


def lr_at(it: int, lr: float, warmup_steps: int, max_iters: int = 0, use_cosine: bool = False,
          min_lr_ratio: float = 0.0) -> float:
    """finetune/ger.py:255-266 (note `<=`: the peak is reached AT `warmup_steps`)."""
    if it <= warmup_steps:
        return lr * it / warmup_steps
    if use_cosine:
        prog = min((it - warmup_steps) / (max_iters - warmup_steps), 1.0)
        lo = lr * min_lr_ratio
        return lo + (lr - lo) * (1 + math.cos(math.pi * prog)) / 2
    return lr


def step_schedule(n_micro: int, accum: int, reference_accumulation: bool = False) -> List[int]:
    """Indices of the micro-batches after which the optimizer steps.  Reference semantics
    (finetune/ger.py:272-292): `micro_step += 1` happens before the `(micro_step + 1) % accum == 0`
    test, so a step fires after accum-1 micro-batches and the counter restarts (quirk Q3)."""
    out, micro = [], 0
    for i in range(n_micro):
        micro += 1
        hit = (micro + 1) % accum == 0 if reference_accumulation else micro % accum == 0
        if hit:
            out.append(i)
            micro = 0
    return out


def epoch_order(n: int, epoch: int, rank: int, world: int, seed: int = 1337, shuffle: bool = True) -> List[int]:
    """This rank's utterances for one epoch: strided slice of a seeded permutation (same on every rank)."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
    usable = n - n % world                      # equal work per rank: drop the ragged tail
    return perm[rank:usable:world]


class FlatGradBucket:
    """All LoRA gradients as ONE contiguous fp32 buffer so a data-parallel step is a single collective
    (18 MB for TinyLlama r=16: latency-bound on the 8-GPU xGMI mesh)."""

    def __init__(self, params: Sequence[torch.nn.Parameter]) -> None:
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)     # autograd accumulates in place into the bucket
            off += p.numel()

    def zero(self) -> None:
        self.flat.zero_()
        off = 0
        for p in self.params:                                       # re-attach (optimizers may set grads to None)
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def all_reduce_mean(self) -> None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():        # also with one rank under a launcher: the collective still runs
            if dist.get_backend() == "gloo" and self.flat.is_cuda:      # rehearsal on one GPU: stage through the host
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)        # RCCL over xGMI: one 18 MB bucket
            self.flat.div_(dist.get_world_size())


@dataclass
class TrainConfig:
    learning_rate: float = 1e-4
    weight_decay: float = 0.02
    num_epochs: int = 5
    batch_size: int = 32                 # global batch per optimizer step (finetune/ger.py:381)
    micro_batch_size: int = 1
    warmup_frac: float = 0.2             # warmup_steps = int(epoch_size * 0.2) // world  (`:182`)
    use_cosine_scheduler: bool = False
    min_lr_ratio: float = 0.0
    lm_head_chunk_size: int = 128
    save_interval: int = 0               # micro-iterations between validations; 0 = only at the end
    reference_accumulation: bool = False
    shuffle: bool = True                 # False: utterances in file order (trajectory fixtures)
    # RelPrompt (finetune/relprompt.py:625,641): the reliability classifiers train beside the LoRA parameters, in
    # their own AdamW group with their own learning rate, on the mask cross entropy weighted by mask_loss_weight
    classifier_learning_rate: float = 1e-4
    mask_loss_weight: float = 0.02
    # up to `pack` micro-batches of one accumulation window run as ONE packed forward / backward replayed from a
    # hipGraph (train.GraphedTrainStep): same per-sequence losses and the sum of their gradients, but the GEMMs see
    # pack x T rows.  1 = every micro-batch on its own through torch autograd (the reference's schedule, launch by launch)
    pack: int = 1


def micro_loss(model, input_ids: torch.Tensor, labels: torch.Tensor, chunk: int) -> torch.Tensor:
    """finetune/ger.py:278-281."""
    logits = model(input_ids, lm_head_chunk_size=chunk)
    logits[-1] = logits[-1][..., :-1, :]
    return chunked_cross_entropy(logits, labels[..., 1:], chunk_size=chunk)


def _all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """SUM over the ranks of a small tensor (validation totals, flags): RCCL on device tensors, staged through the host under the
    one-GPU gloo rehearsal; the identity without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if dist.get_backend() == "gloo" and t.is_cuda:
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        return host.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def _barrier() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


@torch.no_grad()
def validate(model, batches: Iterable[Dict[str, torch.Tensor]], rank: int = 0, world: int = 1) -> float:
    """finetune/ger.py:331-353: mean over batches of the CE on valid positions; all-masked batches skipped.

    Data-parallel form (the reference runs the WHOLE set on every rank and synchronises once per batch through `.item()`):
    rank r takes batches r, r + world, ... of the iterable (host tensors are moved to the model's device only for those),
    accumulates (loss sum, batch count) on the device, and ONE all-reduce of the two numbers + ONE host read end the
    validation — every rank returns the same global mean, so the `best_val` decision is identical everywhere."""
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    tot = torch.zeros(2, dtype=torch.float64, device=dev)            # [sum of per-batch losses, batches counted]
    for i, b in enumerate(batches):
        if i % world != rank:
            continue
        ids, tg = b["input_ids"], b["labels"]
        # the skip test of finetune/ger.py:341-344 on the HOST copy of the labels when there is one (no device sync)
        if int((tg[..., 1:] != -1).sum()) == 0:
            continue
        ids, tg = ids.to(dev), tg.to(dev)
        lg = model(ids)
        tot[0] += chunked_cross_entropy(lg[..., :-1, :], tg[..., 1:], chunk_size=0).double()
        tot[1] += 1
    model.reset_cache()
    model.train(was_training)
    if world > 1:
        tot = _all_reduce_sum(tot)
    s, n = tot.tolist()                                              # the validation's one host synchronisation
    return s / max(n, 1.0)


from .checkpoint import save_checkpoint  # noqa: E402,F401  (finetune/ger.py:356-358)


def fit(model, train_examples: Sequence[Dict[str, torch.Tensor]], collate: Callable, cfg: TrainConfig, *,
        val_batches: Optional[Callable[[], Iterable[Dict[str, torch.Tensor]]]] = None, out_dir: Optional[str] = None,
        rank: int = 0, world: int = 1, device="cuda", log: Callable[[str], None] = print,
        on_step: Optional[Callable[[int, List[torch.nn.Parameter]], None]] = None,
        on_micro: Optional[Callable[[int, torch.Tensor], None]] = None) -> Dict[str, float]:
    """Runs the fine-tune; returns {'final_train_loss', 'best_val_loss', 'optimizer_steps'}.
    `on_step(step_index, lora_parameters)` is called after every optimizer step, `on_micro(iteration, loss)` after
    every micro-batch (tests, progress reporting); neither is needed for training."""
    from .train import prepare_for_training
    model.train()
    params = prepare_for_training(model)
    # RelPrompt: the noise-mask classifiers stay trainable (ger/relprompt.py:79-119) and form the optimizer's second
    # group with their own learning rate (finetune/relprompt.py:175-195); their gradients ride the same flat bucket
    cls_params: List[torch.nn.Parameter] = []
    if hasattr(model, "audio_noise_classifier"):
        from .relprompt import prepare_classifiers_for_training
        cls_params = prepare_classifiers_for_training(model)
    groups = [{"params": params, "lr": cfg.learning_rate, "weight_decay": cfg.weight_decay}]
    if cls_params:
        groups.append({"params": cls_params, "lr": cfg.classifier_learning_rate, "weight_decay": cfg.weight_decay})
    opt = torch.optim.AdamW(groups)
    base_lrs = [cfg.learning_rate, cfg.classifier_learning_rate]
    bucket = FlatGradBucket(params + cls_params)
    accum = max(cfg.batch_size // world // cfg.micro_batch_size, 1)
    epoch_size = len(train_examples) // cfg.micro_batch_size
    warmup = max(int(epoch_size * cfg.warmup_frac) // world, 1)
    max_iters = cfg.num_epochs * epoch_size // world
    it, micro, steps, best_val, last = 0, 0, 0, float("inf"), float("nan")
    loss_acc = torch.zeros((), device=device)          # no per-micro-step .item(): one host sync per log line
    packed = None
    if cfg.pack > 1:
        assert cfg.micro_batch_size == 1 and not cls_params, "TrainConfig.pack: micro_batch_size 1, decoder-only fine-tune"
        from .train import GraphedTrainStep
        packed = GraphedTrainStep(model, bucket)
    pending: List[tuple] = []                           # (iteration, example) of micro-batches not yet run (pack > 1)
    saves = 0

    def checkpoint_if_best(tag: str) -> None:
        """finetune/ger.py:309-317: validate, save `best_model.pth` when the validation loss improved, barrier.  The loss is the
        global mean (validate all-reduces), so every rank takes the same branch; rank 0 alone writes (the LoRA tensors are
        replicated: every rank holds the same ones after the all-reduced optimizer step) and the others wait at the barrier, so
        the next optimizer step cannot start under the save."""
        nonlocal best_val, saves
        v = validate(model, val_batches(), rank, world)
        log(f"{tag} val loss {v:.4f}")
        if v < best_val:
            best_val = v
            if out_dir and rank == 0:
                save_checkpoint(model, Path(out_dir) / "best_model.pth")
                saves += 1
        if world > 1:
            _barrier()

    def flush() -> None:
        """run the pending micro-batches as one packed step; per-micro-batch bookkeeping as if they had run in turn"""
        nonlocal loss_acc
        if not pending:
            return
        exs = [ex for _, ex in pending]
        batch = collate(exs)
        losses = packed(batch["input_ids"].to(device), batch["labels"].to(device), 1.0 / accum,
                        lengths=[int(ex["input_ids"].numel()) for ex in exs],
                        n_targets=int((batch["labels"][:, 1:] != -1).sum()))      # counted on the host copy: no device sync
        loss_acc += losses.sum()
        if on_micro is not None:
            for k, (it_k, _) in enumerate(pending):
                on_micro(it_k, losses[k])
        pending.clear()

    for epoch in range(cfg.num_epochs):
        order = epoch_order(len(train_examples), epoch, rank, world, shuffle=cfg.shuffle)
        for b0 in range(0, len(order) - cfg.micro_batch_size + 1, cfg.micro_batch_size):
            for g, base in zip(opt.param_groups, base_lrs):       # both groups follow the same schedule (relprompt.py:321-341)
                g["lr"] = lr_at(it, base, warmup, max_iters, cfg.use_cosine_scheduler, cfg.min_lr_ratio)
            if packed is not None:
                pending.append((it, train_examples[order[b0]]))
            else:
                batch = collate([train_examples[i] for i in order[b0:b0 + cfg.micro_batch_size]])
                ids, labels = batch["input_ids"].to(device), batch["labels"].to(device)
                loss = micro_loss(model, ids, labels, cfg.lm_head_chunk_size)
                if cls_params and "audio_enc_features" in batch and "audio_mask_targets" in batch:
                    from .relprompt import mask_loss                   # finetune/relprompt.py:356-403
                    a_lg = model.audio_noise_classifier(batch["audio_enc_features"].to(device))
                    v_lg = model.visual_noise_classifier(batch["visual_enc_features"].to(device))
                    loss = loss + cfg.mask_loss_weight * mask_loss(a_lg, v_lg, batch["audio_mask_targets"].to(device),
                                                                   batch["visual_mask_targets"].to(device))
                (loss / accum).backward()
                loss_acc += loss.detach()
                if on_micro is not None:
                    on_micro(it, loss.detach())
            micro += 1
            hit = (micro + 1) % accum == 0 if cfg.reference_accumulation else micro % accum == 0
            if len(pending) >= cfg.pack or hit or (cfg.save_interval and (it + 1) % cfg.save_interval == 0):
                flush()
            if hit:
                bucket.all_reduce_mean()
                opt.step()
                bucket.zero()
                model.refresh_engine()
                if on_step is not None:
                    on_step(steps, params + cls_params)
                micro, steps = 0, steps + 1
            it += 1
            if cfg.save_interval and it % cfg.save_interval == 0:
                last = (loss_acc / cfg.save_interval).item()
                loss_acc.zero_()
                if val_batches is not None:
                    checkpoint_if_best(f"iter {it}: train loss {last:.4f}")
    flush()     # micro-batches after the last optimizer step of the run (their gradients are dropped, as in the reference's loop)
    if val_batches is not None:
        checkpoint_if_best(f"iter {it} (end)")
    if out_dir and rank == 0:
        save_checkpoint(model, Path(out_dir) / "lit_model_lora_finetuned.pth")
    if world > 1:
        _barrier()                          # nobody leaves (and tears the group down) while rank 0 still writes
    return {"final_train_loss": last, "best_val_loss": best_val, "optimizer_steps": steps, "checkpoints_written": saves}


# ------------------------------------------------------------------------------------------ harness
def main(argv: Optional[Sequence[str]] = None) -> Dict[str, float]:
    """`python -m dualhyp_amd.finetune --train_path train.json --val_path val.json --dual_hypotheses --prompts_format
    DualHyp --micro_batch_size 1 --lr 1e-4 --num_epochs 5` — the flags of finetune/ger.py:373-407.  Base weights come
    from `<llm_checkpoint>/lit_model.pth` (finetune/ger.py:113,122-124); `--d N` trains data-parallel on N GPUs, one
    process each, with one RCCL all-reduce of the flat LoRA-gradient bucket per optimizer step."""
    import argparse
    import json
    import logging
    import os
    import random
    import time
    from .inference import add_lora_arguments, config_from_args, init_distributed
    p = argparse.ArgumentParser(prog="python -m dualhyp_amd.finetune")
    p.add_argument("--train_path", type=str, nargs="+", required=True)
    p.add_argument("--val_path", type=str, default=None)
    p.add_argument("--exp_name", type=str, default="finetune")
    p.add_argument("--llm_checkpoint", type=str, default="checkpoints/TinyLlama/TinyLlama-1.1B-Chat-v1.0")
    p.add_argument("--nhyps_key", type=str, default="nhyps_asr")
    p.add_argument("--dual_hypotheses", action="store_true")
    p.add_argument("--max_nhyps", type=int, default=None)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--micro_batch_size", type=int, default=1)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--num_epochs", type=int, default=5)
    p.add_argument("--weight_decay", type=float, default=0.02)
    p.add_argument("--d", type=int, default=1, help="number of GPUs")
    p.add_argument("--lamda", type=float, default=0.5, help="accepted for compatibility (unused in finetune/ger.py too)")
    p.add_argument("--wp", type=float, default=0.2, help="warm-up proportion of an epoch")
    p.add_argument("--use_cosine_scheduler", action="store_true")
    p.add_argument("--min_lr_ratio", type=float, default=0.01)
    p.add_argument("--log_interval", type=int, default=100)
    p.add_argument("--save_interval", type=int, default=10000)
    p.add_argument("--audio_corruption_disabled", action="store_true", help="accepted for compatibility")
    p.add_argument("--visual_corruption_disabled", action="store_true", help="accepted for compatibility")
    p.add_argument("--prompts_format", type=str, default="GER")
    p.add_argument("--apply_chat_template", action="store_true")
    p.add_argument("--language", type=str, default=None)
    add_lora_arguments(p)
    # additions of this build
    p.add_argument("--tokenizer", choices=("auto", "hf", "byte"), default="auto")
    p.add_argument("--config_name", type=str, default=None)
    p.add_argument("--random_init", action="store_true", help="synthetic base weights instead of <llm_checkpoint>/lit_model.pth")
    p.add_argument("--out_dir", type=str, default=None, help="default ./runs/<exp_name>")
    p.add_argument("--reference_accumulation", action="store_true", help="quirk Q3: step every batch_size-1 micro-batches")
    p.add_argument("--seed", type=int, default=1337)
    p.add_argument("--pack", type=int, default=8,
                   help="micro-batches of one accumulation window run as one packed, hipGraph-replayed step (same losses, summed "
                        "gradients; 1 = the reference's launch-by-launch schedule)")
    # finetune/relprompt.py:625,641-643 (RelPrompt: the reliability classifiers train in a second AdamW group)
    p.add_argument("--classifier_lr", type=float, default=1e-4)
    p.add_argument("--mask_loss_weight", type=float, default=0.02)
    p.add_argument("--mask_threshold", type=int, default=None)
    p.add_argument("--time_window", type=float, default=0.4)
    p.add_argument("--pool_size", type=int, default=10)
    p.add_argument("--enc_features_dir", type=str, default=None,
                   help="addition of this build: <dir>/<Uid>.pt = {'audio': [T, whisper_dim], 'visual': [T, raven_dim]} encoder features "
                        "(the Whisper / BRAVEn encoders of finetune/relprompt.py:346-352 are upstream of this path)")
    args = p.parse_args(argv)
    if args.apply_chat_template:
        raise NotImplementedError("--apply_chat_template is outside the hot path")
    rank, world, dev = init_distributed(args.d)
    out_dir = Path(args.out_dir or f"./runs/{args.exp_name}")
    if rank == 0:
        out_dir.mkdir(parents=True, exist_ok=True)
        logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s",
                            handlers=[logging.FileHandler(out_dir / "train.log"), logging.StreamHandler()], force=True)
        logging.info(f"CLI arguments: {args.__dict__}")
    random.seed(args.seed + rank)
    torch.manual_seed(args.seed + rank)                    # finetune/ger.py:135
    from .checkpoint import load_checkpoint
    from .data import HypothesesDataset, collate
    from .gpt import GPT
    from .tokenizer import load_tokenizer
    cfg = config_from_args(args)
    tokenizer = load_tokenizer(args.llm_checkpoint, args.tokenizer)
    max_input_length = 1024                                # finetune/ger.py:417-421
    tc_path = Path(args.llm_checkpoint) / "tokenizer_config.json"
    if tc_path.is_file():
        max_input_length = json.loads(tc_path.read_text()).get("model_max_length") or 1024
    max_input_length = min(int(max_input_length), cfg.block_size)
    rel = args.prompts_format == "RelPrompt"
    if rel:                                                # finetune/relprompt.py:141-147: classifiers + 3 reliability tokens
        from .relprompt import GPT as RelGPT
        cfg.pool_size = args.pool_size
        model = RelGPT(cfg)
    else:
        model = GPT(cfg)
    if args.random_init:
        from .synth import synth_state_dict
        model.load_state_dict(synth_state_dict(cfg, seed=args.seed), strict=not rel)
    else:
        model.load_state_dict(load_checkpoint(Path(args.llm_checkpoint) / "lit_model.pth"), strict=False)   # LoRA tensors keep their init
    if rel:
        tokenizer.add_reliability_tokens(cfg.padded_vocab_size)
        model.resize_token_embeddings(3)
    model = model.to(device=dev, dtype=torch.bfloat16)
    fmt = args.prompts_format if (args.dual_hypotheses or rel) else "GER"
    enc_features = None
    _variants: dict = {}          # Uid -> its items, filled by dataset() below
    if rel and args.enc_features_dir:
        def enc_features(s1, s2, _d=Path(args.enc_features_dir)):
            # <dir>/<Uid>.pt = {'audio', 'visual'} when the Uid has one variant; with several, the audio features of s1's corruption
            # and the visual features of s2's come from <Uid>.<hash of that corruption record>.pt (data.feature_key)
            from .data import feature_key
            n_var = len(_variants.get(s1["Uid"], (s1,)))
            fa = torch.load(_d / f"{feature_key(s1, 'Audio_Corruption', n_var)}.pt", map_location="cpu")
            fv = fa if n_var <= 1 else torch.load(_d / f"{feature_key(s2, 'Visual_Corruption', n_var)}.pt", map_location="cpu")
            return fa["audio"].float(), fv["visual"].float()

    def dataset(path, seed):
        items = []
        for one in ([path] if isinstance(path, str) else path):
            with open(one, encoding="utf-8") as f:
                items += json.load(f)
        for it in items:
            _variants.setdefault(it["Uid"], []).append(it)
        return HypothesesDataset(items, tokenizer, prompts_format=fmt, nhyps_key=args.nhyps_key, max_nhyps=args.max_nhyps,
                                 max_input_length=max_input_length, language=args.language, seed=seed,
                                 mask_threshold=args.mask_threshold, time_window=args.time_window, enc_features=enc_features)
    train_ds = dataset(args.train_path, args.seed + rank)
    train = [train_ds[i] for i in range(len(train_ds))]
    val_batches = None
    if args.val_path:
        val_ds = dataset(args.val_path, args.seed)
        val = [val_ds[i] for i in range(len(val_ds))]

        def val_batches():                 # host tensors of the WHOLE set: validate() takes this rank's share and moves only that
            for b in range(0, len(val), args.micro_batch_size):
                c = collate(val[b:b + args.micro_batch_size])
                yield {"input_ids": c["input_ids"], "labels": c["labels"]}
    tc = TrainConfig(learning_rate=args.lr, weight_decay=args.weight_decay, num_epochs=args.num_epochs, batch_size=args.batch_size,
                     micro_batch_size=args.micro_batch_size, warmup_frac=args.wp, use_cosine_scheduler=args.use_cosine_scheduler,
                     min_lr_ratio=args.min_lr_ratio, save_interval=max(args.save_interval // world, 1),
                     reference_accumulation=args.reference_accumulation, classifier_learning_rate=args.classifier_lr,
                     mask_loss_weight=args.mask_loss_weight,
                     pack=1 if (rel or args.micro_batch_size != 1) else max(args.pack, 1))
    log = logging.info if rank == 0 else (lambda s: None)
    t0 = time.perf_counter()
    out = fit(model, train, collate, tc, val_batches=val_batches, out_dir=str(out_dir), rank=rank, world=world, device=dev, log=log)
    if rank == 0:
        logging.info(f"Total training time: {time.perf_counter() - t0:.2f}s")
        logging.info(f"Memory used: {torch.cuda.max_memory_allocated() / 1e9:.02f} GB")
        logging.info(f"result: {out}")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()        # fit() ended on a barrier behind rank 0's last checkpoint
    return out


if __name__ == "__main__":
    main()
