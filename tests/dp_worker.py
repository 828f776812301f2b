"""Rank program of the data-parallel equivalence tests (tests/test_harness.py): launched with torch.distributed.run,
two ranks sharing cuda:0 over gloo (DUALHYP_DP_REHEARSAL=1 — a one-GPU box has no second device; on an 8-GPU node
the same code runs over RCCL).  Writes rank 0's results to --out."""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def toy_examples(n, vocab):
    from dualhyp_amd.synth import hash_u24, stream_id
    exs = []
    for i in range(n):
        T = 18 + (i * 5) % 11
        ids = hash_u24(T, stream_id(9, f"dp{i}")) % (vocab - 3) + 3
        lab = ids.clone()
        lab[: T - 7] = -1
        exs.append({"input_ids": ids, "labels": lab, "input_ids_no_response": ids[: T - 7], "input": "", "uid": str(i),
                    "ground_truth": f"ref {i}"})
    return exs


def build_model(dev):
    from dualhyp_amd import GPT, Config
    from dualhyp_amd.synth import synth_state_dict
    cfg = Config.from_name("parity-tiny", r=4, alpha=8, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
    m = GPT(cfg).to(device=dev, dtype=torch.bfloat16)
    m.load_state_dict(synth_state_dict(cfg, seed=5, weight_scale=4.0, embed_scale=64.0, head_tie=1.0, device=dev))
    return cfg, m


def run(out_path, global_batch):
    from dualhyp_amd.data import collate
    from dualhyp_amd.finetune import TrainConfig, fit
    from dualhyp_amd.generate import generate_batch
    from dualhyp_amd.inference import init_distributed, run_inference
    rank, world, dev = init_distributed(int(os.environ.get("WORLD_SIZE", "1")))
    cfg, m = build_model(dev)
    exs = toy_examples(16, cfg.padded_vocab_size)
    # ---- sharded inference through the real decode path
    m.eval()
    gen = lambda ps: [o.cpu() for o in generate_batch(m, [p.to(dev) for p in ps], 6, temperature=0.2, top_k=1, prefill_batch=4)]
    dec = lambda ids: " ".join(str(int(i)) for i in ids)
    inf = run_inference(gen, exs[:7], dec, batch_size=3, rank=rank, world=world)
    # ---- data-parallel fine-tune: global batch fixed, per-rank accumulation = global_batch / world
    tc = TrainConfig(learning_rate=2e-3, num_epochs=1, batch_size=global_batch, micro_batch_size=1, lm_head_chunk_size=8)
    stats = fit(m, exs, collate, tc, rank=rank, world=world, device=dev, log=lambda s: None)
    if rank == 0:
        torch.save({"world": world, "stats": stats, "inference": inf,
                    "lora": {k: p.detach().float().cpu() for k, p in m.named_parameters() if "lora_" in k}}, out_path)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--global_batch", type=int, default=4)
    a = ap.parse_args()
    run(a.out, a.global_batch)
