#!/usr/bin/env python3
"""Headline benchmark: corrected utterances/s of the DualHyp decoder on MI355X.

Workload (BASELINE.json configs[1]): TinyLlama-1.1B + LoRA r=16 on q,k,v,proj, bf16, batch of
32 synthetic 5+5-hypothesis prompts of 512 tokens per GPU, 64 generated tokens each
(eos disabled so exactly 64 are produced), greedy (top_k=1, temperature 0.2) — one "step" is
one such batch through packed prefill + 63 hipGraph decode steps.  Weights are random-init of
the TinyLlama architecture from a counter hash (no checkpoints offline); prompts are
token-id level (no tokenizer offline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Data-parallel replicas: every rank decodes its own batch, no collective in the data path
(SURVEY.md §8e); value = utterances of all ranks / max-over-ranks time.  Rank 0 prints ONE JSON
line with the `roofline` of the dominant kernel class (prefill MFMA GEMMs, timed live with HIP
events on the launch stream) and, at N=1, the `cpu_baseline` (the oracle = CPU restatement of
the reference, timed on this host's cores on 8 utterances of the same workload after one warm-up).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

PROMPT_LEN, NEW_TOKENS, BATCH = 512, 64, 32
# --config: the headline (BASELINE configs[1]) and the Llama-3-8B side lines (configs[4]: ~1.5k-token 10+10-hypothesis
# prompts; fp8 = merged LoRA, e4m3 weights with per-channel scales, fp8 MFMA GEMMs)
WORKLOADS = {
    "tinyllama-bf16": dict(model="tiny-llama-1.1b-chat", prompt=512, fp8=False, peak=2500.0, in_flight=64, prefill_batches=2,
                           what="DualHyp inference, TinyLlama-1.1B bf16 + LoRA r16 (q,k,v,proj), batch 32/GPU synthetic 5+5-hyp prompts, "
                                "512-token prompt -> 64 generated tokens, greedy",
                           metric="corrected utterances/sec (TinyLlama-1.1B, 5+5 hyps, 512->64 tok)", dtype="bf16",
                           kernel="gemm_nt256w4_kernel (prefill GEMMs: qkv+LoRA, proj+LoRA, fc_1/fc_2 SwiGLU, mlp proj)"),
    "llama3-8b-bf16": dict(model="Llama-3-8B", prompt=1536, fp8=False, peak=2500.0, in_flight=4, prefill_batches=1,
                           what="DualHyp inference, Llama-3-8B bf16 + LoRA r16, batch 32/GPU synthetic 10+10-hyp prompts, 1536-token prompt -> 64 tokens, greedy",
                           metric="corrected utterances/sec (Llama-3-8B, 10+10 hyps, 1536->64 tok)", dtype="bf16",
                           kernel="gemm_nt256w4_kernel (prefill GEMMs)"),
    "finetune-tinyllama": dict(model="tiny-llama-1.1b-chat", prompt=560, fp8=False, peak=2500.0, in_flight=1, prefill_batches=1, train=True,
                               what="DualHyp LoRA fine-tune (finetune/ger.py --dual_hypotheses), TinyLlama-1.1B bf16 base + fp32 LoRA masters r16, "
                                    "micro-batch 1 x 560 tokens (512 masked prompt + 47 response + EOS), optimizer step = 32 utterances "
                                    "shared by the ranks, one flat-bucket all-reduce of the LoRA gradients per step, AdamW",
                               metric="fine-tuned utterances/sec (TinyLlama-1.1B LoRA r16, T=560, global batch 32)", dtype="bf16",
                               kernel="whole packed micro-step (forward + chunked CE + backward of --pack sequences, one hipGraph replay)"),
    "llama3-8b-fp8": dict(model="Llama-3-8B", prompt=1536, fp8=True, peak=5000.0, in_flight=4, prefill_batches=1,
                          what="DualHyp inference, Llama-3-8B with merged LoRA, fp8 e4m3 weights (per-channel scales) and per-token fp8 "
                               "activations, batch 32/GPU synthetic 10+10-hyp prompts, 1536-token prompt -> 64 tokens, greedy",
                          metric="corrected utterances/sec (Llama-3-8B fp8, 10+10 hyps, 1536->64 tok)", dtype="fp8",
                          kernel="gemm_fp8_kernel (prefill GEMMs on v_mfma_scale_f32_16x16x128_f8f6f4: qkv, proj, fc_1/fc_2 SwiGLU, mlp proj)"),
}
# random-init weights with the head tied to the (scaled) embedding through a fixed permutation (dualhyp_amd.synth):
# same arithmetic and bytes as plain random init, but the arg-max is separated from the runner-up by tens of bf16 ulps
# (>= 20 sigma of the noise between two bf16 implementations) on every step, so the greedy ids of the HIP run can be
# compared with the oracle's token for token (the `parity` object); plain N(0, 0.02) logits are near-ties on a
# quarter of the steps
SYNTH_KW = dict(embed_scale=50.0, head_tie=1.0)
MFMA_PEAK_TFLOPS = 2500.0     # bf16 dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def gemm_flops_per_prefill(cfg, n_tok: int, n_seq: int, merged_lora: bool = False, last_rows_only: bool = False) -> float:
    """Algorithmic FLOPs of the GEMM launches of one packed prefill THAT ARE EXECUTED: per layer qkv + proj +
    fc_1 + fc_2 + mlp proj on every token; LoRA rank-16 side products; the lm_head runs on the last position only (M = n_seq
    <= 32: not in this class).  `last_rows_only` (the engine's prompt forward, csrc/engine.hip g_prune_last_layer): the
    LAST block's proj / fc_1 / fc_2 / mlp proj run on the n_seq last rows only, as small weight-streaming launches OUTSIDE the
    timed class: neither the skipped rows' FLOPs nor those launches' FLOPs are counted."""
    d, I = cfg.n_embd, cfg.intermediate_size
    qkv = (cfg.n_head + 2 * cfg.n_query_groups) * cfg.head_size
    per_tok = 2 * d * (qkv + d + 3 * I)
    lora = (2 * d * (48 + 16) + 2 * 16 * (qkv + d)) if merged_lora is False else 0
    total = float(cfg.n_layer * n_tok * (per_tok + lora))
    if last_rows_only:
        tail = 2 * d * (d + 3 * I) + ((2 * d * 16 + 2 * 16 * d) if merged_lora is False else 0)   # per row of the last block after QKV
        total -= float(n_tok * tail)
    return total


def decode_bytes_per_step(cfg, rows: float, mean_ctx: float, weight_bytes: int, merged_lora: bool) -> float:
    """Algorithmic HBM bytes of one decode step: every dense weight once for all rows + each row's
    KV prefix (compact GQA cache, bf16)."""
    d, I = cfg.n_embd, cfg.intermediate_size
    qkv = (cfg.n_head + 2 * cfg.n_query_groups) * cfg.head_size
    layer = d * (qkv + d + 3 * I)
    lora = 0 if merged_lora else cfg.n_layer * (64 * d + 16 * (qkv + d))
    weights = weight_bytes * (cfg.n_layer * layer + cfg.padded_vocab_size * d) + 2 * lora
    kv = cfg.n_layer * 2 * cfg.n_query_groups * cfg.head_size * 2
    return float(weights + rows * kv * mean_ctx)


# HBM-side traffic of the dominant kernel class, per launch, from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled + WRITE_SIZE,
# MI355X_MICROARCH.md §HBM; separate passes, so it cannot be collected inside the timed run): newest file first
PMC_TRAFFIC = {
    "tinyllama-bf16": ["profiles/r04_pmc_gemm.json", "profiles/r03_pmc_gemm_w4.json"],
    "llama3-8b-fp8": ["profiles/r04_pmc_fp8_gemm256.json", "profiles/r03_pmc_fp8_gemm256.json"],
}


def pmc_traffic(config: str):
    """(bytes per launch or None, where it came from / why it is null)"""
    for rel in PMC_TRAFFIC.get(config, []):
        path = ROOT / rel
        if not path.exists():
            continue
        with open(path) as fh:
            d = json.load(fh)
        if "traffic_bytes_per_launch_mean" in d:
            return d["traffic_bytes_per_launch_mean"], (f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_gemm.py, mean bytes per launch "
                                                          "of the layer's four GEMM launches; L2-miss traffic on the fabric, Infinity-Cache hits included)")
        if "derived" in d:   # per-kernel fabric bytes (tools/pmc_fp8_summary.py): mean over the kernels of the class
            v = [k["fabric_read_bytes"] + k["fabric_write_bytes"] for k in d["derived"].values() if "fabric_read_bytes" in k]
            if v:
                return sum(v) / len(v), f"{rel} (rocprofv3 --pmc passes over tools/pmc_fp8.py at M = 49152: mean fabric read + write bytes per launch of the profiled kernels)"
    return None, f"no PMC pass committed for --config {config} (profiles/ holds them for: {', '.join(sorted(PMC_TRAFFIC))})"


def _sysfs_clocks(dev_index: int) -> dict:
    """current sclk / mclk of the card from sysfs (the starred line of pp_dpm_*), best effort: {} when not readable"""
    out = {}
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        base = Path("/sys/bus/pci/devices") / bdf
        for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
            txt = (base / fn).read_text()
            cur = [l for l in txt.splitlines() if l.strip().endswith("*")]
            if cur:
                out[key] = int("".join(c for c in cur[0].split(":")[1] if c.isdigit()))
        out["pci"] = bdf
    except Exception:        # noqa: BLE001 - identity is informational
        pass
    return out


def device_identity(dev_index: int) -> dict:
    """What the line was measured on (device constants recorded with every run): arch / CU count / HBM bytes from the
    library's own dh_device_info, name and maximum clocks from the HIP properties, current clocks from sysfs where readable."""
    import ctypes as C
    from dualhyp_amd import _lib
    lib = _lib.load()
    buf, cu, hbm = C.create_string_buffer(64), C.c_int(0), C.c_int64(0)
    _lib.check(lib.dh_device_info(buf, 64, C.byref(cu), C.byref(hbm)))
    p = torch.cuda.get_device_properties(dev_index)
    d = {"arch": buf.value.decode(), "compute_units": int(cu.value), "hbm_bytes": int(hbm.value), "name": p.name,
         "max_sclk_mhz": int(getattr(p, "clock_rate", 0) // 1000) or None,
         "max_mclk_mhz": int(getattr(p, "memory_clock_rate", 0) // 1000) or None,
         "hip": torch.version.hip, "torch": torch.__version__}
    d.update({f"{k}_after_run": v for k, v in _sysfs_clocks(dev_index).items()})
    return d


_RESULT_FD = None


def emit(result: dict) -> None:
    """the bench's one line, on the process's ORIGINAL stdout"""
    line = (json.dumps(result) + "\n").encode()
    sys.stdout.flush()
    if _RESULT_FD is None:
        os.write(1, line)
    else:
        os.write(_RESULT_FD, line)


def launch_cmd(n: int, argv: list) -> list:
    """The command the driver itself uses for N > 1: torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def selftest_launch(a, world: int) -> None:
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest_launch": True, "n_gpus": world, "max_over_ranks": float(t.item())}), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="tinyllama-bf16",
                    help="workload: the headline TinyLlama config (default, the driver's) or a Llama-3-8B side line")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="batches (steps) decoded together per GPU: each batch of --batch prompts is prefilled on its "
                         "own, then the decode loop runs over all in-flight sequences at once")
    ap.add_argument("--schedule", choices=("merged", "threads"), default="merged",
                    help="merged: one engine, chunked prefill + joint decode (generate_batch); threads: one engine, "
                         "HIP stream and host thread per in-flight batch (dualhyp_amd.pipeline)")
    ap.add_argument("--engines", type=int, default=2, help="--schedule threads: engines (each decodes --in-flight batches jointly)")
    ap.add_argument("--prefill-batches", type=int, default=None,
                    help="batches per prefill launch (merged schedule): at 2 x 32 x 512 tokens every GEMM of the layer is a whole "
                         "number of 256-tile rounds on 256 CUs (the qkv GEMM is 2.5 rounds at one batch)")
    ap.add_argument("--ragged", action="store_true",
                    help="SURVEY §8d ragged variant: prompt lengths uniform in [384, 640] instead of 512 (not the headline config)")
    ap.add_argument("--pack", type=int, default=32,
                    help="--config finetune-tinyllama: micro-batches of the accumulation window per packed forward/backward launch")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions of --steps steps each, run back to back; the line is the MEDIAN one (its own barriers, wall time and "
                         "HIP-event timings), the others are listed in `repeats`.  Default: 3 when --steps <= 20 (a 0.9-s region moves by +-1.5 %% "
                         "from run to run and box to box), else 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-probe", action="store_true")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="dh_set_tuning(key, value) before the run (kernel-selection experiments; recorded in config.tuning)")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="no GPU work: ranks rendezvous over gloo, reduce a value and rank 0 prints one JSON line (CPU test of the launcher)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE anything
        # here touches the GPU — a process that has initialised HIP must never re-exec — forward their output
        # (rank 0 prints the JSON line) and exit with their return code.
        sys.exit(subprocess.run(launch_cmd(a.gpus, sys.argv[1:])).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.selftest_launch:
        return selftest_launch(a, world)
    # ONE JSON line on stdout, whatever the libraries print (RCCL writes a five-line version banner to stdout when its first
    # communicator comes up): file descriptor 1 is pointed at stderr for the whole run and the result line goes to the saved one
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    wl = WORKLOADS[a.config]
    PROMPT_LEN = wl["prompt"]
    if a.in_flight is None:
        a.in_flight = wl["in_flight"]
    if a.prefill_batches is None:
        a.prefill_batches = wl["prefill_batches"]
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # rehearsal on a one-GPU box: DUALHYP_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    rehearsal = os.environ.get("DUALHYP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(1337 + rank)      # the fine-tune's LoRA-dropout masks are keyed by torch's seed: the same line from run to run
    # under a launcher (WORLD_SIZE set) the process group is created even for ONE rank, so the collectives of the N > 1 path
    # (barrier, MAX all-reduce of the wall time, the fine-tune's flat-bucket all-reduce) run through RCCL on a one-GPU box too
    use_pg = world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if use_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from dualhyp_amd import GPT, Config, GER_LORA
    from dualhyp_amd.pipeline import BatchPipeline
    from dualhyp_amd.synth import synth_state_dict, synth_prompts

    if a.tune:
        from dualhyp_amd import _lib
        for kv in a.tune:
            k, v = kv.split("=")
            _lib.check(_lib.load().dh_set_tuning(int(k), int(v)))
    # inference: dropout plays no role; the fine-tune runs at the reference's LoRA dropout (finetune/ger.py:401: 0.05), masks drawn in the kernel
    cfg = Config.from_name(wl["model"], **{**GER_LORA, "dropout": 0.05 if wl.get("train") else 0.0})
    if "llama-3" in cfg.name.lower():
        cfg.block_size = 4096                      # inference/ger.py:189-190
    sd = synth_state_dict(cfg, seed=1337, device=dev, **SYNTH_KW)
    model = GPT(cfg).to(device=dev, dtype=torch.bfloat16)
    model.load_state_dict(sd, strict=True)
    del sd
    model.eval()
    if wl.get("train"):
        return bench_finetune(a, wl, cfg, model, dev, rank, world, rehearsal, use_pg)
    if wl["fp8"]:
        from dualhyp_amd import quantize_model_fp8
        quantize_model_fp8(model)                  # merge_lora_weights, then e4m3 rows + channel scales
        torch.cuda.empty_cache()
    B, G = a.batch, max(1, a.in_flight)
    from dualhyp_amd.generate import generate_batch
    gen_kw = dict(temperature=0.2, top_k=1, eos_id=None)
    # every rank gets its own utterances (strided shard of one synthetic corpus)
    n_warm = max(a.warmup, 1)
    n_batches = a.steps + n_warm
    corpus = synth_prompts(B * n_batches * world, PROMPT_LEN, cfg.padded_vocab_size, seed=1337, ragged=a.ragged,
                           lo=PROMPT_LEN * 3 // 4, hi=PROMPT_LEN * 5 // 4)
    mine = [p.to(dev) for p in corpus[rank::world]]
    max_len = max(p.numel() for p in mine)
    batches = [mine[i * B:(i + 1) * B] for i in range(n_batches)]

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    phase = {}     # HIP-event time of the prefill and decode phases of the merged schedule (timed region only)

    def run_merged(bs):
        """a "step" is ONE batch of B utterances; up to G consecutive steps share a decode loop"""
        outs = []
        for g in range(0, len(bs), G):
            flat = [p for b in bs[g:g + G] for p in b]
            o = generate_batch(model, flat, NEW_TOKENS, prefill_batch=B * a.prefill_batches, timing=phase, **gen_kw)
            outs += [o[i:i + B] for i in range(0, len(o), B)]
        return outs

    if a.schedule == "merged":
        # one allocation for the whole run (ragged prompts would otherwise grow the engine inside the timed region)
        model.set_capacity(B * min(G, a.steps), max_len + NEW_TOKENS, B * a.prefill_batches * max_len)
        # untimed: decode-graph capture at the in-flight size(s) the timed region uses
        run_merged([batches[i % n_warm] for i in range(min(G, a.steps))])
        if a.steps % G and a.steps > G:
            run_merged([batches[0]] * (a.steps % G))
        run_merged(batches[:n_warm])
        engs = [model.engine()]
    else:
        # `engines` engines, each with its own KV cache, workspace, decode graph, HIP stream and host thread,
        # sharing one copy of the weights; each takes groups of G batches (chunked prefill + joint decode), so
        # one engine's latency-bound decode loop overlaps the other's prefills
        pipe = BatchPipeline(model, a.engines, B * G, max_len + NEW_TOKENS, B * max_len)
        flat = lambda bs: [p for b in bs for p in b]
        pipe.warm(flat([batches[0]] * min(G, a.steps)), NEW_TOKENS, prefill_batch=B, **gen_kw)
        engs = [m.engine() for m in pipe.models]
    timed_prompts = [p for b in batches[n_warm:] for p in b]
    n_rep = a.repeats if a.repeats > 0 else (3 if a.steps <= 20 else 1)
    runs = []
    for _rep in range(n_rep):
        # one timed region: EXACTLY --steps steps between barrier + synchronize on both sides, MAX over ranks
        for e in engs:
            e.set_timing(True)          # (re)starts the HIP-event record of the kernel classes
        phase.clear()
        barrier()
        t0 = time.perf_counter()
        if a.schedule == "merged":
            outs = run_merged(batches[n_warm:])
        else:
            timed = batches[n_warm:]
            futs = [pipe.submit(flat(timed[g:g + G]), NEW_TOKENS, prefill_batch=B, **gen_kw) for g in range(0, len(timed), G)]
            outs = [o[i:i + B] for f in futs for o in [f.result()] for i in range(0, len(o), B)]
        barrier()
        dt = time.perf_counter() - t0
        gemm_ms = gemm_n = attn_ms = 0
        for e in engs:
            ms, n = e.get_timing(0)
            gemm_ms, gemm_n = gemm_ms + ms, gemm_n + n
            attn_ms += e.get_timing(2)[0]
            e.set_timing(False)
        assert all(o.numel() == p.numel() + NEW_TOKENS for o, p in zip([o for out in outs for o in out], timed_prompts))
        if use_pg:
            tt = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        runs.append(dict(dt=dt, gemm_ms=gemm_ms, gemm_n=gemm_n, attn_ms=attn_ms, phase=dict(phase), outs=outs))
    # the line describes the MEDIAN region (by wall time, the same one on every rank: dt is already the max over ranks)
    order = sorted(range(n_rep), key=lambda i: runs[i]["dt"])
    med = runs[order[(n_rep - 1) // 2]]
    dt, gemm_ms, gemm_n, attn_ms, outs = med["dt"], med["gemm_ms"], med["gemm_n"], med["attn_ms"], med["outs"]
    phase.clear()
    phase.update(med["phase"])

    result = None
    if rank == 0:
        utt = B * a.steps * world
        pruned = not any(kv.split("=")[0] == "23" and kv.split("=")[1] == "0" for kv in a.tune)
        flops = gemm_flops_per_prefill(cfg, sum(p.numel() for p in timed_prompts), B * a.steps, merged_lora=wl["fp8"],
                                       last_rows_only=pruned)
        achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_source = pmc_traffic(a.config)   # HBM-side bytes per launch from the committed PMC passes
        result = {
            "metric": wl["metric"],
            "value": utt / dt, "unit": "utterances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": wl["dtype"], "data": "synthetic",
            "repeats": {"timed_regions": n_rep, "reported": "median by wall time", "ms_per_step_all": [r["dt"] / a.steps * 1e3 for r in runs],
                        "value_all": [utt / r["dt"] for r in runs],
                        "spread_pct": (max(r["dt"] for r in runs) - min(r["dt"] for r in runs)) / dt * 100.0},
            "device": device_identity(local),
            "config": {"workload": wl["what"],
                       "batch_per_gpu": B, "prompt_tokens": f"uniform {PROMPT_LEN * 3 // 4}..{PROMPT_LEN * 5 // 4}" if a.ragged else PROMPT_LEN,
                       "new_tokens": NEW_TOKENS,
                       "parallelism": f"replicas x{world}", "batches_in_flight_per_gpu": G, "schedule": a.schedule,
                       "decode_rows_per_launch": B * min(G, a.steps) if a.schedule == "merged" else B,
                       "prefill_tokens_per_launch": B * PROMPT_LEN * (a.prefill_batches if a.schedule == "merged" else 1),
                       "last_block_rows": "last token of each sequence (proj / MLP of the last block feed only the last-position logits; "
                                          "K / V of every token are computed; results bit-equal to the all-rows form, which `--tune 23=0` runs: "
                                          "759 vs 772 utt/s on one box, DESIGN.md §5)" if pruned else "all",
                       **({"tuning": a.tune} if a.tune else {})},
            "roofline": {"bound": "mfma", "kernel": wl["kernel"],
                         "achieved": achieved, "peak": wl["peak"], "unit": "TFLOP/s",
                         "frac": achieved / wl["peak"], "traffic": traffic,
                         "traffic_source": traffic_source,
                         "launches": gemm_n, "avg_launch_ms": gemm_ms / max(gemm_n, 1),
                         "prefill_attention_ms_per_step": attn_ms / max(a.steps, 1)},
        }
    if rank == 0 and phase.get("decode_steps"):
        # second bound of the path: the decode step streams every weight once for all rows of the
        # launch plus each sequence's KV prefix; algorithmic bytes per step = 2.078 GB + rows x 22,528 B x S (S ~ 544)
        rows = phase["decode_row_steps"] / phase["decode_steps"]
        step_ms = phase["decode_ms"] / phase["decode_steps"]
        step_bytes = decode_bytes_per_step(cfg, rows, sum(p.numel() for p in timed_prompts) / len(timed_prompts) + NEW_TOKENS / 2,
                                           1 if wl["fp8"] else 2, merged_lora=wl["fp8"])
        result["phases"] = {"prefill_ms_per_step": phase["prefill_ms"] / a.steps, "decode_ms_per_step": phase["decode_ms"] / a.steps,
                            "decode_loop_ms_per_token": step_ms, "decode_rows_per_launch": rows}
        result["roofline_decode"] = {"bound": "hbm", "achieved": step_bytes / (step_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                     "frac": step_bytes / (step_ms * 1e-3) / 1e9 / 8000.0,
                                     "bytes_per_decode_step": step_bytes,
                                     "kernel": "decode step (fp8 weight-streaming GEMMs + split-KV attention)" if wl["fp8"]
                                     else "decode step (157 launches: streaming GEMMs + fused attention)"}
    if a.config != "tinyllama-bf16":
        a.no_overlap_probe = True
    if rank == 0 and world == 1 and a.schedule == "merged" and not a.no_overlap_probe and G > 1:
        # informational: the same engine with ONE batch in flight (prefill, decode 32 rows, next batch)
        reps = [batches[i % n_batches] for i in range(6)]
        generate_batch(model, reps[0], NEW_TOKENS, prefill_batch=B, **gen_kw)      # decode graph for 32 rows
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for b in reps:
            generate_batch(model, b, NEW_TOKENS, prefill_batch=B, **gen_kw)
        torch.cuda.synchronize()
        result["one_batch_in_flight"] = {"steps": len(reps), "value": B * len(reps) / (time.perf_counter() - t1), "unit": "utterances/s"}
    if rank == 0 and world == 1 and a.schedule == "merged" and not a.no_overlap_probe:
        # informational, outside the timed region of `value`: two engines on two HIP streams, each decoding G
        # batches jointly, so one engine's latency-bound decode loop runs under the other's prefills.  Higher
        # throughput, but the concurrent kernels time-slice the CUs and every per-kernel duration (hence a
        # roofline measured there) is inflated — which is why it is not the default schedule.
        pipe = BatchPipeline(model, 2, B * G, max_len + NEW_TOKENS, B * max_len)
        flat = lambda bs: [p for b in bs for p in b]
        pipe.warm(flat([batches[0]] * G), NEW_TOKENS, prefill_batch=B, **gen_kw)
        reps = [batches[i % n_batches] for i in range(2 * G)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in [pipe.submit(flat(reps[g:g + G]), NEW_TOKENS, prefill_batch=B, **gen_kw) for g in range(0, len(reps), G)]:
            f.result()
        torch.cuda.synchronize()
        result["overlap_probe"] = {"schedule": "threads, 2 engines x %d batches each" % G, "steps": len(reps),
                                   "value": B * len(reps) / (time.perf_counter() - t1), "unit": "utterances/s"}
        pipe.close()
    if rank == 0 and a.config != "tinyllama-bf16":
        # the CPU restatement of an 8B decoder on a 1536-token prompt takes minutes per utterance: outside the
        # "bounded sample" contract of the default run; parity of this path is asserted in tests/test_hip_fp8.py
        result["cpu_baseline"] = None
        # what CAN be checked in the run without the oracle: the first and the last utterance of the timed region decoded ALONE
        # give the tokens they got inside the timed joint run (kernel choice by phase, per-row summation order fixed)
        all_out = [o for out in outs for o in out]
        same = []
        for k in (0, len(timed_prompts) - 1):
            alone = generate_batch(model, [timed_prompts[k]], NEW_TOKENS, **gen_kw)[0]
            same.append(bool(torch.equal(alone.cpu(), all_out[k].cpu())))
        result["parity"] = {"oracle": None, "note": "oracle / reference parity of this configuration at its real prompt length is asserted in tests/: "
                            "test_hip_model.py::test_llama3_8b_shape_vs_reference[llama3_shape_1536] (bf16) and "
                            "test_hip_fp8.py::test_fp8_llama3_shape_vs_reference[llama3_shape_1536] (fp8) — tests/golden/llama3_shape_1536: the reference's own "
                            "logits, top-8 and greedy ids at T = 1536 + 16 decode steps, two layers of this shape — and the hs-128 attention kernels against the "
                            "oracle's SDPA up to 1700 keys (test_hip_ops.py, \"long\" cases); in the run: timed rows against the same prompts decoded alone",
                            "timed_rows_equal_alone_runs": same, "pass": all(same)}
    elif rank == 0 and world == 1 and not a.no_cpu_baseline:
        # the oracle decodes the first utterances of the timed region; its ids and logits are the checker for what the
        # timed run produced for those prompts (parity), its wall time is the CPU baseline
        result["cpu_baseline"], refs = cpu_baseline(cfg, timed_prompts[:9], NEW_TOKENS)
        result["parity"] = parity_vs_oracle(model, timed_prompts, [o for out in outs for o in out], refs, gen_kw)
    if rank == 0:
        if use_pg:
            result["config"]["process_group"] = dist.get_backend()
        emit(result)
    if use_pg:
        dist.destroy_process_group()


def bench_finetune(a, wl, cfg, model, dev, rank: int, world: int, rehearsal: bool, use_pg: bool = False) -> None:
    """BASELINE configs[2]: data-parallel LoRA fine-tune.  A step = ONE optimizer step over a global batch of 32
    utterances (finetune/ger.py:381: batch_size 32, micro_batch_size 1): every rank runs 32 / world micro-steps
    (GraphedTrainStep: forward + chunked CE + backward as one hipGraph replay, gradients accumulated into the flat fp32
    bucket), then ONE all-reduce of the bucket (RCCL over xGMI when world > 1), AdamW on the fp32 LoRA masters.  Total
    work per step is fixed, so `scaling` is "strong"."""
    import torch.distributed as dist
    from dualhyp_amd.finetune import FlatGradBucket
    from dualhyp_amd.synth import synth_prompts
    from dualhyp_amd.train import GraphedTrainStep, prepare_for_training
    T, GLOBAL = wl["prompt"], 32
    assert GLOBAL % world == 0, "--gpus must divide the global batch of 32"
    per_rank = GLOBAL // world
    model.train()
    params = prepare_for_training(model)
    bucket = FlatGradBucket(params)
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.02)
    step_fn = GraphedTrainStep(model, bucket)
    n_steps = a.steps + max(a.warmup, 1)
    corpus = synth_prompts(GLOBAL * 2, T, cfg.padded_vocab_size, seed=1337)
    # PACK micro-batches of the accumulation window run as one packed forward / backward (micro_batch_size 1 semantics
    # kept: per-sequence losses, summed gradients; tests/test_hip_train.py::test_packed_micro_steps_equal_the_sequential_ones)
    PACK = max(1, min(a.pack, per_rank))
    while per_rank % PACK:
        PACK -= 1
    seqs = torch.stack(corpus[rank::world]).to(dev)                       # [n, T]
    labs_all = seqs.clone()
    labs_all[:, :512] = -1
    n_groups = seqs.size(0) // PACK
    mine = [seqs[g * PACK:(g + 1) * PACK].contiguous() for g in range(n_groups)]
    labs = [labs_all[g * PACK:(g + 1) * PACK].contiguous() for g in range(n_groups)]

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def opt_step(k):
        losses = []
        for i in range(per_rank // PACK):
            j = (k * (per_rank // PACK) + i) % len(mine)
            losses.append(step_fn(mine[j], labs[j], 1.0 / per_rank, n_targets=PACK * (T - 512)))   # mean over this rank's micro-batches; all_reduce_mean then averages the ranks (as fit() does)
        bucket.all_reduce_mean()
        opt.step()
        bucket.zero()
        return torch.cat(losses)           # this rank's per-utterance losses of the step
    for k in range(max(a.warmup, 1)):
        opt_step(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        last = opt_step(k)
    barrier()
    dt = time.perf_counter() - t0
    mean_loss = last.mean().to(torch.float64)
    if use_pg:
        tt = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ml = mean_loss.cpu() if rehearsal else mean_loss     # the global batch's mean loss: the same 32 utterances whatever the world size
        dist.all_reduce(ml, op=dist.ReduceOp.SUM)
        mean_loss = ml / world
    if rank == 0:
        # fwd 2NT + dX 2NT over the frozen base (no dW), LoRA 6 x 4.5M x T, attention fwd+bwd
        d, I = cfg.n_embd, cfg.intermediate_size
        qkv = (cfg.n_head + 2 * cfg.n_query_groups) * cfg.head_size
        n_lin = cfg.n_layer * d * (qkv + d + 3 * I)
        n_lora = cfg.n_layer * (64 * d + 16 * (qkv + d))
        attn = 3.5 * cfg.n_layer * 4 * cfg.n_head * cfg.head_size * (T * (T + 1) / 2)
        # the lm_head (forward + dX) runs on the rows that carry a target, in whole 128-row tiles per packed launch
        # (train.GraphedTrainStep): EXECUTED FLOPs, not the T rows the reference pushes through it
        head_rows = max(128, -(-(PACK * (T - 512)) // 128) * 128) / PACK
        flop_utt = 4.0 * n_lin * T + 4.0 * cfg.padded_vocab_size * d * head_rows + 6.0 * n_lora * T + attn
        utt = GLOBAL * a.steps
        achieved = flop_utt * utt / dt / 1e12 / world          # per GPU
        emit({
            "metric": wl["metric"], "value": utt / dt, "unit": "utterances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": wl["dtype"],
            "data": "synthetic", "device": device_identity(dev.index),
            "config": {"workload": wl["what"], "global_batch": GLOBAL, "micro_batches_per_rank_per_step": per_rank, "tokens": T,
                       "micro_batches_per_packed_launch": PACK, "lora_dropout": cfg.dropout,
                       "parallelism": f"data-parallel x{world}, flat LoRA-gradient bucket of {bucket.flat.numel()} fp32 elements",
                       **({"process_group": dist.get_backend()} if use_pg else {})},
            "roofline": {"bound": "mfma", "kernel": wl["kernel"], "achieved": achieved, "peak": wl["peak"], "unit": "TFLOP/s",
                         "frac": achieved / wl["peak"], "traffic": None, "flop_per_utterance": flop_utt,
                         "note": "algorithmic FLOP of the whole micro-step / wall time per GPU: kernels inside a hipGraph replay cannot be bracketed by events"},
            "last_loss": float(last.mean().item()), "mean_loss_last_step": float(mean_loss.item()), "cpu_baseline": None})
    if use_pg:
        dist.destroy_process_group()


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


CLOSENESS_GATE = 1.25   # relRMS(hip, oracle bf16) / relRMS(oracle bf16, fp32) allowed on the pooled last-position logits (measured 1.08)
N_PARITY = 4      # utterances of the timed region the oracle's ids / prefill logits are compared on


def cpu_baseline(cfg, prompts, new_tokens: int):
    """The oracle (CPU restatement of ger/lora.py + generate/base.py, pinned to the reference by tests/golden) timed on
    this host in BASELINE.md §4's form: batch 1 as the reference runs it (inference/ger.py:60-81), ONE warm-up
    utterance, then 8 utterances of the same workload one after the other (~25 s of CPU work); utterances/s, prefill
    seconds and ms per generated token, thread and core counts, CPU model.  The ids and per-step logits the oracle
    produced for the first N_PARITY timed utterances are returned as the checker for the GPU run (`parity`), together
    with the oracle's fp32 last-position prefill logits for them (the yardstick of the logit gate)."""
    from dualhyp_amd.synth import synth_state_dict
    from oracle import ger_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 16))   # the box's CPU share for one GPU
    sd = synth_state_dict(cfg, seed=1337, device="cpu", **SYNTH_KW)
    m = O.OracleGPT(cfg, sd)
    kw = dict(temperature=0.2, top_k=1, eos_id=None, mode="argmax")
    n_timed = min(8, len(prompts) - 1)
    O.generate(m, prompts[-1].cpu(), prompts[-1].numel() + new_tokens, **kw)      # warm-up (an utterance outside the sample)
    m.reset_cache()
    tm, refs = {}, []
    t0 = time.perf_counter()
    for k in range(n_timed):
        T = prompts[k].numel()
        ids, trace = O.generate(m, prompts[k].cpu(), T + new_tokens, return_logits=True, timing=tm, **kw)
        assert ids.numel() == T + new_tokens
        m.reset_cache()
        if k < N_PARITY:
            refs.append([ids, trace])
    dt = (time.perf_counter() - t0) / n_timed
    # fp32 run of the same function on the parity utterances' prompts: relRMS(oracle bf16, oracle fp32) is the
    # distance a bf16 implementation of the reference has from the real-valued function (not part of the timing)
    m32 = O.OracleGPT(cfg, {k: v.float() for k, v in sd.items()})
    del sd
    for k, r in enumerate(refs):
        T = prompts[k].numel()
        with torch.inference_mode():
            r.append(m32(prompts[k].cpu().view(1, -1), torch.arange(T))[0, -1].clone())
        m32.reset_cache()
    del m32
    T0 = prompts[0].numel()
    base = {"value": 1.0 / dt, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
            "prefill_s": tm["prefill_s"] / n_timed, "decode_ms_per_token": tm["decode_s"] / max(tm["decode_tokens"], 1) * 1e3,
            "os_cpu_count": os.cpu_count(), "cpu_model": _cpu_model(),
            "reference_anchor": "the reference itself (ger.lora.GPT + generate.base.generate), 8 vCPU build container: 0.112 utterances/s, "
                                "prefill 2.97 s, 88.5 ms/token (BASELINE.md §2)",
            "sample": f"{n_timed} utterances one after the other after 1 warm-up utterance, {T0}-token prompt -> {new_tokens} generated "
                      f"tokens, batch 1 (as the reference runs), bf16, {torch.get_num_threads()} threads ({dt:.1f} s each)"}
    return base, refs


def parity_vs_oracle(model, prompts, hip_ids_timed, refs, gen_kw) -> dict:
    """Greedy ids and prefill logits of the HIP path against the oracle's for the first N_PARITY utterances of the timed
    region.  Ids: with the tied-head synthetic weights the oracle's top-2 margin is tens of bf16 ulps on
    every step (reported), so all generated ids must agree — this checks the token feedback loop, sampling and cache
    positions, NOT the attention numerics (the tied ids are a permutation chain of the last token).  The
    numerics are gated by the logits, which see every layer (the gate is spelled out where `logits_ok` is computed)."""
    from dualhyp_amd.generate import generate_batch
    G = refs[0][1].size(0)

    def margins_of(ref_logits):
        top = torch.topk(ref_logits.float(), 2, dim=-1).values
        return (top[:, 0] - top[:, 1]) / torch.exp2(torch.floor(torch.log2(top[:, 0].abs().clamp_min(1e-30))) - 7)

    def prefix(ids, ref_ids, T) -> int:
        ne = (ids.cpu()[T:T + G] != ref_ids[T:T + G]).nonzero().flatten().tolist()
        return ne[0] if ne else G

    per, ok_ids = [], True
    keep = model.cpu_rsqrt_vec_width
    hip_last, bf_last, f32_last = [], [], []
    for k, (ref_ids, ref_logits, ref_f32) in enumerate(refs):
        prompt, T = prompts[k], prompts[k].numel()
        mg = margins_of(ref_logits)
        unsafe = (mg < 4).nonzero().flatten().tolist()
        safe = unsafe[0] if unsafe else G
        row = {"utterance": k, "oracle_steps_with_margin_ge_4ulp": int((mg >= 4).sum()), "oracle_tie_free_prefix": safe,
               "min_margin_ulps": float(mg.min()), "ids_equal_prefix_timed_run": prefix(hip_ids_timed[k], ref_ids, T)}
        # the same prompt alone (product default rounding, then the CPU-rsqrt emulation the tests use against CPU tensors, Q11)
        for tag, width in (("alone", keep), ("alone_cpu_rsqrt_emulation", 32)):
            model.cpu_rsqrt_vec_width = width
            ids = generate_batch(model, [prompt], G, **gen_kw)[0]
            row[f"ids_equal_prefix_{tag}"] = prefix(ids, ref_ids, T)
            if tag == "alone":
                row["timed_row_equals_alone_run"] = bool(torch.equal(ids.cpu(), hip_ids_timed[k].cpu()))
        with torch.no_grad():
            model.reset_cache()
            lg = model(prompt.view(1, -1), torch.arange(T, device=prompt.device))[0, -1].float().cpu()
            model.reset_cache()
        model.cpu_rsqrt_vec_width = keep
        want, f32 = ref_logits[0].float(), ref_f32.float()
        rel = lambda a, b: float(((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item())
        row["prefill_last_logits_rel_rms"] = rel(lg, want)
        row["oracle_bf16_vs_fp32_rel_rms"] = rel(want, f32)
        row["hip_vs_oracle_fp32_rel_rms"] = rel(lg, f32)
        row["prefill_last_logits_bit_exact_frac"] = float((lg == want).float().mean().item())
        row["prefill_argmax_equal"] = bool(int(lg.argmax()) == int(want.argmax()))
        ok_ids = ok_ids and row["ids_equal_prefix_timed_run"] >= safe and row["ids_equal_prefix_alone_cpu_rsqrt_emulation"] >= safe \
            and row["timed_row_equals_alone_run"]
        hip_last.append(lg); bf_last.append(want); f32_last.append(f32)
        per.append(row)
    H, Bf, F = torch.stack(hip_last), torch.stack(bf_last), torch.stack(f32_last)
    rr = float(((H - Bf).pow(2).mean().sqrt() / Bf.pow(2).mean().sqrt()).item())
    yard = float(((Bf - F).pow(2).mean().sqrt() / F.pow(2).mean().sqrt()).item())
    acc = float(((H - F).pow(2).mean().sqrt() / F.pow(2).mean().sqrt()).item())
    e_hip, e_ref = float((H - F).abs().max().item()), float((Bf - F).abs().max().item())
    # The gate.  north_star's "within 1e-3" is not a property the reference's bf16 run has with respect to itself (one
    # bf16 ulp of a logit is 4e-3..1.6e-2).  What is required of a second bf16 implementation of the function f (= the
    # oracle's fp32 run): (i) it is as ACCURATE as the reference's own bf16 arithmetic — relRMS(HIP, f) <= 1.05 x
    # relRMS(oracle bf16, f) and max|HIP - f| <= 1.5 x max|oracle bf16 - f| — and (ii) it is no further from the
    # reference's bf16 run than an implementation with INDEPENDENT rounding errors of that size would be:
    # relRMS(HIP, oracle bf16) <= sqrt(2) x relRMS(oracle bf16, f).  (A ratio below 1 needs the two runs' rounding errors
    # to be correlated by more than 0.5 — true of the tests' short fixtures, where HIP reproduces the CPU kernels'
    # rounding points bit for bit; at T = 512 over 22 layers the online-softmax tiling of the two attention kernels
    # differs and the measured ratio is 0.93..1.18 by prompt, 1.08 pooled over the four.  The gate is CLOSENESS_GATE = 1.25 on the pooled
    # figure — set from that measurement (sqrt(2) allowed fully independent errors and hid that the HIP logits sit
    # further from the reference's bf16 run than that run sits from fp32); the tests' 1.0 x gate is the teacher-forced one
    # (tests/test_hip_model.py::gate).  Both numbers and their ratio are on the line.)
    logits_ok = acc <= 1.05 * yard and e_hip <= 1.5 * e_ref and rr <= CLOSENESS_GATE * yard
    out = {"utterances": f"first {len(refs)} utterances of the timed region", "generated_tokens": G,
           "ids_equal_prefix_timed_run": [r["ids_equal_prefix_timed_run"] for r in per],
           "oracle_tie_free_prefix": [r["oracle_tie_free_prefix"] for r in per],
           "ids_note": "tied-head synthetic weights: the ids are a permutation chain of the last token (feedback loop, sampling and "
                       "cache positions); numerics are gated by the logits below and by tests/ (untied 22-layer fixture)",
           "prefill_last_logits_rel_rms": rr, "oracle_bf16_vs_fp32_rel_rms": yard, "hip_vs_oracle_fp32_rel_rms": acc,
           "rel_rms_hip_vs_oracle_bf16_over_yardstick": rr / yard,
           "prefill_last_logits_max_abs_vs_fp32": e_hip, "oracle_bf16_max_abs_vs_fp32": e_ref,
           "logits_gate": "relRMS(hip, fp32) <= 1.05 x relRMS(oracle bf16, fp32) and max|hip - fp32| <= 1.5 x max|oracle bf16 - fp32| and "
                          f"relRMS(hip, oracle bf16) <= {CLOSENESS_GATE} x relRMS(oracle bf16, fp32), last-position prefill logits of the utterances pooled "
                          "(free-running prompts of the timed region; the tests' 1.0 x gate is the teacher-forced one on the reference's own fixtures)",
           "ids_pass": bool(ok_ids), "logits_pass": bool(logits_ok), "per_utterance": per}
    out["pass"] = bool(out["ids_pass"] and out["logits_pass"])
    return out


if __name__ == "__main__":
    main()
