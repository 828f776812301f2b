"""CPU tests of the host-side logic either side of the decoder: prompt packing, labels, collate,
reliability tokens, WER, utterance sharding + counter all-reduce over gloo (world_size 2)."""
import json
import os
import socket
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

from dualhyp_amd import data as D
from dualhyp_amd.inference import extract_answer, run_inference, shard_indices
from dualhyp_amd.wer import edit_counts, post_normalize, wer, wer_counts

GOLDEN = Path(__file__).parent / "golden"


class CharTok:
    """Stand-in tokenizer (no HF tokenizer offline): BOS + one id per character."""
    eos_token = "\x03"

    def encode(self, s):
        return [1] + [ord(c) + 3 for c in s]

    def decode(self, ids):
        return "".join(chr(int(i) - 3) for i in ids if int(i) > 3)


def sample(uid="u1", snr=-5):
    return {"Uid": uid, "Caption": "the cat sat", "Dataset": "LRS2",
            "nhyps_asr": {"hyps": ["the cat sat", "the cat sad", "a cat sat", "the bat sat", "the cat set"], "scores": [0] * 5},
            "nhyps_vsr": {"hyps": ["the cap sat", "the cat sat", "the gap sat", "the cab sat", "he cat sat"], "scores": [0] * 5},
            "Audio_Corruption": {"total_len": 32000, "start_fr": 8000, "occ_len": 2000, "snr": snr, "noise_name": "babble"},
            "Visual_Corruption": {"total_len": 50, "start_fr": 20, "occ_len": 12}}


def test_templates_equal_the_reference():
    ref = json.loads((GOLDEN / "prompts.json").read_text())["prompts"]   # dumped from data/prompts.py by make_golden.py
    for name in ("GER", "DualHyp", "RelPrompt"):
        assert D.get_prompts_format(name) == ref[name]
    with pytest.raises(ValueError):
        D.get_prompts_format("nope")


def test_prompt_strings():
    s = sample()
    g = D.ger_prompt(s)
    assert g.endswith("### Best-hypothesis:\nthe cat sat\n\n### Other-hypothesis:\nthe cat sad\na cat sat\nthe bat sat\nthe cat set\n\n### Response:\n")
    d = D.dualhyp_prompt(s, s)
    assert "### ASR Best-hypothesis:\nthe cat sat\n\n### VSR Best-hypothesis:\nthe cap sat\n\n### ASR Other-hypotheses:\nthe cat sad\n" in d
    assert d.endswith("### VSR Other-hypotheses:\nthe cat sat\nthe gap sat\nthe cab sat\nhe cat sat\n\n### Response:\n")
    assert D.dualhyp_prompt(s, s, max_nhyps=3).count("\n") < d.count("\n")
    # reliability tokens: 0.4 s chunks = 6400 samples / 10 frames (data/av_dataset.py:444-445)
    am = D.noise_mask(s, "audio")
    assert am.count("N") == 2000 and am[8000] == "N" and am[7999] == "C"
    scores, labels = D.chunk_reliability(am, 6400)
    assert labels == ["<<C>>", "<<M>>", "<<C>>", "<<C>>", "<<C>>"] and abs(scores[1] - 4400 / 6400) < 1e-9
    assert D.noise_mask(sample(snr=10), "audio", mask_threshold=0).count("N") == 0          # snr above threshold: clean
    _, vl = D.chunk_reliability(D.noise_mask(s, "video"), 10)
    assert vl == ["<<C>>", "<<C>>", "<<N>>", "<<M>>", "<<C>>"]
    r = D.relprompt_prompt(s, s, labels, vl)
    assert "### Audio Mask:\n<<C>><<M>><<C>><<C>><<C>>\n\n\n### VSR Best-hypothesis:\nthe cap sat" in r and r.endswith("\n\n\n### Response:\n")


def test_encode_collate_and_dataset():
    tok = CharTok()
    s = sample()
    p = D.dualhyp_prompt(s, s)
    ex = D.encode_example(tok, p, s["Caption"])
    n_prompt = ex["input_ids_no_response"].numel()
    assert (ex["labels"][:n_prompt] == -1).all() and torch.equal(ex["labels"][n_prompt:], ex["input_ids"][n_prompt:])
    assert ex["input_ids"].numel() == n_prompt + len(s["Caption"]) + 1 and ex["input"].endswith(tok.eos_token)
    short = D.encode_example(tok, p, s["Caption"], max_input_length=50)
    assert short["input_ids"].numel() == 50 and short["labels"].numel() == 50
    ex2 = D.encode_example(tok, D.ger_prompt(s), s["Caption"])
    b = D.collate([ex, ex2])
    n = max(ex["input_ids"].numel(), ex2["input_ids"].numel())
    assert b["input_ids"].shape == (2, n) and (b["labels"][1, ex2["labels"].numel():] == -1).all()
    assert (b["input_ids"][1, ex2["input_ids"].numel():] == 0).all()
    ds = D.HypothesesDataset([sample("a"), sample("a", snr=3), sample("b")], tok, "RelPrompt", seed=0)
    assert len(ds) == 2 and ds[0]["uid"] == "a" and "<<" in ds[1]["input_no_response"]


def test_wer_known_answers():
    assert edit_counts("a b c".split(), "a b c".split()) == (0, 0, 0)
    assert edit_counts("a b c".split(), "a x c".split()) == (1, 0, 0)
    assert edit_counts("a b c".split(), "a c".split()) == (0, 1, 0)
    assert edit_counts("a c".split(), "a b c".split()) == (0, 0, 1)
    assert sum(edit_counts("the cat sat on the mat".split(), "cat sat on a mat today".split())) == 3
    # corpus level: total errors / total reference words, NOT the mean of per-utterance rates
    assert abs(wer(["a b c d", "x"], ["a b c d", "y z"]) - 2 / 6) < 1e-12
    assert wer_counts(["hello  world "], ["hello world"])["errors"] == 0          # whitespace collapsed
    assert post_normalize("It's A-OK, right?") == "its aok right"
    assert extract_answer("PROMPT the cat sat\nextra line", "PROMPT ") == "the cat sat"


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    exs, gen, dec = _toy_corpus()
    out = run_inference(gen, exs, dec, batch_size=3, rank=rank, world=world)
    q.put((rank, {k: v for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _toy_corpus():
    tok = CharTok()
    refs = ["the cat sat", "a dog ran", "birds fly high", "no way", "yes sir", "so it goes", "ok"]
    hyps = ["the cat sat", "a dog run", "birds fly", "no way", "yes sir yes", "so it goes", "okay"]
    exs = [{"input_ids_no_response": torch.tensor(tok.encode(f"P{i}: ")), "ground_truth": r} for i, r in enumerate(refs)]

    def gen(prompts):   # deterministic stub "model": appends the canned hypothesis of that prompt
        outs = []
        for p in prompts:
            i = int(tok.decode(p)[1])
            outs.append(torch.cat([p, torch.tensor(tok.encode(hyps[i] + "\nignored"))[1:]]))
        return outs
    return exs, gen, tok.decode


def test_sharded_inference_matches_single_process():
    assert shard_indices(7, 0, 2) == [0, 2, 4, 6] and shard_indices(7, 1, 2) == [1, 3, 5]
    exs, gen, dec = _toy_corpus()
    single = run_inference(gen, exs, dec, batch_size=4)
    assert single["n"] == 7 and abs(single["WER"] - 4 / 17) < 1e-12 and abs(single["gtms"] - 3 / 7) < 1e-12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        for k in ("WER", "gtms", "post_ST_wer", "post_gtms", "n"):
            assert got[r][k] == single[k], (r, k)
    assert got[0]["predictions"] == single["predictions"] and got[1]["predictions"] is None


def test_lr_schedule_and_accumulation_trace():
    from dualhyp_amd.finetune import lr_at, step_schedule, epoch_order
    from oracle import ger_oracle as O
    for it in (0, 5, 10, 11, 50, 99, 100, 250):
        for cos in (False, True):
            assert lr_at(it, 1e-4, 10, 100, cos, 0.1) == pytest.approx(O.lr_at(it, 1e-4, 10, 100, cos, 0.1))
    assert lr_at(10, 1e-4, 10) == 1e-4 and lr_at(5, 1e-4, 10) == 5e-5 and lr_at(11, 1e-4, 10) == 1e-4
    # quirk Q3: with accum 32 the reference steps after 31 micro-batches (indices 30, 61, 92, ...)
    assert step_schedule(100, 32, reference_accumulation=True) == [30, 61, 92]
    assert step_schedule(100, 32) == [31, 63, 95]
    # sharded epoch order: ranks partition one permutation, equal sizes, different per epoch
    a, b = epoch_order(11, 0, 0, 2), epoch_order(11, 0, 1, 2)
    assert len(a) == len(b) == 5 and not set(a) & set(b) and epoch_order(11, 1, 0, 2) != a


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from dualhyp_amd.finetune import FlatGradBucket
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))]
    bkt = FlatGradBucket(ps)
    assert bkt.flat.numel() == 17 and ps[0].grad.data_ptr() == bkt.flat.data_ptr()
    # "backward" accumulates in place into the bucket views
    ps[0].grad += (rank + 1)
    ps[1].grad += 10 * (rank + 1)
    bkt.all_reduce_mean()
    q.put((rank, ps[0].grad.clone(), ps[1].grad.clone()))
    bkt.zero()
    assert float(bkt.flat.abs().sum()) == 0 and ps[1].grad.data_ptr() == bkt.flat[12:].data_ptr()
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, g0, g1 in got:
        assert torch.all(g0 == 1.5) and torch.all(g1 == 15.0)       # mean over the two ranks


def test_byte_tokenizer_and_reliability_tokens():
    from dualhyp_amd.tokenizer import ByteTokenizer, load_tokenizer
    tok = ByteTokenizer()
    ids = tok.encode("héllo\n</s>")
    assert ids[0] == tok.bos_token_id and ids[-1] == tok.eos_token_id and tok.decode(ids) == "héllo\n"
    tok.add_reliability_tokens(320)
    ids = tok.encode("a<<C>><<N>><<M>>b")
    assert ids == [1, ord("a") + 3, 320, 322, 321, ord("b") + 3] and tok.decode(ids) == "ab"
    # the reference's answer extraction works on the decoded text of prompt and prompt+continuation
    p = tok.encode("### Response:\n")
    full = p + tok.encode("the cat sat\nmore")[1:]
    assert extract_answer(tok.decode(full), tok.decode(p)) == "the cat sat"
    assert isinstance(load_tokenizer("/nonexistent-dir", "byte"), ByteTokenizer)
    with pytest.raises(FileNotFoundError):
        load_tokenizer("/nonexistent-dir", "hf")


def test_harness_flags_are_the_references():
    """python -m dualhyp_amd.inference / .finetune accept every flag of inference/ger.py:129-153 and
    finetune/ger.py:373-407 (parsed here without touching the GPU: argparse runs before anything else)."""
    import argparse
    import dualhyp_amd.inference as inf
    p = argparse.ArgumentParser()
    inf.add_lora_arguments(p)
    a = p.parse_args(["--lora_r", "8", "--lora_alpha", "32", "--lora_dropout", "0.1"])
    assert (a.lora_r, a.lora_alpha, a.lora_dropout, a.lora_query, a.lora_mlp) == (8, 32, 0.1, True, False)
    a.llm_checkpoint, a.config_name = "checkpoints/TinyLlama/tiny-llama-1.1b-chat", None
    cfg = inf.config_from_args(a)
    assert cfg.name == "tiny-llama-1.1b-chat" and cfg.r == 8 and cfg.alpha == 32 and cfg.to_projection and not cfg.to_mlp
    a.llm_checkpoint = "checkpoints/meta-llama/Llama-3-8B"
    assert inf.config_from_args(a).block_size == 4096          # inference/ger.py:189-190


def test_bench_flop_accounting_counts_executed_launches_only():
    """bench.py's numerator for the prefill-GEMM roofline: with the last block on the last rows (csrc/engine.hip
    g_prune_last_layer) the last layer contributes its QKV GEMM (+ the q/k/v LoRA side products) alone — neither the rows it
    skips nor the n_seq-row launches outside the timed class."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from dualhyp_amd import Config, GER_LORA
    cfg = Config.from_name("tiny-llama-1.1b-chat", **{**GER_LORA, "dropout": 0.0})
    d, I, L = cfg.n_embd, cfg.intermediate_size, cfg.n_layer
    qkv = (cfg.n_head + 2 * cfg.n_query_groups) * cfg.head_size
    n_tok, n_seq = 32 * 512, 32
    full = bench.gemm_flops_per_prefill(cfg, n_tok, n_seq)
    assert full == L * n_tok * (2 * d * (qkv + d + 3 * I) + 2 * d * 64 + 2 * 16 * (qkv + d))
    pruned = bench.gemm_flops_per_prefill(cfg, n_tok, n_seq, last_rows_only=True)
    last_qkv_only = n_tok * (2 * d * qkv + 2 * d * 48 + 2 * 16 * qkv)
    assert pruned == (L - 1) * n_tok * (2 * d * (qkv + d + 3 * I) + 2 * d * 64 + 2 * 16 * (qkv + d)) + last_qkv_only
    merged = bench.gemm_flops_per_prefill(cfg, n_tok, n_seq, merged_lora=True, last_rows_only=True)
    assert merged == (L - 1) * n_tok * 2 * d * (qkv + d + 3 * I) + n_tok * 2 * d * qkv


# ---- data-parallel fit(): sharded validation, one best_model.pth, barriers ------------------------------------
class _StubLM(torch.nn.Module):
    """CPU stand-in with the surface fit() / validate() use (the HIP decoder cannot run here): `transformer.h[l].attn.{attn,proj}`
    carrying `r`, `lora_A`, `lora_B`; forward(idx, lm_head_chunk_size) -> logits or the list of chunks; engine hooks as no-ops."""

    def __init__(self, vocab=64, d=8, r=2):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        rnd = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g) * 0.3)
        self.wte, self.head = rnd(vocab, d), rnd(vocab, d)

        class Lin(torch.nn.Module):
            def __init__(s):
                super().__init__()
                s.r, s.lora_A, s.lora_B = r, rnd(r, d), rnd(d, r)
        blk = torch.nn.Module()
        blk.attn = torch.nn.Module()
        blk.attn.attn, blk.attn.proj = Lin(), Lin()
        self.transformer = torch.nn.Module()
        self.transformer.h = torch.nn.ModuleList([blk])

    def forward(self, idx, lm_head_chunk_size=0):
        x = self.wte[idx]
        for m in (self.transformer.h[0].attn.attn, self.transformer.h[0].attn.proj):
            x = x + (x @ m.lora_A.t()) @ m.lora_B.t()
        lg = x @ self.head.t()
        return list(lg.split(lm_head_chunk_size, dim=1)) if lm_head_chunk_size > 0 else lg

    def reset_cache(self): pass
    def refresh_engine(self): pass
    def _drop_engine(self): pass


def _fit_examples(n, vocab=64):
    g = torch.Generator().manual_seed(17)
    exs = []
    for i in range(n):
        T = 9 + i % 5
        ids = torch.randint(3, vocab, (T,), generator=g)
        lab = ids.clone()
        lab[: T - 4] = -1
        if i == 3:
            lab[:] = -1                       # an all-masked utterance: skipped by validate (finetune/ger.py:341-344)
        exs.append({"input_ids": ids, "labels": lab})
    return exs


def _fit_collate(exs):
    T = max(e["input_ids"].numel() for e in exs)
    pad = lambda t, v: torch.cat([t, torch.full((T - t.numel(),), v, dtype=t.dtype)])
    return {"input_ids": torch.stack([pad(e["input_ids"], 0) for e in exs]), "labels": torch.stack([pad(e["labels"], -1) for e in exs])}


def _fit_run(rank, world, out_dir, log):
    from dualhyp_amd.finetune import TrainConfig, fit, validate
    m = _StubLM()
    train, val = _fit_examples(16), _fit_examples(7)
    val_batches = lambda: (_fit_collate([e]) for e in val)
    v0 = validate(m, val_batches(), rank, world)
    tc = TrainConfig(learning_rate=5e-2, num_epochs=2, batch_size=4, micro_batch_size=1, lm_head_chunk_size=4, save_interval=4 // world, shuffle=False)
    stats = fit(m, train, _fit_collate, tc, val_batches=val_batches, out_dir=out_dir, rank=rank, world=world, device="cpu", log=log)
    return v0, stats, {k: v.detach().clone() for k, v in m.state_dict().items() if "lora_" in k}


def _fit_worker(rank, world, port, q, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lines = []
    v0, stats, lora = _fit_run(rank, world, out_dir, lines.append)
    q.put((rank, v0, stats, {k: v.tolist() for k, v in lora.items()}, lines))     # plain lists: no shared-memory handles outliving the rank
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_fit_validates_sharded_and_writes_one_checkpoint(tmp_path):
    """fit() on two gloo ranks (CPU stand-in model: host logic only): validate() runs each rank's share of the batches and
    all-reduces (loss sum, count) once, so both ranks see the SAME validation loss — the single-process one — and take the same
    best_val decisions; rank 0 alone writes best_model.pth; every rank leaves with the weights of the 1-rank run."""
    from dualhyp_amd.checkpoint import load_checkpoint
    v_single, stats_single, lora_single = _fit_run(0, 1, str(tmp_path / "single"), lambda s: None)
    assert stats_single["checkpoints_written"] >= 1 and (tmp_path / "single" / "best_model.pth").is_file()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = tmp_path / "dp"
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q, str(out))) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=180) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (v0a, sa, la, _), (v0b, sb, lb, lines_b) = got[0], got[1]
    # sharded validation == the whole set on one process (6 counted batches: one of the 7 is all-masked)
    assert v0a == v0b == pytest.approx(v_single, rel=1e-12)
    assert sa["best_val_loss"] == sb["best_val_loss"] == pytest.approx(stats_single["best_val_loss"], rel=1e-5)
    assert sa["optimizer_steps"] == sb["optimizer_steps"] == stats_single["optimizer_steps"]
    # one writer: rank 0 counted its saves, rank 1 wrote nothing, the file holds rank 0's (= everybody's) LoRA tensors
    assert sa["checkpoints_written"] >= 1 and sb["checkpoints_written"] == 0
    assert sorted(p.name for p in out.iterdir()) == ["best_model.pth", "lit_model_lora_finetuned.pth"]
    final = load_checkpoint(out / "lit_model_lora_finetuned.pth")
    la, lb = ({k: torch.tensor(v) for k, v in d.items()} for d in (la, lb))
    for k in la:
        assert torch.equal(la[k], lb[k]), k                                   # replicas stay replicas
        assert torch.allclose(la[k], lora_single[k], rtol=1e-5, atol=1e-6), k   # global batch 4 = 2 ranks x 2 = 1 rank x 4
        assert torch.equal(final[k], la[k]), k
    assert lines_b == [] or all("val loss" in l for l in lines_b)


def test_feature_files_are_keyed_by_the_corruption_variant():
    """RelPrompt encoder features were looked up by Uid alone while the mask targets come from the CHOSEN variants'
    corruption records.  One item per Uid keeps <Uid>.pt; several get one file per corruption record."""
    a, b = sample("u1", snr=-5), sample("u1", snr=5)
    b["Audio_Corruption"] = dict(b["Audio_Corruption"], start_fr=100)
    assert D.feature_key(a, "Audio_Corruption", 1) == "u1"
    ka, kb = D.feature_key(a, "Audio_Corruption", 2), D.feature_key(b, "Audio_Corruption", 2)
    assert ka != kb and ka.startswith("u1.") and kb.startswith("u1.") and ka == D.feature_key(dict(a), "Audio_Corruption", 2)
    assert D.feature_key(a, "Visual_Corruption", 2) == D.feature_key(b, "Visual_Corruption", 2)    # same visual record -> same file


def test_attention_backward_plan_is_the_inverse_map():
    """ops.attn_bwd_plan (host logic of the fine-tune's attention backward, built once per micro-step): every sequence starts at a
    multiple of 32 in the padded order, pad_tok maps each padded position back to its token (or -1 for padding), and the map is a
    bijection between tokens and non-padding positions — checked on the CPU, no kernel involved."""
    import torch
    from dualhyp_amd import ops
    lens = [70, 33, 1, 128, 31]
    n_tok = sum(lens)
    i32 = torch.int32
    starts = torch.tensor([sum(lens[:i]) for i in range(len(lens))], dtype=i32)
    plan = ops.attn_bwd_plan(starts, torch.tensor(lens, dtype=i32), n_tok, lens)
    assert plan["n_pad"] == sum(-(-n // 32) * 32 for n in lens) and plan["n_seq"] == len(lens)
    ps, pt = plan["pad_start"].tolist(), plan["pad_tok"].tolist()
    assert all(p % 32 == 0 for p in ps) and ps == [0, 96, 160, 192, 320]
    assert sorted(t for t in pt if t >= 0) == list(range(n_tok))
    for i, n in enumerate(lens):
        assert pt[ps[i]:ps[i] + n] == list(range(int(starts[i]), int(starts[i]) + n))      # tokens in order, then padding
        end = ps[i + 1] if i + 1 < len(lens) else plan["n_pad"]
        assert all(t == -1 for t in pt[ps[i] + n:end])
