"""End-to-end harness runs on the GPU (SURVEY.md §8 a17 / f2 / e):

* `python -m dualhyp_amd.inference` — the counterpart of inference/ger.py:127-221 with its flag names — on a 7-item JSON
  in the merged wire format (data/merge_json.py:5-63): args -> Config.from_name -> checkpoint -> tokenizer -> JSON
  dataset -> prompt packer -> generate_batch on the HIP model -> predictions JSON + WER fields, checked against
  `run_inference` fed the ORACLE's ids for the same prompts;
* `python -m dualhyp_amd.finetune` — finetune/ger.py:371-435 — a short run writing the reference's checkpoint files;
* data-parallel equivalence without an 8-GPU node: two rank processes on cuda:0 over gloo (the RCCL path itself has not
  run on hardware yet): fit(world=2, accumulation 2) == fit(world=1, accumulation 4), sharded inference == single process."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = str(ROOT) + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra)
    return env


def merged_items(n=7, caps=None):
    """Items with every key of the merged schema (data/merge_json.py:5-63; README.md:78-89)."""
    caps = caps or ["the cat sat on the mat", "we went home early", "it is raining again", "open the window please", "she sells sea shells",
                    "nothing to see here", "turn left at the light"]
    items = []
    for i in range(n):
        c = caps[i % len(caps)]
        w = c.split()
        asr = [c, " ".join(w[:-1]), c.replace("the", "a"), " ".join(w[1:]), c + " now"]
        vsr = [" ".join(reversed(w)), c, c.replace("e", "a"), w[0] + " " + c, " ".join(w[:2])]
        items.append({"Dataset": "LRS2", "Uid": f"utt{i:03d}", "Caption": c, "Clean_Wav": f"a/{i}.wav", "Noise_Wav": f"n/{i}.wav",
                      "Noise_Category": {"asr": "babble", "vsr": "occlusion"}, "SNR": -5 + i, "Mouthroi": f"m/{i}.npz",
                      "Video": f"v/{i}.mp4", "Face_landmark": f"l/{i}.pkl",
                      "nhyps_asr": {"hyps": asr, "scores": [-0.1 * k for k in range(5)]},
                      "nhyps_vsr": {"hyps": vsr, "scores": [-0.2 * k for k in range(5)]},
                      "Audio_Corruption": {"total_len": 32000, "start_fr": 6400, "occ_len": 6400, "snr": -5, "noise_name": "babble"},
                      "Visual_Corruption": {"total_len": 50, "start_fr": 10, "occ_len": 12},
                      "WER_1st-hyp": {"asr": 0.0, "vsr": 1.0}})
    return items


NEW = 10
# The harness fixture's decoder SAYS something (random weights gave WER 1.0 == 1.0 on both sides, an equality
# that proves nothing).  dualhyp_amd.synth ties the head to the embedding through a successor permutation; here the permutation
# is chosen so that the cycle through "\n" — the last token of every prompt, "### Response:\n" — spells SAYS: every utterance is
# answered "up down", and the captions are picked so the corpus WER is a known fraction strictly between 0 and 1.
SAYS = "up down"
CAPTIONS = ["up down", "up town", "sit down", "up down now", "up", "we went up down there", "down up"]
# word errors against "up down": 0 + 1 (sub) + 1 (sub) + 1 (del) + 1 (ins) + 3 (del) + 2 = 9 over 2 + 2 + 2 + 3 + 1 + 5 + 2 = 17 reference words
WANT_WER, WANT_EXACT = 9 / 17, 1


def speaking_state_dict(cfg, seed):
    """synth_state_dict(weight_scale 4, embed_scale 64) with the head tied through a permutation whose cycle through the
    byte-tokenizer's "\n" reads SAYS: lm_head[v] = random[v] + wte_unscaled[pred(v)], pred(first) = "\n", pred(next) = previous, ...,
    pred("\n") = last; every other token keeps a successor too (one more cycle), so the map stays a permutation."""
    import math
    from dualhyp_amd.synth import synth_state_dict, uniform, stream_id
    sd = synth_state_dict(cfg, seed=seed, weight_scale=4.0, embed_scale=64.0, head_tie=0.0)
    V, d = cfg.padded_vocab_size, cfg.n_embd
    cyc = [ord("\n") + 3] + [b + 3 for b in SAYS.encode()]
    assert len(set(cyc)) == len(cyc), "a permutation cycle visits a token once: SAYS must not repeat a character"
    rest = [v for v in range(V) if v not in cyc]
    pred = torch.empty(V, dtype=torch.int64)
    for chain in (cyc, rest):
        for i, v in enumerate(chain):
            pred[v] = chain[i - 1]
    a = 0.02 * math.sqrt(3.0) * 4.0
    raw = uniform((V, d), a, stream_id(seed, "transformer.wte.weight"), "cpu", torch.float32)     # the embedding before embed_scale
    sd["lm_head.linear.weight"] = (sd["lm_head.linear.weight"].float() + raw[pred]).to(torch.bfloat16)
    return sd


@pytest.mark.parametrize("fmt", ["DualHyp", "GER"])
def test_inference_harness_end_to_end(tmp_path, fmt):
    """`python -m dualhyp_amd.inference` on a merged-schema JSON, in the DualHyp format (data/av_dataset.py:373-429) and in the
    ASR-only GER format of BASELINE configs[0] (data/av_dataset.py:210-256, data/prompts.py:3-7; VERDICT r03 missing #6), against
    `run_inference` fed the ORACLE's ids for the same prompts — with a decoder whose corpus WER is a known value in (0, 1)."""
    from dualhyp_amd import Config
    from dualhyp_amd.data import HypothesesDataset
    from dualhyp_amd.inference import run_inference
    from dualhyp_amd.tokenizer import ByteTokenizer
    from dualhyp_amd.wer import wer_counts, post_normalize
    from oracle import ger_oracle as O
    items = merged_items(caps=CAPTIONS)
    test_json = tmp_path / "test.json"
    test_json.write_text(json.dumps(items))
    ckpt_dir = tmp_path / "checkpoints" / "parity-harness"
    ckpt_dir.mkdir(parents=True)
    lora = dict(r=16, alpha=16, dropout=0.05, to_query=True, to_key=True, to_value=True, to_projection=True)
    cfg = Config.from_name("parity-harness", **lora)
    sd = speaking_state_dict(cfg, seed=31)
    run_dir = tmp_path / "runs" / "exp"
    run_dir.mkdir(parents=True)
    torch.save({"model": sd}, run_dir / "best_model.pth")                      # finetune/ger.py:356-358 format
    cmd = [sys.executable, "-m", "dualhyp_amd.inference", "--test_path", str(test_json), "--model_path", str(run_dir / "best_model.pth"),
           "--llm_checkpoint", str(ckpt_dir), "--prompts_format", fmt, "--tokenizer", "byte",
           "--max_new_tokens", str(NEW), "--decode_batch", "4"] + (["--dual_hypotheses"] if fmt == "DualHyp" else [])
    out = subprocess.run(cmd, cwd=tmp_path, env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    pred_file = run_dir / "predictions" / "best_model.json"                    # inference/ger.py:120-122
    js = json.loads(pred_file.read_text())
    assert len(js) == 7 + 2 and set(js[0]) == {"inference", "ground_truth"} and set(js[-2]) == {"wer", "gtms"} and set(js[-1]) == {"post_wer", "post_gtms"}
    preds = js[:7]
    assert [p["ground_truth"] for p in preds] == [it["Caption"] for it in items]
    c = wer_counts([p["inference"] for p in preds], [p["ground_truth"] for p in preds])
    assert js[-2]["wer"] == pytest.approx(c["errors"] / c["ref_words"]) and js[-2]["gtms"] == f"{c['exact']}/7"
    cp = wer_counts([post_normalize(p["inference"]) for p in preds], [post_normalize(p["ground_truth"]) for p in preds])
    assert js[-1]["post_wer"] == pytest.approx(cp["errors"] / cp["ref_words"])
    # ---- the same pipeline with the ORACLE as the decoder (CPU, batch 1 as the reference): ids and margins per utterance
    tok = ByteTokenizer()
    ds = HypothesesDataset(str(test_json), tok, prompts_format=fmt, seed=1337)
    exs = [ds[i] for i in range(len(ds))]
    assert all(e["input_no_response"].endswith("### Response:\n") for e in exs)
    if fmt == "DualHyp":
        assert all("### VSR Other-hypotheses:\n" in e["input_no_response"] for e in exs)
    else:       # the ASR-only prompt: best hypothesis + the other ASR hypotheses, nothing of the VSR stream (data/prompts.py:3-7)
        assert all("VSR" not in e["input_no_response"] and it["nhyps_asr"]["hyps"][4] in e["input_no_response"] for e, it in zip(exs, items))
    om = O.OracleGPT(cfg, sd)
    safe_all, table = [], {}
    for e in exs:
        p = e["input_ids_no_response"]
        ids, trace = O.generate(om, p, p.numel() + NEW, temperature=0.2, top_k=1, eos_id=tok.eos_token_id, mode="argmax", return_logits=True)
        om.reset_cache()
        top = torch.topk(trace.float(), 2, dim=-1).values
        margins = (top[:, 0] - top[:, 1]) / torch.exp2(torch.floor(torch.log2(top[:, 0].abs().clamp_min(1e-30))) - 7)
        safe_all.append(bool((margins >= 4).all()))
        table[tuple(p.tolist())] = ids
    ref = run_inference(lambda ps: [table[tuple(p.tolist())] for p in ps], exs, tok.decode, batch_size=4)
    n_safe = sum(safe_all)
    from conftest import record_parity
    record_parity(f"harness.inference_vs_oracle.{fmt}", utterances=7, utterances_all_steps_margin_ge4=n_safe,
                  predictions_equal=sum(a["inference"] == b["inference"] for a, b in zip(preds, ref["predictions"])),
                  predictions_as_designed=sum(p["inference"] == SAYS for p in preds),
                  wer_hip=js[-2]["wer"], wer_oracle=ref["WER"], wer_by_hand=WANT_WER)
    assert n_safe >= 3, "fixture too tie-prone to say anything"
    for i, ok in enumerate(safe_all):
        if ok:
            assert preds[i]["inference"] == ref["predictions"][i]["inference"], f"utterance {i}: HIP harness and oracle disagree on a tie-free decode"
    # the oracle's decoder says what the permutation was built to say, so the corpus WER is the hand-computed 9 / 17 ...
    assert [p["inference"] for p in ref["predictions"]] == [SAYS] * 7 and ref["WER"] == pytest.approx(WANT_WER)
    assert 0.0 < ref["WER"] < 1.0
    if n_safe == 7:   # ... and so is the HIP harness's, word errors and exact matches alike
        assert js[-2]["wer"] == pytest.approx(WANT_WER) and js[-2]["gtms"] == f"{WANT_EXACT}/7"
        assert js[-2]["wer"] == pytest.approx(ref["WER"]) and js[-1]["post_wer"] == pytest.approx(ref["post_ST_wer"])


def test_finetune_harness_writes_reference_checkpoints(tmp_path):
    items = merged_items(6)
    (tmp_path / "train.json").write_text(json.dumps(items))
    (tmp_path / "val.json").write_text(json.dumps(items[:2]))
    ckpt_dir = tmp_path / "checkpoints" / "parity-harness"
    ckpt_dir.mkdir(parents=True)
    cmd = [sys.executable, "-m", "dualhyp_amd.finetune", "--train_path", str(tmp_path / "train.json"), "--val_path", str(tmp_path / "val.json"),
           "--exp_name", "t", "--llm_checkpoint", str(ckpt_dir), "--dual_hypotheses", "--prompts_format", "DualHyp", "--tokenizer", "byte",
           "--random_init", "--batch_size", "2", "--micro_batch_size", "1", "--lr", "1e-3", "--num_epochs", "1", "--save_interval", "4",
           "--out_dir", str(tmp_path / "runs" / "t")]
    out = subprocess.run(cmd, cwd=tmp_path, env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    run = tmp_path / "runs" / "t"
    assert (run / "train.log").is_file() and (run / "lit_model_lora_finetuned.pth").is_file() and (run / "best_model.pth").is_file()
    ck = torch.load(run / "lit_model_lora_finetuned.pth")
    assert set(ck) == {"model"} and "transformer.h.0.attn.attn.lora_A" in ck["model"] and "lm_head.linear.weight" in ck["model"]
    assert float(ck["model"]["transformer.h.0.attn.attn.lora_B"].float().abs().sum()) > 0      # B starts at 0: it was trained
    assert "optimizer_steps': 3" in (run / "train.log").read_text()


def _launch(n, out_path):
    # --standalone: the launcher picks and holds its own rendezvous port on 127.0.0.1 (no fixed-port collisions)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", str(ROOT / "tests" / "dp_worker.py"), "--out", str(out_path)]
    out = subprocess.run(cmd, cwd=ROOT, env=_env(DUALHYP_DP_REHEARSAL="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return torch.load(out_path)


def test_data_parallel_equivalence(tmp_path):
    """the same global batch whether 1 rank accumulates 4 micro-batches or 2 ranks accumulate 2 each and
    all-reduce the flat LoRA-gradient bucket; utterance-sharded inference equals the single-process run."""
    one = _launch(1, tmp_path / "w1.pt")
    two = _launch(2, tmp_path / "w2.pt")
    assert one["world"] == 1 and two["world"] == 2
    assert one["stats"]["optimizer_steps"] == two["stats"]["optimizer_steps"] == 4
    worst = 0.0
    for k, a in one["lora"].items():
        b = two["lora"][k]
        worst = max(worst, (a - b).abs().max().item() / max(a.abs().max().item(), 1e-12))
    from conftest import record_parity
    record_parity("dp_equivalence.fit_world2_vs_world1", worst_rel_diff_lora_masters=worst, optimizer_steps=4)
    assert worst <= 2e-3, f"LoRA masters of the 2-rank run differ from the 1-rank run by {worst:.2e} of their max"
    for k in ("WER", "gtms", "post_ST_wer", "post_gtms", "n"):
        assert one["inference"][k] == two["inference"][k], k
    assert one["inference"]["predictions"] == two["inference"]["predictions"] and len(one["inference"]["predictions"]) == 7


def test_finetune_bench_two_rank_rehearsal():
    """BASELINE config 3's bench line through the real launcher path with TWO ranks (both on cuda:0, gloo: gpurun boxes have
    one GPU) — the only bench where a collective sits inside the timed loop: `bench.py --config finetune-tinyllama --gpus 2`
    self-launches two fresh rank processes, every rank runs 16 packed micro-batches per optimizer step, the flat LoRA-gradient
    bucket is all-reduced, AdamW steps.  The global batch's mean loss after two optimizer steps must equal the 1-rank run's
    (same 32 utterances per step, same averaged gradient): the multi-rank bench really optimises what it describes."""
    import json

    def run(n):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "finetune-tinyllama", "--gpus", str(n), "--steps", "2",
                              "--warmup", "1", "--pack", "8"], cwd=ROOT, env=_env(DUALHYP_BENCH_REHEARSAL="1"), capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])
    one, two = run(1), run(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["micro_batches_per_rank_per_step"] == 16 and two["config"]["micro_batches_per_packed_launch"] == 8
    from conftest import record_parity
    record_parity("bench_finetune.two_rank_rehearsal", mean_loss_1rank=one["mean_loss_last_step"], mean_loss_2ranks=two["mean_loss_last_step"],
                  utt_per_s_1rank=one["value"], utt_per_s_2ranks_one_gpu=two["value"])
    assert abs(one["mean_loss_last_step"] - two["mean_loss_last_step"]) <= 2e-3 * abs(one["mean_loss_last_step"])


def test_relprompt_finetune_harness_trains_the_classifiers(tmp_path):
    """`python -m dualhyp_amd.finetune --prompts_format RelPrompt` (finetune/relprompt.py's flags): the RelPrompt decoder with its
    three reliability tokens, the two NoiseMaskClassifiers trained in the second AdamW group on encoder features read from
    --enc_features_dir (the encoders themselves are upstream of this path), targets = the chunk labels of the items' corruption
    fields; the checkpoint carries trained classifier weights and the grown embedding."""
    items = merged_items(4)
    (tmp_path / "train.json").write_text(json.dumps(items))
    ckpt_dir = tmp_path / "checkpoints" / "parity-harness"
    ckpt_dir.mkdir(parents=True)
    feats = tmp_path / "feats"
    feats.mkdir()
    g = torch.Generator().manual_seed(0)
    for it in items:       # 2 s of audio at 50 fps / video at 25 fps: five 0.4-s chunks each
        torch.save({"audio": torch.randn(100, 1280, generator=g), "visual": torch.randn(50, 1024, generator=g)}, feats / f"{it['Uid']}.pt")
    run = tmp_path / "runs" / "rp"
    cmd = [sys.executable, "-m", "dualhyp_amd.finetune", "--train_path", str(tmp_path / "train.json"), "--exp_name", "rp",
           "--llm_checkpoint", str(ckpt_dir), "--prompts_format", "RelPrompt", "--tokenizer", "byte", "--random_init", "--batch_size", "2",
           "--micro_batch_size", "1", "--lr", "1e-3", "--classifier_lr", "5e-3", "--mask_loss_weight", "0.02", "--num_epochs", "1",
           "--enc_features_dir", str(feats), "--out_dir", str(run)]
    out = subprocess.run(cmd, cwd=tmp_path, env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    ck = torch.load(run / "lit_model_lora_finetuned.pth")["model"]
    from dualhyp_amd import Config
    cfg = Config.from_name("parity-harness")
    assert ck["transformer.wte.weight"].size(0) == cfg.padded_vocab_size + 3           # <<C>> / <<M>> / <<N>> rows
    for k in ("audio_noise_classifier.conv1.weight", "visual_noise_classifier.classifier.bias"):
        assert k in ck and ck[k].dtype == torch.float32                                  # fp32 masters, trained
    assert "optimizer_steps': 2" in (run / "train.log").read_text()
    # the classifier really moved: a fresh RelGPT with the same seed has other weights
    import random
    from dualhyp_amd.relprompt import GPT as RelGPT
    torch.manual_seed(1337); random.seed(1337)
    fresh = RelGPT(cfg)
    assert not torch.equal(fresh.audio_noise_classifier.classifier.bias.detach().float(), ck["audio_noise_classifier.classifier.bias"].float())


def test_relprompt_inference_predicts_the_masks(tmp_path):
    """inference/relprompt.py:113-153: with encoder features at hand the prompt's reliability tokens come from the classifiers
    (arg-max per chunk), the prompt is re-encoded with them, and the predictions are scored against the chunk labels of the
    corruption records.  Checked against the same steps done by hand on the module API."""
    from dualhyp_amd import Config
    from dualhyp_amd.data import HypothesesDataset
    from dualhyp_amd.relprompt import GPT as RelGPT, predicted_mask_prompt
    from dualhyp_amd.synth import synth_state_dict
    from dualhyp_amd.tokenizer import ByteTokenizer
    items = merged_items(3)
    (tmp_path / "test.json").write_text(json.dumps(items))
    ckpt_dir = tmp_path / "checkpoints" / "parity-harness"
    ckpt_dir.mkdir(parents=True)
    feats = tmp_path / "feats"
    feats.mkdir()
    g = torch.Generator().manual_seed(1)
    for it in items:
        torch.save({"audio": torch.randn(100, 1280, generator=g), "visual": torch.randn(50, 1024, generator=g)}, feats / f"{it['Uid']}.pt")
    # a checkpoint with known classifier weights
    cfg = Config.from_name("parity-harness", r=16, alpha=16, dropout=0.05, to_query=True, to_key=True, to_value=True, to_projection=True)
    torch.manual_seed(7)
    m = RelGPT(cfg)
    m.load_state_dict(synth_state_dict(cfg, seed=31, weight_scale=4.0, embed_scale=64.0, head_tie=1.0), strict=False)
    m.resize_token_embeddings(3)
    run = tmp_path / "runs" / "rp"
    run.mkdir(parents=True)
    torch.save({"model": {k: v.detach() for k, v in m.state_dict().items()}}, run / "best_model.pth")
    cmd = [sys.executable, "-m", "dualhyp_amd.inference", "--test_path", str(tmp_path / "test.json"), "--model_path", str(run / "best_model.pth"),
           "--llm_checkpoint", str(ckpt_dir), "--prompts_format", "RelPrompt", "--tokenizer", "byte", "--max_new_tokens", "6",
           "--decode_batch", "2", "--enc_features_dir", str(feats)]
    out = subprocess.run(cmd, cwd=tmp_path, env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "reliability masks predicted by the classifiers" in out.stdout
    js = json.loads((run / "predictions" / "best_model.json").read_text())
    assert len(js) == 3 + 2
    # by hand: the same model, the dataset with placeholders left in, the classifiers' arg-max tokens
    tok = ByteTokenizer()
    tok.add_reliability_tokens(cfg.padded_vocab_size)
    mg = m.to(device="cuda:0", dtype=torch.bfloat16).eval()
    ds = HypothesesDataset(str(tmp_path / "test.json"), tok, prompts_format="RelPrompt", seed=1337, leave_masks=True,
                           enc_features=lambda s1, s2: tuple(torch.load(feats / f"{s1['Uid']}.pt")[k].float() for k in ("audio", "visual")))
    accs = []
    for i in range(len(ds)):
        ex = ds[i]
        assert "<<<ASR_MASKS>>>" in ex["input_no_response"] and "<<<VSR_MASKS>>>" in ex["input_no_response"]
        prompt, a, v = predicted_mask_prompt(mg, ex["input_no_response"], ex["audio_enc_features"], ex["visual_enc_features"])
        assert "<<<" not in prompt and a.numel() == 5 and v.numel() == 5 and prompt.count("<<") == 10
        accs.append((int((a == ex["audio_mask_targets"][:5]).sum()), int((v == ex["visual_mask_targets"][:5]).sum())))
    want_a, want_v = sum(x for x, _ in accs) / 15, sum(y for _, y in accs) / 15
    import re
    got = re.search(r"'audio_mask_accuracy': ([0-9.]+), 'visual_mask_accuracy': ([0-9.]+)", out.stdout)
    assert got and abs(float(got.group(1)) - want_a) < 1e-9 and abs(float(got.group(2)) - want_v) < 1e-9, (out.stdout[-500:], want_a, want_v)
